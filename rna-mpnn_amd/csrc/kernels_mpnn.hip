// Fused ResMPNN step for gfx950, round-4 form: plain program order per wave, no helper MFMAs, blocks dealt dynamically inside a workgroup.
//
//   DO_EDGE: e <- e + MLP_e(P_e[i] + Q_e[j] + e Wc_e)           (edge update of the previous layer, mpnn.py:229-265)
//   DO_MSG : agg = h + mean_valid MLP_m(P_m[i] + Q_m[j] + e Wc_m)   (message + aggregation + residual, mpnn.py:154-227)
//   EMBED  : (layer 1, with DO_MSG only) e0 = the edge featurisation + embedding MLP of the block (feature.py:386-571), formed in front of the
//            message MLP in the same launch: stored once, consumed from registers (bit-identical to k_edge_embed_bf16 + the plain launch)
//
// Same mathematics, data layouts and weight images as k_mpnn_bf16 (kernels_bf16.hip: transposed per-edge Linears on
// v_mfma_f32_32x32x16_f16, accumulator tile -> operand of the next Linear, un-transposed last message Linear).  What differs:
//   * a wave runs plain program order on ONE accumulator tile - the 8 dependent MFMAs of a chain, then that tile's activation arithmetic -
//     instead of the hand-placed MFMA / VALU interleave over two tiles; the two waves of a SIMD overlap the phases (what that buys and what it
//     does not: tools/ubench/mfma_valu_coexec.hip - the SIMD arbitrates oldest wave first, a stagger or per-phase s_setprio changes nothing, a
//     workgroup barrier per phase is slower);
//   * every per-row constant that k_mpnn_bf16 adds with helper MFMAs (32 of its 160 per block) enters without the matrix pipe: the P row of the
//     residue (first Linears) and the bias (edge second Linear) are the INITIAL VALUE of the accumulator tile, read from LDS while the previous
//     tile's activation arithmetic runs (P rows are staged per block as f32; an absent edge reads a zero row instead, so its hidden activations,
//     its e row and - through the closed form below - its message stay exactly what the branch-free loop relies on); the gathered Q row - f16 in
//     natural channel order: a lane's 16 accumulator registers of a channel block are 16 consecutive channels - is added to the f16-converted
//     pre-activation by two packed adds per quarter tile, the bias of the un-transposed message Linear likewise.  128 MFMAs per block;
//   * the blocks of a workgroup are dealt to its waves through an LDS counter (the older wave of a SIMD runs ~1.4x faster than the younger).
// Measured (C2, one box, tools/ab_mpnn.sh): 155 - 158 us per <edge, message> launch against 161 - 164 us for k_mpnn_bf16.  Timing ablations of
// this kernel (RM_EXP_*, wrong results): no Q gathers 154, no e stores 152, no e loads / stores 145, no activation arithmetic 145, no MFMA 128,
// memory skeleton only 123 - 127, compute only (no e / Q traffic) 135; of the compute-only 135: no LDS weight / init reads 100, no MFMA 85, no
// activation arithmetic 98.  Its issue slots per block - ~1,115 vector instructions x 4.3 cycles, 128 MFMA x 8, 184 LDS reads x 4.5 - are 77 % of
// the compute-only time: the kernel is bound by instruction issue; three waves per SIMD (RM_WAVES=12, <= 168 VGPRs, 9 - 14 spilled) are slower.
// Block = 32 edge slots of one residue (k > 16; slots >= k are padding and stay zero), one wave per block.
#include "kernels_bf16.h"
#include "bf16_dev.h"
#include <cstdlib>

static_assert(RN_E_F16 == 1 && RN_P_F16 == 1, "k_resmpnn works on f16 e and f16 P / Q tables");
// Timing ablations (RM_EXP_*: parts compiled out, WRONG results) and structural variants (RM_WAVES, RM_RING8, RM_EDB, RM_QROLL, RM_NO_EPI_FENCE) are
// compile-time switches behind ONE build flag: tools/build_mpnn_variant.sh passes -DRN_EXPERIMENTS; build() never does.
#if !defined(RN_EXPERIMENTS) && (defined(RM_EXP_NOQ) || defined(RM_EXP_NOESTORE) || defined(RM_EXP_NOELOAD) || defined(RM_EXP_NOMFMA) || defined(RM_EXP_NOGELU) || \
                                 defined(RM_EXP_NOLDS) || defined(RM_EXP_STAGGER) || defined(RM_EXP_PRIO) || defined(RM_WAVES) || defined(RM_RING8) || defined(RM_EDB) || \
                                 defined(RM_QROLL) || defined(RM_NO_EPI_FENCE))
#error "experimental variants of k_resmpnn need -DRN_EXPERIMENTS (tools/build_mpnn_variant.sh)"
#endif

#ifndef RM_WAVES
#define RM_WAVES 8                 // waves per workgroup (one workgroup per CU).  Measured at C2 (tools/ab_mpnn.sh, one box): 8 waves 155 - 158 us per
#endif                             // launch, 12 waves (<= 168 VGPRs: the compiler spills 9 - 14) 176 - 180 us, the round-3 kernel 161 - 164 us
#if RM_WAVES <= 8 && !defined(RM_QROLL)
#define RM_QFULL 1                 // 256 VGPRs per wave: all four channel blocks of a gathered Q row one MLP phase ahead (-2 % against two chains ahead)
#endif
#define RM_LDS_P 131072                              // per wave: [P_e f32 128][P_m f32 128]
#define RM_LDS_BE (RM_LDS_P + RM_WAVES * 1024)       // bias of the edge MLP's second Linear, f32, accumulator order (MpnnWB::b2p)
#define RM_LDS_BM (RM_LDS_BE + 512)                  // bias of the message MLP's second Linear as a splat f16 pair per channel
#define RM_LDS_GB (RM_LDS_BM + 512)                  // GELU(bias) of that Linear as the epilogue computes it (f32)
#define RM_LDS_Z (RM_LDS_GB + 512 + 128)             // 512 zero bytes: accumulator init of absent edges (= 128 mod 256: banks 32.. of the
                                                     // row, which no P / bias read of the same instruction touches)
#define RM_LDS_CTR (RM_LDS_Z + 512)                  // block-dealing counter of the workgroup
#define RM_LDS_B0 (RM_LDS_CTR + 16)                  // EMBED: bias of the edge embedding's first Linear (a b0), natural channel order
#define RM_LDS_BYTES (RM_LDS_B0 + 512)
static_assert(RM_LDS_Z % 256 == 128 && RM_LDS_BYTES <= 160 * 1024, "LDS layout");

__device__ __forceinline__ u32x4* rm_efrag(bf16_t* e, int blk, int lane) { return reinterpret_cast<u32x4*>(e) + (size_t)blk * 512 + lane; }
__device__ __forceinline__ f32x16 rm_ld16(const unsigned char* smem, unsigned off) {      // 4 x ds_read_b128
    const f32x4* p = reinterpret_cast<const f32x4*>(smem + off);
    const f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
    return f32x16{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}
// f32 += f16 * f16 (both halves low / both high): one v_fma_mix_f32
__device__ __forceinline__ float rm_fma_ll(f16x2 a, f16x2 b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float rm_fma_hh(f16x2 a, f16x2 b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f16x4 rm_h4(unsigned w0, unsigned w1) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
    return __builtin_bit_cast(f16x4, u32x2_{w0, w1});
}

#ifdef RM_EXP_NOGELU
#define phi4s(y) (y)
#endif
template <bool DO_EDGE, bool DO_MSG, bool EMBED = false>
__global__ void __launch_bounds__(RM_WAVES * 64, RM_WAVES / 4) k_resmpnn(PackInfo pk, int k, const int* __restrict__ nbr, bf16_t* __restrict__ e,
        const bf16_t* __restrict__ p_e, const bf16_t* __restrict__ q_e, const bf16_t* __restrict__ p_m, const bf16_t* __restrict__ q_m,
        const float* __restrict__ h_res, const bf16_t* __restrict__ img_e_g, const float* __restrict__ b2e,
        const bf16_t* __restrict__ img_m_g, const float* __restrict__ b2m, float* __restrict__ agg,
        const float* __restrict__ geomh = nullptr, const float* __restrict__ ee_b0 = nullptr) {
    static_assert(!EMBED || (!DO_EDGE && DO_MSG), "EMBED: the edge embedding (feature.py:386-571) in front of the message of layer 1");
    extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
    u32x4* img_e = reinterpret_cast<u32x4*>(smem);
    u32x4* img_m = img_e + 4096;
    constexpr int NW = RM_WAVES;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int ntot = pk.cu[pk.B];
    const int nblocks = ntot;                            // one residue per block
    const int zero_row = pk.Nmax;
    const bool slot_ok = r < k;
    const int last_idx = ntot * k - 1;

    // Block -> wave mapping.  XCD-aware (as k_mpnn_bf16): workgroup ids go round-robin to the 8 XCDs; XCD x owns a contiguous eighth of the
    // residues, so the P / Q rows its gathers touch stay in that XCD's L2; workgroup g of the XCD's group owns blocks first + t + i * stride
    // (t < NW).  WITHIN the workgroup those blocks are DEALT DYNAMICALLY through an LDS counter (ordinal n -> t = n % NW, i = n / NW): the SIMD
    // arbitrates oldest wave first, so of two co-resident waves running this stream the older runs ~1.4x faster than the younger
    // (tools/ubench/mfma_valu_coexec.hip: a static equal split waits for the slowest wave, 453 cycles per chain + epilogue per SIMD; with both
    // waves busy to the end it is 390).  Which wave computes a block does not change its result.  A wave claims its next-but-one block at the
    // top of every block and reads the answer at the bottom.
    int blk_end, stride, first;
    if ((gridDim.x & 7) == 0) {
        const int chunk = (nblocks + 7) >> 3, x = blockIdx.x & 7;
        blk_end = min(nblocks, (x + 1) * chunk);
        stride = (gridDim.x >> 3) * NW;
        first = x * chunk + (blockIdx.x >> 3) * NW;
    } else {
        blk_end = nblocks; stride = gridDim.x * NW; first = blockIdx.x * NW;
    }
    int* lds_ctr = reinterpret_cast<int*>(smem + RM_LDS_CTR);
    auto claim_issue = [&]() -> int {                    // lane 0 draws the ordinal ...
        int v = 0;
        if (lane == 0) v = __hip_atomic_fetch_add(lds_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return v;
    };
    auto claim_block = [&](int v) -> int {               // ... and the wave reads it (wave-uniform block id; >= blk_end: none left)
        const int n = __builtin_amdgcn_readfirstlane(v);
        return first + (n % NW) + (n / NW) * stride;
    };
    int blk = first + wave, nxt = blk + stride;          // the first two blocks of a wave are its static ones: ordinals 0 .. 2 NW - 1
    // the first block's HBM requests go out before the weight images are staged (a wave without a block reads block 0 and discards it)
    u32x4 ef[8];
    unsigned pn_e = 0u, pn_m = 0u;
    int jraw = -1;
    if (nblocks > 0) {
        const int b0 = blk < blk_end ? blk : 0;
        const int i0 = b0 * k + r;
        jraw = nbr[i0 > last_idx ? last_idx : i0];
        if (!EMBED) {
#pragma unroll
            for (int s = 0; s < 8; ++s) ef[s] = rm_efrag(e, b0, lane)[64 * s];
        }
        if (DO_EDGE) pn_e = reinterpret_cast<const unsigned*>(p_e + (size_t)b0 * RN_D)[lane];
        if (DO_MSG) pn_m = reinterpret_cast<const unsigned*>(p_m + (size_t)b0 * RN_D)[lane];
    }
    if (DO_EDGE) stage_image_dma<NW * 64>(img_e, reinterpret_cast<const u32x4*>(img_e_g), tid);
    if (EMBED) stage_image_dma<NW * 64, (4 * EMB_KS + 32) * 64>(img_e, reinterpret_cast<const u32x4*>(img_e_g), tid);      // [4 mb][7 s] first Linear | [4 ob][8 ks] second
    if (DO_MSG) stage_image_dma<NW * 64>(img_m, reinterpret_cast<const u32x4*>(img_m_g), tid);
    if (EMBED && tid >= 256 + 64 && tid < 256 + 64 + 128) reinterpret_cast<float*>(smem + RM_LDS_B0)[tid - 320] = kGA * ee_b0[tid - 320];
    if (tid < 128) {
        reinterpret_cast<float*>(smem + RM_LDS_BE)[tid] = (DO_EDGE || EMBED) ? b2e[tid] : 0.f;       // (EMBED: the embedding's second bias, e-fragment row order)
        const float bm = DO_MSG ? b2m[tid] : 0.f;            // (a b2: the images' biases are stored in the scaled activation domain)
        const f16x2 bh = cvt_h2(bm, bm);
        reinterpret_cast<unsigned*>(smem + RM_LDS_BM)[tid] = __builtin_bit_cast(unsigned, bh);
        const f16x2 ph = phi2s(bh);
        reinterpret_cast<float*>(smem + RM_LDS_GB)[tid] = (float)bh[0] * (float)ph[0];      // exactly what a row of zeros contributes to the sum
    } else if (tid < 256) {
        reinterpret_cast<unsigned*>(smem + RM_LDS_Z)[tid - 128] = 0u;
    } else if (tid == 256) {
        *lds_ctr = 2 * NW;
    }
    dma_landed();
    __syncthreads();
    if (blk >= blk_end) return;

    const unsigned off_pe = RM_LDS_P + wave * 1024 + 64 * h, off_pm = off_pe + 512, off_be = RM_LDS_BE + 64 * h;
    float* lds_pw = reinterpret_cast<float*>(smem + RM_LDS_P + wave * 1024);
    auto stage_p = [&]() {                                    // this block's P rows: f16 words as loaded -> f32, natural channel order
        if (DO_EDGE) {
            const f16x2 v = __builtin_bit_cast(f16x2, pn_e);
            *reinterpret_cast<f32x2*>(lds_pw + 2 * lane) = f32x2{(float)v[0], (float)v[1]};
        }
        if (DO_MSG) {
            const f16x2 v = __builtin_bit_cast(f16x2, pn_m);
            *reinterpret_cast<f32x2*>(lds_pw + 128 + 2 * lane) = f32x2{(float)v[0], (float)v[1]};
        }
    };
    // ---- chain sequence of a block: c = 0..3 edge Linear 1, 4..7 edge Linear 2, 8..11 message Linear 1, 12..15 message Linear 2
    constexpr int C_FIRST = DO_EDGE ? 0 : 8, C_LAST = DO_MSG ? 15 : 7;
#ifdef RM_EXP_NOLDS      /* timing ablation: weight fragments are not re-read */
#define RM_FRAG(c, s) (wf[s])
#else
#define RM_FRAG(c, s) (((c) < 8 ? img_e : img_m)[(((c) >> 2) & 1) * 2048 + (((c) & 3) * 8 + (s)) * 64 + lane])
#endif
#define RM_NEXT(c) ((c) == C_LAST ? C_FIRST : (c) + 1)
#define RM_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef RM_NO_EPI_FENCE
#define RM_EFENCE() do { } while (0)
#else
#define RM_EFENCE() RM_FENCE()
#endif
#ifdef RM_RING8
#define RM_RING 8
#else
#define RM_RING 4
#endif
    u32x4 q[8], hb[8], wf[8];
    f32x16 T;
    // the lane's 16 channels of channel block cb of a gathered Q row: 2 x 16 B at channel 32 cb + 16 h (natural order = accumulator order).
    // Requested two chains before the epilogue that adds them, so that at most two channel blocks are in registers at a time.
    auto gather_q = [&](auto cbc, const bf16_t* table, int jj) {
        constexpr int cb = decltype(cbc)::value;
        const int row = jj >= 0 ? (jj > zero_row ? zero_row : jj) : zero_row;
        const u32x4* src = reinterpret_cast<const u32x4*>(table + (size_t)row * RN_D + 32 * cb + 16 * h);
#ifdef RM_EXP_NOQ
        (void)src; q[2 * cb] = u32x4{0u, 0u, 0u, 0u}; q[2 * cb + 1] = u32x4{0u, 0u, 0u, 0u};
#else
        q[2 * cb] = src[0]; q[2 * cb + 1] = src[1];
#endif
    };
    // ---- the 8 MFMAs of chain c on T (initialised by the previous epilogue, or from the literal 0).  Weight fragments: k-steps 0..3 of a
    // chain were requested during the previous chain (into the registers its MFMAs 0..3 had consumed), k-steps 4..7 are requested at the
    // start of the chain itself (four MFMAs = 128+ cycles ahead of their use) - so only 16 fragment registers stay live across an epilogue
    auto chain = [&](auto cc, auto&& extra) {
        constexpr int c = decltype(cc)::value, kind = c >> 2, cn = RM_NEXT(c);
#pragma unroll
        for (int s = RM_RING; s < 8; ++s) wf[s] = RM_FRAG(c, s);
        RM_FENCE();
        static_for<8>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
#ifdef RM_EXP_NOMFMA
            if constexpr (s == 0) T[0] += __uint_as_float(wf[s][0] ^ (kind == 0 || kind == 2 ? ef[s][0] : hb[s][0]));
            else T[s] += __uint_as_float(wf[s][0] ^ (kind == 0 || kind == 2 ? ef[s][0] : hb[s][0]));
#else
            if constexpr (kind == 3) {
                if constexpr (s == 0) T = mfma32h(hb[0], wf[0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
                else T = mfma32h(hb[s], wf[s], T);
            } else if constexpr (kind == 1) T = mfma32h(wf[s], hb[s], T);
            else T = mfma32h(wf[s], ef[s], T);
#endif
            if constexpr (s < RM_RING) wf[s] = RM_FRAG(cn, s);
            extra(sc);
        });
        // (a compiler barrier that owns T: the chain's MFMAs and memory requests stay in front of it, the epilogue behind it - the MFMA and
        // conversion intrinsics have no side effects, so scheduling fences alone let instruction selection deal four chains into four tiles)
        asm volatile("" : "+v"(T) :: "memory");
    };
    auto no_extra = [](auto) {};
#ifdef RM_QFULL       /* all four channel blocks of a Q row one MLP phase ahead (32 registers) instead of two chains ahead (16) */
#define RM_GQ_ROLL(cb, tab, jj) do { } while (0)
#define RM_GQ_FULL(tab, jj) do { gather_q(std::integral_constant<int, 0>{}, tab, jj); gather_q(std::integral_constant<int, 1>{}, tab, jj); gather_q(std::integral_constant<int, 2>{}, tab, jj); gather_q(std::integral_constant<int, 3>{}, tab, jj); } while (0)
#else
#define RM_GQ_ROLL(cb, tab, jj) gather_q(cb, tab, jj)
#define RM_GQ_FULL(tab, jj) do { } while (0)
#endif
    // ---- epilogues.  All four quarter tiles are converted first (T is dead after 8 instructions), then the accumulator init of the NEXT chain
    // is requested into T (next_init: LDS byte offset of this lane's 64 bytes, or < 0: none) and lands under the activation arithmetic.
    auto cvt_tile = [&](f16x4 (&x)[4], int next_init) {
#pragma unroll
        for (int v = 0; v < 4; ++v) x[v] = cvt_h4(T[4 * v], T[4 * v + 1], T[4 * v + 2], T[4 * v + 3]);
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) :: "memory");     // T is dead from here on
#ifndef RM_EXP_NOLDS
        if (next_init >= 0) T = rm_ld16(smem, (unsigned)next_init);
#endif
    };
    // first Linears: x = f16(P + e Wc) + Q;  hidden = x Phi(x) -> f16 operand fragments of the second Linear
    auto epi_first = [&](auto cbc, int next_init) {
        constexpr int cb = decltype(cbc)::value;
        f16x4 x[4];
        cvt_tile(x, next_init);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const f16x4 y = x[v] + rm_h4(q[2 * cb + (v >> 1)][2 * (v & 1)], q[2 * cb + (v >> 1)][2 * (v & 1) + 1]);
            const f16x4 g = y * phi4s(y);
            hb[2 * cb + (v >> 1)][2 * (v & 1)] = __builtin_bit_cast(unsigned, lo2(g));
            hb[2 * cb + (v >> 1)][2 * (v & 1) + 1] = __builtin_bit_cast(unsigned, hi2(g));
        }
        RM_EFENCE();
    };
    // edge second Linear (rows in e-fragment order): e <- e + x Phi(x) on the fragment words as loaded, 16-byte stores
    auto epi_edge = [&](auto obc, int next_init, int gblk) {
        constexpr int ob = decltype(obc)::value;
        f16x4 x[4];
        cvt_tile(x, next_init);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const f16x4 ph = phi4s(x[v]);
            const int sp = v >> 1, t = 2 * (v & 1);
            const unsigned o0 = ef[2 * ob + sp][t], o1 = ef[2 * ob + sp][t + 1];
            ef[2 * ob + sp][t] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(lo2(x[v]), lo2(ph), __builtin_bit_cast(f16x2, o0)));
            ef[2 * ob + sp][t + 1] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(hi2(x[v]), hi2(ph), __builtin_bit_cast(f16x2, o1)));
#ifndef RM_EXP_NOESTORE
            if (v & 1) rm_efrag(e, gblk, lane)[64 * (2 * ob + sp)] = ef[2 * ob + sp];
#endif
        }
        RM_EFENCE();
    };
    // message second Linear (un-transposed: the 32 edge slots on the accumulator registers, channel 32 nb + r on the lane): mean over the
    // real edges.  Every row is summed unmasked; a row of an absent edge holds exactly GELU(bias) and is taken out again in closed form.
    auto epi_mean = [&](auto nbc, int next_init, int gblk, float cabs, float inv, float hres) {
        constexpr int nb = decltype(nbc)::value;
        const unsigned bw = reinterpret_cast<const unsigned*>(smem + RM_LDS_BM)[32 * nb + r];
        const float gbv = reinterpret_cast<const float*>(smem + RM_LDS_GB)[32 * nb + r];
        f16x4 x[4];
        cvt_tile(x, next_init);
        const f16x4 bh = rm_h4(bw, bw);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const f16x4 y = x[v] + bh;
            const f16x4 ph = phi4s(y);
            s0 = rm_fma_ll(lo2(y), lo2(ph), s0);
            s1 = rm_fma_hh(lo2(y), lo2(ph), s1);
            s0 = rm_fma_ll(hi2(y), hi2(ph), s0);
            s1 = rm_fma_hh(hi2(y), hi2(ph), s1);
        }
        float sum = s0 + s1;
        sum += __shfl_xor(sum, 32, 64);
        // both lane halves hold the total: the duplicate store of half 1 saves a divergent branch
        agg[(size_t)gblk * RN_D + 32 * nb + r] = fmaf(sum - cabs * gbv, inv, hres);
        RM_EFENCE();
    };

    // EMBED: the neighbour's half-split geometry record (28 floats of the items this lane half owns; kernels_f32.hip: geomh_record)
    f32x4 nrec[7];
    auto load_rec = [&](int jj) {
        const f32x4* src = reinterpret_cast<const f32x4*>(geomh + (size_t)(jj >= 0 ? jj : 0) * RN_GEOMH + 32 * h);
#pragma unroll
        for (int v = 0; v < 7; ++v) nrec[v] = src[v];
    };
    // ---- prologue: state of the first block, fragments and accumulator init of its first chain
    int j = (slot_ok && blk < ntot) ? jraw : -1;
    if constexpr (EMBED) load_rec(j);
    gather_q(std::integral_constant<int, 0>{}, DO_EDGE ? q_e : q_m, j);
    gather_q(std::integral_constant<int, 1>{}, DO_EDGE ? q_e : q_m, j);
#ifdef RM_QFULL
    gather_q(std::integral_constant<int, 2>{}, DO_EDGE ? q_e : q_m, j);
    gather_q(std::integral_constant<int, 3>{}, DO_EDGE ? q_e : q_m, j);
#endif
    stage_p();
#pragma unroll
    for (int s = 0; s < RM_RING; ++s) wf[s] = RM_FRAG(C_FIRST, s);
    if constexpr (!EMBED) T = rm_ld16(smem, j >= 0 ? (DO_EDGE ? off_pe : off_pm) : (unsigned)RM_LDS_Z);
    RM_FENCE();

#ifdef RM_EXP_STAGGER      /* experiment: delay the second-dispatched half of the workgroup (waves that share SIMDs with the first half) */
    if (wave >= NW / 2) { for (int i = 0; i < RM_EXP_STAGGER; ++i) __builtin_amdgcn_s_sleep(4); }
#endif
#ifdef RM_EXP_PRIO
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(RM_EXP_PRIO);
#endif
    while (true) {
        const int nblk = nxt;
        const int claim_v = claim_issue();
        const bool has_next = nblk < blk_end;
        const int nb_c = has_next ? nblk : blk;            // the last iteration re-reads its own block (results unused)
        const int in_ = nb_c * k + r;
        const int jn_raw = nbr[in_ > last_idx ? last_idx : in_];
        if (DO_EDGE) pn_e = reinterpret_cast<const unsigned*>(p_e + (size_t)nb_c * RN_D)[lane];
        if (DO_MSG) pn_m = reinterpret_cast<const unsigned*>(p_m + (size_t)nb_c * RN_D)[lane];
        const bool valid = j >= 0;
        const unsigned vmask = (unsigned)(__ballot(valid) & 0xffffffffull);
        const int cnt = __popc(vmask);
        const float cabs = (float)(32 - cnt);
        const float inv = cnt > 0 ? kGAi * __builtin_amdgcn_rcpf((float)cnt) : 0.f;      // (the summed messages are a m)
        const int a_pe = valid ? (int)off_pe : RM_LDS_Z, a_pm = valid ? (int)off_pm : RM_LDS_Z, a_be = valid ? (int)off_be : RM_LDS_Z;
        const int jn = (slot_ok && nb_c < ntot) ? jn_raw : -1;   // (consumed from the message MLP on: its load has long landed by then)
        const int a_next = jn >= 0 ? (int)(DO_EDGE ? off_pe : off_pm) : RM_LDS_Z;
#ifdef RM_EDB         /* the next block's e fragments a whole block ahead, in registers of their own (+32) */
        u32x4 efn[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) efn[s] = rm_efrag(e, nb_c, lane)[64 * s];
#endif
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        RM_FENCE();

        if constexpr (DO_EDGE) {
            // edge Linear 1: T = P_e + Wc_e . e  (Q_e added in the epilogue)
            chain(std::integral_constant<int, 0>{}, no_extra); RM_GQ_ROLL(I2{}, q_e, j); epi_first(I0{}, a_pe + 128);
            chain(std::integral_constant<int, 1>{}, no_extra); RM_GQ_ROLL(I3{}, q_e, j); epi_first(I1{}, a_pe + 256);
            chain(std::integral_constant<int, 2>{}, no_extra); epi_first(I2{}, a_pe + 384);
            chain(std::integral_constant<int, 3>{}, no_extra); epi_first(I3{}, a_be);
            // edge Linear 2: T = b2 + W2_e . hidden;  e += GELU(T)
            chain(std::integral_constant<int, 4>{}, no_extra); if constexpr (DO_MSG) RM_GQ_FULL(q_m, j); epi_edge(I0{}, a_be + 128, blk);
            chain(std::integral_constant<int, 5>{}, no_extra); epi_edge(I1{}, a_be + 256, blk);
            if constexpr (DO_MSG) {
                chain(std::integral_constant<int, 6>{}, no_extra); RM_GQ_ROLL(I0{}, q_m, j); epi_edge(I2{}, a_be + 384, blk);
                chain(std::integral_constant<int, 7>{}, no_extra); RM_GQ_ROLL(I1{}, q_m, j); epi_edge(I3{}, a_pm, blk);
            } else {
                chain(std::integral_constant<int, 6>{}, no_extra); epi_edge(I2{}, a_be + 384, blk);
                chain(std::integral_constant<int, 7>{}, no_extra);
                // edge update only (taps, stage API): the last epilogue rewrites e, so the next block's state is requested behind it
                stage_p();
                epi_edge(I3{}, a_next, blk);
                RM_GQ_ROLL(I0{}, q_e, jn); RM_GQ_ROLL(I1{}, q_e, jn); RM_GQ_FULL(q_e, jn);
#pragma unroll
                for (int s = 0; s < 8; ++s) ef[s] = rm_efrag(e, nb_c, lane)[64 * s];
            }
        }
        if constexpr (EMBED) {
            // ---- edge featurisation + embedding MLP of THIS block (feature.py:386-571; the arithmetic of k_edge_embed_bf16): e0 is formed in the
            // registers the message MLP reads it from and stored once - the 236 MB re-read of e0 and one launch disappear.  The CENTRAL residue
            // is wave-uniform: its record is read with scalar loads and enters the vector arithmetic as SGPR operands.
            const float* __restrict__ gc = geomh + (size_t)__builtin_amdgcn_readfirstlane(blk) * RN_GEOMH;
            auto catom = [&](int a, int d) { return a < 4 ? gc[3 * a + d] : gc[32 + 3 * (a - 4) + d]; };
            auto cbond = [&](int a, int d) { return a < 3 ? gc[12 + 3 * a + d] : gc[32 + 12 + 3 * (a - 3) + d]; };
            auto cnorm = [&](int a, int d) { return a < 2 ? gc[21 + 3 * a + d] : gc[32 + 21 + 3 * (a - 2) + d]; };
            float nl[28];
#pragma unroll
            for (int v = 0; v < 7; ++v) { nl[4 * v] = nrec[v][0]; nl[4 * v + 1] = nrec[v][1]; nl[4 * v + 2] = nrec[v][2]; nl[4 * v + 3] = nrec[v][3]; }
            float ft[EMB_SLOTS];
#pragma unroll
            for (int a = 0; a < 7; ++a) {                          // 28 distances
                const float ax = catom(a, 0), ay = catom(a, 1), az = catom(a, 2);
#pragma unroll
                for (int bl = 0; bl < 4; ++bl) {
                    const float dx = nl[3 * bl] - ax, dy = nl[3 * bl + 1] - ay, dz = nl[3 * bl + 2] - az;
                    ft[4 * a + bl] = __builtin_amdgcn_sqrtf(fmaf(dz, dz, fmaf(dy, dy, fmaf(dx, dx, kSEPS))));
                }
            }
#pragma unroll
            for (int a = 0; a < 5; ++a) {                          // 15 bond cosines
                const float ax = cbond(a, 0), ay = cbond(a, 1), az = cbond(a, 2);
#pragma unroll
                for (int bl = 0; bl < 3; ++bl) ft[28 + 3 * a + bl] = fmaf(az, nl[12 + 3 * bl + 2], fmaf(ay, nl[12 + 3 * bl + 1], ax * nl[12 + 3 * bl]));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {                          // 8 normal cosines
                const float ax = cnorm(a, 0), ay = cnorm(a, 1), az = cnorm(a, 2);
#pragma unroll
                for (int bl = 0; bl < 2; ++bl) ft[43 + 2 * a + bl] = fmaf(az, nl[21 + 3 * bl + 2], fmaf(ay, nl[21 + 3 * bl + 1], ax * nl[21 + 3 * bl]));
            }
#pragma unroll
            for (int pz = 51; pz < EMB_SLOTS; ++pz) ft[pz] = 0.f;
            u32x4 xf[EMB_KS];
#pragma unroll
            for (int s7 = 0; s7 < EMB_KS; ++s7)
#pragma unroll
                for (int t = 0; t < 4; ++t) xf[s7][t] = p_pack2(ft[8 * s7 + 2 * t], ft[8 * s7 + 2 * t + 1]);
            RM_FENCE();
            const unsigned emask = valid ? 0xffffffffu : 0u;
            static_for<4>([&](auto mbc) {                      // Linear(90, 128): T = a b0 + a W0 . features
                constexpr int mb = decltype(mbc)::value;
                T = rm_ld16(smem, RM_LDS_B0 + 128 * mb + 64 * h);
#pragma unroll
                for (int s7 = 0; s7 < EMB_KS; ++s7) T = mfma32h(img_e[(mb * EMB_KS + s7) * 64 + lane], xf[s7], T);
                asm volatile("" : "+v"(T) :: "memory");
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f16x4 x = cvt_h4(T[4 * v], T[4 * v + 1], T[4 * v + 2], T[4 * v + 3]);
                    const f16x4 g = x * phi4s(x);
                    hb[2 * mb + (v >> 1)][2 * (v & 1)] = __builtin_bit_cast(unsigned, lo2(g));
                    hb[2 * mb + (v >> 1)][2 * (v & 1) + 1] = __builtin_bit_cast(unsigned, hi2(g));
                }
                RM_FENCE();
            });
            static_for<4>([&](auto obc) {                      // Linear(128, 128) -> GELU -> e0 (rows in e-fragment order); absent edges and padding slots: zeros
                constexpr int ob = decltype(obc)::value;
                T = rm_ld16(smem, RM_LDS_BE + 128 * ob + 64 * h);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) T = mfma32h(img_e[(4 * EMB_KS + ob * 8 + ks) * 64 + lane], hb[ks], T);
                asm volatile("" : "+v"(T) :: "memory");
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f16x4 x = cvt_h4(T[4 * v], T[4 * v + 1], T[4 * v + 2], T[4 * v + 3]);
                    const f16x4 g = x * phi4s(x);
                    const int sp = v >> 1, t = 2 * (v & 1);
                    ef[2 * ob + sp][t] = __builtin_bit_cast(unsigned, lo2(g)) & emask;
                    ef[2 * ob + sp][t + 1] = __builtin_bit_cast(unsigned, hi2(g)) & emask;
                    if (v & 1) rm_efrag(e, blk, lane)[64 * (2 * ob + sp)] = ef[2 * ob + sp];
                }
                RM_FENCE();
            });
            T = rm_ld16(smem, (unsigned)a_pm);                 // accumulator init of the message MLP's first chain
            RM_FENCE();
        }
        if constexpr (DO_MSG) {
            // message Linear 1 on the updated e
            chain(std::integral_constant<int, 8>{}, no_extra); RM_GQ_ROLL(I2{}, q_m, j); epi_first(I0{}, a_pm + 128);
            chain(std::integral_constant<int, 9>{}, no_extra); RM_GQ_ROLL(I3{}, q_m, j); epi_first(I1{}, a_pm + 256);
            chain(std::integral_constant<int, 10>{}, no_extra); epi_first(I2{}, a_pm + 384);
            // (k-step s of the last chain is the last reader of ef[s]: the next block's fragment is requested behind it)
            chain(std::integral_constant<int, 11>{}, [&](auto sc) {
#if !defined(RM_EXP_NOELOAD) && !defined(RM_EDB)
                if constexpr (!EMBED) { constexpr int s = decltype(sc)::value; ef[s] = rm_efrag(e, nb_c, lane)[64 * s]; }
#endif
            });
            epi_first(I3{}, -1);
            stage_p();                                     // the next block's P rows (this block's accumulator inits have all been read)
            if constexpr (EMBED) load_rec(jn);             // ... and its neighbour records
            float hres[4] = {0.f, 0.f, 0.f, 0.f};
            if (h_res) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) hres[nb] = h_res[(size_t)blk * RN_D + 32 * nb + r];
            }
            // message Linear 2, un-transposed: T[edge][channel] = hidden . W2_m^T
            chain(std::integral_constant<int, 12>{}, no_extra); RM_GQ_FULL(DO_EDGE ? q_e : q_m, jn); epi_mean(I0{}, -1, blk, cabs, inv, hres[0]);
            chain(std::integral_constant<int, 13>{}, no_extra); epi_mean(I1{}, -1, blk, cabs, inv, hres[1]);
            chain(std::integral_constant<int, 14>{}, no_extra); RM_GQ_ROLL(I0{}, DO_EDGE ? q_e : q_m, jn); epi_mean(I2{}, -1, blk, cabs, inv, hres[2]);
            chain(std::integral_constant<int, 15>{}, no_extra); RM_GQ_ROLL(I1{}, DO_EDGE ? q_e : q_m, jn); epi_mean(I3{}, EMBED ? -1 : a_next, blk, cabs, inv, hres[3]);
        }
#ifdef RM_EDB
#pragma unroll
        for (int s = 0; s < 8; ++s) ef[s] = efn[s];
#endif
        if (!has_next) break;
        blk = nblk;
        nxt = claim_block(claim_v);
        j = jn;
    }
    // The requests for the next block's state (e fragments, Q rows, accumulator init) are unused on the exit path; without a use there the
    // compiler sinks them behind the loop's exit test, i.e. to the END of the iteration - and the prefetch distance is gone.
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" :: "v"(ef[s]));
#pragma unroll
    for (int s = 0; s < 4; ++s) { asm volatile("" :: "v"(q[s])); asm volatile("" :: "v"(wf[s])); }
    if (RM_RING == 8) { for (int s = 4; s < 8; ++s) asm volatile("" :: "v"(wf[s])); }
    asm volatile("" :: "v"(T));
    if constexpr (EMBED) {
#pragma unroll
        for (int v = 0; v < 7; ++v) asm volatile("" :: "v"(nrec[v]));
    }
#undef RM_GQ_ROLL
#undef RM_GQ_FULL
#undef RM_EFENCE
#undef RM_FENCE
#undef RM_NEXT
#undef RM_FRAG
}

static bool rm_enabled() { const char* v = getenv("RNAMPNN_MPNN_V3"); return !(v && v[0] == '1'); }      // RNAMPNN_MPNN_V3=1: the round-3 kernel (A/B; read per call)
bool resmpnn_covers(int k, bool edge1, bool msg_out) { return k > 16 && k <= 32 && !edge1 && !msg_out && rm_enabled(); }

void launch_resmpnn_bf16(const PackInfo& pk, int k, bool do_edge, bool do_msg, const int* nbr, bf16_t* e, const bf16_t* p_e, const bf16_t* q_e,
                         const bf16_t* p_m, const bf16_t* q_m, MpnnWB we, MpnnWB wm, float* agg, const float* h_res, hipStream_t s) {
    const int max_blocks = pk.Nmax;
    int grid = (max_blocks + RM_WAVES - 1) / RM_WAVES;
    if (grid >= 8) grid = (grid + 7) & ~7;            // a multiple of 8 switches the kernel to its XCD-aware block mapping
    const int cus = rn_num_cus();
    if (grid > cus) grid = cus;
    if (grid < 1) grid = 1;
#define RM_LAUNCH(E, M)                                                                                        \
    do {                                                                                                       \
        static DevAttr attr;                                                                                   \
        ensure_dyn_lds((const void*)k_resmpnn<E, M>, RM_LDS_BYTES, attr);                                      \
        hipLaunchKernelGGL((k_resmpnn<E, M>), dim3(grid), dim3(RM_WAVES * 64), RM_LDS_BYTES, s, pk, k, nbr, e, p_e, q_e, p_m, q_m, h_res, \
                           we.img, we.b2p, wm.img, wm.b2p, agg);                                               \
    } while (0)
    if (do_edge && do_msg) RM_LAUNCH(true, true);
    else if (do_edge) RM_LAUNCH(true, false);
    else RM_LAUNCH(false, true);
#undef RM_LAUNCH
}

// Layer 1's message launch with the edge embedding in front of it: e0 is computed from the geometry records, stored once and consumed from registers.
void launch_resmpnn_embed_bf16(const PackInfo& pk, int k, const int* nbr, bf16_t* e, const float* geomh, const bf16_t* ee_img, const float* ee_b0,
                               const float* ee_b1p, const bf16_t* p_m, const bf16_t* q_m, MpnnWB wm, float* agg, const float* h_res, hipStream_t s) {
    int grid = (pk.Nmax + RM_WAVES - 1) / RM_WAVES;
    if (grid >= 8) grid = (grid + 7) & ~7;
    const int cus = rn_num_cus();
    if (grid > cus) grid = cus;
    if (grid < 1) grid = 1;
    static DevAttr attr;
    ensure_dyn_lds((const void*)k_resmpnn<false, true, true>, RM_LDS_BYTES, attr);
    hipLaunchKernelGGL((k_resmpnn<false, true, true>), dim3(grid), dim3(RM_WAVES * 64), RM_LDS_BYTES, s, pk, k, nbr, e, (const bf16_t*)nullptr,
                       (const bf16_t*)nullptr, p_m, q_m, h_res, ee_img, ee_b1p, wm.img, wm.b2p, agg, geomh, ee_b0);
}
