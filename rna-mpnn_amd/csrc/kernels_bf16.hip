// bf16 MFMA fast path for gfx950 (CDNA4, wave64) - see kernels_bf16.h.
//
// The per-edge MLPs are evaluated "transposed": D[channel][edge] = W . X^T with
// v_mfma_f32_32x32x16_bf16, 32 edges on the MFMA columns (= lanes), output channels on the rows
// (= accumulator registers).  Consequences this file is built around:
//   * the B operand of the first Linear is a row-major e row: lane (edge r, half h) loads
//     e[edge][16s + 8h .. +8] (16 B) for k-step s straight from HBM - no transpose, no LDS;
//   * the accumulator tile of one Linear is, after GELU and a pairwise bf16 convert, directly
//     the B operand (or A operand) of the next Linear - the MLP chain never leaves registers
//     (the k order inside a step is permuted, so the next weight fragment is built with the same
//     permutation at weight-load time);
//   * the output-row order of a Linear is free (it is a row permutation of its weight), and is
//     chosen so that each lane ends up holding exactly the e channels it loaded: the residual
//     add and the 16-byte stores of e' need no lane movement;
//   * the LAST Linear of the message MLP runs un-transposed (A = activations from registers,
//     B = weights), which puts the k edges of a residue on the accumulator registers: the masked
//     mean over the neighbourhood is 15 in-lane adds + one cross-half shuffle.
// Weights live in LDS as ready-made 1 KiB MFMA fragments (lane-linear, conflict-free
// ds_read_b128); a 512-thread workgroup per CU loads them once and then walks 32-edge blocks.
#include "kernels_bf16.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "bf16_dev.h"

// ------------------------------------------------------------------------------------------
// weight preparation
__global__ void k_convert_rows_bf16(const float* __restrict__ src, int ld_src, int rows, int cols, int cols_pad,
                                    bf16_t* __restrict__ dst) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * cols_pad) return;
    int r = id / cols_pad, c = id - r * cols_pad;
    dst[id] = c < cols ? f2bf(src[(size_t)r * ld_src + c]) : (bf16_t)0;
}
void launch_convert_rows_bf16(const float* src, int ld_src, int rows, int cols, int cols_pad, bf16_t* dst, hipStream_t s) {
    int total = rows * cols_pad;
    hipLaunchKernelGGL(k_convert_rows_bf16, dim3((total + 255) / 256), dim3(256), 0, s, src, ld_src, rows, cols, cols_pad, dst);
}

// Fragment image of one 2-Linear MLP acting on e (see file head).
//   image 0 [mb][s][lane][8]:  A fragments of Linear 0's e-part, rows in ch_nat order
//   image 1 [ob][ks][lane][8]: fragments of Linear 1; k-step ks = 2*mb + s' carries channels
//       32mb + 16h + 8s' + j (the accumulator-as-operand order); rows (edge MLP, out_perm=1):
//       ch_efrag order, bias permuted alike; columns (message MLP, out_perm=0): natural.
//   depth-1 MLP (w2 == null; edge update only: e += GELU(P + Q + e Wc)): image 0 with its rows in ch_efrag order, so the
//   accumulator tile of the ONE Linear lines up with the e fragments of the residual add; image 1 zero.
__global__ void k_build_mlp_image(const float* __restrict__ wc, int ld_wc, const float* __restrict__ w2, int ld_w2,
                                  const float* __restrict__ b2, int out_perm, bf16_t* __restrict__ img, float* __restrict__ b2p) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;      // one element of one fragment
    if (!w2) {
        if (id < 2 * 32 * 64 * 8) {
            int j = id & 7, lane = (id >> 3) & 63, f = (id >> 9) & 31, which = id >> 14;
            int r = lane & 31, h = lane >> 5, mb = f >> 3, s = f & 7;
            img[id] = which == 0 ? e_enc(wc[(size_t)ch_efrag(mb, r) * ld_wc + 16 * s + 8 * h + j]) : (bf16_t)0;
        }
        if (id < 128) b2p[id] = 0.f;
        return;
    }
    if (id < 2 * 32 * 64 * 8) {
        int j = id & 7, lane = (id >> 3) & 63, f = (id >> 9) & 31, which = id >> 14;
        int r = lane & 31, h = lane >> 5;
        float v;
        if (which == 0) {
            int mb = f >> 3, s = f & 7;
            v = wc[(size_t)ch_nat(mb, r) * ld_wc + 16 * s + 8 * h + j];
        } else {
            int ob = f >> 3, ks = f & 7, mb = ks >> 1, sp = ks & 1;
            int row = out_perm ? ch_efrag(ob, r) : 32 * ob + r;
            v = w2[(size_t)row * ld_w2 + 32 * mb + 16 * h + 8 * sp + j];
        }
        // the second Linear consumes the hidden activations, which the fused kernel keeps in f16: f16 operands; the first one consumes e (e_enc)
        img[id] = which == 1 ? __builtin_bit_cast(bf16_t, (_Float16)v) : e_enc(v);
    }
    if (id < 128) {       // (scaled activation domain: the second Linear's output is a x2 = W2 . (y1 Phi) + a b2)
        if (out_perm) {
            int ob = id >> 5, m = id & 31;                // position 32*ob + 16*h + i  <-> row m
            int h = (m >> 2) & 1, i = (m & 3) + 4 * (m >> 3);
            b2p[32 * ob + 16 * h + i] = kGA * b2[ch_efrag(ob, m)];
        } else {
            b2p[id] = kGA * b2[id];
        }
    }
}
void launch_build_mlp_image(const float* wc, int ld_wc, const float* w2, int ld_w2, const float* b2, int out_perm,
                            bf16_t* img, float* b2p, hipStream_t s) {
    int total = 2 * 32 * 64 * 8;
    hipLaunchKernelGGL(k_build_mlp_image, dim3((total + 255) / 256), dim3(256), 0, s, wc, ld_wc, w2, ld_w2, b2, out_perm, img, b2p);
}

__global__ void k_build_embed_image(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ b1,
                                    bf16_t* __restrict__ img, float* __restrict__ b1p) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = 4 * EMB_KS * 64 * 8, n1 = 32 * 64 * 8;
    if (id < n0) {
        int j = id & 7, lane = (id >> 3) & 63, f = id >> 9;
        int mb = f / EMB_KS, s = f % EMB_KS, r = lane & 31, h = lane >> 5;
        int feat = emb_feature_of_slot(h, 8 * s + j);
        img[id] = feat >= 0 ? __builtin_bit_cast(bf16_t, (_Float16)(kGA * w0[(size_t)ch_nat(mb, r) * RN_ERAW + feat])) : (bf16_t)0;     // f16 (scaled domain: y1 = a (W0 f + b0))
    } else if (id < n0 + n1) {
        int e = id - n0;
        int j = e & 7, lane = (e >> 3) & 63, f = e >> 9;
        int ob = f >> 3, ks = f & 7, mb = ks >> 1, sp = ks & 1, r = lane & 31, h = lane >> 5;
        // second Linear: f16 operand (its input, the hidden activation, is produced in f16)
        img[id] = __builtin_bit_cast(bf16_t, (_Float16)w1[(size_t)ch_efrag(ob, r) * RN_D + 32 * mb + 16 * h + 8 * sp + j]);
    }
    if (id < 128) {
        int ob = id >> 5, m = id & 31, h = (m >> 2) & 1, i = (m & 3) + 4 * (m >> 3);
        b1p[32 * ob + 16 * h + i] = kGA * b1[ch_efrag(ob, m)];
    }
}
void launch_build_embed_image(const float* w0, const float* w1, const float* b1, bf16_t* img, float* b1p, hipStream_t s) {
    int total = (4 * EMB_KS + 32) * 64 * 8;
    hipLaunchKernelGGL(k_build_embed_image, dim3((total + 255) / 256), dim3(256), 0, s, w0, w1, b1, img, b1p);
}

// fragment-major bf16 e  <->  row-major f32 rows [(p*k + slot)][128]  (taps and the stage API only)
__global__ void k_efrag_to_rows(const bf16_t* __restrict__ ef, const int* __restrict__ ntot_p, int k, float* __restrict__ rows) {
    const int ntot = *ntot_p;
    const int npb = k > 16 ? 1 : 32 / k;
    const size_t total = (size_t)ntot * k * 16;            // one 8-channel chunk per thread
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(id & 15);
        const size_t row = id >> 4;
        const int p = (int)(row / k), sl = (int)(row - (size_t)p * k);
        const int blk = p / npb, r = (p - blk * npb) * k + sl;
        const int s = c8 >> 1, h = c8 & 1;
        const bf16_t* src = ef + ((size_t)blk * 512 + s * 64 + 32 * h + r) * 8;
        float* dst = rows + row * RN_D + 8 * c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = kGAi * e_dec(src[j]);         // (the tensor is stored as a e)
    }
}
// row-major f32 edge rows -> fragment-major bf16.  Every slot of every block is written: padding slots and absent
// edges (nbr < 0) as zeros - the fused kernel's unmasked loads / stores rely on that.
__global__ void k_rows_to_efrag(const float* __restrict__ rows, const int* __restrict__ ntot_p, int k, const int* __restrict__ nbr,
                                bf16_t* __restrict__ ef) {
    const int ntot = *ntot_p;
    const int npb = k > 16 ? 1 : 32 / k;
    const int nblocks = (ntot + npb - 1) / npb;
    const size_t total = (size_t)nblocks * 32 * 16;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(id & 15);
        const int r = (int)((id >> 4) & 31);
        const int blk = (int)(id >> 9);
        const int q = r / k, p = blk * npb + q, sl = r - q * k;
        const size_t row = (size_t)p * k + sl;
        const bool real = q < npb && p < ntot && nbr[row] >= 0;
        const int s = c8 >> 1, h = c8 & 1;
        bf16_t* dst = ef + ((size_t)blk * 512 + s * 64 + 32 * h + r) * 8;
        const float* src = rows + row * RN_D + 8 * c8;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = real ? e_enc(kGA * src[j]) : (bf16_t)0;
    }
}
static unsigned conv_grid(size_t max_elems) { size_t g = (max_elems + 255) / 256; return (unsigned)(g < 8192 ? (g ? g : 1) : 8192); }
void launch_efrag_to_rows(const bf16_t* ef, const int* ntot, int nmax, int k, float* rows, hipStream_t s) {
    hipLaunchKernelGGL(k_efrag_to_rows, dim3(conv_grid((size_t)nmax * k * 16)), dim3(256), 0, s, ef, ntot, k, rows);
}
void launch_rows_to_efrag(const float* rows, const int* ntot, int nmax, int k, const int* nbr, bf16_t* ef, hipStream_t s) {
    hipLaunchKernelGGL(k_rows_to_efrag, dim3(conv_grid((size_t)nmax * 32 * 16)), dim3(256), 0, s, rows, ntot, k, nbr, ef);
}

// ------------------------------------------------------------------------------------------
// shared pieces of the two edge kernels
struct BlockLane {            // what one lane knows about its edge in the current 32-edge block
    int node;                 // packed residue of this lane's edge (valid when ok)
    int j;                    // packed neighbour row (or -1)
    bool ok;                  // lane maps to an existing (residue, slot)
    bool valid;               // ... and the slot holds an edge
    size_t erow;              // row of nbr (and of the row-major e of the API / f32 path)
};
// The bf16 edge tensor lives in HBM in FRAGMENT-MAJOR order: block b (32 edge slots) is 8 KiB =
// [k-step s 0..7][lane 0..63][8 bf16], i.e. exactly the register image of the B fragments, so every
// load/store instruction of the edge kernels moves one fully contiguous 1 KiB.
__device__ __forceinline__ u32x4* efrag_ptr(bf16_t* e, int blk, int lane) {
    return reinterpret_cast<u32x4*>(e) + (size_t)blk * 512 + lane;
}

__device__ __forceinline__ BlockLane block_lane(int blk, int npb, int k, int ntot, int r, const int* __restrict__ nbr) {
    BlockLane b;
    int q = r / k;
    int node0 = blk * npb;
    b.node = node0 + q;
    b.ok = q < npb && b.node < ntot;
    b.erow = (size_t)node0 * k + r;
    b.j = b.ok ? nbr[b.erow] : -1;
    b.valid = b.j >= 0;
    return b;
}

__device__ __forceinline__ f32x16 init_vec16(const float* __restrict__ p) {
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        f32x4 a = *reinterpret_cast<const f32x4*>(p + 4 * v);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * v + t] = a[t];
    }
    return acc;
}

// GELU on one 32-channel accumulator block and repack as the two bf16 k-step fragments it feeds
__device__ __forceinline__ void gelu_pack(const f32x16& acc, u32x4& lo, u32x4& hi) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        lo[t] = pack2(gelu_fast(acc[2 * t]), gelu_fast(acc[2 * t + 1]));
        hi[t] = pack2(gelu_fast(acc[8 + 2 * t]), gelu_fast(acc[8 + 2 * t + 1]));
    }
}

// ------------------------------------------------------------------------------------------
// Fused ResMPNN step on 32-edge blocks (mpnn.py:154-265), bf16 MFMA:
//   DO_EDGE: e <- e + MLP_e(P_e[i] + Q_e[j] + e Wc_e)   (edge update of the previous layer)
//   DO_MSG : agg = mean_valid MLP_m(P_m[i] + Q_m[j] + e Wc_m)   (message + aggregation; the residual
//            h + agg of mpnn.py:222 is taken by the graph-norm kernel that follows)
// One wave owns one block: lanes (r, h) = (edge r of the block, k-half h).  k <= 16 packs
// npb = 32/k residues into a block (edges of consecutive residues are contiguous in e).
struct NodeTabs {             // per-residue parts of the first Linears (k_node_update outputs): f16 of a P / a Q, natural channel order
    const bf16_t* p_e;        // [N+1][128] f16  P = h.Wa_e^T + b1_e
    const bf16_t* q_e;        // [N+1][128] f16  h.Wb_e^T         (row N = zeros)
    const bf16_t* p_m;
    const bf16_t* q_m;
    const float* h_res;       // [N][128] f32 or null: the residue's h, added to the mean so that the launch writes h + agg (mpnn.py:222): the
                              // statistics and update kernels behind it then read ONE node tensor instead of two
};

#ifndef RN_MPNN_WAVES
#define RN_MPNN_WAVES 8           // waves per workgroup (one workgroup per CU): 2 per SIMD, <= 256 VGPRs each
#endif
#define RN_MPNN_LDS (131072 + RN_MPNN_WAVES * 1536 + 1024 + 1024 + 512 + 2048 + 2048)
// Execution shape.  A block runs 16 chains - 4 channel blocks x {edge Linear 1, edge Linear 2, message Linear 1,
// message Linear 2} - of 9-11 dependent MFMAs on a 32x32 accumulator tile, each followed by the activation
// arithmetic of that tile ("epilogue": packed-f16 VALU).  A wave issues in order, so the two only overlap if they are
// interleaved in program order: slot c of the loop issues chain c into one tile while the epilogue of chain c-1 runs
// out of the other tile, one quarter tile after every second MFMA (an MFMA holds the matrix pipe for 32 cycles and
// the vector issue for 8 of them; the ~20 VALU instructions between two MFMAs ride in that shadow).  The slot order
// is pinned with scheduling fences.  Data dependencies between chains exist only where one Linear feeds the next
// (hb, the updated e): there the consumer's k-steps 6 and 7 - the ones the last epilogue quarter produces - come
// last in the chain.  The chain of the NEXT block's first channel block runs against the last epilogue of this one.
//   No operand of a chain waits on LDS or HBM at issue: weight fragment s of the next chain is requested right
// after MFMA s of this one has consumed its register (wf ring), every accumulator starts from the literal 0 - the
// P row of the residue (first Linears) and the bias (second Linears) enter as a k = 2 MFMA of (hi, lo) bf16 words
// against a ones column of the REAL edges (so an absent edge keeps a zero accumulator everywhere: its hidden
// activations are 0, its e row is rewritten unchanged, its message row is GELU(bias)) - and the e fragments and
// Q rows of the next block are requested during the last four chains of this one.
// EDGE1: the edge MLP has ONE Linear (num_mpnn_edge_layers = 1, the reference's recorded alternative configuration, train.py:9-43):
// chains 0..3 carry the residual epilogue themselves (their rows are in e-fragment order: image, P words and routing fragments are
// built for that), chains 4..7 do not exist.
template <bool DO_EDGE, bool DO_MSG, bool SMALLK, bool MSGOUT, bool EDGE1 = false>
__global__ void __launch_bounds__(RN_MPNN_WAVES * 64, RN_MPNN_WAVES / 4) k_mpnn_bf16(PackInfo pk, int k, const int* __restrict__ nbr, bf16_t* __restrict__ e,
        NodeTabs tab, MpnnWB we, MpnnWB wm, float* __restrict__ agg, float* __restrict__ msg_out) {
    // LDS: [img_e 64 KiB][img_m 64 KiB][per wave: P_e | P_m words | h row, 1.5 KiB][bias words edge 1 KiB][bias words msg 1 KiB][GELU(bias) msg 512 B][routing 2 KiB][routing, e-fragment row order (EDGE1) 2 KiB]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* img_e = reinterpret_cast<u32x4*>(smem);
    u32x4* img_m = img_e + 4096;
    constexpr int NW = RN_MPNN_WAVES;
    const int tid = threadIdx.x;
    unsigned* lds_p = reinterpret_cast<unsigned*>(smem + 131072) + (tid >> 6) * 384;
    unsigned* lds_bwe = reinterpret_cast<unsigned*>(smem + 131072) + NW * 384;  // [ob][lane]: (hi, lo) bf16 split of the bias of accumulator row lane&31
    unsigned* lds_bwm = lds_bwe + 256;                                          // [nb][lane]: same for output channel 32nb + (lane&31)
    float* lds_gb = reinterpret_cast<float*>(lds_bwm + 256);                    // [nb][r]: GELU(bias), as the epilogue computes it
    u32x4* lds_perm = reinterpret_cast<u32x4*>(lds_gb + 128);                   // [2][lane]: constant 0/1 routing fragments
    const int ntot = pk.cu[pk.B];
    const int npb = SMALLK ? 32 / k : 1;              // SMALLK <=> k <= 16
    const int nblocks = (ntot + npb - 1) / npb;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int zero_row = pk.Nmax;
    const int q0 = SMALLK ? r / k : 0;                // residue of the block this lane's edge slot belongs to
    const bool slot_ok = SMALLK ? q0 < npb : r < k;
    const int last_idx = ntot * k - 1;
    const unsigned ones_w = h == 0 ? 0x3F803F80u : 0u;     // k = 0, 1 of lane half 0

    // XCD-aware block mapping: consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2.  XCD x
    // owns a CONTIGUOUS eighth of the residues (whole RNAs, mostly), so the Q / P rows its gathers touch (2 MB instead of
    // the full 16 MB tables) stay resident in that L2 next to the streaming e blocks; within the eighth the waves stride.
    int blk, blk_end, stride;
    if ((gridDim.x & 7) == 0) {
        const int chunk = (nblocks + 7) >> 3, x = blockIdx.x & 7;
        blk_end = min(nblocks, (x + 1) * chunk);
        stride = (gridDim.x >> 3) * NW;
        blk = x * chunk + (blockIdx.x >> 3) * NW + wave;
    } else {
        blk_end = nblocks; stride = gridDim.x * NW; blk = blockIdx.x * NW + wave;
    }
    // The first block's HBM requests (neighbour indices, e fragments, P words) go out BEFORE the weight images are staged
    // into LDS, so that their latency passes under the staging instead of after it (a wave without a block reads block 0
    // and discards it; it must stay for the cooperative staging).
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    u32x4 ef[8];
    u32x2 pn_e = {0u, 0u}, pn_m = {0u, 0u};
    f32x2 hn = {0.f, 0.f};                            // the block's h row (two channels per lane), staged like the P words
    int jraw_first = -1;
    if (nblocks > 0) {
        const int b0 = blk < blk_end ? blk : 0;
        const int i0 = b0 * npb * k + r;
        jraw_first = nbr[i0 > last_idx ? last_idx : i0];
#pragma unroll
        for (int s = 0; s < 8; ++s) ef[s] = efrag_ptr(e, b0, lane)[64 * s];
        if (!SMALLK) {
            if (DO_EDGE) pn_e[0] = reinterpret_cast<const unsigned*>(tab.p_e + (size_t)b0 * RN_D)[lane];
            if (DO_MSG) pn_m[0] = reinterpret_cast<const unsigned*>(tab.p_m + (size_t)b0 * RN_D)[lane];
            if (DO_MSG && tab.h_res) hn = *reinterpret_cast<const f32x2*>(tab.h_res + (size_t)b0 * RN_D + 2 * lane);
        }
    }
    if (DO_EDGE) stage_image_dma<NW * 64>(img_e, reinterpret_cast<const u32x4*>(we.img), tid);
    if (DO_MSG) stage_image_dma<NW * 64>(img_m, reinterpret_cast<const u32x4*>(wm.img), tid);
    if (tid < 128) {
        // The gathered Q row (bf16, fetched in the e-fragment layout: q[s] = channels 16s+8h..) is added by the MATRIX
        // pipe: two extra MFMAs per channel block whose A operand is the constant 0/1 matrix that routes channel
        // 16s+8h+j to the accumulator row holding it (identical for every block; exact: x 1.0, f32 accumulate).
        const int sp = tid >> 6, rr = tid & 31, hh = (tid >> 5) & 1;
        const int c16 = (rr & 3) + 4 * (rr >> 3), sp_r = (rr >> 2) & 1, jstar = c16 - 8 * hh;  // element jstar of this lane's k-half
        u32x4 pv;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            pv[t] = (sp == sp_r && jstar == 2 * t ? 0x3C00u : 0u) | (sp == sp_r && jstar == 2 * t + 1 ? 0x3C000000u : 0u);      // f16 ones
        lds_perm[tid] = pv;
        if (EDGE1) {
            // rows in ch_efrag order: row (h' = sp_r, i = c16) holds channel 16 (i >> 3) + 8 h' + (i & 7) of its 32-block
            const bool hit = sp == (c16 >> 3) && hh == sp_r;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                pv[t] = (hit && (c16 & 7) == 2 * t ? 0x3C00u : 0u) | (hit && (c16 & 7) == 2 * t + 1 ? 0x3C000000u : 0u);
            lds_perm[128 + tid] = pv;
        }
    }
    if (tid < 256) {
        const int blk4 = tid >> 6, rr = tid & 31, hh = (tid >> 5) & 1;
        const float bm = DO_MSG ? wm.b2p[32 * blk4 + rr] : 0.f;
        const float be = DO_EDGE ? we.b2p[32 * blk4 + 16 * ((rr >> 2) & 1) + (rr & 3) + 4 * (rr >> 3)] : 0.f;
        const unsigned mw = split_word(bm), ew = split_word(be);
        lds_bwm[tid] = hh == 0 ? mw : 0u;
        lds_bwe[tid] = hh == 0 ? ew : 0u;
        if (hh == 0) {
            const float bs = __uint_as_float(mw << 16) + __uint_as_float(mw & 0xffff0000u);
            lds_gb[32 * blk4 + rr] = bs * (float)phi2s(cvt_h2(bs, bs))[0];      // (bs = a b2: wm.b2p is stored scaled)
        }
    }
    dma_landed();          // (the first block's e / P / index loads issued above have landed too: they are needed right after the barrier)
    __syncthreads();

    const u32x4 perm0 = lds_perm[lane], perm1 = lds_perm[64 + lane];
    const u32x4 perm0e = EDGE1 ? lds_perm[128 + lane] : perm0, perm1e = EDGE1 ? lds_perm[192 + lane] : perm1;   // routing of the edge MLP's Q
    const u32x4 ones_a = {ones_w, 0u, 0u, 0u};
    unsigned bwn = 0u;                                 // bias word of the NEXT chain (second Linears), requested mid-chain
    float gbv = 0.f;                                   // GELU(bias) of the channel whose mean is being formed
    if (blk >= blk_end) return;

    // chain sequence of a block: c = 0..3 edge Linear 1, 4..7 edge Linear 2, 8..11 message Linear 1, 12..15 message Linear 2
    constexpr int C_FIRST = DO_EDGE ? 0 : 8, C_LAST = DO_MSG ? 15 : (EDGE1 ? 3 : 7);
#define RN_FRAG(c, s) (((c) < 8 ? img_e : img_m)[(((c) >> 2) & 1) * 2048 + (((c) & 3) * 8 + (s)) * 64 + lane])
#define RN_NEXT(c) ((c) == C_LAST ? C_FIRST : (EDGE1 && (c) == 3) ? 8 : (c) + 1)
#define RN_IDX(b) ({ int i_ = (b) * npb * k + r; i_ > last_idx ? last_idx : i_; })
#define RN_QROW(jj) ((jj) >= 0 ? ((jj) > zero_row ? zero_row : (jj)) : zero_row)
#define RN_FENCE() __builtin_amdgcn_sched_barrier(0)

    u32x4 q[8], hb[8], wf[8];                         // q: gathered Q rows of the MLP whose first Linear runs next
    f32x16 tA, tB;                                     // accumulator tiles: chain c runs in (c & 1 ? tB : tA)
    u32x4 pwe = {0u, 0u, 0u, 0u}, pwm = {0u, 0u, 0u, 0u};      // !SMALLK: P words of the block's residue, [mb]
    int j;                                             // packed neighbour row of this lane's edge, -1: no edge
    // ---- block state: everything a block needs from HBM (addresses clamped, loads unconditional)
    auto load_e = [&](int b, int s) {
        ef[s] = efrag_ptr(e, b, lane)[64 * s];
    };
    auto gather_q = [&](u32x4 (&dst)[8], const bf16_t* table, int row, int s) {
        dst[s] = (reinterpret_cast<const u32x4*>(table + (size_t)row * RN_D) + h)[2 * s];
    };
    auto load_p = [&](int b) {                         // !SMALLK: the residue's P words (coalesced 512 B rows), requested early ...
        if (SMALLK) return;
        if (DO_EDGE) pn_e[0] = reinterpret_cast<const unsigned*>(tab.p_e + (size_t)b * RN_D)[lane];
        if (DO_MSG) pn_m[0] = reinterpret_cast<const unsigned*>(tab.p_m + (size_t)b * RN_D)[lane];
        if (DO_MSG && tab.h_res) hn = *reinterpret_cast<const f32x2*>(tab.h_res + (size_t)b * RN_D + 2 * lane);
    };
    auto stage_p = [&]() {                             // ... through the wave's LDS slot ...
        if (SMALLK) return;
        if (DO_EDGE) lds_p[lane] = pn_e[0];                  // 128 halfwords per MLP, natural channel order
        if (DO_MSG) lds_p[128 + lane] = pn_m[0];
        if (DO_MSG) *reinterpret_cast<f32x2*>(lds_p + 256 + 2 * lane) = hn;
    };
    auto fetch_p = [&]() {                             // ... back as this lane's four words per MLP
        if (SMALLK) return;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            // accumulator row r of channel block mb holds channel ch_nat(mb, r) (EDGE1's one edge Linear: ch_efrag); (P, 0): k = 1 carries nothing
            if (DO_EDGE) pwe[mb] = h == 0 ? (unsigned)reinterpret_cast<const bf16_t*>(lds_p)[EDGE1 ? ch_efrag(mb, r) : ch_nat(mb, r)] : 0u;
            if (DO_MSG) pwm[mb] = h == 0 ? (unsigned)reinterpret_cast<const bf16_t*>(lds_p + 128)[ch_nat(mb, r)] : 0u;
        }
    };
    // ---- one MFMA of chain c (compile-time c, i), accumulating in T
    //   first Linears  (11): [P words x ones(real edges)] [W . e, k-steps 0..7] [routing x Q, 2]
    //   second Linears ( 9): [bias words x ones]          [W . hidden, k-steps 0..7]
    unsigned onesb = 0u;                               // ones column of this block's real edges (B operand, k = 0, 1)
    auto inject_p = [&](f32x16& T, const bf16_t* ptab, unsigned pw, int mb, bool efr) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (!SMALLK) { T = mfma32h(u32x4{pw, 0u, 0u, 0u}, u32x4{onesb ? 0x3C003C00u : 0u, 0u, 0u, 0u}, z); return; }      // f16 ones where the bf16 ones column has them
        // several residues per block: k-pair 4h + jj of group g carries residue g + 4h + jj against the indicator of its edges
        T = z;
        const int chn = efr ? ch_efrag(mb, r) : ch_nat(mb, r);
        for (int g = 0; g < npb; g += 8) {
            u32x4 aw, bwv;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int qq = g + 4 * h + jj, node = blk * npb + qq;
                aw[jj] = (qq < npb && node < ntot) ? (unsigned)ptab[(size_t)node * RN_D + chn] : 0u;
                bwv[jj] = (j >= 0 && q0 == qq) ? 0x3C003C00u : 0u;
            }
            T = mfma32h(aw, bwv, T);
        }
    };
    auto chain_step = [&](auto cc, auto ii, f32x16& T) {
        constexpr int c = decltype(cc)::value, i = decltype(ii)::value;
        constexpr int kind = c >> 2, cb = c & 3;       // kind 0: E1, 1: E2, 2: M1, 3: M2
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (i == 0) {
            if constexpr (kind == 0) inject_p(T, tab.p_e, pwe[cb], cb, EDGE1);
            else if constexpr (kind == 2) inject_p(T, tab.p_m, pwm[cb], cb, false);
            else if constexpr (kind == 1) T = mfma32(u32x4{bwn, 0u, 0u, 0u}, u32x4{onesb, 0u, 0u, 0u}, z);
            else T = mfma32(ones_a, u32x4{bwn, 0u, 0u, 0u}, z);
        } else if constexpr (i <= 8) {
            constexpr int s = i - 1;
            if constexpr (kind == 0 || kind == 2) T = mfma_e(wf[s], ef[s], T);
            else if constexpr (kind == 1) T = mfma32h(wf[s], hb[s], T);
            else T = mfma32h(hb[s], wf[s], T);
            wf[s] = RN_FRAG(RN_NEXT(c), s);            // the register is free again: request the next chain's fragment
            if constexpr (i == 4) {
                constexpr int cn = RN_NEXT(c), kn = cn >> 2;
                if constexpr (kn == 1) bwn = lds_bwe[(cn & 3) * 64 + lane];
                else if constexpr (kn == 3) bwn = lds_bwm[(cn & 3) * 64 + lane];
            }
        } else {
            if constexpr (EDGE1 && kind == 0) T = mfma32h(i == 9 ? perm0e : perm1e, q[2 * cb + (i - 9)], T);      // (the Q tables are f16)
            else T = mfma32h(i == 9 ? perm0 : perm1, q[2 * cb + (i - 9)], T);
        }
    };
    // ---- the epilogue of chain c in GRANULES of ~8 vector instructions (one rides behind each MFMA of the next chain).
    // Quarter v = accumulator registers 4v..4v+3 (two packed-f16 pairs, evaluated side by side so that no VOP3P result
    // is consumed by the very next instruction); a quarter is 2 granules for the first Linears
    // (A: x, x^2, clamp, first Horner step; B: rest of Phi, x Phi -> hb) and 3 for the second ones (C: the f32 tail -
    // residual add and bf16 repack of e, or the running sum of the mean).  g* = context of the block the chain belongs to.
    static_assert(RN_PHI_DEG == 4, "the granule form of the epilogue implements the 4-coefficient Phi (scaled domain, phi4s)");
    float s0 = 0.f, s1 = 0.f;
    float ghres[4] = {0.f, 0.f, 0.f, 0.f};           // h of the block whose message epilogues are running (channel 32 cb + r)
    f16x4 gx = h4(0.f), gs = h4(0.f), gq = h4(0.f);    // state carried between the granules of a quarter
    auto epi_granule = [&](auto cc, auto gg, f32x16& T, int gblk, unsigned gvmask, float gcntf, float ginv) {
        constexpr int c = decltype(cc)::value, g = decltype(gg)::value;
        constexpr int kind = c >> 2, cb = c & 3;
        constexpr bool resid = kind == 1 || (EDGE1 && kind == 0);     // this chain's output is added to e
        constexpr int gpq = (kind == 3 || (resid && !RN_E_F16)) ? 3 : 2;   // granules per quarter (f16 e: the residual add is two instructions of granule B)
        constexpr int v = g / gpq, ph = g % gpq;
        if constexpr (ph == 0) {
            gx = cvt_h4(T[4 * v], T[4 * v + 1], T[4 * v + 2], T[4 * v + 3]);
            gs = clamp01h(gx * gx);                    // (scaled domain: the clamp is the multiply's output modifier)
            gq = __builtin_elementwise_fma(gs, h4(RN_QS3), h4(RN_QS2));
        } else if constexpr (ph == 1) {
            gq = __builtin_elementwise_fma(gq, gs, h4(RN_QS1));
            gq = __builtin_elementwise_fma(gq, gs, h4(RN_QS0));
            const f16x4 pp = __builtin_elementwise_fma(gx, gq, h4(0.5f));
            gq = __builtin_elementwise_min(__builtin_elementwise_max(pp, h4(0.f)), h4(1.f));      // gq now holds Phi
            if constexpr ((kind == 0 && !EDGE1) || kind == 2) {    // hidden activations -> f16 operand fragments of the second Linear
                const f16x4 gv = gx * gq;
                hb[2 * cb + (v >> 1)][2 * (v & 1)] = __builtin_bit_cast(unsigned, lo2(gv));
                hb[2 * cb + (v >> 1)][2 * (v & 1) + 1] = __builtin_bit_cast(unsigned, hi2(gv));
            }
            if constexpr (resid) {                     // e <- e + GELU(.): e is f16 as stored, x Phi + e is one packed fma per fragment word
                constexpr int sp = v >> 1, t = 2 * (v & 1);
                const unsigned o0 = ef[2 * cb + sp][t], o1 = ef[2 * cb + sp][t + 1];
                ef[2 * cb + sp][t] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(lo2(gx), lo2(gq), __builtin_bit_cast(f16x2, o0)));
                ef[2 * cb + sp][t + 1] = __builtin_bit_cast(unsigned, __builtin_elementwise_fma(hi2(gx), hi2(gq), __builtin_bit_cast(f16x2, o1)));
                if constexpr (v & 1) efrag_ptr(e, gblk, lane)[64 * (2 * cb + sp)] = ef[2 * cb + sp];
            }
        } else if constexpr (resid) {                  // (bf16 storage) e <- e + GELU(.), registers 8sp + 2t.. <-> ef[2ob + sp][t]
            constexpr int sp = v >> 1, t = 2 * (v & 1);
            const unsigned o0 = ef[2 * cb + sp][t], o1 = ef[2 * cb + sp][t + 1];
            ef[2 * cb + sp][t] = pack2(fma_mix_lo(T[4 * v], lo2(gq), lo_bf(o0)), fma_mix_hi(T[4 * v + 1], lo2(gq), hi_bf(o0)));
            ef[2 * cb + sp][t + 1] = pack2(fma_mix_lo(T[4 * v + 2], hi2(gq), lo_bf(o1)), fma_mix_hi(T[4 * v + 3], hi2(gq), hi_bf(o1)));
            if constexpr (v & 1) efrag_ptr(e, gblk, lane)[64 * (2 * cb + sp)] = ef[2 * cb + sp];
        } else if constexpr (!SMALLK && !MSGOUT) {     // mean over the real edges of the residue
            // every one of the 32 rows is summed unmasked; rows of absent edges hold GELU(bias) exactly and are taken out again
            if constexpr (v == 0) { s0 = 0.f; s1 = 0.f; gbv = lds_gb[32 * cb + r]; }
            s0 = fma_mix_lo(T[4 * v], lo2(gq), s0);
            s1 = fma_mix_hi(T[4 * v + 1], lo2(gq), s1);
            s0 = fma_mix_lo(T[4 * v + 2], hi2(gq), s0);
            s1 = fma_mix_hi(T[4 * v + 3], hi2(gq), s1);
            if constexpr (v == 3) {
                float sum = s0 + s1;
                sum += __shfl_xor(sum, 32, 64);
                // both lane halves hold the total: the duplicate store of half 1 saves a divergent branch
                agg[(size_t)gblk * RN_D + 32 * cb + r] = fmaf(sum - gcntf * gbv, ginv, ghres[cb]);
            }
        } else {                                       // several residues per block / per-edge message output
            T[4 * v] *= (float)gq[0]; T[4 * v + 1] *= (float)gq[1]; T[4 * v + 2] *= (float)gq[2]; T[4 * v + 3] *= (float)gq[3];
            if constexpr (v == 3) {                    // ... the whole activated tile at once
                if (MSGOUT) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int er = (i & 3) + 8 * (i >> 2) + 4 * h;
                        const int node = gblk * npb + er / k;
                        if (er < npb * k && node < ntot)
                            msg_out[((size_t)gblk * npb * k + er) * RN_D + 32 * cb + r] = kGAi * T[i] * (float)((gvmask >> er) & 1u);
                    }
                }
                for (int q1 = 0; q1 < npb; ++q1) {
                    const int node = gblk * npb + q1;
                    if (node >= ntot) break;
                    const unsigned seg = SMALLK ? (gvmask & (((1u << k) - 1u) << (q1 * k))) : gvmask;
                    const unsigned segh = seg >> (4 * h);
                    float sum = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        sum = fmaf(T[i], (float)((segh >> ((i & 3) + 8 * (i >> 2))) & 1u), sum);
                    sum += __shfl_xor(sum, 32, 64);
                    const int cnt = __popc(seg);
                    if (h == 0) agg[(size_t)node * RN_D + 32 * cb + r] = kGAi * sum / (float)(cnt > 0 ? cnt : 1) + (tab.h_res ? tab.h_res[(size_t)node * RN_D + 32 * cb + r] : 0.f);
                }
            }
        }
    };
#define RN_NGRAN(c) ((((c) >> 2) == 3 || (!RN_E_F16 && (((c) >> 2) == 1 || (EDGE1 && (c) < 4)))) ? 12 : 8)

    // ---- prologue: state of the first block, fragments of its first chain
    j = (slot_ok && blk * npb + q0 < ntot) ? jraw_first : -1;
    {
        const int qr = RN_QROW(j);
#pragma unroll
        for (int s = 0; s < 8; ++s) gather_q(q, DO_EDGE ? tab.q_e : tab.q_m, qr, s);
    }
    stage_p();
    fetch_p();
#pragma unroll
    for (int s = 0; s < 8; ++s) wf[s] = RN_FRAG(C_FIRST, s);
    onesb = j >= 0 ? ones_w : 0u;

    // slot c: chain c against the epilogue of chain cprev (the one before it in the sequence; for the first chain of a
    // block that is the last chain of the previous block, whose context g* is still in place)
    int gblk = blk; unsigned gvmask = 0u; float gcntf = 0.f, ginv = 0.f;
#define RN_TILE(c) (((c) & 1) ? tB : tA)
// MFMA order of a chain: the k-steps fed by the epilogue running beside it come last (BOUNDARY slots: the consumer of a
// Linear boundary; a first Linear then issues its two routing MFMAs before k-steps 6 and 7)
#define RN_SLOT(c, cprev, WITH_EPI, EXTRA)                                                                     \
    do {                                                                                                       \
        constexpr int nst_ = (((c) >> 2) & 1) ? 9 : 11;                                                        \
        constexpr bool bnd_ = (WITH_EPI) && ((c) >> 2) != ((cprev) >> 2) && ((cprev) >> 2) != 3;               \
        constexpr int ngr_ = (WITH_EPI) ? RN_NGRAN(cprev) : 0;                                                 \
        constexpr int nfree_ = bnd_ ? nst_ - 2 : nst_;      /* MFMAs the granules are spread behind */         \
        static_for<nst_>([&](auto ii) {                                                                        \
            constexpr int p_ = decltype(ii)::value;                                                            \
            constexpr int i_ = (bnd_ && nst_ == 11) ? (p_ <= 6 ? p_ : p_ <= 8 ? p_ + 2 : p_ - 2) : p_;         \
            chain_step(std::integral_constant<int, (c)>{}, std::integral_constant<int, i_>{}, RN_TILE(c));     \
            EXTRA(i_);                                                                                         \
            RN_FENCE();                                                                                        \
            if constexpr (ngr_ > 0 && p_ < nfree_) {                                                           \
                constexpr int g0_ = p_ * ngr_ / nfree_, g1_ = (p_ + 1) * ngr_ / nfree_;                        \
                static_for<g1_ - g0_>([&](auto dg) {                                                           \
                    epi_granule(std::integral_constant<int, (cprev)>{}, std::integral_constant<int, g0_ + decltype(dg)::value>{}, RN_TILE(cprev), gblk, gvmask, gcntf, ginv); \
                });                                                                                            \
                RN_FENCE();                                                                                    \
            }                                                                                                  \
        });                                                                                                    \
    } while (0)
#define RN_NOEXTRA(i) do { } while (0)

    RN_SLOT(C_FIRST, C_FIRST, false, RN_NOEXTRA);      // first chain of the first block: nothing to overlap with yet
    while (true) {
        const int nblk = blk + stride;
        const bool has_next = nblk < blk_end;
        const int nb_c = has_next ? nblk : blk;        // the last iteration re-reads its own block (results unused)
        const int jn_raw = nbr[RN_IDX(nb_c)];          // next block's neighbour indices and P words; consumed behind the message MLP
        load_p(nb_c);
        {
            const unsigned vmask = (unsigned)(__ballot(j >= 0) & 0xffffffffull);   // bit r = edge r is real
            const int cnt_all = __popc(vmask);
            gblk = blk; gvmask = vmask; gcntf = (float)(32 - cnt_all);
            ginv = cnt_all > 0 ? kGAi * __builtin_amdgcn_rcpf((float)cnt_all) : 0.f;      // (the summed messages are a m)
            if (DO_MSG && !SMALLK && !MSGOUT) {       // (the wave's slot still holds THIS block's row: the next one is staged behind chain 12)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) ghres[mb] = __uint_as_float(lds_p[256 + 32 * mb + r]);
            }
        }
        int jn = -1;
        // requests for the next block: its e fragment s as soon as the last reader of ef[s] - k-step s of the last message
        // chain of Linear 1 - has issued (4+ chains before it is needed) ...
#define RN_EXTRA_E(i) do { if constexpr ((i) >= 1 && (i) <= 8) load_e(nb_c, (i) - 1); } while (0)
        // ... its Q rows and P words behind the MFMAs of the last four chains
#define RN_EXTRA_N(cc, i) do { if constexpr ((cc) == 0 && (i) == 0) { jn = (slot_ok && nb_c * npb + q0 < ntot) ? jn_raw : -1; stage_p(); } \
                               if constexpr ((i) == 1 || (i) == 2) gather_q(q, DO_EDGE ? tab.q_e : tab.q_m, RN_QROW(jn), 2 * (cc) + (i) - 1); } while (0)
#define RN_EXTRA_N0(i) RN_EXTRA_N(0, i)
#define RN_EXTRA_N1(i) RN_EXTRA_N(1, i)
#define RN_EXTRA_N2(i) RN_EXTRA_N(2, i)
#define RN_EXTRA_N3(i) RN_EXTRA_N(3, i)
        // ... and the Q rows of this block's message MLP behind the first chain of the second edge Linear
#define RN_EXTRA_QM(i) do { if constexpr ((i) >= 1 && (i) <= 8 && DO_MSG) gather_q(q, tab.q_m, RN_QROW(j), (i) - 1); } while (0)
        if constexpr (DO_EDGE) {
            RN_SLOT(1, 0, true, RN_NOEXTRA); RN_SLOT(2, 1, true, RN_NOEXTRA); RN_SLOT(3, 2, true, RN_NOEXTRA);
            if constexpr (!EDGE1) {
                RN_SLOT(4, 3, true, RN_EXTRA_QM); RN_SLOT(5, 4, true, RN_NOEXTRA); RN_SLOT(6, 5, true, RN_NOEXTRA); RN_SLOT(7, 6, true, RN_NOEXTRA);
            }
        }
        if constexpr (DO_EDGE && DO_MSG && !EDGE1) RN_SLOT(8, 7, true, RN_NOEXTRA);
        if constexpr (DO_EDGE && DO_MSG && EDGE1) RN_SLOT(8, 3, true, RN_EXTRA_QM);    // (the message MLP's Q rows are requested at its own first chain)
        if constexpr (DO_MSG) {
                        RN_SLOT(9, 8, true, RN_NOEXTRA); RN_SLOT(10, 9, true, RN_NOEXTRA); RN_SLOT(11, 10, true, RN_EXTRA_E);
            RN_SLOT(12, 11, true, RN_EXTRA_N0); RN_SLOT(13, 12, true, RN_EXTRA_N1);
            RN_SLOT(14, 13, true, RN_EXTRA_N2); RN_SLOT(15, 14, true, RN_EXTRA_N3);
        } else {
            // edge update only (stand-alone API): the last epilogue rewrites e, so the next block's state cannot be requested early
            static_for<RN_NGRAN(C_LAST)>([&](auto gg) { epi_granule(std::integral_constant<int, C_LAST>{}, gg, RN_TILE(C_LAST), gblk, gvmask, gcntf, ginv); });
            RN_FENCE();
            jn = (slot_ok && nb_c * npb + q0 < ntot) ? jn_raw : -1;
            stage_p();
#pragma unroll
            for (int s = 0; s < 8; ++s) { load_e(nb_c, s); gather_q(q, tab.q_e, RN_QROW(jn), s); }
        }
        if (!has_next) break;
        fetch_p();
        blk = nblk;
        j = jn;
        onesb = j >= 0 ? ones_w : 0u;
        RN_FENCE();
        if constexpr (DO_MSG) RN_SLOT(C_FIRST, 15, true, RN_NOEXTRA);      // first chain of the next block || last epilogue of this one
        else RN_SLOT(C_FIRST, C_FIRST, false, RN_NOEXTRA);
    }
    if constexpr (DO_MSG) static_for<RN_NGRAN(15)>([&](auto gg) { epi_granule(std::integral_constant<int, 15>{}, gg, RN_TILE(15), gblk, gvmask, gcntf, ginv); });
#undef RN_EXTRA_QM
#undef RN_EXTRA_N3
#undef RN_EXTRA_N2
#undef RN_EXTRA_N1
#undef RN_EXTRA_N0
#undef RN_EXTRA_E
#undef RN_EXTRA_N
#undef RN_NOEXTRA
#undef RN_SLOT
#undef RN_NGRAN
#undef RN_TILE
#undef RN_FENCE
#undef RN_QROW
#undef RN_IDX
#undef RN_NEXT
#undef RN_FRAG
}

static int num_cus() { return rn_num_cus(); }

void launch_mpnn_bf16(const PackInfo& pk, int k, bool do_edge, bool do_msg, const int* nbr, bf16_t* e,
                      const bf16_t* p_e, const bf16_t* q_e, const bf16_t* p_m, const bf16_t* q_m, MpnnWB we, MpnnWB wm,
                      float* agg, float* msg_out, bool edge1, const float* h_res, hipStream_t s) {
    if (resmpnn_covers(k, edge1, msg_out != nullptr)) {
        launch_resmpnn_bf16(pk, k, do_edge, do_msg, nbr, e, p_e, q_e, p_m, q_m, we, wm, agg, h_res, s);
        return;
    }
    const int npb = k > 16 ? 1 : 32 / k;
    const int max_blocks = (pk.Nmax + npb - 1) / npb;
    int grid = (max_blocks + RN_MPNN_WAVES - 1) / RN_MPNN_WAVES;
    if (grid >= 8) grid = (grid + 7) & ~7;            // a multiple of 8 switches the kernel to its XCD-aware block mapping
    if (grid > num_cus()) grid = num_cus();
    if (grid < 1) grid = 1;
    size_t lds = RN_MPNN_LDS;
    NodeTabs tab{p_e, q_e, p_m, q_m, h_res};
    const bool smallk = k <= 16, mo = msg_out != nullptr;
#define RN_LAUNCH(E, M, S, O, E1)                                                                              \
    do {                                                                                                       \
        static DevAttr attr;                                                                                   \
        ensure_dyn_lds((const void*)k_mpnn_bf16<E, M, S, O, E1>, RN_MPNN_LDS, attr);                           \
        hipLaunchKernelGGL((k_mpnn_bf16<E, M, S, O, E1>), dim3(grid), dim3(RN_MPNN_WAVES * 64), lds, s, pk, k, nbr, e, tab, we, wm, agg, msg_out); \
    } while (0)
    if (do_edge && do_msg && edge1) { if (smallk) RN_LAUNCH(true, true, true, false, true); else RN_LAUNCH(true, true, false, false, true); }
    else if (do_edge && edge1)      { if (smallk) RN_LAUNCH(true, false, true, false, true); else RN_LAUNCH(true, false, false, false, true); }
    else if (do_edge && do_msg) { if (smallk) RN_LAUNCH(true, true, true, false, false); else RN_LAUNCH(true, true, false, false, false); }
    else if (do_edge)      { if (smallk) RN_LAUNCH(true, false, true, false, false); else RN_LAUNCH(true, false, false, false, false); }
    else if (mo)           { if (smallk) RN_LAUNCH(false, true, true, true, false); else RN_LAUNCH(false, true, false, true, false); }
    else                   { if (smallk) RN_LAUNCH(false, true, true, false, false); else RN_LAUNCH(false, true, false, false, false); }
#undef RN_LAUNCH
}

// ------------------------------------------------------------------------------------------
// Edge featurisation + embedding MLP on MFMA (feature.py:386-571).  One wave per 32-edge block; lane (r, h) = (edge r,
// k-half h) computes the EMB_SLOTS raw features of its half (emb_feature_of_slot) straight into the B fragments of
// Linear(90,128) - the 90-wide tensor exists only in registers - then GELU -> Linear(128,128) -> GELU -> e0 in bf16
// (16-byte stores, fragment-major); absent edges and padding slots store zeros.
//   * the CENTRAL residue of a block is wave-uniform (k > 16: one residue per block): its record is read with scalar
//     loads and enters the vector arithmetic as SGPR operands - no per-lane loads, no address arithmetic, no selects;
//   * of the NEIGHBOUR's record each lane half gathers only the 28 floats (7 x 16 B) of the items it owns;
//   * the loop is software-pipelined on its only dependent memory chain: the neighbour index of block i+2 and the
//     neighbour record of block i+1 are requested while block i runs through the matrix pipe;
//   * no divergent branch: validity is an AND mask on the packed output words.
#define RN_EE_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifndef EE_WAVES
#define EE_WAVES 8                // waves per workgroup; two workgroups (61 KiB of LDS each) per CU
#endif
template <bool SMALLK>
__global__ void __launch_bounds__(EE_WAVES * 64, EE_WAVES / 2) k_edge_embed_bf16(PackInfo pk, int k, const float* __restrict__ geomh,
        const int* __restrict__ nbr, const bf16_t* __restrict__ img_g, const float* __restrict__ b0,
        const float* __restrict__ b1p, bf16_t* __restrict__ e) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* img = reinterpret_cast<u32x4*>(smem);
    constexpr int NFRAG = 4 * EMB_KS + 32;
    const int tid = threadIdx.x;
    float* lds_b = reinterpret_cast<float*>(smem + (size_t)NFRAG * 1024);      // both bias vectors: a global read per channel block
    if (tid < 128) lds_b[tid] = kGA * b0[tid];                                  // of every edge block would expose an L2 round trip each (scaled domain)
    else if (tid < 256) lds_b[tid] = b1p[tid - 128];
    {   // 60 KiB image: loads of a thread first, LDS writes after (see stage_image)
        constexpr int NT = EE_WAVES * 64, PER = (NFRAG * 64 + NT - 1) / NT;
        u32x4 t[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) t[i] = reinterpret_cast<const u32x4*>(img_g)[min(tid + i * NT, NFRAG * 64 - 1)];
#pragma unroll
        for (int i = 0; i < PER; ++i) if (tid + i * NT < NFRAG * 64) img[tid + i * NT] = t[i];
    }
    __syncthreads();
    const int ntot = pk.cu[pk.B];
    const int npb = SMALLK ? 32 / k : 1;
    const int nblocks = (ntot + npb - 1) / npb;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int q0 = SMALLK ? r / k : 0;
    const bool slot_ok = SMALLK ? q0 < npb : r < k;
    const int last_idx = ntot * k - 1;
    const int stride = gridDim.x * EE_WAVES;

    // neighbour row of this lane's edge in block b (-1: no edge), addresses clamped, loads unconditional
    auto load_j = [&](int b) -> int {
        const int i = b * npb * k + r;
        const int jr = nbr[min(max(i, 0), max(last_idx, 0))];
        return (b < nblocks && slot_ok && b * npb + q0 < ntot) ? jr : -1;
    };
    f32x4 nrec[7];
    auto load_rec = [&](int j) {
        const f32x4* src = reinterpret_cast<const f32x4*>(geomh + (size_t)(j >= 0 ? j : 0) * RN_GEOMH + 32 * h);
#pragma unroll
        for (int v = 0; v < 7; ++v) nrec[v] = src[v];
    };
    int blk = blockIdx.x * EE_WAVES + wave;
    if (blk >= nblocks) return;
    int j_cur = load_j(blk), j_nxt = load_j(blk + stride);
    load_rec(j_cur);

    for (; blk < nblocks; blk += stride) {
        // ---- central record: wave-uniform pointer (k > 16) -> scalar loads; per-lane pointer otherwise
        const int cnode = SMALLK ? min(blk * npb + q0, ntot - 1) : __builtin_amdgcn_readfirstlane(blk);
        const float* __restrict__ gc = geomh + (size_t)cnode * RN_GEOMH;
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
        for (int p = 51; p < EMB_SLOTS; ++p) ft[p] = 0.f;
        u32x4 xf[EMB_KS];
#pragma unroll
        for (int s = 0; s < EMB_KS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) xf[s][t] = p_pack2(ft[8 * s + 2 * t], ft[8 * s + 2 * t + 1]);      // f16: distances keep 11 significand bits (8 as bf16)
        RN_EE_FENCE();
        // ---- requests of the next two blocks (their latency passes under this block's matrix work)
        const unsigned vmask = j_cur >= 0 ? 0xffffffffu : 0u;
        load_rec(j_nxt);
        j_cur = j_nxt;
        j_nxt = load_j(blk + 2 * stride);
        RN_EE_FENCE();

        u32x4 hb[8];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            f32x16 acc = init_vec16(lds_b + 32 * mb + 16 * h);
#pragma unroll
            for (int s = 0; s < EMB_KS; ++s) acc = mfma32h(img[(mb * EMB_KS + s) * 64 + lane], xf[s], acc);
#pragma unroll
            for (int t = 0; t < 4; ++t) {           // packed-f16 GELU, hidden activations stay f16 (as in the fused kernel)
                const f16x4 x = cvt_h4(acc[2 * t], acc[2 * t + 1], acc[8 + 2 * t], acc[8 + 2 * t + 1]);
                const f16x4 g = x * phi4s(x);
                hb[2 * mb][t] = __builtin_bit_cast(unsigned, lo2(g));
                hb[2 * mb + 1][t] = __builtin_bit_cast(unsigned, hi2(g));
            }
            RN_EE_FENCE();
        }
        u32x4* ewp = efrag_ptr(e, blk, lane);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x16 acc = init_vec16(lds_b + 128 + 32 * ob + 16 * h);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = mfma32h(img[(4 * EMB_KS + ob * 8 + ks) * 64 + lane], hb[ks], acc);
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {        // padding slots of the block and absent edges are stored as zeros
                u32x4 nw;
#pragma unroll
                for (int t = 0; t < 4; t += 2) {
                    const f16x4 xq = cvt_h4(acc[8 * sp + 2 * t], acc[8 * sp + 2 * t + 1], acc[8 * sp + 2 * t + 2], acc[8 * sp + 2 * t + 3]);
                    const f16x4 gq4 = xq * phi4s(xq);
                    nw[t] = __builtin_bit_cast(unsigned, lo2(gq4)) & vmask;
                    nw[t + 1] = __builtin_bit_cast(unsigned, hi2(gq4)) & vmask;
                }
                ewp[64 * (2 * ob + sp)] = nw;
            }
            RN_EE_FENCE();
        }
    }
}

void launch_edge_embed_bf16(const PackInfo& pk, int k, const float* geomh, const int* nbr, const bf16_t* img,
                            const float* b0, const float* b1p, bf16_t* e, hipStream_t s) {
    const int npb = k > 16 ? 1 : 32 / k;
    const int max_blocks = (pk.Nmax + npb - 1) / npb;
    int grid = (max_blocks + EE_WAVES - 1) / EE_WAVES;
    if (grid > 2 * num_cus()) grid = 2 * num_cus();
    if (grid < 1) grid = 1;
    size_t lds = (size_t)(4 * EMB_KS + 32) * 1024 + 1024;
    if (k > 16) hipLaunchKernelGGL(k_edge_embed_bf16<false>, dim3(grid), dim3(EE_WAVES * 64), lds, s, pk, k, geomh, nbr, img, b0, b1p, e);
    else hipLaunchKernelGGL(k_edge_embed_bf16<true>, dim3(grid), dim3(EE_WAVES * 64), lds, s, pk, k, geomh, nbr, img, b0, b1p, e);
}

// ------------------------------------------------------------------------------------------
// Node-level Linear on MFMA: Y = act([X | X2] . W^T + bias) (+ res), X f32 rows (converted to
// bf16 while staging), W bf16 [N][K] row-major (both operands are k-contiguous: A and B
// fragments are plain 16-byte LDS reads).  128 x 128 tile per 256-thread workgroup, K staged
// through LDS in steps of 64 with a padded row stride (72 bf16 = 36 dwords: conflict-free
// ds_read_b128 for 16 consecutive rows); next tile's global loads are issued before the MFMAs.
#define GB_LD 72
__global__ void __launch_bounds__(256) k_gemm_bf16(const int* __restrict__ ntot_p, const float* __restrict__ X, int ldx, int K1,
        const float* __restrict__ X2, int ldx2, int K2, const bf16_t* __restrict__ W, const float* __restrict__ bias,
        int N, int act, const float* __restrict__ res, int ldres, float* __restrict__ Y, int ldy) {
    __shared__ __attribute__((aligned(16))) bf16_t As[128 * GB_LD];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[128 * GB_LD];
    const int ntot = *ntot_p;
    const int row0 = blockIdx.x * 128;
    if (row0 >= ntot) return;
    const int col0 = blockIdx.y * 128;
    const int K = K1 + K2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int srow = tid >> 1, shalf = tid & 1;           // staging: one 32-wide half row per thread

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    f32x4 ar[8];
    u32x4 br[4];
    auto load_tile = [&](int k0) {
        const int kk = k0 + shalf * 32;
        const int row = row0 + srow;
        const bool okA = row < ntot && kk < K;
        const float* src = kk < K1 ? X + (size_t)row * ldx + kk : X2 + (size_t)row * ldx2 + (kk - K1);
#pragma unroll
        for (int v = 0; v < 8; ++v) ar[v] = okA ? *reinterpret_cast<const f32x4*>(src + 4 * v) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = col0 + srow;
        const bool okB = n < N && kk < K;
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(W + (size_t)n * K + kk);
#pragma unroll
        for (int v = 0; v < 4; ++v) br[v] = okB ? wsrc[v] : u32x4{0u, 0u, 0u, 0u};
    };
    auto store_tile = [&]() {
        u32x4* da = reinterpret_cast<u32x4*>(As + srow * GB_LD + shalf * 32);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            u32x4 w;
            w[0] = pack2(ar[2 * v][0], ar[2 * v][1]); w[1] = pack2(ar[2 * v][2], ar[2 * v][3]);
            w[2] = pack2(ar[2 * v + 1][0], ar[2 * v + 1][1]); w[3] = pack2(ar[2 * v + 1][2], ar[2 * v + 1][3]);
            da[v] = w;
        }
        u32x4* db = reinterpret_cast<u32x4*>(Bs + srow * GB_LD + shalf * 32);
#pragma unroll
        for (int v = 0; v < 4; ++v) db[v] = br[v];
    };

    load_tile(0);
    for (int k0 = 0; k0 < K; k0 += 64) {
        __syncthreads();                                   // previous tile's fragment reads are done
        store_tile();
        __syncthreads();
        if (k0 + 64 < K) load_tile(k0 + 64);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 af[2], bfm[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = *reinterpret_cast<const u32x4*>(As + (64 * wr + 32 * t + r) * GB_LD + 16 * ks + 8 * h);
                bfm[t] = *reinterpret_cast<const u32x4*>(Bs + (64 * wc + 32 * t + r) * GB_LD + 16 * ks + 8 * h);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = mfma32(af[a], bfm[b], acc[a][b]);
        }
    }
    // epilogue: the 16 rows a lane holds of one 32x32 tile are handled together - the residual loads go out as a batch
    // (a per-element load -> wait -> store chain costs one L2 round trip per element) and the activation is a uniform branch
    const bool has_res = res != nullptr;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + 64 * wc + 32 * b + r;
        const bool colok = col < N;
        const int cc = colok ? col : 0;
        const float bv = bias ? bias[cc] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int rbase = row0 + 64 * wr + 32 * a + 4 * h;
            float v[16], rv[16];
            if (has_res) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = rbase + (i & 3) + 8 * (i >> 2);
                    rv[i] = res[(size_t)(row < ntot ? row : ntot - 1) * ldres + cc];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = acc[a][b][i] + bv;
            if (act == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = gelu_erf(v[i]);
            }
            if (has_res) {
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] += rv[i];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = rbase + (i & 3) + 8 * (i >> 2);
                if (colok && row < ntot) Y[(size_t)row * ldy + col] = v[i];
            }
        }
    }
}

void launch_gemm_bf16(const int* ntot, int mmax, const float* X, int ldx, int K1, const float* X2, int ldx2, int K2,
                      const bf16_t* W, const float* bias, int N, int act, const float* res, int ldres,
                      float* Y, int ldy, hipStream_t s) {
    dim3 grid((mmax + 127) / 128, (N + 127) / 128);
    hipLaunchKernelGGL(k_gemm_bf16, grid, dim3(256), 0, s, ntot, X, ldx, K1, X2, ldx2, K2, W, bias, N, act, res, ldres, Y, ldy);
}

// ------------------------------------------------------------------------------------------
// Fused node-level FFN chain on MFMA (functional.py:119-127,170 / 179-187,200 / 62-74,86):
//   X (f32 rows, K0 wide) -> Linear(K0,H)+GELU -> NH x [Linear(H,H)+GELU] -> Linear(H,NOUT)
// evaluated transposed like the edge MLPs (rows of the batch on the MFMA columns = lanes, channels
// on the accumulator registers), so the hidden activations of a 32-row block never leave the
// registers of their wave: 128 VGPRs hold 512 bf16 channels as ready-made B fragments.  A
// 256-thread workgroup (4 waves x 32 rows, one wave per SIMD, whole 512-register file) streams
// the weights ONCE per 128 rows: fragment images (1 KiB = one MFMA A fragment, consumption order)
// are copied global -> registers -> LDS double buffer, one chunk = one 32-channel output block.
struct ChainW {
    const bf16_t* img;        // all layers' fragment chunks, consumption order
    const float* bias[5];     // per layer, natural channel order (last layer padded to NOUT)
};

// The weight stream is a uniform sequence of 32 KiB chunks (32 fragments; a layer with NKS k-steps
// packs 32/NKS output blocks per chunk) moved global -> LDS by LDS-DMA (global_load_lds_dwordx4:
// no VGPRs) into a 4-slot ring, three chunks ahead of the matrix pipe; each thread issues 8 DMAs per
// chunk, so "chunk c has landed" is the counted wait vmcnt(8 * chunks issued after c) followed by a
// raw s_barrier (a __syncthreads() would drain the ring: cdna guide, "Pipelining across barriers").
#define CH_RING 4
// timing ablations of k_ffn_chain (WRONG results; tools/build_mpnn_variant.sh with RN_VARIANT_SRC=kernels_bf16.hip passes -DRN_EXPERIMENTS)
#if !defined(RN_EXPERIMENTS) && (defined(CH_EXP_NODMA) || defined(CH_EXP_NOGELU) || defined(CH_EXP_NOLDS) || defined(CH_EXP_NOBAR) || defined(CH_GRANULES1))
#error "k_ffn_chain ablations need -DRN_EXPERIMENTS"
#endif
#ifndef CH_GELU
#define CH_GELU 1
#endif
__device__ __forceinline__ void chain_issue(const u32x4* __restrict__ img, u32x4* ring, int c, int tid) {
    const u32x4* src = img + (size_t)c * 2048 + tid;
    u32x4* dst = ring + (c % CH_RING) * 2048 + (tid & ~63);          // wave-uniform base; the DMA adds lane * 16
#pragma unroll
    for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 256),
                                         (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
}
// one of the 8 DMA pieces of chunk c (issued between the MFMAs of the chunk that runs three chunks earlier: the ~100
// cycles a DMA holds the wave's issue then pass under a busy matrix pipe instead of in front of it)
__device__ __forceinline__ void chain_issue_piece(const u32x4* __restrict__ img, u32x4* ring, int c, int tid, int i) {
#ifdef CH_EXP_NODMA
    return;
#endif
    const u32x4* src = img + (size_t)c * 2048 + tid + i * 256;
    u32x4* dst = ring + (c % CH_RING) * 2048 + (tid & ~63) + i * 256;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}
__device__ __forceinline__ void chain_wait(int chunks_after) {       // folded to one s_waitcnt after unrolling
    if (chunks_after >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (chunks_after == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef CH_EXP_NOBAR
    __builtin_amdgcn_s_barrier();
#endif
}

// one Linear of the chain = NOB output blocks of NKS k-steps = NOB*NKS/32 chunks, starting at global
// chunk C0 of NCH_T; in[] are the B fragments of its input; GELU + repack into out[] or f32 store.
// The wave is alone on its SIMD (404 registers), so matrix and vector work only overlap if they alternate in PROGRAM order: the
// activation arithmetic of output block ob - 1 (two accumulator tiles alternate) is cut into four granules (one per quarter of the
// tile, ~18 packed-f16 instructions) that ride behind the MFMAs of block ob, pinned with scheduling fences - the scheme of the fused
// ResMPNN kernel.  The last block of a layer is flushed before the next layer starts (its output is that layer's operand).
template <int NKS, int NOB, int C0, int NCH_T, bool LAST>
__device__ __forceinline__ void chain_layer(const u32x4* __restrict__ img, u32x4* ring, int tid, int lane, int h,
                                            const u32x4 (&in)[32], u32x4 (&out)[32], const float* bias_lds,
                                            float* __restrict__ yrow, int n_valid, bool row_ok) {
    constexpr int OPC = 32 / NKS, NCH = NOB / OPC;
    f32x16 accA, accB;                                  // block ob accumulates in (ob & 1 ? accB : accA)
    // quarter t of block ob's epilogue: GELU + repack into out[] (hidden layers) or the f32 store of 4 channels (last layer)
    auto granule = [&](const f32x16& acc, auto obc, auto tc) {
        constexpr int ob = decltype(obc)::value, t = decltype(tc)::value;
#ifdef CH_EXP_NOGELU
        if constexpr (!LAST) { out[2 * ob][t] = __float_as_uint(acc[2 * t]); out[2 * ob + 1][t] = __float_as_uint(acc[8 + 2 * t]); return; }
#endif
        if constexpr (!LAST) {
#if CH_GELU == 2           /* experiment: f32 sigmoid-form GELU (2.7e-4) */
            const f16x4 g = cvt_h4(gelu_fast(acc[2 * t]), gelu_fast(acc[2 * t + 1]), gelu_fast(acc[8 + 2 * t]), gelu_fast(acc[8 + 2 * t + 1]));
#else
            const f16x4 x = cvt_h4(acc[2 * t], acc[2 * t + 1], acc[8 + 2 * t], acc[8 + 2 * t + 1]);
#if CH_GELU == 1
            const f16x4 g = x * phi5n(x);          // packed-f16 GELU in the chain's scaled domain (x = a pre-activation), five-coefficient Phi (1.2e-3):
                                                   // what follows a node chain is a GraphNormalization, which amplifies the approximation error
#else
            const f16x4 g = x * phi4(x);           // packed-f16 GELU (phi4), hidden activations stay f16: as in the edge kernels
#endif
#endif
            out[2 * ob][t] = __builtin_bit_cast(unsigned, lo2(g));
            out[2 * ob + 1][t] = __builtin_bit_cast(unsigned, hi2(g));
        } else if (row_ok) {
            const int c0 = 32 * ob + 16 * h + 4 * t;
            if (c0 < n_valid) *reinterpret_cast<f32x4*>(yrow + c0) = f32x4{acc[4 * t], acc[4 * t + 1], acc[4 * t + 2], acc[4 * t + 3]};
        }
    };
    static_for<NCH>([&](auto chc) {
        constexpr int ch = decltype(chc)::value, c = C0 + ch;
        chain_wait(NCH_T - 1 - c < 2 ? NCH_T - 1 - c : 2);
        const u32x4* buf = ring + (c % CH_RING) * 2048 + lane;
        // the 32 weight fragments of the chunk go through an 8-deep register ring, so that no MFMA waits on its LDS read
        u32x4 fr[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) fr[m] = buf[m * 64];
        static_for<OPC>([&](auto oc) {
            constexpr int o = decltype(oc)::value, ob = ch * OPC + o;
            f32x16& acc = (ob & 1) ? accB : accA;
            f32x16& prev = (ob & 1) ? accA : accB;
            acc = init_vec16(bias_lds + 32 * ob + 16 * h);
            static_for<NKS>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, m = o * NKS + ks;      // m: MFMA number within the chunk (32 per chunk)
                acc = mfma32h(fr[m & 7], in[ks], acc);    // f16 operands in every layer
#ifndef CH_EXP_NOLDS
                if constexpr (m + 8 < 32) fr[m & 7] = buf[(m + 8) * 64];
#endif
                if constexpr (c + 3 < NCH_T && (m & 3) == 1) chain_issue_piece(img, ring, c + 3, tid, m >> 2);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ob > 0) {                        // the previous block's activation arithmetic behind the MFMAs of this one
#ifndef CH_GRANULES1     /* (default: one quarter of the previous tile at a time) */
                    if constexpr (NKS >= 4) {
                        if constexpr ((ks + 1) % (NKS / 4) == 0)
                            granule(prev, std::integral_constant<int, (ob > 0 ? ob - 1 : 0)>{}, std::integral_constant<int, (ks + 1) / (NKS / 4) - 1>{});
                    } else {
                        granule(prev, std::integral_constant<int, (ob > 0 ? ob - 1 : 0)>{}, std::integral_constant<int, (2 * ks) & 3>{});
                        granule(prev, std::integral_constant<int, (ob > 0 ? ob - 1 : 0)>{}, std::integral_constant<int, (2 * ks + 1) & 3>{});
                    }
#else
                    // experiment CH_GRANULES1: all four quarters in one place (eight independent word chains) - 63.0 against 62.3 us: no gain
                    if constexpr (ks == (NKS >= 2 ? NKS / 2 - 1 : 0))
                        static_for<4>([&](auto tc) { granule(prev, std::integral_constant<int, (ob > 0 ? ob - 1 : 0)>{}, tc); });
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        });
    });
    static_for<4>([&](auto tc) { granule(((NOB - 1) & 1) ? accB : accA, std::integral_constant<int, NOB - 1>{}, tc); });
    __builtin_amdgcn_sched_barrier(0);
}

template <int K0, int H, int NH, int NOUT>
__global__ void __launch_bounds__(256, 1) k_ffn_chain(const int* __restrict__ ntot_p, const float* __restrict__ X, int ldx,
        const float* __restrict__ X2, int ldx2, ChainW w, float* __restrict__ Y, int ldy, int n_valid) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [ring 4 x 32 KiB][biases]
    u32x4* ring = reinterpret_cast<u32x4*>(smem);
    float* bias_lds = reinterpret_cast<float*>(smem + CH_RING * 32768);
    const int ntot = *ntot_p;
    const int row_blk = blockIdx.x * 128;
    if (row_blk >= ntot) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int row = row_blk + 32 * wave + r;
    const bool row_ok = row < ntot;
    constexpr int HK = H / 16, HB = H / 32, OB = NOUT / 32;
    constexpr int NCH0 = HB * (K0 / 16) / 32, NCHH = HB * HK / 32, NCHL = OB * HK / 32, NCH_T = NCH0 + NH * NCHH + NCHL;
    // biases -> LDS (layer l at l*H), input rows -> B fragments (natural k order; X for k < 128, X2 beyond)
    // (all loads of a phase are issued before their first use: a load -> wait -> use chain per iteration costs a memory
    //  round trip each; rows beyond the batch read row ntot - 1 and are never stored)
    {
        constexpr int NBL = ((NH + 1) * H + NOUT + 255) / 256;
        float bv[NBL];
#pragma unroll
        for (int t = 0; t < NBL; ++t) {
            const int idx = tid + 256 * t;                     // position in the concatenated [layer][channel] list
            const int l = idx / H < NH + 1 ? idx / H : NH + 1, i = idx - l * H;
            bv[t] = (idx < (NH + 1) * H + NOUT) ? (CH_GELU == 1 && l < NH + 1 ? kGAn : 1.f) * w.bias[l][i] : 0.f;      // (scaled domain: hidden biases x a)
        }
#pragma unroll
        for (int t = 0; t < NBL; ++t) if (tid + 256 * t < (NH + 1) * H + NOUT) bias_lds[tid + 256 * t] = bv[t];
    }
    u32x4 a[32], b[32];
    {
        const int rr = row_ok ? row : ntot - 1;
        f32x4 v0[K0 / 16], v1[K0 / 16];
#pragma unroll
        for (int s = 0; s < K0 / 16; ++s) {
            const int kk = 16 * s + 8 * h;
            const float* src = (K0 > 128 && kk >= 128) ? X2 + (size_t)rr * ldx2 + (kk - 128) : X + (size_t)rr * ldx + kk;
            v0[s] = *reinterpret_cast<const f32x4*>(src);
            v1[s] = *reinterpret_cast<const f32x4*>(src + 4);
        }
#pragma unroll
        for (int s = 0; s < K0 / 16; ++s)
            a[s] = u32x4{p_pack2(v0[s][0], v0[s][1]), p_pack2(v0[s][2], v0[s][3]), p_pack2(v1[s][0], v1[s][1]), p_pack2(v1[s][2], v1[s][3])};      // f16
    }
    __syncthreads();                                          // biases visible; no DMA is in flight yet
    const u32x4* img = reinterpret_cast<const u32x4*>(w.img);
    chain_issue(img, ring, 0, tid);
    if (NCH_T > 1) chain_issue(img, ring, 1, tid);
    if (NCH_T > 2) chain_issue(img, ring, 2, tid);
    float* yrow = Y + (size_t)row * ldy;
    chain_layer<K0 / 16, HB, 0, NCH_T, false>(img, ring, tid, lane, h, a, b, bias_lds, yrow, n_valid, row_ok);
    if (NH == 0) {
        chain_layer<HK, OB, NCH0, NCH_T, true>(img, ring, tid, lane, h, b, a, bias_lds + H, yrow, n_valid, row_ok);
    } else if (NH == 1) {
        chain_layer<HK, HB, NCH0, NCH_T, false>(img, ring, tid, lane, h, b, a, bias_lds + H, yrow, n_valid, row_ok);
        chain_layer<HK, OB, NCH0 + NCHH, NCH_T, true>(img, ring, tid, lane, h, a, b, bias_lds + 2 * H, yrow, n_valid, row_ok);
    } else {
        chain_layer<HK, HB, NCH0, NCH_T, false>(img, ring, tid, lane, h, b, a, bias_lds + H, yrow, n_valid, row_ok);
        chain_layer<HK, HB, NCH0 + NCHH, NCH_T, false>(img, ring, tid, lane, h, a, b, bias_lds + 2 * H, yrow, n_valid, row_ok);
        chain_layer<HK, OB, NCH0 + 2 * NCHH, NCH_T, true>(img, ring, tid, lane, h, b, a, bias_lds + 3 * H, yrow, n_valid, row_ok);
    }
}

// chain weight image: layer with K inputs, N outputs (rows >= n_real are zero): chunks [ob][ks][lane][8];
// first layer: k natural (16 ks + 8h + j); later layers: ks = 2mb + s' <-> channel 32mb + 16h + 8s' + j
__global__ void k_build_chain_image(const float* __restrict__ wraw, int K_real, int K, int N, int n_real, int first, float wscale,
                                    bf16_t* __restrict__ dst) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    int nks = K / 16;
    if (id >= (N / 32) * nks * 64 * 8) return;
    int j = id & 7, lane = (id >> 3) & 63, f = id >> 9;
    int ob = f / nks, ks = f % nks, r = lane & 31, h = lane >> 5;
    int row = ch_nat(ob, r);
    int col = first ? 16 * ks + 8 * h + j : 32 * (ks >> 1) + 16 * h + 8 * (ks & 1) + j;
    const float wv = (row < n_real && col < K_real) ? wraw[(size_t)row * K_real + col] : 0.f;
    dst[id] = __builtin_bit_cast(bf16_t, (_Float16)(wscale * wv));      // every layer consumes f16 operands (the first one too since round 4: the input rows keep 11
                                                             // significand bits instead of 8 - see tools/tap_errors.py: the GraphNorm behind the node stacks amplifies input rounding)
}
// position: 0 first Linear of the chain, 1 hidden, 2 last (scaled activation domain of the chain's GELU: first x a, last / a)
void launch_build_chain_image(const float* wraw, int K_real, int K, int N, int n_real, int position, bf16_t* dst, hipStream_t s) {
    int total = (N / 32) * (K / 16) * 512;
    const float wscale = CH_GELU == 1 ? (position == 0 ? kGAn : position == 2 ? kGAni : 1.f) : 1.f;
    hipLaunchKernelGGL(k_build_chain_image, dim3((total + 255) / 256), dim3(256), 0, s, wraw, K_real, K, N, n_real, position == 0 ? 1 : 0, wscale, dst);
}

// returns 0 when the (K0, H, NH, NOUT) shape has a fused kernel, 1 otherwise (caller falls back to GEMMs)
int launch_ffn_chain(const int* ntot, int mmax, const float* X, int ldx, const float* X2, int ldx2, int K0, int H, int NH,
                     int NOUT, const bf16_t* img, const float* const* bias, float* Y, int ldy, int n_valid, hipStream_t s) {
    ChainW w;
    w.img = img;
    for (int i = 0; i < 5; ++i) w.bias[i] = i < NH + 2 ? bias[i] : nullptr;
    dim3 grid((mmax + 127) / 128);
    const size_t lds = CH_RING * 32768 + (size_t)((NH + 1) * H + NOUT) * sizeof(float);
#define RN_CHAIN(k0, hh, nh, no) \
    if (K0 == k0 && H == hh && NH == nh && NOUT == no) { \
        static DevAttr attr; \
        ensure_dyn_lds((const void*)k_ffn_chain<k0, hh, nh, no>, lds, attr); \
        hipLaunchKernelGGL((k_ffn_chain<k0, hh, nh, no>), grid, dim3(256), lds, s, ntot, X, ldx, X2, ldx2, w, Y, ldy, n_valid); return 0; }
    RN_CHAIN(128, 512, 2, 128)
    RN_CHAIN(32, 512, 2, 128)
    RN_CHAIN(256, 512, 0, 32)
#undef RN_CHAIN
    return 1;
}

// ------------------------------------------------------------------------------------------
// Fused node update between two ResMPNN steps (mpnn.py:222-225, 289; functional.py:33-46):
//   x = h + agg  ->  GraphNormalization  ->  h'  ->  [P | Q] = h' . [Wa | Wb]^T (+ b1) for up to two
// first Linears (edge MLP of this layer, message MLP of the next).  k_gn_coef reduces the per-RNA
// statistics to an affine map y = a x + b per (RNA, channel); k_node_update applies it to 32-row
// blocks held as B fragments, writes h' (f32) and runs the projections on MFMA with the
// [P | Q] weight images resident in LDS (64 KiB per job): P -> f32, Q -> bf16 (gather tables of
// the edge kernel).
__global__ void __launch_bounds__(256) k_gn_coef(PackInfo pk, const float* __restrict__ x, const float* __restrict__ add,
        const float* __restrict__ scale, const float* __restrict__ shift, int t_tot, float* __restrict__ coef) {
    // TWO-PASS statistics (mean, then sum of squared deviations): the one-pass form sum x^2 + (T - 2n) mu^2 cancels once |h| >> sigma
    __shared__ float4 red[4][8];
    const int b = blockIdx.x, cg = blockIdx.y;                 // RNA, group of 32 channels
    const int n = pk.len[b];
    if (n <= 0) return;
    const size_t base = (size_t)pk.cu[b] * RN_D + 32 * cg;
    const int cq = threadIdx.x & 7, g = threadIdx.x >> 3;      // channel quad, row group (32)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float4* xb = reinterpret_cast<const float4*>(x + base) + cq;
    const float4* ab = add ? reinterpret_cast<const float4*>(add + base) + cq : nullptr;
    // reduce a per-thread float4 over the 32 row groups: 4 lanes-of-8 per wave via shuffles, then the 4 waves via LDS (fixed order)
    auto block_sum = [&](float4 s) -> float4 {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
        }
        __syncthreads();                                       // red[] of a previous call has been consumed
        if (lane < 8) red[wave][lane] = s;
        __syncthreads();
        float4 t = red[0][cq];
#pragma unroll
        for (int w = 1; w < 4; ++w) { const float4 u = red[w][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        return t;
    };
    const float fn = (float)n;
    float4 S, Q;
    if (n <= 160) {         // rows stay in registers between the two passes (all loads of the thread in flight together)
        float4 v[5], a[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) { const int r = g + 32 * i, rc = r < n ? r : n - 1; v[i] = xb[(size_t)rc * 32]; }
        if (ab) {
#pragma unroll
            for (int i = 0; i < 5; ++i) { const int r = g + 32 * i, rc = r < n ? r : n - 1; a[i] = ab[(size_t)rc * 32]; }
#pragma unroll
            for (int i = 0; i < 5; ++i) { v[i].x += a[i].x; v[i].y += a[i].y; v[i].z += a[i].z; v[i].w += a[i].w; }
        }
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 5; ++i) if (g + 32 * i < n) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
        S = block_sum(s);
        const float4 mu = make_float4(S.x / fn, S.y / fn, S.z / fn, S.w / fn);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 5; ++i) if (g + 32 * i < n) {
            const float dx = v[i].x - mu.x, dy = v[i].y - mu.y, dz = v[i].z - mu.z, dw = v[i].w - mu.w;
            q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
        }
        Q = block_sum(q);
    } else if (n <= 256) {  // (C4: 200 nt) the same with eight rows per thread
        float4 v[8], a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int r = g + 32 * i, rc = r < n ? r : n - 1; v[i] = xb[(size_t)rc * 32]; }
        if (ab) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int r = g + 32 * i, rc = r < n ? r : n - 1; a[i] = ab[(size_t)rc * 32]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i].x += a[i].x; v[i].y += a[i].y; v[i].z += a[i].z; v[i].w += a[i].w; }
        }
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) if (g + 32 * i < n) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
        S = block_sum(s);
        const float4 mu = make_float4(S.x / fn, S.y / fn, S.z / fn, S.w / fn);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) if (g + 32 * i < n) {
            const float dx = v[i].x - mu.x, dy = v[i].y - mu.y, dz = v[i].z - mu.z, dw = v[i].w - mu.w;
            q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
        }
        Q = block_sum(q);
    } else {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = g; r < n; r += 32) {
            float4 v = xb[(size_t)r * 32];
            if (ab) { const float4 a = ab[(size_t)r * 32]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        S = block_sum(s);
        const float4 mu = make_float4(S.x / fn, S.y / fn, S.z / fn, S.w / fn);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = g; r < n; r += 32) {
            float4 v = xb[(size_t)r * 32];
            if (ab) { const float4 a = ab[(size_t)r * 32]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
            const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
            q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
        }
        Q = block_sum(q);
    }
    if (threadIdx.x < 8) {
        const float pad = (float)(t_tot - n);
        const float4 sc = reinterpret_cast<const float4*>(scale + 32 * cg)[cq], sh = reinterpret_cast<const float4*>(shift + 32 * cg)[cq];
        const float mean[4] = {S.x / fn, S.y / fn, S.z / fn, S.w / fn};
        const float sq[4] = {Q.x, Q.y, Q.z, Q.w}, scl[4] = {sc.x, sc.y, sc.z, sc.w}, shf[4] = {sh.x, sh.y, sh.z, sh.w};
        float a[4], bb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // var = [sum_valid (x - mu)^2 + (T - n) mu^2] / n : padded rows enter as (0 - mu)^2 (functional.py:33-38)
            const float var = (sq[i] + pad * mean[i] * mean[i]) / fn;
            a[i] = scl[i] / sqrtf(var + kSEPS);
            bb[i] = shf[i] - mean[i] * a[i];
        }
        float* cb = coef + (size_t)b * 256 + 32 * cg + 4 * cq;
        *reinterpret_cast<float4*>(cb) = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(cb + 128) = make_float4(bb[0], bb[1], bb[2], bb[3]);
    }
}

typedef NodeJob PqJob;     // img: [8 ob][8 ks][64][8]; ob < 4 -> P rows, >= 4 -> Q rows

template <int NJOBS>
__global__ void __launch_bounds__(256, 2) k_node_update(PackInfo pk, const float* __restrict__ x, const float* __restrict__ add,
        const float* __restrict__ coef, float* __restrict__ h_out, PqJob j0, PqJob j1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* img = reinterpret_cast<u32x4*>(smem);
    const int ntot = pk.cu[pk.B];
    const int row_blk = blockIdx.x * 128;
    if (row_blk >= ntot) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int row = row_blk + 32 * wave + r;
    const bool ok = row < ntot;
    const int rr = ok ? row : 0;
    const float* cf = coef + (size_t)pk.node_b[rr] * 256;
    // the HBM loads of the rows go out first; the weight images (L2-resident) are staged into LDS while they fly
    f32x4 vx[8][2], va[8][2];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int c0 = 16 * s + 8 * h;
        vx[s][0] = *reinterpret_cast<const f32x4*>(x + (size_t)rr * RN_D + c0);
        vx[s][1] = *reinterpret_cast<const f32x4*>(x + (size_t)rr * RN_D + c0 + 4);
        if (add) {
            va[s][0] = *reinterpret_cast<const f32x4*>(add + (size_t)rr * RN_D + c0);
            va[s][1] = *reinterpret_cast<const f32x4*>(add + (size_t)rr * RN_D + c0 + 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_image_dma<256>(img, reinterpret_cast<const u32x4*>(j0.img), tid);
    if (NJOBS > 1) stage_image_dma<256>(img + 4096, reinterpret_cast<const u32x4*>(j1.img), tid);
    __builtin_amdgcn_sched_barrier(0);
    // the biases of the P halves go through LDS too (a global read per channel block would expose an L2 round trip each)
    float* lds_bias = reinterpret_cast<float*>(smem + NJOBS * 65536);
    if (tid < 128) lds_bias[tid] = j0.bias[tid];
    else if (NJOBS > 1) lds_bias[tid] = j1.bias[tid - 128];
    // residual, then GraphNorm as one branch-free pass (its 32 coefficient loads go out together), then stores + bf16 pack
    if (add) {
#pragma unroll
        for (int s = 0; s < 8; ++s) { vx[s][0] += va[s][0]; vx[s][1] += va[s][1]; }
    }
    if (coef) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int c0 = 16 * s + 8 * h;
            const f32x4 ca0 = *reinterpret_cast<const f32x4*>(cf + c0), ca1 = *reinterpret_cast<const f32x4*>(cf + c0 + 4);
            const f32x4 cb0 = *reinterpret_cast<const f32x4*>(cf + 128 + c0), cb1 = *reinterpret_cast<const f32x4*>(cf + 128 + c0 + 4);
            vx[s][0] = vx[s][0] * ca0 + cb0; vx[s][1] = vx[s][1] * ca1 + cb1;
        }
    }
    u32x4 xf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const f32x4 v0 = vx[s][0], v1 = vx[s][1];
        xf[s] = u32x4{p_pack2(v0[0], v0[1]), p_pack2(v0[2], v0[3]), p_pack2(v1[0], v1[1]), p_pack2(v1[2], v1[3])};      // f16 rows, f16 images
    }
    __builtin_amdgcn_sched_barrier(0);
    dma_landed();          // (every row / coefficient load above has been consumed: only the image pieces can still be in flight; the h_out
    __builtin_amdgcn_sched_barrier(0);     //  stores below then drain under the matrix work)
    if (ok && h_out) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int c0 = 16 * s + 8 * h;
            *reinterpret_cast<f32x4*>(h_out + (size_t)row * RN_D + c0) = vx[s][0];
            *reinterpret_cast<f32x4*>(h_out + (size_t)row * RN_D + c0 + 4) = vx[s][1];
        }
    }
    __syncthreads();
#pragma unroll
    for (int jb = 0; jb < NJOBS; ++jb) {
        const PqJob& jbq = jb == 0 ? j0 : j1;
        const u32x4* im = img + jb * 4096;
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) {
            f32x16 acc;
            if (ob < 4) acc = init_vec16(lds_bias + jb * 128 + 32 * ob + 16 * h);
            else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = mfma32h(im[(ob * 8 + ks) * 64 + lane], xf[ks], acc);
            if (ok) {
                // both tables leave as f16 in NATURAL channel order: register i of lane half h holds channel 32 ob' + 16 h + i of its block (rows in
                // ch_nat order), or 32 ob' + 8 h + i / 32 ob' + 16 + 8 h + (i - 8) (P rows of a depth-1 edge MLP: ch_efrag order) - two 16-byte stores
                bf16_t* row_p = (ob < 4 ? jbq.p : jbq.q) + (size_t)row * RN_D + 32 * (ob & 3);
                const bool efr = ob < 4 && jbq.p_efrag;
                u32x4* d0 = reinterpret_cast<u32x4*>(row_p + (efr ? 8 * h : 16 * h));
                u32x4* d1 = reinterpret_cast<u32x4*>(row_p + (efr ? 16 + 8 * h : 16 * h + 8));
                *d0 = u32x4{p_pack2(acc[0], acc[1]), p_pack2(acc[2], acc[3]), p_pack2(acc[4], acc[5]), p_pack2(acc[6], acc[7])};
                *d1 = u32x4{p_pack2(acc[8], acc[9]), p_pack2(acc[10], acc[11]), p_pack2(acc[12], acc[13]), p_pack2(acc[14], acc[15])};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the same update with the GraphNormalization statistics computed IN the kernel: one workgroup per RNA (n <= 32 NW rows), every row of the RNA
// in the registers of its lanes.  Per-channel sums over the rows: DPP tree over the 32 lanes of a half-wave (fixed order), the waves' partials
// combined through LDS by 128 threads in wave order - two passes (mean, then squared deviations: functional.py:33-38 and the note in k_gn_coef).
// Replaces the k_gn_coef launch in front of k_node_update (8 us of a 2.2 ms C2 forward, ten times) when the batch's padded length allows it.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {      // (bound_ctrl with the full row mask: lets the compiler fold the move into v_add_f32_dpp)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf));
}
// Sums over the 32 lanes of a half-wave, two values at a time: v_permlane16_swap_b32 exchanges the odd 16-lane rows of `a` with the even rows of
// `b`, so a' + b' holds a[r] + a[r + 16] on the even rows and b[r] + b[r + 16] on the odd rows - half the values, each on half the lanes; four
// butterfly steps inside the rows finish both.  Result: the total of `a` on lanes 0..15 of the half-wave, of `b` on lanes 16..31 (fixed order).
__device__ __forceinline__ float half_wave_sum2(float a, float b) {
    // (inline asm: this compiler folds the builtin's two results into one register - tools/ubench/permlane_probe.hip; the s_nops are the
    //  VALU-write -> permlane-read and permlane-write -> DPP-read wait states the compiler would have inserted)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float v = a + b;
    v += dpp_f32<0xB1, 0xf>(v);        // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E, 0xf>(v);        // quad_perm [2,3,0,1]
    v += dpp_f32<0x141, 0xf>(v);       // row_half_mirror
    v += dpp_f32<0x140, 0xf>(v);       // row_mirror
    return v;
}
// NB row blocks of 32 residues per workgroup, one wave each.  (Two waves per row block - one job / one P-or-Q half each, 2 NB waves over the four
// SIMDs - measured slower at the C2 lengths: 27.3 against 24.7 us; twice the row loads and ten waves at the barriers cost more than the better
// spread of the MFMAs gains.)
template <int NJOBS, int NB>
__global__ void __launch_bounds__(NB * 64) k_node_update_rna(PackInfo pk, const float* __restrict__ x, const float* __restrict__ add,
        const float* __restrict__ scale, const float* __restrict__ shift, int t_tot, float* __restrict__ h_out, PqJob j0, PqJob j1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* img = reinterpret_cast<u32x4*>(smem);
    float* lds_bias = reinterpret_cast<float*>(smem + NJOBS * 65536);      // [2][128]
    constexpr int NW = NB;
    float* lds_part = lds_bias + 256;                                       // [NB][128] per-row-block partial sums
    float* lds_mean = lds_part + NB * 128;                                  // [128]
    float* lds_ab = lds_mean + 128;                                         // [128] a | [128] b
    const int b = blockIdx.x;
    const int n = pk.len[b];
    if (n <= 0) return;
    const int base = pk.cu[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int rb = wave;                                     // row block
    const int lrow = 32 * rb + r;
    const bool ok = lrow < n;
    const bool own = ok;
    const int row = base + (ok ? lrow : 0);
    f32x4 vx[8][2], va[8][2];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int c0 = 16 * s + 8 * h;
        vx[s][0] = *reinterpret_cast<const f32x4*>(x + (size_t)row * RN_D + c0);
        vx[s][1] = *reinterpret_cast<const f32x4*>(x + (size_t)row * RN_D + c0 + 4);
        if (add) {
            va[s][0] = *reinterpret_cast<const f32x4*>(add + (size_t)row * RN_D + c0);
            va[s][1] = *reinterpret_cast<const f32x4*>(add + (size_t)row * RN_D + c0 + 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_image_dma<NW * 64>(img, reinterpret_cast<const u32x4*>(j0.img), tid);
    if (NJOBS > 1) stage_image_dma<NW * 64>(img + 4096, reinterpret_cast<const u32x4*>(j1.img), tid);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < 128) lds_bias[tid] = j0.bias[tid];
    else if (NJOBS > 1 && tid < 256) lds_bias[tid] = j1.bias[tid - 128];
    float sc_c = 0.f, sh_c = 0.f;
    if (tid < 128) { sc_c = scale[tid]; sh_c = shift[tid]; }
    if (add) {
#pragma unroll
        for (int s = 0; s < 8; ++s) { vx[s][0] += va[s][0]; vx[s][1] += va[s][1]; }
    }
    const int nwb = (n + 31) >> 5;                         // waves that hold rows
    auto reduce_rows = [&](auto&& value) {                 // value(s, p, i) of this lane -> lds_part[wave][channel]
#pragma unroll
        for (int s = 0; s < 4; ++s) {                      // lane r = 0 of a half ends with channel block s, lane r = 16 with block s + 4
            f32x4 t0, t1;
#pragma unroll
            for (int i = 0; i < 4; ++i) { t0[i] = half_wave_sum2(value(s, 0, i), value(s + 4, 0, i)); t1[i] = half_wave_sum2(value(s, 1, i), value(s + 4, 1, i)); }
            if ((r & 15) == 0) {
                const int sb = s + (r >> 4) * 4;
                *reinterpret_cast<f32x4*>(lds_part + rb * 128 + 16 * sb + 8 * h) = t0;
                *reinterpret_cast<f32x4*>(lds_part + rb * 128 + 16 * sb + 8 * h + 4) = t1;
            }
        }
    };
    auto combine = [&]() -> float {                        // thread c < 128: the waves' partials of channel c, in wave order
        float t = lds_part[tid];
        for (int w = 1; w < nwb; ++w) t += lds_part[w * 128 + tid];
        return t;
    };
    const float fn = (float)n;
    // pass 1: mean
    reduce_rows([&](int s, int p, int i) { return own ? vx[s][p][i] : 0.f; });
    __syncthreads();
    float mean_c = 0.f;
    if (tid < 128) { mean_c = combine() / fn; lds_mean[tid] = mean_c; }
    __syncthreads();
    f32x4 mu[8][2];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        mu[s][0] = *reinterpret_cast<const f32x4*>(lds_mean + 16 * s + 8 * h);
        mu[s][1] = *reinterpret_cast<const f32x4*>(lds_mean + 16 * s + 8 * h + 4);
    }
    // pass 2: squared deviations of the valid rows; the T - n padded rows enter as (0 - mean)^2
    reduce_rows([&](int s, int p, int i) { const float d = vx[s][p][i] - mu[s][p][i]; return own ? d * d : 0.f; });
    __syncthreads();
    if (tid < 128) {
        const float sq = combine(), pad = (float)(t_tot - n);
        const float var = (sq + pad * mean_c * mean_c) / fn;
        const float a = sc_c / sqrtf(var + kSEPS);
        lds_ab[tid] = a;
        lds_ab[128 + tid] = sh_c - mean_c * a;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int c0 = 16 * s + 8 * h;
        const f32x4 ca0 = *reinterpret_cast<const f32x4*>(lds_ab + c0), ca1 = *reinterpret_cast<const f32x4*>(lds_ab + c0 + 4);
        const f32x4 cb0 = *reinterpret_cast<const f32x4*>(lds_ab + 128 + c0), cb1 = *reinterpret_cast<const f32x4*>(lds_ab + 128 + c0 + 4);
        vx[s][0] = vx[s][0] * ca0 + cb0; vx[s][1] = vx[s][1] * ca1 + cb1;
    }
    u32x4 xf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const f32x4 v0 = vx[s][0], v1 = vx[s][1];
        xf[s] = u32x4{p_pack2(v0[0], v0[1]), p_pack2(v0[2], v0[3]), p_pack2(v1[0], v1[1]), p_pack2(v1[2], v1[3])};
    }
    __builtin_amdgcn_sched_barrier(0);
    dma_landed();
    __builtin_amdgcn_sched_barrier(0);
    if (own && h_out) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int c0 = 16 * s + 8 * h;
            *reinterpret_cast<f32x4*>(h_out + (size_t)row * RN_D + c0) = vx[s][0];
            *reinterpret_cast<f32x4*>(h_out + (size_t)row * RN_D + c0 + 4) = vx[s][1];
        }
    }
    __syncthreads();
    if (rb >= nwb) return;                                  // (no barrier below)
#pragma unroll
    for (int jb = 0; jb < NJOBS; ++jb) {
        const PqJob& jbq = jb == 0 ? j0 : j1;
        const u32x4* im = img + jb * 4096;
#pragma unroll
        for (int ob = 0; ob < 8; ++ob) {
            f32x16 acc;
            if (ob < 4) acc = init_vec16(lds_bias + jb * 128 + 32 * ob + 16 * h);
            else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = mfma32h(im[(ob * 8 + ks) * 64 + lane], xf[ks], acc);
            if (ok) {
                bf16_t* row_p = (ob < 4 ? jbq.p : jbq.q) + (size_t)row * RN_D + 32 * (ob & 3);
                const bool efr = ob < 4 && jbq.p_efrag;
                u32x4* d0 = reinterpret_cast<u32x4*>(row_p + (efr ? 8 * h : 16 * h));
                u32x4* d1 = reinterpret_cast<u32x4*>(row_p + (efr ? 16 + 8 * h : 16 * h + 8));
                *d0 = u32x4{p_pack2(acc[0], acc[1]), p_pack2(acc[2], acc[3]), p_pack2(acc[4], acc[5]), p_pack2(acc[6], acc[7])};
                *d1 = u32x4{p_pack2(acc[8], acc[9]), p_pack2(acc[10], acc[11]), p_pack2(acc[12], acc[13]), p_pack2(acc[14], acc[15])};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// [P | Q] image of one MLP's first Linear w0 [128][384]: ob < 4 rows ch_nat(ob) of w0[:, 0:128], ob >= 4 of w0[:, 128:256].
// efrag = 1 (depth-1 edge MLP, EDGE1 of the fused kernel): the P rows - hence the P words k_node_update writes - follow ch_efrag,
// the order of that kernel's one accumulator tile; b1p = the bias in the matching order (b1p[32 ob + 16 h + i] <-> accumulator
// register i of lane half h).  The Q rows stay in natural channel order (the table is gathered as stored).
__global__ void k_build_pq_image(const float* __restrict__ w0, const float* __restrict__ b1, int efrag, bf16_t* __restrict__ dst,
                                 float* __restrict__ b1p) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 8 * 8 * 64 * 8) return;
    int j = id & 7, lane = (id >> 3) & 63, f = id >> 9, ob = f >> 3, ks = f & 7, r = lane & 31, h = lane >> 5;
    int row = (efrag && ob < 4) ? ch_efrag(ob, r) : ch_nat(ob & 3, r), col = (ob < 4 ? 0 : 128) + 16 * ks + 8 * h + j;
    dst[id] = __builtin_bit_cast(bf16_t, (_Float16)(kGA * w0[(size_t)row * 384 + col]));          // f16 (scaled domain: the tables hold a P and a Q)
    if (id < 128) {
        int ob2 = id >> 5, m = id & 31, hh = (m >> 2) & 1, i = (m & 3) + 4 * (m >> 3);
        b1p[32 * ob2 + 16 * hh + i] = kGA * b1[efrag ? ch_efrag(ob2, m) : ch_nat(ob2, m)];
    }
}
void launch_build_pq_image(const float* w0, const float* b1, int efrag, bf16_t* dst, float* b1p, hipStream_t s) {
    hipLaunchKernelGGL(k_build_pq_image, dim3(8 * 8 * 64 * 8 / 256), dim3(256), 0, s, w0, b1, efrag, dst, b1p);
}

void launch_node_update(const PackInfo& pk, const float* x, const float* add, const float* scale, const float* shift, int t_tot,
                        float* coef, float* h_out, int njobs, const NodeJob& job0, const NodeJob& job1, hipStream_t s) {
    const PqJob j0 = job0, j1 = njobs > 1 ? job1 : NodeJob{};
    // Every RNA of the batch fits the rows of one workgroup (padded length <= 256) and is long enough for a workgroup of its own: the statistics
    // are computed inside the update kernel (RNAMPNN_NODE_UPDATE_RNA=0: the two-launch form; read per call)
    static const auto rna_form_env = []() { const char* v = getenv("RNAMPNN_NODE_UPDATE_RNA"); return !(v && v[0] == '0'); };
    if (scale && pk.T >= 48 && pk.T <= 256 && rna_form_env()) {
#define NU_RNA(J, NBK)                                                                                                          \
        do {                                                                                                                    \
            static DevAttr attr;                                                                                                \
            constexpr size_t lds = (size_t)(J) * 65536 + 1024 + (NBK) * 512 + 3 * 512;                                          \
            ensure_dyn_lds((const void*)k_node_update_rna<J, NBK>, lds, attr);                                                  \
            hipLaunchKernelGGL((k_node_update_rna<J, NBK>), dim3(pk.B), dim3((NBK) * 64), lds, s, pk, x, add, scale, shift, t_tot, h_out, j0, j1); \
        } while (0)
        if (njobs == 1) { if (pk.T <= 128) NU_RNA(1, 4); else if (pk.T <= 160) NU_RNA(1, 5); else NU_RNA(1, 8); }
        else { if (pk.T <= 128) NU_RNA(2, 4); else if (pk.T <= 160) NU_RNA(2, 5); else NU_RNA(2, 8); }
#undef NU_RNA
        return;
    }
    if (scale) hipLaunchKernelGGL(k_gn_coef, dim3(pk.B, 4), dim3(256), 0, s, pk, x, add, scale, shift, t_tot, coef);
    dim3 grid((pk.Nmax + 127) / 128);
    // Measured and NOT kept (round 3, C2, 21.9 us per launch for this form): (a) four workgroups per 128-row block (one per job and P / Q half,
    // 32 KiB of LDS, four per CU): 28.3 us; (b) every global access coalesced through a per-wave LDS tile (1 KiB row loads, 128-byte-line
    // P / Q stores): 26.2 us.  Phase stamps (NU_STAMPS) explain both: the launch moves 16 MB in and 63 MB out (h' f32, two P tables of
    // (hi, lo) words, two Q tables) - the load phase ends when the LAST row of the chip-wide 16 MB burst has arrived, the MFMA + store phase
    // runs at the rate the 63 MB drain; neither the access shape nor the occupancy is what bounds them.
    static DevAttr attr1, attr2;
    ensure_dyn_lds((const void*)k_node_update<1>, 65536 + 1024, attr1);
    ensure_dyn_lds((const void*)k_node_update<2>, 131072 + 1024, attr2);
    const float* cf = scale ? coef : nullptr;
    if (njobs == 1) hipLaunchKernelGGL(k_node_update<1>, grid, dim3(256), 65536 + 1024, s, pk, x, add, cf, h_out, j0, j1);
    else hipLaunchKernelGGL(k_node_update<2>, grid, dim3(256), 131072 + 1024, s, pk, x, add, cf, h_out, j0, j1);
}

// ------------------------------------------------------------------------------------------
// nn.MultiheadAttention over the valid keys of one RNA on MFMA (functional.py:164-168), head dim 16.
// One workgroup per (RNA, head, 256-query slab); wave w owns queries 32w .. 32w+31 of the slab.
// Transposed like everything else here: S^T[key][query] = K . Q^T puts ONE query on each lane and the
// keys of a 32-key block on the accumulator registers, so the online softmax is in-lane (+ one
// cross-half exchange), and the P tile is, as it stands, the B operand of O^T[d][query] += V^T . P.
// K rows and the (k-permuted) V^T image sit in LDS as ready fragments, staged in CHUNKS of at most `chunk_blocks`
// 32-key blocks (2 KiB each): any RNA length runs through the same kernel (the reference pads to 4,500,
// functional.py:153-159); the softmax state of a wave's one query tile lives in registers across the chunks.
__global__ void __launch_bounds__(512) k_attention_bf16_hd16(PackInfo pk, const float* __restrict__ qkv, float* __restrict__ out,
                                                             int chunk_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, hd = blockIdx.y;
    const int n = pk.len[b];
    const int qbase = blockIdx.z * 256;
    if (n <= 0 || qbase >= n) return;                          // (whole workgroup: uniform)
    const int base = pk.cu[b];
    const int nkb = (n + 31) / 32;
    const int chb = min(chunk_blocks, nkb);
    u32x4* Kimg = reinterpret_cast<u32x4*>(smem);              // [key][2 halves] : 8 bf16 each
    u32x4* Vt = Kimg + (size_t)chb * 64;                       // [kb][s][h][d 0..15] : 8 permuted keys each
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int q0 = qbase + 32 * wave;
    const bool wave_live = q0 < n;                             // wave-uniform; dead waves still stage and hit the barriers
    const int qi = q0 + r;
    u32x4 qf = zero4;
    if (qi < n) {
        const float* qp = qkv + (size_t)(base + qi) * 384 + hd * 16 + 8 * h;
        f32x4 a = *reinterpret_cast<const f32x4*>(qp), c = *reinterpret_cast<const f32x4*>(qp + 4);
        qf = u32x4{pack2(0.25f * a[0], 0.25f * a[1]), pack2(0.25f * a[2], 0.25f * a[3]),
                   pack2(0.25f * c[0], 0.25f * c[1]), pack2(0.25f * c[2], 0.25f * c[3])};
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float m_run = -3.0e38f, l_run = 0.f;
    for (int kb0 = 0; kb0 < nkb; kb0 += chb) {
        const int cb = min(chb, nkb - kb0);                    // blocks in this chunk
        if (kb0 > 0) __syncthreads();                          // the previous chunk has been consumed
        for (int idx = tid; idx < cb * 64; idx += 512) {       // K rows
            const int key = 32 * kb0 + (idx >> 1), hh = idx & 1;
            // (rows clamped and the result masked: a conditional load per element would serialise into one memory round trip each)
            const float* kp = qkv + (size_t)(base + (key < n ? key : n - 1)) * 384 + 128 + hd * 16 + 8 * hh;
            const f32x4 a = *reinterpret_cast<const f32x4*>(kp), c = *reinterpret_cast<const f32x4*>(kp + 4);
            const u32x4 v = u32x4{pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(c[0], c[1]), pack2(c[2], c[3])};
            Kimg[idx] = key < n ? v : zero4;
        }
        for (int idx = tid; idx < cb * 64; idx += 512) {       // V^T fragments
            const int d = idx & 15, hh = (idx >> 4) & 1, sblk = (idx >> 5) & 1, kb = kb0 + (idx >> 6);
            float vals[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = 32 * kb + 16 * sblk + 8 * (j >> 2) + 4 * hh + (j & 3);
                vals[j] = qkv[(size_t)(base + (key < n ? key : n - 1)) * 384 + 256 + hd * 16 + d];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = 32 * kb + 16 * sblk + 8 * (j >> 2) + 4 * hh + (j & 3);
                if (key >= n) vals[j] = 0.f;
            }
            Vt[idx] = u32x4{pack2(vals[0], vals[1]), pack2(vals[2], vals[3]), pack2(vals[4], vals[5]), pack2(vals[6], vals[7])};
        }
        __syncthreads();
        if (wave_live) {
            for (int kl = 0; kl < cb; ++kl) {
                const int kb = kb0 + kl;
                f32x16 sc;
#pragma unroll
                for (int i = 0; i < 16; ++i) sc[i] = 0.f;
                sc = mfma32(Kimg[(32 * kl + r) * 2 + h], qf, sc);                  // S^T[key][query]
                float mx = -3.0e38f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    sc[i] = key < n ? sc[i] : -3.0e38f;
                    mx = fmaxf(mx, sc[i]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run, mx);
                const float corr = __expf(m_run - m_new);
                float ps = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) { sc[i] = __expf(sc[i] - m_new); ps += sc[i]; }
                l_run = l_run * corr + ps;
                m_run = m_new;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] *= corr;
#pragma unroll
                for (int sblk = 0; sblk < 2; ++sblk) {
                    const u32x4 pf = {pack2(sc[8 * sblk], sc[8 * sblk + 1]), pack2(sc[8 * sblk + 2], sc[8 * sblk + 3]),
                                      pack2(sc[8 * sblk + 4], sc[8 * sblk + 5]), pack2(sc[8 * sblk + 6], sc[8 * sblk + 7])};
                    const u32x4 vf = r < 16 ? Vt[((kl * 2 + sblk) * 2 + h) * 16 + r] : zero4;
                    acc = mfma32(vf, pf, acc);                                   // O^T[d][query]
                }
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (qi < n) {
        const float inv = 1.0f / l_tot;
        float* op = out + (size_t)(base + qi) * RN_D + hd * 16 + 4 * h;      // rows d = (i&3) + 8(i>>2) + 4h, i < 8
        *reinterpret_cast<f32x4*>(op) = f32x4{acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
        *reinterpret_cast<f32x4*>(op + 8) = f32x4{acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv};
    }
}

// ------------------------------------------------------------------------------------------
// One attention layer of RNABert (functional.py:161-169) as ONE kernel per RNA, for RNAs of at most AL_NR residues:
//     x <- GraphNorm_P( x + out_proj( MHA(x, x, x; valid keys) ) )
// replaces four launches (QKV GEMM, attention, out-proj GEMM + residual, GraphNorm) and three HBM round trips of [N][384] /
// [N][128] f32 tensors by one pass: x rows in, x rows out (in place: a workgroup touches only its own RNA).  512 threads:
//   phase 0  x -> bf16 B-fragment image in LDS                       (rows of the batch on the MFMA columns, as everywhere here)
//   phase 1  [Q | K | V]^T = Wqkv . x^T + b, 32 x 32 tiles over the 8 waves.  The accumulator tile of a Q (K) block, packed to bf16, IS
//            the B (A) operand of S^T = K . Q^T for its two heads - the d order inside a head is permuted identically on both sides
//            (lane half h holds d = {4h..4h+3, 8+4h..8+4h+3}) - so Q and K go to LDS as ready fragments; V is stored transposed
//            ([channel][key]) so that a V^T fragment is two 8-byte reads
//   phase 2  wave = head: online softmax over the valid keys in the accumulator layout (one query per lane), O^T += V^T . P; the
//            normalised O^T tile is again, as it stands, the B fragment of the out-projection's k-step `head` (same d permutation,
//            applied to the weight fragment when it is loaded)
//   phase 3  y = Wout . O^T + b + x   -> f32 rows in LDS
//   phase 4  GraphNormalization over the RNA (two-pass statistics, padded rows enter through (P - n) mu^2), x rows out.
// Head dim 16, 8 heads.  LDS: [A 40,960: x image, later O image | red][BC 76,032: Q, K fragments, later y][D 40,960: V^T][E: biases, coef].
#define AL_NR 144
#define AL_KS 160
#define AL_YLD 132
#define AL_LDS (40960 + 76032 + 40960 + 3072)
// Fragment images of the two weights (built once per load_state_dict): 1 KiB per MFMA A fragment, lane-linear, so a wave's fragment load is
// one coalesced 1 KiB request (row-major weights cost 64 separate 16-byte accesses per load: the QKV phase was TA-bound on them).
//   qkv [cb 0..11][s 0..7][lane][8]: W[32 cb + r][16 s + 8 h + j];   out [cb 0..3][s 0..7][lane][8]: Wout[32 cb + r][16 s + d(h, j)],
//   d(h, j) = j < 4 ? 4 h + j : 8 + 4 h + (j - 4) - the d order in which the attention phase leaves O^T (see the kernel).
__global__ void k_build_attn_images(const float* __restrict__ wqkv, const float* __restrict__ wout, bf16_t* __restrict__ img_qkv,
                                    bf16_t* __restrict__ img_out) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < 12 * 8 * 64 * 8) {
        const int j = id & 7, lane = (id >> 3) & 63, f = id >> 9, cb = f >> 3, sft = f & 7, r = lane & 31, h = lane >> 5;
        img_qkv[id] = f2bf(wqkv[(size_t)(32 * cb + r) * RN_D + 16 * sft + 8 * h + j]);
    }
    if (id < 4 * 8 * 64 * 8) {
        const int j = id & 7, lane = (id >> 3) & 63, f = id >> 9, cb = f >> 3, sft = f & 7, r = lane & 31, h = lane >> 5;
        const int d = j < 4 ? 4 * h + j : 8 + 4 * h + (j - 4);
        img_out[id] = f2bf(wout[(size_t)(32 * cb + r) * RN_D + 16 * sft + d]);
    }
}
void launch_build_attn_images(const float* wqkv, const float* wout, bf16_t* img_qkv, bf16_t* img_out, hipStream_t s) {
    hipLaunchKernelGGL(k_build_attn_images, dim3(12 * 8 * 64 * 8 / 256), dim3(256), 0, s, wqkv, wout, img_qkv, img_out);
}
#define AL_NW 16
__global__ void __launch_bounds__(AL_NW * 64) k_attn_layer_rna(PackInfo pk, float* __restrict__ x, const bf16_t* __restrict__ wqkv,
        const float* __restrict__ bqkv, const bf16_t* __restrict__ wout, const float* __restrict__ bout,
        const float* __restrict__ gscale, const float* __restrict__ gshift, int t_tot) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* Xs = reinterpret_cast<u32x4*>(smem);                                 // [rb][s][lane]; later Os [qb][head][lane]
    u32x4* Qs = reinterpret_cast<u32x4*>(smem + 40960);                         // [head][row < AL_NR][h]
    u32x4* Ks = Qs + 8 * AL_NR * 2;
    float* Ys = reinterpret_cast<float*>(smem + 40960);                         // [row][AL_YLD]   (after phase 2)
    bf16_t* Vs = reinterpret_cast<bf16_t*>(smem + 40960 + 76032);               // [channel][AL_KS keys]
    float* lb = reinterpret_cast<float*>(smem + 40960 + 76032 + 40960);         // [384 qkv bias][128 out bias]
    float* red = reinterpret_cast<float*>(smem);                                // [8][128] x 2   (phase 4; the O image is dead)
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const int b = blockIdx.x;
    const int n = pk.len[b];
    if (n <= 0 || n > AL_NR) return;                                            // (host launches this kernel only when T <= AL_NR)
    const int base = pk.cu[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int nrb = (n + 31) >> 5;
    float* xb = x + (size_t)base * RN_D;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    // ---- phase 0
    if (tid < 384) lb[tid] = bqkv[tid];
    else if (tid < 512) lb[tid] = bout[tid - 384];
    for (int t = wave; t < nrb * 8; t += AL_NW) {
        const int rb = t >> 3, s = t & 7, row = 32 * rb + r;
        u32x4 v = zero4;
        if (row < n) {
            const float* src = xb + (size_t)row * RN_D + 16 * s + 8 * h;
            const f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + 4);
            v = u32x4{pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(c[0], c[1]), pack2(c[2], c[3])};
        }
        Xs[t * 64 + lane] = v;
    }
    // ---- phase 1: tiles t = cb * nrb + rb (cb 0..11), the weight fragments of a wave's NEXT tile requested before this one's MFMAs
    auto load_wqkv = [&](u32x4 (&af)[8], int t) {
        const int cb = t / nrb;
#pragma unroll
        for (int s = 0; s < 8; ++s) af[s] = reinterpret_cast<const u32x4*>(wqkv)[(cb * 8 + s) * 64 + lane];
    };
    u32x4 afa[8], afb[8];
    const int nt1 = nrb * 12;
    if (wave < nt1) load_wqkv(afa, wave);
    __syncthreads();
    auto qkv_tile = [&](const u32x4 (&af)[8], int t) {
        const int cb = t / nrb, rb = t - cb * nrb;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = mfma32(af[s], Xs[(rb * 8 + s) * 64 + lane], acc);
        const int row = 32 * rb + r;
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = acc[i] + lb[32 * cb + (i & 3) + 4 * h + 8 * (i >> 2)];
        if (cb < 8) {                      // Q (scaled by 1/sqrt(16)) or K: two heads per block, ready fragments
            const float sc = cb < 4 ? 0.25f * 1.44269504088896f : 1.0f;      // (scores in log2 units: the softmax below uses exp2)
            u32x4* dst = cb < 4 ? Qs : Ks;
            const int hd0 = 2 * (cb & 3);
            if (row < AL_NR) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    dst[((hd0 + g) * AL_NR + row) * 2 + h] = u32x4{pack2(sc * v[8 * g], sc * v[8 * g + 1]), pack2(sc * v[8 * g + 2], sc * v[8 * g + 3]),
                                                                    pack2(sc * v[8 * g + 4], sc * v[8 * g + 5]), pack2(sc * v[8 * g + 6], sc * v[8 * g + 7])};
            }
        } else {                           // V^T [channel][key]
#pragma unroll
            for (int i = 0; i < 16; ++i) Vs[(32 * (cb - 8) + (i & 3) + 4 * h + 8 * (i >> 2)) * AL_KS + row] = f2bf(v[i]);
        }
    };
    for (int t = wave; t < nt1; t += 2 * AL_NW) {
        if (t + AL_NW < nt1) load_wqkv(afb, t + AL_NW);
        qkv_tile(afa, t);
        if (t + AL_NW < nt1) {
            if (t + 2 * AL_NW < nt1) load_wqkv(afa, t + 2 * AL_NW);
            qkv_tile(afb, t + AL_NW);
        }
    }
    auto load_wout = [&](u32x4 (&af)[8], int t) {       // (image built in the d order of the O image: {4h..4h+3, 8+4h..8+4h+3} of head s)
        const int cb = t / nrb;
#pragma unroll
        for (int s = 0; s < 8; ++s) af[s] = reinterpret_cast<const u32x4*>(wout)[(cb * 8 + s) * 64 + lane];
    };
    const int nt3 = nrb * 4;
    __syncthreads();
    // ---- phase 2: tasks (head, query block)
    for (int task = wave; task < 8 * nrb; task += AL_NW) {
        const int hd = task & 7, qb = task >> 3;
        const u32x4 qf = Qs[(hd * AL_NR + min(32 * qb + r, AL_NR - 1)) * 2 + h];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float m_run = -3.0e38f, l_run = 0.f;
        for (int kb = 0; kb < nrb; ++kb) {
            f32x16 sc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[i] = 0.f;
            sc = mfma32(Ks[(hd * AL_NR + min(32 * kb + r, AL_NR - 1)) * 2 + h], qf, sc);      // S^T[key][query], log2 units
            if (32 * kb + 32 > n) {            // only the last key block can hold keys beyond the RNA (wave-uniform branch)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                    sc[i] = key < n ? sc[i] : -3.0e38f;
                }
            }
            float mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(sc[8], sc[9]), fmaxf(sc[10], sc[11])), fmaxf(fmaxf(sc[12], sc[13]), fmaxf(sc[14], sc[15]))));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sc[i] = __builtin_amdgcn_exp2f(sc[i] - m_new); ps += sc[i]; }
            l_run = l_run * corr + ps;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] *= corr;
#pragma unroll
            for (int sblk = 0; sblk < 2; ++sblk) {
                const u32x4 pf = {pack2(sc[8 * sblk], sc[8 * sblk + 1]), pack2(sc[8 * sblk + 2], sc[8 * sblk + 3]),
                                  pack2(sc[8 * sblk + 4], sc[8 * sblk + 5]), pack2(sc[8 * sblk + 6], sc[8 * sblk + 7])};
                u32x4 vf = zero4;
                if (r < 16) {
                    const bf16_t* vp = Vs + (16 * hd + r) * AL_KS + 32 * kb + 16 * sblk + 4 * h;
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(vp), hi = *reinterpret_cast<const u32x2*>(vp + 8);
                    vf = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
                acc = mfma32(vf, pf, acc);                                   // O^T[d][query]
            }
        }
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.0f / l_tot;
        // d = (i & 3) + 4h + 8 (i >> 2), i < 8: this lane's k-slice of the out-projection's k-step `hd`.  (The x image in region A is dead:
        // every wave passed the barrier behind phase 1.)
        reinterpret_cast<u32x4*>(smem)[(qb * 8 + hd) * 64 + lane] =
            u32x4{pack2(acc[0] * inv, acc[1] * inv), pack2(acc[2] * inv, acc[3] * inv), pack2(acc[4] * inv, acc[5] * inv), pack2(acc[6] * inv, acc[7] * inv)};
    }
    if (wave < nt3) load_wout(afa, wave);           // (in flight across the barrier)
    __syncthreads();
    // ---- phase 3: y = Wout . O^T + b + x, tiles t = cb * nrb + rb (cb 0..3)
    auto out_tile = [&](const u32x4 (&af)[8], int t) {
        const int cb = t / nrb, rb = t - cb * nrb;
        const int row = 32 * rb + r;
        const bool ok = row < n;
        f32x4 xr[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) xr[v] = *reinterpret_cast<const f32x4*>(xb + (size_t)(ok ? row : 0) * RN_D + 32 * cb + 4 * h + 8 * v);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = mfma32(af[s], reinterpret_cast<const u32x4*>(smem)[(rb * 8 + s) * 64 + lane], acc);
        if (ok) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float* bo = lb + 384 + 32 * cb + 4 * h + 8 * v;
                *reinterpret_cast<f32x4*>(Ys + row * AL_YLD + 32 * cb + 4 * h + 8 * v) =
                    f32x4{acc[4 * v] + bo[0] + xr[v][0], acc[4 * v + 1] + bo[1] + xr[v][1], acc[4 * v + 2] + bo[2] + xr[v][2], acc[4 * v + 3] + bo[3] + xr[v][3]};
            }
        }
    };
    for (int t = wave; t < nt3; t += 2 * AL_NW) {
        if (t + AL_NW < nt3) load_wout(afb, t + AL_NW);
        out_tile(afa, t);
        if (t + AL_NW < nt3) out_tile(afb, t + AL_NW);
    }
    __syncthreads();
    // ---- phase 4: GraphNormalization over the RNA (functional.py:33-46), thread = (channel, row group of 8)
    {
        const int c = tid & 127, g = tid >> 7;
        constexpr int NRW = AL_NR / 8;                          // this thread's rows g, g + 8, ... stay in registers between the passes
        float yv[NRW];
        float s1 = 0.f;
#pragma unroll
        for (int q = 0; q < NRW; ++q) { const int row = g + 8 * q; yv[q] = row < n ? Ys[row * AL_YLD + c] : 0.f; s1 += yv[q]; }
        red[g * 128 + c] = s1;
        const float gsc = gscale[c], gsh = gshift[c];
        __syncthreads();
        const float fn = (float)n;
        float tot = red[c];
#pragma unroll
        for (int q = 1; q < 8; ++q) tot += red[q * 128 + c];
        const float mu = tot / fn;
        float s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NRW; ++q) { const float d = yv[q] - mu; s2 = (g + 8 * q < n) ? fmaf(d, d, s2) : s2; }
        red[1024 + g * 128 + c] = s2;
        __syncthreads();
        float sq = red[1024 + c];
#pragma unroll
        for (int q = 1; q < 8; ++q) sq += red[1024 + q * 128 + c];
        const float var = (sq + (float)(t_tot - n) * mu * mu) / fn;
        const float a = gsc / sqrtf(var + kSEPS), bb = gsh - mu * a;
#pragma unroll
        for (int q = 0; q < NRW; ++q) { const int row = g + 8 * q; if (row < n) xb[(size_t)row * RN_D + c] = fmaf(yv[q], a, bb); }
    }
    __syncthreads();
}

// returns 0 when the fused per-RNA layer ran (head dim 16, 8 heads, every RNA of the batch <= AL_NR residues), 1 otherwise.
// wqkv / wout: the fragment images of launch_build_attn_images.
int launch_attn_layer_rna(const PackInfo& pk, float* x, const bf16_t* wqkv, const float* bqkv, const bf16_t* wout, const float* bout,
                          int heads, const float* gscale, const float* gshift, int t_tot, hipStream_t s) {
    static const bool off = [] { const char* e = getenv("RNAMPNN_NO_ATTN_FUSE"); return e && e[0] == '1'; }();
    if (off || heads != 8 || pk.T > AL_NR) return 1;
    static DevAttr attr;
    ensure_dyn_lds((const void*)k_attn_layer_rna, AL_LDS, attr);
    hipLaunchKernelGGL(k_attn_layer_rna, dim3(pk.B), dim3(AL_NW * 64), AL_LDS, s, pk, x, wqkv, bqkv, wout, bout, gscale, gshift, t_tot);
    return 0;
}

// returns 0 when handled (head dim 16), 1 otherwise (the caller uses the f32 kernel)
int launch_attention_bf16(const PackInfo& pk, const float* qkv, int heads, float* out, hipStream_t s) {
    if (RN_D / heads != 16) return 1;
    const int nkb = (pk.T + 31) / 32;
    const int chunk = nkb < 64 ? nkb : 64;                    // <= 128 KiB of K / V^T fragments per chunk
    const size_t lds = (size_t)chunk * 64 * 16 * 2;
    static DevAttr attr;
    ensure_dyn_lds((const void*)k_attention_bf16_hd16, lds, attr);
    hipLaunchKernelGGL(k_attention_bf16_hd16, dim3(pk.B, heads, (pk.T + 255) / 256), dim3(512), lds, s, pk, qkv, out, chunk);
    return 0;
}
