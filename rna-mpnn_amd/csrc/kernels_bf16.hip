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

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

static constexpr float kSEPS = 1.0e-6f;

__device__ __forceinline__ bf16_t f2bf(float x) { return __builtin_bit_cast(bf16_t, (__bf16)x); }   // RNE, NaN kept
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ unsigned pack2(float a, float b) {        // one v_cvt_pk_bf16_f32
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_bf(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// GELU for the bf16 path: x * sigmoid(x * (c0 + c1 x^2)), coefficients minimax-fitted to the exact
// erf form (max |err| 2.7e-4 at |x| ~ 2-3, 15-30x below the bf16 rounding of the result there);
// monotone argument, so no clamp: 7 VALU instructions, 2 of them transcendental.
// The f32 path and the node-level GEMM epilogues keep erff().
__device__ __forceinline__ float gelu_fast(float x) {
    float t = x * x;
    float p = fmaf(t, -0.10012571f, -2.3087657f);          // -log2(e) * (c0 + c1 t), c0 = 1.60031416, c1 = 0.06940179
    float ex = __builtin_amdgcn_exp2f(x * p);              // exp(-x (c0 + c1 t))
    return x * __builtin_amdgcn_rcpf(1.0f + ex);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// channel held by accumulator row m of a 32-row block when the output order is "natural per lane":
// lane half h = (m>>2)&1, register i = (m&3) + 4*(m>>3)  ->  channel 32*blk + 16*h + i
__host__ __device__ __forceinline__ int ch_nat(int blk, int m) { return 32 * blk + 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3); }
// ... when the output order must equal the e B-fragment layout (lane half h holds channels
// 32*blk + 8h + {0..7} and 32*blk + 16 + 8h + {0..7})
__host__ __device__ __forceinline__ int ch_efrag(int blk, int m) {
    int h = (m >> 2) & 1, i = (m & 3) + 4 * (m >> 3);
    return i < 8 ? 32 * blk + 8 * h + i : 32 * blk + 16 + 8 * h + (i - 8);
}

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
__global__ void k_build_mlp_image(const float* __restrict__ wc, int ld_wc, const float* __restrict__ w2, int ld_w2,
                                  const float* __restrict__ b2, int out_perm, bf16_t* __restrict__ img, float* __restrict__ b2p) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;      // one element of one fragment
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
        img[id] = f2bf(v);
    }
    if (id < 128) {
        if (out_perm) {
            int ob = id >> 5, m = id & 31;                // position 32*ob + 16*h + i  <-> row m
            int h = (m >> 2) & 1, i = (m & 3) + 4 * (m >> 3);
            b2p[32 * ob + 16 * h + i] = b2[ch_efrag(ob, m)];
        } else {
            b2p[id] = b2[id];
        }
    }
}
void launch_build_mlp_image(const float* wc, int ld_wc, const float* w2, int ld_w2, const float* b2, int out_perm,
                            bf16_t* img, float* b2p, hipStream_t s) {
    int total = 2 * 32 * 64 * 8;
    hipLaunchKernelGGL(k_build_mlp_image, dim3((total + 255) / 256), dim3(256), 0, s, wc, ld_wc, w2, ld_w2, b2, out_perm, img, b2p);
}

// Edge-embedding MLP image.  The 90 raw features are computed per lane half in a kernel-private
// order (EMB_SLOTS slots per half, 7 k-steps of 16): slot p of half h sits at k = 16*(p>>3) + 8h + (p&7).
#define EMB_KS 7
#define EMB_SLOTS 56
__host__ __device__ __forceinline__ int emb_feature_of_slot(int h, int p) {      // -> original feature id or -1
    if (p < 28) { int a = 4 * h + p / 7, b = p % 7; return a <= 6 ? a * 7 + b : -1; }
    if (p < 43) { int q = p - 28, ai = q / 5, b = q % 5, a = 2 * h + ai; return (h == 0 ? ai < 2 : a <= 4) ? 49 + a * 5 + b : -1; }
    if (p < 51) { int q = p - 43, a = 2 * h + q / 4, b = q % 4; return 74 + a * 4 + b; }
    return -1;
}
__global__ void k_build_embed_image(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ b1,
                                    bf16_t* __restrict__ img, float* __restrict__ b1p) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = 4 * EMB_KS * 64 * 8, n1 = 32 * 64 * 8;
    if (id < n0) {
        int j = id & 7, lane = (id >> 3) & 63, f = id >> 9;
        int mb = f / EMB_KS, s = f % EMB_KS, r = lane & 31, h = lane >> 5;
        int feat = emb_feature_of_slot(h, 8 * s + j);
        img[id] = feat >= 0 ? f2bf(w0[(size_t)ch_nat(mb, r) * RN_ERAW + feat]) : (bf16_t)0;
    } else if (id < n0 + n1) {
        int e = id - n0;
        int j = e & 7, lane = (e >> 3) & 63, f = e >> 9;
        int ob = f >> 3, ks = f & 7, mb = ks >> 1, sp = ks & 1, r = lane & 31, h = lane >> 5;
        img[id] = f2bf(w1[(size_t)ch_efrag(ob, r) * RN_D + 32 * mb + 16 * h + 8 * sp + j]);
    }
    if (id < 128) {
        int ob = id >> 5, m = id & 31, h = (m >> 2) & 1, i = (m & 3) + 4 * (m >> 3);
        b1p[32 * ob + 16 * h + i] = b1[ch_efrag(ob, m)];
    }
}
void launch_build_embed_image(const float* w0, const float* w1, const float* b1, bf16_t* img, float* b1p, hipStream_t s) {
    int total = (4 * EMB_KS + 32) * 64 * 8;
    hipLaunchKernelGGL(k_build_embed_image, dim3((total + 255) / 256), dim3(256), 0, s, w0, w1, b1, img, b1p);
}

__global__ void k_bf16_to_f32(const bf16_t* __restrict__ src, float* __restrict__ dst, const int* __restrict__ ntot, int per_row) {
    size_t n = (size_t)(*ntot) * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = bf2f(src[i]);
}
__global__ void k_f32_to_bf16(const float* __restrict__ src, bf16_t* __restrict__ dst, const int* __restrict__ ntot, int per_row) {
    size_t n = (size_t)(*ntot) * per_row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = f2bf(src[i]);
}
static unsigned conv_grid(size_t max_elems) { size_t g = (max_elems + 255) / 256; return (unsigned)(g < 8192 ? (g ? g : 1) : 8192); }
void launch_bf16_to_f32(const bf16_t* src, float* dst, size_t max_elems, const int* ntot, int per_row, hipStream_t s) {
    hipLaunchKernelGGL(k_bf16_to_f32, dim3(conv_grid(max_elems)), dim3(256), 0, s, src, dst, ntot, per_row);
}
void launch_f32_to_bf16(const float* src, bf16_t* dst, size_t max_elems, const int* ntot, int per_row, hipStream_t s) {
    hipLaunchKernelGGL(k_f32_to_bf16, dim3(conv_grid(max_elems)), dim3(256), 0, s, src, dst, ntot, per_row);
}

// ------------------------------------------------------------------------------------------
// shared pieces of the two edge kernels
struct BlockLane {            // what one lane knows about its edge in the current 32-edge block
    int node;                 // packed residue of this lane's edge (valid when ok)
    int j;                    // packed neighbour row (or -1)
    bool ok;                  // lane maps to an existing (residue, slot)
    bool valid;               // ... and the slot holds an edge
    size_t erow;              // row of e / nbr
};

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

// acc = P[node][32mb+16h..+16] + Q[j][..]  (first Linear's node parts; rows are [P | Q], 256 wide)
__device__ __forceinline__ f32x16 init_pq(const float* __restrict__ pp, const float* __restrict__ qp, int mb) {
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        f32x4 a = *reinterpret_cast<const f32x4*>(pp + 32 * mb + 4 * v);
        f32x4 b = *reinterpret_cast<const f32x4*>(qp + 32 * mb + 4 * v);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[4 * v + t] = a[t] + b[t];
    }
    return acc;
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

// First Linear (+GELU) of an e-MLP: hb[2mb + s'] = bf16(GELU(P + Q + Wc . e))   [channel blocks mb]
__device__ __forceinline__ void mlp_first(const u32x4* __restrict__ img, int lane, const u32x4 (&ef)[8],
                                          const float* __restrict__ pp, const float* __restrict__ qp, u32x4 (&hb)[8]) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        f32x16 acc = init_pq(pp, qp, mb);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = mfma32(img[(mb * 8 + s) * 64 + lane], ef[s], acc);
        gelu_pack(acc, hb[2 * mb], hb[2 * mb + 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------
// Fused ResMPNN step on 32-edge blocks (mpnn.py:154-265), bf16 MFMA:
//   DO_EDGE: e <- e + MLP_e(P_e[i] + Q_e[j] + e Wc_e)   (edge update of the previous layer)
//   DO_MSG : h_pre = h + mean_valid MLP_m(P_m[i] + Q_m[j] + e Wc_m)   (message + aggregation)
// One wave owns one block: lanes (r, h) = (edge r of the block, k-half h).  k <= 16 packs
// npb = 32/k residues into a block (edges of consecutive residues are contiguous in e).
template <bool DO_EDGE, bool DO_MSG>
__global__ void __launch_bounds__(512, 2) k_mpnn_bf16(PackInfo pk, int k, const int* __restrict__ nbr, bf16_t* __restrict__ e,
        const float* __restrict__ pq_e, const float* __restrict__ pq_m, MpnnWB we, MpnnWB wm,
        const float* __restrict__ h_in, float* __restrict__ h_pre, float* __restrict__ msg_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* img_e = reinterpret_cast<u32x4*>(smem);
    u32x4* img_m = img_e + (DO_EDGE ? 4096 : 0);
    const int tid = threadIdx.x;
    if (DO_EDGE) for (int i = tid; i < 4096; i += 512) img_e[i] = reinterpret_cast<const u32x4*>(we.img)[i];
    if (DO_MSG) for (int i = tid; i < 4096; i += 512) img_m[i] = reinterpret_cast<const u32x4*>(wm.img)[i];
    __syncthreads();

    const int ntot = pk.cu[pk.B];
    const int npb = k > 16 ? 1 : 32 / k;
    const int nblocks = (ntot + npb - 1) / npb;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int zero_row = pk.Nmax;

    for (int blk = blockIdx.x * 8 + wave; blk < nblocks; blk += gridDim.x * 8) {
        BlockLane bl = block_lane(blk, npb, k, ntot, r, nbr);
        const int prow = bl.ok ? bl.node : zero_row;
        const int qrow = bl.valid ? (bl.j > zero_row ? zero_row : bl.j) : zero_row;    // phantom -> zero row
        u32x4 ef[8];
        const u32x4* erp = reinterpret_cast<const u32x4*>(e + bl.erow * RN_D + 8 * h);
#pragma unroll
        for (int s = 0; s < 8; ++s) ef[s] = bl.ok ? erp[2 * s] : u32x4{0u, 0u, 0u, 0u};

        u32x4 hb[8];
        if (DO_EDGE) {
            mlp_first(img_e, lane, ef, pq_e + (size_t)prow * 256 + 16 * h, pq_e + (size_t)qrow * 256 + 128 + 16 * h, hb);
            // second Linear, rows in the e fragment layout: registers 0..7 of block ob <-> ef[2ob], 8..15 <-> ef[2ob+1]
            const bool wr = bl.ok && bl.valid;
            u32x4* ewp = reinterpret_cast<u32x4*>(e + bl.erow * RN_D + 8 * h);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
                f32x16 acc = init_vec16(we.b2p + 32 * ob + 16 * h);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = mfma32(img_e[2048 + (ob * 8 + ks) * 64 + lane], hb[ks], acc);
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    u32x4 old = ef[2 * ob + sp], nw;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        nw[t] = pack2(lo_bf(old[t]) + gelu_fast(acc[8 * sp + 2 * t]),
                                      hi_bf(old[t]) + gelu_fast(acc[8 * sp + 2 * t + 1]));
                    if (wr) { ef[2 * ob + sp] = nw; ewp[2 * (2 * ob + sp)] = nw; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (DO_MSG) {
            mlp_first(img_m, lane, ef, pq_m + (size_t)prow * 256 + 16 * h, pq_m + (size_t)qrow * 256 + 128 + 16 * h, hb);
            const unsigned vmask = (unsigned)(__ballot(bl.ok && bl.valid) & 0xffffffffull);   // bit r = edge r is real
            // last Linear un-transposed: rows = edges of the block (registers), columns = channels 32nb + r
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const float b = wm.b2p[32 * nb + r];
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = b;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) acc = mfma32(hb[ks], img_m[2048 + (nb * 8 + ks) * 64 + lane], acc);
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = gelu_fast(acc[i]);
                if (msg_out) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int er = (i & 3) + 8 * (i >> 2) + 4 * h;
                        const int node = blk * npb + er / k;
                        if (er < npb * k && node < ntot)
                            msg_out[((size_t)blk * npb * k + er) * RN_D + 32 * nb + r] = acc[i] * (float)((vmask >> er) & 1u);
                    }
                }
                for (int q = 0; q < npb; ++q) {
                    const int node = blk * npb + q;
                    if (node >= ntot) break;
                    const unsigned seg = npb == 1 ? vmask : (vmask & (((1u << k) - 1u) << (q * k)));
                    const unsigned segh = seg >> (4 * h);
                    float sum = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        sum = fmaf(acc[i], (float)((segh >> ((i & 3) + 8 * (i >> 2))) & 1u), sum);
                    sum += __shfl_xor(sum, 32, 64);
                    const int cnt = __popc(seg);
                    if (h == 0) {
                        const size_t o = (size_t)node * RN_D + 32 * nb + r;
                        h_pre[o] = h_in[o] + sum / (float)(cnt > 0 ? cnt : 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

static int num_cus() {
    static int n = 0;
    if (!n) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

void launch_mpnn_bf16(const PackInfo& pk, int k, bool do_edge, bool do_msg, const int* nbr, bf16_t* e,
                      const float* pq_e, const float* pq_m, MpnnWB we, MpnnWB wm, const float* h_in,
                      float* h_pre, float* msg_out, hipStream_t s) {
    const int npb = k > 16 ? 1 : 32 / k;
    const int max_blocks = (pk.Nmax + npb - 1) / npb;
    int grid = (max_blocks + 7) / 8;
    if (grid > num_cus()) grid = num_cus();
    if (grid < 1) grid = 1;
    size_t lds = (size_t)((do_edge ? 1 : 0) + (do_msg ? 1 : 0)) * 65536;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_mpnn_bf16<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void*)k_mpnn_bf16<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute((const void*)k_mpnn_bf16<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_done = true;
    }
    if (do_edge && do_msg)
        hipLaunchKernelGGL((k_mpnn_bf16<true, true>), dim3(grid), dim3(512), lds, s, pk, k, nbr, e, pq_e, pq_m, we, wm, h_in, h_pre, msg_out);
    else if (do_edge)
        hipLaunchKernelGGL((k_mpnn_bf16<true, false>), dim3(grid), dim3(512), lds, s, pk, k, nbr, e, pq_e, pq_m, we, wm, h_in, h_pre, msg_out);
    else
        hipLaunchKernelGGL((k_mpnn_bf16<false, true>), dim3(grid), dim3(512), lds, s, pk, k, nbr, e, pq_e, pq_m, we, wm, h_in, h_pre, msg_out);
}

// ------------------------------------------------------------------------------------------
// Edge featurisation + embedding MLP on MFMA (feature.py:386-571): each lane computes the
// EMB_SLOTS raw features of its (edge, half) from the two geometry records straight into the
// B fragments of Linear(90,128) (the 90-wide tensor exists only in registers), then
// GELU -> Linear(128,128) -> GELU -> e0 in bf16 (16-byte stores); invalid slots store 0.
__global__ void __launch_bounds__(512, 2) k_edge_embed_bf16(PackInfo pk, int k, const float* __restrict__ geom,
        const int* __restrict__ nbr, const bf16_t* __restrict__ img_g, const float* __restrict__ b0,
        const float* __restrict__ b1p, bf16_t* __restrict__ e) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4* img = reinterpret_cast<u32x4*>(smem);
    const int NFRAG = 4 * EMB_KS + 32;
    const int tid = threadIdx.x;
    for (int i = tid; i < NFRAG * 64; i += 512) img[i] = reinterpret_cast<const u32x4*>(img_g)[i];
    __syncthreads();
    const int ntot = pk.cu[pk.B];
    const int npb = k > 16 ? 1 : 32 / k;
    const int nblocks = (ntot + npb - 1) / npb;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;

    for (int blk = blockIdx.x * 8 + wave; blk < nblocks; blk += gridDim.x * 8) {
        BlockLane bl = block_lane(blk, npb, k, ntot, r, nbr);
        const int inode = bl.ok ? bl.node : 0;
        const float* gi = geom + (size_t)inode * RN_GEOM;
        const float* gj = geom + (size_t)(bl.valid ? bl.j : inode) * RN_GEOM;
        float cj[RN_GEOM];
#pragma unroll
        for (int v = 0; v < RN_GEOM / 4; ++v) {
            f32x4 t = *reinterpret_cast<const f32x4*>(gj + 4 * v);
            cj[4 * v] = t[0]; cj[4 * v + 1] = t[1]; cj[4 * v + 2] = t[2]; cj[4 * v + 3] = t[3];
        }
        float ft[EMB_SLOTS];
#pragma unroll
        for (int ai = 0; ai < 4; ++ai) {                       // distances: central atoms 4h .. 4h+3
            int a = 4 * h + ai; a = a > 6 ? 6 : a;
            float ax = gi[a * 3], ay = gi[a * 3 + 1], az = gi[a * 3 + 2];
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                float dx = ax - cj[b * 3], dy = ay - cj[b * 3 + 1], dz = az - cj[b * 3 + 2];
                ft[ai * 7 + b] = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz + kSEPS);
            }
        }
#pragma unroll
        for (int ai = 0; ai < 3; ++ai) {                       // bond cosines: central bonds 2h .. 2h+2
            int a = 2 * h + ai; a = a > 4 ? 4 : a;
            float ax = gi[21 + a * 3], ay = gi[22 + a * 3], az = gi[23 + a * 3];
#pragma unroll
            for (int b = 0; b < 5; ++b) ft[28 + ai * 5 + b] = ax * cj[21 + b * 3] + ay * cj[22 + b * 3] + az * cj[23 + b * 3];
        }
#pragma unroll
        for (int ai = 0; ai < 2; ++ai) {                       // normal cosines: central normals 2h, 2h+1
            int a = 2 * h + ai;
            float ax = gi[36 + a * 3], ay = gi[37 + a * 3], az = gi[38 + a * 3];
#pragma unroll
            for (int b = 0; b < 4; ++b) ft[43 + ai * 4 + b] = ax * cj[36 + b * 3] + ay * cj[37 + b * 3] + az * cj[38 + b * 3];
        }
#pragma unroll
        for (int p = 51; p < EMB_SLOTS; ++p) ft[p] = 0.f;
        u32x4 xf[EMB_KS];
#pragma unroll
        for (int s = 0; s < EMB_KS; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) xf[s][t] = pack2(ft[8 * s + 2 * t], ft[8 * s + 2 * t + 1]);
        __builtin_amdgcn_sched_barrier(0);

        u32x4 hb[8];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            f32x16 acc = init_vec16(b0 + 32 * mb + 16 * h);
#pragma unroll
            for (int s = 0; s < EMB_KS; ++s) acc = mfma32(img[(mb * EMB_KS + s) * 64 + lane], xf[s], acc);
            gelu_pack(acc, hb[2 * mb], hb[2 * mb + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        u32x4* ewp = reinterpret_cast<u32x4*>(e + bl.erow * RN_D + 8 * h);
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            f32x16 acc = init_vec16(b1p + 32 * ob + 16 * h);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) acc = mfma32(img[(4 * EMB_KS + ob * 8 + ks) * 64 + lane], hb[ks], acc);
            if (bl.ok) {
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    u32x4 nw;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        nw[t] = bl.valid ? pack2(gelu_fast(acc[8 * sp + 2 * t]), gelu_fast(acc[8 * sp + 2 * t + 1])) : 0u;
                    ewp[2 * (2 * ob + sp)] = nw;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

void launch_edge_embed_bf16(const PackInfo& pk, int k, const float* geom, const int* nbr, const bf16_t* img,
                            const float* b0, const float* b1p, bf16_t* e, hipStream_t s) {
    const int npb = k > 16 ? 1 : 32 / k;
    const int max_blocks = (pk.Nmax + npb - 1) / npb;
    int grid = (max_blocks + 7) / 8;
    if (grid > 2 * num_cus()) grid = 2 * num_cus();
    if (grid < 1) grid = 1;
    size_t lds = (size_t)(4 * EMB_KS + 32) * 1024;
    hipLaunchKernelGGL(k_edge_embed_bf16, dim3(grid), dim3(512), lds, s, pk, k, geom, nbr, img, b0, b1p, e);
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
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + 64 * wc + 32 * b + r;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + 64 * wr + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < ntot) {
                    float v = acc[a][b][i] + bv;
                    if (act == 1) v = gelu_erf(v);
                    if (res) v += res[(size_t)row * ldres + col];
                    Y[(size_t)row * ldy + col] = v;
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
