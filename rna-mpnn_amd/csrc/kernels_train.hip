// f32 kernels of the TRAINING path (loss + gradients of every parameter), gfx950.
// Correctness-first building blocks on packed rows: the training forward materialises the per-edge
// pre-activations ("tape") that the fused inference kernels keep in registers, and the backward pass
// walks the tape with generic row kernels (GEMM, transposed-reduce GEMM, element-wise GELU',
// segment mean / scatter, GraphNorm and attention backward).  Reference: rnampnn.py:151-154,187-207
// (loss = cross_entropy(softmax(logits)[valid], label), mean over valid nucleotides) and the forward
// lines cited in kernels_f32.hip.  Dropout: counter-hash masks (TDrop, kernels_train.h); every cross-workgroup sum is an ordered two-stage reduction (no float atomics).
#include "kernels_train.h"
#include <cstdio>
#include <vector>
#include <cstdlib>

static constexpr float kSEPS = 1.0e-6f;
#define TLD 132

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float x) {     // d/dx [x Phi(x)] = Phi(x) + x phi(x)
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ int nrows(const TRows& r) { return *r.ntot * r.mul; }
// dropout multiplier of one element: 0 or 1/(1-p) (kernels_train.h: TDrop; restated by the oracle's dropout_multiplier).
__device__ __forceinline__ unsigned drop_key(const TDrop& d, unsigned site) {         // wave-uniform part of the hash input
    const unsigned long long sd = d.seed_dev ? *d.seed_dev : d.seed;
    return site * 0x85EBCA6Bu + (unsigned)sd + (unsigned)(sd >> 32) * 0x27D4EB2Fu;
}
__device__ __forceinline__ unsigned drop_hash(unsigned x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float drop_mul(const TDrop& d, unsigned site, unsigned long long idx) {
    if (d.thresh == 0u) return 1.f;
    const unsigned long long P = idx >> 1;
    const unsigned x = drop_hash((unsigned)P + (unsigned)(P >> 32) * 0xC2B2AE35u + drop_key(d, site));
    return ((idx & 1ull) ? x >> 16 : x & 0xffffu) >= d.thresh ? d.scale : 0.f;
}
// both elements of pair P (element indices 2P, 2P + 1) when P is known to fit 32 bits (every [rows][D] tensor of the trainer: the
// entry points bound rows * D / 2 < 2^32); key = drop_key(d, site)
__device__ __forceinline__ void drop_pair(const TDrop& d, unsigned key, unsigned P, float& m0, float& m1) {
#ifdef TE_EXP_NOHASH      // timing experiment only (wrong masks): what the hash costs
    m0 = m1 = __uint_as_float((P + key) & 0x3f800000u); return;
#endif
    if (d.thresh == 0u) { m0 = 1.f; m1 = 1.f; return; }
    const unsigned x = drop_hash(P + key);
    m0 = (x & 0xffffu) >= d.thresh ? d.scale : 0.f;
    m1 = (x >> 16) >= d.thresh ? d.scale : 0.f;
}
// the 8 multipliers of elements 8 * P8 .. 8 * P8 + 7  (P8 = element index / 8)
__device__ __forceinline__ void drop8(const TDrop& d, unsigned key, unsigned P8, float (&m)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) drop_pair(d, key, 4u * P8 + q, m[2 * q], m[2 * q + 1]);
}
// GELU and its derivative for the fused prologues / epilogues of the bf16-mixed GEMMs: Phi(x) ~ sigmoid(x (c0 + c1 x^2)),
// coefficients minimax-fitted to the erf form (max |x Phi - gelu| 2.7e-4, below the bf16 rounding of the operands these
// values are converted to); derivative = Phi + x phi.  The f32 kernels (parity grade) keep erff.
__device__ __forceinline__ float phi_fast(float x) {
#ifdef TE_EXP_NOACT       // timing experiment only (wrong values): what the transcendental GELU costs
    return fmaf(x, 0.25f, 0.5f);
#endif
    const float p = fmaf(x * x, -0.10012571f, -2.3087657f);           // -log2(e) (c0 + c1 x^2)
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
}
__device__ __forceinline__ float gelu_fast(float x) { return x * phi_fast(x); }
__device__ __forceinline__ float gelu_d_fast(float x) {
#ifdef TE_EXP_NOACT
    return fmaf(x, 0.5f, 0.5f);
#endif
    return fmaf(x * 0.3989422804f, __builtin_amdgcn_exp2f(x * x * -0.72134752f), phi_fast(x));
}

// ------------------------------------------------------------------------------------------
// Y[p][0:N] = (beta ? Y : 0) + X[p][0:K] . Wt[0:K][0:N] + bias      (32 x 128 tile, K % 4 == 0)
__global__ void __launch_bounds__(256) k_tgemm(TRows rows, const float* __restrict__ X, int ldx, int K,
        const float* __restrict__ Wt, int ldw, const float* __restrict__ bias, int N, float* __restrict__ Y, int ldy, int beta) {
    __shared__ __attribute__((aligned(16))) float Xs[32 * TLD];
    const int R = nrows(rows);
    const int row0 = blockIdx.x * 32;
    if (row0 >= R) return;
    const int tid = threadIdx.x, c = tid & 127, g = tid >> 7;
    const int cc = blockIdx.y * 128 + c;
    const int cw = cc < N ? cc : N - 1;
    float acc[16];
    const float bv = bias ? bias[cw] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bv;
    for (int k0 = 0; k0 < K; k0 += 128) {
        const int kc = min(128, K - k0);
        for (int idx = tid; idx < 32 * 32; idx += 256) {
            const int r = idx >> 5, q = idx & 31;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int row = row0 + r;
            if (row < R && q * 4 < kc) v = *reinterpret_cast<const float4*>(X + (size_t)row * ldx + k0 + q * 4);
            *reinterpret_cast<float4*>(Xs + r * TLD + q * 4) = v;
        }
        __syncthreads();
        const float* xr = Xs + g * 16 * TLD;
        for (int kk = 0; kk < kc; kk += 4) {
            const float w0 = Wt[(size_t)(k0 + kk) * ldw + cw], w1 = Wt[(size_t)(k0 + kk + 1) * ldw + cw];
            const float w2 = Wt[(size_t)(k0 + kk + 2) * ldw + cw], w3 = Wt[(size_t)(k0 + kk + 3) * ldw + cw];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 x = *reinterpret_cast<const float4*>(xr + r * TLD + kk);
                acc[r] = fmaf(x.x, w0, acc[r]); acc[r] = fmaf(x.y, w1, acc[r]);
                acc[r] = fmaf(x.z, w2, acc[r]); acc[r] = fmaf(x.w, w3, acc[r]);
            }
        }
        __syncthreads();
    }
    if (cc < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + g * 16 + r;
            if (row < R) {
                float* y = Y + (size_t)row * ldy + cc;
                *y = beta ? *y + acc[r] : acc[r];
            }
        }
    }
}
void t_gemm(const TRows& rows, const float* X, int ldx, int K, const float* Wt, int ldw, const float* bias, int N,
            float* Y, int ldy, int beta, hipStream_t s) {
    dim3 grid((rows.maxrows + 31) / 32, (N + 127) / 128);
    hipLaunchKernelGGL(k_tgemm, grid, dim3(256), 0, s, rows, X, ldx, K, Wt, ldw, bias, N, Y, ldy, beta);
}

// Ordered reduction of partial results: out[(i / cols) * ld_out + i % cols] += sum_{s < nparts} part[s * count + i], s ascending.
// Every cross-workgroup sum of the backward goes through this (no float atomics): gradients are bit-reproducible.
// Elements i >= split_at belong to a second, dense output (the bias gradient riding behind a weight-gradient tile): out2[i - split_at].
__global__ void k_reduce_parts(const float* __restrict__ part, int nparts, size_t stride, int count, int cols,
                               float* __restrict__ out, int ld_out, int split_at, float* __restrict__ out2, int cols_keep,
                               int out2_keep, int wrap_rows, int wrap_shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
#pragma unroll 8
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * stride + i];       // (loads batched by the unroll, additions in order)
    if (i < split_at) {
        // (columns >= cols_keep: padding of the operand;  wrap_rows: output rows beyond it continue wrap_shift columns to the right -
        //  the [Wa | Wb] blocks of a first Linear's weight gradient, produced as one 256-row product)
        const int row = i / cols, c = i % cols;
        if (c < cols_keep) out[(size_t)(wrap_rows ? row % wrap_rows : row) * ld_out + (wrap_rows ? (row / wrap_rows) * wrap_shift : 0) + c] += s;
    } else if (i - split_at < out2_keep) out2[i - split_at] += s;
}
// first level of a two-level reduction: group g sums its contiguous run of partials (ascending) into tmp[g][count]
__global__ void k_reduce_groups(const float* __restrict__ part, int nparts, size_t stride, int count, int per_group,
                                float* __restrict__ tmp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (i >= count) return;
    const int p0 = g * per_group, p1 = min(nparts, p0 + per_group);
    float s = 0.f;
#pragma unroll 8
    for (int p = p0; p < p1; ++p) s += part[(size_t)p * stride + i];
    tmp[(size_t)g * count + i] = s;
}
// ---- deferred, batched form (the backward of a training step): a kernel boundary costs ~4.7 us on this part and a step has ~180 of these
// reductions, each a few microseconds of work.  While a queue is active (red_begin .. red_end, one per backward), every producer takes its
// partial buffer from a bump arena (red_acquire) and reduce_parts only RECORDS the job; k_reduce_batch runs up to RED_MAX jobs per launch
// (job table by value in the kernel arguments) when the arena is full, a job targets an output a pending job also targets, a gradient chunk
// becomes final (red_flush) or the backward ends.  Per job the association order is a function of nparts alone: four contiguous runs of
// partials summed in ascending order, then ((g0 + g1) + g2) + g3 - bit-reproducible.
struct RedJob {
    const float* part; float* out; float* out2; size_t stride;
    int nparts, count, cols, ld_out, split_at, cols_keep, out2_keep, wrap_rows, wrap_shift, blk0;
};
#define RED_MAX 48
struct RedBatch { RedJob j[RED_MAX]; int n; };
static_assert(sizeof(RedBatch) <= 4000, "job table travels in the kernel arguments");
__global__ void __launch_bounds__(256) k_reduce_batch(RedBatch b) {
    __shared__ float red[4][64];
    int jb = 0;
    for (int t = 1; t < b.n; ++t) if ((int)blockIdx.x >= b.j[t].blk0) jb = t;      // (blk0 ascending; uniform)
    const RedJob& J = b.j[jb];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int i = ((int)blockIdx.x - J.blk0) * 64 + lane;
    const int per = (J.nparts + 3) >> 2, p0 = g * per, p1 = min(J.nparts, p0 + per);
    float s = 0.f;
    if (i < J.count) {
#pragma unroll 8
        for (int p = p0; p < p1; ++p) s += J.part[(size_t)p * J.stride + i];
    }
    red[g][lane] = s;
    __syncthreads();
    if (g != 0 || i >= J.count) return;
    s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    if (i < J.split_at) {
        const int row = i / J.cols, c = i % J.cols;
        if (c < J.cols_keep)
            J.out[(size_t)(J.wrap_rows ? row % J.wrap_rows : row) * J.ld_out + (J.wrap_rows ? (row / J.wrap_rows) * J.wrap_shift : 0) + c] += s;
    } else if (i - J.split_at < J.out2_keep) J.out2[i - J.split_at] += s;
}
struct RedQueue {
    bool active = false;
    float* arena = nullptr; size_t floats = 0, used = 0;
    hipStream_t s = nullptr;
    RedBatch b;
    int blocks = 0;
};
static thread_local RedQueue g_rq;
static constexpr size_t RED_VIEW = (size_t)16 << 20;       // what one producer may assume (the pre-queue scratch size)
void red_flush() {
    RedQueue& q = g_rq;
    if (q.active && q.b.n > 0) hipLaunchKernelGGL(k_reduce_batch, dim3(q.blocks), dim3(256), 0, q.s, q.b);
    q.b.n = 0; q.blocks = 0; q.used = 0;
}
void red_begin(const TScratch& sc, hipStream_t s) {
    RedQueue& q = g_rq;
    const char* off = getenv("RNAMPNN_NO_RED_BATCH");      // A/B switch (read per call): one reduction launch per producer, as before
    q.active = sc.floats >= 2 * RED_VIEW && !(off && off[0] == '1');
    q.arena = sc.p; q.floats = sc.floats; q.used = 0; q.s = s; q.b.n = 0; q.blocks = 0;
}
void red_end() { red_flush(); g_rq.active = false; }
// the scratch a producer works in: the whole buffer without a queue; with one, the free tail of the arena (flushed first if a producer's
// worst case no longer fits)
TScratch red_acquire(const TScratch& sc) {
    RedQueue& q = g_rq;
    if (!q.active || sc.p != q.arena) return sc;
    if (q.floats - q.used < RED_VIEW) red_flush();
    return TScratch{q.arena + q.used, RED_VIEW};
}
static void reduce_parts(const float* part, int nparts, size_t stride, int count, int cols, float* out, int ld_out, hipStream_t s,
                         float* tmp = nullptr, int split_at = -1, float* out2 = nullptr, int cols_keep = -1, int out2_keep = -1,
                         int wrap_rows = 0, int wrap_shift = 0) {
    if (split_at < 0) split_at = count;
    if (cols_keep < 0) cols_keep = cols;
    if (out2_keep < 0) out2_keep = count;
    RedQueue& q = g_rq;
    if (q.active && s == q.s && part >= q.arena && part < q.arena + q.floats) {
        bool clash = q.b.n == RED_MAX;
        for (int t = 0; t < q.b.n && !clash; ++t) {
            const RedJob& J = q.b.j[t];
            clash = J.out == out || (out2 && (J.out2 == out2 || J.out == out2)) || (J.out2 && J.out2 == out);
        }
        const size_t keep_used = q.used;
        if (clash) { red_flush(); q.used = keep_used; }     // (the partials of THIS job are already in the arena: keep its extent)
        RedJob& J = q.b.j[q.b.n++];
        J = RedJob{part, out, out2, stride, nparts, count, cols, ld_out, split_at, cols_keep, out2_keep, wrap_rows, wrap_shift, q.blocks};
        q.blocks += (count + 63) / 64;
        const size_t end = (size_t)(part - q.arena) + (size_t)(nparts - 1) * stride + (size_t)count;
        if (end > q.used) q.used = (end + 63) & ~(size_t)63;
        return;
    }
    if (tmp && nparts > 96) {
        const int G = 16, per = (nparts + G - 1) / G;
        hipLaunchKernelGGL(k_reduce_groups, dim3((count + 255) / 256, G), dim3(256), 0, s, part, nparts, stride, count, per, tmp);
        hipLaunchKernelGGL(k_reduce_parts, dim3((count + 255) / 256), dim3(256), 0, s, tmp, G, (size_t)count, count, cols, out, ld_out, split_at, out2, cols_keep, out2_keep, wrap_rows, wrap_shift);
        return;
    }
    hipLaunchKernelGGL(k_reduce_parts, dim3((count + 255) / 256), dim3(256), 0, s, part, nparts, stride, count, cols, out, ld_out, split_at, out2, cols_keep, out2_keep, wrap_rows, wrap_shift);
}

// dW[m][k] += sum_p A[p][m] * B[p][k]   (32 x 32 tile per block; the row range is split over blockIdx.z, every split
// writes its own partial tile, k_reduce_parts adds them in split order)
__global__ void __launch_bounds__(256) k_tgemm_tn(TRows rows, const float* __restrict__ A, int lda, int M,
        const float* __restrict__ B, int ldb, int K, float* __restrict__ part, int rows_per_split) {
    __shared__ float As[32][33], Bs[32][33];
    const int R = nrows(rows);
    const int p_begin = blockIdx.z * rows_per_split;
    const int p_end = min(R, p_begin + rows_per_split);
    const int m0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tid = threadIdx.x, tm = tid >> 4, tk = tid & 15;        // thread -> m in {tm, tm+16}, k in {tk, tk+16}
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
    for (int p0 = p_begin; p0 < p_end; p0 += 32) {
        for (int idx = tid; idx < 1024; idx += 256) {
            const int r = idx >> 5, cidx = idx & 31;
            const int p = p0 + r;
            As[r][cidx] = (p < p_end && m0 + cidx < M) ? A[(size_t)p * lda + m0 + cidx] : 0.f;
            Bs[r][cidx] = (p < p_end && k0 + cidx < K) ? B[(size_t)p * ldb + k0 + cidx] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const float x0 = As[r][tm], x1 = As[r][tm + 16], y0 = Bs[r][tk], y1 = Bs[r][tk + 16];
            a00 = fmaf(x0, y0, a00); a01 = fmaf(x0, y1, a01); a10 = fmaf(x1, y0, a10); a11 = fmaf(x1, y1, a11);
        }
        __syncthreads();
    }
    float* dst = part + (size_t)blockIdx.z * M * K;                   // splits beyond the live rows store zeros
    const int mA = m0 + tm, mB = m0 + tm + 16, kA = k0 + tk, kB = k0 + tk + 16;
    if (mA < M && kA < K) dst[(size_t)mA * K + kA] = a00;
    if (mA < M && kB < K) dst[(size_t)mA * K + kB] = a01;
    if (mB < M && kA < K) dst[(size_t)mB * K + kA] = a10;
    if (mB < M && kB < K) dst[(size_t)mB * K + kB] = a11;
}
void t_gemm_tn(const TRows& rows, const float* A, int lda, int M, const float* B, int ldb, int K, float* dW, int ldw,
               const TScratch& sc_in, hipStream_t s) {
    const TScratch sc = red_acquire(sc_in);
    long long cap = (long long)(sc.floats / ((size_t)M * K));
    int splits = (rows.maxrows + 2047) / 2048;
    if (splits > 512) splits = 512;
    if (splits > cap) splits = (int)cap;
    if (splits < 1) splits = 1;
    int rps = ((rows.maxrows + splits - 1) / splits + 31) / 32 * 32;
    dim3 grid((M + 31) / 32, (K + 31) / 32, splits);
    hipLaunchKernelGGL(k_tgemm_tn, grid, dim3(256), 0, s, rows, A, lda, M, B, ldb, K, sc.p, rps);
    reduce_parts(sc.p, splits, (size_t)M * K, M * K, K, dW, ldw, s);
}

// out[m] += sum_p A[p][m]   (per-block partial rows, ordered reduction)
__global__ void __launch_bounds__(256) k_colsum(TRows rows, const float* __restrict__ A, int lda, int M, float* __restrict__ part,
                                                 int rows_per_block) {
    const int R = nrows(rows);
    const int p0 = blockIdx.x * rows_per_block, p1 = min(R, p0 + rows_per_block);
    for (int m = threadIdx.x; m < M; m += 256) {
        float s = 0.f;
        for (int p = p0; p < p1; ++p) s += A[(size_t)p * lda + m];
        part[(size_t)blockIdx.x * M + m] = s;
    }
}
void t_colsum(const TRows& rows, const float* A, int lda, int M, float* out, const TScratch& sc_in, hipStream_t s) {
    const TScratch sc = red_acquire(sc_in);
    int nb = (int)(sc.floats / (size_t)M) - 16;
    if (nb > 512) nb = 512;
    int rpb = (rows.maxrows + nb - 1) / nb;
    if (rpb < 64) rpb = 64;
    nb = (rows.maxrows + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_colsum, dim3(nb), dim3(256), 0, s, rows, A, lda, M, sc.p, rpb);
    reduce_parts(sc.p, nb, (size_t)M, M, M, out, M, s, sc.p + (size_t)nb * M);
}

// element-wise over rows x D (contiguous, ld = D)
__global__ void k_gelu_fwd(TRows rows, const float* __restrict__ x, float* __restrict__ y, int D, TDrop dr, unsigned site) {
    const size_t n = (size_t)nrows(rows) * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = gelu_f(x[i]) * drop_mul(dr, site, i);
}
__global__ void k_gelu_bwd(TRows rows, const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dx, int D,
                           TDrop dr, unsigned site) {
    const size_t n = (size_t)nrows(rows) * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = dy[i] * gelu_d(pre[i]) * drop_mul(dr, site, i);
}
__global__ void k_add(TRows rows, const float* __restrict__ a, float* __restrict__ dst, int D) {
    const size_t n = (size_t)nrows(rows) * D;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += a[i];
}
static unsigned ew_grid(const TRows& r, int D) { size_t g = ((size_t)r.maxrows * D + 255) / 256; return (unsigned)(g < 16384 ? (g ? g : 1) : 16384); }
void t_gelu_fwd(const TRows& rows, const float* x, float* y, int D, const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_gelu_fwd, dim3(ew_grid(rows, D)), dim3(256), 0, s, rows, x, y, D, dr, site);
}
void t_gelu_bwd(const TRows& rows, const float* dy, const float* pre, float* dx, int D, const TDrop& dr, unsigned site,
                hipStream_t s) {
    hipLaunchKernelGGL(k_gelu_bwd, dim3(ew_grid(rows, D)), dim3(256), 0, s, rows, dy, pre, dx, D, dr, site);
}
void t_add(const TRows& rows, const float* a, float* dst, int D, hipStream_t s) {
    hipLaunchKernelGGL(k_add, dim3(ew_grid(rows, D)), dim3(256), 0, s, rows, a, dst, D);
}

// ------------------------------------------------------------------------------------------
// per-edge helpers; edge row = p*k + slot, 128 wide
__global__ void k_edge_features(PackInfo pk, int k, const float* __restrict__ geom, const int* __restrict__ nbr, float* __restrict__ F) {
    const int E = pk.cu[pk.B] * k;
    const int eid = blockIdx.x * blockDim.x + threadIdx.x;
    if (eid >= E) return;
    float* x = F + (size_t)eid * RN_ERAWP;
    const int j = nbr[eid];
    if (j < 0) { for (int i = 0; i < RN_ERAWP; ++i) x[i] = 0.f; return; }
    const float* gi = geom + (size_t)(eid / k) * RN_GEOM;
    const float* gj = geom + (size_t)j * RN_GEOM;
    for (int a = 0; a < 7; ++a)
        for (int b = 0; b < 7; ++b) {
            float dx = gi[a * 3] - gj[b * 3], dy = gi[a * 3 + 1] - gj[b * 3 + 1], dz = gi[a * 3 + 2] - gj[b * 3 + 2];
            x[a * 7 + b] = sqrtf(dx * dx + dy * dy + dz * dz + kSEPS);
        }
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b)
            x[49 + a * 5 + b] = gi[21 + a * 3] * gj[21 + b * 3] + gi[22 + a * 3] * gj[22 + b * 3] + gi[23 + a * 3] * gj[23 + b * 3];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
            x[74 + a * 4 + b] = gi[36 + a * 3] * gj[36 + b * 3] + gi[37 + a * 3] * gj[37 + b * 3] + gi[38 + a * 3] * gj[38 + b * 3];
    for (int i = RN_ERAW; i < RN_ERAWP; ++i) x[i] = 0.f;
}
// bf16 variant for the mixed trainer: [E][128], columns 90..127 and absent edges zero
__global__ void k_edge_features_b(PackInfo pk, int k, const float* __restrict__ geom, const int* __restrict__ nbr, unsigned short* __restrict__ F) {
    const int E = pk.cu[pk.B] * k;
    const int eid = blockIdx.x * blockDim.x + threadIdx.x;
    if (eid >= E) return;
    uint4* x = reinterpret_cast<uint4*>(F + (size_t)eid * RN_D);
    const int j = nbr[eid];
    if (j < 0) { for (int i = 0; i < 16; ++i) x[i] = make_uint4(0u, 0u, 0u, 0u); return; }
    const float* gi = geom + (size_t)(eid / k) * RN_GEOM;
    const float* gj = geom + (size_t)j * RN_GEOM;
    float v[96];
    for (int a = 0; a < 7; ++a)
        for (int b = 0; b < 7; ++b) {
            float dx = gi[a * 3] - gj[b * 3], dy = gi[a * 3 + 1] - gj[b * 3 + 1], dz = gi[a * 3 + 2] - gj[b * 3 + 2];
            v[a * 7 + b] = sqrtf(dx * dx + dy * dy + dz * dz + kSEPS);
        }
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b)
            v[49 + a * 5 + b] = gi[21 + a * 3] * gj[21 + b * 3] + gi[22 + a * 3] * gj[22 + b * 3] + gi[23 + a * 3] * gj[23 + b * 3];
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b)
            v[74 + a * 4 + b] = gi[36 + a * 3] * gj[36 + b * 3] + gi[37 + a * 3] * gj[37 + b * 3] + gi[38 + a * 3] * gj[38 + b * 3];
    for (int i = RN_ERAW; i < 96; ++i) v[i] = 0.f;
    auto pk2 = [](float a, float b) {                     // round-to-nearest-even bf16 pair (tpack2 is defined further down)
        typedef __attribute__((ext_vector_type(2))) float f2; typedef __attribute__((ext_vector_type(2))) __bf16 b2;
        f2 t = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(t, b2));
    };
#pragma unroll
    for (int i = 0; i < 12; ++i) x[i] = make_uint4(pk2(v[8 * i], v[8 * i + 1]), pk2(v[8 * i + 2], v[8 * i + 3]), pk2(v[8 * i + 4], v[8 * i + 5]), pk2(v[8 * i + 6], v[8 * i + 7]));
    for (int i = 12; i < 16; ++i) x[i] = make_uint4(0u, 0u, 0u, 0u);
}
void te_edge_features(const PackInfo& pk, int k, const float* geom, const int* nbr, tb16* F, hipStream_t s) {
    size_t total = (size_t)pk.Nmax * k;
    hipLaunchKernelGGL(k_edge_features_b, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, pk, k, geom, nbr, F);
}
void t_edge_features(const PackInfo& pk, int k, const float* geom, const int* nbr, float* F, hipStream_t s) {
    size_t total = (size_t)pk.Nmax * k;
    hipLaunchKernelGGL(k_edge_features, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, s, pk, k, geom, nbr, F);
}

// mode 0: pre[e] += P[i] + Q[j];  mode 1: x[e] = 0 on invalid slots;  mode 2: dst[e] = src1[e] + (valid ? gelu(src2[e]) : 0)
__global__ void k_edge_elem(PackInfo pk, int k, const int* __restrict__ nbr, int mode, const float* __restrict__ pq,
                            float* __restrict__ x, const float* __restrict__ src1, const float* __restrict__ src2,
                            TDrop dr, unsigned site) {
    const size_t n = (size_t)pk.cu[pk.B] * k * 32;               // float4 units
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(id & 31);
        const size_t er = id >> 5;
        const int j = nbr[er];
        float4* xp = reinterpret_cast<float4*>(x + er * RN_D) + q;
        if (mode == 0) {
            const int i = (int)(er / k);
            const int jj = j < 0 ? pk.Nmax : (j > pk.Nmax ? pk.Nmax : j);
            const float4 p = reinterpret_cast<const float4*>(pq + (size_t)i * 256)[q];
            const float4 qq = reinterpret_cast<const float4*>(pq + (size_t)jj * 256 + 128)[q];
            float4 v = *xp;
            v.x += p.x + qq.x; v.y += p.y + qq.y; v.z += p.z + qq.z; v.w += p.w + qq.w;
            *xp = v;
        } else if (mode == 1) {
            if (j < 0) *xp = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            float4 a = reinterpret_cast<const float4*>(src1 + er * RN_D)[q];
            if (j >= 0) {
                const float4 g = reinterpret_cast<const float4*>(src2 + er * RN_D)[q];
                const unsigned long long i0 = (unsigned long long)er * RN_D + 4 * q;
                a.x += gelu_f(g.x) * drop_mul(dr, site, i0); a.y += gelu_f(g.y) * drop_mul(dr, site, i0 + 1);
                a.z += gelu_f(g.z) * drop_mul(dr, site, i0 + 2); a.w += gelu_f(g.w) * drop_mul(dr, site, i0 + 3);
            }
            *xp = a;
        }
    }
}
static unsigned edge_grid(const PackInfo& pk, int k) { size_t g = ((size_t)pk.Nmax * k * 32 + 255) / 256; return (unsigned)(g < 16384 ? (g ? g : 1) : 16384); }
void t_edge_add_pq(const PackInfo& pk, int k, const int* nbr, const float* pq, float* pre, hipStream_t s) {
    hipLaunchKernelGGL(k_edge_elem, dim3(edge_grid(pk, k)), dim3(256), 0, s, pk, k, nbr, 0, pq, pre, nullptr, nullptr, TDrop{0, 0, 1.f}, 0u);
}
void t_edge_zero_invalid(const PackInfo& pk, int k, const int* nbr, float* x, hipStream_t s) {
    hipLaunchKernelGGL(k_edge_elem, dim3(edge_grid(pk, k)), dim3(256), 0, s, pk, k, nbr, 1, nullptr, x, nullptr, nullptr, TDrop{0, 0, 1.f}, 0u);
}
void t_edge_residual(const PackInfo& pk, int k, const int* nbr, const float* e_in, const float* pre2, float* e_out,
                     const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_edge_elem, dim3(edge_grid(pk, k)), dim3(256), 0, s, pk, k, nbr, 2, nullptr, e_out, e_in, pre2, dr, site);
}

// forward: out[p] = h[p] + sum_valid gelu(pre2[e]) / max(cnt,1);   backward: dpre2[e] = valid ? dagg[p]/cnt * gelu'(pre2[e]) : 0
__global__ void __launch_bounds__(128) k_seg_mean(PackInfo pk, int k, const int* __restrict__ nbr, const float* __restrict__ pre2,
                                                  const float* __restrict__ h, float* __restrict__ out, TDrop dr, unsigned site) {
    const int p = blockIdx.x;
    if (p >= pk.cu[pk.B]) return;
    const int c = threadIdx.x;
    float s = 0.f; int cnt = 0;
    for (int sl = 0; sl < k; ++sl) {
        if (nbr[(size_t)p * k + sl] >= 0) {
            const size_t o = ((size_t)p * k + sl) * RN_D + c;
            s += gelu_f(pre2[o]) * drop_mul(dr, site, o);
            ++cnt;
        }
    }
    out[(size_t)p * RN_D + c] = h[(size_t)p * RN_D + c] + s / (float)(cnt > 0 ? cnt : 1);
}
__global__ void __launch_bounds__(128) k_seg_mean_bwd(PackInfo pk, int k, const int* __restrict__ nbr, const float* __restrict__ dagg,
                                                      const float* __restrict__ pre2, float* __restrict__ dpre2, TDrop dr, unsigned site) {
    const int p = blockIdx.x;
    if (p >= pk.cu[pk.B]) return;
    const int c = threadIdx.x;
    int cnt = 0;
    for (int sl = 0; sl < k; ++sl) cnt += nbr[(size_t)p * k + sl] >= 0;
    const float g = dagg[(size_t)p * RN_D + c] / (float)(cnt > 0 ? cnt : 1);
    for (int sl = 0; sl < k; ++sl) {
        const size_t o = ((size_t)p * k + sl) * RN_D + c;
        dpre2[o] = nbr[(size_t)p * k + sl] >= 0 ? g * gelu_d(pre2[o]) * drop_mul(dr, site, o) : 0.f;
    }
}
void t_seg_mean(const PackInfo& pk, int k, const int* nbr, const float* pre2, const float* h, float* out,
                const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_seg_mean, dim3(pk.Nmax), dim3(128), 0, s, pk, k, nbr, pre2, h, out, dr, site);
}
void t_seg_mean_bwd(const PackInfo& pk, int k, const int* nbr, const float* dagg, const float* pre2, float* dpre2,
                    const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_seg_mean_bwd, dim3(pk.Nmax), dim3(128), 0, s, pk, k, nbr, dagg, pre2, dpre2, dr, site);
}
// dpre2e[e] = valid ? de[e] * gelu'(pre2e[e]) : 0   (edge-update residual branch)
__global__ void k_edge_res_bwd(PackInfo pk, int k, const int* __restrict__ nbr, const float* __restrict__ de, const float* __restrict__ pre2,
                               float* __restrict__ dpre2, TDrop dr, unsigned site) {
    const size_t n = (size_t)pk.cu[pk.B] * k * RN_D;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (size_t)gridDim.x * blockDim.x)
        dpre2[id] = nbr[id >> 7] >= 0 ? de[id] * gelu_d(pre2[id]) * drop_mul(dr, site, id) : 0.f;
}
void t_edge_res_bwd(const PackInfo& pk, int k, const int* nbr, const float* de, const float* pre2, float* dpre2,
                    const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_edge_res_bwd, dim3(edge_grid(pk, k)), dim3(256), 0, s, pk, k, nbr, de, pre2, dpre2, dr, site);
}
// Reverse adjacency of the k-NN graph (built once per training forward): for every packed row j the edge rows
// (p*k + slot) whose neighbour is j, ascending - the gather form of the backward's only true scatter (d Q[j] += ...).
// Counting uses integer atomics (exact), the fill order is then made canonical by a per-row sort.
__global__ void k_rev_count(PackInfo pk, int k, const int* __restrict__ nbr, int* __restrict__ deg) {
    const size_t E = (size_t)pk.cu[pk.B] * k;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (size_t)gridDim.x * blockDim.x) {
        const int j = nbr[e];
        if (j >= 0 && j < pk.Nmax) atomicAdd(deg + j, 1);              // phantom neighbours (>= Nmax) carry no gradient
    }
}
__global__ void __launch_bounds__(1024) k_rev_scan(const int* __restrict__ deg, int n, int* __restrict__ start, int* __restrict__ fill) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, chunk = (n + 1023) / 1024;
    const int lo = min(n, tid * chunk), hi = min(n, lo + chunk);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += deg[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int i = 0; i < 1024; ++i) { int v = part[i]; part[i] = run; run += v; } start[n] = run; }
    __syncthreads();
    int run = part[tid];
    for (int i = lo; i < hi; ++i) { start[i] = run; fill[i] = run; run += deg[i]; }
}
__global__ void k_rev_fill(PackInfo pk, int k, const int* __restrict__ nbr, int* __restrict__ fill, int* __restrict__ list) {
    const size_t E = (size_t)pk.cu[pk.B] * k;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (size_t)gridDim.x * blockDim.x) {
        const int j = nbr[e];
        if (j >= 0 && j < pk.Nmax) list[atomicAdd(fill + j, 1)] = (int)e;
    }
}
// The atomics of k_rev_fill leave every list in arrival order.  Final position of an entry = number of smaller entries of its list
// (lists are short, ~k, and L2-resident): one thread per entry, no serial per-node sort, the result does not depend on the arrival order.
__global__ void k_rev_rank(PackInfo pk, const int* __restrict__ nbr, const int* __restrict__ start, const int* __restrict__ unsorted,
                           int* __restrict__ list) {
    const int total = start[pk.cu[pk.B]];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int er = unsorted[t];
        const int j = nbr[er];
        const int a = start[j], b = start[j + 1];
        int rank = 0;
        for (int u = a; u < b; ++u) rank += unsorted[u] < er;
        list[a + rank] = er;
    }
}
void t_build_reverse(const PackInfo& pk, int k, const int* nbr, int* deg, int* start, int* fill, int* list, int* tmp, hipStream_t s) {
    launch_zero_bytes(deg, (size_t)pk.Nmax * sizeof(int), s, 8);
    size_t E = (size_t)pk.Nmax * k;
    unsigned g = (unsigned)((E + 255) / 256); if (g > 8192) g = 8192; if (g < 1) g = 1;
    hipLaunchKernelGGL(k_rev_count, dim3(g), dim3(256), 0, s, pk, k, nbr, deg);
    hipLaunchKernelGGL(k_rev_scan, dim3(1), dim3(1024), 0, s, deg, pk.Nmax, start, fill);
    hipLaunchKernelGGL(k_rev_fill, dim3(g), dim3(256), 0, s, pk, k, nbr, fill, tmp);
    hipLaunchKernelGGL(k_rev_rank, dim3(g), dim3(256), 0, s, pk, nbr, start, tmp, list);
}
// dpq[i][0:128] = sum_slots dpre1[(i, slot)]  (P part);  dpq[j][128:256] = sum over incoming edges of j, ascending (Q part)
__global__ void __launch_bounds__(128) k_edge_pq_bwd(PackInfo pk, int k, const float* __restrict__ dpre1,
                                                     const int* __restrict__ start, const int* __restrict__ list, float* __restrict__ dpq) {
    const int p = blockIdx.x;
    if (p >= pk.cu[pk.B]) return;
    const int c = threadIdx.x;
    float s = 0.f;
    for (int sl = 0; sl < k; ++sl) s += dpre1[((size_t)p * k + sl) * RN_D + c];
    dpq[(size_t)p * 256 + c] = s;
    float q = 0.f;
    for (int t = start[p]; t < start[p + 1]; ++t) q += dpre1[(size_t)list[t] * RN_D + c];
    dpq[(size_t)p * 256 + 128 + c] = q;
}
void t_edge_pq_bwd(const PackInfo& pk, int k, const float* dpre1, const int* start, const int* list, float* dpq, hipStream_t s) {
    hipLaunchKernelGGL(k_edge_pq_bwd, dim3(pk.Nmax), dim3(128), 0, s, pk, k, dpre1, start, list, dpq);
}

// ------------------------------------------------------------------------------------------
// GraphNormalization backward (functional.py:33-46), one workgroup per RNA, D = 128.
//   y = (x - mu)/sd * scale + shift,  sd^2 = [sum (x-mu)^2 + c mu^2]/n + eps,  c = T_tot - n
//   dx_i = g_i/sd + dL/dvar * 2 (x_i - mu)/n + dL/dmu / n,   g = dy * scale
//   dL/dvar = -0.5 sd^-3 sum g_i (x_i - mu),   dL/dmu = -sum g_i / sd + dL/dvar * 2 c mu / n
__global__ void __launch_bounds__(256) k_gn_bwd(PackInfo pk, const float* __restrict__ x, const float* __restrict__ dy,
        const float* __restrict__ scale, int t_tot, float* __restrict__ dx, float* __restrict__ part) {
    // grid (RNA, 32-channel chunk): thread = (channel c of the chunk, row group rg of 8); the 8 groups are folded in fixed order
    __shared__ float red[3][8][32];
    const int b = blockIdx.x;
    const int n = pk.len[b];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5, c = blockIdx.y * 32 + cl;
    if (n <= 0) { if (rg == 0) { part[(size_t)b * 256 + c] = 0.f; part[(size_t)b * 256 + 128 + c] = 0.f; } return; }      // [b][dscale 128 | dshift 128]
    const size_t base = (size_t)pk.cu[b] * RN_D;
    const float* xb = x + base;
    const float* gb = dy + base;
    auto fold = [&](int w) { float t = 0.f; for (int g = 0; g < 8; ++g) t += red[w][g][cl]; return t; };
    float s = 0.f;
    for (int r = rg; r < n; r += 8) s += xb[(size_t)r * RN_D + c];
    red[0][rg][cl] = s;
    __syncthreads();
    const float mean = fold(0) / (float)n;
    __syncthreads();
    float ss = 0.f, sg = 0.f, sgx = 0.f;
    for (int r = rg; r < n; r += 8) {
        const float d = xb[(size_t)r * RN_D + c] - mean, g = gb[(size_t)r * RN_D + c];
        ss = fmaf(d, d, ss); sg += g; sgx = fmaf(g, d, sgx);
    }
    red[0][rg][cl] = ss; red[1][rg][cl] = sg; red[2][rg][cl] = sgx;
    __syncthreads();
    const float cpad = (float)(t_tot - n), fn = (float)n;
    const float Ss = fold(0), Sg0 = fold(1), Sgx0 = fold(2);
    const float var = (Ss + cpad * mean * mean) / fn + kSEPS;
    const float sd = sqrtf(var), sc = scale[c];
    const float Sg = Sg0 * sc, Sgx = Sgx0 * sc;                 // sums of g = dy*scale
    const float dvar = -0.5f * Sgx / (var * sd);
    const float dmu = -Sg / sd + dvar * 2.f * cpad * mean / fn;
    float* db = dx + base;
    for (int r = rg; r < n; r += 8) {
        const float d = xb[(size_t)r * RN_D + c] - mean;
        db[(size_t)r * RN_D + c] = gb[(size_t)r * RN_D + c] * sc / sd + dvar * 2.f * d / fn + dmu / fn;
    }
    if (rg == 0) {
        part[(size_t)b * 256 + c] = Sgx0 / sd;
        part[(size_t)b * 256 + 128 + c] = Sg0;
    }
}
// ---- the same backward for LONG RNAs (T > GN_SPLIT_T): with one workgroup per (RNA, 32-channel chunk) a batch of a few 4,000-nt RNAs runs on 28 of
// the 256 CUs (231 us per call on the config-3 epoch).  Here the rows of an RNA are cut into `nsplit` ranges (grid.z): pass A accumulates the four
// sums of a range around a per-(RNA, channel) PIVOT (the RNA's first row: sum (x - p), sum (x - p)^2, sum g, sum g (x - p) - the shifted one-pass
// form, whose cancellation is relative to |mean - p| ~ sigma, not to |mean|), pass B folds the ranges in fixed order, forms the same
// quantities as k_gn_bwd and writes dx for its range.  Deterministic; x and dy are read twice instead of three / two times.
#define GN_SPLIT_T 512
#define GN_SPLIT_ROWS 256
__global__ void __launch_bounds__(256) k_gn_bwd_sums(PackInfo pk, const float* __restrict__ x, const float* __restrict__ dy, int nsplit,
                                                     float* __restrict__ sums) {          // sums [B][nsplit][4][128]
    __shared__ float red[4][8][32];
    const int b = blockIdx.x, z = blockIdx.z;
    const int n = pk.len[b];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5, c = blockIdx.y * 32 + cl;
    const int r0 = z * GN_SPLIT_ROWS, r1 = min(n, r0 + GN_SPLIT_ROWS);
    float s1 = 0.f, s2 = 0.f, sg = 0.f, sgx = 0.f;
    if (r0 < n) {
        const size_t base = (size_t)pk.cu[b] * RN_D;
        const float* xb = x + base;
        const float* gb = dy + base;
        const float pv = xb[c];
        for (int r = r0 + rg; r < r1; r += 8) {
            const float d = xb[(size_t)r * RN_D + c] - pv, g = gb[(size_t)r * RN_D + c];
            s1 += d; s2 = fmaf(d, d, s2); sg += g; sgx = fmaf(g, d, sgx);
        }
    }
    red[0][rg][cl] = s1; red[1][rg][cl] = s2; red[2][rg][cl] = sg; red[3][rg][cl] = sgx;
    __syncthreads();
    if (rg < 4) {
        float t = 0.f;
        for (int g = 0; g < 8; ++g) t += red[rg][g][cl];
        sums[(((size_t)b * nsplit + z) * 4 + rg) * 128 + c] = t;
    }
}
__global__ void __launch_bounds__(256) k_gn_bwd_apply(PackInfo pk, const float* __restrict__ x, const float* __restrict__ dy,
        const float* __restrict__ scale, int t_tot, int nsplit, const float* __restrict__ sums, float* __restrict__ dx, float* __restrict__ part) {
    const int b = blockIdx.x, z = blockIdx.z;
    const int n = pk.len[b];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5, c = blockIdx.y * 32 + cl;
    if (n <= 0) { if (rg == 0 && z == 0) { part[(size_t)b * 256 + c] = 0.f; part[(size_t)b * 256 + 128 + c] = 0.f; } return; }
    const int r0 = z * GN_SPLIT_ROWS, r1 = min(n, r0 + GN_SPLIT_ROWS);
    if (r0 >= n) return;
    float S1 = 0.f, S2 = 0.f, Sg0 = 0.f, Sgp = 0.f;
    const int used = (n + GN_SPLIT_ROWS - 1) / GN_SPLIT_ROWS;
    for (int q = 0; q < used; ++q) {                            // fixed order: every workgroup of the RNA forms the same totals
        const float* sp = sums + ((size_t)b * nsplit + q) * 4 * 128 + c;
        S1 += sp[0]; S2 += sp[128]; Sg0 += sp[256]; Sgp += sp[384];
    }
    const size_t base = (size_t)pk.cu[b] * RN_D;
    const float* xb = x + base;
    const float* gb = dy + base;
    const float fn = (float)n, cpad = (float)(t_tot - n);
    const float pv = xb[c];
    const float dm = S1 / fn, mean = pv + dm;                   // mean - pivot, mean
    const float Ss = S2 - fn * dm * dm;                         // sum (x - mean)^2
    const float Sgx0 = Sgp - dm * Sg0;                          // sum g (x - mean)
    const float var = (Ss + cpad * mean * mean) / fn + kSEPS;
    const float sd = sqrtf(var), sc = scale[c];
    const float Sg = Sg0 * sc, Sgx = Sgx0 * sc;
    const float dvar = -0.5f * Sgx / (var * sd);
    const float dmu = -Sg / sd + dvar * 2.f * cpad * mean / fn;
    float* db = dx + base;
    for (int r = r0 + rg; r < r1; r += 8) {
        const float d = xb[(size_t)r * RN_D + c] - mean;
        db[(size_t)r * RN_D + c] = gb[(size_t)r * RN_D + c] * sc / sd + dvar * 2.f * d / fn + dmu / fn;
    }
    if (rg == 0 && z == 0) {
        part[(size_t)b * 256 + c] = Sgx0 / sd;
        part[(size_t)b * 256 + 128 + c] = Sg0;
    }
}
void t_gn_bwd(const PackInfo& pk, const float* x, const float* dy, const float* scale, int t_tot, float* dx, float* dscale,
              float* dshift, const TScratch& sc_in, hipStream_t s) {
    const TScratch sc = red_acquire(sc_in);
    // per-RNA partials of (dscale, dshift), added in RNA order (needs B * 256 floats of scratch)
    const int nsplit = (pk.T + GN_SPLIT_ROWS - 1) / GN_SPLIT_ROWS;
    const size_t need = (size_t)pk.B * 256 + (size_t)pk.B * nsplit * 512;
    const char* nosplit = getenv("RNAMPNN_GN_NOSPLIT");          // A/B switch (read per call)
    if (pk.T > GN_SPLIT_T && need <= sc.floats && !(nosplit && nosplit[0] == '1')) {
        float* sums = sc.p + (size_t)pk.B * 256;
        hipLaunchKernelGGL(k_gn_bwd_sums, dim3(pk.B, 4, nsplit), dim3(256), 0, s, pk, x, dy, nsplit, sums);
        hipLaunchKernelGGL(k_gn_bwd_apply, dim3(pk.B, 4, nsplit), dim3(256), 0, s, pk, x, dy, scale, t_tot, nsplit, sums, dx, sc.p);
    } else {
        hipLaunchKernelGGL(k_gn_bwd, dim3(pk.B, 4), dim3(256), 0, s, pk, x, dy, scale, t_tot, dx, sc.p);
    }
    if (dscale) reduce_parts(sc.p, pk.B, 256, 128, 128, dscale, 128, s);
    if (dshift) reduce_parts(sc.p + 128, pk.B, 256, 128, 128, dshift, 128, s);
}

// ------------------------------------------------------------------------------------------
// Training attention over the valid keys of one RNA (head dim HD), qkv rows [q | k | v], f32, with dropout on the
// attention PROBABILITIES (nn.MultiheadAttention(dropout=p), functional.py:109): O_i = sum_j P_ij M_ij v_j, the softmax
// normaliser is the unmasked sum.  Mask element index = ((query row * heads + head) << 13) + key (keys < 8192).
__device__ __forceinline__ unsigned long long att_idx(int qrow, int heads, int hd, int j) {
    return (((unsigned long long)qrow * heads + hd) << 13) + (unsigned)j;
}
template <int HD>
__global__ void __launch_bounds__(64) k_attn_fwd_t(PackInfo pk, const float* __restrict__ qkv, int heads, float* __restrict__ out,
                                                   float* __restrict__ stat, TDrop dr, unsigned site) {
    const int b = blockIdx.x, hd = blockIdx.y;
    const int n = pk.len[b];
    const int qi = blockIdx.z * 64 + threadIdx.x;
    if (blockIdx.z * 64 >= n) return;
    const int base = pk.cu[b];
    const bool act = qi < n;
    const int qrow = base + (act ? qi : 0);
    const float scale = rsqrtf((float)HD);
    const float* qp = qkv + (size_t)qrow * 384 + hd * HD;
    float q[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = qp[d] * scale; acc[d] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    for (int j = 0; j < n; ++j) {
        const float* kj = qkv + (size_t)(base + j) * 384 + 128 + hd * HD;
        const float* vj = kj + 128;
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) sc = fmaf(q[d], kj[d], sc);
        const float mn = fmaxf(m, sc);
        const float corr = __expf(m - mn), pj = __expf(sc - mn);
        l = l * corr + pj;
        const float pm = pj * drop_mul(dr, site, att_idx(qrow, heads, hd, j));
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = fmaf(pm, vj[d], acc[d] * corr);
        m = mn;
    }
    if (act) {
        const float inv = 1.0f / l;
        float* op = out + (size_t)qrow * RN_D + hd * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = acc[d] * inv;
        if (stat) {                                  // softmax statistics of the row for the backward (tape: one key pass there instead of three)
            float* st = stat + ((size_t)qrow * heads + hd) * 3;
            st[0] = m; st[1] = l;
        }
    }
}
int t_attention_fwd(const PackInfo& pk, const float* qkv, int heads, float* out, float* stat, const TDrop& dr, unsigned site, hipStream_t s) {
    dim3 grid(pk.B, heads, (pk.T + 63) / 64);
    const int hd = RN_D / heads;
#define RN_ATT(H) if (hd == H) { hipLaunchKernelGGL(k_attn_fwd_t<H>, grid, dim3(64), 0, s, pk, qkv, heads, out, stat, dr, site); return 0; }
    RN_ATT(16) RN_ATT(32) RN_ATT(8) RN_ATT(64)
#undef RN_ATT
    return 1;
}

// backward.  The forward taped the row max m and the normaliser l of every (query, head) (stat[0..1]); delta_i = sum_j P_ij dP_ij with
//            dP_ij = M_ij (dO_i . v_j) equals dO_i . O_i (O = the forward output, dropout included), so no pass over the keys is needed for it.
//            pass 1 (thread per query): dq_i = scale * sum_j dS_ij k_j, dS = P (dP - delta); stat[2] = delta    (ONE pass over the keys; it was three)
//            pass 2 (thread per key):   dv_j = sum_i P_ij M_ij dO_i,  dk_j = scale * sum_i dS_ij q_i
template <int HD>
__global__ void __launch_bounds__(64) k_attn_bwd_q(PackInfo pk, const float* __restrict__ qkv, const float* __restrict__ O, const float* __restrict__ dO,
        float* __restrict__ dqkv, float* __restrict__ stat, int heads, TDrop dr, unsigned site) {
    const int b = blockIdx.x, hd = blockIdx.y;
    const int n = pk.len[b];
    const int qi = blockIdx.z * 64 + threadIdx.x;
    if (blockIdx.z * 64 >= n) return;
    const int base = pk.cu[b];
    const bool act = qi < n;
    const int qrow = base + (act ? qi : 0);
    const float scale = rsqrtf((float)HD);
    const float* row = qkv + (size_t)qrow * 384 + hd * HD;
    float q[HD], g[HD], dq[HD];
    float delta = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        q[d] = row[d] * scale; g[d] = dO[(size_t)qrow * RN_D + hd * HD + d]; dq[d] = 0.f;
        delta = fmaf(g[d], O[(size_t)qrow * RN_D + hd * HD + d], delta);
    }
    float* st = stat + ((size_t)qrow * heads + hd) * 3;
    const float m = st[0], linv = 1.0f / st[1];
    for (int j = 0; j < n; ++j) {
        const float* kj = qkv + (size_t)(base + j) * 384 + 128 + hd * HD;
        const float* vj = kj + 128;
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) { sc = fmaf(q[d], kj[d], sc); dp = fmaf(g[d], vj[d], dp); }
        const float ds = __expf(sc - m) * linv * (dp * drop_mul(dr, site, att_idx(qrow, heads, hd, j)) - delta);
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, kj[d], dq[d]);
    }
    if (act) {
        float* o = dqkv + (size_t)(base + qi) * 384 + hd * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = dq[d] * scale;
        st[2] = delta;
    }
}
template <int HD>
__global__ void __launch_bounds__(64) k_attn_bwd_kv(PackInfo pk, const float* __restrict__ qkv, const float* __restrict__ dO,
        float* __restrict__ dqkv, const float* __restrict__ stat, int heads, TDrop dr, unsigned site) {
    const int b = blockIdx.x, hd = blockIdx.y;
    const int n = pk.len[b];
    const int kj = blockIdx.z * 64 + threadIdx.x;
    if (blockIdx.z * 64 >= n) return;
    const int base = pk.cu[b];
    const bool act = kj < n;
    const int kc = act ? kj : 0;
    const float scale = rsqrtf((float)HD);
    const float* row = qkv + (size_t)(base + kc) * 384 + 128 + hd * HD;
    float kk[HD], vv[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { kk[d] = row[d]; vv[d] = row[128 + d]; dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < n; ++i) {
        const float* qi = qkv + (size_t)(base + i) * 384 + hd * HD;
        const float* gi = dO + (size_t)(base + i) * RN_D + hd * HD;
        const float* st = stat + ((size_t)(base + i) * heads + hd) * 3;
        float sc = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) { sc = fmaf(qi[d] * scale, kk[d], sc); dp = fmaf(gi[d], vv[d], dp); }
        const float mk = drop_mul(dr, site, att_idx(base + i, heads, hd, kc));
        const float pij = __expf(sc - st[0]) / st[1];
        const float ds = pij * (dp * mk - st[2]);
        const float pm = pij * mk;
#pragma unroll
        for (int d = 0; d < HD; ++d) { dv[d] = fmaf(pm, gi[d], dv[d]); dk[d] = fmaf(ds, qi[d] * scale, dk[d]); }
    }
    if (act) {
        float* o = dqkv + (size_t)(base + kj) * 384 + 128 + hd * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) { o[d] = dk[d]; o[128 + d] = dv[d]; }
    }
}
int t_attention_bwd(const PackInfo& pk, const float* qkv, const float* O, const float* dO, int heads, float* dqkv, float* stat,
                    const TDrop& dr, unsigned site, hipStream_t s) {
    dim3 grid(pk.B, heads, (pk.T + 63) / 64);
    const int hd = RN_D / heads;
#define RN_ATT(H) \
    if (hd == H) { hipLaunchKernelGGL(k_attn_bwd_q<H>, grid, dim3(64), 0, s, pk, qkv, O, dO, dqkv, stat, heads, dr, site); \
                   hipLaunchKernelGGL(k_attn_bwd_kv<H>, grid, dim3(64), 0, s, pk, qkv, dO, dqkv, stat, heads, dr, site); return 0; }
    RN_ATT(16) RN_ATT(32) RN_ATT(8) RN_ATT(64)
#undef RN_ATT
    return 1;
}

// ------------------------------------------------------------------------------------------
// loss = mean_valid CE(softmax(logits), label)  (softmax applied twice, rnampnn.py:151-154,201-204) and its gradient with
// respect to the logits.  Per-workgroup partial losses, summed in block order by a one-thread second stage.
__global__ void __launch_bounds__(256) k_loss_grad(PackInfo pk, const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                   float* __restrict__ dlogits, float* __restrict__ part) {
    __shared__ float red[4];
    const int ntot = pk.cu[pk.B];
    float local = 0.f;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < ntot; p += gridDim.x * blockDim.x) {
        const int b = pk.node_b[p];
        const int y = labels[(size_t)b * pk.T + (p - pk.cu[b])];
        const float4 z = reinterpret_cast<const float4*>(logits)[p];
        float zz[4] = {z.x, z.y, z.z, z.w}, pr[4], q[4];
        float mx = fmaxf(fmaxf(zz[0], zz[1]), fmaxf(zz[2], zz[3])), s = 0.f;
        for (int c = 0; c < 4; ++c) { pr[c] = expf(zz[c] - mx); s += pr[c]; }
        for (int c = 0; c < 4; ++c) pr[c] /= s;
        float mp = fmaxf(fmaxf(pr[0], pr[1]), fmaxf(pr[2], pr[3])), s2 = 0.f;
        for (int c = 0; c < 4; ++c) { q[c] = expf(pr[c] - mp); s2 += q[c]; }
        for (int c = 0; c < 4; ++c) q[c] /= s2;
        local += -logf(q[y]);
        float dp[4], dot = 0.f;
        for (int c = 0; c < 4; ++c) { dp[c] = (q[c] - (c == y ? 1.f : 0.f)) / (float)ntot; dot += dp[c] * pr[c]; }
        float4 o;
        o.x = pr[0] * (dp[0] - dot); o.y = pr[1] * (dp[1] - dot); o.z = pr[2] * (dp[2] - dot); o.w = pr[3] * (dp[3] - dot);
        reinterpret_cast<float4*>(dlogits)[p] = o;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)(ntot > 0 ? ntot : 1);
}
__global__ void k_loss_sum(const float* __restrict__ part, int n, float* __restrict__ loss) {
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += part[i];
    *loss = s;
}
void t_loss_grad(const PackInfo& pk, const float* logits, const int32_t* labels, float* dlogits, float* loss, const TScratch& sc,
                 hipStream_t s) {
    int grid = (pk.Nmax + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_loss_grad, dim3(grid), dim3(256), 0, s, pk, logits, labels, dlogits, sc.p);
    hipLaunchKernelGGL(k_loss_sum, dim3(1), dim3(1), 0, s, sc.p, grid, loss);
}

// d loss / d logits handed in by the caller (autograd) in the padded (B,T,4) layout -> packed rows
__global__ void k_unpack_dlogits(PackInfo pk, const float* __restrict__ src, float* __restrict__ dst) {
    const int ntot = pk.cu[pk.B];
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < ntot; p += gridDim.x * blockDim.x) {
        const int b = pk.node_b[p];
        reinterpret_cast<float4*>(dst)[p] = reinterpret_cast<const float4*>(src)[(size_t)b * pk.T + (p - pk.cu[b])];
    }
}
void t_pack_dlogits(const PackInfo& pk, const float* dlogits_padded, float* dlogits_p, hipStream_t s) {
    int grid = (pk.Nmax + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_unpack_dlogits, dim3(grid), dim3(256), 0, s, pk, dlogits_padded, dlogits_p);
}

// ==========================================================================================
// bf16-MIXED training GEMMs on MFMA (the reference trains under bf16 autocast, rnampnn/utils/train.py:109: matmul
// operands bf16, accumulation and everything else f32).  f32 tensors in HBM, converted to bf16 while the operand
// fragments are loaded - straight from global memory into registers: every flavour below has one operand whose
// k index runs along ROWS of a row-major tensor (a transposed read), which an LDS staging pass would have to
// scatter element by element, while a fragment is just 8 coalesced dword loads per lane (lane = row / column of the
// tile, the wave covers 32 consecutive floats of 8 rows); reuse across waves comes out of L1/L2.
// v_mfma_f32_32x32x16_bf16: A fragment lane (r, h) = A[row r][k = 8h + j], B fragment = B[k = 8h + j][col r].
// Optional fused element-wise work (saves whole passes over the [E][128] f32 tape):
//   operand prologue  a = drop(gelu(pre))         (GELU + dropout of a taped pre-activation, never materialised)
//   result epilogue   y = acc * gelu'(pre) * mask (backward through GELU + dropout)
typedef __attribute__((ext_vector_type(8))) __bf16 tbf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 tbf16x2;
typedef __attribute__((ext_vector_type(2))) float tf32x2;
typedef __attribute__((ext_vector_type(4))) float tf32x4;
typedef __attribute__((ext_vector_type(16))) float tf32x16;
typedef __attribute__((ext_vector_type(4))) unsigned tu32x4;
__device__ __forceinline__ unsigned tpack2(float a, float b) {
    tf32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, tbf16x2));
}
__device__ __forceinline__ tf32x16 tmfma(tu32x4 a, tu32x4 b, tf32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tbf16x8, a), __builtin_bit_cast(tbf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ tu32x4 tpack8(const float (&v)[8]) {
    return tu32x4{tpack2(v[0], v[1]), tpack2(v[2], v[3]), tpack2(v[4], v[5]), tpack2(v[6], v[7])};
}
// fragment whose k index runs along a ROW of src: 8 consecutive floats at src[row][k0 .. k0+7] (two 16-byte loads);
// act: a = drop(gelu(x)), dropout index row * ld_idx + k
__device__ __forceinline__ tu32x4 frag_row(const float* __restrict__ src, int ld, int row, int nrows, int k0, bool act,
                                           const TDrop& dr, unsigned site, int ld_idx) {
    const bool ok = row < nrows;
    const float* p = src + (size_t)(ok ? row : 0) * ld + k0;
    const tf32x4 a = *reinterpret_cast<const tf32x4*>(p), b = *reinterpret_cast<const tf32x4*>(p + 4);
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (act) {                                  // (ld_idx and k0 are multiples of 8)
        float dm[8];
        drop8(dr, drop_key(dr, site), (unsigned)(((unsigned long long)row * ld_idx + k0) >> 3), dm);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_fast(v[j]) * dm[j];
    }
    if (!ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
    }
    return tpack8(v);
}
// fragment whose k index runs DOWN a column of src: src[k0 + j][col], j = 0..7 (8 dword loads, coalesced across lanes)
__device__ __forceinline__ tu32x4 frag_col(const float* __restrict__ src, int ld, int k0, int nk, int col, int ncols, bool act,
                                           const TDrop& dr, unsigned site, int ld_idx) {
    float v[8];
    const int cc = col < ncols ? col : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int kk = k0 + j; v[j] = src[(size_t)(kk < nk ? kk : 0) * ld + cc]; }
    if (act) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_fast(v[j]) * drop_mul(dr, site, (unsigned long long)(k0 + j) * ld_idx + col);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (k0 + j >= nk || col >= ncols) v[j] = 0.f;
    return tpack8(v);
}

// ---- bf16 fragment images of 128 x 128 weight blocks (32 KiB: [k-step 8][channel block 4][lane 64][8 bf16]) for the
// weights-resident GEMM kernels.  layout 0 (k_mm128): B fragments, lane (r, h) = column 32cb + r; layout 1 (k_emm128): A fragments with
// the row permutation that makes a lane's accumulators two runs of 8 consecutive channels.  A kernel builds its image itself
// (32 rounds of scattered 4-byte loads per thread, ~10 us of serial latency before the first tile) unless the caller's WImageCache has it:
// the cache rebuilds ALL registered images in one launch at the start of a training forward.
struct WImgDesc { const float* W; int ldw; int b_rows; int layout; int kvalid; };     // k >= kvalid: zeros (the 90-wide raw edge features)
__device__ __forceinline__ void build_wimage(unsigned short* img, const float* __restrict__ W, int ldw, bool b_rows, int layout, int kvalid, int tid, int nthreads) {
    for (int e = tid; e < 128 * 64; e += nthreads) {
        int k, c;
        float v0, v1;
        if (b_rows) { c = e >> 6; k = 2 * (e & 63); const float* p = W + (size_t)c * ldw + k; v0 = k < kvalid ? p[0] : 0.f; v1 = k + 1 < kvalid ? p[1] : 0.f; }
        else { k = 2 * (e >> 7); c = e & 127; v0 = k < kvalid ? W[(size_t)k * ldw + c] : 0.f; v1 = k + 1 < kvalid ? W[(size_t)(k + 1) * ldw + c] : 0.f; }
        const int ks = k >> 4, hh = (k >> 3) & 1, j = k & 7, cb = c >> 5, c5 = c & 31;
        int rr = c5;
        if (layout == 1) { const int i = 8 * (c5 >> 4) + (c5 & 7), hq = (c5 >> 3) & 1; rr = (i & 3) + 8 * (i >> 2) + 4 * hq; }
        *reinterpret_cast<unsigned*>(img + (((ks * 4 + cb) * 64 + hh * 32 + rr) * 8 + j)) = tpack2(v0, v1);
    }
}
__device__ __forceinline__ void stage_wimage(unsigned short* img, const unsigned short* __restrict__ prebuilt, const float* __restrict__ W, int ldw,
                                             bool b_rows, int layout, int tid, int kvalid = 128) {
    if (prebuilt) {
        for (int e = tid; e < 2048; e += 256) reinterpret_cast<tu32x4*>(img)[e] = reinterpret_cast<const tu32x4*>(prebuilt)[e];
    } else {
        build_wimage(img, W, ldw, b_rows, layout, kvalid, tid, 256);
    }
}
__global__ void __launch_bounds__(256) k_wimg_build(const WImgDesc* __restrict__ desc, unsigned short* __restrict__ arena) {
    const WImgDesc d = desc[blockIdx.x];
    build_wimage(arena + (size_t)blockIdx.x * 16384, d.W, d.ldw, d.b_rows != 0, d.layout, d.kvalid, threadIdx.x, 256);
}
struct WImageCache {
    std::vector<WImgDesc> host;
    WImgDesc* dev = nullptr;
    unsigned short* arena = nullptr;
    int cap = 0, synced = 0, built = 0;
};
static thread_local WImageCache* g_wimg = nullptr;
WImageCache* t_wimg_create(int capacity) {
    WImageCache* c = new WImageCache();
    c->cap = capacity;
    if (hipMalloc((void**)&c->dev, sizeof(WImgDesc) * capacity) != hipSuccess || hipMalloc((void**)&c->arena, (size_t)capacity * 32768) != hipSuccess) {
        if (c->dev) (void)hipFree(c->dev);
        delete c;
        return nullptr;
    }
    c->host.reserve(capacity);
    return c;
}
void t_wimg_destroy(WImageCache* c) {
    if (!c) return;
    if (g_wimg == c) g_wimg = nullptr;
    (void)hipFree(c->dev); (void)hipFree(c->arena);
    delete c;
}
void t_wimg_bind(WImageCache* c) { g_wimg = c; }
int t_wimg_pending(const WImageCache* c) { return c ? (int)c->host.size() - c->built : 0; }
void t_wimg_clear(WImageCache* c) { if (c) { c->host.clear(); c->synced = 0; c->built = 0; } }
void t_wimg_refresh(WImageCache* c, hipStream_t s) {
    if (!c || c->host.empty()) return;
    if ((int)c->host.size() > c->synced) {        // pageable source: the copy is staged before the call returns
        (void)hipMemcpyAsync(c->dev + c->synced, c->host.data() + c->synced, sizeof(WImgDesc) * (c->host.size() - c->synced), hipMemcpyHostToDevice, s);
        c->synced = (int)c->host.size();
    }
    hipLaunchKernelGGL(k_wimg_build, dim3(c->synced), dim3(256), 0, s, c->dev, c->arena);
    c->built = c->synced;
}
// image of (W, ldw, orientation, layout) if the bound cache has built it; unknown blocks are registered for the next refresh
static const unsigned short* wimg_lookup(const float* W, int ldw, bool b_rows, int layout, int kvalid = 128) {
    WImageCache* c = g_wimg;
    if (!c) return nullptr;
    const int n = (int)c->host.size();
    for (int i = 0; i < n; ++i) {
        const WImgDesc& d = c->host[i];
        if (d.W == W && d.ldw == ldw && d.b_rows == (b_rows ? 1 : 0) && d.layout == layout && d.kvalid == kvalid) return i < c->built ? c->arena + (size_t)i * 16384 : nullptr;
    }
    if (n < c->cap) c->host.push_back(WImgDesc{W, ldw, b_rows ? 1 : 0, layout, kvalid});
    return nullptr;
}

// ---- NT / NN: Y[m][n] (+)= sum_k A[m][k] * B[k][n]  (+ bias[n]) (* gelu'(pre[m][n]) * mask)
//   A = X rows (k along the row; optional GELU+dropout prologue);  B_NT: B[k][n] = W[n][k] (W row-major [N][K]);
//   B_NN: B[k][n] = W[k][n] (W row-major [K][N]).  Wave tile 64 rows x 128 columns, workgroup = 4 waves = 256 rows.
template <bool B_ROWS, int MT>      // B_ROWS true: NT (B fragment from rows of W[N][K]); false: NN (B fragment from columns of W[K][N]); MT row tiles of 32 per wave
__global__ void __launch_bounds__(256) k_mm(TRows rows, const float* __restrict__ X, int ldx, int K, const float* __restrict__ W, int ldw,
        const float* __restrict__ bias, int N, float* __restrict__ Y, int ldy, int beta, int actA, const float* __restrict__ epi_pre,
        int ld_epi, TDrop dr, unsigned site) {
    const int R = nrows(rows);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * (128 * MT) + wave * (32 * MT);
    if (blockIdx.x * (128 * MT) >= R) return;
    const int n0 = blockIdx.y * 128;
    tf32x16 acc[MT][4];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // raw operand loads one k-step ahead of the MFMAs that use them (a fragment straight from global memory is a load -> wait -> use chain
    // per k-step otherwise): rows / columns clamped, out-of-range elements zeroed when the fragment is finished
    float ra[MT][8], rb[4][8];
    auto load_step = [&](int k0) {
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int row = m0 + 32 * a + r;
            const float* p = X + (size_t)(row < R ? row : 0) * ldx + k0 + 8 * h;
            const tf32x4 u = *reinterpret_cast<const tf32x4*>(p), v = *reinterpret_cast<const tf32x4*>(p + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { ra[a][q] = u[q]; ra[a][4 + q] = v[q]; }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int col = n0 + 32 * b + r;
            const int cc = col < N ? col : 0;
            if (B_ROWS) {
                const float* p = W + (size_t)cc * ldw + k0 + 8 * h;
                const tf32x4 u = *reinterpret_cast<const tf32x4*>(p), v = *reinterpret_cast<const tf32x4*>(p + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { rb[b][q] = u[q]; rb[b][4 + q] = v[q]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int kk = k0 + 8 * h + j; rb[b][j] = W[(size_t)(kk < K ? kk : 0) * ldw + cc]; }
            }
        }
    };
    load_step(0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        tu32x4 af[MT], bf[4];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int row = m0 + 32 * a + r;
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ra[a][q];
            if (actA) {
                float dm[8];
                drop8(dr, drop_key(dr, site), (unsigned)(((unsigned long long)row * K + k0 + 8 * h) >> 3), dm);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = gelu_fast(v[q]) * dm[q];
            }
            if (row >= R) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = 0.f;
            }
            af[a] = tpack8(v);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int col = n0 + 32 * b + r;
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = (col < N && (B_ROWS || k0 + 8 * h + q < K)) ? rb[b][q] : 0.f;
            bf[b] = tpack8(v);
        }
        if (k0 + 16 < K) load_step(k0 + 16);
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = tmfma(af[a], bf[b], acc[a][b]);
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int col = n0 + 32 * b + r;
        const bool colok = col < N;
        const int cc = colok ? col : 0;
        const float bv = (bias && colok) ? bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            // the loads of a tile (old Y for beta, the taped pre-activation) go out as one batch with clamped rows
            float yo[16], pr[16];
            if (beta) {
#pragma unroll
                for (int i = 0; i < 16; ++i) yo[i] = Y[(size_t)min(m0 + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h, R - 1) * ldy + cc];
            }
            if (epi_pre) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pr[i] = epi_pre[(size_t)min(m0 + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h, R - 1) * ld_epi + cc];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = m0 + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
                float v = acc[a][b][i] + bv;
                if (epi_pre) v *= gelu_d_fast(pr[i]) * drop_mul(dr, site, (unsigned long long)row * ld_epi + col);
                if (beta) v += yo[i];
                if (colok && row < R) Y[(size_t)row * ldy + col] = v;
            }
        }
    }
}
// ---- NT / NN for the dominant shape K = N = 128 (the per-edge Linears): the whole weight lives in REGISTERS as 32 B
// fragments (built once per workgroup through an LDS fragment image), a wave streams 32-row tiles of X past it:
// 16 coalesced 16-byte loads and 32 MFMAs per tile, no LDS traffic in the loop.
template <bool B_ROWS>
__global__ void __launch_bounds__(256, 1) k_mm128(TRows rows, const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
        const float* __restrict__ bias, float* __restrict__ Y, int ldy, int beta, int actA, const float* __restrict__ epi_pre,
        TDrop dr, unsigned site, const unsigned short* __restrict__ wimg) {
    __shared__ __attribute__((aligned(16))) unsigned short img[32 * 64 * 8];      // [ks][cb][lane][8] bf16
    const int R = nrows(rows);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    stage_wimage(img, wimg, W, ldw, B_ROWS, 0, tid);
    __syncthreads();
    // one wave per SIMD with the whole 512-register file: 128 registers of weight fragments + a full tile of X in flight
    tu32x4 bf[8][4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) bf[ks][cb] = reinterpret_cast<const tu32x4*>(img)[(ks * 4 + cb) * 64 + lane];
    float bv[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bv[cb] = bias ? bias[32 * cb + r] : 0.f;
    const int ntiles = (R + 31) / 32;
    const int tstride = gridDim.x * 4;
    tf32x4 raw[16];
    auto load_raw = [&](int t) {                              // this lane's row of tile t: 8 floats per k-step (rows clamped)
        const int row = min(32 * t + r, R - 1);
        const float* p = X + (size_t)row * ldx + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            raw[2 * ks] = *reinterpret_cast<const tf32x4*>(p + 16 * ks);
            raw[2 * ks + 1] = *reinterpret_cast<const tf32x4*>(p + 16 * ks + 4);
        }
    };
    int t = blockIdx.x * 4 + wave;
    if (t < ntiles) load_raw(t);
    for (; t < ntiles; t += tstride) {
        const int row = 32 * t + r;
        tu32x4 af[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            float v[8] = {raw[2 * ks][0], raw[2 * ks][1], raw[2 * ks][2], raw[2 * ks][3],
                          raw[2 * ks + 1][0], raw[2 * ks + 1][1], raw[2 * ks + 1][2], raw[2 * ks + 1][3]};
            if (actA) {
                float dm[8];
                drop8(dr, drop_key(dr, site), (unsigned)row * 16u + 2 * ks + h, dm);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gelu_fast(v[j]) * dm[j];
            }
            if (row >= R) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.f;
            }
            af[ks] = tpack8(v);
        }
        if (t + tstride < ntiles) load_raw(t + tstride);      // next tile's loads fly under this tile's MFMAs and stores
        tf32x16 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][i] = bv[cb];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = tmfma(af[ks], bf[ks][cb], acc[cb]);
        // epilogue: the loads of a column block (old Y for beta, the taped pre-activation for the GELU backward) go out as
        // one batch with clamped rows - a load -> use -> store chain per element would expose one memory round trip each
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int col = 32 * cb + r;
            float yo[16], pr[16];
            if (beta) {
#pragma unroll
                for (int i = 0; i < 16; ++i) yo[i] = Y[(size_t)min(32 * t + (i & 3) + 8 * (i >> 2) + 4 * h, R - 1) * ldy + col];
            }
            if (epi_pre) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pr[i] = epi_pre[(size_t)min(32 * t + (i & 3) + 8 * (i >> 2) + 4 * h, R - 1) * 128 + col];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int orow = 32 * t + (i & 3) + 8 * (i >> 2) + 4 * h;
                float v = acc[cb][i];
                if (epi_pre) v *= gelu_d_fast(pr[i]) * drop_mul(dr, site, (unsigned long long)orow * 128 + col);
                if (beta) v += yo[i];
                if (orow < R) Y[(size_t)orow * ldy + col] = v;
            }
        }
    }
}
static int mm128_grid(const TRows& rows) {
    int g = (rows.maxrows + 127) / 128;                      // 4 waves x one 32-row tile each
    const int cap = rn_num_cus();                            // one workgroup per CU (one wave per SIMD)
    return g > cap ? cap : (g < 1 ? 1 : g);
}

static bool mm_ok(const void* X, int ldx, int K, const void* W, int ldw, bool b_rows) {
    return K % 16 == 0 && ldx % 4 == 0 && ((uintptr_t)X & 15) == 0 && (!b_rows || (ldw % 4 == 0 && ((uintptr_t)W & 15) == 0));
}
// LDS-tiled general GEMM (defined below, next to the transposed-read helper): false when the shape is not covered
static bool launch_tmm(bool wt, const TRows& rows, const float* X, int ldx, int K, const float* W, int ldw, const float* bias, int N, float* Y,
                       int ldy, int beta, bool actA, const float* epi_pre, int ld_epi, const TDrop& dr, unsigned site, hipStream_t s,
                       const float* W_hi = nullptr, int k_split = 0);
// Y = [beta Y] + actA(X) . W^T + bias            (W [N][K] row-major: nn.Linear.weight as it is stored)
bool tm_gemm_nt(const TRows& rows, const float* X, int ldx, int K, const float* W, int ldw, const float* bias, int N, float* Y,
                int ldy, int beta, bool actA, const TDrop& dr, unsigned site, hipStream_t s) {
    if (!mm_ok(X, ldx, K, W, ldw, true)) return false;
    if (K == 128 && N == 128 && ldw % 2 == 0) {
        hipLaunchKernelGGL(k_mm128<true>, dim3(mm128_grid(rows)), dim3(256), 0, s, rows, X, ldx, W, ldw, bias, Y, ldy, beta,
                           actA ? 1 : 0, (const float*)nullptr, dr, site, wimg_lookup(W, ldw, true, 0));
        return true;
    }
    if (launch_tmm(false, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, actA, nullptr, 0, dr, site, s)) return true;
    // 64-row wave tiles reuse the B fragments twice; with few rows (node tensors) 32-row tiles fill more of the chip
    const bool small = (long long)((rows.maxrows + 255) / 256) * ((N + 127) / 128) < 2 * rn_num_cus();
    if (small) {
        dim3 grid((rows.maxrows + 127) / 128, (N + 127) / 128);
        hipLaunchKernelGGL((k_mm<true, 1>), grid, dim3(256), 0, s, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, actA ? 1 : 0,
                           (const float*)nullptr, 0, dr, site);
    } else {
        dim3 grid((rows.maxrows + 255) / 256, (N + 127) / 128);
        hipLaunchKernelGGL((k_mm<true, 2>), grid, dim3(256), 0, s, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, actA ? 1 : 0,
                           (const float*)nullptr, 0, dr, site);
    }
    return true;
}
// Y = [beta Y] + (X . W) [* gelu'(pre) * mask]    (W [K][N] row-major; backward dX = dY . W with W = nn.Linear.weight [out][in])
bool tm_gemm_nn(const TRows& rows, const float* X, int ldx, int K, const float* W, int ldw, const float* bias, int N, float* Y,
                int ldy, int beta, const float* epi_pre, int ld_epi, const TDrop& dr, unsigned site, hipStream_t s) {
    if (!mm_ok(X, ldx, K, W, ldw, false)) return false;
    if (K == 128 && N == 128 && (!epi_pre || ld_epi == 128)) {
        hipLaunchKernelGGL(k_mm128<false>, dim3(mm128_grid(rows)), dim3(256), 0, s, rows, X, ldx, W, ldw, bias, Y, ldy, beta, 0,
                           epi_pre, dr, site, wimg_lookup(W, ldw, false, 0));
        return true;
    }
    if (launch_tmm(true, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, false, epi_pre, ld_epi, dr, site, s)) return true;
    const bool small = (long long)((rows.maxrows + 255) / 256) * ((N + 127) / 128) < 2 * rn_num_cus();
    if (small) {
        dim3 grid((rows.maxrows + 127) / 128, (N + 127) / 128);
        hipLaunchKernelGGL((k_mm<false, 1>), grid, dim3(256), 0, s, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, 0, epi_pre, ld_epi, dr, site);
    } else {
        dim3 grid((rows.maxrows + 255) / 256, (N + 127) / 128);
        hipLaunchKernelGGL((k_mm<false, 2>), grid, dim3(256), 0, s, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, 0, epi_pre, ld_epi, dr, site);
    }
    return true;
}

// ---- TN: dW[n][kk] += sum_m A[m][n] * actB(B[m][kk]): contraction over ROWS.  Both operands are needed k-major (k = row
// index) while the tensors are row-major: 64-row tiles are staged row-major in LDS as bf16 (coalesced 16-byte global loads,
// 8-byte LDS writes) and read back TRANSPOSED by ds_read_b64_tr_b16 (a 4-row x 16-column block per 16-lane group, delivered
// column-major: exactly the 32x32x16 operand).  Row pitch 320 B makes both the writes and the transposed reads conflict-free.
// Workgroup = 4 waves on one 128 (n) x 128 (kk) output tile (wave = 64 x 64 quadrant); the row range is split over
// blockIdx.z, every split writes its own partial tile, reduce_parts adds them in order.
#define TN_PITCH 160                                          // bf16 elements per LDS row (128 data + 32 pad)
typedef __attribute__((ext_vector_type(4))) short ts16x4;
__device__ __forceinline__ tu32x4 tr_frag(const unsigned short* tile, int row0, int col0, int lane) {
    const int g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3, h = lane >> 5;
    const unsigned short* a = tile + (row0 + 8 * h + q) * TN_PITCH + col0 + 16 * g + 4 * p;
    const ts16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ts16x4*)a);
    const ts16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ts16x4*)(a + 4 * TN_PITCH));
    typedef __attribute__((ext_vector_type(2))) unsigned tu32x2;
    const tu32x2 l2 = __builtin_bit_cast(tu32x2, lo), h2 = __builtin_bit_cast(tu32x2, hi);
    return tu32x4{l2[0], l2[1], h2[0], h2[1]};
}
// ---- LDS-tiled NT / NN for the node-level shapes (K or N = 256 .. 512): a (64 MT) x 128 tile per workgroup (2 x 2 waves), K staged in
// steps of 64 with the next tile's global loads in flight.  X tile: f32 rows -> [GELU + dropout] -> bf16, k-contiguous in LDS.  W tile:
// NT (W [N][K]): k-contiguous rows like X, plain 16-byte fragment reads;  NN (W [K][N], the backward's dX = dY . W): the tile is stored as
// it is read, [k][n], and the B fragments come out of ds_read_b64_tr_b16 (k runs down the rows: the transposed read of k_mm_tn).
// Fragments straight from global memory (k_mm) re-read every operand once per wave: 768 B per MFMA against 256 B here.
#define TM_LD 72                                              // bf16 elements per k-contiguous LDS row (64 data + 8 pad)
template <bool WT, int MT>
__global__ void __launch_bounds__(256) k_tmm(TRows rows, const float* __restrict__ X, int ldx, int K, const float* __restrict__ W, int ldw,
        const float* __restrict__ bias, int N, float* __restrict__ Y, int ldy, int beta, int actA, const float* __restrict__ epi_pre,
        int ld_epi, TDrop dr, unsigned site, const float* __restrict__ W_hi, int k_split) {
    constexpr int BM = 64 * MT;
    __shared__ __attribute__((aligned(16))) unsigned short As[BM * TM_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[WT ? 64 * TN_PITCH : 128 * TM_LD];
    const int R = nrows(rows);
    const int row0 = blockIdx.x * BM;
    if (row0 >= R) return;
    const int col0 = blockIdx.y * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    tf32x16 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    // staging maps.  A: thread -> (row tid / (4 / MT) ..., 16 * MT floats);  B (NT): (n = tid >> 1, 32 floats);  B (NN): (k = tid >> 2, 32 floats of n)
    constexpr int AV = 4 * MT;                                // float4 loads of A per thread: BM * 64 / 256 / 4
    const int arow = MT == 2 ? tid >> 1 : tid >> 2, acol = MT == 2 ? (tid & 1) * 32 : (tid & 3) * 16;
    tf32x4 ar[AV], br[8];
    const unsigned key = drop_key(dr, site);
    auto load_tile = [&](int k0) {
        const int row = row0 + arow, kk = k0 + acol;
        const float* src = X + (size_t)(row < R ? row : R - 1) * ldx + (kk < K ? kk : 0);
#pragma unroll
        for (int v = 0; v < AV; ++v) ar[v] = *reinterpret_cast<const tf32x4*>(src + 4 * v);
        if (!WT) {
            const int n = col0 + (tid >> 1), kb = k0 + (tid & 1) * 32;
            const float* ws = W + (size_t)(n < N ? n : 0) * ldw + (kb < K ? kb : 0);
#pragma unroll
            for (int v = 0; v < 8; ++v) br[v] = *reinterpret_cast<const tf32x4*>(ws + 4 * v);
        } else {
            const int k = k0 + (tid >> 2), nb = col0 + (tid & 3) * 32;
            const int kc = k < K ? k : 0;                     // (NN: rows k >= k_split of W' come from a second block, W_hi)
            const float* ws = (kc < k_split ? W + (size_t)kc * ldw : W_hi + (size_t)(kc - k_split) * ldw) + (nb + 31 < N ? nb : 0);
#pragma unroll
            for (int v = 0; v < 8; ++v) br[v] = *reinterpret_cast<const tf32x4*>(ws + 4 * v);
        }
    };
    auto store_tile = [&](int k0) {
        typedef __attribute__((ext_vector_type(2))) unsigned tu32x2;
        {
            const int row = row0 + arow, kk = k0 + acol;
            const bool ok = row < R && kk < K;
#pragma unroll
            for (int v = 0; v < AV; ++v) {
                float x[4] = {ar[v][0], ar[v][1], ar[v][2], ar[v][3]};
                if (actA) {
                    float m0, m1, m2, m3;
                    const unsigned P = (unsigned)(((unsigned long long)row * K + kk + 4 * v) >> 1);
                    drop_pair(dr, key, P, m0, m1); drop_pair(dr, key, P + 1, m2, m3);
                    x[0] = gelu_fast(x[0]) * m0; x[1] = gelu_fast(x[1]) * m1; x[2] = gelu_fast(x[2]) * m2; x[3] = gelu_fast(x[3]) * m3;
                }
                const tu32x2 w = ok ? tu32x2{tpack2(x[0], x[1]), tpack2(x[2], x[3])} : tu32x2{0u, 0u};
                *reinterpret_cast<tu32x2*>(As + arow * TM_LD + acol + 4 * v) = w;
            }
        }
        if (!WT) {
            const int n = col0 + (tid >> 1), kb = k0 + (tid & 1) * 32;
            const bool ok = n < N && kb < K;
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const tu32x2 w = ok ? tu32x2{tpack2(br[v][0], br[v][1]), tpack2(br[v][2], br[v][3])} : tu32x2{0u, 0u};
                *reinterpret_cast<tu32x2*>(Bs + (tid >> 1) * TM_LD + (tid & 1) * 32 + 4 * v) = w;
            }
        } else {
            const int k = k0 + (tid >> 2), nb = col0 + (tid & 3) * 32;
            const bool ok = k < K && nb + 31 < N;
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                const tu32x2 w = ok ? tu32x2{tpack2(br[v][0], br[v][1]), tpack2(br[v][2], br[v][3])} : tu32x2{0u, 0u};
                *reinterpret_cast<tu32x2*>(Bs + (tid >> 2) * TN_PITCH + (tid & 3) * 32 + 4 * v) = w;
            }
        }
    };
    load_tile(0);
    for (int k0 = 0; k0 < K; k0 += 64) {
        __syncthreads();
        store_tile(k0);
        __syncthreads();
        if (k0 + 64 < K) load_tile(k0 + 64);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            tu32x4 af[MT], bf[2];
#pragma unroll
            for (int a = 0; a < MT; ++a) af[a] = *reinterpret_cast<const tu32x4*>(As + (32 * MT * wr + 32 * a + r) * TM_LD + 16 * ks + 8 * h);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (!WT) bf[b] = *reinterpret_cast<const tu32x4*>(Bs + (64 * wc + 32 * b + r) * TM_LD + 16 * ks + 8 * h);
                else bf[b] = tr_frag(Bs, 16 * ks, 64 * wc + 32 * b, lane);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = tmfma(af[a], bf[b], acc[a][b]);
        }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + 64 * wc + 32 * b + r;
        const bool colok = col < N;
        const int cc = colok ? col : 0;
        const float bv = (bias && colok) ? bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int rbase = row0 + 32 * MT * wr + 32 * a + 4 * h;
            float yo[16], pr[16];
            if (beta) {
#pragma unroll
                for (int i = 0; i < 16; ++i) yo[i] = Y[(size_t)min(rbase + (i & 3) + 8 * (i >> 2), R - 1) * ldy + cc];
            }
            if (epi_pre) {
#pragma unroll
                for (int i = 0; i < 16; ++i) pr[i] = epi_pre[(size_t)min(rbase + (i & 3) + 8 * (i >> 2), R - 1) * ld_epi + cc];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = rbase + (i & 3) + 8 * (i >> 2);
                float v = acc[a][b][i] + bv;
                if (epi_pre) v *= gelu_d_fast(pr[i]) * drop_mul(dr, site, (unsigned long long)row * ld_epi + col);
                if (beta) v += yo[i];
                if (colok && row < R) Y[(size_t)row * ldy + col] = v;
            }
        }
    }
}
static bool launch_tmm(bool wt, const TRows& rows, const float* X, int ldx, int K, const float* W, int ldw, const float* bias, int N, float* Y,
                       int ldy, int beta, bool actA, const float* epi_pre, int ld_epi, const TDrop& dr, unsigned site, hipStream_t s,
                       const float* W_hi, int k_split) {
    if (!W_hi) k_split = K;
    // 16-byte loads of 16 / 32 consecutive floats: K in whole half-tiles, aligned rows; NN: whole 32-column groups of W
    if (K % 32 || ldx % 4 || ldw % 4 || ((uintptr_t)X & 15) || ((uintptr_t)W & 15) || (wt && N % 32) || K < 64) return false;
    const long long big = (long long)((rows.maxrows + 127) / 128) * ((N + 127) / 128);
    if (big >= 2 * rn_num_cus()) {
        dim3 grid((rows.maxrows + 127) / 128, (N + 127) / 128);
        if (wt) hipLaunchKernelGGL((k_tmm<true, 2>), grid, dim3(256), 0, s, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, actA ? 1 : 0, epi_pre, ld_epi, dr, site, W_hi, k_split);
        else hipLaunchKernelGGL((k_tmm<false, 2>), grid, dim3(256), 0, s, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, actA ? 1 : 0, epi_pre, ld_epi, dr, site, W_hi, k_split);
    } else {
        dim3 grid((rows.maxrows + 63) / 64, (N + 127) / 128);
        if (wt) hipLaunchKernelGGL((k_tmm<true, 1>), grid, dim3(256), 0, s, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, actA ? 1 : 0, epi_pre, ld_epi, dr, site, W_hi, k_split);
        else hipLaunchKernelGGL((k_tmm<false, 1>), grid, dim3(256), 0, s, rows, X, ldx, K, W, ldw, bias, N, Y, ldy, beta, actA ? 1 : 0, epi_pre, ld_epi, dr, site, W_hi, k_split);
    }
    return true;
}

__global__ void __launch_bounds__(256) k_mm_tn(TRows rows, const float* __restrict__ A, int lda, int M, const float* __restrict__ B,
        int ldb, int K, float* __restrict__ part, size_t pstride, int rows_per_split, int actB, TDrop dr, unsigned site, float* __restrict__ cs_part) {
    __shared__ __attribute__((aligned(16))) unsigned short tA[64 * TN_PITCH], tB[64 * TN_PITCH];
    __shared__ float cs_red[8][128];
    const int R = nrows(rows);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.x * 128, kk0 = blockIdx.y * 128;
    const int p_begin = blockIdx.z * rows_per_split, p_end = min(R, p_begin + rows_per_split);
    // staging map: thread -> 8 (row, 4-column group) cells of the 64 x 128 tile; cg fastest: coalesced 512-B rows
    const int cg = tid & 31, rg = tid >> 5;                   // columns 4cg .. 4cg+3; rows rg, rg+8, ...
    tf32x4 ra[8], rb[8];
    tf32x4 csum = {0.f, 0.f, 0.f, 0.f};                       // column sums of A over this thread's rows (bias gradient, optional)
    auto load_tile = [&](int m0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + rg + 8 * i;
            const bool ok = m < p_end;
            const int mc = ok ? m : (R > 0 ? R - 1 : 0);
            const int ca = n0 + 4 * cg, cb = kk0 + 4 * cg;
            tf32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (ca + 3 < M) va = *reinterpret_cast<const tf32x4*>(A + (size_t)mc * lda + ca);
            else for (int c = 0; c < 4; ++c) if (ca + c < M) va[c] = A[(size_t)mc * lda + ca + c];
            if (cb + 3 < K) vb = *reinterpret_cast<const tf32x4*>(B + (size_t)mc * ldb + cb);
            else for (int c = 0; c < 4; ++c) if (cb + c < K) vb[c] = B[(size_t)mc * ldb + cb + c];
            if (actB) {
#pragma unroll
                for (int c = 0; c < 4; ++c) vb[c] = gelu_fast(vb[c]) * drop_mul(dr, site, (unsigned long long)m * K + cb + c);
            }
            if (!ok) { va = tf32x4{0.f, 0.f, 0.f, 0.f}; vb = va; }
            ra[i] = va; rb[i] = vb;
        }
    };
    auto store_tile = [&]() {
        typedef __attribute__((ext_vector_type(2))) unsigned tu32x2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = rg + 8 * i;
            *reinterpret_cast<tu32x2*>(tA + row * TN_PITCH + 4 * cg) = tu32x2{tpack2(ra[i][0], ra[i][1]), tpack2(ra[i][2], ra[i][3])};
            *reinterpret_cast<tu32x2*>(tB + row * TN_PITCH + 4 * cg) = tu32x2{tpack2(rb[i][0], rb[i][1]), tpack2(rb[i][2], rb[i][3])};
        }
    };
    tf32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    if (p_begin < p_end) load_tile(p_begin);
    for (int m0 = p_begin; m0 < p_end; m0 += 64) {
        __syncthreads();                                      // the previous tile's fragment reads are done
        store_tile();
        if (cs_part) {
#pragma unroll
            for (int i = 0; i < 8; ++i) csum += ra[i];
        }
        __syncthreads();
        if (m0 + 64 < p_end) load_tile(m0 + 64);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            tu32x4 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = tr_frag(tA, 16 * ks, 64 * wr + 32 * a, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = tr_frag(tB, 16 * ks, 64 * wc + 32 * b, lane);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = tmfma(af[a], bf[b], acc[a][b]);
        }
    }
    if (cs_part && blockIdx.y == 0) {                         // rows rg, rg+8, ... were summed per thread: fold the 8 row groups in order
        *reinterpret_cast<tf32x4*>(&cs_red[rg][4 * cg]) = csum;
        __syncthreads();
        if (tid < 128 && n0 + tid < M) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) t += cs_red[g][tid];
            cs_part[(size_t)blockIdx.z * pstride + n0 + tid] = t;
        }
    }
    float* dst = part + (size_t)blockIdx.z * pstride;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = kk0 + 64 * wc + 32 * b + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = n0 + 64 * wr + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (row < M && col < K) dst[(size_t)row * K + col] = acc[a][b][i];
            }
        }
}
void tm_gemm_tn(const TRows& rows, const float* A, int lda, int M, const float* B, int ldb, int K, float* dW, int ldw,
                const TScratch& sc_in, bool actB, const TDrop& dr, unsigned site, float* dbias, hipStream_t s) {
    const TScratch sc = red_acquire(sc_in);
    const size_t mk = (size_t)M * K;
    const int tiles = ((M + 127) / 128) * ((K + 127) / 128);
    long long cap = (long long)((sc.floats - (size_t)520 * M) / mk) - 16;   // behind the partials: 16 * mk floats for the two-level reduction, 520 * M for the column sums
    int splits = (rows.maxrows + 255) / 256;                  // >= 256 rows (4 tiles) per workgroup
    const int want = (2 * rn_num_cus() + tiles - 1) / tiles;  // enough workgroups for the chip
    if (splits > want) splits = want;
    if (splits > cap) splits = (int)cap;
    if (splits > 500) splits = 500;
    if (splits < 1) splits = 1;
    const int rps = ((rows.maxrows + splits - 1) / splits + 63) / 64 * 64;
    if ((lda % 4) || (ldb % 4) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || actB && cap < 1) {     // 16-byte row loads need aligned rows
        t_gemm_tn(rows, A, lda, M, B, ldb, K, dW, ldw, sc, s);
        if (dbias) t_colsum(rows, A, lda, M, dbias, sc, s);
        return;
    }
    // partial of split z: [M*K tile][M column sums (with dbias)]: one ordered reduction adds both into dW / dbias
    const size_t pstride = mk + (dbias ? (size_t)M : 0);
    float* tmp = sc.p + (size_t)splits * pstride;             // [16][pstride] second-level buffer
    dim3 grid((M + 127) / 128, (K + 127) / 128, splits);
    hipLaunchKernelGGL(k_mm_tn, grid, dim3(256), 0, s, rows, A, lda, M, B, ldb, K, sc.p, pstride, rps, actB ? 1 : 0, dr, site,
                       dbias ? sc.p + mk : (float*)nullptr);
    reduce_parts(sc.p, splits, pstride, (int)pstride, K, dW, ldw, s, tmp, (int)mk, dbias);
}

// The node-side weight gradients of a factored first Linear in ONE product: [dWa ; dWb] = [dP | dQ]^T h (256 x 128), db1 = colsum(dP).
// gw0 = the [128][384] gradient of the Linear's weight ([Wa | Wb | Wc] blocks): rows 0..127 of the product go to columns 0..127, rows
// 128..255 to columns 128..255 of the same 128 output rows (the reduction's wrap).
void tm_gemm_tn_pq(const TRows& rows, const float* dpq, const float* h, float* gw0, float* db1, const TScratch& sc_in, hipStream_t s) {
    const TScratch sc = red_acquire(sc_in);
    const int M = 256, K = 128;
    const size_t mk = (size_t)M * K, pstride = mk + M;
    long long cap = (long long)(sc.floats / pstride) - 16;
    int splits = (rows.maxrows + 255) / 256;
    const int want = rn_num_cus();                            // two 128-row output tiles per split
    if (splits > want) splits = want;
    if (splits > cap) splits = (int)cap;
    if (splits < 1) splits = 1;
    const int rps = ((rows.maxrows + splits - 1) / splits + 63) / 64 * 64;
    float* tmp = sc.p + (size_t)splits * pstride;
    hipLaunchKernelGGL(k_mm_tn, dim3(2, 1, splits), dim3(256), 0, s, rows, dpq, 256, M, h, 128, K, sc.p, pstride, rps, 0, TDrop{0ull, 0u, 1.f}, 0u, sc.p + mk);
    reduce_parts(sc.p, splits, pstride, (int)pstride, K, gw0, 3 * 128, s, tmp, (int)mk, db1, K, 128, 128, 128);
}
// dh += dP . Wa + dQ . Wb as one K = 256 product (w0 = [128][384] weight: Wa = columns 0..127, Wb = columns 128..255)
bool tm_gemm_nn_pq(const TRows& rows, const float* dpq, const float* w0, float* dh, hipStream_t s) {
    return launch_tmm(true, rows, dpq, 256, 256, w0, 3 * 128, nullptr, 128, dh, 128, 1, false, nullptr, 0, TDrop{0ull, 0u, 1.f}, 0u, s, w0 + 128, 128);
}

// ------------------------------------------------------------------------------------------
// Adam with L2 weight decay, torch.optim.Adam semantics (foreach / fused CUDA form): bias-corrected moments,
// denom = sqrt(v) / sqrt(1 - beta2^t) + eps.
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                       float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float pi = p[i];
        if (wd != 0.f) gi = fmaf(wd, pi, gi);
        const float mi = fmaf(b1, m[i], (1.f - b1) * gi);       // lerp form of exp_avg
        const float vi = fmaf(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}
void t_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd,
                 int step, hipStream_t s) {
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    int grid = (int)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, (float)bc1, (float)sqrt(bc2));
}

// ==========================================================================================
// bf16-STORAGE edge kernels of the bf16-mixed trainer (declared at the end of kernels_train.h).
__device__ __forceinline__ float tbf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float tbf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float tbf(tb16 v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ tb16 to_tb(float v) { return (tb16)(tpack2(v, 0.f) & 0xffffu); }
__device__ __forceinline__ void unpack8(const tu32x4& u, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = tbf_lo(u[i]); v[2 * i + 1] = tbf_hi(u[i]); }
}

struct EmmArgs {
    TRows rows;
    const void* X; int ldx;
    const float* W; int ldw; const float* bias;
    tb16* Y; int beta; int actA; const tb16* epi_pre;
    EFuse f; int has_pq, has_res;
    TDrop dr; unsigned site;
    const unsigned short* wimg;      // prebuilt fragment image of W (WImageCache) or null
    int w_yoff; tb16* Y2; const unsigned short* wimg2;     // blockIdx.y = 1: second weight block (W + w_yoff, no bias) -> Y2   (P and Q tables in one launch)
    int kvalid;                      // columns of X / rows of W' beyond it do not exist (treated as zero weights)
};
// The [E][128] x [128][128] GEMM of the per-edge Linears with the output TRANSPOSED in the accumulators: D = W' . X^T, i.e. the weight
// is the A operand (its rows permuted so that a lane's 16 accumulator registers of a 32-channel block are two runs of 8 CONSECUTIVE
// channels) and the 32 edge rows of a tile are the B operand (lane (r, h) = 8 consecutive k of row r: one 16-byte bf16 load).  Every
// lane then owns 64 channels of ITS edge row: the epilogue's loads / stores (old Y, taped pre-activation, P / Q rows, residual input,
// outputs) are all 16-byte row-contiguous accesses, and row-wise data (neighbour index, validity) is per lane.
// The kernel is a stream (256 B in, 256 - 768 B out / side inputs per row) with ~40 VALU operations per element when an activation
// or its derivative is fused: two waves per SIMD (weight fragments from a 32 KiB LDS image instead of 128 registers) so that one
// wave's loads and VALU run under the other's MFMAs, the next tile's X and THIS tile's epilogue operands are requested before the MFMA
// block, and every load of the epilogue sits in registers before the first store (stores would otherwise fence the later loads: the
// compiler cannot prove that Y does not alias them).
// EP: 0 none, 1 + P[row / k] + Q[nbr[row]], 2 * gelu'(pre) * mask, 3 + old Y;  ACT: X' = drop(gelu(X));  RES: second output res_in + drop(gelu(v))
template <bool B_ROWS, typename TX, int EP, bool ACT, bool RES>
__global__ void __launch_bounds__(256, 2) k_emm128(EmmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short img[32 * 64 * 8];      // [ks][cb][lane][8] bf16 A fragments
    __shared__ __attribute__((aligned(16))) float lds_bias[128];
    const int R = nrows(a.rows);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const bool second = blockIdx.y != 0;
    tb16* __restrict__ Yp = second ? a.Y2 : a.Y;
    stage_wimage(img, second ? a.wimg2 : a.wimg, second ? a.W + a.w_yoff : a.W, a.ldw, B_ROWS, 1, tid, a.kvalid);
    if (tid < 128) lds_bias[tid] = (a.bias && !second) ? a.bias[tid] : 0.f;
    __syncthreads();
    const tu32x4* wimg = reinterpret_cast<const tu32x4*>(img) + lane;
    const int ntiles = (R + 31) / 32;
    const int tstride = gridDim.x * 4;
    const unsigned key1 = drop_key(a.dr, a.site), key2 = drop_key(a.dr, a.f.site2);
    constexpr bool XB = sizeof(TX) == 2;
    constexpr bool NEED_J = EP == 1 || RES;
    tu32x4 rawb[XB ? 8 : 1];           // bf16 X: the fragments themselves
    tf32x4 rawf[XB ? 1 : 16];          // f32 X
    int jn = -1;                       // neighbour of the prefetched tile's row
    const TX* __restrict__ X = reinterpret_cast<const TX*>(a.X);
    const tu32x4 z4 = {0u, 0u, 0u, 0u};
    auto load_raw = [&](int t) {
        const int row = min(32 * t + r, R - 1);
        const TX* p = X + (size_t)row * a.ldx + 8 * h;
        if constexpr (XB) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) rawb[ks] = *reinterpret_cast<const tu32x4*>(p + 16 * ks);
        } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                rawf[2 * ks] = *reinterpret_cast<const tf32x4*>(p + 16 * ks);
                rawf[2 * ks + 1] = *reinterpret_cast<const tf32x4*>(p + 16 * ks + 4);
            }
        }
        if constexpr (NEED_J) jn = a.f.nbr[row];
    };
    int t = blockIdx.x * 4 + wave;
    if (t < ntiles) load_raw(t);
    for (; t < ntiles; t += tstride) {
        const int row = 32 * t + r;
        const bool rok = row < R;
        const int rowc = rok ? row : R - 1;
        const int j = jn;
        // ---- epilogue operands of this tile, group u = channels 16u + 8h .. +7 of this lane's row
        tu32x4 e0[EP ? 8 : 1], e1[(EP == 1 || RES) ? 8 : 1];      // e0: P row | taped pre-activation | old Y;  e1: Q row | residual input
        {
            const size_t ro = (size_t)rowc * 128 + 8 * h;
            if constexpr (EP == 1) {
                const tb16* prow = a.f.P + (size_t)(rowc / a.f.k) * 128 + 8 * h;
                const tb16* qrow = a.f.Q + (size_t)(j < 0 ? a.f.zero_row : (j > a.f.zero_row ? a.f.zero_row : j)) * 128 + 8 * h;
#pragma unroll
                for (int u = 0; u < 8; ++u) { e0[u] = *reinterpret_cast<const tu32x4*>(prow + 16 * u); e1[u] = *reinterpret_cast<const tu32x4*>(qrow + 16 * u); }
            } else if constexpr (EP == 2) {
#pragma unroll
                for (int u = 0; u < 8; ++u) e0[u] = *reinterpret_cast<const tu32x4*>(a.epi_pre + ro + 16 * u);
            } else if constexpr (EP == 3) {
#pragma unroll
                for (int u = 0; u < 8; ++u) e0[u] = *reinterpret_cast<const tu32x4*>(Yp + ro + 16 * u);
            }
            if constexpr (RES && EP != 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) e1[u] = *reinterpret_cast<const tu32x4*>(a.f.res_in + ro + 16 * u);
            }
        }
        tu32x4 xf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if constexpr (XB && !ACT) { xf[ks] = rawb[ks]; }
            else {
                float v[8];
                if constexpr (XB) unpack8(rawb[ks], v);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q] = rawf[2 * ks][q]; v[4 + q] = rawf[2 * ks + 1][q]; }
                }
                if constexpr (ACT) {
                    float dm[8];
                    drop8(a.dr, key1, (unsigned)row * 16u + 2 * ks + h, dm);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = gelu_fast(v[q]) * dm[q];
                }
                xf[ks] = tpack8(v);
            }
            if (!rok) xf[ks] = z4;
        }
        if (t + tstride < ntiles) load_raw(t + tstride);
        tf32x16 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
        // weight fragments from LDS, one k-step ahead of the MFMAs that use them (pinned: hoisting all 32 reads costs 128 registers)
        tu32x4 wa[4], wb[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) wa[cb] = wimg[cb * 64];
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) wb[cb] = wimg[((ks + 1) * 4 + cb) * 64];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = tmfma(wa[cb], xf[ks], acc[cb]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < 8) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) wa[cb] = wimg[((ks + 2) * 4 + cb) * 64];
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = tmfma(wb[cb], xf[ks + 1], acc[cb]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue in two halves of 64 channels: compute, then store (all loads are already in registers)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            tu32x4 yo[4], ro2[RES ? 4 : 1];
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {
                const int u = 4 * half + uu, cb = u >> 1, g = u & 1, c = 16 * u + 8 * h;
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = acc[cb][8 * g + q];
                const tf32x4 b0 = *reinterpret_cast<const tf32x4*>(lds_bias + c), b1 = *reinterpret_cast<const tf32x4*>(lds_bias + c + 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] += b0[q]; v[4 + q] += b1[q]; }
                if constexpr (EP == 1) {
                    float pv[8], qv[8];
                    unpack8(e0[u], pv); unpack8(e1[u], qv);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += pv[q] + qv[q];
                } else if constexpr (EP == 2) {
                    float pr[8], dm[8];
                    unpack8(e0[u], pr);
                    drop8(a.dr, key1, (unsigned)row * 16u + (c >> 3), dm);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] *= gelu_d_fast(pr[q]) * dm[q];
                } else if constexpr (EP == 3) {
                    float old[8];
                    unpack8(e0[u], old);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += old[q];
                }
                yo[uu] = tpack8(v);
                if constexpr (RES) {
                    float ei[8];
                    if constexpr (EP == 1) unpack8(*reinterpret_cast<const tu32x4*>(a.f.res_in + (size_t)rowc * 128 + c), ei);   // (depth-1 MLPs only)
                    else unpack8(e1[u], ei);
                    if (j >= 0) {
                        float dm[8];
                        drop8(a.dr, key2, (unsigned)row * 16u + (c >> 3), dm);
#pragma unroll
                        for (int q = 0; q < 8; ++q) ei[q] += gelu_fast(v[q]) * dm[q];
                    }
                    ro2[uu] = tpack8(ei);
                }
            }
            if (rok) {
                tb16* yrow = Yp + (size_t)row * 128 + 64 * half + 8 * h;
#pragma unroll
                for (int uu = 0; uu < 4; ++uu) *reinterpret_cast<tu32x4*>(yrow + 16 * uu) = yo[uu];
                if constexpr (RES) {
                    tb16* rrow = a.f.res_out + (size_t)row * 128 + 64 * half + 8 * h;
#pragma unroll
                    for (int uu = 0; uu < 4; ++uu) *reinterpret_cast<tu32x4*>(rrow + 16 * uu) = ro2[uu];
                }
            }
        }
    }
}
bool te_gemm(const TRows& rows, const void* X, bool x_bf16, int ldx, const float* W, int ldw, bool w_rows, const float* bias, tb16* Y,
             int beta, bool actA, const tb16* epi_pre, const EFuse* fuse, const TDrop& dr, unsigned site, hipStream_t s, int kvalid) {
    EmmArgs a;
    a.rows = rows; a.X = X; a.ldx = ldx; a.W = W; a.ldw = ldw; a.bias = bias; a.Y = Y; a.beta = beta; a.actA = actA ? 1 : 0;
    a.epi_pre = epi_pre; a.dr = dr; a.site = site;
    if (fuse) a.f = *fuse; else a.f = EFuse{nullptr, nullptr, nullptr, 1, 0, nullptr, nullptr, 0u};
    a.has_pq = a.f.P != nullptr; a.has_res = a.f.res_out != nullptr;
    a.kvalid = kvalid;
    a.wimg = wimg_lookup(W, ldw, w_rows, 1, kvalid);
    a.w_yoff = 0; a.Y2 = nullptr; a.wimg2 = nullptr;
    int g = (rows.maxrows + 127) / 128;                      // 4 waves x one 32-row tile each
    const int cap = 2 * rn_num_cus();                        // two workgroups per CU (two waves per SIMD)
    const dim3 grid(g > cap ? cap : (g < 1 ? 1 : g));
    const int ep = a.has_pq ? 1 : (epi_pre ? 2 : (beta ? 3 : 0));
#define EMM_GO(BR, TXT, EPV, ACTV, RESV) hipLaunchKernelGGL((k_emm128<BR, TXT, EPV, ACTV, RESV>), grid, dim3(256), 0, s, a)
    if (!x_bf16 && w_rows && ep == 0 && !actA && !a.has_res) EMM_GO(true, float, 0, false, false);                 // node rows -> bf16 P / Q tables
    else if (x_bf16 && w_rows && ep == 0 && !actA && !a.has_res) EMM_GO(true, tb16, 0, false, false);             // edge embedding: raw features -> pe1
    else if (x_bf16 && w_rows && ep == 1 && !actA && !a.has_res) EMM_GO(true, tb16, 1, false, false);             // first Linear + P + Q
    else if (x_bf16 && w_rows && ep == 1 && !actA && a.has_res) EMM_GO(true, tb16, 1, false, true);               // ... of a depth-1 edge update
    else if (x_bf16 && w_rows && ep == 0 && actA && !a.has_res) EMM_GO(true, tb16, 0, true, false);               // second Linear (message)
    else if (x_bf16 && w_rows && ep == 0 && actA && a.has_res) EMM_GO(true, tb16, 0, true, true);                 // second Linear + edge update
    else if (x_bf16 && !w_rows && ep == 2 && !actA && !a.has_res) EMM_GO(false, tb16, 2, false, false);           // d pre1 = (d pre2 . W2) gelu' mask
    else if (x_bf16 && !w_rows && ep == 3 && !actA && !a.has_res) EMM_GO(false, tb16, 3, false, false);           // dE += d pre1 . Wc
    else return false;                                   // combination not instantiated: nothing was launched
#undef EMM_GO
    return true;
}

// P = h Wa^T + b1 and Q = h Wb^T (bf16 tables [rows][128]) in one launch: w0 = [128][384] weight, Wa = columns 0..127, Wb = columns 128..255
void te_gemm_pq(const TRows& rows, const float* h, const float* w0, const float* b1, tb16* Pt, tb16* Qt, hipStream_t s) {
    EmmArgs a;
    a.rows = rows; a.X = h; a.ldx = 128; a.W = w0; a.ldw = 3 * 128; a.bias = b1; a.Y = Pt; a.beta = 0; a.actA = 0; a.epi_pre = nullptr;
    a.dr = TDrop{0ull, 0u, 1.f}; a.site = 0u;
    a.f = EFuse{nullptr, nullptr, nullptr, 1, 0, nullptr, nullptr, 0u};
    a.has_pq = 0; a.has_res = 0; a.kvalid = 128;
    a.wimg = wimg_lookup(w0, 3 * 128, true, 1); a.w_yoff = 128; a.Y2 = Qt; a.wimg2 = wimg_lookup(w0 + 128, 3 * 128, true, 1);
    int g = (rows.maxrows + 127) / 128;
    const int cap = rn_num_cus();
    hipLaunchKernelGGL((k_emm128<true, float, 0, false, false>), dim3(g > cap ? cap : (g < 1 ? 1 : g), 2), dim3(256), 0, s, a);
}

// ---- the forward of a depth-2 per-edge MLP in ONE kernel: pre1 = e . Wc^T + P[i] + Q[j] ; pre2 = drop(gelu(pre1)) . W2^T + b2 ;
// [e_out = e + valid * drop(gelu(pre2))].  With the accumulators transposed (lane = row, a lane's 16 registers of a channel block = two runs
// of 8 consecutive channels) the activated first-Linear output IS the B-operand fragment of the second Linear (k-step 2 cb + g <-> run g of
// block cb): the hidden activation never leaves the registers, pre1 / pre2 are written once as the tape, and the e tile that was the
// operand of the first Linear is still in registers when the edge update needs it.  Against the two-kernel form this saves the read of pre1
// and (edge update) the second read of e.  Both weight images live in LDS (64 KiB, two workgroups per CU).
__device__ __forceinline__ void gelu_both_fast(float x, float& g, float& d);
#define TE_DROPPED (-1.0e4f)         // taped in place of a dropped pre-activation: gelu_fast, gelu_d_fast and gelu_both_fast give exactly (-)0 there
struct Emm2Args {
    TRows rows;
    const tb16* X;                   // e [R][128]
    const float* W1; int ldw1;       // Wc block of the first Linear ([128][ldw1], nn.Linear layout)
    const float* W2; int ldw2; const float* bias2;
    const unsigned short* wimg1; const unsigned short* wimg2;
    tb16* pre1; tb16* pre2;
    EFuse f;                         // P, Q, nbr, k, zero_row ; res_out (optional), site2
    TDrop dr; unsigned site;         // dropout site of the hidden activation
    int g2tape;                      // RES: pre2 receives gelu'(pre2) * mask(site2) - all the backward needs of it - instead of pre2
    int p1mask;                      // pre1 is taped with its DROPPED elements replaced by TE_DROPPED (gelu = gelu' = 0 there): the backward needs no hash
};
template <bool RES, bool TAPE1>       // TAPE1: pre1 is written (training tape); inference callers keep only pre2
// tape stores (written once, read a whole backward later): TE_EXP_NT_TAPE builds them as non-temporal stores - measured 30.3 ms per step at the C2
// batch against 23.6 (the 16-byte-per-lane pieces of a row no longer merge in L2): kept as the experiment's switch only
#ifdef TE_EXP_NT_TAPE
#define TE_TAPE_STORE(ptr, val) __builtin_nontemporal_store((val), reinterpret_cast<tu32x4*>(ptr))
#else
#define TE_TAPE_STORE(ptr, val) (*reinterpret_cast<tu32x4*>(ptr) = (val))
#endif
__global__ void __launch_bounds__(256, 2) k_emm_fwd2(Emm2Args a) {
    __shared__ __attribute__((aligned(16))) unsigned short img1[32 * 64 * 8], img2[32 * 64 * 8];
    __shared__ __attribute__((aligned(16))) float lds_bias[128];
    const int R = nrows(a.rows);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    stage_wimage(img1, a.wimg1, a.W1, a.ldw1, true, 1, tid);
    stage_wimage(img2, a.wimg2, a.W2, a.ldw2, true, 1, tid);
    if (tid < 128) lds_bias[tid] = a.bias2 ? a.bias2[tid] : 0.f;
    __syncthreads();
    const tu32x4* w1 = reinterpret_cast<const tu32x4*>(img1) + lane;
    const tu32x4* w2 = reinterpret_cast<const tu32x4*>(img2) + lane;
    const int ntiles = (R + 31) / 32;
    const int tstride = gridDim.x * 4;
    const unsigned key1 = drop_key(a.dr, a.site), key2 = drop_key(a.dr, a.f.site2);
    tu32x4 raw[8];
    int jn = -1;
    auto load_raw = [&](int t) {
        const int row = min(32 * t + r, R - 1);
        const tb16* p = a.X + (size_t)row * 128 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) raw[ks] = *reinterpret_cast<const tu32x4*>(p + 16 * ks);
        jn = a.f.nbr[row];
    };
    auto gemm = [&](const tu32x4* wimg, const tu32x4 (&xf)[8], tf32x16 (&acc)[4]) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
        tu32x4 wa[4], wb[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) wa[cb] = wimg[cb * 64];
#pragma unroll
        for (int ks = 0; ks < 8; ks += 2) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) wb[cb] = wimg[((ks + 1) * 4 + cb) * 64];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = tmfma(wa[cb], xf[ks], acc[cb]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < 8) {
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) wa[cb] = wimg[((ks + 2) * 4 + cb) * 64];
            }
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = tmfma(wb[cb], xf[ks + 1], acc[cb]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int t = blockIdx.x * 4 + wave;
    if (t < ntiles) load_raw(t);
    for (; t < ntiles; t += tstride) {
        const int row = 32 * t + r;
        const bool rok = row < R;
        const int rowc = rok ? row : R - 1;
        const int j = jn;
        tu32x4 xe[8];                                        // this row of e: operand of the first Linear, residual input of the edge update
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xe[ks] = rok ? raw[ks] : tu32x4{0u, 0u, 0u, 0u};
        tu32x4 pp[8], qq[8];
        {
            const tb16* prow = a.f.P + (size_t)(rowc / a.f.k) * 128 + 8 * h;
            const tb16* qrow = a.f.Q + (size_t)(j < 0 ? a.f.zero_row : (j > a.f.zero_row ? a.f.zero_row : j)) * 128 + 8 * h;
#pragma unroll
            for (int u = 0; u < 8; ++u) { pp[u] = *reinterpret_cast<const tu32x4*>(prow + 16 * u); qq[u] = *reinterpret_cast<const tu32x4*>(qrow + 16 * u); }
        }
        if (t + tstride < ntiles) load_raw(t + tstride);
        tf32x16 acc[4];
        gemm(w1, xe, acc);
        // epilogue 1: pre1 = acc + P + Q (tape, bf16) ; hidden = drop(gelu(pre1 as stored)) -> operand fragments of the second Linear
        tu32x4 xh[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cb = u >> 1, g = u & 1;
            float v[8], pv[8], qv[8], dm[8];
            unpack8(pp[u], pv); unpack8(qq[u], qv);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = acc[cb][8 * g + q] + pv[q] + qv[q];
            const tu32x4 y = tpack8(v);
            drop8(a.dr, key1, (unsigned)row * 16u + 2 * u + h, dm);
            if (TAPE1 && rok) {
                if (a.p1mask) {                  // (uniform)
                    float vm[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) vm[q] = dm[q] != 0.f ? v[q] : TE_DROPPED;
                    TE_TAPE_STORE(a.pre1 + (size_t)row * 128 + 16 * u + 8 * h, tpack8(vm));
                } else {
                    TE_TAPE_STORE(a.pre1 + (size_t)row * 128 + 16 * u + 8 * h, y);
                }
            }
            unpack8(y, v);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = gelu_fast(v[q]) * dm[q];
            xh[u] = rok ? tpack8(v) : tu32x4{0u, 0u, 0u, 0u};
        }
        gemm(w2, xh, acc);
        // epilogue 2: pre2 (tape) [+ edge update from the e row still in registers]
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int cb = u >> 1, g = u & 1, c = 16 * u + 8 * h;
            float v[8];
            const tf32x4 b0 = *reinterpret_cast<const tf32x4*>(lds_bias + c), b1 = *reinterpret_cast<const tf32x4*>(lds_bias + c + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = acc[cb][8 * g + q] + b0[q]; v[4 + q] = acc[cb][8 * g + 4 + q] + b1[q]; }
            if constexpr (RES) {
                float ei[8], dm[8];
                unpack8(xe[u], ei);
                drop8(a.dr, key2, (unsigned)row * 16u + 2 * u + h, dm);
                if (a.g2tape) {                  // (uniform) the tape gets gelu' * mask, evaluated on the unrounded pre-activation, sharing the sigmoid with the update
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float g, d;
                        gelu_both_fast(v[q], g, d);
                        if (j >= 0) ei[q] += g * dm[q];
                        v[q] = d * dm[q];
                    }
                } else if (j >= 0) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) ei[q] += gelu_fast(v[q]) * dm[q];
                }
                if (rok) TE_TAPE_STORE(a.pre2 + (size_t)row * 128 + c, tpack8(v));
                if (rok) *reinterpret_cast<tu32x4*>(a.f.res_out + (size_t)row * 128 + c) = tpack8(ei);
            } else {
                if (rok) TE_TAPE_STORE(a.pre2 + (size_t)row * 128 + c, tpack8(v));
            }
        }
    }
}
// pre1 = X . W1^T + P[row / k] + Q[nbr[row]] ; pre2 = drop(gelu(pre1), site) . W2^T + bias2 ; [res_out = X + valid drop(gelu(pre2), site2)]
void te_mlp2_fwd(const TRows& rows, const tb16* X, const float* W1, int ldw1, const float* W2, int ldw2, const float* bias2, tb16* pre1,
                 tb16* pre2, const EFuse& f, const TDrop& dr, unsigned site, hipStream_t s, bool g2tape) {
    Emm2Args a;
    a.g2tape = (g2tape && f.res_out) ? 1 : 0;
    a.p1mask = (g2tape && pre1) ? 1 : 0;
    a.rows = rows; a.X = X; a.W1 = W1; a.ldw1 = ldw1; a.W2 = W2; a.ldw2 = ldw2; a.bias2 = bias2; a.pre1 = pre1; a.pre2 = pre2; a.f = f;
    a.dr = dr; a.site = site;
    a.wimg1 = wimg_lookup(W1, ldw1, true, 1); a.wimg2 = wimg_lookup(W2, ldw2, true, 1);
    int g = (rows.maxrows + 127) / 128;
    const int cap = 2 * rn_num_cus();
    const dim3 grid(g > cap ? cap : (g < 1 ? 1 : g));
    if (f.res_out) hipLaunchKernelGGL((k_emm_fwd2<true, true>), grid, dim3(256), 0, s, a);
    else if (pre1) hipLaunchKernelGGL((k_emm_fwd2<false, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_emm_fwd2<false, false>), grid, dim3(256), 0, s, a);
}

// ---- TN with bf16 operands: dW[n][kk] += sum_m A[m][n] actB(B[m][kk]), M = K = 128.  Same scheme as k_mm_tn (64-row tiles row-major
// in LDS, transposed fragment reads, row range split over blockIdx.z, ordered reduction of the partial tiles); the staging is a straight
// 16-byte copy unless the activation prologue is on.
__global__ void __launch_bounds__(256, 3) k_emm_tn(TRows rows, const tb16* __restrict__ A, const tb16* __restrict__ B, float* __restrict__ part,
        size_t pstride, int rows_per_split, int actB, TDrop dr, unsigned site, float* __restrict__ cs_part) {
    // 40 KiB of LDS and <= 168 registers: three workgroups per CU, so that one's staging VALU (the activation) runs under the others' MFMAs
    __shared__ __attribute__((aligned(16))) unsigned short tA[64 * TN_PITCH], tB[64 * TN_PITCH];
    float (*cs_red)[128] = reinterpret_cast<float (*)[128]>(tA);      // [16][128] floats = 8 KiB: reuses the A tile after the last MFMAs
    const int R = nrows(rows);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int p_begin = blockIdx.z * rows_per_split, p_end = min(R, p_begin + rows_per_split);
    const int ch = tid & 15, rg = tid >> 4;                   // 16-byte chunk ch of rows rg, rg+16, rg+32, rg+48
    const unsigned key = drop_key(dr, site);
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const tu32x4 z4 = {0u, 0u, 0u, 0u};
    // raw tile loads one tile ahead of the MFMAs; the activation prologue runs when the tile is moved to LDS
    auto load_tile = [&](int m0, tu32x4 (&xa)[4], tu32x4 (&xb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + rg + 16 * i;
            const int mc = m < p_end ? m : (R > 0 ? R - 1 : 0);
            xa[i] = *reinterpret_cast<const tu32x4*>(A + (size_t)mc * 128 + 8 * ch);
            xb[i] = *reinterpret_cast<const tu32x4*>(B + (size_t)mc * 128 + 8 * ch);
        }
    };
    tf32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    auto stage = [&](int m0, tu32x4 (&xa)[4], tu32x4 (&xb)[4]) {
        __syncthreads();                                      // the previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + rg + 16 * i;
            const bool ok = m < p_end;
            tu32x4 va = ok ? xa[i] : z4, vb = xb[i];
            if (actB) {
                float v[8], dm[8];
                unpack8(vb, v);
                drop8(dr, key, (unsigned)m * 16u + ch, dm);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = gelu_fast(v[q]) * dm[q];
                vb = tpack8(v);
            }
            if (!ok) vb = z4;
            *reinterpret_cast<tu32x4*>(tA + (rg + 16 * i) * TN_PITCH + 8 * ch) = va;
            *reinterpret_cast<tu32x4*>(tB + (rg + 16 * i) * TN_PITCH + 8 * ch) = vb;
            if (cs_part) {
                float v[8];
                unpack8(va, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) csum[q] += v[q];
            }
        }
        __syncthreads();
    };
    auto compute = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            tu32x4 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = tr_frag(tA, 16 * ks, 64 * wr + 32 * a, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = tr_frag(tB, 16 * ks, 64 * wc + 32 * b, lane);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = tmfma(af[a], bf[b], acc[a][b]);
        }
    };
    tu32x4 a0[4], b0[4];
    int m0 = p_begin;
    if (m0 < p_end) load_tile(m0, a0, b0);
    while (m0 < p_end) {
        stage(m0, a0, b0);
        load_tile(m0 + 64, a0, b0);                           // next tile's loads fly under this tile's MFMAs (and the other two workgroups of the CU)
        compute();
        m0 += 64;
    }
    if (cs_part) {                                            // rows rg, rg+16, ... were summed per thread: fold the 16 row groups in order
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) cs_red[rg][8 * ch + q] = csum[q];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += cs_red[g][tid];
            cs_part[(size_t)blockIdx.z * pstride + tid] = t;
        }
    }
    float* dst = part + (size_t)blockIdx.z * pstride;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = 64 * wc + 32 * b + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[(size_t)(64 * wr + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h) * 128 + col] = acc[a][b][i];
        }
}
void te_gemm_tn(const TRows& rows, const tb16* A, const tb16* B, float* dW, int ldw, const TScratch& sc_in, bool actB, const TDrop& dr,
                unsigned site, float* dbias, hipStream_t s, int cols_keep) {
    const TScratch sc = red_acquire(sc_in);
    const size_t mk = 128 * 128;
    long long cap = (long long)((sc.floats - (size_t)800 * 128) / mk) - 16;
    int splits = (rows.maxrows + 1023) / 1024;
    const int want = 3 * rn_num_cus();                       // three resident workgroups per CU
    if (splits > want) splits = want;
    if (splits > cap) splits = (int)cap;
    if (splits > 768) splits = 768;
    if (splits < 1) splits = 1;
    const int rps = ((rows.maxrows + splits - 1) / splits + 63) / 64 * 64;
    const size_t pstride = mk + (dbias ? 128 : 0);
    float* tmp = sc.p + (size_t)splits * pstride;
    hipLaunchKernelGGL(k_emm_tn, dim3(1, 1, splits), dim3(256), 0, s, rows, A, B, sc.p, pstride, rps, actB ? 1 : 0, dr, site,
                       dbias ? sc.p + mk : (float*)nullptr);
    reduce_parts(sc.p, splits, pstride, (int)pstride, 128, dW, ldw, s, tmp, (int)mk, dbias, cols_keep);
}

// ---- the two kernels that consume d pre2 of a depth-2 per-edge MLP, fused: dW2 += dpre2^T a1, db2 += colsum(dpre2) (k_emm_tn with
// the activation recomputed) AND d pre1 = (dpre2 . W2) * gelu'(pre1) * mask (k_emm128 epilogue form).  Separately they read dpre2 and pre1
// twice and evaluate the dropout hash and the sigmoid of the GELU twice per element; here a 64-row tile of each is loaded once, ONE pass over
// pre1 produces a1 = gelu * mask (the TN operand) and g' = gelu' * mask (an LDS tile for the NN epilogue), and the same dpre2 tile in LDS
// feeds the transposed reads of the weight-gradient MFMAs and, as row-major fragments, the MFMAs of d pre1 (output transposed in the
// accumulators as in k_emm128: lane = row, 16-byte stores).  W2's fragment image is staged through the LDS region that then holds g'.
__device__ __forceinline__ void gelu_both_fast(float x, float& g, float& d) {          // (gelu_fast(x), gelu_d_fast(x)) sharing the sigmoid
#ifdef TE_EXP_NOACT
    g = x * fmaf(x, 0.25f, 0.5f); d = fmaf(x, 0.5f, 0.5f); return;
#endif
    const float p = fmaf(x * x, -0.10012571f, -2.3087657f);
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * p));
    g = x * sg;
    d = fmaf(x * 0.3989422804f, __builtin_amdgcn_exp2f(x * x * -0.72134752f), sg);
}
// MODE: where d pre2 comes from.  0: the tensor dY.  1 (edge update, mpnn.py:250-262): d pre2 = valid ? dY * gelu'(PRE2) * mask(site2) : 0 with
// dY = d e_out - the residual backward formed while the tile is staged instead of by a kernel of its own (one read and one write of an
// [E][128] tensor less).  2 (message mean, mpnn.py:212-219): d pre2 = valid ? dagg[row / k] / cnt[row / k] * gelu'(PRE2) * mask(site2) : 0.
// The staged values are rounded to bf16 exactly as the stand-alone kernels stored them: results are bit-identical to the two-kernel form.
struct Bwd2Src { const tb16* pre2; const int* nbr; const float* dagg; const float* inv_cnt; int k; unsigned site2; int g2tape; };
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_emm_bwd2(TRows rows, const tb16* __restrict__ dY, const tb16* __restrict__ PRE, tb16* __restrict__ DX,
        const float* __restrict__ W, int ldw, const unsigned short* __restrict__ wimg, float* __restrict__ part, size_t pstride,
        int rows_per_split, TDrop dr, unsigned site, float* __restrict__ cs_part, Bwd2Src src) {
    __shared__ __attribute__((aligned(16))) unsigned short tA[64 * TN_PITCH], tB[64 * TN_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short tG[32 * 64 * 8];          // W2 image (32 KiB) first, then the g' tile [64][TN_PITCH]
    float (*cs_red)[128] = reinterpret_cast<float (*)[128]>(tA);
    const int R = nrows(rows);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int p_begin = blockIdx.z * rows_per_split, p_end = min(R, p_begin + rows_per_split);
    const int ch = tid & 15, rg = tid >> 4;
    const unsigned key = drop_key(dr, site);
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const tu32x4 z4 = {0u, 0u, 0u, 0u};
    // this wave's share of d pre1: rows 32 wr .. 32 wr + 31 of the tile, channel blocks 2 wc and 2 wc + 1
    stage_wimage(tG, wimg, W, ldw, false, 1, tid);
    __syncthreads();
    tu32x4 wf[8][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) wf[ks][u] = reinterpret_cast<const tu32x4*>(tG)[(ks * 4 + 2 * wc + u) * 64 + lane];
    const unsigned key2 = drop_key(dr, src.site2);
    tu32x4 c0[MODE == 1 ? 4 : 1];                             // MODE 1: the taped pre2 chunk (MODE 2 carries it in the dY slot)
    auto load_tile = [&](int m0, tu32x4 (&xa)[4], tu32x4 (&xb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + rg + 16 * i;
            const int mc = m < p_end ? m : (R > 0 ? R - 1 : 0);
            xa[i] = *reinterpret_cast<const tu32x4*>((MODE == 2 ? src.pre2 : dY) + (size_t)mc * 128 + 8 * ch);
            xb[i] = *reinterpret_cast<const tu32x4*>(PRE + (size_t)mc * 128 + 8 * ch);
            if constexpr (MODE == 1) c0[i] = *reinterpret_cast<const tu32x4*>(src.pre2 + (size_t)mc * 128 + 8 * ch);
        }
    };
    tf32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    tu32x4 a0[4], b0[4];
    int m0 = p_begin;
    if (m0 < p_end) load_tile(m0, a0, b0);
    while (m0 < p_end) {
        __syncthreads();                                      // the previous tile's fragment reads (and, first time, the image reads) are done
        tu32x4 dy4[4];                                        // d pre2 of this thread's four chunks
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dy4[i] = a0[i];
        } else {
            int jv[4];
            tf32x4 ga[MODE == 2 ? 4 : 1], gb[MODE == 2 ? 4 : 1];
            float ic[MODE == 2 ? 4 : 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                     // (all loads of the phase first)
                const int m = m0 + rg + 16 * i;
                const int mc = m < p_end ? m : (R > 0 ? R - 1 : 0);
                jv[i] = src.nbr[mc];
                if constexpr (MODE == 2) {
                    const int res = mc / src.k;
                    ga[i] = *reinterpret_cast<const tf32x4*>(src.dagg + (size_t)res * 128 + 8 * ch);
                    gb[i] = *reinterpret_cast<const tf32x4*>(src.dagg + (size_t)res * 128 + 8 * ch + 4);
                    ic[i] = src.inv_cnt[res];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + rg + 16 * i;
                float up[8], p2[8], dm2[8];
                if constexpr (MODE == 1) { unpack8(a0[i], up); unpack8(c0[i], p2); }
                else {
                    unpack8(a0[i], p2);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { up[q] = ga[i][q] * ic[i]; up[4 + q] = gb[i][q] * ic[i]; }
                }
                if (src.g2tape) {                // (uniform) the tape holds gelu'(pre2) * mask already
#pragma unroll
                    for (int q = 0; q < 8; ++q) up[q] *= p2[q];
                } else {
                    drop8(dr, key2, (unsigned)m * 16u + ch, dm2);
#pragma unroll
                    for (int q = 0; q < 8; ++q) up[q] = MODE == 1 ? up[q] * (gelu_d_fast(p2[q]) * dm2[q]) : up[q] * gelu_d_fast(p2[q]) * dm2[q];   // (the stand-alone kernels' association)
                }
                dy4[i] = jv[i] >= 0 ? tpack8(up) : z4;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + rg + 16 * i;
            const bool ok = m < p_end;
            const tu32x4 va = ok ? dy4[i] : z4;
            float v[8], dm[8], a1[8], gp[8];
            unpack8(b0[i], v);
            if (MODE != 0 && src.g2tape) {       // (uniform) dropped elements are TE_DROPPED on the tape: both functions vanish there, no hash
#pragma unroll
                for (int q = 0; q < 8; ++q) { float g, d; gelu_both_fast(v[q], g, d); a1[q] = g * dr.scale; gp[q] = d * dr.scale; }
            } else {
                drop8(dr, key, (unsigned)m * 16u + ch, dm);
#pragma unroll
                for (int q = 0; q < 8; ++q) { float g, d; gelu_both_fast(v[q], g, d); a1[q] = g * dm[q]; gp[q] = d * dm[q]; }
            }
            *reinterpret_cast<tu32x4*>(tA + (rg + 16 * i) * TN_PITCH + 8 * ch) = va;
            *reinterpret_cast<tu32x4*>(tB + (rg + 16 * i) * TN_PITCH + 8 * ch) = ok ? tpack8(a1) : z4;
            *reinterpret_cast<tu32x4*>(tG + (rg + 16 * i) * TN_PITCH + 8 * ch) = ok ? tpack8(gp) : z4;
            if (cs_part) {
                float w8[8];
                unpack8(va, w8);
#pragma unroll
                for (int q = 0; q < 8; ++q) csum[q] += w8[q];
            }
        }
        __syncthreads();
        load_tile(m0 + 64, a0, b0);
        // weight gradient: acc += dpre2^T a1 (transposed reads of both tiles)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            tu32x4 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = tr_frag(tA, 16 * ks, 64 * wr + 32 * a, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = tr_frag(tB, 16 * ks, 64 * wc + 32 * b, lane);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = tmfma(af[a], bf[b], acc[a][b]);
        }
        // d pre1 of this wave's 32 rows x 64 channels: D = W2' . dpre2^T, the dpre2 rows as row-major fragments of the same LDS tile
        {
            const unsigned short* xrow = tA + (32 * wr + r) * TN_PITCH + 8 * h;
            tf32x16 dn[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) dn[u][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const tu32x4 xf = *reinterpret_cast<const tu32x4*>(xrow + 16 * ks);
#pragma unroll
                for (int u = 0; u < 2; ++u) dn[u] = tmfma(wf[ks][u], xf, dn[u]);
            }
            const int m = m0 + 32 * wr + r;
            const unsigned short* grow = tG + (32 * wr + r) * TN_PITCH + 8 * h;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int c = 32 * (2 * wc + u) + 16 * g;           // + 8 h (in grow / the store address)
                    float gq[8], o[8];
                    unpack8(*reinterpret_cast<const tu32x4*>(grow + c), gq);
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = dn[u][8 * g + q] * gq[q];
                    if (m < p_end) *reinterpret_cast<tu32x4*>(DX + (size_t)m * 128 + c + 8 * h) = tpack8(o);
                }
        }
        m0 += 64;
    }
    if (cs_part) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) cs_red[rg][8 * ch + q] = csum[q];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += cs_red[g][tid];
            cs_part[(size_t)blockIdx.z * pstride + tid] = t;
        }
    }
    float* dst = part + (size_t)blockIdx.z * pstride;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = 64 * wc + 32 * b + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[(size_t)(64 * wr + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h) * 128 + col] = acc[a][b][i];
        }
}
// ---- the two kernels that consume d pre1, fused the same way: dWc += dpre1^T e (k_emm_tn, no activation) AND dE += dpre1 . Wc (k_emm128
// with the old-Y epilogue): one pass over dpre1 instead of two.  The weight image is staged through the tile region before the first tile.
__global__ void __launch_bounds__(256, 2) k_emm_bwd1(TRows rows, const tb16* __restrict__ dY, const tb16* __restrict__ Xin, tb16* __restrict__ DE,
        const float* __restrict__ W, int ldw, const unsigned short* __restrict__ wimg, float* __restrict__ part, size_t pstride,
        int rows_per_split) {
    __shared__ __attribute__((aligned(16))) unsigned short tAB[2 * 64 * TN_PITCH];     // [A tile | B tile]; first the 32-KiB weight image
    unsigned short* tA = tAB;
    unsigned short* tB = tAB + 64 * TN_PITCH;
    const int R = nrows(rows);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int p_begin = blockIdx.z * rows_per_split, p_end = min(R, p_begin + rows_per_split);
    const int ch = tid & 15, rg = tid >> 4;
    const tu32x4 z4 = {0u, 0u, 0u, 0u};
    stage_wimage(tAB, wimg, W, ldw, false, 1, tid);
    __syncthreads();
    tu32x4 wf[8][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) wf[ks][u] = reinterpret_cast<const tu32x4*>(tAB)[(ks * 4 + 2 * wc + u) * 64 + lane];
    auto load_tile = [&](int m0, tu32x4 (&xa)[4], tu32x4 (&xb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + rg + 16 * i;
            const int mc = m < p_end ? m : (R > 0 ? R - 1 : 0);
            xa[i] = *reinterpret_cast<const tu32x4*>(dY + (size_t)mc * 128 + 8 * ch);
            xb[i] = *reinterpret_cast<const tu32x4*>(Xin + (size_t)mc * 128 + 8 * ch);
        }
    };
    tf32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    tu32x4 a0[4], b0[4];
    int m0 = p_begin;
    if (m0 < p_end) load_tile(m0, a0, b0);
    while (m0 < p_end) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = m0 + rg + 16 * i < p_end;
            *reinterpret_cast<tu32x4*>(tA + (rg + 16 * i) * TN_PITCH + 8 * ch) = ok ? a0[i] : z4;
            *reinterpret_cast<tu32x4*>(tB + (rg + 16 * i) * TN_PITCH + 8 * ch) = ok ? b0[i] : z4;
        }
        __syncthreads();
        load_tile(m0 + 64, a0, b0);
        // the old dE of this lane's row (4 x 16 bytes: channel groups of its two channel blocks) goes out before the MFMAs
        const int m = m0 + 32 * wr + r;
        const int mc = m < p_end ? m : (R > 0 ? R - 1 : 0);
        tu32x4 old[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 2; ++g) old[u][g] = *reinterpret_cast<const tu32x4*>(DE + (size_t)mc * 128 + 32 * (2 * wc + u) + 16 * g + 8 * h);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            tu32x4 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = tr_frag(tA, 16 * ks, 64 * wr + 32 * a, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = tr_frag(tB, 16 * ks, 64 * wc + 32 * b, lane);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = tmfma(af[a], bf[b], acc[a][b]);
        }
        {
            const unsigned short* xrow = tA + (32 * wr + r) * TN_PITCH + 8 * h;
            tf32x16 dn[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) dn[u][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const tu32x4 xf = *reinterpret_cast<const tu32x4*>(xrow + 16 * ks);
#pragma unroll
                for (int u = 0; u < 2; ++u) dn[u] = tmfma(wf[ks][u], xf, dn[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float ov[8], o[8];
                    unpack8(old[u][g], ov);
#pragma unroll
                    for (int q = 0; q < 8; ++q) o[q] = dn[u][8 * g + q] + ov[q];
                    if (m < p_end) *reinterpret_cast<tu32x4*>(DE + (size_t)m * 128 + 32 * (2 * wc + u) + 16 * g + 8 * h) = tpack8(o);
                }
        }
        m0 += 64;
    }
    float* dst = part + (size_t)blockIdx.z * pstride;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = 64 * wc + 32 * b + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[(size_t)(64 * wr + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h) * 128 + col] = acc[a][b][i];
        }
}
// dW[128][ldw_out] += dY^T X,  DE += dY . W       (W [128 out][ldw]: the Wc block of a first Linear)
void te_gemm_bwd1(const TRows& rows, const tb16* dY, const tb16* X, tb16* DE, const float* W, int ldw, float* dW, int ldw_out,
                  const TScratch& sc_in, hipStream_t s) {
    const TScratch sc = red_acquire(sc_in);
    const size_t mk = 128 * 128;
    long long cap = (long long)((sc.floats - (size_t)800 * 128) / mk) - 16;
    int splits = (rows.maxrows + 511) / 512;                 // >= 8 tiles per workgroup; fills the chip from ~130 K rows on
    const int want = 2 * rn_num_cus();
    if (splits > want) splits = want;
    if (splits > cap) splits = (int)cap;
    if (splits < 1) splits = 1;
    const int rps = ((rows.maxrows + splits - 1) / splits + 63) / 64 * 64;
    float* tmp = sc.p + (size_t)splits * mk;
    hipLaunchKernelGGL(k_emm_bwd1, dim3(1, 1, splits), dim3(256), 0, s, rows, dY, X, DE, W, ldw, wimg_lookup(W, ldw, false, 1), sc.p, mk, rps);
    reduce_parts(sc.p, splits, mk, (int)mk, 128, dW, ldw_out, s, tmp);
}
// ---- the first-Linear backward of the TWO per-edge MLPs of a layer that share their input e (edge update and message MLP, mpnn.py:212-262) in one
// pass: dW1 += dY1^T e, dW2 += dY2^T e, dE += dY1 . W1 + dY2 . W2.  As two k_emm_bwd1 launches e and dE are read twice and dE is written twice
// (8 [E][128] passes); here once each (5).  One 8-wave workgroup per CU (two waves per SIMD): three 64-row LDS tiles + both weight images
// (32 KiB each, fragments read per use - two register-resident slices would not leave room for the four accumulator tiles of a wave).
// Weight-gradient share of a wave: output rows 32 (wave >> 1), columns 64 (wave & 1) of both products; dE share: rows 32 (wave >> 2) of the
// tile, channel block wave & 3.  Same accumulation order per output element as the two-launch form over a split: results differ from it only
// through the split boundaries (partials are per split) - compared against it in tests/test_round3_gpu.py.
__global__ void __launch_bounds__(512, 1) k_emm_bwd1x2(TRows rows, const tb16* __restrict__ dY1, const tb16* __restrict__ dY2,
        const tb16* __restrict__ Xin, tb16* __restrict__ DE, const float* __restrict__ W1, const float* __restrict__ W2, int ldw,
        const unsigned short* __restrict__ wimg1, const unsigned short* __restrict__ wimg2, float* __restrict__ part, size_t pstride,
        int rows_per_split) {
    __shared__ __attribute__((aligned(16))) unsigned short tA1[64 * TN_PITCH], tA2[64 * TN_PITCH], tB[64 * TN_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short img1[32 * 64 * 8], img2[32 * 64 * 8];
    const int R = nrows(rows);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;                  // weight-gradient share
    const int rh = wave >> 2, cbw = wave & 3;                 // dE share
    const int p_begin = blockIdx.z * rows_per_split, p_end = min(R, p_begin + rows_per_split);
    const int ch = tid & 15, rg = tid >> 4;                   // 16-byte chunk ch of rows rg, rg + 32
    const tu32x4 z4 = {0u, 0u, 0u, 0u};
    if (tid < 256) stage_wimage(img1, wimg1, W1, ldw, false, 1, tid);
    else stage_wimage(img2, wimg2, W2, ldw, false, 1, tid - 256);
    auto load_tile = [&](int m0, tu32x4 (&x1)[2], tu32x4 (&x2)[2], tu32x4 (&xb)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + rg + 32 * i;
            const int mc = m < p_end ? m : (R > 0 ? R - 1 : 0);
            x1[i] = *reinterpret_cast<const tu32x4*>(dY1 + (size_t)mc * 128 + 8 * ch);
            x2[i] = *reinterpret_cast<const tu32x4*>(dY2 + (size_t)mc * 128 + 8 * ch);
            xb[i] = *reinterpret_cast<const tu32x4*>(Xin + (size_t)mc * 128 + 8 * ch);
        }
    };
    tf32x16 acc1[2], acc2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc1[b][i] = 0.f; acc2[b][i] = 0.f; }
    const tu32x4* w1 = reinterpret_cast<const tu32x4*>(img1) + cbw * 64 + lane;
    const tu32x4* w2 = reinterpret_cast<const tu32x4*>(img2) + cbw * 64 + lane;
    tu32x4 a1[2], a2[2], b0[2];
    int m0 = p_begin;
    if (m0 < p_end) load_tile(m0, a1, a2, b0);
    while (m0 < p_end) {
        __syncthreads();                                      // the previous tile's fragment reads (first time: the image writes) are done
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = m0 + rg + 32 * i < p_end;
            *reinterpret_cast<tu32x4*>(tA1 + (rg + 32 * i) * TN_PITCH + 8 * ch) = ok ? a1[i] : z4;
            *reinterpret_cast<tu32x4*>(tA2 + (rg + 32 * i) * TN_PITCH + 8 * ch) = ok ? a2[i] : z4;
            *reinterpret_cast<tu32x4*>(tB + (rg + 32 * i) * TN_PITCH + 8 * ch) = ok ? b0[i] : z4;
        }
        __syncthreads();
        load_tile(m0 + 64, a1, a2, b0);
        const int m = m0 + 32 * rh + r;
        const int mc = m < p_end ? m : (R > 0 ? R - 1 : 0);
        tu32x4 old[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) old[g] = *reinterpret_cast<const tu32x4*>(DE + (size_t)mc * 128 + 32 * cbw + 16 * g + 8 * h);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const tu32x4 af1 = tr_frag(tA1, 16 * ks, 32 * wr, lane), af2 = tr_frag(tA2, 16 * ks, 32 * wr, lane);
            tu32x4 bf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = tr_frag(tB, 16 * ks, 64 * wc + 32 * b, lane);
#pragma unroll
            for (int b = 0; b < 2; ++b) { acc1[b] = tmfma(af1, bf[b], acc1[b]); acc2[b] = tmfma(af2, bf[b], acc2[b]); }
        }
        {
            const unsigned short* x1 = tA1 + (32 * rh + r) * TN_PITCH + 8 * h;
            const unsigned short* x2 = tA2 + (32 * rh + r) * TN_PITCH + 8 * h;
            tf32x16 dn;
#pragma unroll
            for (int i = 0; i < 16; ++i) dn[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                dn = tmfma(w1[ks * 4 * 64], *reinterpret_cast<const tu32x4*>(x1 + 16 * ks), dn);
                dn = tmfma(w2[ks * 4 * 64], *reinterpret_cast<const tu32x4*>(x2 + 16 * ks), dn);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float ov[8], o[8];
                unpack8(old[g], ov);
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = dn[8 * g + q] + ov[q];
                if (m < p_end) *reinterpret_cast<tu32x4*>(DE + (size_t)m * 128 + 32 * cbw + 16 * g + 8 * h) = tpack8(o);
            }
        }
        m0 += 64;
    }
    float* d1 = part + (size_t)blockIdx.z * pstride;
    float* d2 = d1 + 128 * 128;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = 64 * wc + 32 * b + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const size_t o = (size_t)(32 * wr + (i & 3) + 8 * (i >> 2) + 4 * h) * 128 + col;
            d1[o] = acc1[b][i]; d2[o] = acc2[b][i];
        }
    }
}
// dW1[128][ldw_out] += dY1^T X, dW2 += dY2^T X, DE += dY1 . W1 + dY2 . W2      (W1, W2 [128 out][ldw]: the Wc blocks of the two first Linears)
void te_gemm_bwd1x2(const TRows& rows, const tb16* dY1, const tb16* dY2, const tb16* X, tb16* DE, const float* W1, const float* W2, int ldw,
                    float* dW1, float* dW2, int ldw_out, const TScratch& sc_in, hipStream_t s) {
    const TScratch sc = red_acquire(sc_in);
    const size_t mk = 128 * 128, pstride = 2 * mk;
    long long cap = (long long)((sc.floats - (size_t)800 * 128) / pstride) - 16;
    int splits = (rows.maxrows + 1023) / 1024;               // >= 16 tiles per workgroup
    const int want = rn_num_cus();                           // one 8-wave workgroup per CU (127 KiB of LDS)
    if (splits > want) splits = want;
    if (splits > cap) splits = (int)cap;
    if (splits < 1) splits = 1;
    const int rps = ((rows.maxrows + splits - 1) / splits + 63) / 64 * 64;
    float* tmp = sc.p + (size_t)splits * pstride;
    hipLaunchKernelGGL(k_emm_bwd1x2, dim3(1, 1, splits), dim3(512), 0, s, rows, dY1, dY2, X, DE, W1, W2, ldw, wimg_lookup(W1, ldw, false, 1),
                       wimg_lookup(W2, ldw, false, 1), sc.p, pstride, rps);
    reduce_parts(sc.p, splits, pstride, (int)mk, 128, dW1, ldw_out, s, tmp);
    reduce_parts(sc.p + mk, splits, pstride, (int)mk, 128, dW2, ldw_out, s, tmp);
}
// dW[128][ldw_out] += dY^T drop(gelu(PRE)), dbias += colsum(dY), DX = (dY . W) gelu'(PRE) mask        (W [128 out][ldw] as nn.Linear stores it)
void te_gemm_bwd2(const TRows& rows, const tb16* dY, const tb16* PRE, tb16* DX, const float* W, int ldw, float* dW, int ldw_out,
                  const TScratch& sc_in, const TDrop& dr, unsigned site, float* dbias, hipStream_t s, const EBwd2Src* from) {
    const TScratch sc = red_acquire(sc_in);
    const size_t mk = 128 * 128;
    long long cap = (long long)((sc.floats - (size_t)800 * 128) / mk) - 16;
    int splits = (rows.maxrows + 511) / 512;                 // >= 8 tiles per workgroup; fills the chip from ~130 K rows on
    const int want = 2 * rn_num_cus();                       // two resident workgroups per CU (72 KiB of LDS each)
    if (splits > want) splits = want;
    if (splits > cap) splits = (int)cap;
    if (splits < 1) splits = 1;
    const int rps = ((rows.maxrows + splits - 1) / splits + 63) / 64 * 64;
    const size_t pstride = mk + (dbias ? 128 : 0);
    float* tmp = sc.p + (size_t)splits * pstride;
    Bwd2Src src{nullptr, nullptr, nullptr, nullptr, 1, 0u, 0};
    const int mode = from ? from->mode : 0;
    if (from) src = Bwd2Src{from->pre2, from->nbr, from->dagg, from->inv_cnt, from->k, from->site2, from->g2tape};
#define BWD2_GO(M) hipLaunchKernelGGL(k_emm_bwd2<M>, dim3(1, 1, splits), dim3(256), 0, s, rows, dY, PRE, DX, W, ldw, wimg_lookup(W, ldw, false, 1), \
                                       sc.p, pstride, rps, dr, site, dbias ? sc.p + mk : (float*)nullptr, src)
    if (mode == 1) BWD2_GO(1); else if (mode == 2) BWD2_GO(2); else BWD2_GO(0);
#undef BWD2_GO
    reduce_parts(sc.p, splits, pstride, (int)pstride, 128, dW, ldw_out, s, tmp, (int)mk, dbias);
}

// ------------------------------------------------------------------------------------------
// bf16-mixed training attention on MFMA (head dim 16; the f32 kernels above stay the parity-grade path).  The thread-per-query f32
// kernels run at the FMA rate of the vector units (~50 TFLOP/s): on the config-3 epoch - RNAs of up to 4,417 nt, work ~ T^2 - the three of
// them were 28 % of the kernel time.  Same transposed scheme as the inference kernel (kernels_bf16.hip: k_attention_bf16_hd16):
//   forward   S^T[key][query] = K . Q^T (one query per lane, the 32 keys of a block on the accumulator registers), online softmax in
//             lane, dropout mask on the probabilities (the hash of drop_mul / att_idx, one hash per pair of adjacent keys), the masked P
//             tile is the B operand of O^T[d][query] += V^T . P; the row max and normaliser go to the tape (stat[0..1]).
//   dQ        per 32-key block: S^T and dP^T[key][query] = V . dO^T by one MFMA each, dS = P (dP M - delta) with delta = dO . O,
//             dQ^T[d][query] += K^T . dS (the K^T image is staged like the forward's V^T image).
//   dK, dV    the same with the roles swapped (one KEY per lane, 32 queries of a block on the registers; row statistics of the queries
//             from LDS): dV^T[d][key] += dO^T . (P M), dK^T[d][key] += Q^T . dS.
// Q is pre-scaled by 1/4 before its bf16 rounding in all three, so the recomputed S equals the forward's bit for bit and exp(S - m) <= 1.
// K / V / Q / dO tiles are staged in LDS per chunk of 32-row blocks as ready fragments; any RNA length runs through the same kernels.
__device__ __forceinline__ float att_mask(const TDrop& d, unsigned hb, int key) {        // == drop_mul(d, site, att_idx(q, heads, hd, key)); hb = hash base of (q, head)
    if (d.thresh == 0u) return 1.f;
    const unsigned x = drop_hash(hb + ((unsigned)key >> 1));
    return ((key & 1) ? x >> 16 : x & 0xffffu) >= d.thresh ? d.scale : 0.f;
}
__device__ __forceinline__ unsigned att_hash_base(const TDrop& d, unsigned key_site, int qrow, int heads, int hd) {
    const unsigned long long P = ((unsigned long long)((unsigned)qrow * (unsigned)heads + (unsigned)hd)) << 12;     // att_idx(...) >> 1 with key = 0 (low 12 bits free)
    return (unsigned)P + (unsigned)(P >> 32) * 0xC2B2AE35u + key_site;
}
// rows [row0, row0 + 32 nb) of qkv column block `col` (+ 16 hd) -> [row][2 halves] A-fragment image (8 bf16 each), scaled; rows >= n are zero
__device__ __forceinline__ void att_stage_rows(tu32x4* img, const float* __restrict__ src, int ld, int base, int row0, int nb, int n, float scl, int tid, int nthr) {
    const tu32x4 z4 = {0u, 0u, 0u, 0u};
    for (int idx = tid; idx < nb * 64; idx += nthr) {
        const int row = row0 + (idx >> 1), hh = idx & 1;
        const float* p = src + (size_t)(base + (row < n ? row : n - 1)) * ld + 8 * hh;
        const tf32x4 a = *reinterpret_cast<const tf32x4*>(p), c = *reinterpret_cast<const tf32x4*>(p + 4);
        const tu32x4 v = {tpack2(scl * a[0], scl * a[1]), tpack2(scl * a[2], scl * a[3]), tpack2(scl * c[0], scl * c[1]), tpack2(scl * c[2], scl * c[3])};
        img[idx] = row < n ? v : z4;
    }
}
// the same rows TRANSPOSED: [block][s][h][d 0..15] = 8 rows in the order the packed accumulator registers carry them (A operand of X^T . tile)
__device__ __forceinline__ void att_stage_rows_t(tu32x4* img, const float* __restrict__ src, int ld, int base, int row0, int nb, int n, float scl, int tid, int nthr) {
    for (int idx = tid; idx < nb * 64; idx += nthr) {
        const int d = idx & 15, hh = (idx >> 4) & 1, sblk = (idx >> 5) & 1, blk = idx >> 6;
        float vals[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = row0 + 32 * blk + 16 * sblk + 8 * (j >> 2) + 4 * hh + (j & 3);
            vals[j] = src[(size_t)(base + (row < n ? row : n - 1)) * ld + d];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = row0 + 32 * blk + 16 * sblk + 8 * (j >> 2) + 4 * hh + (j & 3);
            vals[j] = row < n ? scl * vals[j] : 0.f;
        }
        img[idx] = tu32x4{tpack2(vals[0], vals[1]), tpack2(vals[2], vals[3]), tpack2(vals[4], vals[5]), tpack2(vals[6], vals[7])};
    }
}
#define ATT_CHUNK 32            // 32-row blocks per LDS chunk
__global__ void __launch_bounds__(512) k_attn_fwd_m16(PackInfo pk, const float* __restrict__ qkv, int heads, float* __restrict__ out,
                                                      float* __restrict__ stat, TDrop dr, unsigned site) {
    __shared__ __attribute__((aligned(16))) tu32x4 Kimg[ATT_CHUNK * 64], Vt[ATT_CHUNK * 64];
    const int b = blockIdx.x, hd = blockIdx.y;
    const int n = pk.len[b];
    const int qbase = blockIdx.z * 256;
    if (n <= 0 || qbase >= n) return;
    const int base = pk.cu[b];
    const int nkb = (n + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const tu32x4 z4 = {0u, 0u, 0u, 0u};
    const int q0 = qbase + 32 * wave;
    const bool wave_live = q0 < n;
    const int qi = q0 + r;
    const int qrow = base + (qi < n ? qi : n - 1);
    tu32x4 qf = z4;
    if (qi < n) {
        const float* qp = qkv + (size_t)qrow * 384 + hd * 16 + 8 * h;
        const tf32x4 a = *reinterpret_cast<const tf32x4*>(qp), c = *reinterpret_cast<const tf32x4*>(qp + 4);
        qf = tu32x4{tpack2(0.25f * a[0], 0.25f * a[1]), tpack2(0.25f * a[2], 0.25f * a[3]), tpack2(0.25f * c[0], 0.25f * c[1]), tpack2(0.25f * c[2], 0.25f * c[3])};
    }
    const unsigned hb = att_hash_base(dr, drop_key(dr, site), qrow, heads, hd);
    tf32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float m_run = -3.0e38f, l_run = 0.f;
    for (int kb0 = 0; kb0 < nkb; kb0 += ATT_CHUNK) {
        const int cb = min(ATT_CHUNK, nkb - kb0);
        if (kb0 > 0) __syncthreads();
        att_stage_rows(Kimg, qkv + 128 + hd * 16, 384, base, 32 * kb0, cb, n, 1.f, tid, 512);
        att_stage_rows_t(Vt, qkv + 256 + hd * 16, 384, base, 32 * kb0, cb, n, 1.f, tid, 512);
        __syncthreads();
        if (!wave_live) continue;
        for (int kl = 0; kl < cb; ++kl) {
            const int kb = kb0 + kl;
            tf32x16 sc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[i] = 0.f;
            sc = tmfma(Kimg[(32 * kl + r) * 2 + h], qf, sc);                      // S^T[key][query]
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
            for (int i = 0; i < 16; ++i) sc[i] *= att_mask(dr, hb, 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h);       // (adjacent registers = adjacent keys: the compiler shares the pair's hash)
#pragma unroll
            for (int sblk = 0; sblk < 2; ++sblk) {
                const tu32x4 pf = {tpack2(sc[8 * sblk], sc[8 * sblk + 1]), tpack2(sc[8 * sblk + 2], sc[8 * sblk + 3]),
                                   tpack2(sc[8 * sblk + 4], sc[8 * sblk + 5]), tpack2(sc[8 * sblk + 6], sc[8 * sblk + 7])};
                const tu32x4 vf = r < 16 ? Vt[((kl * 2 + sblk) * 2 + h) * 16 + r] : z4;
                acc = tmfma(vf, pf, acc);                                       // O^T[d][query]
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (qi < n) {
        const float inv = 1.0f / l_tot;
        float* op = out + (size_t)qrow * RN_D + hd * 16 + 4 * h;                // rows d = (i&3) + 8(i>>2) + 4h, i < 8
        *reinterpret_cast<tf32x4*>(op) = tf32x4{acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
        *reinterpret_cast<tf32x4*>(op + 8) = tf32x4{acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv};
        if (h == 0) { float* st = stat + ((size_t)qrow * heads + hd) * 3; st[0] = m_run; st[1] = l_tot; }
    }
}
__global__ void __launch_bounds__(512) k_attn_bwd_q_m16(PackInfo pk, const float* __restrict__ qkv, const float* __restrict__ O,
        const float* __restrict__ dO, float* __restrict__ dqkv, float* __restrict__ stat, int heads, TDrop dr, unsigned site) {
    __shared__ __attribute__((aligned(16))) tu32x4 Kimg[ATT_CHUNK * 64], Vimg[ATT_CHUNK * 64], Kt[ATT_CHUNK * 64];
    const int b = blockIdx.x, hd = blockIdx.y;
    const int n = pk.len[b];
    const int qbase = blockIdx.z * 256;
    if (n <= 0 || qbase >= n) return;
    const int base = pk.cu[b];
    const int nkb = (n + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const tu32x4 z4 = {0u, 0u, 0u, 0u};
    const int q0 = qbase + 32 * wave;
    const bool wave_live = q0 < n;
    const int qi = q0 + r;
    const int qrow = base + (qi < n ? qi : n - 1);
    tu32x4 qf = z4, gf = z4;
    float delta = 0.f;
    if (qi < n) {
        const float* qp = qkv + (size_t)qrow * 384 + hd * 16 + 8 * h;
        const tf32x4 a = *reinterpret_cast<const tf32x4*>(qp), c = *reinterpret_cast<const tf32x4*>(qp + 4);
        qf = tu32x4{tpack2(0.25f * a[0], 0.25f * a[1]), tpack2(0.25f * a[2], 0.25f * a[3]), tpack2(0.25f * c[0], 0.25f * c[1]), tpack2(0.25f * c[2], 0.25f * c[3])};
        const float* gp = dO + (size_t)qrow * RN_D + hd * 16 + 8 * h;
        const float* op = O + (size_t)qrow * RN_D + hd * 16 + 8 * h;
        const tf32x4 g0 = *reinterpret_cast<const tf32x4*>(gp), g1 = *reinterpret_cast<const tf32x4*>(gp + 4);
        const tf32x4 o0 = *reinterpret_cast<const tf32x4*>(op), o1 = *reinterpret_cast<const tf32x4*>(op + 4);
        gf = tu32x4{tpack2(g0[0], g0[1]), tpack2(g0[2], g0[3]), tpack2(g1[0], g1[1]), tpack2(g1[2], g1[3])};
#pragma unroll
        for (int d = 0; d < 4; ++d) delta = fmaf(g0[d], o0[d], fmaf(g1[d], o1[d], delta));
    }
    delta += __shfl_xor(delta, 32, 64);                          // the two lane halves hold the two halves of d
    const float* stp = stat + ((size_t)qrow * heads + hd) * 3;
    const float m = stp[0], linv = 1.0f / stp[1];
    const unsigned hb = att_hash_base(dr, drop_key(dr, site), qrow, heads, hd);
    tf32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int kb0 = 0; kb0 < nkb; kb0 += ATT_CHUNK) {
        const int cb = min(ATT_CHUNK, nkb - kb0);
        if (kb0 > 0) __syncthreads();
        att_stage_rows(Kimg, qkv + 128 + hd * 16, 384, base, 32 * kb0, cb, n, 1.f, tid, 512);
        att_stage_rows(Vimg, qkv + 256 + hd * 16, 384, base, 32 * kb0, cb, n, 1.f, tid, 512);
        att_stage_rows_t(Kt, qkv + 128 + hd * 16, 384, base, 32 * kb0, cb, n, 1.f, tid, 512);
        __syncthreads();
        if (!wave_live) continue;
        for (int kl = 0; kl < cb; ++kl) {
            const int kb = kb0 + kl;
            tf32x16 sc, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sc[i] = 0.f; dp[i] = 0.f; }
            sc = tmfma(Kimg[(32 * kl + r) * 2 + h], qf, sc);                      // S^T[key][query]
            dp = tmfma(Vimg[(32 * kl + r) * 2 + h], gf, dp);                      // dP^T[key][query] = v_key . dO_query
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h;
                const float p = __expf(sc[i] - m) * linv;
                const float ds = p * (dp[i] * att_mask(dr, hb, key) - delta);
                sc[i] = key < n ? ds : 0.f;
            }
#pragma unroll
            for (int sblk = 0; sblk < 2; ++sblk) {
                const tu32x4 df = {tpack2(sc[8 * sblk], sc[8 * sblk + 1]), tpack2(sc[8 * sblk + 2], sc[8 * sblk + 3]),
                                   tpack2(sc[8 * sblk + 4], sc[8 * sblk + 5]), tpack2(sc[8 * sblk + 6], sc[8 * sblk + 7])};
                const tu32x4 kf = r < 16 ? Kt[((kl * 2 + sblk) * 2 + h) * 16 + r] : z4;
                acc = tmfma(kf, df, acc);                                       // dQ^T[d][query]
            }
        }
    }
    if (qi < n) {
        float* o = dqkv + (size_t)qrow * 384 + hd * 16 + 4 * h;
        *reinterpret_cast<tf32x4*>(o) = tf32x4{0.25f * acc[0], 0.25f * acc[1], 0.25f * acc[2], 0.25f * acc[3]};
        *reinterpret_cast<tf32x4*>(o + 8) = tf32x4{0.25f * acc[4], 0.25f * acc[5], 0.25f * acc[6], 0.25f * acc[7]};
        if (h == 0) stat[((size_t)qrow * heads + hd) * 3 + 2] = delta;
    }
}
__global__ void __launch_bounds__(512) k_attn_bwd_kv_m16(PackInfo pk, const float* __restrict__ qkv, const float* __restrict__ dO,
        float* __restrict__ dqkv, const float* __restrict__ stat, int heads, TDrop dr, unsigned site) {
    __shared__ __attribute__((aligned(16))) tu32x4 Qimg[ATT_CHUNK * 64], Gimg[ATT_CHUNK * 64], Qt[ATT_CHUNK * 64], Gt[ATT_CHUNK * 64];
    __shared__ float s_m[ATT_CHUNK * 32], s_li[ATT_CHUNK * 32], s_de[ATT_CHUNK * 32];
    __shared__ unsigned s_hb[ATT_CHUNK * 32];
    const int b = blockIdx.x, hd = blockIdx.y;
    const int n = pk.len[b];
    const int kbase = blockIdx.z * 256;
    if (n <= 0 || kbase >= n) return;
    const int base = pk.cu[b];
    const int nqb = (n + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const tu32x4 z4 = {0u, 0u, 0u, 0u};
    const int k0 = kbase + 32 * wave;
    const bool wave_live = k0 < n;
    const int kj = k0 + r;
    const int krow = base + (kj < n ? kj : n - 1);
    tu32x4 kf = z4, vf = z4;
    if (kj < n) {
        const float* kp = qkv + (size_t)krow * 384 + 128 + hd * 16 + 8 * h;
        const tf32x4 a = *reinterpret_cast<const tf32x4*>(kp), c = *reinterpret_cast<const tf32x4*>(kp + 4);
        const tf32x4 v0 = *reinterpret_cast<const tf32x4*>(kp + 128), v1 = *reinterpret_cast<const tf32x4*>(kp + 132);
        kf = tu32x4{tpack2(a[0], a[1]), tpack2(a[2], a[3]), tpack2(c[0], c[1]), tpack2(c[2], c[3])};
        vf = tu32x4{tpack2(v0[0], v0[1]), tpack2(v0[2], v0[3]), tpack2(v1[0], v1[1]), tpack2(v1[2], v1[3])};
    }
    const unsigned key_site = drop_key(dr, site);
    tf32x16 dk, dv;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
    for (int qb0 = 0; qb0 < nqb; qb0 += ATT_CHUNK) {
        const int cb = min(ATT_CHUNK, nqb - qb0);
        if (qb0 > 0) __syncthreads();
        att_stage_rows(Qimg, qkv + hd * 16, 384, base, 32 * qb0, cb, n, 0.25f, tid, 512);
        att_stage_rows(Gimg, dO + hd * 16, RN_D, base, 32 * qb0, cb, n, 1.f, tid, 512);
        att_stage_rows_t(Qt, qkv + hd * 16, 384, base, 32 * qb0, cb, n, 0.25f, tid, 512);
        att_stage_rows_t(Gt, dO + hd * 16, RN_D, base, 32 * qb0, cb, n, 1.f, tid, 512);
        for (int idx = tid; idx < cb * 32; idx += 512) {
            const int q = 32 * qb0 + idx;
            const int qr = base + (q < n ? q : n - 1);
            const float* st = stat + ((size_t)qr * heads + hd) * 3;
            s_m[idx] = st[0]; s_li[idx] = 1.0f / st[1]; s_de[idx] = st[2];
            s_hb[idx] = att_hash_base(dr, key_site, qr, heads, hd);
        }
        __syncthreads();
        if (!wave_live) continue;
        for (int ql = 0; ql < cb; ++ql) {
            const int qb = qb0 + ql;
            tf32x16 sc, dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) { sc[i] = 0.f; dp[i] = 0.f; }
            sc = tmfma(Qimg[(32 * ql + r) * 2 + h], kf, sc);                      // S[query][key], one key per lane
            dp = tmfma(Gimg[(32 * ql + r) * 2 + h], vf, dp);                      // dP[query][key] = dO_query . v_key
            tf32x16 pm;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ql_i = (i & 3) + 8 * (i >> 2) + 4 * h, q = 32 * qb + ql_i;
                const int si = 32 * ql + ql_i;
                const float p = __expf(sc[i] - s_m[si]) * s_li[si];
                const float mk = att_mask(dr, s_hb[si], kj);
                const bool ok = q < n;
                pm[i] = ok ? p * mk : 0.f;
                sc[i] = ok ? p * (dp[i] * mk - s_de[si]) : 0.f;
            }
#pragma unroll
            for (int sblk = 0; sblk < 2; ++sblk) {
                const tu32x4 pf = {tpack2(pm[8 * sblk], pm[8 * sblk + 1]), tpack2(pm[8 * sblk + 2], pm[8 * sblk + 3]),
                                   tpack2(pm[8 * sblk + 4], pm[8 * sblk + 5]), tpack2(pm[8 * sblk + 6], pm[8 * sblk + 7])};
                const tu32x4 df = {tpack2(sc[8 * sblk], sc[8 * sblk + 1]), tpack2(sc[8 * sblk + 2], sc[8 * sblk + 3]),
                                   tpack2(sc[8 * sblk + 4], sc[8 * sblk + 5]), tpack2(sc[8 * sblk + 6], sc[8 * sblk + 7])};
                const tu32x4 gt = r < 16 ? Gt[((ql * 2 + sblk) * 2 + h) * 16 + r] : z4;
                const tu32x4 qt = r < 16 ? Qt[((ql * 2 + sblk) * 2 + h) * 16 + r] : z4;
                dv = tmfma(gt, pf, dv);                                         // dV^T[d][key] += dO^T . (P M)
                dk = tmfma(qt, df, dk);                                         // dK^T[d][key] += (Q / 4)^T . dS
            }
        }
    }
    if (kj < n) {
        float* o = dqkv + (size_t)krow * 384 + 128 + hd * 16 + 4 * h;
        *reinterpret_cast<tf32x4*>(o) = tf32x4{dk[0], dk[1], dk[2], dk[3]};
        *reinterpret_cast<tf32x4*>(o + 8) = tf32x4{dk[4], dk[5], dk[6], dk[7]};
        *reinterpret_cast<tf32x4*>(o + 128) = tf32x4{dv[0], dv[1], dv[2], dv[3]};
        *reinterpret_cast<tf32x4*>(o + 136) = tf32x4{dv[4], dv[5], dv[6], dv[7]};
    }
}
// returns 0 when handled (head dim 16); the caller falls back to the f32 kernels otherwise
int te_attention_fwd(const PackInfo& pk, const float* qkv, int heads, float* out, float* stat, const TDrop& dr, unsigned site, hipStream_t s) {
    if (RN_D / heads != 16 || pk.T > 8192 || !stat) return 1;
    hipLaunchKernelGGL(k_attn_fwd_m16, dim3(pk.B, heads, (pk.T + 255) / 256), dim3(512), 0, s, pk, qkv, heads, out, stat, dr, site);
    return 0;
}
int te_attention_bwd(const PackInfo& pk, const float* qkv, const float* O, const float* dO, int heads, float* dqkv, float* stat,
                     const TDrop& dr, unsigned site, hipStream_t s) {
    if (RN_D / heads != 16 || pk.T > 8192) return 1;
    const dim3 grid(pk.B, heads, (pk.T + 255) / 256);
    hipLaunchKernelGGL(k_attn_bwd_q_m16, grid, dim3(512), 0, s, pk, qkv, O, dO, dqkv, stat, heads, dr, site);
    hipLaunchKernelGGL(k_attn_bwd_kv_m16, grid, dim3(512), 0, s, pk, qkv, dO, dqkv, stat, heads, dr, site);
    return 0;
}

// ---- row kernels on bf16 edge tensors: one thread = two adjacent channels (a 32-bit load), 64 threads per edge row
// One wave per residue, lane = two adjacent channels.  The k validity flags are taken in ONE load + ballot (a per-slot `if (nbr >= 0)` in the
// loop is a dependent load -> branch -> load chain per slot: 83 % of the wave cycles were waits), the row loads are unconditional and go
// out in batches of the unrolled loop.
__global__ void __launch_bounds__(256) k_eseg_mean(PackInfo pk, int k, const int* __restrict__ nbr, const tb16* pre2,
                                                   const float* __restrict__ hin, float* __restrict__ out, TDrop dr, unsigned site, tb16* g2_out) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= pk.cu[pk.B]) return;
    const int lane = threadIdx.x & 63, c = 2 * lane;
    const unsigned key = drop_key(dr, site);
    const unsigned long long vm = __ballot(lane < k && nbr[(size_t)p * k + (lane < k ? lane : 0)] >= 0);
    const int cnt = __popcll(vm);
    const tb16* base = pre2 + (size_t)p * k * RN_D + c;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 6
    for (int sl = 0; sl < k; ++sl) {
        const unsigned w = *reinterpret_cast<const unsigned*>(base + (size_t)sl * RN_D);
        float m0, m1;
        drop_pair(dr, key, (unsigned)(p * k + sl) * 64u + lane, m0, m1);
        const bool valid = (vm >> sl) & 1ull;               // (a select, not a multiply by 0: an absent slot's row may hold anything)
        if (g2_out) {                                       // (uniform) leave gelu' * mask for the backward, in place of pre2 when the two alias
            float g0, d0, g1, d1;
            gelu_both_fast(tbf_lo(w), g0, d0); gelu_both_fast(tbf_hi(w), g1, d1);
            s0 += valid ? g0 * m0 : 0.f;
            s1 += valid ? g1 * m1 : 0.f;
            *reinterpret_cast<unsigned*>(g2_out + (size_t)p * k * RN_D + c + (size_t)sl * RN_D) = valid ? tpack2(d0 * m0, d1 * m1) : 0u;
        } else {
            s0 += valid ? gelu_fast(tbf_lo(w)) * m0 : 0.f;
            s1 += valid ? gelu_fast(tbf_hi(w)) * m1 : 0.f;
        }
    }
    const float inv = 1.0f / (float)(cnt > 0 ? cnt : 1);
    const tf32x2 hv = *reinterpret_cast<const tf32x2*>(hin + (size_t)p * RN_D + c);
    *reinterpret_cast<tf32x2*>(out + (size_t)p * RN_D + c) = tf32x2{hv[0] + s0 * inv, hv[1] + s1 * inv};
}
__global__ void __launch_bounds__(256) k_eseg_mean_bwd(PackInfo pk, int k, const int* __restrict__ nbr, const float* __restrict__ dagg,
                                                       const tb16* __restrict__ pre2, tb16* __restrict__ dpre2, TDrop dr, unsigned site) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= pk.cu[pk.B]) return;
    const int lane = threadIdx.x & 63, c = 2 * lane;
    const unsigned key = drop_key(dr, site);
    const unsigned long long vm = __ballot(lane < k && nbr[(size_t)p * k + (lane < k ? lane : 0)] >= 0);
    const int cnt = __popcll(vm);
    const float inv = 1.0f / (float)(cnt > 0 ? cnt : 1);
    const tf32x2 g = *reinterpret_cast<const tf32x2*>(dagg + (size_t)p * RN_D + c);
    const float g0 = g[0] * inv, g1 = g[1] * inv;
    const tb16* base = pre2 + (size_t)p * k * RN_D + c;
    tb16* obase = dpre2 + (size_t)p * k * RN_D + c;
#pragma unroll 6
    for (int sl = 0; sl < k; ++sl) {
        const unsigned w = *reinterpret_cast<const unsigned*>(base + (size_t)sl * RN_D);
        float m0, m1;
        drop_pair(dr, key, (unsigned)(p * k + sl) * 64u + lane, m0, m1);
        const bool valid = (vm >> sl) & 1ull;
        const unsigned o = tpack2(g0 * gelu_d_fast(tbf_lo(w)) * m0, g1 * gelu_d_fast(tbf_hi(w)) * m1);
        *reinterpret_cast<unsigned*>(obase + (size_t)sl * RN_D) = valid ? o : 0u;
    }
}
// 1 / max(#valid neighbour slots, 1) per residue: the k-NN graph is fixed across the layers, the mean's backward needs it per edge row
__global__ void k_inv_count(PackInfo pk, int k, const int* __restrict__ nbr, float* __restrict__ inv_cnt) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pk.cu[pk.B]) return;
    int cnt = 0;
    for (int sl = 0; sl < k; ++sl) cnt += nbr[(size_t)p * k + sl] >= 0;
    inv_cnt[p] = 1.0f / (float)(cnt > 0 ? cnt : 1);
}
void te_inv_count(const PackInfo& pk, int k, const int* nbr, float* inv_cnt, hipStream_t s) {
    hipLaunchKernelGGL(k_inv_count, dim3((pk.Nmax + 255) / 256), dim3(256), 0, s, pk, k, nbr, inv_cnt);
}
void te_seg_mean(const PackInfo& pk, int k, const int* nbr, const tb16* pre2, const float* h, float* out, const TDrop& dr, unsigned site, hipStream_t s,
                 tb16* g2_out) {
    hipLaunchKernelGGL(k_eseg_mean, dim3((pk.Nmax + 3) / 4), dim3(256), 0, s, pk, k, nbr, pre2, h, out, dr, site, g2_out);
}
void te_seg_mean_bwd(const PackInfo& pk, int k, const int* nbr, const float* dagg, const tb16* pre2, tb16* dpre2, const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_eseg_mean_bwd, dim3((pk.Nmax + 3) / 4), dim3(256), 0, s, pk, k, nbr, dagg, pre2, dpre2, dr, site);
}
// mode 0: dpre2 = valid ? de * gelu'(pre2) * mask : 0;  mode 1: x = 0 on invalid slots       (16-byte units: 8 channels)
__global__ void k_eelem(PackInfo pk, int k, const int* __restrict__ nbr, int mode, const tb16* __restrict__ de, const tb16* __restrict__ pre2,
                        tb16* __restrict__ x, TDrop dr, unsigned site) {
    const size_t n = (size_t)pk.cu[pk.B] * k * 16;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < n; id += (size_t)gridDim.x * blockDim.x) {
        const bool valid = nbr[id >> 4] >= 0;
        tu32x4* xp = reinterpret_cast<tu32x4*>(x) + id;
        if (mode == 1) { if (!valid) *xp = tu32x4{0u, 0u, 0u, 0u}; continue; }
        if (mode == 2) {                                      // x = valid ? drop(gelu(pre2)) : 0
            tu32x4 o2 = {0u, 0u, 0u, 0u};
            if (valid) {
                float pr[8], dm[8];
                unpack8(reinterpret_cast<const tu32x4*>(pre2)[id], pr);
                drop8(dr, drop_key(dr, site), (unsigned)id, dm);
#pragma unroll
                for (int q = 0; q < 8; ++q) pr[q] = gelu_fast(pr[q]) * dm[q];
                o2 = tpack8(pr);
            }
            *xp = o2;
            continue;
        }
        tu32x4 o = {0u, 0u, 0u, 0u};
        if (valid) {
            float d[8], pr[8];
            unpack8(reinterpret_cast<const tu32x4*>(de)[id], d);
            unpack8(reinterpret_cast<const tu32x4*>(pre2)[id], pr);
            float dm[8];
            drop8(dr, drop_key(dr, site), (unsigned)id, dm);
#pragma unroll
            for (int q = 0; q < 8; ++q) d[q] *= gelu_d_fast(pr[q]) * dm[q];
            o = tpack8(d);
        }
        *xp = o;
    }
}
static unsigned eelem_grid(const PackInfo& pk, int k) { size_t g = ((size_t)pk.Nmax * k * 16 + 255) / 256; return (unsigned)(g < 16384 ? (g ? g : 1) : 16384); }
void te_edge_res_bwd(const PackInfo& pk, int k, const int* nbr, const tb16* de, const tb16* pre2, tb16* dpre2, const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_eelem, dim3(eelem_grid(pk, k)), dim3(256), 0, s, pk, k, nbr, 0, de, pre2, dpre2, dr, site);
}
void te_edge_act(const PackInfo& pk, int k, const int* nbr, const tb16* pre, tb16* out, const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_eelem, dim3(eelem_grid(pk, k)), dim3(256), 0, s, pk, k, nbr, 2, nullptr, pre, out, dr, site);
}
void te_zero_invalid(const PackInfo& pk, int k, const int* nbr, tb16* x, hipStream_t s) {
    hipLaunchKernelGGL(k_eelem, dim3(eelem_grid(pk, k)), dim3(256), 0, s, pk, k, nbr, 1, nullptr, nullptr, x, TDrop{0, 0, 1.f}, 0u);
}
// one wave per residue: lane = (row group g of 4, 16-byte chunk c16 of the 256-byte row); the 4 groups are folded with fixed-order shuffles
__global__ void __launch_bounds__(256) k_epq_bwd(PackInfo pk, int k, const tb16* __restrict__ dpre1, const int* __restrict__ start,
                                                 const int* __restrict__ list, float* __restrict__ dpq) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= pk.cu[pk.B]) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = 8 * (lane & 15);
    float sp[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sl = g; sl < k; sl += 4) {
        float v[8];
        unpack8(*reinterpret_cast<const tu32x4*>(dpre1 + ((size_t)p * k + sl) * RN_D + c), v);
#pragma unroll
        for (int q = 0; q < 8; ++q) sp[q] += v[q];
    }
    const int t1 = start[p + 1];
    for (int t = start[p] + g; t < t1; t += 4) {
        float v[8];
        unpack8(*reinterpret_cast<const tu32x4*>(dpre1 + (size_t)list[t] * RN_D + c), v);
#pragma unroll
        for (int q = 0; q < 8; ++q) sq[q] += v[q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        sp[q] += __shfl_xor(sp[q], 16, 64); sp[q] += __shfl_xor(sp[q], 32, 64);
        sq[q] += __shfl_xor(sq[q], 16, 64); sq[q] += __shfl_xor(sq[q], 32, 64);
    }
    if (g == 0) {
        float* o = dpq + (size_t)p * 256 + c;
        *reinterpret_cast<tf32x4*>(o) = tf32x4{sp[0], sp[1], sp[2], sp[3]};
        *reinterpret_cast<tf32x4*>(o + 4) = tf32x4{sp[4], sp[5], sp[6], sp[7]};
        *reinterpret_cast<tf32x4*>(o + 128) = tf32x4{sq[0], sq[1], sq[2], sq[3]};
        *reinterpret_cast<tf32x4*>(o + 132) = tf32x4{sq[4], sq[5], sq[6], sq[7]};
    }
}
void te_edge_pq_bwd(const PackInfo& pk, int k, const tb16* dpre1, const int* start, const int* list, float* dpq, hipStream_t s) {
    hipLaunchKernelGGL(k_epq_bwd, dim3((pk.Nmax + 3) / 4), dim3(256), 0, s, pk, k, dpre1, start, list, dpq);
}
__global__ void k_egelu_fwd_out(TRows rows, const float* __restrict__ x, tb16* __restrict__ y, int D, TDrop dr, unsigned site) {
    const size_t n = (size_t)nrows(rows) * D / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const tf32x2 v = reinterpret_cast<const tf32x2*>(x)[i];
        float m0, m1;
        drop_pair(dr, drop_key(dr, site), (unsigned)i, m0, m1);
        reinterpret_cast<unsigned*>(y)[i] = tpack2(gelu_f(v[0]) * m0, gelu_f(v[1]) * m1);
    }
}
__global__ void k_egelu_bwd_in(TRows rows, const tb16* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dx, int D, TDrop dr, unsigned site) {
    const size_t n = (size_t)nrows(rows) * D / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned w = reinterpret_cast<const unsigned*>(dy)[i];
        const tf32x2 p = reinterpret_cast<const tf32x2*>(pre)[i];
        float m0, m1;
        drop_pair(dr, drop_key(dr, site), (unsigned)i, m0, m1);
        reinterpret_cast<tf32x2*>(dx)[i] = tf32x2{tbf_lo(w) * gelu_d(p[0]) * m0, tbf_hi(w) * gelu_d(p[1]) * m1};
    }
}
void te_gelu_fwd_out(const TRows& rows, const float* x, tb16* y, int D, const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_egelu_fwd_out, dim3(ew_grid(rows, D)), dim3(256), 0, s, rows, x, y, D, dr, site);
}
void te_gelu_bwd_in(const TRows& rows, const tb16* dy, const float* pre, float* dx, int D, const TDrop& dr, unsigned site, hipStream_t s) {
    hipLaunchKernelGGL(k_egelu_bwd_in, dim3(ew_grid(rows, D)), dim3(256), 0, s, rows, dy, pre, dx, D, dr, site);
}
