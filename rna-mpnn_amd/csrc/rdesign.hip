// MI355X path of the reference's sibling model `rdesign` (C ABI: include/rdesign_hip.h).
// RNAModel.forward + Readout (rdesign/model/rdesign.py:82-88,104): k-NN graph on the P atoms (self included), 101 node / 115
// edge geometric features (RBF distances, local-frame directions, frame quaternions, backbone dihedrals), Linear + Normalize
// embeddings, L x MPNNLayer (3-Linear message MLP on [h_E | h_V[centre] | h_V[neighbour]], sum over the neighbourhood / 30,
// LayerNorm, dense FFN, LayerNorm), Linear read-out.  Edges are NOT updated in this model, and every valid residue owns a dense
// row of K slots (the reference's edge list is sorted by centre: its scatter_sum is a segmented sum, no atomics).
// Layout: packed residues (row p = cu[b] + t) as in the rnampnn path; the GEMMs are the training path's (kernels_train.hip):
// exact f32 (`t_gemm`, K-major weight copies) or bf16 MFMA with f32 accumulate (`tm_gemm_nt`, weights as stored, GELU fused into the
// operand load).  The first message Linear is factored W.[h_E | h_i | h_j] = W_e.h_E + P[i] + Q[j] like the rnampnn kernels.
// PARITY UNPINNED: see oracle/rdesign_oracle.py (the reference modules cannot be imported here, no fixture ships).
#include "../../include/rdesign_hip.h"
#include "rnampnn_internal.h"
#include "kernels_train.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define RD_H 128
#define RD_NODE 101
#define RD_NODEP 104
#define RD_EDGE 115
#define RD_EDGEP 116
#define RD_KMAX 64

static thread_local char rd_err[512] = "";
static int rd_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(rd_err, sizeof(rd_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* rdesign_last_error(void) { return rd_err; }
#define RD_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return rd_fail(RDESIGN_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); } while (0)

// ------------------------------------------------------------------------------------------ device helpers
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 scale(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
// utils.data.normalize: v / |v|, NaN (0/0) -> 0           (rdesign/utils/data.py:169-171)
__device__ __forceinline__ V3 unit_nan0(V3 a) { const float n = sqrtf(dot(a, a)); return n > 0.f ? scale(a, 1.0f / n) : V3{0.f, 0.f, 0.f}; }
// F.normalize: v / max(|v|, 1e-12)
__device__ __forceinline__ V3 unit_eps(V3 a) { return scale(a, 1.0f / fmaxf(sqrtf(dot(a, a)), 1e-12f)); }
__device__ __forceinline__ float gelu_e(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ tb16 rd_bf(float v) {          // round-to-nearest-even bf16
    typedef __attribute__((ext_vector_type(2))) float f2; typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    f2 t = {v, 0.f};
    return (tb16)(__builtin_bit_cast(unsigned, __builtin_convertvector(t, b2)) & 0xffffu);
}

// atom q of the FLATTENED 6-atom chain of batch row b (feature.py:85-86,138): coordinates of a valid residue, zeros for a padded
// one (the collate zero-fills, utils/data.py:113-115); `inside` = the position exists in the (B, T) tensor at all
__device__ __forceinline__ V3 chain_atom(const float* __restrict__ X, int b, int T, int n, int q, bool& inside) {
    inside = q >= 0 && q < 6 * T;
    const int t = q / 6, a = q - 6 * t;
    if (!inside || t >= n) return V3{0.f, 0.f, 0.f};
    const float* p = X + ((size_t)(b * T + t) * 6 + a) * 3;
    return V3{p[0], p[1], p[2]};
}

// ------------------------------------------------------------------------------------------ per-residue kernel
// packed coordinates [N][18], RNA id, local frame Q [N][9] (rows b_1, n_0, b_1 x n_0; feature.py:88-101) and the 101 raw node
// features: 12 dihedral cos | sin (feature.py:136-155), 5 x 16 RBF of intra-residue distances to P (:196-203), 3 x 3 unit directions of
// P, C5', C4' in the local frame (:131-133).
__global__ void k_rd_residue(const float* __restrict__ X, PackInfo pk, float* __restrict__ coords_p, float* __restrict__ frame,
                             float* __restrict__ node_raw) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= pk.B * pk.T) return;
    const int b = id / pk.T, t = id - b * pk.T, T = pk.T, n = pk.len[b];
    if (t >= n) return;
    const int p = pk.cu[b] + t;
    pk.node_b[p] = b;
    V3 at[6];
    bool in;
#pragma unroll
    for (int a = 0; a < 6; ++a) { at[a] = chain_atom(X, b, T, n, 6 * t + a, in); coords_p[(size_t)p * 18 + 3 * a] = at[a].x; coords_p[(size_t)p * 18 + 3 * a + 1] = at[a].y; coords_p[(size_t)p * 18 + 3 * a + 2] = at[a].z; }
    // local frame at C3' (chain position 6t+4): u_0 = unit(O3' - C3'), u_1 = unit(P(t+1) - O3'); the tensor's last residue has the zero frame
    float Q[9];
    if (t + 1 < T) {
        const V3 pn = chain_atom(X, b, T, n, 6 * t + 6, in);
        const V3 u0 = unit_nan0(at[5] - at[4]), u1 = unit_nan0(pn - at[5]);
        const V3 n0 = unit_nan0(cross(u0, u1)), b1 = unit_nan0(u0 - u1), c = cross(b1, n0);
        Q[0] = b1.x; Q[1] = b1.y; Q[2] = b1.z; Q[3] = n0.x; Q[4] = n0.y; Q[5] = n0.z; Q[6] = c.x; Q[7] = c.y; Q[8] = c.z;
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) Q[i] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) frame[(size_t)p * 9 + i] = Q[i];
    float* f = node_raw + (size_t)p * RD_NODEP;
    // dihedral a of residue t = D[6t + a - 3] of the stride-5 chain (pad (3,4), feature.py:152-153); D = 0 outside
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const int m = 6 * t + a - 3;
        float c = 1.f, sn = 0.f;
        if (m >= 0 && m <= 6 * T - 8) {
            bool i0;
            const V3 x0 = chain_atom(X, b, T, n, m, i0), x1 = chain_atom(X, b, T, n, m + 1, i0), x2 = chain_atom(X, b, T, n, m + 2, i0);
            const V3 x5 = chain_atom(X, b, T, n, m + 5, i0), x6 = chain_atom(X, b, T, n, m + 6, i0), x7 = chain_atom(X, b, T, n, m + 7, i0);
            const V3 u2 = unit_eps(x5 - x0), u1 = unit_eps(x6 - x1), u0 = unit_eps(x7 - x2);
            const V3 n2 = unit_eps(cross(u2, u1)), n1 = unit_eps(cross(u1, u0));
            const float cd = fminf(fmaxf(dot(n2, n1), -1.f + 1e-7f), 1.f - 1e-7f);
            const float sg = dot(u2, n1);
            if (sg != 0.f) { c = cd; sn = (sg > 0.f ? 1.f : -1.f) * sqrtf(fmaxf(1.f - cd * cd, 0.f)); }     // cos / sin of sign * acos(cd)
        }
        f[a] = c; f[6 + a] = sn;
    }
    // RBF of |atom_a - P|, a = O5', C5', C4', C3', O3' (16 Gaussians on [0, 20], sigma 1.25; feature.py:50-56)
#pragma unroll
    for (int a = 1; a < 6; ++a) {
        const V3 d = at[a] - at[0];
        const float dist = sqrtf(dot(d, d) + 1e-6f);
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float z = (dist - (20.0f / 15.0f) * r) / 1.25f; f[12 + 16 * (a - 1) + r] = expf(-z * z); }
    }
    // unit directions of P, C5', C4' seen from C3' in the local frame
    const int inner[3] = {0, 2, 3};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const V3 d = at[inner[i]] - at[4];
        const V3 u = unit_nan0(V3{Q[0] * d.x + Q[1] * d.y + Q[2] * d.z, Q[3] * d.x + Q[4] * d.y + Q[5] * d.z, Q[6] * d.x + Q[7] * d.y + Q[8] * d.z});
        f[92 + 3 * i] = u.x; f[92 + 3 * i + 1] = u.y; f[92 + 3 * i + 2] = u.z;
    }
#pragma unroll
    for (int i = RD_NODE; i < RD_NODEP; ++i) f[i] = 0.f;
}

// ------------------------------------------------------------------------------------------ k-NN on the P atoms
// One wave per residue: the min(K, n) nearest valid residues INCLUDING itself, ascending (distance, index) (feature.py:42-48 +
// mask_attend :186-187).  nbr[p][s] = packed row of the neighbour, -1 for the slots the reference filters out.
__global__ void __launch_bounds__(256) k_rd_knn(const float* __restrict__ coords_p, PackInfo pk, int K, int* __restrict__ nbr,
                                                 int64_t* __restrict__ eidx) {
    extern __shared__ float sm[];                    // P atoms of the RNA [n][3], then one distance row per wave
    const int b = blockIdx.x, n = pk.len[b], T = pk.T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int base = pk.cu[b];
    for (int j = threadIdx.x; j < n; j += 256) { sm[3 * j] = coords_p[(size_t)(base + j) * 18]; sm[3 * j + 1] = coords_p[(size_t)(base + j) * 18 + 1]; sm[3 * j + 2] = coords_p[(size_t)(base + j) * 18 + 2]; }
    __syncthreads();
    float* drow = sm + 3 * T + wave * T;
    for (int i = blockIdx.y * 4 + wave; i < T; i += gridDim.y * 4) {
        if (i >= n) {
            if (eidx) for (int s = lane; s < K; s += 64) eidx[((size_t)b * T + i) * K + s] = -1;
            continue;
        }
        const float cx = sm[3 * i], cy = sm[3 * i + 1], cz = sm[3 * i + 2];
        for (int j = lane; j < n; j += 64) {
            const float dx = __fsub_rn(sm[3 * j], cx), dy = __fsub_rn(sm[3 * j + 1], cy), dz = __fsub_rn(sm[3 * j + 2], cz);
            drow[j] = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)), 1e-6f));
        }
        const int nreal = min(K, n);
        unsigned long long prev = 0ull;
        bool first = true;
        for (int s = 0; s < nreal; ++s) {
            unsigned long long best = ~0ull;
            for (int j = lane; j < n; j += 64) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(drow[j]) << 32) | (unsigned)j;
                if ((first || key > prev) && key < best) best = key;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(best, o, 64); best = other < best ? other : best; }
            prev = best; first = false;
            if (lane == 0) {
                const int j = (int)(best & 0xffffffffu);
                nbr[(size_t)(base + i) * K + s] = base + j;
                if (eidx) eidx[((size_t)b * T + i) * K + s] = j;
            }
        }
        for (int s = nreal + lane; s < K; s += 64) {
            nbr[(size_t)(base + i) * K + s] = -1;
            if (eidx) eidx[((size_t)b * T + i) * K + s] = -1;
        }
    }
}

// ------------------------------------------------------------------------------------------ per-edge features
// 115 raw edge features of edge (p, s): 4 frame quaternion (feature.py:62-81,127-128), 6 x 16 RBF of |atom_a(centre) - P(neighbour)|
// (:58-60, 211-219), 5 x 3 unit directions of the neighbour's P, O5', C5', C4', O3' in the centre's frame (:119-126).  Absent slots: zeros.
__global__ void k_rd_edge(PackInfo pk, int K, const int* __restrict__ nbr, const float* __restrict__ coords_p,
                          const float* __restrict__ frame, float* __restrict__ edge_raw) {
    const size_t E = (size_t)pk.cu[pk.B] * K;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    float* f = edge_raw + e * RD_EDGEP;
    const int j = nbr[e];
    if (j < 0) { for (int i = 0; i < RD_EDGEP; ++i) f[i] = 0.f; return; }
    const int p = (int)(e / K);
    const float* ci = coords_p + (size_t)p * 18;
    const float* cj = coords_p + (size_t)j * 18;
    const float* Qi = frame + (size_t)p * 9;
    const float* Qj = frame + (size_t)j * 9;
    // R = Q_i^T Q_j
    float R[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) R[a][c] = Qi[a] * Qj[c] + Qi[3 + a] * Qj[3 + c] + Qi[6 + a] * Qj[6 + c];
    const float xx = R[0][0], yy = R[1][1], zz = R[2][2];
    float q[4];
    const float m0 = 0.5f * sqrtf(fabsf(1.f + xx - yy - zz)), m1 = 0.5f * sqrtf(fabsf(1.f - xx + yy - zz)), m2 = 0.5f * sqrtf(fabsf(1.f - xx - yy + zz));
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    q[0] = sgn(R[2][1] - R[1][2]) * m0; q[1] = sgn(R[0][2] - R[2][0]) * m1; q[2] = sgn(R[1][0] - R[0][1]) * m2;
    q[3] = sqrtf(fmaxf(1.f + xx + yy + zz, 0.f)) * 0.5f;
    const float qn = 1.0f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = q[i] * qn;
    const V3 pj = v3(cj[0], cj[1], cj[2]);
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const V3 d = v3(ci[3 * a], ci[3 * a + 1], ci[3 * a + 2]) - pj;
        const float dist = sqrtf(dot(d, d) + 1e-6f);
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float z = (dist - (20.0f / 15.0f) * r) / 1.25f; f[4 + 16 * a + r] = expf(-z * z); }
    }
    const V3 c3 = v3(ci[12], ci[13], ci[14]);
    const int atoms[5] = {0, 1, 2, 3, 5};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const V3 d = v3(cj[3 * atoms[i]], cj[3 * atoms[i] + 1], cj[3 * atoms[i] + 2]) - c3;
        const V3 u = unit_nan0(V3{Qi[0] * d.x + Qi[1] * d.y + Qi[2] * d.z, Qi[3] * d.x + Qi[4] * d.y + Qi[5] * d.z, Qi[6] * d.x + Qi[7] * d.y + Qi[8] * d.z});
        f[100 + 3 * i] = u.x; f[100 + 3 * i + 1] = u.y; f[100 + 3 * i + 2] = u.z;
    }
    f[RD_EDGE] = 0.f;
}

// ------------------------------------------------------------------------------------------ row normalisations (one wave per row, D = 128)
// mode 0: functional.Normalize (functional.py:83-101): unbiased variance, gain (x - mu) / (sqrt(var + eps) + eps) + bias, eps 1e-6
// mode 1: nn.LayerNorm(x + r): biased variance, eps 1e-5
__global__ void __launch_bounds__(256) k_rd_rownorm(const int* __restrict__ ntot_p, int mul, const float* __restrict__ x, const float* __restrict__ res,
                                                    const float* __restrict__ gain, const float* __restrict__ bias, int mode, float* __restrict__ y,
                                                    tb16* __restrict__ yb) {
    const size_t R = (size_t)*ntot_p * mul;
    const int lane = threadIdx.x & 63;
    for (size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < R; row += (size_t)gridDim.x * 4) {
        float v0 = x[row * RD_H + lane], v1 = x[row * RD_H + 64 + lane];
        if (res) { v0 += res[row * RD_H + lane]; v1 += res[row * RD_H + 64 + lane]; }
        float s = v0 + v1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mu = s / 128.f;
        const float d0 = v0 - mu, d1 = v1 - mu;
        float q = d0 * d0 + d1 * d1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        float inv;
        if (mode == 0) inv = 1.0f / (sqrtf(q / 127.f + 1e-6f) + 1e-6f);
        else inv = 1.0f / sqrtf(q / 128.f + 1e-5f);
        const float o0 = gain[lane] * d0 * inv + bias[lane], o1 = gain[64 + lane] * d1 * inv + bias[64 + lane];
        if (yb) { yb[row * RD_H + lane] = rd_bf(o0); yb[row * RD_H + 64 + lane] = rd_bf(o1); }      // bf16 edge tensors of the bf16 path
        else { y[row * RD_H + lane] = o0; y[row * RD_H + 64 + lane] = o1; }
    }
}
static void rd_rownorm(const int* ntot, int mul, size_t maxrows, const float* x, const float* res, const float* gain, const float* bias, int mode,
                       float* y, hipStream_t s, tb16* yb = nullptr) {
    size_t g = (maxrows + 3) / 4;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_rd_rownorm, dim3((unsigned)g), dim3(256), 0, s, ntot, mul, x, res, gain, bias, mode, y, yb);
}

// dh[p][c] = sum over the valid slots of GELU(pre[(p, s)][c]) / scale      (mpnn.py:32-33: scatter_sum / 30, a segmented sum here)
__global__ void __launch_bounds__(128) k_rd_segsum(PackInfo pk, int K, const int* __restrict__ nbr, const float* __restrict__ pre, float inv_scale,
                                                   float* __restrict__ out) {
    const int p = blockIdx.x;
    if (p >= pk.cu[pk.B]) return;
    const int c = threadIdx.x;
    float s = 0.f;
    for (int sl = 0; sl < K; ++sl)
        if (nbr[(size_t)p * K + sl] >= 0) s += gelu_e(pre[((size_t)p * K + sl) * RD_H + c]);
    out[(size_t)p * RD_H + c] = s * inv_scale;
}

// the same on a bf16 pre-activation tensor (bf16 path): one thread = two adjacent channels
__global__ void __launch_bounds__(256) k_rd_segsum_b(PackInfo pk, int K, const int* __restrict__ nbr, const tb16* __restrict__ pre, float inv_scale,
                                                     float* __restrict__ out) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= pk.cu[pk.B]) return;
    const int lane = threadIdx.x & 63, c = 2 * lane;
    // validity of the K slots in one load + ballot (K <= 64), then unconditional row loads in batches
    const unsigned long long vm = __ballot(lane < K && nbr[(size_t)p * K + (lane < K ? lane : 0)] >= 0);
    const tb16* base = pre + (size_t)p * K * RD_H + c;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll 5
    for (int sl = 0; sl < K; ++sl) {
        const unsigned w = *reinterpret_cast<const unsigned*>(base + (size_t)sl * RD_H);
        const bool valid = (vm >> sl) & 1ull;
        s0 += valid ? gelu_e(__uint_as_float(w << 16)) : 0.f;
        s1 += valid ? gelu_e(__uint_as_float(w & 0xffff0000u)) : 0.f;
    }
    out[(size_t)p * RD_H + c] = s0 * inv_scale; out[(size_t)p * RD_H + c + 1] = s1 * inv_scale;
}

// ------------------------------------------------------------------------------------------ handle
struct RdT { std::string key; int64_t numel; size_t off; };
struct RdLin { int in, out, w, b; size_t wt; };          // wt: K-major f32 copy [in_pad][out] (f32 path, and K % 16 != 0 shapes)
struct RdLayer { int n1w, n1b, n2w, n2b; std::vector<RdLin> msg, dense; };
struct rdesign_ctx {
    RDesignConfig cfg;
    std::vector<RdT> raw;
    size_t raw_floats = 0, der_floats = 0;
    float* arena = nullptr;          // caller's flat parameter buffer
    float* der = nullptr;
    bool finalized = false;
    WImageCache* wimg = nullptr;     // prebuilt bf16 fragment images of the 128 x 128 weight blocks (bf16 path; kernels_train.h)
    bool wimg_fresh = false;         // images match the weights (reset by finalize)
    RdLin node_emb, edge_emb;
    int nn_g, nn_b, ne_g, ne_b;
    std::vector<RdLayer> layers;
    std::vector<RdLin> readout;
};
static int rd_add(rdesign_ctx* c, const std::string& key, int64_t numel) {
    c->raw.push_back(RdT{key, numel, c->raw_floats});
    c->raw_floats += (size_t)((numel + 3) / 4 * 4);
    return (int)c->raw.size() - 1;
}
static RdLin rd_lin(rdesign_ctx* c, const std::string& prefix, int in, int out) {
    RdLin l;
    l.in = in; l.out = out;
    l.w = rd_add(c, prefix + ".weight", (int64_t)in * out);
    l.b = rd_add(c, prefix + ".bias", out);
    l.wt = c->der_floats;
    c->der_floats += (size_t)((in + 3) / 4 * 4) * out;
    return l;
}
static inline float* rdp(rdesign_ctx* c, int i) { return c->arena + c->raw[i].off; }

extern "C" int rdesign_create(const RDesignConfig* cfg, rdesign_handle* out) {
    if (!cfg || !out) return rd_fail(RDESIGN_ERR_BAD_ARG, "rdesign_create: null argument");
    const RDesignConfig& g = *cfg;
    if (g.hidden_dim != RD_H) return rd_fail(RDESIGN_ERR_UNSUPPORTED, "hidden_dim must be 128 (got %d)", g.hidden_dim);
    if (g.k_neighbors < 1 || g.k_neighbors > RD_KMAX) return rd_fail(RDESIGN_ERR_UNSUPPORTED, "k_neighbors must be in [1, %d]", RD_KMAX);
    if (g.num_message_layers < 1 || g.num_message_layers > 8 || g.num_dense_layers < 1 || g.num_dense_layers > 8 || g.num_mpnn_layers < 1)
        return rd_fail(RDESIGN_ERR_UNSUPPORTED, "layer counts out of range");
    if (g.dim_dense_layers < 16 || g.dim_dense_layers % 16 || g.dim_dense_layers > 2048 || (g.num_readout_layers > 1 && (g.readout_hidden_dim % 16 || g.readout_hidden_dim < 16)))
        return rd_fail(RDESIGN_ERR_UNSUPPORTED, "dense / read-out widths must be multiples of 16");
    if (g.precision != RDESIGN_PREC_F32 && g.precision != RDESIGN_PREC_BF16) return rd_fail(RDESIGN_ERR_BAD_ARG, "unknown precision");
    rdesign_ctx* c = new rdesign_ctx();
    c->cfg = g;
    // registration order = torch's state_dict order of RNAModel (rdesign.py:52-64)
    c->node_emb = rd_lin(c, "features.node_embedding", RD_NODE, RD_H);
    c->edge_emb = rd_lin(c, "features.edge_embedding", RD_EDGE, RD_H);
    c->nn_g = rd_add(c, "features.norm_nodes.gain", RD_H); c->nn_b = rd_add(c, "features.norm_nodes.bias", RD_H);
    c->ne_g = rd_add(c, "features.norm_edges.gain", RD_H); c->ne_b = rd_add(c, "features.norm_edges.bias", RD_H);
    for (int l = 0; l < g.num_mpnn_layers; ++l) {
        RdLayer L;
        const std::string p = "mpnn_layers." + std::to_string(l);
        L.n1w = rd_add(c, p + ".norm1.weight", RD_H); L.n1b = rd_add(c, p + ".norm1.bias", RD_H);
        L.n2w = rd_add(c, p + ".norm2.weight", RD_H); L.n2b = rd_add(c, p + ".norm2.bias", RD_H);
        int in = 3 * RD_H;
        for (int i = 0; i < g.num_message_layers; ++i) { L.msg.push_back(rd_lin(c, p + ".message_layers." + std::to_string(3 * i), in, RD_H)); in = RD_H; }
        for (int i = 0; i < g.num_dense_layers; ++i) { L.dense.push_back(rd_lin(c, p + ".dense." + std::to_string(3 * i), in, g.dim_dense_layers)); in = g.dim_dense_layers; }
        L.dense.push_back(rd_lin(c, p + ".dense." + std::to_string(3 * g.num_dense_layers), in, RD_H));
        c->layers.push_back(L);
    }
    int in = RD_H;
    for (int i = 0; i + 1 < g.num_readout_layers; ++i) { c->readout.push_back(rd_lin(c, "readout.readout_layers." + std::to_string(3 * i), in, g.readout_hidden_dim)); in = g.readout_hidden_dim; }
    c->readout.push_back(rd_lin(c, "readout.readout_layers." + std::to_string(3 * (g.num_readout_layers > 1 ? g.num_readout_layers - 1 : 0)), in, 4));
    *out = c;
    return RDESIGN_OK;
}
extern "C" int rdesign_destroy(rdesign_handle h) {
    if (!h) return RDESIGN_OK;
    if (h->der) (void)hipFree(h->der);
    t_wimg_destroy(h->wimg);
    delete h;
    return RDESIGN_OK;
}
extern "C" int rdesign_num_weights(rdesign_handle h) { return h ? (int)h->raw.size() : 0; }
extern "C" int64_t rdesign_param_numel(rdesign_handle h) { return h ? (int64_t)h->raw_floats : 0; }
extern "C" int rdesign_weight_info(rdesign_handle h, int32_t i, const char** key, int64_t* numel, int64_t* offset) {
    if (!h || i < 0 || i >= (int)h->raw.size()) return rd_fail(RDESIGN_ERR_BAD_ARG, "weight index out of range");
    if (key) *key = h->raw[i].key.c_str();
    if (numel) *numel = h->raw[i].numel;
    if (offset) *offset = (int64_t)h->raw[i].off;
    return RDESIGN_OK;
}
extern "C" int rdesign_use_weight_arena(rdesign_handle h, float* arena, void* stream) {
    if (!h || !arena || ((uintptr_t)arena & 15)) return rd_fail(RDESIGN_ERR_BAD_ARG, "rdesign_use_weight_arena: null or unaligned arena");
    h->arena = arena;
    t_wimg_clear(h->wimg);
    if (!h->der) {
        RD_TRY(hipMalloc((void**)&h->der, h->der_floats * sizeof(float)));
        RD_TRY(hipMemsetAsync(h->der, 0, h->der_floats * sizeof(float), (hipStream_t)stream));
    }
    h->finalized = false;
    return RDESIGN_OK;
}
extern "C" int rdesign_finalize_weights(rdesign_handle h, void* stream) {
    if (!h || !h->arena) return rd_fail(RDESIGN_ERR_WEIGHTS, "no weight arena set");
    hipStream_t s = (hipStream_t)stream;
    auto tr = [&](const RdLin& l) { launch_transpose(rdp(h, l.w), l.in, l.out, l.in, h->der + l.wt, l.out, s); };
    tr(h->node_emb); tr(h->edge_emb);
    for (auto& L : h->layers) { for (auto& l : L.msg) tr(l); for (auto& l : L.dense) tr(l); }
    for (auto& l : h->readout) tr(l);
    RD_TRY(hipGetLastError());
    h->finalized = true;
    h->wimg_fresh = false;
    return RDESIGN_OK;
}

// ------------------------------------------------------------------------------------------ forward
struct RdWs {
    int *len, *cu, *node_b, *nbr;
    float *coords_p, *frame, *node_raw, *edge_raw, *hV, *hV2, *hE, *E1, *E2, *pq, *dh, *dA, *dB, *logits;
    size_t total;
};
static size_t rd_carve(const rdesign_ctx* c, int B, size_t Nmax, char* base, RdWs* w, bool edges = true) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return base ? base + o : (char*)nullptr; };
    const size_t K = c->cfg.k_neighbors, E = edges ? Nmax * K : 0, D = c->cfg.dim_dense_layers > c->cfg.readout_hidden_dim ? c->cfg.dim_dense_layers : c->cfg.readout_hidden_dim;
    RdWs tmp;
    RdWs& r = w ? *w : tmp;
    r.len = (int*)take(B * sizeof(int)); r.cu = (int*)take((B + 1) * sizeof(int)); r.node_b = (int*)take(Nmax * sizeof(int));
    r.nbr = (int*)take(E * sizeof(int));
    r.coords_p = (float*)take(Nmax * 18 * sizeof(float)); r.frame = (float*)take(Nmax * 9 * sizeof(float));
    r.node_raw = (float*)take(Nmax * RD_NODEP * sizeof(float)); r.edge_raw = (float*)take(E * RD_EDGEP * sizeof(float));
    r.hV = (float*)take((Nmax + 1) * RD_H * sizeof(float)); r.hV2 = (float*)take((Nmax + 1) * RD_H * sizeof(float));
    r.hE = (float*)take(E * RD_H * sizeof(float)); r.E1 = (float*)take(E * RD_H * sizeof(float)); r.E2 = (float*)take(E * RD_H * sizeof(float));
    r.pq = (float*)take((Nmax + 1) * 256 * sizeof(float)); r.dh = (float*)take(Nmax * RD_H * sizeof(float));
    r.dA = (float*)take(Nmax * D * sizeof(float)); r.dB = (float*)take(Nmax * D * sizeof(float));
    r.logits = (float*)take(Nmax * 4 * sizeof(float));
    r.total = off;
    return off;
}
extern "C" size_t rdesign_readout_workspace_bytes(rdesign_handle h, int32_t n_rows) {
    if (!h || n_rows <= 0) return 0;
    return rd_carve(h, 1, (size_t)n_rows, nullptr, nullptr, false);
}
extern "C" size_t rdesign_workspace_bytes(rdesign_handle h, int32_t B, int32_t T) {
    if (!h || B <= 0 || T <= 0) return 0;
    return rd_carve(h, B, (size_t)B * T, nullptr, nullptr);
}

namespace {
struct RdRun { rdesign_ctx* c; PackInfo pk; RdWs w; hipStream_t s; bool mixed; TDrop nodrop; int K; bool bad = false;
    TRows rn() const { return TRows{pk.cu + pk.B, 1, pk.Nmax}; }
    TRows re() const { return TRows{pk.cu + pk.B, K, pk.Nmax * K}; } };
// Y = [beta Y] + act(X)[:, 0:Kc] . W[:, k0:k0+Kc]^T + bias        (act = GELU of the stored pre-activation when `gelu_in`)
void rd_mm(RdRun& r, const TRows& rows, const float* X, int ldx, const RdLin& l, int k0, int Kc, bool use_bias, float* Y, int ldy, int beta,
           bool gelu_in, float* scratch) {
    rdesign_ctx* c = r.c;
    const float* bias = use_bias ? rdp(c, l.b) : nullptr;
    if (r.mixed && tm_gemm_nt(rows, X, ldx, Kc, rdp(c, l.w) + k0, l.in, bias, l.out, Y, ldy, beta, gelu_in, r.nodrop, 0u, r.s)) return;
    const float* xin = X;
    if (gelu_in) { t_gelu_fwd(rows, X, scratch, ldx, r.nodrop, 0u, r.s); xin = scratch; }      // (ldx == width of the activation here)
    const int Kp = (Kc + 3) / 4 * 4;
    t_gemm(rows, xin, ldx, Kp, c->der + l.wt + (size_t)k0 * l.out, l.out, bias, l.out, Y, ldy, beta, r.s);
}
}  // namespace

extern "C" int rdesign_forward(rdesign_handle h, const float* X, const float* mask, int32_t B, int32_t T, float* h_V, float* logits,
                               int64_t* edge_index, float* node_raw, float* edge_raw, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !X || !mask || !ws || B <= 0 || T <= 0) return rd_fail(RDESIGN_ERR_BAD_ARG, "rdesign_forward: null pointer or non-positive B/T");
    if (!h->arena) return rd_fail(RDESIGN_ERR_WEIGHTS, "no weight arena set");
    if (!h->finalized) return rd_fail(RDESIGN_ERR_WEIGHTS, "weights not finalized (call rdesign_finalize_weights)");
    rdesign_ctx* c = h;
    const size_t Nmax = (size_t)B * T;
    const int K = c->cfg.k_neighbors;
    if ((long long)Nmax * K > 0x3fffffffLL / 4) return rd_fail(RDESIGN_ERR_BAD_ARG, "row count out of range for 32-bit edge indexing");
    if (ws_bytes < rd_carve(c, B, Nmax, nullptr, nullptr)) return rd_fail(RDESIGN_ERR_WORKSPACE, "workspace too small");
    if (((uintptr_t)ws & 255) != 0) return rd_fail(RDESIGN_ERR_BAD_ARG, "workspace must be 256-byte aligned");
    const size_t knn_lds = (size_t)(3 + 4) * T * sizeof(float);
    if (knn_lds > 160 * 1024 - 256) return rd_fail(RDESIGN_ERR_UNSUPPORTED, "max_len %d too long for the LDS-resident k-NN row", T);
    RdRun r;
    r.c = c; r.s = (hipStream_t)stream; r.mixed = c->cfg.precision == RDESIGN_PREC_BF16; r.nodrop = TDrop{0ull, 0u, 1.f}; r.K = K;
    rd_carve(c, B, Nmax, (char*)ws, &r.w);
    RdWs& w = r.w;
    hipStream_t s = r.s;
    r.pk.len = w.len; r.pk.cu = w.cu; r.pk.node_b = w.node_b; r.pk.B = B; r.pk.T = T; r.pk.Nmax = (int)Nmax; r.pk.packed_in = 0;
    if (r.mixed) {      // weights are static between finalize calls: the images are rebuilt once, blocks first seen in this call build their own
        if (!c->wimg) c->wimg = t_wimg_create(256);
        if (c->wimg && (!c->wimg_fresh || t_wimg_pending(c->wimg) > 0)) { t_wimg_refresh(c->wimg, s); c->wimg_fresh = true; }
        t_wimg_bind(c->wimg);
    }
    launch_lengths(mask, r.pk, s);
    ZeroRegions z{};
    z.ptr[0] = w.hV + Nmax * RD_H; z.words[0] = RD_H; z.ptr[1] = w.hV2 + Nmax * RD_H; z.words[1] = RD_H;
    z.ptr[2] = w.pq + Nmax * 256; z.words[2] = 256; z.n = 3;
    if (r.mixed) { z.ptr[3] = reinterpret_cast<tb16*>(w.pq) + (2 * Nmax + 1) * RD_H; z.words[3] = RD_H / 2; z.n = 4; }     // row Nmax of the bf16 Q table
    launch_zero_regions(z, s);
    // ---- RNAFeatures.forward (feature.py:157-248)
    hipLaunchKernelGGL(k_rd_residue, dim3((unsigned)((Nmax + 63) / 64)), dim3(64), 0, s, X, r.pk, w.coords_p, w.frame, w.node_raw);
    {
        static DevAttr attr;
        ensure_dyn_lds((const void*)k_rd_knn, knn_lds, attr);
        dim3 grid(B, (T + 31) / 32 > 64 ? 64 : (T + 31) / 32);
        hipLaunchKernelGGL(k_rd_knn, grid, dim3(256), knn_lds, s, w.coords_p, r.pk, K, w.nbr, edge_index);
    }
    hipLaunchKernelGGL(k_rd_edge, dim3((unsigned)((Nmax * K + 127) / 128)), dim3(128), 0, s, r.pk, K, w.nbr, w.coords_p, w.frame, w.edge_raw);
    if (node_raw) RD_TRY(hipMemcpy2DAsync(node_raw, RD_NODE * sizeof(float), w.node_raw, RD_NODEP * sizeof(float), RD_NODE * sizeof(float), Nmax, hipMemcpyDeviceToDevice, s));
    if (edge_raw) RD_TRY(hipMemcpy2DAsync(edge_raw, RD_EDGE * sizeof(float), w.edge_raw, RD_EDGEP * sizeof(float), RD_EDGE * sizeof(float), Nmax * K, hipMemcpyDeviceToDevice, s));
    // embeddings: Linear (101 / 115 inputs: exact-f32 GEMM on the K-major copy in both precisions) + Normalize
    t_gemm(r.rn(), w.node_raw, RD_NODEP, RD_NODEP, c->der + c->node_emb.wt, RD_H, rdp(c, c->node_emb.b), RD_H, w.hV2, RD_H, 0, s);
    rd_rownorm(r.pk.cu + B, 1, Nmax, w.hV2, nullptr, rdp(c, c->nn_g), rdp(c, c->nn_b), 0, w.hV, s);
    t_gemm(r.re(), w.edge_raw, RD_EDGEP, RD_EDGEP, c->der + c->edge_emb.wt, RD_H, rdp(c, c->edge_emb.b), RD_H, w.E1, RD_H, 0, s);
    rd_rownorm(r.pk.cu + B, K, Nmax * K, w.E1, nullptr, rdp(c, c->ne_g), rdp(c, c->ne_b), 0, w.hE, s, r.mixed ? reinterpret_cast<tb16*>(w.hE) : nullptr);
    // ---- L x MPNNLayer (rdesign.py:84-86, mpnn.py:31-37)
    for (auto& L : c->layers) {
        // message Linear 0 on cat[h_E, h_V[centre], h_V[neighbour]] = W_e.h_E + P[centre] + Q[neighbour]
        if (r.mixed) {     // bf16 edge tensors, the edge GEMM of the bf16-mixed trainer (kernels_train.h: te_gemm): P + Q in the epilogue, GELU in the operand load
            tb16* Pt = reinterpret_cast<tb16*>(w.pq);
            tb16* Qt = Pt + (Nmax + 1) * RD_H;
            const float* w0 = rdp(c, L.msg[0].w);                                                            // [128][384] = [W_e | W_centre | W_neighbour]
            r.bad |= !te_gemm(r.rn(), w.hV, false, RD_H, w0 + RD_H, 3 * RD_H, true, rdp(c, L.msg[0].b), Pt, 0, false, nullptr, nullptr, r.nodrop, 0u, s);
            r.bad |= !te_gemm(r.rn(), w.hV, false, RD_H, w0 + 2 * RD_H, 3 * RD_H, true, nullptr, Qt, 0, false, nullptr, nullptr, r.nodrop, 0u, s);
            EFuse f{Pt, Qt, w.nbr, K, (int)Nmax, nullptr, nullptr, 0u};
            tb16* cur = reinterpret_cast<tb16*>(w.E1);
            tb16* nxt = reinterpret_cast<tb16*>(w.E2);
            size_t first = 1;
            if (L.msg.size() >= 2) {      // Linears 0 and 1 in one kernel: the hidden activation stays in registers, only the second pre-activation is written
                te_mlp2_fwd(r.re(), reinterpret_cast<const tb16*>(w.hE), w0, 3 * RD_H, rdp(c, L.msg[1].w), RD_H, rdp(c, L.msg[1].b), nullptr, cur, f, r.nodrop, 0u, s);
                first = 2;
            } else {
                r.bad |= !te_gemm(r.re(), w.hE, true, RD_H, w0, 3 * RD_H, true, nullptr, cur, 0, false, nullptr, &f, r.nodrop, 0u, s);
            }
            for (size_t i = first; i < L.msg.size(); ++i) {
                r.bad |= !te_gemm(r.re(), cur, true, RD_H, rdp(c, L.msg[i].w), RD_H, true, rdp(c, L.msg[i].b), nxt, 0, true, nullptr, nullptr, r.nodrop, 0u, s);
                tb16* t = cur; cur = nxt; nxt = t;
            }
            hipLaunchKernelGGL(k_rd_segsum_b, dim3((unsigned)((Nmax + 3) / 4)), dim3(256), 0, s, r.pk, K, w.nbr, cur, 1.0f / 30.0f, w.dh);
        } else {
        rd_mm(r, r.rn(), w.hV, RD_H, L.msg[0], RD_H, RD_H, true, w.pq, 256, 0, false, nullptr);            // P = h_V W_c^T + b
        rd_mm(r, r.rn(), w.hV, RD_H, L.msg[0], 2 * RD_H, RD_H, false, w.pq + RD_H, 256, 0, false, nullptr); // Q = h_V W_n^T
        rd_mm(r, r.re(), w.hE, RD_H, L.msg[0], 0, RD_H, false, w.E1, RD_H, 0, false, nullptr);
        t_edge_add_pq(r.pk, K, w.nbr, w.pq, w.E1, s);
        float* cur = w.E1;
        float* nxt = w.E2;
        for (size_t i = 1; i < L.msg.size(); ++i) {                                                         // GELU of the previous Linear fused into / before this one
            rd_mm(r, r.re(), cur, RD_H, L.msg[i], 0, RD_H, true, nxt, RD_H, 0, true, cur == w.E1 ? w.E1 : w.E2);
            float* t = cur; cur = nxt; nxt = t;
        }
        hipLaunchKernelGGL(k_rd_segsum, dim3((unsigned)Nmax), dim3(128), 0, s, r.pk, K, w.nbr, cur, 1.0f / 30.0f, w.dh);
        }
        rd_rownorm(r.pk.cu + B, 1, Nmax, w.hV, w.dh, rdp(c, L.n1w), rdp(c, L.n1b), 1, w.hV2, s);            // norm1(h_V + dh)
        // dense FFN
        const float* x = w.hV2;
        int ld = RD_H;
        float* bufs[2] = {w.dA, w.dB};
        for (size_t i = 0; i < L.dense.size(); ++i) {
            const bool last = i + 1 == L.dense.size();
            float* dst = last ? w.dh : bufs[i & 1];
            rd_mm(r, r.rn(), x, ld, L.dense[i], 0, L.dense[i].in, true, dst, L.dense[i].out, 0, i > 0, const_cast<float*>(x));
            x = dst; ld = L.dense[i].out;
        }
        rd_rownorm(r.pk.cu + B, 1, Nmax, w.hV2, w.dh, rdp(c, L.n2w), rdp(c, L.n2b), 1, w.hV, s);            // norm2(h_V + dense(h_V))
    }
    // ---- Readout (functional.py:103-126)
    {
        const float* x = w.hV;
        int ld = RD_H;
        float* bufs[2] = {w.dA, w.dB};
        for (size_t i = 0; i < c->readout.size(); ++i) {
            const bool last = i + 1 == c->readout.size();
            float* dst = last ? w.logits : bufs[i & 1];
            rd_mm(r, r.rn(), x, ld, c->readout[i], 0, c->readout[i].in, true, dst, c->readout[i].out, 0, i > 0, const_cast<float*>(x));
            x = dst; ld = c->readout[i].out;
        }
    }
    t_wimg_bind(nullptr);
    if (r.bad) return rd_fail(RDESIGN_ERR_UNSUPPORTED, "bf16 path: a GEMM variant this configuration needs is not built");
    if (h_V) RD_TRY(hipMemcpyAsync(h_V, w.hV, Nmax * RD_H * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (logits) RD_TRY(hipMemcpyAsync(logits, w.logits, Nmax * 4 * sizeof(float), hipMemcpyDeviceToDevice, s));
    RD_TRY(hipGetLastError());
    return RDESIGN_OK;
}

// Readout.forward (functional.py:123-126) on caller rows: logits[n][4] = readout_layers(h_V[n][128])
__global__ void k_rd_seti(int* p, int v) { *p = v; }
extern "C" int rdesign_readout(rdesign_handle h, const float* h_V, int32_t n_rows, float* logits, void* ws, size_t ws_bytes, void* stream) {
    if (!h || !h_V || !logits || !ws || n_rows <= 0) return rd_fail(RDESIGN_ERR_BAD_ARG, "rdesign_readout: null pointer or non-positive row count");
    if (!h->arena || !h->finalized) return rd_fail(RDESIGN_ERR_WEIGHTS, "weights not set / not finalized");
    if (ws_bytes < rd_carve(h, 1, (size_t)n_rows, nullptr, nullptr, false)) return rd_fail(RDESIGN_ERR_WORKSPACE, "workspace too small");
    if (((uintptr_t)ws & 255) != 0) return rd_fail(RDESIGN_ERR_BAD_ARG, "workspace must be 256-byte aligned");
    RdRun r;
    r.c = h; r.s = (hipStream_t)stream; r.mixed = h->cfg.precision == RDESIGN_PREC_BF16; r.nodrop = TDrop{0ull, 0u, 1.f}; r.K = h->cfg.k_neighbors;
    rd_carve(h, 1, (size_t)n_rows, (char*)ws, &r.w, false);
    r.pk.len = r.w.len; r.pk.cu = r.w.cu; r.pk.node_b = r.w.node_b; r.pk.B = 1; r.pk.T = n_rows; r.pk.Nmax = n_rows; r.pk.packed_in = 0;
    hipLaunchKernelGGL(k_rd_seti, dim3(1), dim3(1), 0, r.s, r.w.cu + 1, (int)n_rows);
    const float* x = h_V;
    int ld = RD_H;
    float* bufs[2] = {r.w.dA, r.w.dB};
    for (size_t i = 0; i < h->readout.size(); ++i) {
        const bool last = i + 1 == h->readout.size();
        float* dst = last ? logits : bufs[i & 1];
        rd_mm(r, r.rn(), x, ld, h->readout[i], 0, h->readout[i].in, true, dst, h->readout[i].out, 0, i > 0, const_cast<float*>(x));
        x = dst; ld = h->readout[i].out;
    }
    RD_TRY(hipGetLastError());
    return RDESIGN_OK;
}
