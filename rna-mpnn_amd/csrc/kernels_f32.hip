// Exact-f32 kernels of the RNA-MPNN forward path for gfx950 (wave64).
// These are the parity-grade kernels (|dlogit| <= 1e-4 vs the oracle): plain f32 FMA
// arithmetic, LDS-staged operand tiles, one 256-thread workgroup per 32-row tile.
// The bf16/MFMA kernels in kernels_bf16.hip replace the GEMM-shaped ones on the fast path;
// the graph / geometry / normalisation / decode kernels here serve both precisions.
// Reference lines restated by each kernel are cited at its head.
#include "rnampnn_internal.h"
#include <cstdlib>

#define WAVE 64
static constexpr float kLEPS = 1.0e6f;
static constexpr float kSEPS = 1.0e-6f;

__device__ __forceinline__ float gelu_erf(float x) {           // nn.GELU() default (erf form)
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// ------------------------------------------------------------------------------------------
// lengths + prefix sum of the prefix masks produced by the reference collate
// (rnampnn/utils/data.py:128-139).
__global__ void k_lengths(const float* __restrict__ mask, int B, int T, int* __restrict__ len) {
    int b = blockIdx.x;
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += WAVE) s += mask[(size_t)b * T + t];
    s = wave_sum(s);
    if (threadIdx.x == 0) len[b] = (int)(s + 0.5f);
}

// exclusive prefix sum of one int per thread over a 1024-thread workgroup (wave shuffles + one LDS hop); *total = the grand sum
// (a serial loop of thread 0 over 1024 LDS words took 10 us of every forward)
__device__ __forceinline__ int block_exclusive_scan_1024(int v, int* wave_tot /* LDS [16] */, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int t = wave_tot[w]; if (w < wave) base += t; tot += t; }
    *total = tot;
    return base + inc - v;
}
__global__ void __launch_bounds__(1024) k_scan(const int* __restrict__ len, int B, int* __restrict__ cu) {
    __shared__ int wave_tot[16];
    int tid = threadIdx.x;
    int chunk = (B + 1023) / 1024;
    int lo = min(B, tid * chunk), hi = min(B, lo + chunk);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += len[i];
    int total;
    int run = block_exclusive_scan_1024(s, wave_tot, &total);
    if (tid == 0) cu[B] = total;
    for (int i = lo; i < hi; ++i) { cu[i] = run; run += len[i]; }
}

// Packed input: lengths and the internal prefix sum from the caller's cu_seqlens.  The caller's contract is
// 0 <= cu[b+1] - cu[b] <= T_max and cu[B] - cu[0] == N_total; a violation must not become an out-of-bounds access
// (k_knn sizes LDS rows by T_max, the workspace is carved for N_total rows), so lengths are clamped to [0, T_max] and the
// running total to N_total - results for an out-of-contract input are meaningless, but memory-safe.
__global__ void __launch_bounds__(1024) k_lengths_from_cu(const int32_t* __restrict__ cu_in, int B, int T, int Nmax,
                                                          int* __restrict__ len, int* __restrict__ cu) {
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x;
    const int chunk = (B + 1023) / 1024;
    const int lo = min(B, tid * chunk), hi = min(B, lo + chunk);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += min(max(cu_in[i + 1] - cu_in[i], 0), T);
    int total;
    int run = block_exclusive_scan_1024(s, wave_tot, &total);
    if (tid == 0) cu[B] = min(total, Nmax);
    for (int i = lo; i < hi; ++i) {
        const int l = min(max(cu_in[i + 1] - cu_in[i], 0), T);
        const int c0 = min(run, Nmax);
        cu[i] = c0;
        len[i] = min(l, Nmax - c0);
        run += l;
    }
}
// zero up to 8 small regions (the all-zero gather rows of the node tables) in one launch
__global__ void k_zero_regions(ZeroRegions z) {
    const int g = blockIdx.x;
    if (g >= z.n) return;
    unsigned* p = reinterpret_cast<unsigned*>(z.ptr[g]);
    for (unsigned i = threadIdx.x; i < z.words[g]; i += blockDim.x) p[i] = 0u;
}
// Plain kernels instead of hipMemsetAsync / hipMemcpyAsync on the training path: inside a captured hipGraph the runtime's memset / copy
// NODES of tens of megabytes misbehaved on this stack (NaN losses at 16 RNAs, memory faults at 64), kernel nodes do not.
__global__ void k_zero_bytes(uint4* __restrict__ p, size_t n16, unsigned char* __restrict__ tail, int ntail) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
// RNAMPNN_DBG_MEMNODE (diagnostic, read per call): bit 0 routes launch_zero_bytes through hipMemsetAsync, bit 1 launch_copy_bytes through
// hipMemcpyAsync - the forms round 3 replaced - so that the captured training step can be compared with and without runtime memset / copy nodes
// RNAMPNN_DBG_MEMNODE_SITES: bit mask of the launch_zero_bytes call sites (1 row zeroes of the taped forward, 2 flat gradient, 4 dE, 8 reverse-
// adjacency counts) that take the runtime path; default all
static int dbg_memnode() { const char* e = getenv("RNAMPNN_DBG_MEMNODE"); return e ? atoi(e) : 0; }
static int dbg_memnode_sites() { const char* e = getenv("RNAMPNN_DBG_MEMNODE_SITES"); return e ? atoi(e) : 0xff; }
void launch_zero_bytes(void* ptr, size_t bytes, hipStream_t s, int site) {       // ptr 16-byte aligned
    if (!bytes) return;
    if ((dbg_memnode() & 1) && (dbg_memnode_sites() & site)) { (void)hipMemsetAsync(ptr, 0, bytes, s); return; }
    const size_t n16 = bytes / 16;
    size_t g = (n16 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_zero_bytes, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<uint4*>(ptr), n16,
                       reinterpret_cast<unsigned char*>(ptr) + n16 * 16, (int)(bytes - n16 * 16));
}
__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t s) {       // both 16-byte aligned, bytes a multiple of 16
    const size_t n16 = bytes / 16;
    if (!n16) return;
    if (dbg_memnode() & 2) { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s); return; }
    size_t g = (n16 + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16);
}
void launch_zero_regions(const ZeroRegions& z, hipStream_t s) {
    if (z.n > 0) hipLaunchKernelGGL(k_zero_regions, dim3(z.n), dim3(256), 0, s, z);
}

void launch_lengths_from_cu(const int32_t* cu_seqlens, const PackInfo& pk, hipStream_t s) {
    hipLaunchKernelGGL(k_lengths_from_cu, dim3(1), dim3(1024), 0, s, cu_seqlens, pk.B, pk.T, pk.Nmax, pk.len, pk.cu);
}

void launch_lengths(const float* mask, const PackInfo& pk, hipStream_t s) {
    hipLaunchKernelGGL(k_lengths, dim3(pk.B), dim3(WAVE), 0, s, mask, pk.B, pk.T, pk.len);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, pk.len, pk.B, pk.cu);
}

// ------------------------------------------------------------------------------------------
// Per-residue geometry: the 28 raw node features (feature.py:298-384, 531-535) and the
// record the edge kernels gather per neighbour: 21 coordinates, 5 unit bond vectors
// (F.normalize eps 1e-12, feature.py:455-456), 4 unit plane normals (cross of consecutive
// raw bond vectors, eps 1e-6, feature.py:496-505).  One thread per (b, t); t == n_b also
// fills the record of the RNA's phantom neighbour (row Nmax + b), see k_knn.
__device__ __forceinline__ void geom_record(const float* c, float* g) {
#pragma unroll
    for (int i = 0; i < 21; ++i) g[i] = c[i];
    float v[5][3];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
#pragma unroll
        for (int d = 0; d < 3; ++d) v[a][d] = c[(a + 1) * 3 + d] - c[a * 3 + d];
        float nr = sqrtf(v[a][0] * v[a][0] + v[a][1] * v[a][1] + v[a][2] * v[a][2]);
        float inv = 1.0f / fmaxf(nr, 1e-12f);
#pragma unroll
        for (int d = 0; d < 3; ++d) g[21 + a * 3 + d] = v[a][d] * inv;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float x = v[a][1] * v[a + 1][2] - v[a][2] * v[a + 1][1];
        float y = v[a][2] * v[a + 1][0] - v[a][0] * v[a + 1][2];
        float z = v[a][0] * v[a + 1][1] - v[a][1] * v[a + 1][0];
        float inv = 1.0f / fmaxf(sqrtf(x * x + y * y + z * z), kSEPS);
        g[36 + a * 3 + 0] = x * inv; g[36 + a * 3 + 1] = y * inv; g[36 + a * 3 + 2] = z * inv;
    }
}

// The same record split by LANE HALF for the bf16 edge-embedding kernel (RN_GEOMH = 64 floats = 256 B, so a record never
// straddles more than two 128-B lines): half h (floats 32h ..) = atoms 4h..4h+3 (12) | unit bonds 3h..3h+2 (9) |
// unit normals 2h, 2h+1 (6) | zeros (5); the items a half does not have (atom 7, bond 5) are zeros.  Lane half h of the
// kernel gathers only ITS 28 floats of the neighbour (7 x 16 B), the central residue's record is read with scalar loads.
__device__ __forceinline__ void geomh_record(const float* g, float* gh) {
#pragma unroll
    for (int i = 0; i < RN_GEOMH; ++i) gh[i] = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (4 * h + a < 7)
#pragma unroll
                for (int d = 0; d < 3; ++d) gh[32 * h + 3 * a + d] = g[(4 * h + a) * 3 + d];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (3 * h + a < 5)
#pragma unroll
                for (int d = 0; d < 3; ++d) gh[32 * h + 12 + 3 * a + d] = g[21 + (3 * h + a) * 3 + d];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int d = 0; d < 3; ++d) gh[32 * h + 21 + 3 * a + d] = g[36 + (2 * h + a) * 3 + d];
    }
}

__device__ __forceinline__ void raw_features(const float* c, float* f) {
    int o = 0;
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = a + 1; b < 7; ++b) {                       // triu(offset=1), row-major (feature.py:322-325)
            float dx = c[a * 3] - c[b * 3], dy = c[a * 3 + 1] - c[b * 3 + 1], dz = c[a * 3 + 2] - c[b * 3 + 2];
            f[o++] = sqrtf(dx * dx + dy * dy + dz * dz + kSEPS);
        }
    float v[5][3], nr[5], u[5][3];
#pragma unroll
    for (int a = 0; a < 5; ++a) {
#pragma unroll
        for (int d = 0; d < 3; ++d) v[a][d] = c[(a + 1) * 3 + d] - c[a * 3 + d];
        nr[a] = sqrtf(v[a][0] * v[a][0] + v[a][1] * v[a][1] + v[a][2] * v[a][2]);
        float inv = 1.0f / fmaxf(nr[a], kSEPS);                // F.normalize(eps=SEPS) feature.py:377
#pragma unroll
        for (int d = 0; d < 3; ++d) u[a][d] = v[a][d] * inv;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)                                 // feature.py:351-355
        f[21 + a] = (v[a][0] * v[a + 1][0] + v[a][1] * v[a + 1][1] + v[a][2] * v[a + 1][2]) / (nr[a] * nr[a + 1] + kSEPS);
    float nm[4][3];
#pragma unroll
    for (int a = 0; a < 4; ++a) {                               // feature.py:381
        float x = u[a][1] * u[a + 1][2] - u[a][2] * u[a + 1][1];
        float y = u[a][2] * u[a + 1][0] - u[a][0] * u[a + 1][2];
        float z = u[a][0] * u[a + 1][1] - u[a][1] * u[a + 1][0];
        float inv = 1.0f / fmaxf(sqrtf(x * x + y * y + z * z), kSEPS);
        nm[a][0] = x * inv; nm[a][1] = y * inv; nm[a][2] = z * inv;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) f[25 + a] = nm[a + 1][0] * nm[a][0] + nm[a + 1][1] * nm[a][1] + nm[a + 1][2] * nm[a][2];
}

__device__ __forceinline__ void store_geomh(const float* g, float* __restrict__ dst) {
    float gh[RN_GEOMH];
    geomh_record(g, gh);
#pragma unroll
    for (int i = 0; i < RN_GEOMH; i += 4) *reinterpret_cast<float4*>(dst + i) = make_float4(gh[i], gh[i + 1], gh[i + 2], gh[i + 3]);
}

__global__ void k_geom(const float* __restrict__ coords, PackInfo pk, float* __restrict__ raw_out,
                       float* __restrict__ raw_p, float* __restrict__ geom, float* __restrict__ geomh) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= pk.B * pk.T) return;
    int b = id / pk.T, t = id - b * pk.T;
    int n = pk.len[b];
    float c[21];
    // padded input: residue (b, t); packed input: row cu[b] + t, and the (absent) padded residues are zeros
    const bool have = !pk.packed_in || t < n;
    const float* src = coords + (size_t)(pk.packed_in ? pk.cu[b] + (t < n ? t : 0) : id) * 21;
#pragma unroll
    for (int i = 0; i < 21; ++i) c[i] = have ? src[i] : 0.f;
    if (t < n) {
        int p = pk.cu[b] + t;
        pk.node_b[p] = b;
        float f[28];
        raw_features(c, f);
        float* rp = raw_p + (size_t)p * RN_RAWP;
#pragma unroll
        for (int i = 0; i < 28; ++i) rp[i] = f[i];
#pragma unroll
        for (int i = 28; i < RN_RAWP; ++i) rp[i] = 0.f;
        if (raw_out) {
            float* ro = raw_out + (size_t)id * RN_RAW;
#pragma unroll
            for (int i = 0; i < 28; ++i) ro[i] = f[i];
        }
        float g[RN_GEOM];
        geom_record(c, g);
        float* gp = geom + (size_t)p * RN_GEOM;
#pragma unroll
        for (int i = 0; i < RN_GEOM; ++i) gp[i] = g[i];
        if (geomh) store_geomh(g, geomh + (size_t)p * RN_GEOMH);
    } else {
        if (raw_out) {                                          // padded rows: 1e6 distances, 0 cosines
            float* ro = raw_out + (size_t)id * RN_RAW;
#pragma unroll
            for (int i = 0; i < 21; ++i) ro[i] = kLEPS;
#pragma unroll
            for (int i = 21; i < 28; ++i) ro[i] = 0.f;
        }
        if (t == n) {                                           // phantom neighbour of RNA b
            float g[RN_GEOM];
            geom_record(c, g);
            float* gp = geom + (size_t)(pk.Nmax + b) * RN_GEOM;
#pragma unroll
            for (int i = 0; i < RN_GEOM; ++i) gp[i] = g[i];
            if (geomh) store_geomh(g, geomh + (size_t)(pk.Nmax + b) * RN_GEOMH);
        }
    }
}

void launch_geom(const float* coords, const PackInfo& pk, float* raw_out, float* raw_p, float* geom, float* geomh, hipStream_t s) {
    int total = pk.B * pk.T;
    hipLaunchKernelGGL(k_geom, dim3((total + 63) / 64), dim3(64), 0, s, coords, pk, raw_out, raw_p, geom, geomh);
}

// ------------------------------------------------------------------------------------------
// k-NN graph over residue centroids (feature.py:205-256).  One wave per row: the row of
// distances d(i, j) = sqrt(|c_i - c_j|^2 + 1e-6) lives in LDS, and the min(k, n-1) nearest
// residues are extracted in ascending (distance, index) order by k wave-wide min-reductions
// of a packed 64-bit key.  Self and padded residues all sit at exactly 1e6 in the reference
// and are never real neighbours.  Slot n-1 (when n-1 < k): the reference keeps one extra edge
// to a PADDED residue iff T > n -> the phantom neighbour (index n in the
// API tensor, row Nmax + b in the packed index); all later slots are -1.
// G = lanes cooperating on one row (64: one wave per row, any T that fits LDS; 16: four rows per
// wave, 4x the rows in flight - the extraction loop is a chain of dependent cross-lane reductions,
// so throughput comes from rows in flight, not from lanes per row).
template <int G>
__global__ void __launch_bounds__(256) k_knn(const float* __restrict__ coords, PackInfo pk, int k, int rows_per_block,
                                              int* __restrict__ nbr, int64_t* __restrict__ eidx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int GROUPS = 256 / G;
    int b = blockIdx.x;
    int n = pk.len[b];
    int T = pk.T;
    int row0 = blockIdx.y * rows_per_block;
    if (row0 >= T) return;
    int gl = threadIdx.x % G, grp = threadIdx.x / G;
    float* cen = sm;                       // [T][3]
    float* drow = sm + 3 * T + grp * T;    // per-group distance row
    if (row0 < n) {
        for (int j = threadIdx.x; j < n; j += 256) {
            const float* c = coords + (size_t)(pk.packed_in ? pk.cu[b] + j : b * T + j) * 21;
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int a = 0; a < 7; ++a) { sx += c[a * 3]; sy += c[a * 3 + 1]; sz += c[a * 3 + 2]; }
            cen[j * 3] = sx / 7.0f; cen[j * 3 + 1] = sy / 7.0f; cen[j * 3 + 2] = sz / 7.0f;
        }
    }
    __syncthreads();
    int base = pk.cu[b];
    const int row_end = min(T, row0 + rows_per_block);
    for (int i0 = row0; i0 < row_end; i0 += GROUPS) {
        const int i = i0 + grp;
        const bool row_live = i < row_end;          // groups of one wave stay in step (shuffles below)
        const bool real = row_live && i < n;
        if (row_live && !real) {                     // padded row: all -1 (feature.py:253-254)
            if (eidx) for (int s = gl; s < k; s += G) eidx[((size_t)b * T + i) * k + s] = -1;
        }
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (real) { cx = cen[i * 3]; cy = cen[i * 3 + 1]; cz = cen[i * 3 + 2]; }
        if (real) {
            for (int j = gl; j < n; j += G) {
                float dx = __fsub_rn(cen[j * 3], cx), dy = __fsub_rn(cen[j * 3 + 1], cy), dz = __fsub_rn(cen[j * 3 + 2], cz);
                float ss = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                drow[j] = (j == i) ? 3.0e38f : sqrtf(__fadd_rn(ss, kSEPS));
            }
        }
        // (group-private LDS row, written and read by the same lanes: program ordered)
        const int nreal = real ? min(k, n - 1) : 0;
        int nmax = nreal;                            // wave-uniform trip count
#pragma unroll
        for (int o = G; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
        unsigned long long prev = 0ull;              // keys are > 0 (distance >= 1e-3)
        const size_t pbase = (size_t)(base + i) * k;
        const size_t obase = ((size_t)b * T + i) * k;
        if constexpr (G == 64) {
            // Long rows (one wave per row, n / 64 candidates per lane): every lane keeps the smallest key of ITS candidates (j = lane mod 64) in a
            // register; a round is a wave-min of those, and only the winner's candidate has to be renewed - by the whole wave, which re-reads the
            // winner's n / 64 entries (two LDS reads per lane at 4,417 nt) and reduces again.  The plain form below rescans the entire row every round
            // (69 LDS reads + key compares per lane and round at 4,417 nt: 1.6 ms per long batch of the config-3 epoch).  Same keys, same order.
            auto wave_min = [](unsigned long long v) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(v, o, 64); v = other < v ? other : v; }
                return v;
            };
            unsigned long long cand = ~0ull;
            if (real)
                for (int j = gl; j < n; j += 64) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(drow[j]) << 32) | (unsigned)j;
                    cand = key < cand ? key : cand;
                }
            for (int s = 0; s < nreal; ++s) {                 // (G == 64: the wave has one row, nreal is wave-uniform)
                const unsigned long long best = wave_min(cand);
                const int jw = (int)(best & 0xffffffffu), w = jw & 63;
                if (gl == 0) {
                    nbr[pbase + s] = base + jw;
                    if (eidx) eidx[obase + s] = jw;
                }
                unsigned long long nc = ~0ull;                 // the winner lane's next candidate: smallest key > best among j = w (mod 64)
                for (int j = w + 64 * gl; j < n; j += 64 * 64) {
                    const unsigned long long key = ((unsigned long long)__float_as_uint(drow[j]) << 32) | (unsigned)j;
                    if (key > best && key < nc) nc = key;
                }
                nc = wave_min(nc);
                if (gl == w) cand = nc;
            }
        } else {
        for (int s = 0; s < nmax; ++s) {
            unsigned long long best = ~0ull;
            if (s < nreal) {
                for (int j = gl; j < n; j += G) {
                    unsigned long long key = ((unsigned long long)__float_as_uint(drow[j]) << 32) | (unsigned)j;
                    if (key > prev && key < best) best = key;
                }
            }
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) {
                unsigned long long other = __shfl_xor(best, o, 64);
                best = other < best ? other : best;
            }
            prev = best;
            if (gl == 0 && s < nreal) {
                int j = (int)(best & 0xffffffffu);
                nbr[pbase + s] = base + j;
                if (eidx) eidx[obase + s] = j;
            }
        }
        }
        if (real) {
            for (int s = nreal + gl; s < k; s += G) {
                bool phantom = (s == n - 1) && (n < T);
                nbr[pbase + s] = phantom ? pk.Nmax + b : -1;
                if (eidx) eidx[obase + s] = phantom ? n : -1;
            }
        }
    }
}

// The same selection with every lane's candidates in REGISTERS as a sorted queue (max_len <= 16 NK): the 16 lanes of a row each sort
// their NK keys once (odd-even transposition network), an extraction round is then one 16-lane min-reduction of the queue HEADS and a
// conditional shift of the winner's queue - no re-scan of the row per round (the re-scan was 3/4 of the instructions of k_knn<16>).
// Identical output: ascending (distance, index) order, same phantom / -1 rule.
template <int NK>
__global__ void __launch_bounds__(256) k_knn_queue(const float* __restrict__ coords, PackInfo pk, int k, int rows_per_block,
                                                    int* __restrict__ nbr, int64_t* __restrict__ eidx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int G = 16, GROUPS = 256 / G;
    const int b = blockIdx.x;
    const int n = pk.len[b];
    const int T = pk.T;
    const int row0 = blockIdx.y * rows_per_block;
    if (row0 >= T) return;
    const int gl = threadIdx.x % G, grp = threadIdx.x / G;
    float* cen = sm;                       // [T][3]
    if (row0 < n) {
        for (int j = threadIdx.x; j < n; j += 256) {
            const float* c = coords + (size_t)(pk.packed_in ? pk.cu[b] + j : b * T + j) * 21;
            float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
            for (int a = 0; a < 7; ++a) { sx += c[a * 3]; sy += c[a * 3 + 1]; sz += c[a * 3 + 2]; }
            cen[j * 3] = sx / 7.0f; cen[j * 3 + 1] = sy / 7.0f; cen[j * 3 + 2] = sz / 7.0f;
        }
    }
    __syncthreads();
    const int base = pk.cu[b];
    const int row_end = min(T, row0 + rows_per_block);
    for (int i0 = row0; i0 < row_end; i0 += GROUPS) {
        const int i = i0 + grp;
        const bool row_live = i < row_end;
        const bool real = row_live && i < n;
        if (row_live && !real) {
            if (eidx) for (int s = gl; s < k; s += G) eidx[((size_t)b * T + i) * k + s] = -1;
        }
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (real) { cx = cen[i * 3]; cy = cen[i * 3 + 1]; cz = cen[i * 3 + 2]; }
        unsigned long long q[NK];
#pragma unroll
        for (int t = 0; t < NK; ++t) {
            const int j = gl + G * t;
            unsigned long long key = ~0ull;
            if (real && j < n && j != i) {
                float dx = __fsub_rn(cen[j * 3], cx), dy = __fsub_rn(cen[j * 3 + 1], cy), dz = __fsub_rn(cen[j * 3 + 2], cz);
                float ss = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                key = ((unsigned long long)__float_as_uint(sqrtf(__fadd_rn(ss, kSEPS))) << 32) | (unsigned)j;
            }
            q[t] = key;
        }
#pragma unroll
        for (int pass = 0; pass < NK; ++pass)        // odd-even transposition sort, ascending
#pragma unroll
            for (int t = pass & 1; t + 1 < NK; t += 2) {
                const unsigned long long lo = q[t] < q[t + 1] ? q[t] : q[t + 1], hi = q[t] < q[t + 1] ? q[t + 1] : q[t];
                q[t] = lo; q[t + 1] = hi;
            }
        const int nreal = real ? min(k, n - 1) : 0;
        int nmax = nreal;                            // wave-uniform trip count
#pragma unroll
        for (int o = G; o < 64; o <<= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
        const size_t pbase = (size_t)(base + i) * k;
        const size_t obase = ((size_t)b * T + i) * k;
        for (int s = 0; s < nmax; ++s) {
            // all-reduce min over the 16 lanes of the row by DPP rotations within the row (row_ror 8, 4, 2, 1): VALU moves instead of
            // four dependent trips through the LDS crossbar (ds_bpermute) - the extraction loop is a latency chain
            unsigned long long best = q[0];
#define KNN_ROR(n) do {                                                                                                      \
                const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)best, 0x120 + (n), 0xf, 0xf, false);            \
                const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(best >> 32), 0x120 + (n), 0xf, 0xf, false);     \
                const unsigned long long other_ = ((unsigned long long)hi_ << 32) | lo_;                                     \
                best = other_ < best ? other_ : best;                                                                        \
            } while (0)
            KNN_ROR(8); KNN_ROR(4); KNN_ROR(2); KNN_ROR(1);
#undef KNN_ROR
            if (q[0] == best && best != ~0ull) {     // keys are unique (they carry j): exactly one lane of the group shifts
#pragma unroll
                for (int t = 0; t + 1 < NK; ++t) q[t] = q[t + 1];
                q[NK - 1] = ~0ull;
            }
            if (gl == 0 && s < nreal) {
                const int j = (int)(best & 0xffffffffu);
                nbr[pbase + s] = base + j;
                if (eidx) eidx[obase + s] = j;
            }
        }
        if (real) {
            for (int s = nreal + gl; s < k; s += G) {
                const bool phantom = (s == n - 1) && (n < T);
                nbr[pbase + s] = phantom ? pk.Nmax + b : -1;
                if (eidx) eidx[obase + s] = phantom ? n : -1;
            }
        }
    }
}

int launch_knn(const float* coords, const PackInfo& pk, int k, int* nbr, int64_t* eidx, hipStream_t s) {
    static const bool no_queue = [] { const char* e = getenv("RNAMPNN_KNN_SCAN"); return e && e[0] == '1'; }();
    if (!no_queue && pk.T <= 256) {                  // register-resident sorted queues
        const int rpb = 32;
        dim3 grid(pk.B, (pk.T + rpb - 1) / rpb);
        const size_t lds = (size_t)3 * pk.T * sizeof(float);
        if (pk.T <= 64) hipLaunchKernelGGL(k_knn_queue<4>, grid, dim3(256), lds, s, coords, pk, k, rpb, nbr, eidx);
        else if (pk.T <= 144) hipLaunchKernelGGL(k_knn_queue<9>, grid, dim3(256), lds, s, coords, pk, k, rpb, nbr, eidx);
        else hipLaunchKernelGGL(k_knn_queue<16>, grid, dim3(256), lds, s, coords, pk, k, rpb, nbr, eidx);
        return 0;
    }
    const size_t lds16 = (size_t)(3 + 16) * pk.T * sizeof(float);
    if (lds16 <= 48 * 1024) {                        // four rows per wave
        int rpb = 32;
        dim3 grid(pk.B, (pk.T + rpb - 1) / rpb);
        hipLaunchKernelGGL(k_knn<16>, grid, dim3(256), lds16, s, coords, pk, k, rpb, nbr, eidx);
        return 0;
    }
    size_t lds = (size_t)(3 + 4) * pk.T * sizeof(float);
    if (lds > 160 * 1024 - 256) return 1;                       // T too long for the LDS-resident row
    static DevAttr attr;
    ensure_dyn_lds((const void*)k_knn<64>, lds, attr);
    int rpb = pk.T <= 512 ? 16 : 64;
    dim3 grid(pk.B, (pk.T + rpb - 1) / rpb);
    hipLaunchKernelGGL(k_knn<64>, grid, dim3(256), lds, s, coords, pk, k, rpb, nbr, eidx);
    return 0;
}

// ------------------------------------------------------------------------------------------
// 32 x 128 tile of a Linear layer on 256 threads: thread (c = tid & 127, g = tid >> 7) owns
// column c of rows g*16 .. g*16+15.  X tile in LDS (row stride ldx floats), weights K-major in
// HBM/L2 (coalesced across c).  All lanes of a wave share g, so X reads are LDS broadcasts.
__device__ __forceinline__ void tile_fma(const float* __restrict__ Xs, int ldx, int K, const float* __restrict__ Wt,
                                         int ldw, int c, int g, float (&acc)[16]) {
    const float* xr = Xs + g * 16 * ldx;
    for (int kk = 0; kk < K; kk += 4) {
        float w0 = Wt[(size_t)(kk + 0) * ldw + c], w1 = Wt[(size_t)(kk + 1) * ldw + c];
        float w2 = Wt[(size_t)(kk + 2) * ldw + c], w3 = Wt[(size_t)(kk + 3) * ldw + c];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float4 x = *reinterpret_cast<const float4*>(xr + r * ldx + kk);
            acc[r] = fmaf(x.x, w0, acc[r]); acc[r] = fmaf(x.y, w1, acc[r]);
            acc[r] = fmaf(x.z, w2, acc[r]); acc[r] = fmaf(x.w, w3, acc[r]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Edge featurisation + edge-embedding MLP (feature.py:386-571).  One workgroup per residue:
// thread s < k computes the 90 raw features of edge (p, s) from the two geometry records
// (49 distances a*7+b, 25 bond cosines a*5+b, 16 normal cosines a*4+b) into LDS - the
// 90-wide tensor never reaches HBM - then Linear(90,128)+GELU [+ Linear(128,128)+GELU].
// Invalid slots (nbr == -1) give e = 0 (feature.py:564-569).
#define EX_LD 100   // 96 + 4 pad (keeps float4 alignment, staggers banks)
#define EH_LD 132
__global__ void __launch_bounds__(256) k_edge_embed_f32(PackInfo pk, int k, const float* __restrict__ geom,
        const int* __restrict__ nbr, const float* __restrict__ w0t, const float* __restrict__ b0,
        const float* __restrict__ w1t, const float* __restrict__ b1, int depth, float* __restrict__ e) {
    __shared__ __attribute__((aligned(16))) float X[32 * EX_LD];
    __shared__ __attribute__((aligned(16))) float H[32 * EH_LD];
    __shared__ int valid_s[32];
    int ntot = pk.cu[pk.B];
    for (int p = blockIdx.x; p < ntot; p += gridDim.x) {
        int tid = threadIdx.x;
        if (tid < 32) {
            int j = (tid < k) ? nbr[(size_t)p * k + tid] : -1;
            valid_s[tid] = j >= 0;
            float* x = X + tid * EX_LD;
            if (j >= 0) {
                const float* gi = geom + (size_t)p * RN_GEOM;
                const float* gj = geom + (size_t)j * RN_GEOM;
                float cj[RN_GEOM];
#pragma unroll
                for (int i = 0; i < RN_GEOM; ++i) cj[i] = gj[i];
#pragma unroll
                for (int a = 0; a < 7; ++a) {
                    float ax = gi[a * 3], ay = gi[a * 3 + 1], az = gi[a * 3 + 2];
#pragma unroll
                    for (int bb = 0; bb < 7; ++bb) {
                        float dx = ax - cj[bb * 3], dy = ay - cj[bb * 3 + 1], dz = az - cj[bb * 3 + 2];
                        x[a * 7 + bb] = sqrtf(dx * dx + dy * dy + dz * dz + kSEPS);
                    }
                }
#pragma unroll
                for (int a = 0; a < 5; ++a) {
                    float ax = gi[21 + a * 3], ay = gi[22 + a * 3], az = gi[23 + a * 3];
#pragma unroll
                    for (int bb = 0; bb < 5; ++bb)
                        x[49 + a * 5 + bb] = ax * cj[21 + bb * 3] + ay * cj[22 + bb * 3] + az * cj[23 + bb * 3];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    float ax = gi[36 + a * 3], ay = gi[37 + a * 3], az = gi[38 + a * 3];
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb)
                        x[74 + a * 4 + bb] = ax * cj[36 + bb * 3] + ay * cj[37 + bb * 3] + az * cj[38 + bb * 3];
                }
#pragma unroll
                for (int i = RN_ERAW; i < RN_ERAWP; ++i) x[i] = 0.f;
            } else {
                for (int i = 0; i < RN_ERAWP; ++i) x[i] = 0.f;
            }
        }
        __syncthreads();
        int c = tid & 127, g = tid >> 7;
        float acc[16];
        float bias = b0[c];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bias;
        tile_fma(X, EX_LD, RN_ERAWP, w0t, RN_D, c, g, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = gelu_erf(acc[r]);
        if (depth > 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) H[(g * 16 + r) * EH_LD + c] = acc[r];
            __syncthreads();
            bias = b1[c];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = bias;
            tile_fma(H, EH_LD, RN_D, w1t, RN_D, c, g, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = gelu_erf(acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int s = g * 16 + r;
            if (s < k) e[((size_t)p * k + s) * RN_D + c] = valid_s[s] ? acc[r] : 0.f;
        }
        __syncthreads();
    }
}

void launch_edge_embed_f32(const PackInfo& pk, int k, const float* geom, const int* nbr, const float* w0t,
                           const float* b0, const float* w1t, const float* b1, int depth, float* e, hipStream_t s) {
    int grid = pk.Nmax < 4096 ? pk.Nmax : 4096;
    hipLaunchKernelGGL(k_edge_embed_f32, dim3(grid), dim3(256), 0, s, pk, k, geom, nbr, w0t, b0, w1t, b1, depth, e);
}

// ------------------------------------------------------------------------------------------
// One ResMPNN step on the k edges of one residue (mpnn.py:154-265), f32.
// The first Linear of both MLPs acts on cat[h_i, h_j, e] (mpnn.py:184-188, 260); it is
// evaluated as P[i] + Q[j] + e.Wc with P = h.Wa^T + b1 and Q = h.Wb^T computed once per
// residue by the node GEMM (pq_* rows are [P | Q], 256 wide) - identical mathematics,
// different summation order.  DO_EDGE: e <- e + MLP_e(...) (mpnn.py:263; invalid slots stay
// 0, the reference leaves never-consumed garbage there).  DO_MSG: messages, masked mean over
// the valid slots, residual: h_pre = h + sum/max(cnt,1) (mpnn.py:191-225).
template <bool DO_EDGE, bool DO_MSG>
__global__ void __launch_bounds__(256) k_mpnn_f32(PackInfo pk, int k, const int* __restrict__ nbr, float* __restrict__ e,
        const float* __restrict__ pq_e, const float* __restrict__ pq_m, MpnnW32 we, MpnnW32 wm,
        const float* __restrict__ h_in, float* __restrict__ h_pre, float* __restrict__ msg_out) {
    __shared__ __attribute__((aligned(16))) float X[32 * EH_LD];
    __shared__ __attribute__((aligned(16))) float H[32 * EH_LD];
    __shared__ int jrow[32];
    __shared__ float part[128];
    int ntot = pk.cu[pk.B];
    int tid = threadIdx.x, c = tid & 127, g = tid >> 7;
    for (int p = blockIdx.x; p < ntot; p += gridDim.x) {
        if (tid < 32) jrow[tid] = (tid < k) ? nbr[(size_t)p * k + tid] : -1;
        for (int idx = tid; idx < 32 * 32; idx += 256) {        // e rows -> LDS (float4 per thread)
            int r = idx >> 5, q = idx & 31;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < k) v = *reinterpret_cast<const float4*>(e + ((size_t)p * k + r) * RN_D + q * 4);
            *reinterpret_cast<float4*>(X + r * EH_LD + q * 4) = v;
        }
        __syncthreads();
        float acc[16];
        if (DO_EDGE) {
            float pi = pq_e[(size_t)p * 256 + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int j = jrow[g * 16 + r];
                j = j < 0 ? 0 : (j > pk.Nmax ? pk.Nmax : j);    // phantom -> the all-zero row Nmax
                acc[r] = pi + pq_e[(size_t)j * 256 + 128 + c];
            }
            tile_fma(X, EH_LD, RN_D, we.wc_t, RN_D, c, g, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = gelu_erf(acc[r]);
            if (we.depth > 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) H[(g * 16 + r) * EH_LD + c] = acc[r];
                __syncthreads();
                float bias = we.b2[c];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bias;
                tile_fma(H, EH_LD, RN_D, we.w2_t, RN_D, c, g, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = gelu_erf(acc[r]);
            } else {
                __syncthreads();
            }
            // every thread has finished reading X for the first Linear (barrier above)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int s = g * 16 + r;
                if (s < k && jrow[s] >= 0) {
                    float v = X[s * EH_LD + c] + acc[r];
                    X[s * EH_LD + c] = v;
                    e[((size_t)p * k + s) * RN_D + c] = v;
                }
            }
            __syncthreads();
        }
        if (DO_MSG) {
            float pi = pq_m[(size_t)p * 256 + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int j = jrow[g * 16 + r];
                j = j < 0 ? 0 : (j > pk.Nmax ? pk.Nmax : j);
                acc[r] = pi + pq_m[(size_t)j * 256 + 128 + c];
            }
            tile_fma(X, EH_LD, RN_D, wm.wc_t, RN_D, c, g, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = gelu_erf(acc[r]);
            if (wm.depth > 1) {
                __syncthreads();                                 // H may still be read by the edge MLP
#pragma unroll
                for (int r = 0; r < 16; ++r) H[(g * 16 + r) * EH_LD + c] = acc[r];
                __syncthreads();
                float bias = wm.b2[c];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = bias;
                tile_fma(H, EH_LD, RN_D, wm.w2_t, RN_D, c, g, acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = gelu_erf(acc[r]);
            }
            float sum = 0.f;
            int cnt = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int s = g * 16 + r;
                bool ok = s < k && jrow[s] >= 0;
                float m = ok ? acc[r] : 0.f;
                sum += m;
                if (msg_out && s < k) msg_out[((size_t)p * k + s) * RN_D + c] = m;
            }
            for (int s = 0; s < k; ++s) cnt += jrow[s] >= 0;
            if (g == 1) part[c] = sum;
            __syncthreads();
            if (g == 0) {
                float tot = sum + part[c];
                float denom = (float)(cnt > 0 ? cnt : 1);
                h_pre[(size_t)p * RN_D + c] = h_in[(size_t)p * RN_D + c] + tot / denom;
            }
        }
        __syncthreads();
    }
}

void launch_mpnn_f32(const PackInfo& pk, int k, bool do_edge, bool do_msg, const int* nbr, float* e,
                     const float* pq_e, const float* pq_m, MpnnW32 we, MpnnW32 wm, const float* h_in,
                     float* h_pre, float* msg_out, hipStream_t s) {
    int grid = pk.Nmax < 4096 ? pk.Nmax : 4096;
    if (do_edge && do_msg)
        hipLaunchKernelGGL((k_mpnn_f32<true, true>), dim3(grid), dim3(256), 0, s, pk, k, nbr, e, pq_e, pq_m, we, wm, h_in, h_pre, msg_out);
    else if (do_edge)
        hipLaunchKernelGGL((k_mpnn_f32<true, false>), dim3(grid), dim3(256), 0, s, pk, k, nbr, e, pq_e, pq_m, we, wm, h_in, h_pre, msg_out);
    else
        hipLaunchKernelGGL((k_mpnn_f32<false, true>), dim3(grid), dim3(256), 0, s, pk, k, nbr, e, pq_e, pq_m, we, wm, h_in, h_pre, msg_out);
}

// ------------------------------------------------------------------------------------------
// GraphNormalization on packed rows (functional.py:18-48), D = 128, one workgroup per RNA:
//   mu = sum_valid x / n ;  var = [sum_valid (x-mu)^2 + (T_tot - n) mu^2] / n
//   y = (x - mu) / sqrt(var + 1e-6) * scale + shift         (padded rows do not exist here)
// x may come as x + add (the residual h + agg of mpnn.py:222 on the bf16 path).  Thread layout:
// 32 channel quads (float4) x 8 row groups; the three passes re-read the L2-resident rows.
__global__ void __launch_bounds__(256) k_graph_norm_packed(PackInfo pk, const float* __restrict__ x, const float* __restrict__ add,
        float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int t_tot) {
    __shared__ float4 red[8][32];
    __shared__ float4 stat[32];
    int b = blockIdx.x;
    int n = pk.len[b];
    if (n <= 0) return;
    const size_t base = (size_t)pk.cu[b] * RN_D;
    const int cq = threadIdx.x & 31, g = threadIdx.x >> 5;
    const float4* xb = reinterpret_cast<const float4*>(x + base) + cq;
    const float4* ab = add ? reinterpret_cast<const float4*>(add + base) + cq : nullptr;
    auto ld = [&](int r) {
        float4 v = xb[(size_t)r * 32];
        if (ab) { float4 a = ab[(size_t)r * 32]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        return v;
    };
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = g; r < n; r += 8) { float4 v = ld(r); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    red[g][cq] = s;
    __syncthreads();
    if (g == 0) {
        float4 t = red[0][cq];
#pragma unroll
        for (int i = 1; i < 8; ++i) { float4 u = red[i][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        float inv = 1.0f / (float)n;
        stat[cq] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
    __syncthreads();
    const float4 mean = stat[cq];
    float4 ss = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = g; r < n; r += 8) {
        float4 v = ld(r);
        float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
        ss.x = fmaf(dx, dx, ss.x); ss.y = fmaf(dy, dy, ss.y); ss.z = fmaf(dz, dz, ss.z); ss.w = fmaf(dw, dw, ss.w);
    }
    __syncthreads();
    red[g][cq] = ss;
    __syncthreads();
    float4 t = red[0][cq];
#pragma unroll
    for (int i = 1; i < 8; ++i) { float4 u = red[i][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    const float pad = (float)(t_tot - n), fn = (float)n;
    float4 sd;
    sd.x = sqrtf((t.x + pad * mean.x * mean.x) / fn + kSEPS); sd.y = sqrtf((t.y + pad * mean.y * mean.y) / fn + kSEPS);
    sd.z = sqrtf((t.z + pad * mean.z * mean.z) / fn + kSEPS); sd.w = sqrtf((t.w + pad * mean.w * mean.w) / fn + kSEPS);
    const float4 sc = reinterpret_cast<const float4*>(scale)[cq], sh = reinterpret_cast<const float4*>(shift)[cq];
    float4* yb = reinterpret_cast<float4*>(y + base) + cq;
#pragma unroll 4
    for (int r = g; r < n; r += 8) {
        float4 v = ld(r), o;
        o.x = (v.x - mean.x) / sd.x * sc.x + sh.x; o.y = (v.y - mean.y) / sd.y * sc.y + sh.y;
        o.z = (v.z - mean.z) / sd.z * sc.z + sh.z; o.w = (v.w - mean.w) / sd.w * sc.w + sh.w;
        yb[(size_t)r * 32] = o;
    }
}

// Same, for RNAs of at most 8 * NR residues: every thread keeps its <= NR rows in registers, so the rows are read from
// memory once (all loads in flight together) instead of three times in dependent loops.
template <int NR>
__global__ void __launch_bounds__(256) k_graph_norm_packed_reg(PackInfo pk, const float* __restrict__ x, const float* __restrict__ add,
        float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int t_tot) {
    __shared__ float4 red[8][32];
    __shared__ float4 stat[32];
    const int b = blockIdx.x;
    const int n = pk.len[b];
    if (n <= 0) return;
    const size_t base = (size_t)pk.cu[b] * RN_D;
    const int cq = threadIdx.x & 31, g = threadIdx.x >> 5;
    const float4* xb = reinterpret_cast<const float4*>(x + base) + cq;
    const float4* ab = add ? reinterpret_cast<const float4*>(add + base) + cq : nullptr;
    float4 v[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = g + 8 * i, rc = r < n ? r : n - 1;
        v[i] = xb[(size_t)rc * 32];
    }
    if (ab) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = g + 8 * i, rc = r < n ? r : n - 1;
            const float4 a = ab[(size_t)rc * 32];
            v[i].x += a.x; v[i].y += a.y; v[i].z += a.z; v[i].w += a.w;
        }
    }
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NR; ++i)
        if (g + 8 * i < n) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    red[g][cq] = s;
    __syncthreads();
    if (g == 0) {
        float4 t = red[0][cq];
#pragma unroll
        for (int i = 1; i < 8; ++i) { float4 u = red[i][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float inv = 1.0f / (float)n;
        stat[cq] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
    __syncthreads();
    const float4 mean = stat[cq];
    float4 ss = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < NR; ++i)
        if (g + 8 * i < n) {
            const float dx = v[i].x - mean.x, dy = v[i].y - mean.y, dz = v[i].z - mean.z, dw = v[i].w - mean.w;
            ss.x = fmaf(dx, dx, ss.x); ss.y = fmaf(dy, dy, ss.y); ss.z = fmaf(dz, dz, ss.z); ss.w = fmaf(dw, dw, ss.w);
        }
    __syncthreads();
    red[g][cq] = ss;
    __syncthreads();
    float4 t = red[0][cq];
#pragma unroll
    for (int i = 1; i < 8; ++i) { float4 u = red[i][cq]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    const float pad = (float)(t_tot - n), fn = (float)n;
    float4 sd;
    sd.x = sqrtf((t.x + pad * mean.x * mean.x) / fn + kSEPS); sd.y = sqrtf((t.y + pad * mean.y * mean.y) / fn + kSEPS);
    sd.z = sqrtf((t.z + pad * mean.z * mean.z) / fn + kSEPS); sd.w = sqrtf((t.w + pad * mean.w * mean.w) / fn + kSEPS);
    const float4 sc = reinterpret_cast<const float4*>(scale)[cq], sh = reinterpret_cast<const float4*>(shift)[cq];
    float4* yb = reinterpret_cast<float4*>(y + base) + cq;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int r = g + 8 * i;
        if (r < n) {
            float4 o;
            o.x = (v[i].x - mean.x) / sd.x * sc.x + sh.x; o.y = (v[i].y - mean.y) / sd.y * sc.y + sh.y;
            o.z = (v[i].z - mean.z) / sd.z * sc.z + sh.z; o.w = (v[i].w - mean.w) / sd.w * sc.w + sh.w;
            yb[(size_t)r * 32] = o;
        }
    }
}

// Same, for LONG RNAs (T > 512): four workgroups per RNA, one per 32-channel chunk (thread = channel quad cq of the chunk x row group g of 32;
// a wave reads 128 bytes of 8 rows per instruction).  With one workgroup per RNA a batch of seven 4,400-nt RNAs used 7 of the 256 CUs
// (95 us per call on the config-3 epoch).  Per-channel arithmetic and its order over the row groups differ from the kernels above
// (32 partial sums instead of 8): results agree to rounding, not bit for bit.
__global__ void __launch_bounds__(256) k_graph_norm_packed_c4(PackInfo pk, const float* __restrict__ x, const float* __restrict__ add,
        float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int t_tot) {
    __shared__ float4 red[32][8];
    __shared__ float4 stat[8];
    const int b = blockIdx.x;
    const int n = pk.len[b];
    if (n <= 0) return;
    const size_t base = (size_t)pk.cu[b] * RN_D;
    const int cl = threadIdx.x & 7, g = threadIdx.x >> 3, cq = 8 * blockIdx.y + cl;
    const float4* xb = reinterpret_cast<const float4*>(x + base) + cq;
    const float4* ab = add ? reinterpret_cast<const float4*>(add + base) + cq : nullptr;
    auto ld = [&](int r) {
        float4 v = xb[(size_t)r * 32];
        if (ab) { float4 a = ab[(size_t)r * 32]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        return v;
    };
    auto fold = [&]() {
        float4 t = red[0][cl];
#pragma unroll
        for (int i = 1; i < 32; ++i) { const float4 u = red[i][cl]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        return t;
    };
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = g; r < n; r += 32) { const float4 v = ld(r); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    red[g][cl] = s;
    __syncthreads();
    if (g == 0) {
        const float4 t = fold();
        const float inv = 1.0f / (float)n;
        stat[cl] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
    __syncthreads();
    const float4 mean = stat[cl];
    float4 ss = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = g; r < n; r += 32) {
        const float4 v = ld(r);
        const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
        ss.x = fmaf(dx, dx, ss.x); ss.y = fmaf(dy, dy, ss.y); ss.z = fmaf(dz, dz, ss.z); ss.w = fmaf(dw, dw, ss.w);
    }
    red[g][cl] = ss;
    __syncthreads();
    const float4 t = fold();
    const float pad = (float)(t_tot - n), fn = (float)n;
    float4 sd;
    sd.x = sqrtf((t.x + pad * mean.x * mean.x) / fn + kSEPS); sd.y = sqrtf((t.y + pad * mean.y * mean.y) / fn + kSEPS);
    sd.z = sqrtf((t.z + pad * mean.z * mean.z) / fn + kSEPS); sd.w = sqrtf((t.w + pad * mean.w * mean.w) / fn + kSEPS);
    const float4 sc = reinterpret_cast<const float4*>(scale)[cq], sh = reinterpret_cast<const float4*>(shift)[cq];
    float4* yb = reinterpret_cast<float4*>(y + base) + cq;
#pragma unroll 4
    for (int r = g; r < n; r += 32) {
        const float4 v = ld(r);
        float4 o;
        o.x = (v.x - mean.x) / sd.x * sc.x + sh.x; o.y = (v.y - mean.y) / sd.y * sc.y + sh.y;
        o.z = (v.z - mean.z) / sd.z * sc.z + sh.z; o.w = (v.w - mean.w) / sd.w * sc.w + sh.w;
        yb[(size_t)r * 32] = o;
    }
}

void launch_graph_norm_packed(const PackInfo& pk, const float* x, const float* add, float* y, const float* scale,
                              const float* shift, int t_tot, hipStream_t s) {
    // (same arithmetic, same summation order per thread: the register variant is bit-identical to the loop variant)
    if (pk.T <= 160) hipLaunchKernelGGL(k_graph_norm_packed_reg<20>, dim3(pk.B), dim3(256), 0, s, pk, x, add, y, scale, shift, t_tot);
    else if (pk.T <= 256) hipLaunchKernelGGL(k_graph_norm_packed_reg<32>, dim3(pk.B), dim3(256), 0, s, pk, x, add, y, scale, shift, t_tot);      // (C4: 200 nt)
    else if (pk.T <= 512) hipLaunchKernelGGL(k_graph_norm_packed, dim3(pk.B), dim3(256), 0, s, pk, x, add, y, scale, shift, t_tot);
    else hipLaunchKernelGGL(k_graph_norm_packed_c4, dim3(pk.B, 4), dim3(256), 0, s, pk, x, add, y, scale, shift, t_tot);
}

// Stand-alone GraphNormalization on the reference's padded layout, any D, mask by value.
__global__ void __launch_bounds__(256) k_graph_norm_padded(const float* __restrict__ x, const float* __restrict__ mask,
        const float* __restrict__ scale, const float* __restrict__ shift, int T, int t_tot, int D, float* __restrict__ y) {
    int b = blockIdx.x;
    const float* xb = x + (size_t)b * T * D;
    const float* mb = mask + (size_t)b * T;
    float* yb = y + (size_t)b * T * D;
    for (int c = threadIdx.x; c < D; c += 256) {
        float cnt = 0.f, s = 0.f;
        for (int t = 0; t < T; ++t) { float m = mb[t]; cnt += m; s += xb[(size_t)t * D + c] * m; }
        float n = cnt == 0.f ? 1.f : cnt;
        float mean = s / n;
        float ss = 0.f;
        for (int t = 0; t < T; ++t) { float d = xb[(size_t)t * D + c] * mb[t] - mean; ss = fmaf(d, d, ss); }
        ss += (float)(t_tot - T) * mean * mean;
        float sd = sqrtf(ss / n + kSEPS);
        float sc = scale[c], sh = shift[c];
        for (int t = 0; t < T; ++t) yb[(size_t)t * D + c] = ((xb[(size_t)t * D + c] - mean) / sd * sc + sh) * mb[t];
    }
}

void launch_graph_norm_padded(const float* x, const float* mask, const float* scale, const float* shift, int B, int T,
                              int t_tot, int D, float* y, hipStream_t s) {
    hipLaunchKernelGGL(k_graph_norm_padded, dim3(B), dim3(256), 0, s, x, mask, scale, shift, T, t_tot, D, y);
}

// ------------------------------------------------------------------------------------------
// Generic node-level Linear on packed rows: Y = act([X | X2] . Wt + bias) (+ res).
// 32 x 128 output tile per workgroup, K staged through LDS in chunks of 128.
__global__ void __launch_bounds__(256) k_gemm_f32(const int* __restrict__ ntot_p, const float* __restrict__ X, int ldx, int K1,
        const float* __restrict__ X2, int ldx2, int K2, const float* __restrict__ Wt, const float* __restrict__ bias,
        int N, int act, const float* __restrict__ res, int ldres, float* __restrict__ Y, int ldy) {
    __shared__ __attribute__((aligned(16))) float Xs[32 * EH_LD];
    int ntot = *ntot_p;
    int row0 = blockIdx.x * 32;
    if (row0 >= ntot) return;
    int col0 = blockIdx.y * 128;
    int tid = threadIdx.x, c = tid & 127, g = tid >> 7;
    int cc = col0 + c;
    int cw = cc < N ? cc : N - 1;                               // clamp: out-of-range columns compute and drop
    float acc[16];
    float bv = bias ? bias[cw] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bv;
    int K = K1 + K2;
    for (int k0 = 0; k0 < K; k0 += 128) {
        int kc = min(128, K - k0);                              // multiple of 4 by construction
        for (int idx = tid; idx < 32 * 32; idx += 256) {
            int r = idx >> 5, q = idx & 31;
            int kk = k0 + q * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int row = row0 + r;
            if (row < ntot && q * 4 < kc) {
                if (kk < K1) v = *reinterpret_cast<const float4*>(X + (size_t)row * ldx + kk);
                else v = *reinterpret_cast<const float4*>(X2 + (size_t)row * ldx2 + (kk - K1));
            }
            *reinterpret_cast<float4*>(Xs + r * EH_LD + q * 4) = v;
        }
        __syncthreads();
        tile_fma(Xs, EH_LD, kc, Wt + (size_t)k0 * N + cw - c, N, c, g, acc);
        __syncthreads();
    }
    if (cc < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = row0 + g * 16 + r;
            if (row < ntot) {
                float v = acc[r];
                if (act == 1) v = gelu_erf(v);
                if (res) v += res[(size_t)row * ldres + cc];
                Y[(size_t)row * ldy + cc] = v;
            }
        }
    }
}

void launch_gemm_f32(const int* ntot, int mmax, const float* X, int ldx, int K1, const float* X2, int ldx2, int K2,
                     const float* Wt, const float* bias, int N, int act, const float* res, int ldres,
                     float* Y, int ldy, hipStream_t s) {
    dim3 grid((mmax + 31) / 32, (N + 127) / 128);
    hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, s, ntot, X, ldx, K1, X2, ldx2, K2, Wt, bias, N, act, res, ldres, Y, ldy);
}

// ------------------------------------------------------------------------------------------
// nn.MultiheadAttention forward over the VALID keys of one RNA (functional.py:164-168):
// qkv rows are [q | k | v] (3 x 128) after the in-projection; one thread per query, keys
// staged through LDS in chunks of 64, online softmax.  Padded keys are masked in the
// reference and padded queries are discarded, so neither exists here.
template <int HD>
__global__ void __launch_bounds__(64) k_attention_f32(PackInfo pk, const float* __restrict__ qkv, int heads,
                                                     float* __restrict__ out) {
    __shared__ float Ks[64 * HD];
    __shared__ float Vs[64 * HD];
    int b = blockIdx.x, hd = blockIdx.y;
    int n = pk.len[b];
    int q0 = blockIdx.z * 64;
    if (q0 >= n) return;
    int base = pk.cu[b];
    int qi = q0 + threadIdx.x;
    bool active = qi < n;
    float q[HD], acc[HD];
    float scale = rsqrtf((float)HD);
    const float* qp = qkv + (size_t)(base + (active ? qi : 0)) * 384 + hd * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = qp[d] * scale; acc[d] = 0.f; }
    float m = -3.0e38f, l = 0.f;
    for (int k0 = 0; k0 < n; k0 += 64) {
        int kn = min(64, n - k0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < kn * HD; idx += 64) {
            int r = idx / HD, d = idx - r * HD;
            const float* row = qkv + (size_t)(base + k0 + r) * 384 + hd * HD + d;
            Ks[idx] = row[128];
            Vs[idx] = row[256];
        }
        __syncthreads();
        for (int j = 0; j < kn; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) sc = fmaf(q[d], Ks[j * HD + d], sc);
            float mn = fmaxf(m, sc);
            float corr = __expf(m - mn), pj = __expf(sc - mn);
            l = l * corr + pj;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(pj, Vs[j * HD + d], acc[d] * corr);
            m = mn;
        }
    }
    if (active) {
        float inv = 1.0f / l;
        float* op = out + (size_t)(base + qi) * RN_D + hd * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) op[d] = acc[d] * inv;
    }
}

int launch_attention_f32(const PackInfo& pk, const float* qkv, int heads, float* out, hipStream_t s) {
    dim3 grid(pk.B, heads, (pk.T + 63) / 64);
    int hd = RN_D / heads;
    if (hd == 16) hipLaunchKernelGGL(k_attention_f32<16>, grid, dim3(64), 0, s, pk, qkv, heads, out);
    else if (hd == 32) hipLaunchKernelGGL(k_attention_f32<32>, grid, dim3(64), 0, s, pk, qkv, heads, out);
    else if (hd == 8) hipLaunchKernelGGL(k_attention_f32<8>, grid, dim3(64), 0, s, pk, qkv, heads, out);
    else if (hd == 64) hipLaunchKernelGGL(k_attention_f32<64>, grid, dim3(64), 0, s, pk, qkv, heads, out);
    else return 1;
    return 0;
}

// ------------------------------------------------------------------------------------------
// pack / unpack between the reference's padded layouts and the packed rows.
__global__ void k_unpack_nodes(PackInfo pk, const float* __restrict__ src, int ld, int D, float* __restrict__ dst,
                               int dst_ld, int dst_col0) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)pk.B * pk.T * D;
    if (id >= total) return;
    int d = (int)(id % D);
    size_t bt = id / D;
    int b = (int)(bt / pk.T), t = (int)(bt - (size_t)b * pk.T);
    dst[bt * dst_ld + dst_col0 + d] = t < pk.len[b] ? src[(size_t)(pk.cu[b] + t) * ld + d] : 0.f;
}
void launch_unpack_nodes_strided(const PackInfo& pk, const float* src, int ld, int D, float* dst, int dst_ld,
                                 int dst_col0, hipStream_t s) {
    size_t total = (size_t)pk.B * pk.T * D;
    hipLaunchKernelGGL(k_unpack_nodes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pk, src, ld, D, dst, dst_ld, dst_col0);
}
void launch_unpack_nodes(const PackInfo& pk, const float* src, int ld, int D, float* dst, hipStream_t s) {
    launch_unpack_nodes_strided(pk, src, ld, D, dst, D, 0, s);
}

__global__ void k_pack_nodes(PackInfo pk, const float* __restrict__ src, int D, float* __restrict__ dst, int ld) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)pk.B * pk.T * D;
    if (id >= total) return;
    int d = (int)(id % D);
    size_t bt = id / D;
    int b = (int)(bt / pk.T), t = (int)(bt - (size_t)b * pk.T);
    if (t < pk.len[b]) {
        dst[(size_t)(pk.cu[b] + t) * ld + d] = src[id];
        if (d == 0) pk.node_b[pk.cu[b] + t] = b;
    }
}
void launch_pack_nodes(const PackInfo& pk, const float* src, int D, float* dst, int ld, hipStream_t s) {
    size_t total = (size_t)pk.B * pk.T * D;
    hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pk, src, D, dst, ld);
}

// edges: (B,T,k,128) <-> packed (N_tot*k, 128); invalid slots and padded rows unpack to 0.
__global__ void k_unpack_edges(PackInfo pk, int k, const float* __restrict__ src, const int* __restrict__ nbr,
                               float* __restrict__ dst) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 per thread
    size_t total = (size_t)pk.B * pk.T * k * 32;
    if (id >= total) return;
    int q = (int)(id & 31);
    size_t row = id >> 5;
    int sl = (int)(row % k);
    size_t bt = row / k;
    int b = (int)(bt / pk.T), t = (int)(bt - (size_t)b * pk.T);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < pk.len[b]) {
        size_t pe = (size_t)(pk.cu[b] + t) * k + sl;
        if (!nbr || nbr[pe] >= 0) v = *reinterpret_cast<const float4*>(src + pe * RN_D + q * 4);
    }
    *reinterpret_cast<float4*>(dst + row * RN_D + q * 4) = v;
}
void launch_unpack_edges(const PackInfo& pk, int k, const float* src, const int* nbr, float* dst, hipStream_t s) {
    size_t total = (size_t)pk.B * pk.T * k * 32;
    hipLaunchKernelGGL(k_unpack_edges, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pk, k, src, nbr, dst);
}

__global__ void k_pack_edges(PackInfo pk, int k, const float* __restrict__ src, float* __restrict__ dst) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)pk.B * pk.T * k * 32;
    if (id >= total) return;
    int q = (int)(id & 31);
    size_t row = id >> 5;
    int sl = (int)(row % k);
    size_t bt = row / k;
    int b = (int)(bt / pk.T), t = (int)(bt - (size_t)b * pk.T);
    if (t < pk.len[b]) {
        size_t pe = (size_t)(pk.cu[b] + t) * k + sl;
        *reinterpret_cast<float4*>(dst + pe * RN_D + q * 4) = *reinterpret_cast<const float4*>(src + row * RN_D + q * 4);
    }
}
void launch_pack_edges(const PackInfo& pk, int k, const float* src, float* dst, hipStream_t s) {
    size_t total = (size_t)pk.B * pk.T * k * 32;
    hipLaunchKernelGGL(k_pack_edges, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pk, k, src, dst);
}

// edge_index (B,T,k) i64, local indices / -1  ->  packed global rows; any padded index (>= n_b)
// is the phantom neighbour (zero embedding) of that RNA.
__global__ void k_pack_index(PackInfo pk, int k, const int64_t* __restrict__ eidx, int* __restrict__ nbr) {
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)pk.B * pk.T * k;
    if (id >= total) return;
    int sl = (int)(id % k);
    size_t bt = id / k;
    int b = (int)(bt / pk.T), t = (int)(bt - (size_t)b * pk.T);
    int n = pk.len[b];
    if (t < n) {
        long long j = eidx[id];
        int v = j < 0 ? -1 : (j >= n ? pk.Nmax + b : pk.cu[b] + (int)j);
        nbr[(size_t)(pk.cu[b] + t) * k + sl] = v;
    }
}
void launch_pack_index(const PackInfo& pk, int k, const int64_t* eidx, int* nbr, hipStream_t s) {
    size_t total = (size_t)pk.B * pk.T * k;
    hipLaunchKernelGGL(k_pack_index, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, pk, k, eidx, nbr);
}

__global__ void k_transpose(const float* __restrict__ src, int ld_src, int rows, int cols, float* __restrict__ dst, int ld_dst) {
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= rows * cols) return;
    int r = id / cols, c = id - r * cols;
    dst[(size_t)c * ld_dst + r] = src[(size_t)r * ld_src + c];
}
void launch_transpose(const float* src, int ld_src, int rows, int cols, float* dst, int ld_dst, hipStream_t s) {
    int total = rows * cols;
    hipLaunchKernelGGL(k_transpose, dim3((total + 255) / 256), dim3(256), 0, s, src, ld_src, rows, cols, dst, ld_dst);
}

// ------------------------------------------------------------------------------------------
// decode: argmax + recovery counts (rnampnn.py:223-230; argmax of softmax = argmax of logits,
// first maximum wins as torch.argmax does), one wave per RNA.
__global__ void __launch_bounds__(64) k_argmax_recovery(const float* __restrict__ logits, const float* __restrict__ mask,
        const int32_t* __restrict__ labels, int T, int8_t* __restrict__ pred, int32_t* __restrict__ correct,
        int32_t* __restrict__ valid) {
    int b = blockIdx.x;
    int ok = 0, nv = 0;
    for (int t = threadIdx.x; t < T; t += 64) {
        size_t i = (size_t)b * T + t;
        if (mask[i] != 0.f) {
            const float4 v = *reinterpret_cast<const float4*>(logits + i * 4);
            int a = 0; float best = v.x;
            if (v.y > best) { best = v.y; a = 1; }
            if (v.z > best) { best = v.z; a = 2; }
            if (v.w > best) { best = v.w; a = 3; }
            if (pred) pred[i] = (int8_t)a;
            nv += 1;
            if (labels) ok += (labels[i] == a);
        } else if (pred) pred[i] = -1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ok += __shfl_xor(ok, o, 64); nv += __shfl_xor(nv, o, 64); }
    if (threadIdx.x == 0) { if (correct) correct[b] = ok; if (valid) valid[b] = nv; }
}
void launch_argmax_recovery(const float* logits, const float* mask, const int32_t* labels, int B, int T,
                            int8_t* pred, int32_t* correct, int32_t* valid, hipStream_t s) {
    hipLaunchKernelGGL(k_argmax_recovery, dim3(B), dim3(64), 0, s, logits, mask, labels, T, pred, correct, valid);
}

// sample(): independent categorical draw per position from softmax(logits / temperature).
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}
__global__ void k_sample(const float* __restrict__ logits, const float* __restrict__ mask, int B, int T, float inv_temp,
                         int n_samples, unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                         int8_t* __restrict__ out) {
    if (seed_dev) seed = *seed_dev;
    size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t per = (size_t)B * T;
    if (id >= per * n_samples) return;
    size_t i = id % per;
    if (mask[i] == 0.f) { out[id] = -1; return; }
    const float4 v = *reinterpret_cast<const float4*>(logits + i * 4);
    float z0 = v.x * inv_temp, z1 = v.y * inv_temp, z2 = v.z * inv_temp, z3 = v.w * inv_temp;
    float mx = fmaxf(fmaxf(z0, z1), fmaxf(z2, z3));
    float p0 = expf(z0 - mx), p1 = expf(z1 - mx), p2 = expf(z2 - mx), p3 = expf(z3 - mx);
    float tot = p0 + p1 + p2 + p3;
    unsigned long long bits = mix64(mix64((id + 1) * 0x9E3779B97F4A7C15ull + seed) ^ (seed * 0xD6E8FEB86659FD93ull));
    float u = (float)(bits >> 40) * (1.0f / 16777216.0f) * tot;     // uniform in [0, tot)
    int a = 3;
    if (u < p0) a = 0; else if (u < p0 + p1) a = 1; else if (u < p0 + p1 + p2) a = 2;
    out[id] = (int8_t)a;
}
void launch_sample(const float* logits, const float* mask, int B, int T, float temperature, int n_samples,
                   uint64_t seed, const uint64_t* seed_dev, int8_t* out, hipStream_t s) {
    size_t total = (size_t)B * T * n_samples;
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, logits, mask, B, T,
                       1.0f / temperature, n_samples, (unsigned long long)seed,
                       reinterpret_cast<const unsigned long long*>(seed_dev), out);
}
