// Internal declarations shared by the C-ABI layer (api.cpp) and the kernel translation units.
// Node tensors are PACKED inside the library: the valid residues of all RNAs of a batch are
// stored back to back (row p = cu[b] + t), so no kernel spends work on padding; `T` survives
// only as the GraphNormalization parameter T_tot (reference functional.py:33-38) and in the
// padded layouts of the API tensors, which pack/unpack kernels translate at the boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define RN_D 128            // hidden width of node and edge embeddings
#define RN_RAW 28           // node raw features
#define RN_RAWP 32          // ... padded to a multiple of 4 for the GEMM K loop
#define RN_ERAW 90          // edge raw features
#define RN_ERAWP 96
#define RN_GEOM 48          // per-residue geometry record: 21 coords, 15 unit bonds, 12 unit normals
#define RN_GEOMH 64         // the same record split by lane half for the bf16 edge-embedding kernel (kernels_f32.hip: geomh_record)
#define RN_KMAX 32

typedef unsigned short bf16_t;   // raw bf16 bits

// Kernels that need more than 64 KiB of dynamic LDS must be told so once PER DEVICE (the attribute lives with the
// device's code object): remembered per device id, thread-safe, grows monotonically.
#include <mutex>
struct DevAttr { std::mutex mu; size_t have[64] = {}; };
static inline void ensure_dyn_lds(const void* fn, size_t bytes, DevAttr& st) {
    if (bytes <= 64 * 1024) return;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(st.mu);
    size_t& h = st.have[dev & 63];
    if (h < bytes) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); h = bytes; }
}
// CU count of the CURRENT device (cached per device id)
static inline int rn_num_cus() {
    static int cus[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& n = cus[dev & 63];
    if (!n) {
        hipDeviceProp_t p;
        n = hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    return n;
}

struct PackInfo {            // device arrays describing the packed batch
    int* len;                // [B]    valid length n_b
    int* cu;                 // [B+1]  exclusive prefix sum; cu[B] = N_tot
    int* node_b;             // [Nmax] RNA id of packed row p
    int  B, T, Nmax;         // Nmax = upper bound for N_tot known on the host (B*T, or the exact total of a packed input)
    int  packed_in;          // 1: the coords tensor is packed like the internal rows (row cu[b]+t), no padded residues
};

// ---- kernels_f32.hip -------------------------------------------------------------------
struct ZeroRegions { void* ptr[8]; unsigned words[8]; int n; };       // 4-byte aligned regions, sizes in 32-bit words
void launch_zero_regions(const ZeroRegions& z, hipStream_t s);
void launch_zero_bytes(void* ptr, size_t bytes, hipStream_t s, int site = 0x80);                    // kernel, not a memset node (hipGraph-safe); ptr 16-byte aligned
void launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t s);   // kernel copy, 16-byte aligned, bytes % 16 == 0
void launch_lengths(const float* mask, const PackInfo& pk, hipStream_t s);
void launch_lengths_from_cu(const int32_t* cu_seqlens, const PackInfo& pk, hipStream_t s);
void launch_geom(const float* coords, const PackInfo& pk, float* raw_out, float* raw_p, float* geom, float* geomh, hipStream_t s);
int  launch_knn(const float* coords, const PackInfo& pk, int k, int* nbr, int64_t* edge_index_out, hipStream_t s);
void launch_edge_embed_f32(const PackInfo& pk, int k, const float* geom, const int* nbr,
                           const float* w0t, const float* b0, const float* w1t, const float* b1, int depth,
                           float* e, hipStream_t s);
struct MpnnW32 {             // one MLP of a ResMPNN layer in the f32 layouts
    const float* wc_t;       // [128][128]  e-part of the first Linear, K-major (transposed)
    const float* w2_t;       // [128][128]  second Linear, K-major (null when depth == 1)
    const float* b2;         // [128]
    int depth;
};
void launch_mpnn_f32(const PackInfo& pk, int k, bool do_edge, bool do_msg, const int* nbr,
                     float* e, const float* pq_e, const float* pq_m, MpnnW32 we, MpnnW32 wm,
                     const float* h_in, float* h_pre, float* msg_out, hipStream_t s);
// y = GraphNorm(x (+ add)) on packed rows; add may be null; y may alias x or add
void launch_graph_norm_packed(const PackInfo& pk, const float* x, const float* add, float* y, const float* scale,
                              const float* shift, int t_tot, hipStream_t s);
void launch_graph_norm_padded(const float* x, const float* mask, const float* scale, const float* shift,
                              int B, int T, int t_tot, int D, float* y, hipStream_t s);
// Y[p][0:N] = act(X[p][0:K1] . Wt[0:K1] + X2[p][0:K2] . Wt[K1:K1+K2] + bias) (+ res[p])   for p < *ntot
void launch_gemm_f32(const int* ntot, int mmax, const float* X, int ldx, int K1, const float* X2, int ldx2, int K2,
                     const float* Wt, const float* bias, int N, int act, const float* res, int ldres,
                     float* Y, int ldy, hipStream_t s);
int  launch_attention_f32(const PackInfo& pk, const float* qkv, int heads, float* out, hipStream_t s);
void launch_unpack_nodes(const PackInfo& pk, const float* src, int ld, int D, float* dst, hipStream_t s);
void launch_unpack_nodes_strided(const PackInfo& pk, const float* src, int ld, int D, float* dst, int dst_ld, int dst_col0, hipStream_t s);
void launch_pack_nodes(const PackInfo& pk, const float* src, int D, float* dst, int ld, hipStream_t s);
void launch_unpack_edges(const PackInfo& pk, int k, const float* src, const int* nbr, float* dst, hipStream_t s);
void launch_pack_edges(const PackInfo& pk, int k, const float* src, float* dst, hipStream_t s);
void launch_pack_index(const PackInfo& pk, int k, const int64_t* edge_index, int* nbr, hipStream_t s);
void launch_transpose(const float* src, int ld_src, int rows, int cols, float* dst, int ld_dst, hipStream_t s);
void launch_argmax_recovery(const float* logits, const float* mask, const int32_t* labels, int B, int T,
                            int8_t* pred, int32_t* correct, int32_t* valid, hipStream_t s);
void launch_sample(const float* logits, const float* mask, int B, int T, float temperature, int n_samples,
                   uint64_t seed, const uint64_t* seed_dev, int8_t* out, hipStream_t s);
