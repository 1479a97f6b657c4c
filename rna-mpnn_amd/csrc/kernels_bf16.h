// Fast path: bf16 MFMA kernels (kernels_bf16.hip).  Operands bf16, accumulation f32,
// edge tensor e stored bf16 in HBM; node tensors stay f32.
#pragma once
#include "rnampnn_internal.h"

struct MpnnWB {              // one MLP of a ResMPNN layer, fast-path layouts
    const bf16_t* img;       // MFMA A/B fragment image of (Wc, W2), see build_mlp_image
    const float* b2p;        // bias of Linear 1 in the kernel's output-channel order
};

// weight preparation (run once per load_state_dict)
void launch_convert_rows_bf16(const float* src, int ld_src, int rows, int cols, int cols_pad, bf16_t* dst, hipStream_t s);
void launch_build_mlp_image(const float* wc, int ld_wc, const float* w2, int ld_w2, const float* b2, int out_perm,
                            bf16_t* img, float* b2p, hipStream_t s);
void launch_build_embed_image(const float* w0, const float* w1, const float* b1, bf16_t* img, float* b1p, hipStream_t s);
// the bf16 edge tensor is stored fragment-major (8 KiB per 32-slot block, see kernels_bf16.hip);
// conversion to / from row-major f32 rows [(p*k + slot)][128] for taps and the stage API
void launch_efrag_to_rows(const bf16_t* ef, const int* ntot, int nmax, int k, float* rows, hipStream_t s);
void launch_rows_to_efrag(const float* rows, const int* ntot, int nmax, int k, const int* nbr, bf16_t* ef, hipStream_t s);
static inline size_t efrag_bytes(int nmax, int k) { int npb = k > 16 ? 1 : 32 / k; return (size_t)((nmax + npb - 1) / npb) * 8192; }
// node-level Linear on MFMA: Y = act([X | X2] . W^T + bias) (+ res);  W bf16 [N][K] row-major
void launch_gemm_bf16(const int* ntot, int mmax, const float* X, int ldx, int K1, const float* X2, int ldx2, int K2,
                      const bf16_t* W, const float* bias, int N, int act, const float* res, int ldres,
                      float* Y, int ldy, hipStream_t s);   // cols >= col_split -> bf16 Yb (if Yb)
void launch_edge_embed_bf16(const PackInfo& pk, int k, const float* geom, const int* nbr, const bf16_t* img,
                            const float* b0, const float* b1p, bf16_t* e, hipStream_t s);
// node tables (launch_node_update): p_* = h.Wa^T + b1, q_* = h.Wb^T, f16 [N+1][128], natural channel order (row Nmax of q_* = zeros)
void launch_mpnn_bf16(const PackInfo& pk, int k, bool do_edge, bool do_msg, const int* nbr, bf16_t* e,
                      const bf16_t* p_e, const bf16_t* q_e, const bf16_t* p_m, const bf16_t* q_m, MpnnWB we, MpnnWB wm,
                      float* agg, float* msg_out, bool edge1, const float* h_res, hipStream_t s);   // agg [N][128]: h_res + masked mean of the messages (h_res null: the mean alone); edge1: the edge MLP has one Linear

// round-4 form of the fused step (kernels_mpnn.hip: two waves per SIMD, no helper MFMAs; layer 1 with the edge embedding in front); launch_mpnn_bf16 routes to it when it covers the case
bool resmpnn_covers(int k, bool edge1, bool msg_out);
void launch_resmpnn_embed_bf16(const PackInfo& pk, int k, const int* nbr, bf16_t* e, const float* geomh, const bf16_t* ee_img, const float* ee_b0,
                               const float* ee_b1p, const bf16_t* p_m, const bf16_t* q_m, MpnnWB wm, float* agg, const float* h_res, hipStream_t s);
void launch_resmpnn_bf16(const PackInfo& pk, int k, bool do_edge, bool do_msg, const int* nbr, bf16_t* e, const bf16_t* p_e, const bf16_t* q_e,
                         const bf16_t* p_m, const bf16_t* q_m, MpnnWB we, MpnnWB wm, float* agg, const float* h_res, hipStream_t s);

// fused FFN chain  X -> Linear(K0,H)+GELU -> NH x [Linear(H,H)+GELU] -> Linear(H,NOUT)  (see kernels_bf16.hip)
void launch_build_chain_image(const float* wraw, int K_real, int K, int N, int n_real, int position, bf16_t* dst, hipStream_t s);   // position: 0 first, 1 hidden, 2 last Linear
static inline size_t chain_image_bytes(int K0, int H, int NH, int NOUT) {
    return ((size_t)(H / 32) * (K0 / 16) + (size_t)NH * (H / 32) * (H / 16) + (size_t)(NOUT / 32) * (H / 16)) * 1024;
}
int launch_ffn_chain(const int* ntot, int mmax, const float* X, int ldx, const float* X2, int ldx2, int K0, int H, int NH,
                     int NOUT, const bf16_t* img, const float* const* bias, float* Y, int ldy, int n_valid, hipStream_t s);

// fused node update: h' = GraphNorm(x + add) (scale == null: h' = x + add, no norm), then the [P | Q] projections of up to two first
// Linears.  Both tables are f16 of (a P), (a Q) in NATURAL channel order, [N+1][128]; coef = scratch [B][256]
struct NodeJob {
    const bf16_t* img = nullptr;     // launch_build_pq_image
    const float* bias = nullptr;     // its b1p
    bf16_t* p = nullptr;             // P = h . Wa^T + b1
    bf16_t* q = nullptr;             // Q = h . Wb^T
    int p_efrag = 0;                 // the image's P rows are in ch_efrag order (depth-1 edge MLP)
};
void launch_build_pq_image(const float* w0, const float* b1, int efrag, bf16_t* dst, float* b1p, hipStream_t s);   // b1p: bias in the P-row order
void launch_node_update(const PackInfo& pk, const float* x, const float* add, const float* scale, const float* shift, int t_tot,
                        float* coef, float* h_out, int njobs, const NodeJob& job0, const NodeJob& job1, hipStream_t s);

// MFMA attention over valid keys (head dim 16); returns 1 when the shape is not covered (caller uses the f32 kernel)
int launch_attention_bf16(const PackInfo& pk, const float* qkv, int heads, float* out, hipStream_t s);

// one whole attention layer of RNABert per RNA (QKV, attention over valid keys, out-projection + residual, GraphNorm with T_tot), in
// place on x; returns 1 (nothing launched) unless heads == 8 and every RNA of the batch has at most 144 residues (pk.T <= 144)
void launch_build_attn_images(const float* wqkv, const float* wout, bf16_t* img_qkv, bf16_t* img_out, hipStream_t s);   // 96 KiB + 32 KiB
int launch_attn_layer_rna(const PackInfo& pk, float* x, const bf16_t* wqkv, const float* bqkv, const bf16_t* wout, const float* bout,
                          int heads, const float* gscale, const float* gshift, int t_tot, hipStream_t s);
