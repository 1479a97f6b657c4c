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
// element conversion of packed edge rows (stage API / taps)
void launch_bf16_to_f32(const bf16_t* src, float* dst, size_t max_elems, const int* ntot, int per_row, hipStream_t s);
void launch_f32_to_bf16(const float* src, bf16_t* dst, size_t max_elems, const int* ntot, int per_row, hipStream_t s);
// node-level Linear on MFMA: Y = act([X | X2] . W^T + bias) (+ res);  W bf16 [N][K] row-major
void launch_gemm_bf16(const int* ntot, int mmax, const float* X, int ldx, int K1, const float* X2, int ldx2, int K2,
                      const bf16_t* W, const float* bias, int N, int act, const float* res, int ldres,
                      float* Y, int ldy, hipStream_t s);
void launch_edge_embed_bf16(const PackInfo& pk, int k, const float* geom, const int* nbr, const bf16_t* img,
                            const float* b0, const float* b1p, bf16_t* e, hipStream_t s);
void launch_mpnn_bf16(const PackInfo& pk, int k, bool do_edge, bool do_msg, const int* nbr, bf16_t* e,
                      const float* pq_e, const float* pq_m, MpnnWB we, MpnnWB wm, const float* h_in,
                      float* h_pre, float* msg_out, hipStream_t s);
