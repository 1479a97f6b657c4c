// Training-path building blocks (kernels_train.hip): f32, packed rows.
#pragma once
#include "rnampnn_internal.h"

struct TRows { const int* ntot; int mul; int maxrows; };   // row count = *ntot * mul (device side), maxrows = host upper bound

void t_gemm(const TRows& rows, const float* X, int ldx, int K, const float* Wt, int ldw, const float* bias, int N,
            float* Y, int ldy, int beta, hipStream_t s);                      // Y = beta*Y + X.Wt + bias   (Wt K-major, row stride ldw)
void t_gemm_tn(const TRows& rows, const float* A, int lda, int M, const float* B, int ldb, int K, float* dW, int ldw, hipStream_t s);  // dW += A^T B
void t_colsum(const TRows& rows, const float* A, int lda, int M, float* out, hipStream_t s);                // out += column sums
void t_gelu_fwd(const TRows& rows, const float* x, float* y, int D, hipStream_t s);
void t_gelu_bwd(const TRows& rows, const float* dy, const float* pre, float* dx, int D, hipStream_t s);     // dx = dy * gelu'(pre)
void t_add(const TRows& rows, const float* a, float* dst, int D, hipStream_t s);                            // dst += a
void t_edge_features(const PackInfo& pk, int k, const float* geom, const int* nbr, float* F, hipStream_t s);        // [E][96]
void t_edge_add_pq(const PackInfo& pk, int k, const int* nbr, const float* pq, float* pre, hipStream_t s);          // pre += P[i] + Q[j]
void t_edge_zero_invalid(const PackInfo& pk, int k, const int* nbr, float* x, hipStream_t s);
void t_edge_residual(const PackInfo& pk, int k, const int* nbr, const float* e_in, const float* pre2, float* e_out, hipStream_t s);
void t_seg_mean(const PackInfo& pk, int k, const int* nbr, const float* pre2, const float* h, float* out, hipStream_t s);
void t_seg_mean_bwd(const PackInfo& pk, int k, const int* nbr, const float* dagg, const float* pre2, float* dpre2, hipStream_t s);
void t_edge_res_bwd(const PackInfo& pk, int k, const int* nbr, const float* de, const float* pre2, float* dpre2, hipStream_t s);
void t_edge_pq_bwd(const PackInfo& pk, int k, const int* nbr, const float* dpre1, float* dpq, hipStream_t s);
void t_gn_bwd(const PackInfo& pk, const float* x, const float* dy, const float* scale, int t_tot, float* dx, float* dscale,
              float* dshift, hipStream_t s);
int  t_attention_bwd(const PackInfo& pk, const float* qkv, const float* dO, int heads, float* dqkv, float* stat, hipStream_t s);
void t_loss_grad(const PackInfo& pk, const float* logits, const int32_t* labels, float* dlogits, float* loss, hipStream_t s);
