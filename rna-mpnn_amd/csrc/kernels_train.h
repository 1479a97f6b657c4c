// Training-path building blocks (kernels_train.hip): f32, packed rows.
#pragma once
#include "rnampnn_internal.h"

struct TRows { const int* ntot; int mul; int maxrows; };   // row count = *ntot * mul (device side), maxrows = host upper bound

// Dropout of the training path (the reference: nn.Dropout(p) after every GELU, mpnn.py:140,150, feature.py:200,
// functional.py:69,124,184, and on the attention probabilities, functional.py:109).  The keep decision of one element is a
// pure function of (seed, site, element index), restated by the CPU oracle, so that the HIP forward / backward and the oracle's
// autograd see THE SAME mask.  One 32-bit counter hash (murmur3 finaliser) serves the PAIR of elements (2P, 2P + 1): the low
// 16 bits decide the even element, the high 16 bits the odd one (keep iff >= thresh = round(p * 65536)): the kernels that own 8
// consecutive channels of a row pay 4 hashes for 8 decisions.  element index = row * D + channel with rows in the packed order
// (node row p, edge row p*k + slot); attention: ((query row * heads + head) << 13) + key.
// seed_dev (optional): the seed is read from DEVICE memory at kernel time instead (a hipGraph that captured a training step then draws
// fresh masks on every replay once the caller has updated *seed_dev - the decode path's rnampnn_sample_dev_seed idea).
struct TDrop { unsigned long long seed; unsigned thresh; float scale; const unsigned long long* seed_dev; };   // scale = 1/(1-p)
static inline TDrop t_drop(float p, unsigned long long seed, const unsigned long long* seed_dev = nullptr) {
    TDrop d;
    d.seed = seed;
    d.seed_dev = seed_dev;
    d.thresh = p > 0.f ? (unsigned)(p * 65536.0f + 0.5f) : 0u;
    d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    return d;
}

void t_gemm(const TRows& rows, const float* X, int ldx, int K, const float* Wt, int ldw, const float* bias, int N,
            float* Y, int ldy, int beta, hipStream_t s);                      // Y = beta*Y + X.Wt + bias   (Wt K-major, row stride ldw)
struct TScratch { float* p; size_t floats; };                // partial results of the ordered two-stage reductions
// deferred, batched reductions of one backward (kernels_train.hip: RedQueue): between red_begin and red_end the producers' ordered reductions are
// recorded and run many per launch; red_flush makes everything recorded so far final (before a gradient chunk's event)
void red_begin(const TScratch& sc, hipStream_t s);
void red_flush();
void red_end();
TScratch red_acquire(const TScratch& sc);
void t_gemm_tn(const TRows& rows, const float* A, int lda, int M, const float* B, int ldb, int K, float* dW, int ldw,
               const TScratch& sc, hipStream_t s);                                                           // dW += A^T B
// bf16-mixed MFMA versions (kernels_train.hip, second half).  _nt / _nn return false when the shape is not covered
// (K not a multiple of 16, unaligned rows): the caller then uses the f32 kernel.
bool tm_gemm_nt(const TRows& rows, const float* X, int ldx, int K, const float* W, int ldw, const float* bias, int N, float* Y,
                int ldy, int beta, bool actA, const TDrop& dr, unsigned site, hipStream_t s);
bool tm_gemm_nn(const TRows& rows, const float* X, int ldx, int K, const float* W, int ldw, const float* bias, int N, float* Y,
                int ldy, int beta, const float* epi_pre, int ld_epi, const TDrop& dr, unsigned site, hipStream_t s);
// dbias (optional): += column sums of A, computed from the tiles the kernel stages anyway (the bias gradient of the same Linear)
void tm_gemm_tn(const TRows& rows, const float* A, int lda, int M, const float* B, int ldb, int K, float* dW, int ldw,
                const TScratch& sc, bool actB, const TDrop& dr, unsigned site, float* dbias, hipStream_t s);
void t_colsum(const TRows& rows, const float* A, int lda, int M, float* out, const TScratch& sc, hipStream_t s);   // out += column sums
void t_gelu_fwd(const TRows& rows, const float* x, float* y, int D, const TDrop& dr, unsigned site, hipStream_t s);   // y = drop(gelu(x))
void t_gelu_bwd(const TRows& rows, const float* dy, const float* pre, float* dx, int D, const TDrop& dr, unsigned site,
                hipStream_t s);                                                                             // dx = dy * mask * gelu'(pre)
void t_add(const TRows& rows, const float* a, float* dst, int D, hipStream_t s);                            // dst += a
void t_edge_features(const PackInfo& pk, int k, const float* geom, const int* nbr, float* F, hipStream_t s);        // [E][96]
void t_edge_add_pq(const PackInfo& pk, int k, const int* nbr, const float* pq, float* pre, hipStream_t s);          // pre += P[i] + Q[j]
void t_edge_zero_invalid(const PackInfo& pk, int k, const int* nbr, float* x, hipStream_t s);
void t_edge_residual(const PackInfo& pk, int k, const int* nbr, const float* e_in, const float* pre2, float* e_out,
                     const TDrop& dr, unsigned site, hipStream_t s);
void t_seg_mean(const PackInfo& pk, int k, const int* nbr, const float* pre2, const float* h, float* out,
                const TDrop& dr, unsigned site, hipStream_t s);
void t_seg_mean_bwd(const PackInfo& pk, int k, const int* nbr, const float* dagg, const float* pre2, float* dpre2,
                    const TDrop& dr, unsigned site, hipStream_t s);
void t_edge_res_bwd(const PackInfo& pk, int k, const int* nbr, const float* de, const float* pre2, float* dpre2,
                    const TDrop& dr, unsigned site, hipStream_t s);
// reverse adjacency (deg/start/fill: [Nmax+1] ints, list: [Nmax*k] ints) and the gather-form backward of P[i] + Q[j]
void t_build_reverse(const PackInfo& pk, int k, const int* nbr, int* deg, int* start, int* fill, int* list, int* tmp, hipStream_t s);   // tmp: [Nmax*k] ints of scratch
void t_edge_pq_bwd(const PackInfo& pk, int k, const float* dpre1, const int* start, const int* list, float* dpq, hipStream_t s);
void t_gn_bwd(const PackInfo& pk, const float* x, const float* dy, const float* scale, int t_tot, float* dx, float* dscale,
              float* dshift, const TScratch& sc, hipStream_t s);
// stat [N][heads][3]: the forward writes (row max, normaliser) per (query, head); the backward reads them and adds delta (it is a TAPE: one per attention layer)
int  t_attention_fwd(const PackInfo& pk, const float* qkv, int heads, float* out, float* stat, const TDrop& dr, unsigned site, hipStream_t s);
int  t_attention_bwd(const PackInfo& pk, const float* qkv, const float* O, const float* dO, int heads, float* dqkv, float* stat,
                     const TDrop& dr, unsigned site, hipStream_t s);
// the same on MFMA for the bf16-mixed trainer (head dim 16): returns 1 (nothing launched) when the shape is not covered
int  te_attention_fwd(const PackInfo& pk, const float* qkv, int heads, float* out, float* stat, const TDrop& dr, unsigned site, hipStream_t s);
int  te_attention_bwd(const PackInfo& pk, const float* qkv, const float* O, const float* dO, int heads, float* dqkv, float* stat,
                      const TDrop& dr, unsigned site, hipStream_t s);
// loss = mean_valid CE(softmax(logits), label) and d loss / d logits (packed rows)
void t_pack_dlogits(const PackInfo& pk, const float* dlogits_padded, float* dlogits_p, hipStream_t s);
void t_loss_grad(const PackInfo& pk, const float* logits, const int32_t* labels, float* dlogits, float* loss, const TScratch& sc,
                 hipStream_t s);
void t_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd,
                 int step, hipStream_t s);

// ---- bf16-STORAGE edge kernels of the bf16-mixed trainer (kernels_train.hip, last part).  Under the reference's bf16 autocast
// (rnampnn/utils/train.py:109) every nn.Linear output - hence the whole per-edge chain e, pre-activations and their gradients - is a
// bf16 tensor; here all [E][128] tensors of the ResMPNN layers are bf16 row-major in HBM (half the bytes of the f32 tape), f32
// only inside the kernels (MFMA accumulators, GELU, sums).  `tb16` = raw bf16 bits, round-to-nearest-even on store.
typedef unsigned short tb16;
struct EFuse {                       // optional epilogue fusions of te_gemm
    const tb16* P; const tb16* Q;    // v += P[row / k] + Q[nbr[row]]       (bf16 [N+1][128] each; row `zero_row` of Q = zeros)
    const int* nbr; int k; int zero_row;
    const tb16* res_in; tb16* res_out;   // res_out = res_in + (nbr[row] >= 0 ? drop(gelu(v), site2) : 0)     (edge update, mpnn.py:250-262)
    unsigned site2;
};
// Y[R][128] (bf16) = [beta Y] + actA(X)[R][128] . W' + bias, [* gelu'(epi_pre) * mask(site)]; W' = W^T (w_rows: W [128][ldw] as
// nn.Linear stores it) or W (W [128][ldw] k-major).  X is bf16 (x_bf16) or f32, row stride ldx.  Returns false (nothing launched) for a
// combination of options that is not instantiated.
bool te_gemm(const TRows& rows, const void* X, bool x_bf16, int ldx, const float* W, int ldw, bool w_rows, const float* bias, tb16* Y,
             int beta, bool actA, const tb16* epi_pre, const EFuse* fuse, const TDrop& dr, unsigned site, hipStream_t s, int kvalid = 128);   // kvalid: live columns of X (the rest is zero padding)
// dW[128][ldw] += A^T . actB(B), dbias += colsum(A)     (A, B bf16 [R][128])
void te_gemm_tn(const TRows& rows, const tb16* A, const tb16* B, float* dW, int ldw, const TScratch& sc, bool actB, const TDrop& dr,
                unsigned site, float* dbias, hipStream_t s, int cols_keep = 128);      // cols_keep: live columns of B
// the node-side GEMMs of a factored first Linear, each pair as one launch: P / Q tables, [dWa ; dWb] (+ db1), dh += dP Wa + dQ Wb
void te_gemm_pq(const TRows& rows, const float* h, const float* w0, const float* b1, tb16* Pt, tb16* Qt, hipStream_t s);
void tm_gemm_tn_pq(const TRows& rows, const float* dpq, const float* h, float* gw0, float* db1, const TScratch& sc, hipStream_t s);
bool tm_gemm_nn_pq(const TRows& rows, const float* dpq, const float* w0, float* dh, hipStream_t s);
// forward of a depth-2 per-edge MLP in one kernel (the hidden activation stays in registers; pre1 / pre2 written once as the tape;
// pre1 = null: not kept)
void te_mlp2_fwd(const TRows& rows, const tb16* X, const float* W1, int ldw1, const float* W2, int ldw2, const float* bias2, tb16* pre1,
                 tb16* pre2, const EFuse& f, const TDrop& dr, unsigned site, hipStream_t s, bool g2tape = false);   // g2tape (edge update only): pre2 receives gelu'(pre2) * mask(site2)
// fused pair of a first Linear's backward: dW += dY^T X, DE += dY . W   (one pass over dY)
void te_gemm_bwd1(const TRows& rows, const tb16* dY, const tb16* X, tb16* DE, const float* W, int ldw, float* dW, int ldw_out,
                  const TScratch& sc, hipStream_t s);
// fused pair of a depth-2 MLP's backward: dW += dY^T drop(gelu(PRE)), dbias += colsum(dY), DX = (dY . W) gelu'(PRE) mask   (one pass over dY and PRE)
// `from` (optional): d pre2 is formed on the fly while the tile is staged - mode 1: dY = d e_out, d pre2 = valid ? dY gelu'(pre2) mask(site2) : 0
// (the edge update's residual backward); mode 2: d pre2 = valid ? dagg[row / k] inv_cnt[row / k] gelu'(pre2) mask(site2) : 0 (the message mean's)
// g2tape: the `pre2` tensor holds gelu'(pre2) * mask(site2) as the forward left it (k_emm_fwd2 / k_eseg_mean, below) instead of pre2: the staging pass
// multiplies, where it evaluated a sigmoid, an exp2 and a dropout hash per element
struct EBwd2Src { int mode; const tb16* pre2; const int* nbr; const float* dagg; const float* inv_cnt; int k; unsigned site2; int g2tape; };
void te_gemm_bwd1x2(const TRows& rows, const tb16* dY1, const tb16* dY2, const tb16* X, tb16* DE, const float* W1, const float* W2, int ldw,
                    float* dW1, float* dW2, int ldw_out, const TScratch& sc, hipStream_t s);
void te_gemm_bwd2(const TRows& rows, const tb16* dY, const tb16* PRE, tb16* DX, const float* W, int ldw, float* dW, int ldw_out,
                  const TScratch& sc, const TDrop& dr, unsigned site, float* dbias, hipStream_t s, const EBwd2Src* from = nullptr);
void te_inv_count(const PackInfo& pk, int k, const int* nbr, float* inv_cnt, hipStream_t s);   // 1 / max(#valid slots, 1) per residue
// g2_out (optional, may alias pre2): gelu'(pre2) * mask(site) per element - what the message MLP's backward needs of pre2
void te_seg_mean(const PackInfo& pk, int k, const int* nbr, const tb16* pre2, const float* h, float* out, const TDrop& dr, unsigned site, hipStream_t s,
                 tb16* g2_out = nullptr);
void te_seg_mean_bwd(const PackInfo& pk, int k, const int* nbr, const float* dagg, const tb16* pre2, tb16* dpre2, const TDrop& dr, unsigned site, hipStream_t s);
void te_edge_res_bwd(const PackInfo& pk, int k, const int* nbr, const tb16* de, const tb16* pre2, tb16* dpre2, const TDrop& dr, unsigned site, hipStream_t s);
void te_edge_pq_bwd(const PackInfo& pk, int k, const tb16* dpre1, const int* start, const int* list, float* dpq, hipStream_t s);
void te_zero_invalid(const PackInfo& pk, int k, const int* nbr, tb16* x, hipStream_t s);
void te_edge_features(const PackInfo& pk, int k, const float* geom, const int* nbr, tb16* F, hipStream_t s);          // raw edge features, bf16 [E][128] (90 live columns)
void te_edge_act(const PackInfo& pk, int k, const int* nbr, const tb16* pre, tb16* out, const TDrop& dr, unsigned site, hipStream_t s);   // out = valid ? drop(gelu(pre)) : 0
void te_gelu_fwd_out(const TRows& rows, const float* x, tb16* y, int D, const TDrop& dr, unsigned site, hipStream_t s);      // y = bf16(drop(gelu(x)))
void te_gelu_bwd_in(const TRows& rows, const tb16* dy, const float* pre, float* dx, int D, const TDrop& dr, unsigned site, hipStream_t s);   // dx = dy gelu'(pre) mask

// ---- cache of prebuilt bf16 fragment images of the 128 x 128 weight blocks the weights-resident GEMMs use (kernels_train.hip):
// bind it for the calling thread, refresh at the start of a training forward (one launch rebuilds every image registered so far).
struct WImageCache;
WImageCache* t_wimg_create(int capacity);
void t_wimg_destroy(WImageCache* c);
void t_wimg_bind(WImageCache* c);            // null: kernels build their images themselves
void t_wimg_clear(WImageCache* c);           // the weights moved (new arena)
void t_wimg_refresh(WImageCache* c, hipStream_t s);
int t_wimg_pending(const WImageCache* c);     // blocks registered since the last refresh (their kernels still build their own image)
