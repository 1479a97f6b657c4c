/*
 * rnampnn_hip.h - C ABI of the MI355X-native RNA-MPNN forward path (librnampnn_hip.so).
 *
 * The reference (givemeone1astkiss/RNA-MPNN) has no FFI: its boundary for this path is the
 * Python module surface of `rnampnn.model`.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference root).  The host-side mirror in
 * `rna-mpnn_amd/rnampnn/model/` binds these symbols with ctypes and re-exposes the
 * reference's class/method names; INTEGRATION.md shows the stub a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; all tensor pointers are DEVICE pointers (HBM) unless
 *     a parameter says "host"; tensors are dense row-major in the reference's layouts:
 *       coords (B,T,7,3) f32, mask (B,T) f32 0/1 prefix masks as produced by the reference
 *       collate (rnampnn/utils/data.py:110-142), logits (B,T,4) f32, edge_index (B,T,k) i64.
 *   - inputs are const, outputs/workspace are caller-owned; the library keeps no reference
 *     to them after return.  Weights are copied into library-owned HBM at set time.
 *   - every launch goes to the caller's stream (`stream` = hipStream_t passed as void*), is
 *     asynchronous and never synchronises the device (hipGraph-capturable).
 *   - return value: 0 on success, an RNAMPNN_ERR_* code otherwise;
 *     rnampnn_last_error() gives the message of the calling thread's last failure.
 *   - one handle may be used by one host thread at a time (the reference: one Python
 *     thread per process, one process per GPU).
 */
#ifndef RNAMPNN_HIP_H
#define RNAMPNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNAMPNN_OK               0
#define RNAMPNN_ERR_BAD_ARG      1  /* null pointer, negative size ...            -> ValueError        */
#define RNAMPNN_ERR_UNSUPPORTED  2  /* hyper-parameter outside kernel support     -> NotImplementedError */
#define RNAMPNN_ERR_T_GT_P       3  /* max_len > padding_len (functional.py:155)  -> RuntimeError      */
#define RNAMPNN_ERR_K_TOO_LARGE  4  /* num_res_neighbours > RNAMPNN_KMAX          -> NotImplementedError */
#define RNAMPNN_ERR_WORKSPACE    5  /* workspace smaller than rnampnn_workspace_bytes -> RuntimeError  */
#define RNAMPNN_ERR_WEIGHTS      6  /* unknown key / wrong shape / weight not set -> KeyError          */
#define RNAMPNN_ERR_HIP          7  /* HIP runtime error (message carries hipGetErrorString)           */

#define RNAMPNN_KMAX 32             /* neighbours per residue supported by the edge kernels */

#define RNAMPNN_PREC_F32  0         /* exact-f32 kernels (parity grade: |dlogit| <= 1e-4)   */
#define RNAMPNN_PREC_BF16 1         /* bf16 MFMA operands, f32 accumulate, bf16 edge tensor */

typedef struct rnampnn_ctx* rnampnn_handle;

/* Hyper-parameters of RNAMPNN.__init__ that shape the network (rnampnn/model/rnampnn.py:19-47).
 * Atom counts are fixed at the reference defaults 7/6/6 (28 node and 90 edge raw features). */
typedef struct RnaMpnnConfig {
    int32_t num_res_neighbours;
    int32_t res_embedding_dim;            /* must be 128 */
    int32_t num_embedding_attn_layers;
    int32_t num_embedding_heads;
    int32_t embedding_ffn_dim;
    int32_t num_embedding_ffn_layers;
    int32_t res_edge_embedding_dim;       /* must be 128 */
    int32_t depth_res_edge_feature;       /* 1..2 */
    int32_t num_res_mpnn_layers;
    int32_t depth_res_mpnn;               /* 1..2 */
    int32_t num_mpnn_edge_layers;         /* 1..2 */
    int32_t padding_len;
    int32_t num_post_fusion_attn_layers;
    int32_t num_post_fusion_heads;
    int32_t post_fusion_ffn_dim;
    int32_t num_post_fusion_ffn_layers;
    int32_t num_raw_ffn_dim;
    int32_t num_raw_ffn_layers;
    int32_t raw_embedding_dim;            /* must be 128 */
    int32_t readout_hidden_dim;
    int32_t num_readout_layers;
    int32_t precision;                    /* RNAMPNN_PREC_* */
} RnaMpnnConfig;

/* Inputs, outputs and optional intermediate taps of one forward pass.  Null taps are skipped.
 * All taps are written in the reference's padded layouts with the reference's padding values. */
typedef struct RnaMpnnForwardIO {
    const float* coords;      /* (B,T,7,3) */
    const float* mask;        /* (B,T)     */
    int32_t B, T;
    int32_t T_norm;           /* node-axis length seen by GraphNormalization (functional.py:33-38);
                                 0 = T.  A data-parallel shard passes the GLOBAL batch max_len here. */
    int32_t stop_after;       /* 0 = whole forward; 1 = stop after ResFeature.forward (feature.py:573-592) */
    float*   logits;          /* (B,T,4)    RNAMPNN.forward          rnampnn.py:161-185 (required if stop_after==0) */
    float*   embedding;       /* (B,T,256)  RNAMPNN.embedding        rnampnn.py:269-278 */
    int64_t* edge_index;      /* (B,T,k)    ResFeature._get_res_graph feature.py:205-256 */
    float*   raw;             /* (B,T,28)   ResFeature._res_embedding feature.py:531-535 */
    float*   h0;              /* (B,T,128)  node embedding after ResFeature.graph_norm feature.py:591 */
    float*   e0;              /* (B,T,k,128) ResFeature._res_edge_embedding feature.py:540-571 */
    int32_t  tap_layer;       /* 1-based ResMPNN layer whose outputs go to h_layer/e_layer; 0 = none */
    float*   h_layer;         /* (B,T,128)   ResMPNN.forward h  mpnn.py:283-294 */
    float*   e_layer;         /* (B,T,k,128) ResMPNN.forward e (valid edges; invalid slots are 0) */
    float*   h_post;          /* (B,T,128)  RNABert post_fusion      functional.py:161-172 */
    float*   raw_emb;         /* (B,T,128)  RawFFN                    functional.py:200-202 */
} RnaMpnnForwardIO;

/* -- lifetime / weights ------------------------------------------------------------------ */
/* RNAMPNN.__init__ (rnampnn.py:94-134): validates hyper-parameters, allocates weight storage. */
int rnampnn_create(const RnaMpnnConfig* cfg, rnampnn_handle* out);
int rnampnn_destroy(rnampnn_handle h);
/* nn.Module.load_state_dict for one entry: `key` is the reference state_dict key (SURVEY.md
 * row A1), `data` an f32 tensor with `numel` elements on the device (is_host=0) or host (1). */
int rnampnn_set_weight(rnampnn_handle h, const char* key, const float* data, int64_t numel,
                       int32_t is_host, void* stream);
/* Number of state-dict entries the configuration expects; key/numel of entry i (host strings). */
int rnampnn_num_weights(rnampnn_handle h);
int rnampnn_weight_info(rnampnn_handle h, int32_t i, const char** key, int64_t* numel);
/* Builds the kernel-side layouts (transposes, bf16 fragment images) from the weights set so far. */
int rnampnn_finalize_weights(rnampnn_handle h, void* stream);

/* -- forward ----------------------------------------------------------------------------- */
size_t rnampnn_workspace_bytes(rnampnn_handle h, int32_t B, int32_t T);
/* RNAMPNN.forward / RNAMPNN.embedding / ResFeature.forward, by `io->stop_after` and taps. */
int rnampnn_forward(rnampnn_handle h, const RnaMpnnForwardIO* io, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Packed (var-len) form of the same forward - SURVEY.md section 8 row F1: the reference's collate pads every
 * RNA to the batch max_len (rnampnn/utils/data.py:110-142); here the caller hands over the valid residues
 * only, back to back: coords_packed (N_total,7,3) f32, cu_seqlens (B+1) i32 on the device (exclusive prefix
 * sum of the lengths), T_max = longest RNA (host value; decides the phantom-edge rule exactly as a batch
 * padded to T_max would), T_norm as above (0 = T_max).  Outputs are packed rows as well:
 * logits_packed (N_total,4), embedding_packed (N_total,256); either may be null. */
size_t rnampnn_workspace_bytes_packed(rnampnn_handle h, int32_t B, int32_t N_total);
int rnampnn_forward_packed(rnampnn_handle h, const float* coords_packed, const int32_t* cu_seqlens, int32_t B,
                           int32_t N_total, int32_t T_max, int32_t T_norm, float* logits_packed,
                           float* embedding_packed, void* workspace, size_t workspace_bytes, void* stream);

/* -- stage entry points (parity tests and the standalone reference classes) -------------- */
/* ResMPNN.forward / ResMPNN.message (mpnn.py:154-194, 267-294) for layer `layer` (0-based) on
 * caller-supplied h (B,T,128), e (B,T,k,128), edge_index (B,T,k) i64.  msg_out (B,T,k,128),
 * h_out, e_out are optional (null = skip). */
int rnampnn_mpnn_layer(rnampnn_handle h, int32_t layer, const float* h_in, const float* e_in,
                       const int64_t* edge_index, const float* mask, int32_t B, int32_t T, int32_t T_norm,
                       float* msg_out, float* h_out, float* e_out,
                       void* workspace, size_t workspace_bytes, void* stream);
/* GraphNormalization.forward (functional.py:18-48); T_tot = node-axis length entering the variance. */
int rnampnn_graph_norm(const float* x, const float* mask, const float* scale, const float* shift,
                       int32_t B, int32_t T, int32_t T_tot, int32_t D, float* y, void* stream);
/* RNABert.forward (functional.py:161-172): which = 0 res_feature.res_embedding, 1 post_fusion. */
int rnampnn_rnabert(rnampnn_handle h, int32_t which, const float* x, const float* mask, int32_t B, int32_t T,
                    float* y, void* workspace, size_t workspace_bytes, void* stream);
/* RawFFN.forward (functional.py:200-202): raw (B,T,28) -> (B,T,128). */
int rnampnn_raw_ffn(rnampnn_handle h, const float* raw, const float* mask, int32_t B, int32_t T, int32_t T_norm,
                    float* y, void* workspace, size_t workspace_bytes, void* stream);
/* Readout.forward (functional.py:86-90): emb (B,T,256) -> logits (B,T,4). */
int rnampnn_readout(rnampnn_handle h, const float* emb, const float* mask, int32_t B, int32_t T,
                    float* logits, void* workspace, size_t workspace_bytes, void* stream);

/* -- decode ------------------------------------------------------------------------------ */
/* argmax decode + recovery (rnampnn.py:223-230, utils/train.py:18-21): per-RNA counts of correct
 * and valid positions, pred (B,T) int8 (-1 on padding).  labels (B,T) int32 class ids. */
int rnampnn_argmax_recovery(const float* logits, const float* mask, const int32_t* labels,
                            int32_t B, int32_t T, int8_t* pred, int32_t* correct, int32_t* valid, void* stream);
/* sample(): independent categorical draw per position from softmax(logits / temperature)
 * (no reference counterpart).  out (n_samples,B,T) int8, -1 on padding.
 * Counter-based RNG: draw = f(seed, sample, b, t) - reproducible and graph-replay safe. */
int rnampnn_sample(const float* logits, const float* mask, int32_t B, int32_t T, float temperature,
                   int32_t n_samples, uint64_t seed, int8_t* out, void* stream);
/* Same, with the seed read from DEVICE memory at kernel time: a hipGraph that captured this launch
 * draws fresh samples on every replay once the caller has updated *seed_device (BASELINE config 5:
 * "hipGraph-captured decode step"). */
int rnampnn_sample_dev_seed(const float* logits, const float* mask, int32_t B, int32_t T, float temperature,
                            int32_t n_samples, const uint64_t* seed_device, int8_t* out, void* stream);

/* -- training ---------------------------------------------------------------------------- */
/* The training surface of RNAMPNN (rnampnn.py:187-207 + Lightning's loss.backward()):
 *   rnampnn_train_forward  - `self(coords, mask)` in train mode.  Dropout with probability `dropout` after every GELU
 *       (mpnn.py:140,150; feature.py:200; functional.py:69,124,184) and on the attention probabilities
 *       (nn.MultiheadAttention(dropout=...), functional.py:109).  The reference draws its masks from torch's global RNG;
 *       here the keep decision of an element is a pure function of (seed, site, element index) - one 32-bit counter hash per
 *       PAIR of elements, its low / high 16 bits deciding the even / odd one (csrc/kernels_train.h: TDrop), restated by
 *       oracle/rnampnn_oracle.py - so a step is reproducible and testable against autograd.  A call accepts
 *       B*T*k < 2^26 edge rows and B*T < 2^23 residues (32-bit pair indices); larger batches return RNAMPNN_ERR_BAD_ARG.  The
 *       activations the backward needs (the "tape") stay in `workspace`, which must be left untouched until
 *       rnampnn_train_backward has run.  logits (B,T,4) out; *tape_id receives the identity of this tape (ids grow
 *       monotonically per handle, 0 is never issued).  A later forward / loss_and_grad into the SAME workspace destroys the
 *       tape; forwards into different workspaces may be outstanding together (gradient accumulation over micro-batches,
 *       `(loss1 + loss2).backward()`).
 *   rnampnn_train_backward - gradient of every parameter from dlogits (B,T,4) = d loss / d logits of ANY loss the
 *       caller built on the logits (torch autograd: torch.autograd.Function in rnampnn/model/rnampnn.py), for the tape
 *       `tape_id` living in `workspace`; RNAMPNN_ERR_BAD_ARG when that tape has been overwritten or belongs to another
 *       workspace / shape - never a silent backward through another forward's activations.  A tape may be walked more than once
 *       (retain_graph).  accumulate = 0 overwrites `grad`, 1 adds to it.
 *   rnampnn_loss_and_grad  - both in one call around the reference loss: cross_entropy(softmax(logits)[valid], label)
 *       (softmax twice, rnampnn.py:151-154), mean over valid nucleotides.  labels (B,T) int32 class ids (ignored on
 *       padding); loss: device scalar; logits optional; grad overwritten.
 *   grad: flat f32 buffer of rnampnn_grad_numel() elements; parameter i of rnampnn_weight_info() lies at
 *   rnampnn_weight_offset(i) - one buffer = ONE RCCL all-reduce per step (what Lightning DDP does for the reference,
 *   rnampnn/utils/train.py:106-117).  Every cross-workgroup sum of the backward is an ordered two-stage reduction (no
 *   float atomics): the same inputs give bit-identical gradients. */
size_t  rnampnn_train_workspace_bytes(rnampnn_handle h, int32_t B, int32_t T);
int64_t rnampnn_grad_numel(rnampnn_handle h);
int     rnampnn_weight_offset(rnampnn_handle h, int32_t i, int64_t* offset);
#define RNAMPNN_TRAIN_F32        0  /* exact-f32 GEMMs: the parity-grade path (gradients vs oracle autograd to 2e-3)         */
#define RNAMPNN_TRAIN_BF16_MIXED 1  /* the reference's `bf16-mixed` (rnampnn/utils/train.py:109): GEMM operands bf16 on MFMA, */
                                    /* f32 accumulate; as under autocast, every per-edge activation and its gradient is a bf16 */
                                    /* tensor in HBM (the tape), node-sized tensors and all reductions stay f32                */
int     rnampnn_train_forward(rnampnn_handle h, const float* coords, const float* mask, int32_t B, int32_t T,
                              int32_t T_norm, float dropout, uint64_t seed, int32_t flags, float* logits,
                              void* workspace, size_t workspace_bytes, void* stream, int64_t* tape_id);
int     rnampnn_train_backward(rnampnn_handle h, int64_t tape_id, const float* dlogits, int32_t B, int32_t T, int32_t accumulate,
                               float* grad, void* workspace, size_t workspace_bytes, void* stream);
int     rnampnn_loss_and_grad(rnampnn_handle h, const float* coords, const float* mask, const int32_t* labels,
                              int32_t B, int32_t T, int32_t T_norm, float dropout, uint64_t seed, int32_t flags,
                              float* loss, float* logits, float* grad, void* workspace, size_t workspace_bytes, void* stream);
/* Overlap of the data-parallel gradient all-reduce with the backward (Lightning DDP's bucketed overlap in the
 * reference, rnampnn/utils/train.py:106-117).  The flat gradient is final in three contiguous chunks, in this order:
 * 0 = [post_fusion .. readout], 1 = ResMPNN layers L/2 .. L-1, 2 = the rest (end of the backward).  rnampnn_grad_chunks reports
 * their float ranges (begin[3], end[3]); rnampnn_set_grad_events registers two hipEvent_t (or null) that every later backward
 * records on its stream when chunk 0 / chunk 1 is final: the caller's side stream waits on them and all-reduces that range
 * while the rest of the backward runs. */
int     rnampnn_grad_chunks(rnampnn_handle h, int64_t* begin, int64_t* end);
int     rnampnn_set_grad_events(rnampnn_handle h, void* ev0, void* ev1);
/* hipGraph capture of a training step (the size-independent part of a small-batch step is ~600 launches): with a device seed source set,
 * the training kernels read the dropout seed from *seed_device at kernel time (the `seed` arguments are ignored), so ONE captured
 * rnampnn_loss_and_grad replays with fresh masks after the caller has updated the word.  No reference counterpart (the reference draws
 * from torch's global RNG, which CUDA graphs handle by the same device-side-offset idea).  null restores the argument. */
int     rnampnn_set_seed_source(rnampnn_handle h, const uint64_t* seed_device);
/* Optimiser support (F2).  rnampnn_use_weight_arena: the caller's flat f32 buffer (rnampnn_grad_numel() elements, tensor i
 * at rnampnn_weight_offset(i)) becomes the library's weight storage - the nn.Parameters of the Python module are views
 * of it, so an optimiser step needs no re-upload.  rnampnn_adam_step: torch.optim.Adam (betas, eps, L2 weight decay:
 * rnampnn.py:156-159) as ONE launch over the flat parameter / gradient / moment buffers; `step` counts from 1. */
int     rnampnn_use_weight_arena(rnampnn_handle h, float* arena, void* stream);
int     rnampnn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel, float lr,
                          float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);

/* -- measurement ------------------------------------------------------------------------- */
/* Live timing of the dominant kernel (the fused ResMPNN edge kernel, mpnn.py:154-265): when
 * enabled, HIP events bracket its launches on the caller's stream - every `enable`-th launch (1 = all; an event
 * pair costs ~12 us of stream idle, so a timed production loop samples, e.g. 7 against the 10 launches of a forward);
 * _read synchronises the recorded events and returns the summed duration and the number of TIMED launches since the
 * last reset.
 * No reference counterpart (the reference has no profiling hooks). */
int rnampnn_profile_enable(rnampnn_handle h, int32_t enable);
int rnampnn_profile_read(rnampnn_handle h, double* kernel_ms, int64_t* launches, int32_t reset);
/* The same split by launch kind: index 0 = launches that run the edge update of layer l AND the message of layer l + 1 (L - 1 per forward),
 * index 1 = the other fused launches (the message-only launch of layer 1, an edge-only launch of a tap). */
int rnampnn_profile_read_kinds(rnampnn_handle h, double* kernel_ms2, int64_t* launches2, int32_t reset);

const char* rnampnn_last_error(void);
const char* rnampnn_version(void);

/* Test tap of the 90 raw edge features of ResFeature (rnampnn/model/feature.py:386-517: `_cross_dists` 49, `_cross_angles` 25,
 * `_cross_dihedrals` 16) - the tensor the inference kernels keep in registers.  feats (B,T,k,96) f32: columns 90..95, padded residues and
 * absent neighbour slots are zero (the reference holds 1e6 distances there); edge_index (B,T,k) i64 optional. */
size_t rnampnn_edge_raw_workspace_bytes(rnampnn_handle h, int32_t B, int32_t T);
int rnampnn_edge_raw_features(rnampnn_handle h, const float* coords, const float* mask, int32_t B, int32_t T, int64_t* edge_index,
                              float* feats, void* ws, size_t ws_bytes, void* stream);

/* ---- Gradient-boosted-tree read-out (PARITY UNPINNED: xgboost is not installed, no fitted model ships).
 * Replaces `self.xgb_readout.predict(embedding)` (rnampnn/model/rnampnn.py:136-145,297-298) for a fitted `multi:softmax` gbtree model
 * given as the arrays of XGBoost's JSON model format (`learner.gradient_booster.model.trees[*].{left_children,right_children,
 * split_indices,split_conditions,default_left}` concatenated over the trees, `tree_info` = class of each tree).  Host arrays, copied.
 * Rule: at an internal node go left iff x[split_index] < split_condition (NaN: default_left); a leaf adds split_conditions[leaf] to the
 * margin of its tree's class; prediction = first argmax of the margins.  X (n_rows, ldx >= num_feature) f32 device, outputs device. */
typedef struct rnampnn_gbdt* rnampnn_gbdt_handle;
int rnampnn_gbdt_create(int32_t num_trees, int32_t num_class, int32_t num_feature, float base_score, const int32_t* tree_offsets,
                        const int32_t* tree_class, const int32_t* left_children, const int32_t* right_children,
                        const int32_t* split_indices, const float* split_conditions, const uint8_t* default_left,
                        rnampnn_gbdt_handle* out);
int rnampnn_gbdt_destroy(rnampnn_gbdt_handle g);
int rnampnn_gbdt_predict(rnampnn_gbdt_handle g, const float* X, int32_t n_rows, int32_t ldx, float* margin /* (n_rows,num_class) or null */,
                         int32_t* argmax_out /* (n_rows) or null */, void* stream);
const char* rnampnn_gbdt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RNAMPNN_HIP_H */
