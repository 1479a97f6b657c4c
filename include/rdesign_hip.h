/*
 * rdesign_hip.h - C ABI of the MI355X path of the reference's sibling model `rdesign`,
 * exported by the same librnampnn_hip.so as rnampnn_hip.h.
 *
 * The reference boundary is the Python module surface of `rdesign.model` (the model `main.py:24-31` and
 * `train.py:67-80` of the reference run today).  Every entry point names the reference interface it replaces;
 * `rna-mpnn_amd/rdesign/model/rdesign.py` binds these symbols with ctypes and re-exposes the reference's names.
 *
 * PARITY UNPINNED: the reference's `rdesign.model` modules import `rdesign/utils/data.py`, which needs BioPython,
 * pytorch_lightning and seaborn - absent from this image - and no rdesign fixture or checkpoint ships with the reference.
 * The checker is the CPU restatement `oracle/rdesign_oracle.py` (numpy, citing the reference line by line) alone.
 *
 * Conventions: as rnampnn_hip.h (device pointers, caller's stream, no synchronisation, 0 = success).
 *   X (B,T,6,3) f32 backbone atoms P, O5', C5', C4', C3', O3' (rdesign/utils/data.py:90-115, zero-filled padding),
 *   mask (B,T) f32 0/1 prefix masks.  Outputs are PACKED: the reference drops padded residues
 *   (feature.py:206 `mask_select`), row p = (number of valid residues of RNAs < b) + t, N = mask.sum().
 */
#ifndef RDESIGN_HIP_H
#define RDESIGN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDESIGN_OK               0
#define RDESIGN_ERR_BAD_ARG      1
#define RDESIGN_ERR_UNSUPPORTED  2
#define RDESIGN_ERR_WORKSPACE    5
#define RDESIGN_ERR_WEIGHTS      6
#define RDESIGN_ERR_HIP          7

#define RDESIGN_PREC_F32  0   /* exact-f32 GEMMs                                                              */
#define RDESIGN_PREC_BF16 1   /* bf16 MFMA operands, f32 accumulate (the embeddings' 101/115-wide GEMMs stay f32) */

typedef struct rdesign_ctx* rdesign_handle;

/* Hyper-parameters of RNAModel.__init__ (rdesign/model/rdesign.py:19-50). */
typedef struct RDesignConfig {
    int32_t hidden_dim;           /* must be 128 */
    int32_t k_neighbors;          /* 1..64 (reference default 25) */
    int32_t num_mpnn_layers;      /* reference default 9 */
    int32_t num_message_layers;   /* Linears of MPNNLayer.message_layers (mpnn.py:13-19), default 3 */
    int32_t num_dense_layers;     /* hidden Linears of MPNNLayer.dense (mpnn.py:21-29), default 3 */
    int32_t dim_dense_layers;     /* default 256 */
    int32_t num_readout_layers;   /* Readout (functional.py:98-121): num_layers - 1 hidden Linears + the 4-way Linear; default 0 */
    int32_t readout_hidden_dim;   /* default 256 */
    int32_t precision;            /* RDESIGN_PREC_* */
} RDesignConfig;

/* RNAModel.__init__ (rdesign.py:19-64): validates the configuration and builds the parameter table. */
int rdesign_create(const RDesignConfig* cfg, rdesign_handle* out);
int rdesign_destroy(rdesign_handle h);
const char* rdesign_last_error(void);

/* state_dict() surface (the keys of RNAModel.state_dict(), registration order): key, element count and the offset
 * (in floats) of the tensor inside the flat parameter arena. */
int rdesign_num_weights(rdesign_handle h);
int rdesign_weight_info(rdesign_handle h, int32_t index, const char** key, int64_t* numel, int64_t* offset);
int64_t rdesign_param_numel(rdesign_handle h);
/* load_state_dict(): the caller owns ONE flat f32 device buffer of rdesign_param_numel() floats (16-byte aligned)
 * whose slices are the parameters (torch Parameters are views of it); the library reads it in place. */
int rdesign_use_weight_arena(rdesign_handle h, float* arena, void* stream);
/* derive the kernel-side weight images after the arena changed (optimizer step, load_state_dict). */
int rdesign_finalize_weights(rdesign_handle h, void* stream);

size_t rdesign_workspace_bytes(rdesign_handle h, int32_t B, int32_t T);

/* RNAModel.forward (rdesign.py:82-88) followed by Readout (rdesign.py:104): every output is optional (null = skipped).
 *   h_V      (B*T,128) packed node embeddings after the MPNN stack (rows >= N untouched)
 *   logits   (B*T,4)   packed read-out logits
 *   edge_index (B,T,k) i64: neighbour position inside the RNA per slot, -1 for padded residues and for the slots
 *            beyond the RNA's length (the edges the reference's `mask_attend` filter removes, feature.py:186-194)
 *   node_raw (B*T,101) / edge_raw (B*T*k,115): raw geometric features (feature.py:221-236), test taps */
int rdesign_forward(rdesign_handle h, const float* X, const float* mask, int32_t B, int32_t T, float* h_V, float* logits,
                    int64_t* edge_index, float* node_raw, float* edge_raw, void* ws, size_t ws_bytes, void* stream);

/* Readout.forward (functional.py:123-126) on n_rows caller rows of 128 floats -> logits (n_rows,4).
 * Workspace: rdesign_readout_workspace_bytes(h, n_rows) (node-sized buffers only). */
size_t rdesign_readout_workspace_bytes(rdesign_handle h, int32_t n_rows);
int rdesign_readout(rdesign_handle h, const float* h_V, int32_t n_rows, float* logits, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
