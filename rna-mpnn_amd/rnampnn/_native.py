"""ctypes binding of ``librnampnn_hip.so`` (C ABI: ``include/rnampnn_hip.h``).

The product path has no fallback: if the shared library is missing (or a call fails) this
module raises - it never routes to PyTorch ops or to the CPU oracle.  Build the library with
``python -c "import __graft_entry__ as g; g.build()"`` (hipcc --offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  - FIRST: the process must use ONE HIP runtime (the one PyTorch-ROCm loads); loading
#                librnampnn_hip.so before torch would bring in a second libamdhip64 that sees no device

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RNAMPNN_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "librnampnn_hip.so")   # RNAMPNN_LIB: A/B builds of the same ABI (tools/build_variant.sh)

PREC_F32, PREC_BF16 = 0, 1
TRAIN_F32, TRAIN_BF16_MIXED = 0, 1
KMAX = 32

ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_T_GT_P, ERR_K_TOO_LARGE, ERR_WORKSPACE, ERR_WEIGHTS, ERR_HIP = 1, 2, 3, 4, 5, 6, 7
_EXC = {ERR_BAD_ARG: ValueError, ERR_UNSUPPORTED: NotImplementedError, ERR_T_GT_P: RuntimeError,
        ERR_K_TOO_LARGE: NotImplementedError, ERR_WORKSPACE: RuntimeError, ERR_WEIGHTS: KeyError,
        ERR_HIP: RuntimeError}


class RnaMpnnConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_res_neighbours", "res_embedding_dim", "num_embedding_attn_layers", "num_embedding_heads",
        "embedding_ffn_dim", "num_embedding_ffn_layers", "res_edge_embedding_dim", "depth_res_edge_feature",
        "num_res_mpnn_layers", "depth_res_mpnn", "num_mpnn_edge_layers", "padding_len",
        "num_post_fusion_attn_layers", "num_post_fusion_heads", "post_fusion_ffn_dim",
        "num_post_fusion_ffn_layers", "num_raw_ffn_dim", "num_raw_ffn_layers", "raw_embedding_dim",
        "readout_hidden_dim", "num_readout_layers", "precision")]


class RnaMpnnForwardIO(C.Structure):
    _fields_ = [("coords", C.c_void_p), ("mask", C.c_void_p), ("B", C.c_int32), ("T", C.c_int32),
                ("T_norm", C.c_int32), ("stop_after", C.c_int32),
                ("logits", C.c_void_p), ("embedding", C.c_void_p), ("edge_index", C.c_void_p),
                ("raw", C.c_void_p), ("h0", C.c_void_p), ("e0", C.c_void_p),
                ("tap_layer", C.c_int32), ("h_layer", C.c_void_p), ("e_layer", C.c_void_p),
                ("h_post", C.c_void_p), ("raw_emb", C.c_void_p)]


# every symbol include/rnampnn_hip.h declares: (restype, argtypes)
_VP, _I32, _I64, _SZ, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float
SYMBOLS = {
    "rnampnn_create": (C.c_int, [C.POINTER(RnaMpnnConfig), C.POINTER(_VP)]),
    "rnampnn_destroy": (C.c_int, [_VP]),
    "rnampnn_set_weight": (C.c_int, [_VP, C.c_char_p, _VP, _I64, _I32, _VP]),
    "rnampnn_num_weights": (C.c_int, [_VP]),
    "rnampnn_weight_info": (C.c_int, [_VP, _I32, C.POINTER(C.c_char_p), C.POINTER(_I64)]),
    "rnampnn_finalize_weights": (C.c_int, [_VP, _VP]),
    "rnampnn_workspace_bytes": (_SZ, [_VP, _I32, _I32]),
    "rnampnn_forward": (C.c_int, [_VP, C.POINTER(RnaMpnnForwardIO), _VP, _SZ, _VP]),
    "rnampnn_workspace_bytes_packed": (_SZ, [_VP, _I32, _I32]),
    "rnampnn_forward_packed": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _SZ, _VP]),
    "rnampnn_mpnn_layer": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "rnampnn_graph_norm": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP]),
    "rnampnn_rnabert": (C.c_int, [_VP, _I32, _VP, _VP, _I32, _I32, _VP, _VP, _SZ, _VP]),
    "rnampnn_raw_ffn": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _SZ, _VP]),
    "rnampnn_readout": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP, _VP, _SZ, _VP]),
    "rnampnn_argmax_recovery": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP]),
    "rnampnn_sample": (C.c_int, [_VP, _VP, _I32, _I32, _F, _I32, C.c_uint64, _VP, _VP]),
    "rnampnn_sample_dev_seed": (C.c_int, [_VP, _VP, _I32, _I32, _F, _I32, _VP, _VP, _VP]),
    "rnampnn_train_workspace_bytes": (_SZ, [_VP, _I32, _I32]),
    "rnampnn_grad_numel": (_I64, [_VP]),
    "rnampnn_weight_offset": (C.c_int, [_VP, _I32, C.POINTER(_I64)]),
    "rnampnn_train_forward": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _I32, _F, C.c_uint64, _I32, _VP, _VP, _SZ, _VP, C.POINTER(_I64)]),
    "rnampnn_grad_chunks": (C.c_int, [_VP, C.POINTER(_I64), C.POINTER(_I64)]),
    "rnampnn_set_grad_events": (C.c_int, [_VP, _VP, _VP]),
    "rnampnn_set_seed_source": (C.c_int, [_VP, _VP]),
    "rnampnn_use_weight_arena": (C.c_int, [_VP, _VP, _VP]),
    "rnampnn_adam_step": (C.c_int, [_VP, _VP, _VP, _VP, _I64, _F, _F, _F, _F, _F, _I32, _VP]),
    "rnampnn_train_backward": (C.c_int, [_VP, _I64, _VP, _I32, _I32, _I32, _VP, _VP, _SZ, _VP]),
    "rnampnn_loss_and_grad": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _F, C.c_uint64, _I32, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "rnampnn_profile_enable": (C.c_int, [_VP, _I32]),
    "rnampnn_profile_read": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(_I64), _I32]),
    "rnampnn_profile_read_kinds": (C.c_int, [_VP, C.POINTER(C.c_double), C.POINTER(_I64), _I32]),
    "rnampnn_edge_raw_workspace_bytes": (_SZ, [_VP, _I32, _I32]),
    "rnampnn_edge_raw_features": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP, _VP, _VP, _SZ, _VP]),
    "rnampnn_gbdt_create": (C.c_int, [_I32, _I32, _I32, _F, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(_VP)]),
    "rnampnn_gbdt_destroy": (C.c_int, [_VP]),
    "rnampnn_gbdt_predict": (C.c_int, [_VP, _VP, _I32, _I32, _VP, _VP, _VP]),
    "rnampnn_gbdt_last_error": (C.c_char_p, []),
    "rnampnn_last_error": (C.c_char_p, []),
    "rnampnn_version": (C.c_char_p, []),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP library once; raise loudly when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the RNA-MPNN HIP extension is not built. "
                "Run `python -c \"import __graft_entry__ as g; g.build()\"` (hipcc, gfx950). "
                "There is no PyTorch/CPU fallback for this path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().rnampnn_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, RuntimeError)(f"rnampnn_hip error {rc}: {msg}")


def make_config(hp, precision: int) -> RnaMpnnConfig:
    cfg = RnaMpnnConfig()
    for name, _ in RnaMpnnConfig._fields_:
        if name != "precision":
            setattr(cfg, name, int(hp[name]))
    cfg.precision = int(precision)
    return cfg


class Handle:
    """Owns one ``rnampnn_handle``."""

    def __init__(self, hp, precision: int):
        self._h = _VP()
        self.cfg = make_config(hp, precision)
        check(lib().rnampnn_create(C.byref(self.cfg), C.byref(self._h)))

    @property
    def ptr(self):
        return self._h

    def weight_schema(self):
        n = lib().rnampnn_num_weights(self._h)
        out = []
        for i in range(n):
            key, numel = C.c_char_p(), _I64()
            check(lib().rnampnn_weight_info(self._h, i, C.byref(key), C.byref(numel)))
            out.append((key.value.decode(), int(numel.value)))
        return out

    def close(self):
        if self._h:
            lib().rnampnn_destroy(self._h)
            self._h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
