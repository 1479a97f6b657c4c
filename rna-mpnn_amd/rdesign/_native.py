"""ctypes binding of the ``rdesign_*`` entry points of ``librnampnn_hip.so`` (C ABI: ``include/rdesign_hip.h``).
Same library, same loader and the same rule as ``rnampnn._native``: no fallback - a missing library raises."""
from __future__ import annotations

import ctypes as C

from rnampnn import _native as _rn

PREC_F32, PREC_BF16 = 0, 1
_EXC = {1: ValueError, 2: NotImplementedError, 5: RuntimeError, 6: KeyError, 7: RuntimeError}


class RDesignConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "hidden_dim", "k_neighbors", "num_mpnn_layers", "num_message_layers", "num_dense_layers", "dim_dense_layers",
        "num_readout_layers", "readout_hidden_dim", "precision")]


_VP, _I32, _I64, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
SYMBOLS = {
    "rdesign_create": (C.c_int, [C.POINTER(RDesignConfig), C.POINTER(_VP)]),
    "rdesign_destroy": (C.c_int, [_VP]),
    "rdesign_last_error": (C.c_char_p, []),
    "rdesign_num_weights": (C.c_int, [_VP]),
    "rdesign_weight_info": (C.c_int, [_VP, _I32, C.POINTER(C.c_char_p), C.POINTER(_I64), C.POINTER(_I64)]),
    "rdesign_param_numel": (_I64, [_VP]),
    "rdesign_use_weight_arena": (C.c_int, [_VP, _VP, _VP]),
    "rdesign_finalize_weights": (C.c_int, [_VP, _VP]),
    "rdesign_workspace_bytes": (_SZ, [_VP, _I32, _I32]),
    "rdesign_readout_workspace_bytes": (_SZ, [_VP, _I32]),
    "rdesign_forward": (C.c_int, [_VP, _VP, _VP, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "rdesign_readout": (C.c_int, [_VP, _VP, _I32, _VP, _VP, _SZ, _VP]),
}
_bound = False


def lib() -> C.CDLL:
    global _bound
    handle = _rn.lib()
    if not _bound:
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return handle


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().rdesign_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, RuntimeError)(f"rdesign_hip error {rc}: {msg}")


class Handle:
    """Owns one ``rdesign_handle``."""

    def __init__(self, hp, precision: int):
        self._h = _VP()
        self.cfg = RDesignConfig()
        for name, _ in RDesignConfig._fields_:
            if name != "precision":
                setattr(self.cfg, name, int(hp[name]))
        self.cfg.precision = int(precision)
        check(lib().rdesign_create(C.byref(self.cfg), C.byref(self._h)))

    @property
    def ptr(self):
        return self._h

    def weight_schema(self):
        """-> [(key, numel, offset in floats)] in state_dict order."""
        out = []
        for i in range(lib().rdesign_num_weights(self._h)):
            key, numel, off = C.c_char_p(), _I64(), _I64()
            check(lib().rdesign_weight_info(self._h, i, C.byref(key), C.byref(numel), C.byref(off)))
            out.append((key.value.decode(), int(numel.value), int(off.value)))
        return out

    def close(self):
        if self._h:
            lib().rdesign_destroy(self._h)
            self._h = _VP()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
