"""Shared host-side plumbing of the drop-in modules: parameter containers with the reference's
state-dict keys, weight upload into the HIP library, workspace and stream handling.
PyTorch is used for device memory, streams and parameters only; all arithmetic of the path
runs in ``librnampnn_hip.so``."""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict
from typing import Dict, Optional

import torch
from torch import nn

from .. import _native
from ._schema import DEFAULT_HPARAMS, state_dict_shapes

DEFAULT_PRECISION = os.environ.get("RNAMPNN_PRECISION", "bf16")
_PREC = {"f32": _native.PREC_F32, "fp32": _native.PREC_F32, "float32": _native.PREC_F32,
         "bf16": _native.PREC_BF16, "bfloat16": _native.PREC_BF16}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _prep(t: torch.Tensor, device, dtype=torch.float32) -> torch.Tensor:
    """Caller tensors are const inputs: cast/move/contiguous-ify into a private copy if needed."""
    return t.detach().to(device=device, dtype=dtype).contiguous()


def _register_nested(root: nn.Module, key: str, param: nn.Parameter) -> None:
    parts = key.split(".")
    mod = root
    for name in parts[:-1]:
        if name not in mod._modules:
            mod.add_module(name, nn.Module())
        mod = mod._modules[name]
    mod.register_parameter(parts[-1], param)


def _init_like_reference(key: str, shape) -> torch.Tensor:
    """torch's default initialisers of the reference's layers (nn.Linear, nn.MultiheadAttention,
    GraphNormalization ones/zeros)."""
    leaf = key.rsplit(".", 1)[-1]
    t = torch.empty(shape, dtype=torch.float32)
    if leaf == "scale":
        return t.fill_(1.0)
    if leaf == "shift":
        return t.zero_()
    if leaf == "in_proj_weight":
        return nn.init.xavier_uniform_(t)
    if leaf == "in_proj_bias" or key.endswith("out_proj.bias"):
        return t.zero_()
    if leaf == "weight":
        return nn.init.kaiming_uniform_(t, a=5 ** 0.5)
    fan_in = None
    return t.uniform_(-1, 1)  # bias: scaled by the caller (needs the sibling's fan-in)


class NativeModule(nn.Module):
    """Base of every drop-in class.  ``hp`` are the full RNAMPNN hyper-parameters the C handle
    is created with; ``prefix`` selects which state-dict entries this module owns (the others,
    needed only to make the handle complete, are zero-filled at upload)."""

    def _setup(self, hp: Dict, prefix: str, precision: Optional[str]) -> None:
        full = OrderedDict(DEFAULT_HPARAMS)
        full.update(hp)
        self._hp = full
        self._prefix = prefix
        prec = (precision or DEFAULT_PRECISION).lower()
        if prec not in _PREC:
            raise ValueError(f"precision must be one of {sorted(_PREC)}, got {precision!r}")
        self.precision = "bf16" if _PREC[prec] == _native.PREC_BF16 else "f32"
        self._shapes = state_dict_shapes(full)
        for key, shape in self._shapes.items():
            if not key.startswith(prefix):
                continue
            t = _init_like_reference(key, shape)
            if key.endswith("bias") and not key.endswith("in_proj_bias") and not key.endswith("out_proj.bias"):
                fan_in = self._shapes[key[:-4] + "weight"][-1]
                t.mul_(1.0 / fan_in ** 0.5)
            _register_nested(self, key[len(prefix):], nn.Parameter(t))
        self._handle: Optional[_native.Handle] = None
        self._param_slices = None
        self._flat_param: Optional[torch.Tensor] = None
        self._finalized_sig = None
        self._ext_version = 0
        self._ws: Optional[torch.Tensor] = None
        self._warned_train = False

    # ------------------------------------------------------------------ native state
    def _device(self) -> torch.device:
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise RuntimeError("the RNA-MPNN HIP path runs on an MI355X: move the module to 'cuda' first "
                               "(there is no CPU fallback)")
        return p.device

    def _bind_flat_params(self, device) -> None:
        """All parameters are VIEWS of one flat f32 buffer laid out like the library's weight arena, and the library uses
        that buffer as its weight storage (``rnampnn_use_weight_arena``): an optimiser step - torch's per-tensor Adam or
        the fused ``FlatAdam`` - updates the weights the kernels read, nothing is re-uploaded.  Re-established whenever a
        parameter stopped aliasing its slice (``module.to(...)`` replaces ``p.data``)."""
        lib = _native.lib()
        if self._param_slices is None:
            named = dict(self.named_parameters())
            self._param_slices = []
            for i, (key, numel) in enumerate(self._handle.weight_schema()):
                off = C.c_int64()
                _native.check(lib.rnampnn_weight_offset(self._handle.ptr, i, C.byref(off)))
                if key.startswith(self._prefix):
                    self._param_slices.append((named[key[len(self._prefix):]], int(off.value), int(numel)))
        flat = self._flat_param
        if flat is not None and flat.device == device and all(p.data_ptr() == flat.data_ptr() + 4 * off
                                                               for p, off, _ in self._param_slices):
            return
        flat = torch.zeros(int(lib.rnampnn_grad_numel(self._handle.ptr)), dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, off, numel in self._param_slices:
                flat[off: off + numel].copy_(p.data.reshape(-1))
                p.data = flat[off: off + numel].view(p.shape)
        with torch.cuda.device(device):
            _native.check(lib.rnampnn_use_weight_arena(self._handle.ptr, C.c_void_p(flat.data_ptr()), _stream(device)))
        self._flat_param = flat
        self._finalized_sig = None

    def _weights_touched(self) -> None:
        """Called by code that changed the flat parameter buffer behind autograd's back (the fused optimiser)."""
        self._ext_version += 1

    def _ensure(self, for_mixed_training: bool = False) -> torch.device:
        """Handle + weight storage ready on the parameters' device.  The kernel-side layouts (K-major f32 copies, bf16
        fragment images) are rebuilt by ``rnampnn_finalize_weights`` only when the weights changed since the last build AND
        the caller needs them: the bf16-mixed training kernels read ``nn.Linear.weight`` as stored."""
        device = self._device()
        if self._handle is None:
            self._handle = _native.Handle(self._hp, _PREC[self.precision])
        self._bind_flat_params(device)
        if not for_mixed_training:
            sig = (self._ext_version,) + tuple(p._version for p, _, _ in self._param_slices)
            if sig != self._finalized_sig:
                with torch.cuda.device(device):
                    _native.check(_native.lib().rnampnn_finalize_weights(self._handle.ptr, _stream(device)))
                self._finalized_sig = sig
        return device

    def _workspace(self, B: int, T: int, device) -> torch.Tensor:
        """The module's shared scratch (grown on demand).  A hipGraph must never bake pointers into it - a later, larger
        call would free the buffer under the graph - so captures bring their own (``_ws_args(..., private=...)``)."""
        need = int(_native.lib().rnampnn_workspace_bytes(self._handle.ptr, B, T))
        if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=device)
        return self._ws

    def new_workspace(self, B: int, T: int) -> torch.Tensor:
        """A private workspace tensor for (B, T), owned by the caller (hipGraph captures)."""
        device = self._ensure()
        need = int(_native.lib().rnampnn_workspace_bytes(self._handle.ptr, B, T))
        return torch.empty(need + 256, dtype=torch.uint8, device=device)

    def _ws_args(self, B: int, T: int, device, private: Optional[torch.Tensor] = None):
        ws = private if private is not None else self._workspace(B, T, device)
        base = ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        return C.c_void_p(aligned), C.c_size_t(ws.numel() - (aligned - base))

    # ------------------------------------------------------------------ measurement hooks
    def profile_enable(self, enable: bool = True, every: int = 1) -> None:
        """HIP-event timing of every ``every``-th launch of the dominant (fused ResMPNN edge) kernel."""
        self._ensure()
        _native.check(_native.lib().rnampnn_profile_enable(self._handle.ptr, max(1, int(every)) if enable else 0))

    def profile_read(self, reset: bool = True):
        """-> (summed kernel milliseconds, launches) since the last reset (synchronises the events)."""
        ms, n = C.c_double(), C.c_int64()
        _native.check(_native.lib().rnampnn_profile_read(self._handle.ptr, C.byref(ms), C.byref(n), 1 if reset else 0))
        return float(ms.value), int(n.value)

    def profile_read_kinds(self, reset: bool = True):
        """-> ((ms, launches) of the <edge update, message> launches, (ms, launches) of the other fused launches) since the last reset."""
        ms, n = (C.c_double * 2)(), (C.c_int64 * 2)()
        _native.check(_native.lib().rnampnn_profile_read_kinds(self._handle.ptr, ms, n, 1 if reset else 0))
        return (float(ms[0]), int(n[0])), (float(ms[1]), int(n[1]))
