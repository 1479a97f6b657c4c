"""Drop-in for ``ResMPNN`` of the reference's ``rnampnn/model/mpnn.py:115-294``."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import _native
from ._base import NativeModule, _prep, _ptr, _stream


class ResMPNN(NativeModule):
    """``ResMPNN(res_embedding_dim, res_edge_embedding_dim, depth_res_mpnn, num_edge_layers, dropout)``
    (mpnn.py:116-121).  The number of neighbours is read from the edge tensor at call time."""

    def __init__(self, res_embedding_dim: int, res_edge_embedding_dim: int, depth_res_mpnn: int,
                 num_edge_layers: int, dropout: float = 0.0, precision: Optional[str] = None):
        super().__init__()
        self._args = dict(res_embedding_dim=res_embedding_dim, res_edge_embedding_dim=res_edge_embedding_dim,
                          depth_res_mpnn=depth_res_mpnn, num_mpnn_edge_layers=num_edge_layers)
        self._k = None
        self._precision_arg = precision
        self._build(3)

    def _build(self, k: int):
        old = {n: p for n, p in self.named_parameters()} if self._k is not None else None
        self._setup(dict(self._args, num_res_neighbours=k, num_res_mpnn_layers=1, padding_len=1 << 30,
                         num_post_fusion_attn_layers=0, num_post_fusion_ffn_layers=1, post_fusion_ffn_dim=128,
                         num_embedding_ffn_layers=1, embedding_ffn_dim=128, num_raw_ffn_layers=1,
                         num_raw_ffn_dim=128, num_readout_layers=1),
                    "res_mpnn_layers.0.", self._precision_arg)
        if old is not None:                      # keep the parameter objects when k changes
            for n, p in old.items():
                mod = self
                parts = n.split(".")
                for part in parts[:-1]:
                    mod = mod._modules[part]
                mod._parameters[parts[-1]] = p
        self._k = k

    def _call(self, h, e, idx, mask, want_msg, want_h, want_e):
        k = int(e.shape[2])
        if k != self._k:
            self._build(k)
        device = self._ensure()
        B, T = int(h.shape[0]), int(h.shape[1])
        hh, ee, mm = _prep(h, device), _prep(e, device), _prep(mask, device)
        ii = _prep(idx, device, torch.int64)
        msg = torch.empty(B, T, k, 128, dtype=torch.float32, device=device) if want_msg else None
        ho = torch.empty(B, T, 128, dtype=torch.float32, device=device) if want_h else None
        eo = torch.empty(B, T, k, 128, dtype=torch.float32, device=device) if want_e else None
        with torch.cuda.device(device):
            ws, ws_bytes = self._ws_args(B, T, device)
            _native.check(_native.lib().rnampnn_mpnn_layer(self._handle.ptr, 0, _ptr(hh), _ptr(ee), _ptr(ii), _ptr(mm),
                                                           B, T, 0, _ptr(msg), _ptr(ho), _ptr(eo), ws, ws_bytes,
                                                           _stream(device)))
        return msg, ho, eo

    def message(self, res_embedding, res_edge_embedding, edge_index, mask) -> torch.Tensor:
        """mpnn.py:154-194 -> messages (B, T, k, 128), zero on invalid edges."""
        return self._call(res_embedding, res_edge_embedding, edge_index, mask, True, False, False)[0]

    def forward(self, res_embedding, res_edge_embedding, edge_index, mask) -> Tuple[torch.Tensor, torch.Tensor]:
        """mpnn.py:267-294 -> (h, e).  The reference updates ``res_edge_embedding`` in place and leaves
        never-consumed values on invalid edge slots; here the caller's tensor is const and invalid
        slots of the returned e are 0."""
        _, h, e = self._call(res_embedding, res_edge_embedding, edge_index, mask, False, True, True)
        return h, e
