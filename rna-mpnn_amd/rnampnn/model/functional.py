"""Drop-ins for ``GraphNormalization``, ``Readout``, ``RNABert`` and ``RawFFN`` of the reference's
``rnampnn/model/functional.py``."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import _native
from ..config.glob import NUM_RES_TYPES
from ._base import NativeModule, _prep, _ptr, _stream


class GraphNormalization(nn.Module):
    """functional.py:7-48.  ``forward(features, mask, t_tot=None)``: ``t_tot`` (extension) is the
    node-axis length entering the padding-dependent variance; default = features.shape[1]."""

    def __init__(self, embedding_dim: int):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(1, 1, embedding_dim))
        self.shift = nn.Parameter(torch.zeros(1, 1, embedding_dim))

    def forward(self, features: torch.Tensor, mask: torch.Tensor, t_tot: Optional[int] = None) -> torch.Tensor:
        device = self.scale.device
        if device.type != "cuda":
            raise RuntimeError("GraphNormalization HIP kernel needs the module on 'cuda' (no CPU fallback)")
        B, T, D = (int(s) for s in features.shape)
        x, m = _prep(features, device), _prep(mask, device)
        y = torch.empty_like(x)
        with torch.cuda.device(device):
            _native.check(_native.lib().rnampnn_graph_norm(_ptr(x), _ptr(m), _ptr(_prep(self.scale.data, device)),
                                                           _ptr(_prep(self.shift.data, device)), B, T,
                                                           int(t_tot or T), D, _ptr(y), _stream(device)))
        return y


_MINIMAL = dict(num_res_neighbours=1, num_res_mpnn_layers=1, num_embedding_ffn_layers=1, embedding_ffn_dim=128,
                num_post_fusion_attn_layers=0, num_post_fusion_ffn_layers=1, post_fusion_ffn_dim=128,
                num_raw_ffn_layers=1, num_raw_ffn_dim=128, num_readout_layers=1, padding_len=1 << 30)


class Readout(NativeModule):
    """functional.py:51-90: ``Readout(embedding_dim, readout_hidden_dim, num_layers, dropout)``."""

    def __init__(self, embedding_dim: int, readout_hidden_dim: int, num_layers: int, dropout: float = 0.1,
                 precision: Optional[str] = None):
        super().__init__()
        if embedding_dim != 256:
            raise NotImplementedError("Readout HIP path expects cat(res_embedding, raw_embedding) = 256 inputs")
        self._setup(dict(_MINIMAL, readout_hidden_dim=readout_hidden_dim, num_readout_layers=num_layers),
                    "readout.", precision)

    def forward(self, res_embedding: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        device = self._ensure()
        B, T = int(res_embedding.shape[0]), int(res_embedding.shape[1])
        x, m = _prep(res_embedding, device), _prep(mask, device)
        y = torch.empty(B, T, NUM_RES_TYPES, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            ws, ws_bytes = self._ws_args(B, T, device)
            _native.check(_native.lib().rnampnn_readout(self._handle.ptr, _ptr(x), _ptr(m), B, T, _ptr(y), ws, ws_bytes,
                                                        _stream(device)))
        return y


class RNABert(NativeModule):
    """functional.py:93-172 (the never-called ``_position_embedding`` is not mirrored)."""

    def __init__(self, padding_len: int, res_embedding_dim: int, num_attn_layers: int, num_heads: int, ffn_dim: int,
                 num_ffn_layers: int, dropout: float = 0.1, precision: Optional[str] = None):
        super().__init__()
        self.padding_len = padding_len
        self._setup(dict(_MINIMAL, padding_len=padding_len, res_embedding_dim=res_embedding_dim,
                         num_post_fusion_attn_layers=num_attn_layers, num_post_fusion_heads=num_heads,
                         post_fusion_ffn_dim=ffn_dim, num_post_fusion_ffn_layers=num_ffn_layers),
                    "post_fusion.", precision)

    def forward(self, res_embedding: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        device = self._ensure()
        B, T = int(res_embedding.shape[0]), int(res_embedding.shape[1])
        x, m = _prep(res_embedding, device), _prep(mask, device)
        y = torch.empty(B, T, 128, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            ws, ws_bytes = self._ws_args(B, T, device)
            _native.check(_native.lib().rnampnn_rnabert(self._handle.ptr, 1, _ptr(x), _ptr(m), B, T, _ptr(y), ws,
                                                        ws_bytes, _stream(device)))
        return y


class RawFFN(NativeModule):
    """functional.py:175-202: ``RawFFN(raw_dim, num_raw_ffn_dim, num_raw_ffn_layers, raw_embedding_dim, dropout)``."""

    def __init__(self, raw_dim: int, num_raw_ffn_dim: int, num_raw_ffn_layers: int, raw_embedding_dim: int,
                 dropout: float = 0.1, precision: Optional[str] = None):
        super().__init__()
        if raw_dim != 28:
            raise NotImplementedError("RawFFN HIP path expects the 28 raw node features")
        self._setup(dict(_MINIMAL, num_raw_ffn_dim=num_raw_ffn_dim, num_raw_ffn_layers=num_raw_ffn_layers,
                         raw_embedding_dim=raw_embedding_dim), "raw_embedding.", precision)

    def forward(self, raw: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        device = self._ensure()
        B, T = int(raw.shape[0]), int(raw.shape[1])
        x, m = _prep(raw, device), _prep(mask, device)
        y = torch.empty(B, T, 128, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            ws, ws_bytes = self._ws_args(B, T, device)
            _native.check(_native.lib().rnampnn_raw_ffn(self._handle.ptr, _ptr(x), _ptr(m), B, T, 0, _ptr(y), ws,
                                                        ws_bytes, _stream(device)))
        return y
