"""Drop-in for ``ResFeature`` of the reference's ``rnampnn/model/feature.py:133-592``."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from .. import _native
from ..config.glob import DEFAULT_HIDDEN_DIM, NUM_MAIN_SEQ_ATOMS
from ._base import NativeModule, _prep, _ptr, _stream


class ResFeature(NativeModule):
    """Same constructor arguments and defaults as the reference (feature.py:134-150);
    ``forward(coords, mask) -> (raw, res_embedding, res_edge_embedding, edge_index)`` (feature.py:573-592)."""

    def __init__(self, num_neighbours: int,
                 num_inside_dist_atoms: int = NUM_MAIN_SEQ_ATOMS,
                 num_inside_angle_atoms: int = NUM_MAIN_SEQ_ATOMS - 1,
                 num_inside_dihedral_atoms: int = NUM_MAIN_SEQ_ATOMS - 1,
                 num_cross_dist_atoms: int = NUM_MAIN_SEQ_ATOMS,
                 num_cross_angle_atoms: int = NUM_MAIN_SEQ_ATOMS - 1,
                 num_cross_dihedral_atoms: int = NUM_MAIN_SEQ_ATOMS - 1,
                 res_embedding_dim: int = DEFAULT_HIDDEN_DIM, padding_len: int = 4500, num_attn_layers: int = 2,
                 num_heads=4, ffn_dim=DEFAULT_HIDDEN_DIM, num_ffn_layers=2,
                 res_edge_embedding_dim: int = DEFAULT_HIDDEN_DIM, num_edge_layers: int = 2, dropout: float = 0.1,
                 precision: Optional[str] = None):
        super().__init__()
        # AssertionError on impossible atom counts, as feature.py:169-179 raises
        minimum = {"dist": 2, "angle": 3, "dihedral": 4}
        for scope, counts in (("inside", (num_inside_dist_atoms, num_inside_angle_atoms, num_inside_dihedral_atoms)),
                              ("cross", (num_cross_dist_atoms, num_cross_angle_atoms, num_cross_dihedral_atoms))):
            for (kind, least), got in zip(minimum.items(), counts):
                assert got >= least, f"num_{scope}_{kind}_atoms = {got}: a {kind} feature needs at least {least} atoms"
        if (num_inside_dist_atoms, num_inside_angle_atoms, num_inside_dihedral_atoms, num_cross_dist_atoms,
                num_cross_angle_atoms, num_cross_dihedral_atoms) != (7, 6, 6, 7, 6, 6):
            raise NotImplementedError("the HIP featurisation kernels are built for atom counts 7/6/6")
        self.num_neighbours = num_neighbours
        self.raw_dim = 28
        self._setup(dict(num_res_neighbours=num_neighbours, res_embedding_dim=res_embedding_dim,
                         padding_len=padding_len, num_embedding_attn_layers=num_attn_layers,
                         num_embedding_heads=num_heads, embedding_ffn_dim=ffn_dim,
                         num_embedding_ffn_layers=num_ffn_layers, res_edge_embedding_dim=res_edge_embedding_dim,
                         depth_res_edge_feature=num_edge_layers, num_res_mpnn_layers=1,
                         num_post_fusion_attn_layers=0, num_post_fusion_ffn_layers=1, post_fusion_ffn_dim=128,
                         num_raw_ffn_layers=1, num_raw_ffn_dim=128, num_readout_layers=1),
                    "res_feature.", precision)

    def forward(self, coords: torch.Tensor, mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        device = self._ensure()
        B, T = int(coords.shape[0]), int(coords.shape[1])
        k = self.num_neighbours
        c, m = _prep(coords, device), _prep(mask, device)
        raw = torch.empty(B, T, 28, dtype=torch.float32, device=device)
        h = torch.empty(B, T, 128, dtype=torch.float32, device=device)
        e = torch.empty(B, T, k, 128, dtype=torch.float32, device=device)
        idx = torch.empty(B, T, k, dtype=torch.int64, device=device)
        io = _native.RnaMpnnForwardIO()
        io.coords, io.mask, io.B, io.T, io.stop_after = _ptr(c), _ptr(m), B, T, 1
        io.raw, io.h0, io.e0, io.edge_index = _ptr(raw), _ptr(h), _ptr(e), _ptr(idx)
        with torch.cuda.device(device):
            ws, ws_bytes = self._ws_args(B, T, device)
            _native.check(_native.lib().rnampnn_forward(self._handle.ptr, C.byref(io), ws, ws_bytes, _stream(device)))
        return raw, h, e, idx
