"""CPU ORACLE for the sibling `rdesign` model of the reference.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference's `rdesign/model/{feature,mpnn,functional}.py` cannot be imported in the build container:
all three import `rdesign/utils/data.py`, which pulls BioPython, pytorch_lightning and seaborn at import time (none is
installed, none is in the wheelhouse), and the reference holds no numeric fixture for this model (no test, no checkpoint,
`out/logs/RDesign-X/*/hparams.yaml` only).  This file is therefore a restatement from reading the source, line by line, and is
checked only for self-consistency (`tests/test_rdesign_cpu.py`: shapes, invariances, gradient flow).  Every function cites the
reference lines it restates (paths relative to the reference root).

Only `tests/` may import this module; the product path (`rna-mpnn_amd/rdesign`, `librnampnn_hip.so`) never does.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Mapping, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class RDesignConfig:
    """`RNAModel.__init__` (rdesign/model/rdesign.py:20-38)."""
    hidden_dim: int = 128
    k_neighbors: int = 25
    num_message_layers: int = 3
    num_dense_layers: int = 3
    dim_dense_layers: int = 256
    num_mpnn_layers: int = 9
    readout_hidden_dim: int = 256
    num_readout_layers: int = 0
    num_rbf: int = 16
    scale: float = 30.0            # MPNNLayer(scale=30): messages are summed and divided by 30 (mpnn.py:6,33)


def state_dict_shapes(cfg: RDesignConfig) -> "Dict[str, Tuple[int, ...]]":
    """Keys / shapes of `RNAModel.state_dict()` in torch's registration order (rdesign.py:52-64, feature.py:22-27,
    mpnn.py:11-29, functional.py:103-121)."""
    H, D = cfg.hidden_dim, cfg.dim_dense_layers
    out: Dict[str, Tuple[int, ...]] = {}
    out["features.node_embedding.weight"] = (H, 101); out["features.node_embedding.bias"] = (H,)
    out["features.edge_embedding.weight"] = (H, 115); out["features.edge_embedding.bias"] = (H,)
    out["features.norm_nodes.gain"] = (H,); out["features.norm_nodes.bias"] = (H,)
    out["features.norm_edges.gain"] = (H,); out["features.norm_edges.bias"] = (H,)
    for l in range(cfg.num_mpnn_layers):
        p = f"mpnn_layers.{l}"
        out[p + ".norm1.weight"] = (H,); out[p + ".norm1.bias"] = (H,)
        out[p + ".norm2.weight"] = (H,); out[p + ".norm2.bias"] = (H,)
        n_in = 3 * H
        for i in range(cfg.num_message_layers):
            out[f"{p}.message_layers.{3 * i}.weight"] = (H, n_in); out[f"{p}.message_layers.{3 * i}.bias"] = (H,)
            n_in = H
        for i in range(cfg.num_dense_layers):
            out[f"{p}.dense.{3 * i}.weight"] = (D, n_in); out[f"{p}.dense.{3 * i}.bias"] = (D,)
            n_in = D
        out[f"{p}.dense.{3 * cfg.num_dense_layers}.weight"] = (H, n_in); out[f"{p}.dense.{3 * cfg.num_dense_layers}.bias"] = (H,)
    n_in = H
    for i in range(max(cfg.num_readout_layers - 1, 0)):
        out[f"readout.readout_layers.{3 * i}.weight"] = (cfg.readout_hidden_dim, n_in)
        out[f"readout.readout_layers.{3 * i}.bias"] = (cfg.readout_hidden_dim,)
        n_in = cfg.readout_hidden_dim
    last = 3 * max(cfg.num_readout_layers - 1, 0)
    out[f"readout.readout_layers.{last}.weight"] = (4, n_in); out[f"readout.readout_layers.{last}.bias"] = (4,)
    return out


# ----------------------------------------------------------------------------- helpers
def _normalize_nan0(t: Tensor) -> Tensor:
    """`utils.data.normalize` (rdesign/utils/data.py:169-171): t / ||t||, NaN (0/0) -> 0."""
    return torch.nan_to_num(t / torch.norm(t, dim=-1, keepdim=True))


def _gelu(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * 0.7071067811865476))


def _custom_norm(x: Tensor, gain: Tensor, bias: Tensor, eps: float = 1e-6) -> Tensor:
    """`functional.Normalize` (functional.py:83-101): unbiased variance, eps inside the sqrt AND added to sigma."""
    mu = x.mean(-1, keepdim=True)
    sigma = torch.sqrt(x.var(-1, keepdim=True) + eps)
    return gain * (x - mu) / (sigma + eps) + bias


def _rbf(D: Tensor, num_rbf: int) -> Tensor:
    """`RNAFeatures._rbf` (feature.py:50-56): 16 Gaussians, centres linspace(0, 20), sigma = 20 / 16."""
    mu = torch.linspace(0.0, 20.0, num_rbf, dtype=D.dtype)
    sigma = 20.0 / num_rbf
    return torch.exp(-(((D.unsqueeze(-1) - mu) / sigma) ** 2))


def _gather_nodes(nodes: Tensor, idx: Tensor) -> Tensor:
    """feature.py:30-35: nodes (B,N,C), idx (B,N,K) -> (B,N,K,C)."""
    B, N, K = idx.shape
    b = torch.arange(B).view(B, 1, 1).expand(B, N, K)
    return nodes[b, idx]


# ----------------------------------------------------------------------------- features
def knn(X: Tensor, mask: Tensor, top_k: int) -> Tuple[Tensor, Tensor]:
    """`RNAFeatures._dist` on the P atoms (feature.py:42-48, 183): -> (E_idx (B,N,K'), mask_attend (B,N,K') bool).
    The K' = min(top_k, N) nearest residues INCLUDING the residue itself (distance sqrt(1e-6)); pairs with a padded member sit at
    10000 + (row max + 1), so a valid row lists its valid residues first (ascending distance, ties by index here) and
    `mask_attend` (feature.py:186-187) drops the rest."""
    P = X[:, :, 0, :]
    m2 = mask.unsqueeze(1) * mask.unsqueeze(2)
    dX = P.unsqueeze(1) - P.unsqueeze(2)
    D = (1.0 - m2) * 10000 + m2 * torch.sqrt((dX ** 2).sum(3) + 1e-6)
    D_adj = D + (1.0 - m2) * (D.max(-1, keepdim=True).values + 1)
    K = min(top_k, D_adj.shape[-1])
    order = torch.argsort(D_adj, dim=-1, stable=True)[..., :K]
    attend = (mask.unsqueeze(-1) * _gather_nodes(mask.unsqueeze(-1), order).squeeze(-1)) == 1
    return order, attend


def dihedrals(X: Tensor) -> Tensor:
    """`RNAFeatures._dihedrals` (feature.py:136-155): 6 angles per residue from the flattened 6-atom chain with a stride-5
    difference (as written in the reference), cos | sin -> (B,N,12).  Windows run over the PADDED chain (zeros)."""
    B, N = X.shape[:2]
    Xf = X[:, :, :6, :].reshape(B, 6 * N, 3)
    dX = Xf[:, 5:, :] - Xf[:, :-5, :]
    U = F.normalize(dX, dim=-1)
    u_2, u_1, u_0 = U[:, :-2, :], U[:, 1:-1, :], U[:, 2:, :]
    n_2 = F.normalize(torch.linalg.cross(u_2, u_1), dim=-1)
    n_1 = F.normalize(torch.linalg.cross(u_1, u_0), dim=-1)
    cosD = torch.clamp((n_2 * n_1).sum(-1), -1 + 1e-7, 1 - 1e-7)
    D = torch.sign((u_2 * n_1).sum(-1)) * torch.acos(cosD)
    D = F.pad(D, (3, 4), "constant", 0).view(B, N, 6)
    return torch.cat((torch.cos(D), torch.sin(D)), 2)


def orientations(X: Tensor, E_idx: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """`RNAFeatures._orientations_coarse` (feature.py:83-134) -> (V_direct (B,N,9), E_direct (B,N,K,15), E_orient (B,N,K,4)).
    Local frame at C3' from the flattened chain: u = unit(O3' - C3'), v = unit(P_next - O3'); the last residue of the TENSOR has
    the zero frame (feature.py:101)."""
    B, N = X.shape[:2]
    V = X
    Xf = X[:, :, :6, :].reshape(B, 6 * N, 3)
    U = _normalize_nan0(Xf[:, 1:, :] - Xf[:, :-1, :])
    u_0, u_1 = U[:, :-2, :], U[:, 1:-1, :]
    n_0 = _normalize_nan0(torch.linalg.cross(u_0, u_1))[:, 4::6, :]
    b_1 = _normalize_nan0(u_0 - u_1)[:, 4::6, :]
    C3 = Xf[:, 4::6, :]
    Q = torch.stack((b_1, n_0, torch.linalg.cross(b_1, n_0)), 2).reshape(B, -1, 9)
    Q = F.pad(Q, (0, 0, 0, 1), "constant", 0)                                  # (B,N,9)
    Qn = _gather_nodes(Q, E_idx).view(B, N, -1, 3, 3)
    nb = torch.stack([_gather_nodes(V[:, :, a, :], E_idx) for a in (0, 1, 2, 3, 5)], dim=3)   # P O5' C5' C4' O3'  (B,N,K,5,3)
    Q3 = Q.view(B, N, 3, 3).unsqueeze(2)                                        # (B,N,1,3,3)
    dX = nb - C3[:, :, None, None, :]
    dU = torch.matmul(Q3[:, :, :, None, :, :], dX[..., None]).squeeze(-1)
    E_direct = _normalize_nan0(dU).reshape(B, N, dU.shape[2], 15)
    R = torch.matmul(Q3.transpose(-1, -2), Qn)
    E_orient = quaternions(R)
    dXi = V[:, :, [0, 2, 3], :] - C3.unsqueeze(-2)
    dUi = torch.matmul(Q3, dXi.unsqueeze(-1)).squeeze(-1)
    V_direct = _normalize_nan0(dUi).reshape(B, N, 9)
    return V_direct, E_direct, E_orient


def quaternions(R: Tensor) -> Tensor:
    """`RNAFeatures._quaternions` (feature.py:62-81)."""
    diag = torch.diagonal(R, dim1=-2, dim2=-1)
    Rxx, Ryy, Rzz = diag.unbind(-1)
    mag = 0.5 * torch.sqrt(torch.abs(1 + torch.stack([Rxx - Ryy - Rzz, -Rxx + Ryy - Rzz, -Rxx - Ryy + Rzz], -1)))
    r = lambda i, j: R[..., i, j]
    signs = torch.sign(torch.stack([r(2, 1) - r(1, 2), r(0, 2) - r(2, 0), r(1, 0) - r(0, 1)], -1))
    xyz = signs * mag
    w = torch.sqrt(F.relu(1 + diag.sum(-1, keepdim=True))) / 2.0
    return F.normalize(torch.cat((xyz, w), -1), dim=-1)


def raw_features(X: Tensor, mask: Tensor, cfg: RDesignConfig) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """The feature tensors of `RNAFeatures.forward` (feature.py:168-223) before the embeddings, on the padded layout:
    -> (node (B,N,101), edge (B,N,K,115), E_idx (B,N,K), mask_attend (B,N,K))."""
    atom = [X[:, :, a, :] for a in range(6)]                                    # P O5' C5' C4' C3' O3'
    E_idx, attend = knn(X, mask, cfg.k_neighbors)
    V_angle = dihedrals(X)
    pair_d = lambda A, Bv: torch.sqrt(((A - Bv) ** 2).sum(-1) + 1e-6)
    V_dist = torch.cat([_rbf(pair_d(atom[a], atom[0]), cfg.num_rbf) for a in (1, 2, 3, 4, 5)], -1)    # O5'-P ... O3'-P  (:196-203)
    V_direct, E_direct, E_orient = orientations(X, E_idx)
    E_dist = []
    for a in (0, 1, 2, 3, 4, 5):                                                # 'P-P','O5_-P',...: atom1 of the CENTRE, P of the neighbour (:211-219, 58-60)
        D = torch.sqrt(((atom[a][:, :, None, :] - atom[0][:, None, :, :]) ** 2).sum(-1) + 1e-6)
        E_dist.append(_rbf(torch.gather(D, 2, E_idx), cfg.num_rbf))
    node = torch.cat([V_angle, V_dist, V_direct], -1)                           # angle, distance, direction (:221-226)
    edge = torch.cat([E_orient, torch.cat(E_dist, -1), E_direct], -1)           # orientation, distance, direction (:228-233)
    return node, edge, E_idx, attend


# ----------------------------------------------------------------------------- model
def forward(X: Tensor, mask: Tensor, sd: Mapping[str, Tensor], cfg: RDesignConfig, taps: dict = None) -> Tuple[Tensor, Tensor]:
    """`RNAModel.forward` + `readout` (rdesign.py:82-88, 104) -> (h_V (N_valid,128), logits (N_valid,4)), residues in
    (batch, position) order as `masked_select` produces them (feature.py:187-189)."""
    node, edge, E_idx, attend = raw_features(X, mask, cfg)
    B, N, K = E_idx.shape
    mb = mask == 1
    h_V = _custom_norm(node[mb] @ sd["features.node_embedding.weight"].T + sd["features.node_embedding.bias"],
                       sd["features.norm_nodes.gain"], sd["features.norm_nodes.bias"])            # :236
    h_E = _custom_norm(edge[attend] @ sd["features.edge_embedding.weight"].T + sd["features.edge_embedding.bias"],
                       sd["features.norm_edges.gain"], sd["features.norm_edges.bias"])            # :237
    shift = (mask.sum(1).cumsum(0) - mask.sum(1)).long()                                          # :241-246
    src = (shift.view(B, 1, 1) + E_idx)[attend]
    dst = (shift.view(B, 1, 1) + torch.arange(N).view(1, N, 1).expand(B, N, K))[attend]
    if taps is not None:
        taps.update(node_raw=node, edge_raw=edge, E_idx=E_idx, attend=attend, h_V0=h_V, h_E=h_E, dst=dst, src=src)
    n_nodes = h_V.shape[0]
    for l in range(cfg.num_mpnn_layers):                                                           # rdesign.py:84-86, mpnn.py:31-37
        p = f"mpnn_layers.{l}"
        x = torch.cat([h_E, h_V[dst], h_V[src]], -1)
        for i in range(cfg.num_message_layers):
            x = _gelu(x @ sd[f"{p}.message_layers.{3 * i}.weight"].T + sd[f"{p}.message_layers.{3 * i}.bias"])
        dh = torch.zeros(n_nodes, x.shape[1], dtype=x.dtype).index_add_(0, dst, x) / cfg.scale
        h_V = F.layer_norm(h_V + dh, (cfg.hidden_dim,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
        y = h_V
        for i in range(cfg.num_dense_layers):
            y = _gelu(y @ sd[f"{p}.dense.{3 * i}.weight"].T + sd[f"{p}.dense.{3 * i}.bias"])
        y = y @ sd[f"{p}.dense.{3 * cfg.num_dense_layers}.weight"].T + sd[f"{p}.dense.{3 * cfg.num_dense_layers}.bias"]
        h_V = F.layer_norm(h_V + y, (cfg.hidden_dim,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
        if taps is not None:
            taps[f"h_V{l + 1}"] = h_V
    x = h_V
    for i in range(max(cfg.num_readout_layers - 1, 0)):                                            # functional.py:113-121
        x = _gelu(x @ sd[f"readout.readout_layers.{3 * i}.weight"].T + sd[f"readout.readout_layers.{3 * i}.bias"])
    last = 3 * max(cfg.num_readout_layers - 1, 0)
    logits = x @ sd[f"readout.readout_layers.{last}.weight"].T + sd[f"readout.readout_layers.{last}.bias"]
    return h_V, logits
