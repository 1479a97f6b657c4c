"""CPU ORACLE for the RNA-MPNN forward path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch-CPU restatement of the arithmetic of the reference's
``rnampnn.model`` forward path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path (the HIP library
behind ``include/rnampnn_hip.h``) never does and fails loudly without its ``.so``.

Pinning: the reference ships no numerical fixtures for this path,
so the oracle is pinned against outputs of the reference's own L1 modules, generated in
the build container by ``tools/gen_golden.py`` (which imports them from /root/reference)
and committed under ``tests/golden/``; ``tests/test_oracle_golden.py`` is the check.

Each function cites the reference lines it restates (paths relative to the reference root).
All functions take/return torch CPU tensors; dtype follows the inputs (fp32 or fp64).
Weights come as a ``state_dict``-shaped mapping with the reference's key names
(``rnampnn/model/rnampnn.py:94-134``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F

LEPS = 1e6     # rnampnn/config/glob.py:16
SEPS = 1e-6    # rnampnn/config/glob.py:17
Tensor = torch.Tensor


@dataclass
class OracleConfig:
    """The model hyper-parameters of ``RNAMPNN.__init__`` (rnampnn/model/rnampnn.py:19-47)."""
    num_res_neighbours: int = 3
    res_embedding_dim: int = 128
    num_embedding_attn_layers: int = 0
    num_embedding_heads: int = 8
    embedding_ffn_dim: int = 512
    num_embedding_ffn_layers: int = 3
    res_edge_embedding_dim: int = 128
    depth_res_edge_feature: int = 2
    num_res_mpnn_layers: int = 10
    depth_res_mpnn: int = 2
    num_mpnn_edge_layers: int = 2
    padding_len: int = 4500
    num_post_fusion_attn_layers: int = 2
    num_post_fusion_heads: int = 8
    post_fusion_ffn_dim: int = 512
    num_post_fusion_ffn_layers: int = 3
    num_raw_ffn_dim: int = 512
    num_raw_ffn_layers: int = 3
    raw_embedding_dim: int = 128
    readout_hidden_dim: int = 512
    num_readout_layers: int = 2


# ----------------------------------------------------------------------------- dropout (train mode)
# The reference applies nn.Dropout(p) after every GELU (mpnn.py:140,150; feature.py:200; functional.py:69,124,184)
# and dropout on the attention probabilities (nn.MultiheadAttention(dropout=p), functional.py:109), with masks from
# torch's global RNG.  The HIP training path draws the keep decision of an element from a 32-bit counter hash of
# (seed, site, element index) instead (csrc/kernels_train.h: TDrop); this class restates that hash so oracle autograd
# and the HIP backward see the same masks.  element index = row * D + channel, rows in the packed order of the valid
# residues (node row p = cu[b] + t, edge row p*k + slot); attention: ((query row * heads + head) << 13) + key.
SITE_EE, SITE_EMB_FFN, SITE_EMB_ATT, SITE_POST_FFN, SITE_POST_ATT, SITE_RAW, SITE_RO = 1, 10, 20, 30, 40, 50, 60


def site_msg(layer: int, i: int) -> int:
    return 100 + 4 * layer + i


def site_edge(layer: int, i: int) -> int:
    return 102 + 4 * layer + i


def dropout_multiplier(seed: int, site: int, idx, p: float):
    """-> float32 numpy array of 0 or 1/(1-p) for the element indices ``idx`` (any integer array): the counter hash of
    csrc/kernels_train.hip (drop_mul / drop_pair), all arithmetic modulo 2^32.  One murmur3-finaliser hash per PAIR of elements
    (2P, 2P + 1): its low 16 bits decide the even element, its high 16 bits the odd one; keep iff >= round(p * 65536)."""
    import numpy as np
    u32 = np.uint32
    i64 = np.asarray(idx).astype(np.uint64)
    pair = i64 >> np.uint64(1)
    lo, hi = (pair & np.uint64(0xFFFFFFFF)).astype(u32), (pair >> np.uint64(32)).astype(u32)
    const = (int(site) * 0x85EBCA6B + (int(seed) & 0xFFFFFFFF) + ((int(seed) >> 32) & 0xFFFFFFFF) * 0x27D4EB2F) & 0xFFFFFFFF
    with np.errstate(over="ignore"):
        x = lo + hi * u32(0xC2B2AE35) + u32(const)
        x ^= x >> u32(16); x *= u32(0x85EBCA6B)
        x ^= x >> u32(13); x *= u32(0xC2B2AE35)
        x ^= x >> u32(16)
    r16 = np.where((i64 & np.uint64(1)) == 1, x >> u32(16), x & u32(0xFFFF)).astype(np.int64)
    thresh = int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))
    keep = r16 >= thresh
    return np.where(keep, np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32)


class Drop:
    """Dropout context of one forward: p, seed and the packed row of every (b, t)."""

    def __init__(self, p: float, seed: int, mask: Tensor):
        self.p, self.seed = float(p), int(seed)
        n = mask.sum(-1).long()
        cu = torch.cumsum(n, 0) - n
        t = torch.arange(mask.shape[1]).unsqueeze(0)
        self.rows = torch.where(mask > 0, cu.unsqueeze(1) + t, torch.zeros_like(t))      # (B,T); padded rows: don't care

    def nodes(self, x: Tensor, site: int) -> Tensor:           # x (B,T,D)
        if self.p <= 0.0:
            return x
        D = x.shape[-1]
        idx = self.rows.unsqueeze(-1) * D + torch.arange(D)
        return x * torch.from_numpy(dropout_multiplier(self.seed, site, idx.numpy(), self.p)).to(x.dtype)

    def edges(self, x: Tensor, site: int) -> Tensor:           # x (B,T,k,D)
        if self.p <= 0.0:
            return x
        k, D = x.shape[2], x.shape[3]
        er = self.rows.unsqueeze(-1) * k + torch.arange(k)
        idx = er.unsqueeze(-1) * D + torch.arange(D)
        return x * torch.from_numpy(dropout_multiplier(self.seed, site, idx.numpy(), self.p)).to(x.dtype)

    def attention(self, att: Tensor, site: int) -> Tensor:     # att (B,H,T,T) probabilities
        if self.p <= 0.0:
            return att
        H, T = att.shape[1], att.shape[3]
        q = self.rows.view(-1, 1, att.shape[2], 1) * H + torch.arange(H).view(1, H, 1, 1)
        idx = q * 8192 + torch.arange(T).view(1, 1, 1, T)
        return att * torch.from_numpy(dropout_multiplier(self.seed, site, idx.numpy(), self.p)).to(att.dtype)

    def apply(self, x: Tensor, site: int) -> Tensor:
        return self.edges(x, site) if x.dim() == 4 else self.nodes(x, site)


# ----------------------------------------------------------------------------- helpers
def _gelu(x: Tensor) -> Tensor:
    # nn.GELU() default = exact erf form (mpnn.py:139,149; functional.py:68,123,183)
    return 0.5 * x * (1.0 + torch.erf(x * 0.7071067811865476))


def _linear(x: Tensor, sd: Mapping[str, Tensor], prefix: str) -> Tensor:
    return x @ sd[prefix + ".weight"].T + sd[prefix + ".bias"]


def _mlp_all_gelu(x: Tensor, sd, prefix: str, depth: int, drop: Optional["Drop"] = None, site0: int = 0) -> Tensor:
    """Sequential(Linear, GELU, Dropout) x depth - ends in GELU (feature.py:195-203, mpnn.py:135-152)."""
    for i in range(depth):
        x = _gelu(_linear(x, sd, f"{prefix}.{3 * i}"))
        if drop is not None:
            x = drop.apply(x, site0 + i)
    return x


def _ffn_last_plain(x: Tensor, sd, prefix: str, n_hidden: int, drop: Optional["Drop"] = None, site0: int = 0) -> Tensor:
    """n_hidden x (Linear, GELU, Dropout) + final plain Linear (functional.py:119-127,179-187)."""
    for i in range(n_hidden):
        x = _gelu(_linear(x, sd, f"{prefix}.{3 * i}"))
        if drop is not None:
            x = drop.apply(x, site0 + i)
    return _linear(x, sd, f"{prefix}.{3 * n_hidden}")


def _unit(v: Tensor, eps: float) -> Tensor:
    # F.normalize: v / max(||v||, eps)
    return v / torch.clamp(torch.sqrt((v * v).sum(-1, keepdim=True)), min=eps)


def _cross(a: Tensor, b: Tensor) -> Tensor:
    return torch.stack((a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                        a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                        a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]), dim=-1)


# ----------------------------------------------------------------------------- A8
def graph_norm(x: Tensor, mask: Tensor, scale: Tensor, shift: Tensor) -> Tensor:
    """GraphNormalization (functional.py:18-48).  The node axis of ``x`` is T_tot: padded rows
    enter the variance as (0 - mean)^2, i.e. var = [sum_valid (x-mu)^2 + (T_tot-n) mu^2] / n."""
    m = mask.unsqueeze(-1).to(x.dtype)
    xm = x * m
    cnt = m.sum(dim=1, keepdim=True)
    cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)
    mean = xm.sum(dim=1, keepdim=True) / cnt
    var = ((xm - mean) ** 2).sum(dim=1, keepdim=True) / cnt
    y = (x - mean) / torch.sqrt(var + SEPS)
    return (y * scale.reshape(1, 1, -1) + shift.reshape(1, 1, -1)) * m


# ----------------------------------------------------------------------------- A2
def knn_graph(coords: Tensor, mask: Tensor, k: int) -> Tensor:
    """Residue k-NN graph (feature.py:205-256) -> edge_index (B,T,k) int64, -1 = no edge.

    Canonical form of the reference result: real neighbours ascending by centroid distance
    (ties by lower index); if a valid row has fewer than k real neighbours and the batch is
    padded (T > n) the reference keeps ONE extra edge to a padded residue (fp32: self and all
    padded residues tie at exactly 1e6 and CPU topk returns a padded index).  All padded
    residues are interchangeable (zero coords from the collate, zero embedding), so the
    oracle names the first one, index n.  With T == n that slot is the row itself -> -1."""
    B, T = mask.shape
    cen = coords.mean(dim=2)                                            # :218
    d = torch.sqrt(((cen.unsqueeze(1) - cen.unsqueeze(2)) ** 2).sum(-1) + SEPS)   # :220-221
    m2 = mask.unsqueeze(1) * mask.unsqueeze(2)
    d = d * m2 + (1.0 - m2) * LEPS                                      # :223-224
    d = d + torch.eye(T, dtype=d.dtype).unsqueeze(0) * LEPS             # :226-227
    n = mask.sum(-1).long()                                             # (B,)
    order = torch.argsort(d, dim=-1, stable=True)                       # ascending, ties by index
    kk = min(k, T)
    idx = order[..., :kk]
    if k > T:
        idx = torch.cat([idx, torch.full((B, T, k - T), -1, dtype=idx.dtype)], dim=-1)
    slot = torch.arange(k).view(1, 1, k)
    nb = n.view(B, 1, 1)
    real = slot < (nb - 1).clamp(min=0)                                  # slots holding true neighbours
    phantom = (slot == (nb - 1)) & (nb < T) & (nb >= 1)                   # :248-251 keeps slot n-1
    out = torch.where(real, idx, torch.full_like(idx, -1))
    out = torch.where(phantom, nb.expand_as(out), out)
    out = torch.where((mask == 0).unsqueeze(-1), torch.full_like(out, -1), out)   # :253-254
    return out


def canonical_edge_index(edge_index: Tensor, mask: Tensor) -> Tensor:
    """Map any padded neighbour index (>= n) to n: the equivalence class the tie rule allows."""
    n = mask.sum(-1).long().view(-1, 1, 1)
    return torch.where(edge_index >= n, n.expand_as(edge_index), edge_index)


# ----------------------------------------------------------------------------- A6
def node_raw_features(coords: Tensor, mask: Tensor) -> Tensor:
    """28 intra-residue features (feature.py:298-384, 531-535) -> (B,T,28)."""
    pad = (mask == 0).unsqueeze(-1)
    m = mask.unsqueeze(-1).to(coords.dtype)
    diff = coords.unsqueeze(3) - coords.unsqueeze(2)
    dist = torch.sqrt((diff ** 2).sum(-1) + SEPS)
    iu = torch.triu_indices(7, 7, offset=1)
    d21 = dist[:, :, iu[0], iu[1]].masked_fill(pad, 1e6)               # :322-329
    c6 = coords[:, :, :6]
    v = c6[:, :, 1:] - c6[:, :, :-1]                                    # 5 bond vectors
    nrm = torch.sqrt((v * v).sum(-1))
    ang = (v[:, :, :-1] * v[:, :, 1:]).sum(-1) / (nrm[:, :, :-1] * nrm[:, :, 1:] + SEPS) * m   # :349-357
    u = _unit(v, SEPS)                                                  # :377
    nor = _unit(_cross(u[:, :, :-1], u[:, :, 1:]), SEPS)                # :381
    dih = (nor[:, :, 1:] * nor[:, :, :-1]).sum(-1) * m                  # :382-383
    return torch.cat([d21, ang, dih], dim=-1)


# ----------------------------------------------------------------------------- A3/A4
def edge_raw_features(coords: Tensor, mask: Tensor, edge_index: Tensor) -> Tensor:
    """90 cross features per edge (feature.py:386-517, 553-558) -> (B,T,k,90)."""
    B, T = mask.shape
    k = edge_index.shape[-1]
    invalid = edge_index == -1
    safe = edge_index.clamp(min=0)
    bidx = torch.arange(B).view(B, 1, 1).expand(B, T, k)
    nb = coords[bidx, safe]                                             # (B,T,k,7,3)  :280-286
    pad_row = (mask == 0).view(B, T, 1, 1, 1)
    nb = torch.where(pad_row, torch.full_like(nb, LEPS), nb)            # :293-294
    valid = (mask.view(B, T, 1, 1) * (~invalid).unsqueeze(-1)).to(coords.dtype)
    # 49 distances, index a*7+b (a = central atom)                      :414-421
    diff = coords.view(B, T, 1, 7, 1, 3) - nb.view(B, T, k, 1, 7, 3)
    dist = torch.sqrt((diff ** 2).sum(-1) + SEPS).reshape(B, T, k, 49)
    dist = dist * valid + (1 - valid) * LEPS
    # 25 cosines between unit bond vectors of the first 6 atoms         :451-467
    c6, n6 = coords[:, :, :6], nb[:, :, :, :6]
    cv = c6[:, :, 1:] - c6[:, :, :-1]
    nv = n6[:, :, :, 1:] - n6[:, :, :, :-1]
    cu, nu = _unit(cv, 1e-12), _unit(nv, 1e-12)
    ang = (cu.view(B, T, 1, 5, 1, 3) * nu.view(B, T, k, 1, 5, 3)).sum(-1).reshape(B, T, k, 25) * valid
    # 16 cosines between unit plane normals                              :493-515
    cn = _unit(_cross(cv[:, :, :-1], cv[:, :, 1:]), SEPS)
    nn_ = _unit(_cross(nv[:, :, :, :-1], nv[:, :, :, 1:]), SEPS)
    dih = (cn.view(B, T, 1, 4, 1, 3) * nn_.view(B, T, k, 1, 4, 3)).sum(-1).reshape(B, T, k, 16) * valid
    return torch.cat([dist, ang, dih], dim=-1)


# ----------------------------------------------------------------------------- A13
def rnabert(h: Tensor, mask: Tensor, sd, prefix: str, n_attn: int, n_heads: int,
            n_ffn: int, padding_len: int, drop: Optional["Drop"] = None, site_att0: int = 0, site_ffn0: int = 0) -> Tensor:
    """RNABert.forward (functional.py:153-172).  The reference zero-pads to ``padding_len`` P;
    padded keys are masked out of the softmax and padded queries are discarded, so only the
    GraphNormalization sees P (its T_tot).  Computed here on the T valid-capable rows with the
    closed-form variance term, which is exact."""
    B, T, D = h.shape
    if T > padding_len:
        raise RuntimeError(f"max_len {T} exceeds padding_len {padding_len} (functional.py:155)")
    m = mask.to(h.dtype)
    key_bias = torch.zeros(B, 1, 1, T, dtype=h.dtype).masked_fill((mask == 0).view(B, 1, 1, T), float("-inf"))
    hd = D // n_heads
    x = h
    for j in range(n_attn):
        p = f"{prefix}.bi_attention_layers.{j}"
        qkv = x @ sd[p + ".in_proj_weight"].T + sd[p + ".in_proj_bias"]
        q, kx, v = qkv.split(D, dim=-1)
        q = q.view(B, T, n_heads, hd).transpose(1, 2) / (hd ** 0.5)
        kx = kx.view(B, T, n_heads, hd).transpose(1, 2)
        v = v.view(B, T, n_heads, hd).transpose(1, 2)
        att = torch.softmax(q @ kx.transpose(-1, -2) + key_bias, dim=-1)
        att = torch.nan_to_num(att, nan=0.0)          # rows of an all-padding RNA (no valid key)
        if drop is not None:
            att = drop.attention(att, site_att0 + j)  # nn.MultiheadAttention(dropout=p): on the probabilities (functional.py:109)
        o = (att @ v).transpose(1, 2).reshape(B, T, D)
        x = x + (o @ sd[p + ".out_proj.weight"].T + sd[p + ".out_proj.bias"])     # :165
        g = f"{prefix}.graph_norm_layers.{j}"
        x = _graph_norm_ttot(x, m, sd[g + ".scale"], sd[g + ".shift"], padding_len)  # :169
    x = _ffn_last_plain(x, sd, f"{prefix}.ffn_layers", n_ffn, drop, site_ffn0)   # :170
    return x * m.unsqueeze(-1)                                          # :171-172


def _graph_norm_ttot(x: Tensor, m: Tensor, scale: Tensor, shift: Tensor, t_tot: int) -> Tensor:
    """GraphNormalization on a tensor whose node axis is conceptually ``t_tot`` long while only
    the first T rows are materialised (rows beyond T are padding = zeros, mask 0)."""
    mm = m.unsqueeze(-1)
    cnt = mm.sum(dim=1, keepdim=True)
    n_valid = cnt.clone()
    cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)
    xm = x * mm
    mean = xm.sum(dim=1, keepdim=True) / cnt
    ss = (((xm - mean) ** 2) * mm).sum(dim=1, keepdim=True) + (t_tot - n_valid) * mean ** 2
    var = ss / cnt
    y = (x - mean) / torch.sqrt(var + SEPS)
    return (y * scale.reshape(1, 1, -1) + shift.reshape(1, 1, -1)) * mm


# ----------------------------------------------------------------------------- A1-A7
def res_feature(coords: Tensor, mask: Tensor, sd, cfg: OracleConfig, drop: Optional["Drop"] = None
                ) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """ResFeature.forward (feature.py:573-592) -> (raw, h0, e0, edge_index)."""
    idx = knn_graph(coords, mask, cfg.num_res_neighbours)
    x = edge_raw_features(coords, mask, idx)
    e = _mlp_all_gelu(x, sd, "res_feature.res_edge_embedding_layers", cfg.depth_res_edge_feature, drop, SITE_EE)
    e = e.masked_fill((idx == -1).unsqueeze(-1), 0.0)                   # :564-568
    e = e.masked_fill((mask == 0).view(*mask.shape, 1, 1), 0.0)         # :569
    raw = node_raw_features(coords, mask)
    h = _linear(raw, sd, "res_feature.raw_project")                     # :536
    h = rnabert(h, mask, sd, "res_feature.res_embedding", cfg.num_embedding_attn_layers,
                cfg.num_embedding_heads, cfg.num_embedding_ffn_layers, cfg.padding_len, drop, SITE_EMB_ATT, SITE_EMB_FFN)
    h = graph_norm(h, mask, sd["res_feature.graph_norm.scale"], sd["res_feature.graph_norm.shift"])  # :591
    return raw, h, e, idx


# ----------------------------------------------------------------------------- A9-A12
def _gather_nodes(h: Tensor, idx: Tensor) -> Tensor:
    B, T, k = idx.shape
    bidx = torch.arange(B).view(B, 1, 1).expand(B, T, k)
    return h[bidx, idx.clamp(min=0)]                                    # -1 -> row 0 (mpnn.py:173-181)


def mpnn_message(h: Tensor, e: Tensor, idx: Tensor, mask: Tensor, sd, prefix: str, depth: int,
                 drop: Optional["Drop"] = None, site0: int = 0) -> Tensor:
    """ResMPNN.message (mpnn.py:154-194)."""
    h = h * mask.unsqueeze(-1).to(h.dtype)
    k = idx.shape[-1]
    x = torch.cat([h.unsqueeze(2).expand(-1, -1, k, -1), _gather_nodes(h, idx), e], dim=-1)
    msg = _mlp_all_gelu(x, sd, prefix + ".message_layers", depth, drop, site0)
    return msg * (idx != -1).unsqueeze(-1).to(h.dtype)


def mpnn_aggregate(h: Tensor, msg: Tensor, idx: Tensor, mask: Tensor) -> Tensor:
    """ResMPNN.aggregation (mpnn.py:196-227): masked MEAN over k + residual."""
    cnt = (idx != -1).sum(-1, keepdim=True).to(h.dtype)
    cnt = torch.where(cnt == 0, torch.ones_like(cnt), cnt)
    return (h + msg.sum(dim=2) / cnt) * mask.unsqueeze(-1).to(h.dtype)


def mpnn_update_edges(h: Tensor, e: Tensor, idx: Tensor, sd, prefix: str, depth: int,
                      drop: Optional["Drop"] = None, site0: int = 0) -> Tensor:
    """ResMPNN._update_edges (mpnn.py:229-265): residual, NOT masked."""
    k = idx.shape[-1]
    x = torch.cat([h.unsqueeze(2).expand(-1, -1, k, -1), _gather_nodes(h, idx), e], dim=-1)
    return e + _mlp_all_gelu(x, sd, prefix + ".edge_layers", depth, drop, site0)


def mpnn_layer(h: Tensor, e: Tensor, idx: Tensor, mask: Tensor, sd, layer: int, cfg: OracleConfig,
               update_edges: bool = True, drop: Optional["Drop"] = None) -> Tuple[Tensor, Tensor]:
    """ResMPNN.forward (mpnn.py:283-294)."""
    p = f"res_mpnn_layers.{layer}"
    msg = mpnn_message(h, e, idx, mask, sd, p, cfg.depth_res_mpnn, drop, site_msg(layer, 0))
    h = mpnn_aggregate(h, msg, idx, mask)
    h = graph_norm(h, mask, sd[p + ".graph_norm.scale"], sd[p + ".graph_norm.shift"])
    if update_edges:
        e = mpnn_update_edges(h, e, idx, sd, p, cfg.num_mpnn_edge_layers, drop, site_edge(layer, 0))
    return h, e


# ----------------------------------------------------------------------------- A14-A16
def raw_ffn(raw: Tensor, mask: Tensor, sd, cfg: OracleConfig, drop: Optional["Drop"] = None) -> Tensor:
    """RawFFN.forward (functional.py:200-202)."""
    x = _ffn_last_plain(raw, sd, "raw_embedding.raw_ffn", cfg.num_raw_ffn_layers, drop, SITE_RAW)
    return graph_norm(x, mask, sd["raw_embedding.graph_norm.scale"], sd["raw_embedding.graph_norm.shift"])


def readout(x: Tensor, mask: Tensor, sd, cfg: OracleConfig, drop: Optional["Drop"] = None) -> Tensor:
    """Readout.forward (functional.py:86-90)."""
    for i in range(cfg.num_readout_layers - 1):
        x = _gelu(_linear(x, sd, f"readout.readout_layers.{3 * i}"))
        if drop is not None:
            x = drop.nodes(x, SITE_RO + i)
    x = _linear(x, sd, f"readout.readout_layers.{3 * (cfg.num_readout_layers - 1)}")
    return x * mask.unsqueeze(-1).to(x.dtype)


def forward(coords: Tensor, mask: Tensor, sd, cfg: OracleConfig, taps: Optional[dict] = None,
            skip_dead_edge_update: bool = True, dropout: float = 0.0, seed: int = 0) -> Tuple[Tensor, Tensor]:
    """RNAMPNN.forward + embedding (rnampnn.py:173-185, 269-278) -> (logits (B,T,4), embedding (B,T,256)).
    ``taps`` (a dict) receives named intermediates when given.  ``dropout`` > 0: train-mode forward with the
    counter-hash masks of ``Drop`` (eval mode, the default, is what the golden fixtures pin)."""
    drop = Drop(dropout, seed, mask) if dropout > 0.0 else None
    raw, h, e, idx = res_feature(coords, mask, sd, cfg, drop)
    if taps is not None:
        taps.update(raw=raw, h0=h, e0=e, edge_index=idx)
    L = cfg.num_res_mpnn_layers
    for l in range(L):
        dead = skip_dead_edge_update and l == L - 1 and taps is None   # layer-L edge update is never consumed
        h, e = mpnn_layer(h, e, idx, mask, sd, l, cfg, update_edges=not dead, drop=drop)
        if taps is not None:
            taps[f"h{l + 1}"] = h
            taps[f"e{l + 1}"] = e
    hp = rnabert(h, mask, sd, "post_fusion", cfg.num_post_fusion_attn_layers, cfg.num_post_fusion_heads,
                 cfg.num_post_fusion_ffn_layers, cfg.padding_len, drop, SITE_POST_ATT, SITE_POST_FFN)
    re = raw_ffn(raw, mask, sd, cfg, drop)
    emb = torch.cat((hp, re), dim=-1)
    logits = readout(emb, mask, sd, cfg, drop)
    if taps is not None:
        taps.update(h_post=hp, raw_emb=re, embedding=emb, logits=logits)
    return logits, emb


# ----------------------------------------------------------------------------- A17/A18
def loss_double_softmax(logits: Tensor, mask: Tensor, labels: Tensor) -> Tensor:
    """training/validation loss (rnampnn.py:151-154, 200-204): cross_entropy applied to softmax
    PROBABILITIES (softmax twice) of the valid positions, mean over valid nucleotides."""
    probs = torch.softmax(logits, dim=-1)[mask.bool()]
    return F.cross_entropy(probs, labels[mask.bool()], reduction="mean")


def recovery(logits: Tensor, mask: Tensor, labels: Tensor) -> Tuple[float, float, Tensor]:
    """argmax recovery (rnampnn.py:223-230; utils/train.py:18-21) -> (micro, macro, per-RNA)."""
    pred = logits.argmax(dim=-1)
    ok = ((pred == labels) & mask.bool()).to(torch.float64)
    n = mask.sum(-1).to(torch.float64)
    per_rna = ok.sum(-1) / n.clamp(min=1)
    micro = float(ok.sum() / n.sum().clamp(min=1))
    has = n > 0
    macro = float(per_rna[has].mean()) if bool(has.any()) else 0.0
    return micro, macro, per_rna


def sample_probs(logits: Tensor, temperature: float) -> Tensor:
    """Distribution the build's ``sample()`` draws from, independently per position:
    softmax(logits / temperature).  No reference counterpart (parity
    unpinned); at temperature -> 0 it degenerates to the reference argmax, which is pinned."""
    return torch.softmax(logits / temperature, dim=-1)


def state_dict_from_numpy(arrs: Mapping[str, "object"], dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: torch.as_tensor(v).to(dtype) for k, v in arrs.items()}
