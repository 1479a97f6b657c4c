"""``RNAModel`` of the reference's ``rdesign`` package (``rdesign/model/rdesign.py:18-209``) on the MI355X path.

The module tree (``features.node_embedding`` ... ``mpnn_layers.{i}.message_layers.{0,3,6}`` ... ``readout.readout_layers.{j}``)
reproduces the reference's ``state_dict`` keys and shapes, so a reference checkpoint's ``state_dict`` loads with
``load_state_dict``; the sub-modules are parameter holders, the arithmetic is ``rdesign_forward`` of ``librnampnn_hip.so``.
Inference surface only (``forward``, ``readout``, ``validation_step`` / ``test_step`` metrics, ``predict``); the XGBoost head of
``predict`` is out of scope like the main model's (xgboost is not installed: the reference's own ``NotFittedError`` branch, argmax of
``Readout``, is the one taken - ``rdesign.py:152-155``).  PARITY UNPINNED (``oracle/rdesign_oracle.py``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native
from rnampnn.model._base import _prep, _stream

_PREC = {"f32": _native.PREC_F32, "bf16": _native.PREC_BF16}


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("this sub-module only holds parameters: call RNAModel.forward (HIP path, no CPU fallback)")


class Normalize(_Holder):
    """``functional.Normalize`` (rdesign/model/functional.py:79-96)."""

    def __init__(self, features: int):
        super().__init__()
        self.gain = nn.Parameter(torch.ones(features))
        self.bias = nn.Parameter(torch.zeros(features))


class RNAFeatures(_Holder):
    """Parameters of ``feature.RNAFeatures`` (rdesign/model/feature.py:8-27)."""

    def __init__(self, hidden: int):
        super().__init__()
        self.node_embedding = nn.Linear(101, hidden)
        self.edge_embedding = nn.Linear(115, hidden)
        self.norm_nodes = Normalize(hidden)
        self.norm_edges = Normalize(hidden)


def _mlp(sizes, act_last: bool, dropout: float) -> nn.Sequential:
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if act_last or i + 2 < len(sizes):
            layers += [nn.GELU(), nn.Dropout(dropout)]
    return nn.Sequential(*layers)


class MPNNLayer(_Holder):
    """Parameters of ``mpnn.MPNNLayer`` (rdesign/model/mpnn.py:5-29)."""

    def __init__(self, hidden, num_in, num_message_layers, num_dense_layers, dim_dense_layers, dropout=0.1, scale=30):
        super().__init__()
        self.scale = scale
        self.norm1 = nn.LayerNorm(hidden)
        self.norm2 = nn.LayerNorm(hidden)
        self.message_layers = _mlp([hidden + num_in] + [hidden] * num_message_layers, True, dropout)
        self.dense = _mlp([hidden] + [dim_dense_layers] * num_dense_layers + [hidden], False, dropout)


class Readout(nn.Module):
    """``functional.Readout`` (rdesign/model/functional.py:98-126); ``forward`` runs ``rdesign_readout``."""

    def __init__(self, owner: "RNAModel", hidden: int, readout_hidden_dim: int, num_layers: int, dropout: float):
        super().__init__()
        self.readout_layers = _mlp([hidden] + [readout_hidden_dim] * max(num_layers - 1, 0) + [4], False, dropout)
        object.__setattr__(self, "_owner", owner)          # not a sub-module: no cycle in the module tree

    def forward(self, res_embedding: torch.Tensor) -> torch.Tensor:
        return self._owner._readout_native(res_embedding)


class RNAModel(nn.Module):
    def __init__(self, hidden_dim: int = 128, vocab_size: int = 4, k_neighbors: int = 25, dropout: float = 0.1,
                 node_feat_types=None, edge_feat_types=None, num_message_layers: int = 3, num_dense_layers: int = 3,
                 dim_dense_layers: int = 256, num_mpnn_layers: int = 9, readout_hidden_dim: int = 256,
                 num_readout_layers: int = 0, lr: float = 0.002, n_estimators: int = 100, xgb_max_depth: int = 6,
                 xgb_learning_rate: float = 0.1, xgb_subsample: float = 0.8, xgb_colsample_bytree: float = 0.8,
                 precision: str = "bf16"):
        super().__init__()
        if node_feat_types not in (None, ["angle", "distance", "direction"]) or \
                edge_feat_types not in (None, ["orientation", "distance", "direction"]):
            raise NotImplementedError("the HIP path builds the reference's default feature set (101 node / 115 edge features)")
        if vocab_size != 4:
            raise NotImplementedError("vocab_size must be 4 (AUCG)")
        if precision not in _PREC:
            raise ValueError(f"precision must be one of {sorted(_PREC)}")
        self.name, self.version, self.precision = "RDesign-X", 0, precision
        self.hparams = dict(hidden_dim=hidden_dim, vocab_size=vocab_size, k_neighbors=k_neighbors, dropout=dropout,
                            num_message_layers=num_message_layers, num_dense_layers=num_dense_layers,
                            dim_dense_layers=dim_dense_layers, num_mpnn_layers=num_mpnn_layers,
                            readout_hidden_dim=readout_hidden_dim, num_readout_layers=num_readout_layers, lr=lr)
        self.hidden_dim, self.vocab = hidden_dim, vocab_size
        self._handle = _native.Handle(self.hparams, _PREC[precision])      # validates like the reference constructor would fail later
        self.features = RNAFeatures(hidden_dim)
        self.mpnn_layers = nn.ModuleList([
            MPNNLayer(hidden_dim, hidden_dim * 2, num_message_layers, num_dense_layers, dim_dense_layers, dropout=dropout)
            for _ in range(num_mpnn_layers)])
        self.readout = Readout(self, hidden_dim, readout_hidden_dim, num_readout_layers, dropout)
        self.loss_fn = nn.CrossEntropyLoss()
        self.val_step_outputs = {"val_loss": [], "correct": [], "len": [], "recovery_rates": []}
        self.test_step_outputs = {"test_loss": [], "correct": [], "len": [], "recovery_rates": []}
        schema = self._handle.weight_schema()
        named = dict(self.named_parameters())
        assert [k for k, _, _ in schema] == list(named), "parameter table of the library != module tree"
        assert all(named[k].numel() == n for k, n, _ in schema)
        self._slices = [(named[k], off, n) for k, n, off in schema]
        self._flat: Optional[torch.Tensor] = None
        self._sig = None
        self._ws: Optional[torch.Tensor] = None
        self._ext_version = 0

    # ------------------------------------------------------------------ native state
    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def _ensure(self) -> torch.device:
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("the rdesign HIP path runs on an MI355X: move the module to 'cuda' first (there is no CPU fallback)")
        lib = _native.lib()
        flat = self._flat
        if flat is None or flat.device != dev or any(p.data_ptr() != flat.data_ptr() + 4 * off for p, off, _ in self._slices):
            flat = torch.zeros(int(lib.rdesign_param_numel(self._handle.ptr)), dtype=torch.float32, device=dev)
            with torch.no_grad():
                for p, off, n in self._slices:
                    flat[off: off + n].copy_(p.data.reshape(-1))
                    p.data = flat[off: off + n].view(p.shape)
            with torch.cuda.device(dev):
                _native.check(lib.rdesign_use_weight_arena(self._handle.ptr, C.c_void_p(flat.data_ptr()), _stream(dev)))
            self._flat, self._sig = flat, None
        sig = (self._ext_version,) + tuple(p._version for p, _, _ in self._slices)
        if sig != self._sig:
            with torch.cuda.device(dev):
                _native.check(lib.rdesign_finalize_weights(self._handle.ptr, _stream(dev)))
            self._sig = sig
        return dev

    def _ws_args(self, B: int, T: int, dev, readout_rows: int = 0):
        lib = _native.lib()
        need = int(lib.rdesign_readout_workspace_bytes(self._handle.ptr, readout_rows) if readout_rows
                   else lib.rdesign_workspace_bytes(self._handle.ptr, B, T))
        if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        base = self._ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        return C.c_void_p(aligned), C.c_size_t(self._ws.numel() - (aligned - base))

    @staticmethod
    def _check_mask(mask: torch.Tensor) -> None:
        m = mask.detach().cpu()
        if not bool((((m == 0) | (m == 1)).all()) and bool((m[:, 1:] <= m[:, :-1]).all())):
            raise ValueError("mask must be a 0/1 prefix mask per RNA (rdesign/utils/data.py:104-115 compacts valid residues)")

    def _run(self, X, mask, want=("h_V", "logits"), n_valid: Optional[int] = None):
        dev = self._ensure()
        if X.dim() != 4 or X.shape[2:] != (6, 3) or mask.shape != X.shape[:2]:
            raise ValueError(f"X must be (B, T, 6, 3) and mask (B, T); got {tuple(X.shape)}, {tuple(mask.shape)}")
        B, T = int(X.shape[0]), int(X.shape[1])
        if B == 0 or T == 0:
            raise ValueError("empty batch")
        if n_valid is None:                      # a caller that passes n_valid vouches for the mask (no host sync on the hot path)
            self._check_mask(mask)
        n = int(mask.sum().item()) if n_valid is None else n_valid
        Xd, md = _prep(X, dev), _prep(mask, dev)
        K = self.hparams["k_neighbors"]
        out = {}
        alloc = dict(h_V=lambda: torch.zeros(B * T, 128, device=dev), logits=lambda: torch.zeros(B * T, 4, device=dev),
                     edge_index=lambda: torch.full((B, T, K), -1, dtype=torch.int64, device=dev),
                     node_raw=lambda: torch.zeros(B * T, 101, device=dev), edge_raw=lambda: torch.zeros(B * T * K, 115, device=dev))
        for k in want:
            out[k] = alloc[k]()
        ptr = lambda k: C.c_void_p(out[k].data_ptr()) if k in out else None
        ws, ws_bytes = self._ws_args(B, T, dev)
        with torch.cuda.device(dev):
            _native.check(_native.lib().rdesign_forward(self._handle.ptr, C.c_void_p(Xd.data_ptr()), C.c_void_p(md.data_ptr()), B, T,
                                                        ptr("h_V"), ptr("logits"), ptr("edge_index"), ptr("node_raw"), ptr("edge_raw"),
                                                        ws, ws_bytes, _stream(dev)))
        for k in ("h_V", "logits", "node_raw"):
            if k in out:
                out[k] = out[k][:n]
        if "edge_raw" in out:
            out["edge_raw"] = out["edge_raw"][:n * K]
        return out

    def _readout_native(self, h_V: torch.Tensor) -> torch.Tensor:
        dev = self._ensure()
        x = _prep(h_V, dev)
        if x.dim() != 2 or x.shape[1] != 128 or x.shape[0] == 0:
            raise ValueError(f"res_embedding must be (N, 128), N > 0; got {tuple(x.shape)}")
        n = int(x.shape[0])
        logits = torch.empty(n, 4, device=dev)
        ws, ws_bytes = self._ws_args(1, n, dev, readout_rows=n)
        with torch.cuda.device(dev):
            _native.check(_native.lib().rdesign_readout(self._handle.ptr, C.c_void_p(x.data_ptr()), n, C.c_void_p(logits.data_ptr()),
                                                        ws, ws_bytes, _stream(dev)))
        return logits

    # ------------------------------------------------------------------ reference surface
    def forward(self, X, S, mask, is_predict: bool = False):
        """-> (h_V (N, 128), S (N,)): packed over the valid residues (``rdesign.py:82-88``, ``feature.py:187-189``)."""
        if self.training:
            raise NotImplementedError("the rdesign HIP path is inference-only (training stays on the main model, SURVEY.md 8 F3): call .eval()")
        out = self._run(X, mask, want=("h_V",))
        S_packed = torch.masked_select(S.to(out["h_V"].device), mask.to(out["h_V"].device) == 1)
        return out["h_V"], S_packed

    def forward_logits(self, X, mask) -> torch.Tensor:
        """``readout(forward(...)[0])`` in one call (no second launch sequence)."""
        return self._run(X, mask, want=("logits",))["logits"]

    def configure_optimizers(self):
        optimizer = torch.optim.Adam(self.parameters(), lr=self.hparams["lr"])
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=40, gamma=0.8)
        return [optimizer], [scheduler]

    def training_step(self, batch):
        raise NotImplementedError("the rdesign HIP path is inference-only; its backward is not built (DESIGN.md section 9)")

    def _eval_step(self, batch, store, loss_key):
        X, S, mask, lengths, _ = batch
        out = self._run(X, mask, want=("logits",))
        logits = out["logits"]
        S_p = torch.masked_select(S.to(logits.device), mask.to(logits.device) == 1)
        loss = self.loss_fn(logits, S_p)
        correct = (logits.argmax(dim=-1) == S_p).to(torch.float32)
        rates, start = [], 0
        for n in [int(v) for v in lengths]:
            rates.append(float(correct[start:start + n].sum() / n)); start += n
        store[loss_key].append(loss * correct.shape[0])
        store["correct"].append(correct.sum(dim=-1).item())
        store["len"].append(correct.shape[0])
        store["recovery_rates"] += rates
        return loss, rates

    def validation_step(self, batch):
        loss, rates = self._eval_step(batch, self.val_step_outputs, "val_loss")
        return {"validation loss": loss, "recovery_rates": rates}

    def test_step(self, batch):
        loss, rates = self._eval_step(batch, self.test_step_outputs, "test_loss")
        return {"test loss": loss, "recovery_rates": rates}

    def predict(self, batch, batch_id, output_dir, filename):
        """``rdesign.py:143-173``: argmax of the read-out (the reference's branch for an unfitted XGBoost head), one CSV row per RNA."""
        self.eval()
        X, S, mask, lengths, pdb_ids = batch
        samples = self._run(X, mask, want=("logits",))["logits"].argmax(dim=-1).tolist()
        os.makedirs(output_dir, exist_ok=True)
        start = 0
        with open(os.path.join(output_dir, filename), "a") as f:
            if batch_id == 0:
                f.write("pdb_id,seq\n")
            for length, pdb_id in zip(lengths, pdb_ids):
                n = int(length)
                f.write(f"{pdb_id},{''.join('AUCG'[i] for i in samples[start:start + n])}\n")
                start += n
