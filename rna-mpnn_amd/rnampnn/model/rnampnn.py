"""Drop-in for the reference's ``rnampnn/model/rnampnn.py``: ``RNAMPNN`` with the same constructor
hyper-parameters (``rnampnn.py:19-54``), ``forward`` / ``embedding`` signatures (``:161-185,269-278``),
state-dict keys, loss (``:151-154``), optimiser (``:156-159``) and validation metrics (``:209-236``),
running on hand-written HIP kernels through the C ABI of ``include/rnampnn_hip.h``.

Not mirrored: the Lightning base class, the XGBoost read-out (third-party, not installed, no
pickle ships; ``predict`` uses the ``Readout`` argmax as ``rdesign/model/rdesign.py:152-155`` does)
and checkpoint hooks.  New (no reference counterpart): ``sample()``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn
import torch.nn.functional as F

from .. import _native
from ..config.glob import REVERSE_VOCAB
from ..utils.data import check_prefix_mask
from ._base import NativeModule, _prep, _ptr, _stream
from ._schema import DEFAULT_HPARAMS

_CHECK_MASKS = os.environ.get("RNAMPNN_CHECK_MASKS", "0") == "1"

_FIXED_ATOMS = dict(num_inside_dist_atoms=7, num_inside_angle_atoms=6, num_inside_dihedral_atoms=6,
                    num_cross_dist_atoms=7, num_cross_angle_atoms=6, num_cross_dihedral_atoms=6)


class RNAMPNN(NativeModule):
    def __init__(self, precision: Optional[str] = None, **hparams):
        super().__init__()
        unknown = set(hparams) - set(DEFAULT_HPARAMS)
        if unknown:
            raise TypeError(f"RNAMPNN.__init__() got unexpected keyword arguments {sorted(unknown)}")
        for k, v in _FIXED_ATOMS.items():
            if hparams.get(k, v) != v:
                raise NotImplementedError(f"{k}={hparams[k]}: the HIP featurisation kernels are built for the "
                                          f"reference default {v} (28 node / 90 edge raw features)")
        self.name = 'RNAMPNN-X'
        self.version = 0
        self._setup(hparams, "", precision)
        self.hparams = dict(self._hp)
        # GEMM arithmetic of the TRAINING kernels: "bf16" = the reference's bf16-mixed (MFMA, bf16 operands, f32 accumulate),
        # "f32" = exact (gradient-parity grade).  Follows the inference precision unless set.
        self.train_precision = self.precision
        self.val_step_outputs = {'val_loss': [], 'correct': [], 'len': [], 'recovery_rates': []}

    # ------------------------------------------------------------------ forward surface
    def _run(self, coords, mask, want_logits=True, want_embedding=False, T_norm: int = 0, taps: Optional[Dict] = None,
             workspace: Optional[torch.Tensor] = None):
        device = self._ensure()
        if coords.dim() != 4 or coords.shape[2:] != (7, 3):
            raise ValueError(f"coords must be (B, T, 7, 3), got {tuple(coords.shape)}")
        B, T = int(coords.shape[0]), int(coords.shape[1])
        if tuple(mask.shape) != (B, T):
            raise ValueError(f"mask must be (B, T) = {(B, T)}, got {tuple(mask.shape)}")
        if _CHECK_MASKS:            # opt-in (host sync): the kernels take sum(mask) as the length of a PREFIX mask
            check_prefix_mask(mask)
        c = _prep(coords, device)
        m = _prep(mask, device)
        k = self._hp["num_res_neighbours"]
        io = _native.RnaMpnnForwardIO()
        io.coords, io.mask, io.B, io.T, io.T_norm, io.stop_after = _ptr(c), _ptr(m), B, T, int(T_norm), 0
        out = {}
        if want_logits:
            out["logits"] = torch.empty(B, T, 4, dtype=torch.float32, device=device)
            io.logits = _ptr(out["logits"])
        if want_embedding:
            out["embedding"] = torch.empty(B, T, 256, dtype=torch.float32, device=device)
            io.embedding = _ptr(out["embedding"])
        if taps:
            shapes = {"edge_index": ((B, T, k), torch.int64), "raw": ((B, T, 28), torch.float32),
                      "h0": ((B, T, 128), torch.float32), "e0": ((B, T, k, 128), torch.float32),
                      "h_layer": ((B, T, 128), torch.float32), "e_layer": ((B, T, k, 128), torch.float32),
                      "h_post": ((B, T, 128), torch.float32), "raw_emb": ((B, T, 128), torch.float32)}
            for name in taps.get("names", ()):
                shp, dt = shapes[name]
                out[name] = torch.empty(shp, dtype=dt, device=device)
                setattr(io, name, _ptr(out[name]))
            io.tap_layer = int(taps.get("tap_layer", 0))
            io.stop_after = int(taps.get("stop_after", 0))
        with torch.cuda.device(device):
            ws, ws_bytes = self._ws_args(B, T, device, workspace)
            _native.check(_native.lib().rnampnn_forward(self._handle.ptr, C.byref(io), ws, ws_bytes, _stream(device)))
        return out

    def forward(self, coords: torch.Tensor, mask: torch.Tensor, is_predict: bool = False, T_norm: int = 0) -> torch.Tensor:
        """``RNAMPNN.forward`` (rnampnn.py:161-185): logits (B, T, 4), zero on padded rows.
        ``T_norm`` (extension): node-axis length GraphNormalization should see when this call holds
        only a shard of a padded global batch (0 = this tensor's T)."""
        if self.training:       # train mode = the reference's nn.Dropout / MHA dropout active (rnampnn.py:47,109-134)
            return self._forward_train(coords, mask, T_norm)
        return self._run(coords, mask, T_norm=T_norm)["logits"]

    # ------------------------------------------------------------------ training forward (autograd-visible)
    def manual_seed(self, seed: int) -> None:
        """Base seed of the dropout masks; call ``t`` of ``forward`` in train mode uses (seed << 32) + t."""
        self._drop_base, self._drop_calls = int(seed) & 0x7FFFFFFF, 0

    def _next_seed(self) -> int:
        self._drop_calls = getattr(self, "_drop_calls", 0) + 1
        return (getattr(self, "_drop_base", 0) << 32) + self._drop_calls

    def _train_ws(self, B: int, T: int, device):
        need = int(_native.lib().rnampnn_train_workspace_bytes(self._handle.ptr, B, T))
        if getattr(self, "_tws", None) is None or self._tws.numel() < need + 256 or self._tws.device != device:
            self._tws = None
            self._tws = torch.empty(need + 256, dtype=torch.uint8, device=device)
        base = self._tws.data_ptr()
        aligned = (base + 255) // 256 * 256
        return C.c_void_p(aligned), C.c_size_t(self._tws.numel() - (aligned - base))

    def _train_flags(self) -> int:
        if self.train_precision not in ("f32", "bf16"):
            raise ValueError("train_precision must be 'f32' or 'bf16'")
        return _native.TRAIN_BF16_MIXED if self.train_precision == "bf16" else _native.TRAIN_F32

    def _train_forward_native(self, coords, mask, T_norm: int, dropout: float, seed: int) -> torch.Tensor:
        device = self._ensure(for_mixed_training=self.train_precision == "bf16")
        B, T = int(coords.shape[0]), int(coords.shape[1])
        c, m = _prep(coords, device), _prep(mask, device)
        logits = torch.empty(B, T, 4, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            ws, ws_bytes = self._train_ws(B, T, device)
            _native.check(_native.lib().rnampnn_train_forward(self._handle.ptr, _ptr(c), _ptr(m), B, T, int(T_norm), float(dropout),
                                                              C.c_uint64(int(seed) & (2 ** 64 - 1)), self._train_flags(),
                                                              _ptr(logits), ws, ws_bytes, _stream(device)))
        return logits

    def _train_backward_native(self, dlogits: torch.Tensor) -> None:
        device = self._device()
        B, T = int(dlogits.shape[0]), int(dlogits.shape[1])
        fresh = self._bind_flat_grad(device)
        d = _prep(dlogits, device)
        with torch.cuda.device(device):
            ws, ws_bytes = self._train_ws(B, T, device)
            _native.check(_native.lib().rnampnn_train_backward(self._handle.ptr, _ptr(d), B, T, 0 if fresh else 1,
                                                               _ptr(self.flat_grad), ws, ws_bytes, _stream(device)))

    def _forward_train(self, coords, mask, T_norm: int = 0, dropout: Optional[float] = None, seed: Optional[int] = None):
        """The reference's ``self(coords, mask)`` inside ``training_step`` (rnampnn.py:199): logits carrying an autograd
        node, so ``loss.backward()`` (Lightning's automatic optimisation) reaches every parameter.  Runs the taped f32
        kernels; the parameter gradients are written into ``flat_grad`` (``p.grad`` are views of it) by the HIP backward."""
        if coords.dim() != 4 or coords.shape[2:] != (7, 3) or tuple(mask.shape) != tuple(coords.shape[:2]):
            raise ValueError(f"coords must be (B, T, 7, 3) and mask (B, T), got {tuple(coords.shape)} / {tuple(mask.shape)}")
        p = float(self._hp["dropout"] if dropout is None else dropout)
        sd = self._next_seed() if seed is None else int(seed)
        if not torch.is_grad_enabled():
            return self._train_forward_native(coords, mask, T_norm, p, sd)
        return _TrainForward.apply(self, coords, mask, int(T_norm), p, sd, *self.parameters())

    def embedding(self, coords: torch.Tensor, mask: torch.Tensor, is_predict: bool = False, T_norm: int = 0) -> torch.Tensor:
        """``RNAMPNN.embedding`` (rnampnn.py:269-278): cat(post-fusion h, raw embedding), (B, T, 256)."""
        return self._run(coords, mask, want_logits=False, want_embedding=True, T_norm=T_norm)["embedding"]

    @torch.no_grad()
    def forward_packed(self, coords_packed: torch.Tensor, cu_seqlens: torch.Tensor, max_len: int, T_norm: int = 0,
                       want_embedding: bool = False):
        """Var-len form of ``forward``: ``coords_packed`` (N,7,3) holds the valid residues
        of all RNAs back to back, ``cu_seqlens`` (B+1) int32 their exclusive prefix sum, ``max_len`` the longest
        RNA.  Returns packed logits (N,4) [and embedding (N,256)] - identical, row for row, to the valid rows
        of ``forward`` on the same batch padded to ``max_len`` - without moving or touching any padding."""
        device = self._ensure()
        if coords_packed.dim() != 3 or coords_packed.shape[1:] != (7, 3):
            raise ValueError(f"coords_packed must be (N, 7, 3), got {tuple(coords_packed.shape)}")
        if cu_seqlens.device.type == "cpu" or _CHECK_MASKS:      # free on the host (pack_batch's output); opt-in sync otherwise
            cu_h = cu_seqlens.detach().cpu().to(torch.int64)
            d = cu_h[1:] - cu_h[:-1]
            if cu_h.numel() < 2 or int(cu_h[0]) != 0 or bool((d < 0).any()) or int(cu_h[-1]) != int(coords_packed.shape[0]):
                raise ValueError("cu_seqlens must be a non-decreasing prefix sum from 0 to N = coords_packed.shape[0]")
            if int(d.max()) > int(max_len):
                raise ValueError(f"an RNA of {int(d.max())} nt exceeds max_len = {int(max_len)}")
        c = _prep(coords_packed, device)
        cu = _prep(cu_seqlens, device, torch.int32)
        B, N = int(cu.numel()) - 1, int(c.shape[0])
        logits = torch.empty(N, 4, dtype=torch.float32, device=device)
        emb = torch.empty(N, 256, dtype=torch.float32, device=device) if want_embedding else None
        lib = _native.lib()
        need = int(lib.rnampnn_workspace_bytes_packed(self._handle.ptr, B, N))
        if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=device)
        base = self._ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        with torch.cuda.device(device):
            _native.check(lib.rnampnn_forward_packed(self._handle.ptr, _ptr(c), _ptr(cu), B, N, int(max_len), int(T_norm),
                                                     _ptr(logits), _ptr(emb), C.c_void_p(aligned),
                                                     C.c_size_t(self._ws.numel() - (aligned - base)), _stream(device)))
        return (logits, emb) if want_embedding else logits

    @torch.no_grad()
    def raw_edge_features(self, coords: torch.Tensor, mask: torch.Tensor):
        """-> (feats (B,T,k,90), edge_index (B,T,k)): the raw edge features of ``ResFeature`` (``feature.py:386-517``: cross distances,
        bond-angle and dihedral-normal dot products) - a parity tap; zero on padded residues and absent neighbour slots."""
        device = self._ensure()
        c, m = _prep(coords, device), _prep(mask, device)
        B, T, k = int(c.shape[0]), int(c.shape[1]), int(self._hp["num_res_neighbours"])
        lib = _native.lib()
        need = int(lib.rnampnn_edge_raw_workspace_bytes(self._handle.ptr, B, T))
        ws = torch.empty(need + 256, dtype=torch.uint8, device=device)
        base = ws.data_ptr()
        aligned = (base + 255) // 256 * 256
        feats = torch.empty(B, T, k, 96, dtype=torch.float32, device=device)
        idx = torch.empty(B, T, k, dtype=torch.int64, device=device)
        with torch.cuda.device(device):
            _native.check(lib.rnampnn_edge_raw_features(self._handle.ptr, _ptr(c), _ptr(m), B, T, _ptr(idx), _ptr(feats),
                                                        C.c_void_p(aligned), C.c_size_t(need), _stream(device)))
        return feats[..., :90], idx

    def forward_taps(self, coords, mask, names, tap_layer: int = 0, T_norm: int = 0):
        """Forward with intermediate tensors (parity tests): ``names`` from edge_index, raw, h0, e0,
        h_layer, e_layer, h_post, raw_emb; ``tap_layer`` is the 1-based ResMPNN layer of h_/e_layer."""
        return self._run(coords, mask, want_logits=True, want_embedding=True, T_norm=T_norm,
                         taps=dict(names=tuple(names), tap_layer=tap_layer))

    # ------------------------------------------------------------------ decode
    @torch.no_grad()
    def sample(self, coords: torch.Tensor, mask: torch.Tensor, temperature: float = 0.1, n_samples: int = 8,
               seed: int = 0, T_norm: int = 0) -> torch.Tensor:
        """Draw ``n_samples`` sequences per structure, independently per position, from
        softmax(logits / temperature) -> int8 (n_samples, B, T), -1 on padding.  The reference has no
        sampler (its decode is one-shot argmax); temperature -> 0 reproduces that argmax."""
        logits = self.forward(coords, mask, T_norm=T_norm)
        return sample_from_logits(logits, mask, temperature, n_samples, seed)

    def load_xgb_readout(self, model) -> None:
        """Attach a fitted multi:softmax XGBoost model (JSON model file / dict, ``Booster.save_model``) as ``self.xgb_readout``: the
        device-side tree read-out of ``rnampnn/model/xgb.py`` (parity unpinned).  The reference unpickles its
        ``XGB-V*.pkl`` in ``on_load_checkpoint`` (rnampnn.py:232-266); pickles are never loaded here."""
        from .xgb import GBDTReadout
        self.xgb_readout = GBDTReadout.from_xgboost_json(model)

    @torch.no_grad()
    def predict_sequences(self, coords: torch.Tensor, mask: torch.Tensor) -> List[str]:
        """What ``predict`` writes (rnampnn.py:280-305): with a tree model loaded (``load_xgb_readout``) the reference's route -
        ``embedding`` (B*T, 256) -> ``xgb_readout.predict`` -> valid positions; otherwise the Readout argmax stands in for the
        unfitted classifier."""
        if getattr(self, "xgb_readout", None) is not None:
            emb = self.embedding(coords, mask)
            pred = self.xgb_readout.predict(emb).cpu()
            valid = mask.cpu() == 1
            return ["".join(REVERSE_VOCAB[int(i)] for i in row[v]) for row, v in zip(pred, valid)]
        logits = self.forward(coords, mask)
        pred, _, _ = argmax_recovery(logits, mask, None)
        pred = pred.cpu()
        return ["".join(REVERSE_VOCAB[int(i)] for i in row[row >= 0]) for row in pred]

    # ------------------------------------------------------------------ training / validation surface
    def loss_and_grad(self, sequences: torch.Tensor, coords: torch.Tensor, mask: torch.Tensor, T_norm: int = 0,
                      return_logits: bool = False, dropout: Optional[float] = None, seed: Optional[int] = None):
        """``training_step`` + ``loss.backward()`` of the reference (rnampnn.py:187-207) in one native call:
        forward, loss = cross_entropy(softmax(logits)[valid], label) and the gradient of every parameter
        (f32 HIP kernels, bit-reproducible).  ``dropout``: None = the module's hyper-parameter in train mode and 0 in eval
        mode; masks are a function of ``seed`` (None = the module's running counter, ``manual_seed``).  ``sequences`` is the
        collate's one-hot (B,T,4) or class ids (B,T).  Afterwards every ``p.grad`` is a view into ONE flat buffer
        (``self.flat_grad``, overwritten), so a data-parallel job all-reduces gradients with a single RCCL call
        (``allreduce_gradients``)."""
        device = self._ensure(for_mixed_training=self.train_precision == "bf16")
        B, T = int(coords.shape[0]), int(coords.shape[1])
        c, m = _prep(coords, device), _prep(mask, device)
        lab = sequences.argmax(dim=-1) if sequences.dim() == 3 else sequences
        lab = _prep(lab, device, torch.int32)
        lib = _native.lib()
        self._bind_flat_grad(device)
        loss = torch.zeros((), dtype=torch.float32, device=device)
        logits = torch.empty(B, T, 4, dtype=torch.float32, device=device) if return_logits else None
        p = float((self._hp["dropout"] if self.training else 0.0) if dropout is None else dropout)
        sd = self._next_seed() if seed is None else int(seed)
        with torch.cuda.device(device):
            ws, ws_bytes = self._train_ws(B, T, device)
            _native.check(lib.rnampnn_loss_and_grad(self._handle.ptr, _ptr(c), _ptr(m), _ptr(lab), B, T, int(T_norm), p,
                                                    C.c_uint64(sd & (2 ** 64 - 1)), self._train_flags(), _ptr(loss), _ptr(logits),
                                                    _ptr(self.flat_grad), ws, ws_bytes, _stream(device)))
        return (loss, logits) if return_logits else loss

    def _bind_flat_grad(self, device) -> bool:
        """(Re-)bind every ``p.grad`` to its slice of ONE flat buffer laid out like the weight arena.  Checked on every call:
        ``optimizer.zero_grad()`` (set_to_none=True, torch's and Lightning's default) drops the views, and a parameter
        whose ``.grad`` no longer aliases the buffer would silently stop being updated.  Returns True when any view had
        to be (re)made, i.e. the gradients were reset since the last backward (the next backward overwrites)."""
        lib = _native.lib()
        if getattr(self, "flat_grad", None) is None or self.flat_grad.device != device:
            self.flat_grad = torch.zeros(int(lib.rnampnn_grad_numel(self._handle.ptr)), dtype=torch.float32, device=device)
            self._grad_slices = None
        if getattr(self, "_grad_slices", None) is None:
            named = dict(self.named_parameters())
            self._grad_slices = []
            for i, (key, _) in enumerate(self._handle.weight_schema()):
                off = C.c_int64()
                _native.check(lib.rnampnn_weight_offset(self._handle.ptr, i, C.byref(off)))
                self._grad_slices.append((named[key], int(off.value)))
        base = self.flat_grad.data_ptr()
        fresh = False
        for p, off in self._grad_slices:
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                p.grad = self.flat_grad[off: off + p.numel()].view(p.shape)
                fresh = True
        return fresh

    def training_step(self, batch):
        """rnampnn.py:187-207, line for line: forward -> softmax -> boolean-mask select -> mix_loss.  The returned loss
        carries an autograd graph: ``loss.backward()`` runs the HIP backward and fills ``p.grad``."""
        sequences, coords, mask, _ = batch
        device = self._device()
        sequences, mask = sequences.to(device), mask.to(device)
        logits = self(coords, mask)
        probs = F.softmax(logits, dim=-1)
        valid = mask.bool()
        return self.mix_loss(probs[valid], sequences[valid])

    def allreduce_gradients(self) -> None:
        """Average ``flat_grad`` over the ranks of the default process group (RCCL on the GPUs): the one
        collective of data-parallel training (Lightning DDP in the reference, utils/train.py:106-117)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM)
            self.flat_grad.div_(dist.get_world_size())

    @staticmethod
    def mix_loss(valid_probs, valid_sequences):
        # rnampnn.py:151-154 - cross_entropy applied to PROBABILITIES (softmax twice), a reference quirk
        return F.cross_entropy(valid_probs, valid_sequences.argmax(dim=-1), reduction='mean')

    def configure_optimizers(self, fused: bool = False):
        """rnampnn.py:156-159: Adam(lr, weight_decay) + StepLR(15, 0.8).  ``fused=True``: the same update as ONE HIP launch
        over the flat parameter / gradient buffers (``FlatAdam``) instead of torch's per-tensor kernels."""
        if fused:
            optimizer = FlatAdam(self, lr=self._hp["lr"], weight_decay=self._hp["weight_decay"])
        else:
            optimizer = torch.optim.Adam(self.parameters(), lr=self._hp["lr"], weight_decay=self._hp["weight_decay"])
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=15, gamma=0.8)
        return [optimizer], [scheduler]

    @torch.no_grad()
    def validation_step(self, batch):
        """rnampnn.py:209-236: loss + per-RNA recovery; the argmax / per-RNA reduction that the
        reference does on the CPU (``separate``, utils/data.py:594-604) runs in one HIP kernel."""
        sequences, coords, mask, _ = batch
        device = self._device()
        sequences, mask = sequences.to(device), mask.to(device)
        logits = self.forward(coords, mask)
        probs = F.softmax(logits, dim=-1)
        valid = mask.bool()
        loss = self.mix_loss(probs[valid], sequences[valid])
        _, correct, nvalid = argmax_recovery(logits, mask, sequences.argmax(dim=-1))
        recovery_rates = (correct.float() / nvalid.float().clamp(min=1)).tolist()
        n_tot = int(nvalid.sum())
        self.val_step_outputs['val_loss'].append(loss * n_tot)
        self.val_step_outputs['correct'].append(int(correct.sum()))
        self.val_step_outputs['len'].append(n_tot)
        self.val_step_outputs['recovery_rates'] += recovery_rates
        return {'validation loss': loss, 'recovery_rates': recovery_rates}


class FlatAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (default betas / eps, L2 weight decay) for an ``RNAMPNN`` whose parameters and gradients are
    views of two flat buffers: one ``rnampnn_adam_step`` launch per step.  One parameter group, so ``StepLR`` and friends
    work on it unchanged; ``zero_grad`` zeroes the flat gradient buffer in place (the views stay bound)."""

    def __init__(self, model: "RNAMPNN", lr: float = 2e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        super().__init__(list(model.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.t = 0
        self.exp_avg = self.exp_avg_sq = None

    def zero_grad(self, set_to_none: bool = True) -> None:
        if getattr(self.model, "flat_grad", None) is not None:
            self.model.flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        m = self.model
        device = m._ensure(for_mixed_training=True)              # parameters aliased to the flat buffer; no finalize here
        if getattr(m, "flat_grad", None) is None:
            return loss                                          # no backward has run yet
        m._bind_flat_grad(device)
        if self.exp_avg is None or self.exp_avg.device != device:
            self.exp_avg = torch.zeros_like(m._flat_param)
            self.exp_avg_sq = torch.zeros_like(m._flat_param)
        g = self.param_groups[0]
        self.t += 1
        with torch.cuda.device(device):
            _native.check(_native.lib().rnampnn_adam_step(_ptr(m._flat_param), _ptr(m.flat_grad), _ptr(self.exp_avg),
                                                          _ptr(self.exp_avg_sq), m._flat_param.numel(), float(g["lr"]),
                                                          float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                                          float(g["weight_decay"]), self.t, _stream(device)))
        m._weights_touched()
        return loss


class _TrainForward(torch.autograd.Function):
    """Autograd node of the taped HIP forward.  The parameters are inputs only so that autograd schedules ``backward``;
    their gradients are written by the HIP backward straight into the module's flat buffer (``p.grad`` are views of it,
    accumulated across backward calls until the gradients are reset, as torch does), so ``None`` is returned for them."""

    @staticmethod
    def forward(ctx, model, coords, mask, T_norm, dropout, seed, *params):
        ctx.model = model
        ctx.n_params = len(params)
        return model._train_forward_native(coords, mask, T_norm, dropout, seed)

    @staticmethod
    def backward(ctx, dlogits):
        ctx.model._train_backward_native(dlogits)
        return (None,) * (6 + ctx.n_params)


def argmax_recovery(logits: torch.Tensor, mask: torch.Tensor, labels: Optional[torch.Tensor]
                    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """HIP decode kernel: (pred int8 (B,T) with -1 on padding, correct int32 (B,), valid int32 (B,))."""
    device = logits.device
    B, T = int(logits.shape[0]), int(logits.shape[1])
    lg, m = _prep(logits, device), _prep(mask, device)
    lab = None if labels is None else _prep(labels, device, torch.int32)
    pred = torch.empty(B, T, dtype=torch.int8, device=device)
    correct = torch.zeros(B, dtype=torch.int32, device=device)
    nvalid = torch.zeros(B, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        _native.check(_native.lib().rnampnn_argmax_recovery(_ptr(lg), _ptr(m), _ptr(lab), B, T, _ptr(pred),
                                                            _ptr(correct), _ptr(nvalid), _stream(device)))
    return pred, correct, nvalid


def sample_from_logits(logits: torch.Tensor, mask: torch.Tensor, temperature: float, n_samples: int, seed: int = 0
                       ) -> torch.Tensor:
    device = logits.device
    B, T = int(logits.shape[0]), int(logits.shape[1])
    lg, m = _prep(logits, device), _prep(mask, device)
    out = torch.empty(n_samples, B, T, dtype=torch.int8, device=device)
    with torch.cuda.device(device):
        _native.check(_native.lib().rnampnn_sample(_ptr(lg), _ptr(m), B, T, float(temperature), int(n_samples),
                                                   C.c_uint64(int(seed) & (2 ** 64 - 1)), _ptr(out), _stream(device)))
    return out


class CapturedSampler:
    """hipGraph-captured decode step (BASELINE config 5): ONE graph holding the whole forward and the
    sampling kernel for a fixed (B, T) bucket; replay costs one graph launch instead of ~60 kernel
    launches.  Inputs are copied into static device buffers, the RNG seed lives in device memory so
    every replay draws fresh samples.  ``torch.cuda.CUDAGraph`` is used for capture/replay plumbing
    only (on ROCm it is hipGraph); every node of the graph is a kernel of ``librnampnn_hip.so``."""

    def __init__(self, model: RNAMPNN, B: int, T: int, temperature: float = 0.1, n_samples: int = 8, T_norm: int = 0):
        device = model._ensure()
        self.model, self.B, self.T = model, B, T
        self.temperature, self.n_samples, self.T_norm = float(temperature), int(n_samples), int(T_norm)
        self.coords = torch.zeros(B, T, 7, 3, dtype=torch.float32, device=device)
        self.mask = torch.zeros(B, T, dtype=torch.float32, device=device)
        self.mask[:, 0] = 1
        self.seed = torch.zeros(1, dtype=torch.int64, device=device)
        # private workspace: the graph bakes raw pointers into it, so it must not be the module's shared scratch (which a later,
        # larger eager call reallocates) and it lives exactly as long as this object
        self._ws = model.new_workspace(B, T)
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):                       # warm-up: one-time function attributes, workspace growth
            for _ in range(2):
                self._step()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.logits, self.samples = self._step()

    def _step(self):
        logits = self.model._run(self.coords, self.mask, T_norm=self.T_norm, workspace=self._ws)["logits"]
        out = torch.empty(self.n_samples, self.B, self.T, dtype=torch.int8, device=logits.device)
        with torch.cuda.device(logits.device):
            _native.check(_native.lib().rnampnn_sample_dev_seed(
                _ptr(logits), _ptr(self.mask), self.B, self.T, self.temperature, self.n_samples, _ptr(self.seed),
                _ptr(out), _stream(logits.device)))
        return logits, out

    @torch.no_grad()
    def __call__(self, coords: torch.Tensor, mask: torch.Tensor, seed: int = 0):
        """-> (logits (B,T,4), samples int8 (n_samples,B,T)) - static tensors, overwritten by the next call.
        Weights changed since the capture (optimizer step, load_state_dict) are re-uploaded first: the graph reads the
        library's weight arena, whose address does not change."""
        self.model._ensure()
        self.coords.copy_(coords, non_blocking=True)
        self.mask.copy_(mask, non_blocking=True)
        self.seed.fill_(int(seed) & (2 ** 63 - 1))
        self.graph.replay()
        return self.logits, self.samples
