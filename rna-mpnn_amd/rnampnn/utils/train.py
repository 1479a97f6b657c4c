"""The training loop of the MI355X path - the role of the reference's ``rnampnn/utils/train.py:91-118`` (``get_trainer``: Lightning
``Trainer(precision='bf16-mixed', strategy='ddp...')``) without Lightning: one process per GPU, ONE gradient exchange per step
(chunked over a side stream under the tail of the backward), fused Adam, and NO host synchronisation inside an epoch - the
loss is accumulated on the device, nucleotide counts come from the host-side lengths the loader already has, the input
batches arrive through ``PaddedLoader`` (pinned memory, copies on a side stream).

Batch statistics semantics: every rank normalises with ITS OWN padded batch length (``T_norm = 0``), which is what the
reference's DDP ranks do (each rank's GraphNormalization sees its own collated batch, functional.py:33-39);
``global_t_norm=True`` instead passes the MAX over the ranks (one all-reduce of the per-step lengths per EPOCH, computed
on the host from the batch plan) - the setting under which an N-rank step equals the 1-rank step on the union batch.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Sequence

import torch

from . import shard
from .data import PaddedLoader, bucket_batches


def plan_epoch(lengths: Sequence[int], rank: int, world: int, batch_size: int, max_rows: int, seed: int):
    """-> (batches of GLOBAL item indices for this rank, per-step padded lengths of this rank).  The batch plan is a pure function
    of (lengths, world, seed): every rank computes the whole plan and takes its column - no communication.  Length-bucketed
    batches are dealt to the ranks round-robin in length order, so step s of every rank holds RNAs of similar length (balanced
    work per step) and all ranks take the same number of steps.

    What the reference's loaders do and this plan keeps (``DataLoader(shuffle=True)`` + Lightning's ``DistributedSampler``, which
    uses ``drop_last=False``): EVERY sample is seen in every epoch, and the composition of the batches changes from epoch to
    epoch.  (i) The length order that the buckets are cut from is jittered by the seed - an RNA's sort key is its length times a
    seeded factor in [0.95, 1.05] - so neighbours in length trade places between batches; (ii) a last round that the batches do
    not fill is padded by REPEATING batches of that round (the sampler pads by wrapping around), never dropped: with a fixed
    length-sorted order the dropped tail would be the same longest RNAs in every epoch."""
    from . import synth
    import numpy as np
    n_items = len(lengths)
    jit = synth.uniform01(synth._fnv1a64(f"epoch_jitter/{seed}"), np.arange(n_items, dtype=np.uint64))
    order = sorted(range(n_items), key=lambda i: (int(lengths[i]) * (0.95 + 0.1 * float(jit[i])), i))
    batches: List[List[int]] = []
    cur: List[int] = []
    cur_max = 0
    for i in order:
        n = int(lengths[i])
        if cur and (len(cur) >= batch_size or (len(cur) + 1) * max(cur_max, n) > max_rows):
            batches.append(cur)
            cur, cur_max = [], 0
        cur.append(i)
        cur_max = max(cur_max, n)
    if cur:
        batches.append(cur)
    if not batches:
        raise ValueError("no items")
    rem = len(batches) % world
    if rem:                 # pad the unfilled last round by wrapping around its own batches (similar lengths: balanced steps)
        tail = batches[-rem:]
        batches = batches + [list(tail[q % rem]) for q in range(world - rem)]
    rounds = len(batches) // world
    u = synth.uniform01(synth._fnv1a64(f"epoch_order/{seed}"), np.arange(rounds, dtype=np.uint64))
    perm = np.argsort(u, kind="stable")
    mine = [batches[int(r) * world + rank] for r in perm]
    t_glob = [max(max(int(lengths[i]) for i in batches[int(r) * world + q]) for q in range(world)) for r in perm]
    return mine, t_glob


class Trainer:
    """``fit``-style loop around ``RNAMPNN.loss_and_grad`` + ``allreduce_gradients`` + the optimiser."""

    def __init__(self, model, optimizer, scheduler=None, world: int = 1, rank: int = 0, overlap_allreduce: bool = True,
                 global_t_norm: bool = False, seed: int = 0):
        self.model, self.opt, self.sched = model, optimizer, scheduler
        self.world, self.rank, self.seed = int(world), int(rank), int(seed)
        self.global_t_norm = bool(global_t_norm)
        self.device = model._device()
        if self.world > 1 and overlap_allreduce:
            model.enable_allreduce_overlap(True)
        self._loss = torch.zeros((), dtype=torch.float32, device=self.device)

    def step(self, labels, coords, mask, T_norm: int = 0, seed: Optional[int] = None, timing=None):
        """forward + backward (one native call), gradient exchange, optimiser step; returns the device loss (no sync)."""
        loss = self.model.loss_and_grad(labels, coords, mask, T_norm=T_norm, seed=seed)
        self.model.allreduce_gradients(timing=timing)
        self.opt.step()
        return loss

    def run_epoch(self, items, lengths: Sequence[int], epoch: int, batch_size: int, max_rows: int) -> Dict:
        """One pass over ``items`` ((id, coords, labels) or (coords, labels)); -> dict(train_loss, steps, nt, seconds, nt_per_s)
        for THIS rank's share.  The only host synchronisation is the one at the end of the epoch."""
        self.model.train()
        mine, t_glob = plan_epoch(lengths, self.rank, self.world, batch_size, max_rows, self.seed + epoch)
        loader = PaddedLoader(items, mine, device=self.device)
        # (the jittered plan has other batch shapes in every epoch: size the tape for this epoch's largest step before the clock starts)
        self.model.reserve_training((len(b), max(int(lengths[i]) for i in b)) for b in mine)
        self._loss.zero_()
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        nt = 0
        for it, (y, c, m, lens, _) in enumerate(loader):
            tn = t_glob[it] if (self.global_t_norm and self.world > 1) else 0
            loss = self.step(y, c, m, T_norm=tn, seed=((self.seed << 20) + epoch * 100003 + it) * max(self.world, 1) + self.rank)
            self._loss += loss
            nt += sum(lens)
        if self.sched is not None:
            self.sched.step()
        torch.cuda.synchronize(self.device)
        dt = time.perf_counter() - t0
        steps = len(mine)
        return dict(train_loss=float(self._loss) / max(steps, 1), steps=steps, nt=nt, seconds=dt, nt_per_s=nt / dt)

    @torch.no_grad()
    def validate(self, items, lengths: Sequence[int], batch_size: int, max_rows: int):
        """Micro / macro recovery (utils/train.py:15-26 of the reference) over all ranks; every rank scores a strided share."""
        from ..model.rnampnn import argmax_recovery
        self.model.eval()
        batches = bucket_batches(lengths, batch_size, max_rows, seed=0)[self.rank::self.world]
        correct_all, valid_all = [], []
        for y, c, m, _, _ in PaddedLoader(items, batches, device=self.device):
            logits = self.model(c, m)
            _, correct, nvalid = argmax_recovery(logits, m, y)
            correct_all.append(correct); valid_all.append(nvalid)
        if correct_all:
            c, v = torch.cat(correct_all), torch.cat(valid_all)
        else:
            c = v = torch.zeros(0, dtype=torch.int32, device=self.device)
        return shard.reduce_recovery(c, v)
