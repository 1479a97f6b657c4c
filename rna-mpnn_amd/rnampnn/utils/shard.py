"""Data-parallel sharding of the forward path.

RNAs are independent units: the forward needs NO data-path collective.  Whole RNAs go to ranks
(never split); the only cross-rank quantities are (a) the global batch max_len, which the
padding-dependent GraphNormalization needs as ``T_norm`` to make an N-rank run reproduce the
1-rank result (reference functional.py:33-38), and (b) the throughput / recovery counters.
Works with any ``torch.distributed`` backend (RCCL = "nccl" on the GPUs, "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def strided_shard(n_items: int, rank: int, world: int) -> List[int]:
    """``DistributedSampler`` semantics without shuffling/padding: rank r takes items r::world."""
    return list(range(rank, n_items, world))


def balanced_shards(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Length-balanced assignment of whole RNAs: longest-first greedy onto the lightest rank
    (load = nucleotides, ties -> lower rank), each rank's list sorted by index.  Deterministic."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (loads[j], j))
        out[r].append(i)
        loads[r] += int(lengths[i])
    return [sorted(x) for x in out]


def _dev(device):
    return device if device is not None else torch.device("cpu")


def global_max_len(local_max_len: int, device=None) -> int:
    """MAX all-reduce of the per-rank max_len -> the T_norm every rank passes to its kernels."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(local_max_len)
    t = torch.tensor([int(local_max_len)], dtype=torch.int64, device=_dev(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def reduce_job(elapsed_s: float, units: float, device=None) -> Tuple[float, float]:
    """-> (max-over-ranks elapsed, sum-over-ranks units): the two numbers of a whole-job rate."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), float(units)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=_dev(device))
    u = torch.tensor([float(units)], dtype=torch.float64, device=_dev(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def reduce_recovery(correct: torch.Tensor, valid: torch.Tensor) -> Tuple[float, float]:
    """Micro / macro recovery over all ranks from per-RNA (correct, valid) counts
    (reference utils/train.py:18-21 with ``sync_dist=True`` semantics)."""
    c = correct.to(torch.float64)
    v = valid.to(torch.float64)
    stats = torch.stack([c.sum(), v.sum(), (c / v.clamp(min=1)).sum(), (v > 0).to(torch.float64).sum()])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    micro = float(stats[0] / stats[1].clamp(min=1))
    macro = float(stats[2] / stats[3].clamp(min=1))
    return micro, macro


def allreduce_mean_chunks(flat: torch.Tensor, chunks: Sequence[Tuple[int, int]]) -> None:
    """Average ``flat`` over the ranks in the given [begin, end) ranges, one collective per range (the host-side form of
    ``RNAMPNN.allreduce_gradients`` with overlap enabled, without the stream plumbing).  The ranges must tile the buffer."""
    covered = sorted(chunks)
    if covered[0][0] != 0 or covered[-1][1] != flat.numel() or any(a[1] != b[0] for a, b in zip(covered, covered[1:])):
        raise ValueError(f"gradient chunks {list(chunks)} do not tile a buffer of {flat.numel()} elements")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    inv = 1.0 / dist.get_world_size()
    for b, e in chunks:
        if e > b:
            part = flat[b:e]
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
            part.mul_(inv)
