"""Host-side input contract of the hot path: the batch layout the reference's collate produces and
``separate`` (``rnampnn/utils/data.py:110-142, 594-604``).  Dataset / PDB parsing / plotting of that
file are host I/O outside the hot path and are not mirrored."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple, Union

import torch
from torch.nn.utils.rnn import pad_sequence

Item = Dict[str, Union[str, torch.Tensor]]


def featurize(batch: Sequence[Item]) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, List[str]]:
    """Collate {'sequence' (L,4) one-hot, 'coordinates' (L,7,3), 'id'} items to the batch max length ->
    (sequences (B,T,4), coords (B,T,7,3), mask (B,T), ids); zero padding, mask = prefix of ones
    (the layout every kernel of this package assumes)."""
    seqs = pad_sequence([it['sequence'].to(torch.float32) for it in batch], batch_first=True)
    xyz = pad_sequence([it['coordinates'].to(torch.float32) for it in batch], batch_first=True)
    lengths = torch.tensor([it['sequence'].shape[0] for it in batch])
    mask = (torch.arange(seqs.shape[1]).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.float32)
    return seqs, xyz, mask, [it['id'] for it in batch]


def separate(concat: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """Inverse of boolean-mask selection: a concatenation of per-RNA vectors -> zero-padded (B, max_len)."""
    pieces = torch.split(concat, [int(n) for n in lengths.tolist()])
    return pad_sequence(list(pieces), batch_first=True) if pieces else concat.new_zeros((0, 0))


def check_prefix_mask(mask: torch.Tensor) -> None:
    """The kernels assume the collate's prefix masks (ones then zeros per row)."""
    m = mask.detach().cpu() != 0
    if bool((m[:, 1:] & ~m[:, :-1]).any()):
        raise ValueError("mask rows must be prefixes of ones (as produced by featurize)")


def pack_batch(coords_list: Sequence[torch.Tensor], pin: bool = True):
    """Var-len collate: a list of (L_i,7,3) tensors -> (coords_packed (N,7,3) f32,
    cu_seqlens (B+1) int32, max_len), in pinned host memory so ``.to(device, non_blocking=True)`` overlaps
    the copy with compute.  No padding is materialised or copied (the padded collate moves B*T rows)."""
    lengths = torch.tensor([int(c.shape[0]) for c in coords_list], dtype=torch.int32)
    cu = torch.zeros(len(coords_list) + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lengths, 0)
    packed = torch.empty((int(cu[-1]), 7, 3), dtype=torch.float32, pin_memory=pin and torch.cuda.is_available())
    torch.cat([c.to(torch.float32) for c in coords_list], dim=0, out=packed)
    if pin and torch.cuda.is_available():
        cu = cu.pin_memory()
    return packed, cu, int(lengths.max()) if len(coords_list) else 0


# --------------------------------------------------------------------------------------------------------------
# Host input pipeline: dataset directory -> length buckets -> packed pinned batches
# -> asynchronous host-to-device copies.  Mirrors the ROLE of the reference's RNADataset / RNADataModule
# (rnampnn/utils/data.py:155-258, 397-539) for the hot path; PDB parsing, plotting and CSV tools are not mirrored.
import glob as _glob
import os as _os
import queue as _queue
import threading as _threading

import numpy as _np

from ..config.glob import NUM_RES_TYPES, VOCAB
from . import synth as _synth


def fill_nan_deterministic(coords: "_np.ndarray", rna_id: str) -> "_np.ndarray":
    """Deterministic replacement for the reference's NaN fill (``RNADataset.__fill_nan_with_mean``,
    rnampnn/utils/data.py:190-222).  Same geometry, atom by atom and in place like the reference: a missing atom
    a < 6 goes 1.5 A from the first atom of its residue that is present at that moment, a missing atom 6 goes
    ``NUM_RES_TYPES`` (= 4) A from atom 5; whatever is still NaN becomes 0.  The reference draws the direction from the
    UNSEEDED global ``np.random`` stream (order- and process-dependent, so no two runs see the same structure); here it
    is a unit vector from the counter RNG of ``rnampnn.utils.synth`` keyed by (rna id, residue, atom): the same file
    gives the same coordinates on every box, rank and epoch."""
    c = _np.array(coords, dtype=_np.float32, copy=True)
    stream = _synth._fnv1a64("nan_fill/" + rna_id)
    for a in range(c.shape[1]):
        missing = _np.where(_np.isnan(c[:, a, :]).any(axis=1))[0]
        for r in missing:
            g = _synth.normal01(stream, _np.arange(3, dtype=_np.uint64) + _np.uint64(3 * (r * c.shape[1] + a)))
            g = g / max(float(_np.linalg.norm(g)), 1e-12)
            if a < 6:
                present = _np.where(~_np.isnan(c[r]).any(axis=1))[0]
                if len(present) > 0:
                    c[r, a] = c[r, present[0]] + 1.5 * g
            elif not _np.isnan(c[r, 5]).any():
                c[r, a] = c[r, 5] + float(NUM_RES_TYPES) * g
    c[_np.isnan(c)] = 0.0
    return c


def read_fasta(path: str) -> str:
    return "".join(line.strip() for line in open(path) if not line.startswith(">"))


def load_rna_dir(path: str, max_len: int = 1 << 30, nan_policy: str = "skip"):
    """The reference's data layout (``RNADataset.from_path``, utils/data.py:155-187): ``coords/<id>.npy`` (L,7,3) +
    ``seqs/<id>.fasta`` -> list of (id, coords f32 (L,7,3), labels int64 (L,)) in id order.  ``nan_policy``: "skip"
    drops structures with missing atoms, "fill" applies ``fill_nan_deterministic``.  Entries whose sequence length
    differs from the coordinate count, or with letters outside AUCG, are dropped."""
    if nan_policy not in ("skip", "fill"):
        raise ValueError("nan_policy must be 'skip' or 'fill'")
    items = []
    for f in sorted(_glob.glob(_os.path.join(path, "coords", "*.npy"))):
        rid = _os.path.splitext(_os.path.basename(f))[0]
        fa = _os.path.join(path, "seqs", rid + ".fasta")
        if not _os.path.exists(fa):
            continue
        c = _np.load(f, allow_pickle=False).astype(_np.float32)
        seq = read_fasta(fa)
        if c.ndim != 3 or c.shape[1:] != (7, 3) or c.shape[0] != len(seq) or c.shape[0] > max_len or c.shape[0] == 0:
            continue
        if any(ch not in VOCAB for ch in seq):
            continue
        if _np.isnan(c).any():
            if nan_policy == "skip":
                continue
            c = fill_nan_deterministic(c, rid)
        items.append((rid, c, _np.array([VOCAB[ch] for ch in seq], dtype=_np.int64)))
    return items


def bucket_batches(lengths: Sequence[int], batch_size: int, max_rows: int = 1 << 30, seed: int = 0) -> List[List[int]]:
    """Length-bucketed batches: indices sorted by length (ties by index) are cut greedily into batches of at most
    ``batch_size`` RNAs and at most ``max_rows`` padded rows (B * longest); a single RNA longer than ``max_rows`` gets a
    batch of its own.  The ORDER of the batches is shuffled by ``seed`` (counter RNG: identical on every rank)."""
    order = sorted(range(len(lengths)), key=lambda i: (int(lengths[i]), i))
    batches: List[List[int]] = []
    cur: List[int] = []
    for i in order:
        n = int(lengths[i])            # ascending, so n is the longest of cur + [i]
        if cur and (len(cur) >= batch_size or (len(cur) + 1) * n > max_rows):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    u = _synth.uniform01(_synth._fnv1a64(f"bucket_order/{seed}"), _np.arange(len(batches), dtype=_np.uint64))
    return [batches[i] for i in _np.argsort(u, kind="stable")]


class PackedLoader:
    """Asynchronous var-len loader: a background thread collates the next batches with ``pack_batch`` into pinned host
    memory; ``__iter__`` yields (coords_packed, cu_seqlens, max_len, indices) ALREADY ON THE DEVICE - the host-to-device
    copy of batch i+1 is issued on a side stream while the kernels of batch i run, and the consumer's stream waits on
    the copy's event only (no host synchronisation).  ``items`` is a list of coords arrays or of (id, coords, labels)
    tuples as ``load_rna_dir`` returns; ``batches`` a list of index lists (``bucket_batches``)."""

    def __init__(self, items, batches: Sequence[Sequence[int]], device=None, prefetch: int = 2):
        self.items, self.batches, self.prefetch = items, [list(b) for b in batches], max(1, int(prefetch))
        self.device = torch.device(device) if device is not None else None

    def _coords(self, i):
        it = self.items[i]
        c = it[1] if isinstance(it, tuple) else it
        return torch.as_tensor(c, dtype=torch.float32)

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        q: "_queue.Queue" = _queue.Queue(maxsize=self.prefetch)
        use_gpu = self.device is not None and self.device.type == "cuda"
        side = torch.cuda.Stream(self.device) if use_gpu else None

        def worker():
            try:
                for b in self.batches:
                    packed, cu, max_len = pack_batch([self._coords(i) for i in b], pin=use_gpu)
                    if use_gpu:
                        with torch.cuda.stream(side):
                            dp = packed.to(self.device, non_blocking=True)
                            dc = cu.to(self.device, non_blocking=True)
                            ev = torch.cuda.Event()
                            ev.record(side)
                        q.put((dp, dc, max_len, b, ev, (packed, cu)))      # keep the pinned source alive until consumed
                    else:
                        q.put((packed, cu, max_len, b, None, None))
                q.put(None)
            except BaseException as exc:       # surface loader errors in the consumer
                q.put(exc)

        th = _threading.Thread(target=worker, daemon=True)
        th.start()
        while True:
            got = q.get()
            if got is None:
                break
            if isinstance(got, BaseException):
                raise got
            dp, dc, max_len, b, ev, _keep = got
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                dp.record_stream(torch.cuda.current_stream(self.device))
                dc.record_stream(torch.cuda.current_stream(self.device))
            yield dp, dc, max_len, b
        th.join()


def pad_batch(items: Sequence, pin: bool = True):
    """The reference collate's layout (``featurize``, utils/data.py:110-142) for (coords (L,7,3), labels (L,)) pairs, built in
    pinned host memory: (labels (B,T) int32, coords (B,T,7,3) f32, mask (B,T) f32, lengths (B,) list).  Labels are class ids
    (the one-hot of the reference is an argmax away and 4x the bytes)."""
    lens = [int(c.shape[0]) for c, _ in items]
    B, T = len(items), max(lens)
    pin = pin and torch.cuda.is_available()
    coords = torch.zeros((B, T, 7, 3), dtype=torch.float32, pin_memory=pin)
    mask = torch.zeros((B, T), dtype=torch.float32, pin_memory=pin)
    labels = torch.zeros((B, T), dtype=torch.int32, pin_memory=pin)
    for i, (c, y) in enumerate(items):
        n = lens[i]
        coords[i, :n] = torch.as_tensor(c, dtype=torch.float32)
        labels[i, :n] = torch.as_tensor(y).to(torch.int32)
        mask[i, :n] = 1.0
    return labels, coords, mask, lens


class PaddedLoader:
    """``PackedLoader``'s sibling for the TRAINING entry points, which take the collate's padded layout: a background thread
    builds the next batches with ``pad_batch`` in pinned memory and issues their host-to-device copies on a side stream;
    ``__iter__`` yields (labels, coords, mask, lengths, indices) with the three tensors already on the device and the
    consumer's stream waiting on the copy's event only.  ``lengths`` stays on the host: the trainer counts nucleotides without
    a device round trip.  ``items``: (id, coords, labels) tuples (``load_rna_dir``) or (coords, labels) pairs."""

    def __init__(self, items, batches: Sequence[Sequence[int]], device=None, prefetch: int = 2):
        self.items, self.batches, self.prefetch = items, [list(b) for b in batches], max(1, int(prefetch))
        self.device = torch.device(device) if device is not None else None

    def _pair(self, i):
        it = self.items[i]
        return (it[1], it[2]) if len(it) == 3 else (it[0], it[1])

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        q: "_queue.Queue" = _queue.Queue(maxsize=self.prefetch)
        use_gpu = self.device is not None and self.device.type == "cuda"
        side = torch.cuda.Stream(self.device) if use_gpu else None

        def worker():
            try:
                for b in self.batches:
                    y, c, m, lens = pad_batch([self._pair(i) for i in b], pin=use_gpu)
                    if use_gpu:
                        with torch.cuda.stream(side):
                            dy, dc, dm = (t.to(self.device, non_blocking=True) for t in (y, c, m))
                            ev = torch.cuda.Event()
                            ev.record(side)
                        q.put((dy, dc, dm, lens, b, ev, (y, c, m)))        # keep the pinned sources alive until consumed
                    else:
                        q.put((y, c, m, lens, b, None, None))
                q.put(None)
            except BaseException as exc:
                q.put(exc)

        th = _threading.Thread(target=worker, daemon=True)
        th.start()
        while True:
            got = q.get()
            if got is None:
                break
            if isinstance(got, BaseException):
                raise got
            dy, dc, dm, lens, b, ev, _keep = got
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in (dy, dc, dm):
                    t.record_stream(cur)
            yield dy, dc, dm, lens, b
        th.join()
