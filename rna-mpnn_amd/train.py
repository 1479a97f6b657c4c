#!/usr/bin/env python3
"""Drop-in training driver for the MI355X RNA-MPNN path (the role of the reference's ``train.py:1-59`` +
``rnampnn/utils/train.py:91-118`` without Lightning): Adam(lr=2e-3, wd=2e-4) + StepLR(15, 0.8)
(``rnampnn.py:156-159``), loss = cross_entropy(softmax(logits)) (``rnampnn.py:151-154``), macro / micro
recovery on a validation split (``utils/train.py:15-26``).  Gradients come from the HIP backward
(``rnampnn_loss_and_grad``); data-parallel runs are one process per GPU with ONE flat RCCL all-reduce
per step (``RNAMPNN.allreduce_gradients``):

    python rna-mpnn_amd/train.py --data /path/to/data --epochs 2                    # coords/*.npy + seqs/*.fasta
    python rna-mpnn_amd/train.py --synthetic 512 --epochs 3                         # seeded synthetic RNAs
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 rna-mpnn_amd/train.py --synthetic 4096
"""
from __future__ import annotations

import argparse
import glob
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from rnampnn.config.glob import VOCAB  # noqa: E402
from rnampnn.model.rnampnn import RNAMPNN, argmax_recovery  # noqa: E402
from rnampnn.utils import shard, synth  # noqa: E402


def load_dir(path, max_len):
    """coords/<id>.npy (L,7,3) + seqs/<id>.fasta, the reference's data layout (utils/data.py:155-187).
    RNAs with NaN coordinates are skipped (the reference fills them with unseeded random vectors)."""
    items = []
    for f in sorted(glob.glob(os.path.join(path, "coords", "*.npy"))):
        rid = os.path.splitext(os.path.basename(f))[0]
        fa = os.path.join(path, "seqs", rid + ".fasta")
        if not os.path.exists(fa):
            continue
        c = np.load(f).astype(np.float32)
        seq = "".join(l.strip() for l in open(fa) if not l.startswith(">"))
        if c.shape[0] != len(seq) or c.shape[0] > max_len or np.isnan(c).any() or any(ch not in VOCAB for ch in seq):
            continue
        items.append((c, np.array([VOCAB[ch] for ch in seq], dtype=np.int64)))
    return items


def collate(items):
    T = max(c.shape[0] for c, _ in items)
    coords = np.zeros((len(items), T, 7, 3), np.float32)
    mask = np.zeros((len(items), T), np.float32)
    labels = np.zeros((len(items), T), np.int64)
    for i, (c, y) in enumerate(items):
        n = c.shape[0]
        coords[i, :n], mask[i, :n], labels[i, :n] = c, 1, y
    return torch.from_numpy(labels), torch.from_numpy(coords), torch.from_numpy(mask)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--max-len", type=int, default=200)
    ap.add_argument("--neighbours", type=int, default=30)
    ap.add_argument("--layers", type=int, default=10)
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.data:
        items = load_dir(args.data, args.max_len)
    else:
        lens = synth.synth_lengths(args.synthetic or 256, 30, min(140, args.max_len), seed=1)
        items = [(synth.synth_rna(int(n), i, seed=1), synth.synth_labels(int(n), i, seed=1)) for i, n in enumerate(lens)]
    n_val = max(1, len(items) // 20)
    val, train = items[:n_val], items[n_val:]
    model = RNAMPNN(num_res_neighbours=args.neighbours, num_res_mpnn_layers=args.layers, padding_len=args.max_len,
                    precision="f32").to(dev)
    if world > 1:                                   # identical initial weights on every rank
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    (opt,), (sched,) = model.configure_optimizers()
    mine = shard.balanced_shards([c.shape[0] for c, _ in train], world)[rank]
    steps = min(len(s) for s in shard.balanced_shards([c.shape[0] for c, _ in train], world)) // args.batch_size
    for epoch in range(args.epochs):
        order = np.random.RandomState(epoch).permutation(len(mine))
        t0, nt, tot = time.perf_counter(), 0, 0.0
        for it in range(steps):
            batch = [train[mine[j]] for j in order[it * args.batch_size:(it + 1) * args.batch_size]]
            y, c, m = collate(batch)
            T_norm = shard.global_max_len(int(m.shape[1]), dev)
            loss = model.loss_and_grad(y, c, m, T_norm=T_norm)
            model.allreduce_gradients()
            opt.step()
            nt += int(m.sum()); tot += float(loss)
        sched.step()
        y, c, m = collate(val)
        with torch.no_grad():
            logits = model(c, m)
            _, correct, nvalid = argmax_recovery(logits, m.to(dev), y.to(dev))
        micro, macro = shard.reduce_recovery(correct, nvalid) if world == 1 else (float(correct.sum()) / float(nvalid.sum()), float((correct.float() / nvalid.float()).mean()))
        torch.cuda.synchronize(dev)
        if rank == 0:
            dt = time.perf_counter() - t0
            print(f"epoch {epoch}: train_loss {tot / max(steps, 1):.4f}  val_recovery micro {micro:.4f} macro {macro:.4f}  "
                  f"{nt * world / dt:.0f} nt/s ({steps} steps x {args.batch_size} RNAs x {world} ranks)", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
