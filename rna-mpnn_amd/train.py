#!/usr/bin/env python3
"""Drop-in training driver for the MI355X RNA-MPNN path (the role of the reference's ``train.py:1-59`` +
``rnampnn/utils/train.py:91-118`` without Lightning): Adam(lr=2e-3, wd=2e-4) + StepLR(15, 0.8)
(``rnampnn.py:156-159``), loss = cross_entropy(softmax(logits)) (``rnampnn.py:151-154``), macro / micro
recovery on a validation split (``utils/train.py:15-26``).  Gradients come from the HIP backward; data-parallel
runs are one process per GPU with ONE flat RCCL all-reduce per step (``RNAMPNN.allreduce_gradients``):

    python rna-mpnn_amd/train.py --data /path/to/data --epochs 2                    # coords/*.npy + seqs/*.fasta
    python rna-mpnn_amd/train.py --synthetic 512 --epochs 3                         # seeded synthetic RNAs
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 rna-mpnn_amd/train.py --synthetic 4096

Batches are length-bucketed (``rnampnn.utils.data.bucket_batches``): the reference's collate pads every RNA of a batch
to the longest one, so mixing a 2,436-nt ribosomal RNA with 20-nt hairpins would spend > 99 % of the rows on padding.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from rnampnn.model.rnampnn import RNAMPNN, argmax_recovery  # noqa: E402
from rnampnn.utils import shard, synth  # noqa: E402
from rnampnn.utils.data import bucket_batches, load_rna_dir  # noqa: E402


def collate(items):
    T = max(c.shape[0] for c, _ in items)
    coords = np.zeros((len(items), T, 7, 3), np.float32)
    mask = np.zeros((len(items), T), np.float32)
    labels = np.zeros((len(items), T), np.int64)
    for i, (c, y) in enumerate(items):
        n = c.shape[0]
        coords[i, :n], mask[i, :n], labels[i, :n] = c, 1, y
    return torch.from_numpy(labels), torch.from_numpy(coords), torch.from_numpy(mask)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=16)
    ap.add_argument("--max-nt", type=int, default=8192, help="upper bound on padded rows (B*T) of one batch")
    ap.add_argument("--max-len", type=int, default=200, help="longest RNA kept (the reference trained with 100, train.py:57)")
    ap.add_argument("--neighbours", type=int, default=30)
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--dropout", type=float, default=None, help="default: the reference's 0.4 (rnampnn.py:47)")
    ap.add_argument("--nan-policy", default="skip", choices=["skip", "fill"])
    ap.add_argument("--train-precision", default="bf16", choices=["bf16", "f32"],
                    help="bf16 = the reference's bf16-mixed trainer setting (utils/train.py:109); f32 = exact")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the fused flat-buffer Adam")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args(argv)


def run(args, log=print):
    """-> dict(epochs=[dict(train_loss, val_micro, val_macro, nt_per_s, steps)], n_train, n_val)."""
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.data:
        items = [(c, y) for _, c, y in load_rna_dir(args.data, max_len=args.max_len, nan_policy=args.nan_policy)]
    else:
        lens = synth.synth_lengths(args.synthetic or 256, 30, min(140, args.max_len), seed=1)
        items = [(synth.synth_rna(int(n), i, seed=1), synth.synth_labels(int(n), i, seed=1)) for i, n in enumerate(lens)]
    n_val = max(1, len(items) // 20)
    order0 = np.random.RandomState(args.seed).permutation(len(items))          # id-order-independent split
    val = [items[i] for i in order0[:n_val]]
    train = [items[i] for i in order0[n_val:]]
    hp = dict(num_res_neighbours=args.neighbours, num_res_mpnn_layers=args.layers, padding_len=max(args.max_len, 1))
    if args.dropout is not None:
        hp["dropout"] = args.dropout
    model = RNAMPNN(**hp).to(dev)
    model.train_precision = args.train_precision
    if world > 1:                                   # identical initial weights on every rank
        for p in model.parameters():
            dist.broadcast(p.data, 0)
        model._weights_touched()
    (opt,), (sched,) = model.configure_optimizers(fused=not args.torch_adam)
    shards = shard.balanced_shards([c.shape[0] for c, _ in train], world)
    mine = shards[rank]
    out = dict(epochs=[], n_train=len(train), n_val=len(val))
    for epoch in range(args.epochs):
        model.train()
        batches = bucket_batches([train[j][0].shape[0] for j in mine], args.batch_size, args.max_nt, seed=args.seed + epoch)
        steps = len(batches)
        if world > 1:                               # every rank takes the same number of optimiser steps
            t = torch.tensor([steps], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            steps = int(t.item())
        t0, nt, tot = time.perf_counter(), 0, 0.0
        for it in range(steps):
            y, c, m = collate([train[mine[j]] for j in batches[it]])
            T_norm = shard.global_max_len(int(m.shape[1]), dev)
            opt.zero_grad()
            loss = model.loss_and_grad(y, c, m, T_norm=T_norm, seed=(args.seed << 20) + epoch * 100003 + it)
            model.allreduce_gradients()
            opt.step()
            nt += int(m.sum()); tot += float(loss)
        sched.step()
        model.eval()
        correct_all, valid_all = [], []
        with torch.no_grad():
            for b in bucket_batches([c.shape[0] for c, _ in val], args.batch_size, args.max_nt, seed=0):
                y, c, m = collate([val[j] for j in b])
                logits = model(c, m)
                _, correct, nvalid = argmax_recovery(logits, m.to(dev), y.to(dev))
                correct_all.append(correct); valid_all.append(nvalid)
        micro, macro = shard.reduce_recovery(torch.cat(correct_all), torch.cat(valid_all))
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        rec = dict(train_loss=tot / max(steps, 1), val_micro=micro, val_macro=macro, nt_per_s=nt * world / dt, steps=steps)
        out["epochs"].append(rec)
        if rank == 0:
            log(f"epoch {epoch}: train_loss {rec['train_loss']:.4f}  val_recovery micro {micro:.4f} macro {macro:.4f}  "
                f"{rec['nt_per_s']:.0f} nt/s ({steps} steps x <= {args.batch_size} RNAs x {world} ranks)")
    out["model"] = model
    return out


def main():
    run(parse())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
