#!/usr/bin/env python3
"""Drop-in training driver for the MI355X RNA-MPNN path (the role of the reference's ``train.py:1-59`` +
``rnampnn/utils/train.py:91-118`` without Lightning): Adam(lr=2e-3, wd=2e-4) + StepLR(15, 0.8)
(``rnampnn.py:156-159``), loss = cross_entropy(softmax(logits)) (``rnampnn.py:151-154``), macro / micro
recovery on a validation split (``utils/train.py:15-26``).  Gradients come from the HIP backward; data-parallel
runs are one process per GPU with ONE flat RCCL all-reduce per step (``RNAMPNN.allreduce_gradients``):

    python rna-mpnn_amd/train.py --data /path/to/data --epochs 2                    # coords/*.npy + seqs/*.fasta
    python rna-mpnn_amd/train.py --synthetic 512 --epochs 3                         # seeded synthetic RNAs
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 rna-mpnn_amd/train.py --synthetic 4096

Batches are length-bucketed (``rnampnn.utils.train.plan_epoch``): the reference's collate pads every RNA of a batch
to the longest one, so mixing a 2,436-nt ribosomal RNA with 20-nt hairpins would spend > 99 % of the rows on padding.
The loop itself is ``rnampnn.utils.train.Trainer``: inputs through ``PaddedLoader`` (pinned memory, copies on a side stream),
loss accumulated on the device, no host synchronisation inside an epoch.
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

from rnampnn.model.rnampnn import RNAMPNN  # noqa: E402
from rnampnn.utils import synth  # noqa: E402
from rnampnn.utils.data import load_rna_dir  # noqa: E402
from rnampnn.utils.train import Trainer  # noqa: E402


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--lengths-file", default=None,
                    help="synthetic RNAs of exactly these lengths (.npy int array; tests/data/c3_train_lengths.npy = the 2,083 ids of the "
                         "reference's data/train_data.csv)")
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch-size", type=int, default=512, help="upper bound on RNAs per step per rank")
    ap.add_argument("--max-nt", type=int, default=32768, help="upper bound on padded rows (B*T) of one step per rank")
    ap.add_argument("--max-len", type=int, default=4500, help="longest RNA kept (= padding_len; the reference trained with 100, train.py:57)")
    ap.add_argument("--neighbours", type=int, default=30)
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--dropout", type=float, default=None, help="default: the reference's 0.4 (rnampnn.py:47)")
    ap.add_argument("--nan-policy", default="skip", choices=["skip", "fill"])
    ap.add_argument("--train-precision", default="bf16", choices=["bf16", "f32"],
                    help="bf16 = the reference's bf16-mixed trainer setting (utils/train.py:109); f32 = exact")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the fused flat-buffer Adam")
    ap.add_argument("--global-t-norm", action="store_true",
                    help="normalise with the MAX padded length over the ranks (an N-rank step == the 1-rank step on the union batch); "
                         "default: each rank its own batch length, as the reference's DDP ranks do")
    ap.add_argument("--no-validation", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    return ap.parse_args(argv)


def run(args, log=print):
    """-> dict(epochs=[dict(train_loss, val_micro, val_macro, nt_per_s, steps, seconds)], n_train, n_val)."""
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.data:
        items = [(c, y) for _, c, y in load_rna_dir(args.data, max_len=args.max_len, nan_policy=args.nan_policy)]
    else:
        if args.lengths_file:
            lens = [int(n) for n in np.load(args.lengths_file, allow_pickle=False) if 0 < int(n) <= args.max_len]
        else:
            lens = synth.synth_lengths(args.synthetic or 256, 30, min(140, args.max_len), seed=1)
        items = [(synth.synth_rna(int(n), i, seed=1), synth.synth_labels(int(n), i, seed=1)) for i, n in enumerate(lens)]
    n_val = 0 if args.no_validation else max(1, len(items) // 20)
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
    trainer = Trainer(model, opt, sched, world=world, rank=rank, global_t_norm=args.global_t_norm, seed=args.seed)
    train_lens = [c.shape[0] for c, _ in train]
    val_lens = [c.shape[0] for c, _ in val]
    out = dict(epochs=[], n_train=len(train), n_val=len(val))
    for epoch in range(args.epochs):
        rec = trainer.run_epoch(train, train_lens, epoch, args.batch_size, args.max_nt)
        if world > 1:       # whole-job rate: all nucleotides / slowest rank
            t = torch.tensor([rec["seconds"], float(rec["nt"])], dtype=torch.float64, device=dev)
            tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            rec["nt_per_s"] = float(t[1] / tmax[0])
        micro, macro = trainer.validate(val, val_lens, args.batch_size, args.max_nt) if val else (float("nan"), float("nan"))
        rec.update(val_micro=micro, val_macro=macro)
        out["epochs"].append(rec)
        if rank == 0:
            log(f"epoch {epoch}: train_loss {rec['train_loss']:.4f}  val_recovery micro {micro:.4f} macro {macro:.4f}  "
                f"{rec['nt_per_s']:.0f} nt/s end to end ({rec['steps']} steps, {rec['seconds']:.2f} s, {world} rank(s))")
    out["model"] = model
    return out


def main():
    run(parse())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
