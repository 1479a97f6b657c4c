#!/usr/bin/env python3
"""Training-step throughput (taped forward + HIP backward + fused Adam) at the C2 shape.
usage: python tools/train_probe.py [B=64] [precision=bf16|f32] [steps=5] [lo=100] [hi=140] [graph=0|1]   (C2 shape: lo=100 hi=500)
graph=1: the step is one replay of a captured hipGraph (rnampnn.model.rnampnn.CapturedTrainStep) + the fused Adam launch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rna-mpnn_amd"))
import torch
from rnampnn.model.rnampnn import RNAMPNN
from rnampnn.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lo = int(sys.argv[4]) if len(sys.argv) > 4 else 100
hi = int(sys.argv[5]) if len(sys.argv) > 5 else 140
graph = int(sys.argv[6]) if len(sys.argv) > 6 else 0
lens = synth.synth_lengths(B, lo, hi, seed=0)
coords, mask, labels = synth.synth_batch(lens)
model = RNAMPNN(precision=prec, num_res_neighbours=30, padding_len=int(mask.shape[1])).to("cuda:0").train()
(opt,), _ = model.configure_optimizers(fused=True)
c, m, y = (torch.from_numpy(a).cuda() for a in (coords, mask, labels))
cap = None
if graph:
    from rnampnn.model.rnampnn import CapturedTrainStep
    cap = CapturedTrainStep(model, int(c.shape[0]), int(c.shape[1]))
    y = y.to(torch.int32)
for it in range(steps + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    loss = cap(y, c, m, seed=it + 1) if cap is not None else model.loss_and_grad(y, c, m)
    opt.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"train step {prec}{' (hipGraph)' if graph else ''}: B={B} nt={int(lens.sum())}  {dt * 1e3:.1f} ms/step  {int(lens.sum()) / dt:.0f} nt/s  loss {float(loss):.4f}")
