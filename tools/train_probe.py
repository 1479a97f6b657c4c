#!/usr/bin/env python3
"""Time a few training steps (loss_and_grad) on a C2-shaped batch; used under rocprofv3 for the backward profile."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
from rnampnn.model.rnampnn import RNAMPNN
from rnampnn.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lens = synth.synth_lengths(B, 100, 140, seed=0)
coords, mask, labels = synth.synth_batch(lens)
model = RNAMPNN(precision="f32", num_res_neighbours=30, padding_len=int(mask.shape[1])).to("cuda").eval()
c, m, y = (torch.from_numpy(a).cuda() for a in (coords, mask, labels))
model.loss_and_grad(y, c, m); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = model.loss_and_grad(y, c, m)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"train step: B={B} nt={int(lens.sum())} {dt*1e3:.1f} ms/step -> {lens.sum()/dt:.0f} nt/s, loss {float(loss):.4f}")
