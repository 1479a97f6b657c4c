#!/usr/bin/env python3
"""Does the hipGraph-captured training step still equal the eager one when the runtime's memset / copy graph NODES come back?
(RNAMPNN_DBG_MEMNODE: 1 = memsets, 2 = copies, 3 = both; round 3 replaced them by kernels after NaN losses at 16 RNAs.)
usage (GPU box): python tools/capture_memnode_probe.py <n_rnas> <mode>"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))
n_rnas, mode = int(sys.argv[1]), sys.argv[2]
import torch
from rnampnn.model.rnampnn import RNAMPNN, CapturedTrainStep
from rnampnn.utils import synth
lens = synth.synth_lengths(n_rnas, 100, 140, seed=0, first_index=0)
coords, mask, labels = synth.synth_batch(lens, first_index=0, seed=0)
model = RNAMPNN(precision="bf16", num_res_neighbours=30, padding_len=int(mask.shape[1]))
sd = synth.closed_form_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to("cuda:0").train()
c, m, y = (torch.from_numpy(x).cuda() for x in (coords, mask, labels))
eager = float(model.loss_and_grad(y, c, m, seed=4242))
g_eager = model.flat_grad.clone()
os.environ["RNAMPNN_DBG_MEMNODE"] = mode              # from here on: capture + replay with the runtime nodes
cap = CapturedTrainStep(model, n_rnas, int(mask.shape[1]))
out = []
for rep in range(3):
    l = float(cap(y, c, m, seed=4242))
    torch.cuda.synchronize()
    same = torch.equal(model.flat_grad, g_eager)
    out.append((l, same, bool(torch.isfinite(model.flat_grad).all())))
print(f"n_rnas {n_rnas} mode {mode}: eager loss {eager:.6f}; replays (loss, grad == eager, finite): {out}")
