#!/usr/bin/env python3
"""Throughput of the rdesign forward (row F3) on a C2-shaped batch (64 RNAs x 100..500 nt): tools/rdesign_probe.py [precision] [steps].
Prints one JSON line (nt/s, ms/step).  Run under tools/kstats_rdesign.sh for the per-kernel breakdown."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rna-mpnn_amd"))
import __graft_entry__ as g  # noqa: E402

g.load_only()
from rdesign.model.rdesign import RNAModel  # noqa: E402
from rnampnn.utils import synth  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lens = [int(v) for v in synth.synth_lengths(64, 100, 500, seed=0)]
T = max(lens)
X = np.zeros((64, T, 6, 3), np.float32); mask = np.zeros((64, T), np.float32)
for i, n in enumerate(lens):
    X[i, :n] = synth.synth_rna(n, i)[:, :6]; mask[i, :n] = 1
torch.manual_seed(0)
m = RNAModel(precision=prec).cuda().eval()
Xd, md = torch.from_numpy(X).cuda(), torch.from_numpy(mask).cuda()
n_valid = int(mask.sum())
for _ in range(3):
    m._run(Xd, md, want=("logits",), n_valid=n_valid)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    m._run(Xd, md, want=("logits",), n_valid=n_valid)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps(dict(metric="rdesign_forward_nt_per_s", value=n_valid / dt, ms_per_step=dt * 1e3, nt=n_valid, precision=prec,
                      config="RNAModel defaults (k=25, 9 layers, dense 256), 64 RNAs x 100..500 nt synthetic")))
