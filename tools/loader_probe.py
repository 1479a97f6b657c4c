#!/usr/bin/env python3
"""Row F1 measurement: does the var-len host pipeline keep up with the model?  (a) PackedLoader alone (pack into pinned memory + H2D on
the side stream), (b) PackedLoader -> forward_packed end to end, (c) the padded collate + forward the reference's loop would do.
40 batches of the C2 shape (256 RNAs x 100..140 nt) from an in-memory list of per-RNA arrays."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rna-mpnn_amd"))
import __graft_entry__ as g
g.load_only()
from rnampnn.model.rnampnn import RNAMPNN
from rnampnn.utils import synth
from rnampnn.utils.data import PackedLoader, bucket_batches
NB, B = 40, 256
lens = synth.synth_lengths(NB * B, 100, 140, seed=0)
items = [synth.synth_rna(int(n), i) for i, n in enumerate(lens)]
nt = int(lens.sum())
batches = [list(range(i * B, (i + 1) * B)) for i in range(NB)]
model = RNAMPNN(precision="bf16", num_res_neighbours=30, padding_len=140).to("cuda:0").eval()
def loader_only():
    for dp, dc, ml, b in PackedLoader(items, batches, device="cuda:0"):
        pass
    torch.cuda.synchronize()
def end_to_end():
    for dp, dc, ml, b in PackedLoader(items, batches, device="cuda:0"):
        model.forward_packed(dp, dc, ml)
    torch.cuda.synchronize()
def padded():
    for b in batches:
        T = max(items[i].shape[0] for i in b)
        c = np.zeros((len(b), T, 7, 3), np.float32); m = np.zeros((len(b), T), np.float32)
        for j, i in enumerate(b):
            n = items[i].shape[0]; c[j, :n] = items[i]; m[j, :n] = 1
        model(torch.from_numpy(c), torch.from_numpy(m))
    torch.cuda.synchronize()
out = {}
for name, fn in (("packed_loader_only", loader_only), ("packed_loader_forward", end_to_end), ("padded_collate_forward", padded)):
    fn()
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    out[name] = dict(nt_per_s=nt / dt, ms_per_batch=dt / NB * 1e3)
print(json.dumps(out))
