#!/usr/bin/env python3
"""Eager launches vs one captured hipGraph for the C2 forward (bf16): does launch overhead / inter-kernel gap matter?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rna-mpnn_amd"))
import __graft_entry__ as g
g.load_only()
import torch
from rnampnn.model.rnampnn import RNAMPNN
from rnampnn.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lens = synth.synth_lengths(B, 100, 140, seed=0)
coords, mask, _ = synth.synth_batch(lens)
model = RNAMPNN(precision="bf16", num_res_neighbours=30, padding_len=int(mask.shape[1])).to("cuda:0").eval()
c, m = torch.from_numpy(coords).cuda(), torch.from_numpy(mask).cuda()
nt = int(lens.sum())
def run(): return model._run(c, m)["logits"]
for _ in range(5): run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): run()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 30
ws = model.new_workspace(B, int(mask.shape[1]))
def run_p(): return model._run(c, m, workspace=ws)["logits"]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): run_p()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = run_p()
for _ in range(5): graph.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): graph.replay()
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 30
print(f"B={B} nt={nt}: eager {te*1e3:.3f} ms ({nt/te/1e6:.2f} M nt/s)   graph {tg*1e3:.3f} ms ({nt/tg/1e6:.2f} M nt/s)")
