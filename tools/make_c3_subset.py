#!/usr/bin/env python3
"""Build tests/data/c3_subset.npz: a <= 1 MB, NaN-free subset of the reference's data/coords + data/seqs
(DATA only: coordinates and sequences of real RNAs; no reference code) for the config-3 tests
(BASELINE.json configs[2]: one epoch of train.py on data/train_data.csv).  Runs in the build container only.

Selection (deterministic): the three 1-nt and two 2-nt RNAs (shortest), 7S9U_1_A (2,436 nt: beyond the 2,400-nt
LDS limit of the round-1 bf16 attention), 1B23_1_R (the C1 golden), and every 17th NaN-free RNA of 10..160 nt in
id order until the budget of 12,000 nucleotides is reached.
"""
import glob, os, sys
import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "data", "c3_subset.npz")

def seq_of(rid):
    return "".join(l.strip() for l in open(os.path.join(REF, "seqs", rid + ".fasta")) if not l.startswith(">"))

ids = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(REF, "coords", "*.npy")))
clean = {}
for rid in ids:
    a = np.load(os.path.join(REF, "coords", rid + ".npy"))
    if a.dtype == np.float32 and not np.isnan(a).any() and os.path.exists(os.path.join(REF, "seqs", rid + ".fasta")):
        s = seq_of(rid)
        if len(s) == a.shape[0] and set(s) <= set("AUCG"):
            clean[rid] = (a, s)
pick = [r for r in clean if clean[r][0].shape[0] == 1][:3] + [r for r in clean if clean[r][0].shape[0] == 2][:2]
pick += ["7S9U_1_A", "1B23_1_R"]
budget = 12000 - sum(clean[r][0].shape[0] for r in pick)
mid = [r for r in clean if 10 <= clean[r][0].shape[0] <= 160 and r not in pick][::17]
for r in mid:
    n = clean[r][0].shape[0]
    if n <= budget:
        pick.append(r); budget -= n
out = {"ids": np.array(pick)}
for r in pick:
    out["coords/" + r] = clean[r][0]
    out["seq/" + r] = np.array(clean[r][1])
np.savez(OUT, **out)
print(len(pick), "RNAs,", sum(clean[r][0].shape[0] for r in pick), "nt ->", OUT, os.path.getsize(OUT), "bytes")
