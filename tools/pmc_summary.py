#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch.
usage: tools/pmc_summary.py gpurun_out/<dir> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_mpnn_bf16<true, true>"
agg = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(agg):
    s, n = agg[k]
    print(f"{k:40s} dispatches {n:4d}  mean/dispatch {s / n:16.1f}")
