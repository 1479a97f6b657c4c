#!/usr/bin/env python3
"""Build tests/data/c3_train_lengths.npy: the sequence lengths (an int32 array, DATA only) of the 2,083 ids of the reference's
data/train_data.csv, in file order - the length mix of BASELINE.json configs[2] ("train.py one epoch on data/train_data.csv").
The config-3 epoch leg of bench.py (--train-epoch) and rna-mpnn_amd/train.py --lengths-file run seeded synthetic RNAs of exactly
these lengths.  Runs in the build container only."""
import csv, os, sys
import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data/train_data.csv"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "data", "c3_train_lengths.npy")
with open(REF) as f:
    lens = np.array([len(row["seq"]) for row in csv.DictReader(f)], dtype=np.int32)
np.save(OUT, lens)
print(len(lens), "ids,", int(lens.sum()), "nt, min", int(lens.min()), "median", int(np.median(lens)), "max", int(lens.max()), "->", OUT)
