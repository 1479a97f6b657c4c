#!/usr/bin/env python3
"""Throughput of the device-side tree read-out (row F4) at the reference's model shape: 150 rounds x 4 classes, depth 8, 256 features,
on the embeddings of a C2 batch (30,559 rows).  Seeded random forest in XGBoost's JSON schema (rnampnn/utils/synth.py: synth_xgb_json)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rna-mpnn_amd")); 
import __graft_entry__ as g
g.load_only()
from rnampnn.model.xgb import GBDTReadout, parse_xgboost_json
from rnampnn.utils.synth import synth_xgb_json as random_xgb_json
a = parse_xgboost_json(random_xgb_json(150, 4, 256, 8, seed=1))
gb = GBDTReadout(a)
x = torch.randn(30559, 256, device="cuda")
for _ in range(3): gb.predict(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): gb.predict(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(json.dumps(dict(metric="gbdt_readout_rows_per_s", value=30559 / dt, us_per_call=dt * 1e6, trees=len(a["tree_class"]), nodes=int(a["tree_offsets"][-1]))))
