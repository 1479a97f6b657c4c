#!/usr/bin/env python3
"""Where does the bf16 path's logit error come from?  bf16 vs f32 kernels on the same weights, tap by tap, for the real RNA 1B23_1_R
(66 nt, T = 80) at several k: h0, e0, h / e after every ResMPNN layer, h_post, logits - max and mean |delta| next to the tensor's own scale.
usage (GPU box): python tools/tap_errors.py [k ...]"""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))
from rnampnn.model.rnampnn import RNAMPNN
from rnampnn.utils import synth

g = np.load(os.path.join(REPO, "tests", "golden", "c1_1b23_k30_T80.npz"))
coords, mask = torch.from_numpy(g["coords"]), torch.from_numpy(g["mask"])
ks = [int(a) for a in sys.argv[1:]] or [30, 16, 3]
for k in ks:
    models = {}
    for prec in ("f32", "bf16"):
        m = RNAMPNN(precision=prec, num_res_neighbours=k, padding_len=80)
        sd = synth.closed_form_state_dict({n: tuple(v.shape) for n, v in m.state_dict().items()})
        m.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()})
        models[prec] = m.to("cuda:0").eval()
    print(f"--- k = {k}")
    valid = mask.bool()
    for layer in range(1, 11):
        names = ["h0", "e0", "h_layer", "e_layer", "h_post", "edge_index"] if layer == 1 else ["h_layer", "e_layer", "edge_index"]
        if layer == 10:
            names = ["h_layer", "h_post", "edge_index"]
        tf = models["f32"].forward_taps(coords, mask, names, tap_layer=layer)
        tb = models["bf16"].forward_taps(coords, mask, names, tap_layer=layer)
        ok = (tf["edge_index"] >= 0).cpu()
        def rep(name, sel):
            a, b = tf[name].cpu()[sel], tb[name].cpu()[sel]
            d = (a - b).abs()
            print(f"  layer {layer:2d} {name:8s} |ref| mean {a.abs().mean():.3e}  |d| max {d.max():.3e} mean {d.mean():.3e}  rel(mean) {d.mean() / a.abs().mean():.3e}")
        if layer == 1:
            rep("h0", valid); rep("e0", ok)
        rep("h_layer", valid)
        if "e_layer" in names: rep("e_layer", ok)
        if layer == 10: rep("h_post", valid)
    lf, lb = models["f32"](coords, mask).cpu(), models["bf16"](coords, mask).cpu()
    d = (lf - lb).abs()[valid]
    print(f"  logits: std {lf[valid].std():.3e}  |d| max {d.max():.3e} mean {d.mean():.3e}")
