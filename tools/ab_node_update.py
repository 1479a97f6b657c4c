"""Per-layer differences between the per-RNA node update (in-kernel GraphNorm statistics) and the two-launch form (RNAMPNN_NODE_UPDATE_RNA=0)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rna-mpnn_amd"))
from rnampnn.model.rnampnn import RNAMPNN
from rnampnn.utils import synth

for T, lens in ((64, [64, 20, 47, 33, 5, 58, 31, 1]), (150, [150, 97, 1, 129, 33, 140])):
    coords, mask, _ = synth.synth_batch(lens, first_index=11)
    model = RNAMPNN(precision="bf16", num_res_neighbours=30, num_res_mpnn_layers=3, padding_len=T)
    sd = synth.closed_form_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to("cuda:0").eval()
    c, m = torch.from_numpy(coords), torch.from_numpy(mask)
    for layer in (0, 1, 2, 3):
        names = ["h0"] if layer == 0 else ["h_layer", "e_layer"]
        os.environ.pop("RNAMPNN_NODE_UPDATE_RNA", None)
        a = {k: v.clone() for k, v in model.forward_taps(c, m, names, tap_layer=layer).items() if torch.is_tensor(v)}
        os.environ["RNAMPNN_NODE_UPDATE_RNA"] = "0"
        b = {k: v.clone() for k, v in model.forward_taps(c, m, names, tap_layer=layer).items() if torch.is_tensor(v)}
        for k in a:
            d = (a[k] - b[k]).abs()
            print(f"T={T} layer={layer} {k}: max|d|={float(d.max()):.3e} mean|d|={float(d.mean()):.3e} scale={float(b[k].abs().max()):.3e} finite={bool(torch.isfinite(a[k]).all())}")
