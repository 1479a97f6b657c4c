"""A/B of the embed-fused first ResMPNN launch (RNAMPNN_EMBED_FUSED=1) against the two-launch form: equality of logits and of the layer-1 edge tap."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rna-mpnn_amd"))
from rnampnn.model.rnampnn import RNAMPNN
from rnampnn.utils import synth

coords, mask, _ = synth.synth_batch([64, 20, 47, 33, 5, 58, 31, 1], first_index=77)
model = RNAMPNN(precision="bf16", num_res_neighbours=30, num_res_mpnn_layers=4, padding_len=64)
sd = synth.closed_form_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to("cuda:0").eval()
c, m = torch.from_numpy(coords), torch.from_numpy(mask)
os.environ.pop("RNAMPNN_EMBED_FUSED", None)
base = model(c, m).clone()
tb = {k: v.clone() for k, v in model.forward_taps(c, m, ["h_layer", "e_layer"], tap_layer=1).items()}
os.environ["RNAMPNN_EMBED_FUSED"] = "1"
alt = model(c, m).clone()
ta = {k: v.clone() for k, v in model.forward_taps(c, m, ["h_layer", "e_layer"], tap_layer=1).items()}
print("logits equal:", torch.equal(base, alt), "max|d|:", float((base - alt).abs().max()), "finite:", bool(torch.isfinite(alt).all()))
for k in tb:
    if torch.is_tensor(tb[k]):
        print(k, "equal:", torch.equal(tb[k], ta[k]), "max|d|:", float((tb[k] - ta[k]).abs().max()))
