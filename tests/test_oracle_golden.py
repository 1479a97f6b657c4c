"""Pin the CPU oracle against the golden vectors produced by the reference's own modules
(tools/gen_golden.py).  Tolerance: the reference's own fp32-vs-fp64 noise floor is 3.8e-6
on logits; the oracle must sit within 2e-5 of the reference fp32 run."""
import numpy as np
import pytest
import torch

from conftest import FULL_CASES, GOLDEN_CASES
from oracle import rnampnn_oracle as O
from rnampnn.utils import synth

ATOL = 2e-5


def _setup(golden, name, dtype=torch.float32):
    arrs, hp, shapes = golden(name)
    cfg = O.OracleConfig(**{k: v for k, v in hp.items() if k in O.OracleConfig.__dataclass_fields__})
    sd = O.state_dict_from_numpy(synth.closed_form_state_dict(shapes), dtype)
    coords = torch.from_numpy(arrs["coords"]).to(dtype)
    mask = torch.from_numpy(arrs["mask"]).to(dtype)
    return arrs, cfg, sd, coords, mask


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_forward_matches_reference(golden, name):
    arrs, cfg, sd, coords, mask = _setup(golden, name)
    taps = {}
    logits, emb = O.forward(coords, mask, sd, cfg, taps=taps)
    ref_idx = O.canonical_edge_index(torch.from_numpy(arrs["edge_index"]).long(), mask)
    assert torch.equal(taps["edge_index"], ref_idx)
    L = cfg.num_res_mpnn_layers
    # h after 10 layers reaches |h| ~ 6; fp32 re-association noise there is ~1e-5 relative
    assert np.abs(taps[f"h{L}"].numpy() - arrs["hL"]).max() < 2e-4
    assert np.abs(logits.numpy() - arrs["logits"]).max() < ATOL
    # against the fp64 run of the reference: still inside the fp32 noise floor
    assert np.abs(logits.numpy() - arrs["logits_f64"]).max() < 2e-5
    labels = torch.from_numpy(arrs["labels"])
    assert abs(float(O.loss_double_softmax(logits, mask, labels)) - float(arrs["loss"])) < 1e-5
    # argmax agreement (ties excluded by construction of the margin check)
    ref_arg = arrs["logits"].argmax(-1)
    top2 = np.sort(arrs["logits"], -1)
    clear = (top2[..., -1] - top2[..., -2]) > 1e-4
    assert (logits.argmax(-1).numpy() == ref_arg)[clear].all()


@pytest.mark.parametrize("name", FULL_CASES)
def test_stage_taps_match_reference(golden, name):
    arrs, cfg, sd, coords, mask = _setup(golden, name)
    taps = {}
    O.forward(coords, mask, sd, cfg, taps=taps)
    idx = O.canonical_edge_index(torch.from_numpy(arrs["edge_index"]).long(), mask)
    en = arrs["e0"].shape[1]
    raw_edge = O.edge_raw_features(coords, mask, idx)
    # distances of invalid edges are exactly 1e6 in both; compare relative on the 1e6 entries
    assert np.allclose(raw_edge[:, :en].numpy(), arrs["edge_raw"], rtol=1e-5, atol=1e-4)
    for key, tol in (("raw", 1e-3), ("h0", 1e-4), ("h1", 1e-4), ("h_post", 1e-4), ("raw_emb", 1e-4)):
        got = taps[key].numpy()
        ref = arrs[key]
        if key == "raw":   # padded rows hold 1e6: relative compare
            assert np.allclose(got, ref, rtol=1e-5, atol=1e-5), key
        else:
            assert np.abs(got - ref).max() < tol, key
    # edge tensors: only rows of valid nodes AND valid edges are consumed downstream; the reference
    # leaves garbage (un-masked residual) on invalid slots after the first update (mpnn.py:263)
    valid = (idx[:, :en] != -1).unsqueeze(-1).numpy()
    assert np.abs((taps["e0"][:, :en].numpy() - arrs["e0"]) * valid).max() < 1e-4
    assert np.abs((taps["e1"][:, :en].numpy() - arrs["e1"]) * valid).max() < 1e-4


def test_fp64_oracle_tracks_reference_fp64(golden):
    arrs, cfg, sd, coords, mask = _setup(golden, "c1_1b23_k16_P66", torch.float64)
    logits, _ = O.forward(coords, mask, sd, cfg)
    assert np.abs(logits.numpy() - arrs["logits_f64"]).max() < 1e-9


def test_phantom_edge_rule(golden):
    """n=5, T=8, k=6: four real neighbours, ONE phantom edge to a padded residue, then -1."""
    arrs, cfg, sd, coords, mask = _setup(golden, "phantom_n5_T8_k6")
    ref = arrs["edge_index"][0]
    assert (ref[:5, 4] >= 5).all() and (ref[:5, 5] == -1).all() and (ref[5:] == -1).all()
    mine = O.knn_graph(coords, mask, 6)[0].numpy()
    assert (mine[:5, 4] == 5).all() and (mine[:5, :4] == ref[:5, :4]).all()


def test_padding_dependent_graph_norm():
    """Closed form of the variance: padded rows contribute mean^2 each (functional.py:33-38)."""
    torch.manual_seed(0)
    x = torch.randn(2, 9, 16, dtype=torch.float64)
    mask = torch.zeros(2, 9, dtype=torch.float64)
    mask[0, :5] = 1
    mask[1, :9] = 1
    scale, shift = torch.rand(16, dtype=torch.float64) + 0.5, torch.rand(16, dtype=torch.float64)
    a = O.graph_norm(x, mask, scale, shift)
    b = O._graph_norm_ttot(x, mask, scale, shift, 9)
    assert (a - b).abs().max() < 1e-12
    # growing the node axis changes the result of a padded RNA only through the closed form
    xp = torch.cat([x, torch.zeros(2, 4, 16, dtype=torch.float64)], 1)
    mp = torch.cat([mask, torch.zeros(2, 4, dtype=torch.float64)], 1)
    c = O.graph_norm(xp, mp, scale, shift)[:, :9]
    d = O._graph_norm_ttot(x, mask, scale, shift, 13)
    assert (c - d).abs().max() < 1e-12
