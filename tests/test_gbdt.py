"""Row F4: the device-side gradient-boosted-tree read-out (`rnampnn_gbdt_*`) vs the CPU restatement `oracle/gbdt_oracle.py` on seeded
random forests written in XGBoost's JSON model schema.  PARITY UNPINNED (no xgboost in the image, no fitted model in the reference)."""
import json

import numpy as np
import pytest
import torch

from oracle import gbdt_oracle


from rnampnn.utils.synth import synth_xgb_json as random_xgb_json  # noqa: E402


def test_json_parser_and_oracle_on_a_hand_made_tree(tmp_path):
    from rnampnn.model.xgb import parse_xgboost_json
    # x0 < 0.5 ? (x1 < -1 ? 1.0 : 2.0) : 3.0 for class 0; a single leaf 0.25 for class 1
    model = {"learner": {"learner_model_param": {"num_class": "2", "num_feature": "2", "base_score": "0"},
                         "gradient_booster": {"name": "gbtree", "model": {"tree_info": [0, 1], "trees": [
                             dict(left_children=[1, 3, -1, -1, -1], right_children=[2, 4, -1, -1, -1], split_indices=[0, 1, 0, 0, 0],
                                  split_conditions=[0.5, -1.0, 3.0, 1.0, 2.0], default_left=[1, 0, 0, 0, 0], categories_nodes=[]),
                             dict(left_children=[-1], right_children=[-1], split_indices=[0], split_conditions=[0.25], default_left=[0],
                                  categories_nodes=[])]}}}}
    path = tmp_path / "m.json"
    path.write_text(json.dumps(model))
    a = parse_xgboost_json(str(path))
    assert a["tree_offsets"].tolist() == [0, 5, 6] and a["num_class"] == 2
    X = np.array([[0.0, -2.0], [0.0, 0.0], [0.5, 0.0], [np.nan, np.nan]], np.float32)
    pred, m = gbdt_oracle.predict(a, X)
    # strict '<': x0 = 0.5 goes right; NaN at the root follows default_left = 1, NaN at node 1 follows default_left = 0 -> right
    assert m[:, 0].tolist() == [1.0, 2.0, 3.0, 2.0] and m[:, 1].tolist() == [0.25] * 4
    assert pred.tolist() == [0, 0, 0, 0]


def test_create_rejects_malformed_models():
    import __graft_entry__ as g
    g.build()
    from rnampnn.model.xgb import GBDTReadout, parse_xgboost_json
    a = parse_xgboost_json(random_xgb_json(2, 4, 16, 3, 0))
    bad = dict(a); bad["left_children"] = a["left_children"].copy(); bad["left_children"][0] = 0        # a cycle
    with pytest.raises(ValueError):
        GBDTReadout(bad)
    bad = dict(a); bad["split_indices"] = a["split_indices"].copy(); bad["split_indices"][0] = 99        # feature out of range
    with pytest.raises(ValueError):
        GBDTReadout(bad)
    if not torch.cuda.is_available():                               # a well-formed model needs the device for its arrays: no CPU fallback
        with pytest.raises(RuntimeError):
            GBDTReadout(a)


@pytest.mark.gpu
@pytest.mark.parametrize("rounds,depth,nfeat,nrows", [(5, 4, 16, 257), (150, 8, 256, 300)])
def test_device_prediction_matches_oracle(rounds, depth, nfeat, nrows):
    """The reference's shape is 150 rounds x 4 classes, depth 8, 256 features (rnampnn.py:136-145).  Margins are added in tree order on
    both sides: bit-exact."""
    from rnampnn.model.xgb import GBDTReadout, parse_xgboost_json
    a = parse_xgboost_json(random_xgb_json(rounds, 4, nfeat, depth, seed=rounds))
    X = np.random.RandomState(1).randn(nrows, nfeat).astype(np.float32)
    X[::17, 3] = np.nan                                            # missing values
    ref_pred, ref_m = gbdt_oracle.predict(a, X)
    g = GBDTReadout(a)
    xd = torch.from_numpy(X).cuda()
    assert np.array_equal(g.margins(xd).cpu().numpy(), ref_m)
    assert np.array_equal(g.predict(xd).cpu().numpy(), ref_pred)
    assert g.predict(xd.reshape(3, nrows // 3, nfeat) if nrows % 3 == 0 else xd).numel() == nrows


@pytest.mark.gpu
def test_rnampnn_predict_uses_the_tree_readout(tmp_path):
    """`RNAMPNN.predict_sequences` with a loaded tree model = the reference's `predict` (embedding -> xgb_readout.predict -> strings)."""
    from rnampnn.model.rnampnn import RNAMPNN
    from rnampnn.utils import synth
    coords, mask, _ = synth.synth_batch([21, 34], first_index=3)
    model = RNAMPNN(num_res_neighbours=8, num_res_mpnn_layers=2, padding_len=40, precision="f32").cuda().eval()
    path = tmp_path / "xgb.json"
    path.write_text(json.dumps(random_xgb_json(6, 4, 256, 4, seed=9)))
    model.load_xgb_readout(str(path))
    c, m = torch.from_numpy(coords), torch.from_numpy(mask)
    seqs = model.predict_sequences(c, m)
    emb = model.embedding(c, m).cpu().numpy()
    ref_pred, _ = gbdt_oracle.predict(model.xgb_readout.arrays, emb.reshape(-1, 256))
    ref_pred = ref_pred.reshape(2, -1)
    want = ["".join("AUCG"[i] for i in ref_pred[b][: n]) for b, n in enumerate([21, 34])]
    assert seqs == want
