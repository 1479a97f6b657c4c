"""CPU tests of the host-side pipeline and of the oracle's train-mode pieces: no kernel is launched."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO
from oracle import rnampnn_oracle as O
from rnampnn.utils import synth
from rnampnn.utils.data import PackedLoader, bucket_batches, fill_nan_deterministic, load_rna_dir, pack_batch


@pytest.fixture(scope="module")
def c3_dir(tmp_path_factory):
    z = np.load(os.path.join(REPO, "tests", "data", "c3_subset.npz"), allow_pickle=False)
    root = tmp_path_factory.mktemp("c3cpu")
    os.makedirs(root / "coords"); os.makedirs(root / "seqs")
    for rid in z["ids"]:
        rid = str(rid)
        np.save(root / "coords" / (rid + ".npy"), z["coords/" + rid])
        with open(root / "seqs" / (rid + ".fasta"), "w") as f:
            f.write(f">{rid}\n{str(z['seq/' + rid])}\n")
    # one structure with missing atoms (NaN), as 819 of the reference's 2,317 coordinate files have
    c = z["coords/1B23_1_R"].copy().astype(np.float64)
    c[3, 2] = np.nan; c[10, 6] = np.nan; c[20] = np.nan
    np.save(root / "coords" / "NAN_TEST.npy", c)
    with open(root / "seqs" / "NAN_TEST.fasta", "w") as f:
        f.write(">NAN_TEST\n" + str(z["seq/1B23_1_R"]) + "\n")
    return str(root)


def test_load_rna_dir_and_nan_policies(c3_dir):
    skip = load_rna_dir(c3_dir, max_len=4500)
    assert len(skip) == 59 and all(not np.isnan(c).any() for _, c, _ in skip)
    assert [rid for rid, _, _ in skip] == sorted(rid for rid, _, _ in skip)          # id order: deterministic
    lens = sorted(c.shape[0] for _, c, _ in skip)
    assert lens[0] == 1 and lens[-1] == 2436
    assert len(load_rna_dir(c3_dir, max_len=200)) == 58                                 # the ribosomal RNA is filtered
    fill = dict((rid, c) for rid, c, _ in load_rna_dir(c3_dir, max_len=4500, nan_policy="fill"))
    assert len(fill) == 60 and not np.isnan(fill["NAN_TEST"]).any()
    f, ref = fill["NAN_TEST"], fill["1B23_1_R"]
    assert abs(np.linalg.norm(f[3, 2] - f[3, 0]) - 1.5) < 1e-4                          # 1.5 A from the first present atom
    assert abs(np.linalg.norm(f[10, 6] - f[10, 5]) - 4.0) < 1e-4                        # atom 6: 4 A from atom 5
    assert (f[20] == 0).all()                                                           # a residue with no atom at all -> 0
    untouched = np.ones(f.shape[:2], bool); untouched[3, 2] = untouched[10, 6] = False; untouched[20] = False
    assert np.array_equal(f[untouched], ref[untouched])
    again = dict((rid, c) for rid, c, _ in load_rna_dir(c3_dir, max_len=4500, nan_policy="fill"))
    assert np.array_equal(again["NAN_TEST"], f)                                         # same file -> same coordinates
    assert not np.array_equal(fill_nan_deterministic(np.full((2, 7, 3), np.nan, np.float32), "a"), f[:2])
    with pytest.raises(ValueError):
        load_rna_dir(c3_dir, nan_policy="random")


def test_bucket_batches_partition_and_bounds():
    lens = [int(x) for x in synth.synth_lengths(300, 1, 300, seed=5)] + [2436]
    batches = bucket_batches(lens, batch_size=16, max_rows=2048, seed=3)
    assert sorted(sum(batches, [])) == list(range(len(lens)))                           # every RNA exactly once
    for b in batches:
        T = max(lens[i] for i in b)
        assert len(b) <= 16 and (len(b) * T <= 2048 or len(b) == 1)
        assert max(lens[i] for i in b) - min(lens[i] for i in b) <= 300                 # length-sorted buckets
    assert [2436 in [lens[i] for i in b] for b in batches].count(True) == 1
    assert bucket_batches(lens, 16, 2048, seed=3) == batches                            # counter RNG: reproducible
    assert bucket_batches(lens, 16, 2048, seed=4) != batches                            # ... and seed-dependent order
    assert sorted(map(tuple, bucket_batches(lens, 16, 2048, seed=4))) == sorted(map(tuple, batches))


def test_pack_batch_and_packed_loader_on_cpu():
    items = [np.full((n, 7, 3), float(n), np.float32) for n in (5, 9, 3, 12, 7)]
    packed, cu, max_len = pack_batch([torch.from_numpy(c) for c in items[:3]], pin=False)
    assert cu.tolist() == [0, 5, 14, 17] and max_len == 9 and packed.shape == (17, 7, 3)
    assert float(packed[5:14].mean()) == 9.0
    out = list(PackedLoader(items, [[0, 2], [1, 3, 4]]))
    assert [o[3] for o in out] == [[0, 2], [1, 3, 4]]
    assert out[1][1].tolist() == [0, 9, 21, 28] and out[1][2] == 12
    with pytest.raises(IndexError):                                                     # loader errors surface in the consumer
        list(PackedLoader(items, [[0, 99]]))


def test_dropout_hash_statistics_and_reproducibility():
    idx = np.arange(1 << 20, dtype=np.uint64)
    for p in (0.1, 0.4):
        m = O.dropout_multiplier(12345, O.site_msg(3, 1), idx, p)
        assert set(np.unique(m)) == {np.float32(0.0), np.float32(1.0) / (np.float32(1.0) - np.float32(p))}
        assert abs(float((m == 0).mean()) - p) < 3e-3
    a = O.dropout_multiplier(7, 10, idx, 0.4)
    assert np.array_equal(a, O.dropout_multiplier(7, 10, idx, 0.4))
    for other in (O.dropout_multiplier(8, 10, idx, 0.4), O.dropout_multiplier(7, 11, idx, 0.4),
                  O.dropout_multiplier(7 + (1 << 32), 10, idx, 0.4), O.dropout_multiplier(7, 10, idx + np.uint64(1 << 32), 0.4)):
        assert 0.45 < float((a != other).mean()) < 0.51                                  # seed lo / site / seed hi / index hi all matter
    # neighbouring elements are uncorrelated (lag-1 agreement = 0.4^2 + 0.6^2 = 0.52)
    assert abs(float((a[1:] == a[:-1]).mean()) - 0.52) < 5e-3


def test_oracle_train_mode_is_eval_mode_at_p0_and_unbiased():
    lens = [9, 14]
    coords, mask, _ = synth.synth_batch(lens, first_index=3)
    from rnampnn.model._schema import DEFAULT_HPARAMS, state_dict_shapes
    hp = dict(DEFAULT_HPARAMS, num_res_neighbours=5, num_res_mpnn_layers=2, padding_len=16, embedding_ffn_dim=64,
              post_fusion_ffn_dim=64, num_raw_ffn_dim=64, readout_hidden_dim=64)
    sd = O.state_dict_from_numpy(synth.closed_form_state_dict(state_dict_shapes(hp)))
    cfg = O.OracleConfig(**{k: v for k, v in hp.items() if k in O.OracleConfig.__dataclass_fields__})
    c, m = torch.from_numpy(coords), torch.from_numpy(mask)
    base, _ = O.forward(c, m, sd, cfg)
    same, _ = O.forward(c, m, sd, cfg, dropout=0.0, seed=99)
    assert torch.equal(base, same)
    d1, _ = O.forward(c, m, sd, cfg, dropout=0.4, seed=1)
    d2, _ = O.forward(c, m, sd, cfg, dropout=0.4, seed=2)
    assert not torch.equal(d1, d2) and torch.isfinite(d1).all()
    assert (d1[m == 0] == 0).all()                                                      # padded rows stay zero in train mode


def test_epoch_plan_pad_batch_and_config3_length_fixture():
    """Round 3: the trainer's epoch plan is a pure function of (lengths, world, seed) whose per-rank columns are disjoint, equal
    in step count and length-matched per step; pad_batch reproduces the collate layout; the committed config-3 length list is
    the reference's data/train_data.csv (2,083 ids, 1,205,038 nt, 1 ... 4,417 nt, median 76)."""
    import os
    import numpy as np
    import torch
    from conftest import REPO
    from rnampnn.utils.data import pad_batch
    from rnampnn.utils.train import plan_epoch
    lens = np.load(os.path.join(REPO, "tests", "data", "c3_train_lengths.npy"), allow_pickle=False)
    assert lens.dtype == np.int32 and len(lens) == 2083 and int(lens.sum()) == 1205038
    assert int(lens.min()) == 1 and int(lens.max()) == 4417 and int(np.median(lens)) == 76
    cols = [plan_epoch(lens, r, 4, 512, 32768, seed=3) for r in range(4)]
    steps = {len(c[0]) for c in cols}
    assert len(steps) == 1
    seen = [i for c in cols for b in c[0] for i in b]
    assert set(seen) == set(range(len(lens)))                    # nothing is dropped: an unfilled last round is padded by repetition ...
    assert len(seen) - len(set(seen)) <= 3 * 512                  # ... of at most world - 1 of its own batches
    for world in (4, 8):                                          # every index in every epoch, at every world size; batch composition changes per epoch
        per_epoch = []
        for ep in range(3):
            plan = [plan_epoch(lens, r, world, 512, 32768, seed=10 + ep) for r in range(world)]
            assert len({len(p[0]) for p in plan}) == 1
            assert {i for p in plan for b in p[0] for i in b} == set(range(len(lens)))
            per_epoch.append({tuple(sorted(b)) for p in plan for b in p[0]})
        assert len(per_epoch[0] & per_epoch[1]) < 0.5 * len(per_epoch[0])
    assert all(cols[0][1] == c[1] for c in cols)                                  # every rank derives the same global lengths
    for s in range(steps.pop()):
        rows = [len(c[0][s]) * max(int(lens[i]) for i in c[0][s]) for c in cols]
        assert max(rows) <= 32768 or all(len(c[0][s]) == 1 for c in cols if len(c[0][s]) * max(int(lens[i]) for i in c[0][s]) > 32768)
        assert cols[0][1][s] == max(max(int(lens[i]) for i in c[0][s]) for c in cols)
    assert plan_epoch(lens, 1, 4, 512, 32768, seed=3)[0] == cols[1][0] and plan_epoch(lens, 1, 4, 512, 32768, seed=4)[0] != cols[1][0]
    one = plan_epoch(lens, 0, 1, 512, 32768, seed=0)[0]
    assert sorted(i for b in one for i in b) == list(range(len(lens)))
    items = [(np.full((n, 7, 3), float(n), np.float32), np.arange(n) % 4) for n in (3, 5, 2)]
    y, c, m, ln = pad_batch(items, pin=False)
    assert ln == [3, 5, 2] and c.shape == (3, 5, 7, 3) and y.dtype == torch.int32
    assert m.tolist() == [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 0, 0, 0]]
    assert float(c[0, 3:].abs().sum()) == 0 and y[1].tolist() == [0, 1, 2, 3, 0] and float(c[2, 1, 6, 2]) == 2.0
