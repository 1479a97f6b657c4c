"""GPU tests at the workloads of BASELINE.json configs 3, 4 and 5 and on TRAINED (separated) logits.

The configs' full sizes do not fit a CPU oracle run inside a test, so each is checked through size-independent
properties (bit-determinism, batch-permutation equivariance, captured-graph == eager) plus an oracle comparison on
the first RNAs of the batch; config 3 runs on a committed NaN-free subset of the reference's own data
(tests/data/c3_subset.npz: 59 RNAs, 1 ... 2,436 nt; built by tools/make_c3_subset.py).
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def _hp(**kw):
    from rnampnn.model._schema import DEFAULT_HPARAMS
    return dict(DEFAULT_HPARAMS, **kw)


def _model(hp, precision, sd_np=None):
    from rnampnn.model.rnampnn import RNAMPNN
    from rnampnn.model._schema import DEFAULT_HPARAMS, state_dict_shapes
    from rnampnn.utils import synth
    model = RNAMPNN(precision=precision, **{k: v for k, v in hp.items() if k in DEFAULT_HPARAMS})
    if sd_np is None:
        sd_np = synth.closed_form_state_dict(state_dict_shapes(hp))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})
    return model.to("cuda:0").eval(), sd_np


def _oracle(hp, sd_np, coords, mask):
    from oracle import rnampnn_oracle as O
    cfg = O.OracleConfig(**{k: v for k, v in hp.items() if k in O.OracleConfig.__dataclass_fields__})
    return O.forward(torch.from_numpy(coords), torch.from_numpy(mask), O.state_dict_from_numpy(sd_np), cfg)[0]


def bf16_tol(ref: torch.Tensor, mask) -> float:
    """bf16 logit tolerance relative to the logit spread (SURVEY 8c's 5e-2 is 70 % of the signal at random init):
    5 % of the standard deviation of the reference logits over valid positions, floor 3e-2."""
    valid = torch.as_tensor(mask).bool()
    return max(3e-2, 0.05 * float(ref[valid].std()))


# ----------------------------------------------------------------------------------------------- trained weights
def test_trained_weights_bf16_tracks_f32_oracle():
    """SURVEY 8c, second half: on weights whose logits are SEPARATED (trained here with the HIP backward until recovery
    > 0.6) the bf16 path must agree with the f32 oracle: argmax agreement >= 99 %, |delta recovery| <= 0.5 pt, logits
    within 5 % of their spread.  Also covers optimizer.zero_grad(set_to_none=True) between steps (the gradient views of
    the flat buffer are re-bound on every call)."""
    from oracle import rnampnn_oracle as O
    from rnampnn.utils import synth
    lens = [41, 33, 56, 28, 47, 60]
    coords, mask, labels = synth.synth_batch(lens, first_index=1200)
    hp = _hp(num_res_neighbours=30, padding_len=64)
    trainer, _ = _model(hp, "f32")
    trainer.train()
    (opt,), _ = trainer.configure_optimizers()
    c, m, y = torch.from_numpy(coords), torch.from_numpy(mask), torch.from_numpy(labels)
    w0 = trainer.readout.readout_layers._modules["0"].weight.detach().clone()
    losses = []
    for it in range(60):
        opt.zero_grad()                                           # set_to_none=True: drops every p.grad
        losses.append(float(trainer.loss_and_grad(y, c, m, dropout=0.0)))
        opt.step()
        if it == 0:
            assert not torch.equal(w0, trainer.readout.readout_layers._modules["0"].weight.detach()), \
                "weights did not move after zero_grad(): gradient views were not re-bound"
    assert losses[-1] < losses[0] - 0.3, (losses[0], losses[-1])
    sd = {k: v.detach().cpu().numpy() for k, v in trainer.state_dict().items()}
    ref = _oracle(hp, sd, coords, mask)
    micro_ref, _, _ = O.recovery(ref, m, y)
    assert micro_ref > 0.6, micro_ref
    valid = m.bool()
    spread = float(ref[valid].std())
    exact, _ = _model(hp, "f32", sd)
    lf = exact(c, m).cpu()
    assert (lf - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max())), float((lf - ref).abs().max())
    fast, _ = _model(hp, "bf16", sd)
    lb = fast(c, m).cpu()
    err = float((lb - ref).abs().max())
    assert err < bf16_tol(ref, m), (err, spread)
    agree = float((lb.argmax(-1) == ref.argmax(-1))[valid].float().mean())
    micro_b, _, _ = O.recovery(lb, m, y)
    print(f"trained: loss {losses[0]:.3f} -> {losses[-1]:.3f}, logit std {spread:.2f}, bf16 |dlogit| {err:.3e}, "
          f"argmax agreement {agree:.4f}, recovery f32-oracle {micro_ref:.4f} bf16 {micro_b:.4f}")
    assert agree >= 0.99
    assert abs(micro_b - micro_ref) <= 0.005


# ----------------------------------------------------------------------------------------------- config 4 shape
def test_config4_shape_properties():
    """BASELINE configs[3] shape: 1,024 RNAs x 200 nt, T = n (no padding term), k = 30, bf16: bit-determinism,
    batch-permutation equivariance, and the first RNAs against the oracle."""
    from rnampnn.utils import synth
    B, n = 1024, 200
    coords = np.stack([synth.synth_rna(n, i, 0) for i in range(B)]).astype(np.float32)
    mask = np.ones((B, n), np.float32)
    hp = _hp(num_res_neighbours=30, padding_len=n)
    model, sd = _model(hp, "bf16")
    c, m = torch.from_numpy(coords).cuda(), torch.from_numpy(mask).cuda()
    a = model(c, m).clone()
    b = model(c, m).clone()
    assert torch.isfinite(a).all() and torch.equal(a, b)
    perm = torch.from_numpy(np.random.default_rng(11).permutation(B)).cuda()
    p = model(c[perm].contiguous(), m[perm].contiguous())
    assert torch.equal(p, a[perm])
    ref = _oracle(hp, sd, coords[:3], mask[:3])
    err = float((a[:3].cpu() - ref).abs().max())
    assert err < bf16_tol(ref, mask[:3]), err
    # a shard of the job (what one of N ranks runs) reproduces its rows bit for bit: T_norm = T = n for every rank
    part = model(c[100:356].contiguous(), m[100:356].contiguous())
    assert torch.equal(part, a[100:356])


# ----------------------------------------------------------------------------------------------- config 5
def test_config5_captured_decode_1000x120x8():
    """BASELINE configs[4]: 1,000 structures x 120 nt, 8 samples at temperature 0.1 through ONE captured hipGraph
    (forward + sampler).  Replay == eager bit for bit (logits and samples); sampled frequencies on a subset follow
    softmax(logits / 0.1); a larger eager call in between must not disturb the capture (private workspace)."""
    from rnampnn.model.rnampnn import CapturedSampler, sample_from_logits
    from rnampnn.utils import synth
    B, n = 1000, 120
    lens = [n] * B
    coords, mask, _ = synth.synth_batch(lens, first_index=5000)
    hp = _hp(num_res_neighbours=30, padding_len=160)
    model, _ = _model(hp, "bf16")
    c, m = torch.from_numpy(coords).cuda(), torch.from_numpy(mask).cuda()
    cap = CapturedSampler(model, B, n, temperature=0.1, n_samples=8)
    logits, samples = cap(c, m, seed=3)
    eager = model(c, m)
    assert torch.equal(logits, eager)
    assert torch.equal(samples, sample_from_logits(eager, m, 0.1, 8, seed=3))
    assert samples.shape == (8, B, n) and int(samples.min()) >= 0 and int(samples.max()) <= 3
    # grow the module's shared workspace with a bigger eager call: the graph keeps its own
    big_c, big_m, _ = synth.synth_batch([150] * 1200, first_index=9000)
    model(torch.from_numpy(big_c).cuda(), torch.from_numpy(big_m).cuda())
    junk = torch.empty(64 << 20, dtype=torch.uint8, device="cuda").fill_(255)       # reuse freed blocks, if any
    logits2, samples2 = cap(c, m, seed=3)
    assert torch.equal(logits2, eager) and torch.equal(samples2, samples)
    del junk
    # frequencies over 64 replays x 8 samples on the first 40 structures
    counts = torch.zeros(40, n, 4, device="cuda")
    reps = 64
    for s in range(reps):
        _, smp = cap(c, m, seed=100 + s)
        counts += torch.nn.functional.one_hot(smp[:, :40].long(), 4).sum(0)
    freq = counts / (8 * reps)
    probs = torch.softmax(eager[:40] / 0.1, -1)
    assert float((freq - probs).abs().max()) < 0.10                 # 512 draws: sigma <= 0.022


# ----------------------------------------------------------------------------------------------- config 3
@pytest.fixture(scope="module")
def c3_dir(tmp_path_factory):
    """tests/data/c3_subset.npz unpacked into the reference's directory layout (coords/<id>.npy + seqs/<id>.fasta)."""
    z = np.load(os.path.join(REPO, "tests", "data", "c3_subset.npz"), allow_pickle=False)
    root = tmp_path_factory.mktemp("c3")
    os.makedirs(root / "coords"); os.makedirs(root / "seqs")
    for rid in z["ids"]:
        rid = str(rid)
        np.save(root / "coords" / (rid + ".npy"), z["coords/" + rid])
        with open(root / "seqs" / (rid + ".fasta"), "w") as f:
            f.write(f">{rid}\n{str(z['seq/' + rid])}\n")
    return str(root)


def test_config3_epochs_on_reference_data_subset(c3_dir):
    """BASELINE configs[2]: train.py on (a subset of) the reference's data, 1 x MI355X, Adam, lengths 1 ... 2,436 nt with
    length-bucketed steps, dropout as the reference (0.4), the trainer loop of rnampnn.utils.train (PaddedLoader, no host syncs).
    The SAME subset, split and dropout seeds through both trainers: the bf16-mixed loss curve (the reference's setting,
    utils/train.py:109) must track the exact-f32 curve epoch by epoch (measured: max difference 0.0094 over 120 epochs), and the
    loss must fall clearly (ln 4 = 1.386 at chance, floor 0.744 for the double-softmax loss): with the reference's optimiser
    (Adam 2e-3, StepLR(15, 0.8): the step size is down to 0.17 x by epoch 120) on 56 real RNAs with dropout 0.4 the epoch-mean
    TRAIN-mode loss falls 1.3855 -> 1.340 .. 1.346 - the end value moves by ~0.01 between builds whose gradients agree with the oracle to
    4e-6 (a different summation order in one backward kernel is enough: 120 epochs of Adam amplify it; measured 1.3402 / 1.3377 and
    1.3418 / 1.3463 for bf16-mixed / f32 on two such builds) - asserted as >= 0.03; the 64-RNA fixed-batch run of
    tests/test_round3_gpu.py::test_matched_recovery_on_trained_logits_64_rnas falls 0.13 in 300 steps and reaches recovery 0.56.

    WHAT THIS TEST PROVES AND WHAT IT DOES NOT: it pins the PLUMBING of config 3 (data loading, the jittered epoch plan, the
    padded loader, 120 epochs without a non-finite value, validation metrics in range) and that the two arithmetic modes of the trainer stay
    together on identical data and masks.  Both bounds (fall >= 0.03, curves within 0.03) are of the size of the build-to-build noise (~0.01), so a
    missing gradient term could pass here: it does NOT prove learning and does not pin the backward.  Those are pinned elsewhere - the bf16-mixed
    and f32 gradients per parameter against the oracle's autograd (tests/test_hip_parity.py: test_bf16_mixed_gradients_match_oracle_autograd,
    test_loss_and_gradients_match_oracle_autograd, test_gradients_with_dropout_match_oracle_autograd_and_are_bit_reproducible), and learning by the
    300-step / 64-RNA run above (loss -0.125, recovery 0.56) and test_trained_weights_bf16_tracks_f32_oracle (60 steps: loss 1.387 -> 0.83)."""
    sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))
    import train as T
    curves = {}
    for prec in ("bf16", "f32"):
        torch.manual_seed(0)                                  # same torch-default initial weights for both runs
        args = T.parse(["--data", c3_dir, "--epochs", "120", "--batch-size", "8", "--max-len", "4500", "--max-nt", "4096",
                        "--train-precision", prec])
        out = T.run(args, log=lambda s: None)
        assert out["n_train"] + out["n_val"] == 59
        curves[prec] = [e["train_loss"] for e in out["epochs"]]
        assert all(np.isfinite(curves[prec]))
        assert 0.0 <= out["epochs"][-1]["val_micro"] <= 1.0 and 0.0 <= out["epochs"][-1]["val_macro"] <= 1.0
    b, f = np.array(curves["bf16"]), np.array(curves["f32"])
    print("config 3 subset: epoch losses bf16-mixed", np.round(b[::15], 4).tolist(), "f32", np.round(f[::15], 4).tolist(),
          f"max |bf16 - f32| {np.abs(b - f).max():.4f}")
    assert b[-1] < b[0] - 0.03 and f[-1] < f[0] - 0.03, (b[0], b[-1], f[0], f[-1])
    assert np.abs(b - f).max() < 0.03, np.abs(b - f).max()       # same data order and dropout masks: the curves stay together


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_config3_longest_rna_matches_oracle(c3_dir, precision):
    """The 2,436-nt RNA of the subset (beyond the 2,400-nt LDS image of the one-pass bf16 attention) together with the
    shortest ones (1 and 2 nt: no neighbours / a single edge), default 10-layer stack, P = 4500: vs the oracle."""
    from rnampnn.utils.data import load_rna_dir
    items = {rid: c for rid, c, _ in load_rna_dir(c3_dir, max_len=4500)}
    hp = _hp(num_res_neighbours=30, padding_len=4500)
    model, sd = _model(hp, precision)
    long_c = items["7S9U_1_A"][None]
    long_m = np.ones((1, long_c.shape[1]), np.float32)
    lg = model(torch.from_numpy(long_c), torch.from_numpy(long_m)).cpu()
    ref = _oracle(hp, sd, long_c, long_m)
    err = float((lg - ref).abs().max())
    assert torch.isfinite(lg).all()
    assert err < (1e-4 if precision == "f32" else bf16_tol(ref, long_m)), err
    shorts = [c for c in items.values() if c.shape[0] <= 2]
    assert len(shorts) == 5
    T = 2
    sc = np.zeros((5, T, 7, 3), np.float32); sm = np.zeros((5, T), np.float32)
    for i, c in enumerate(shorts):
        sc[i, :c.shape[0]] = c; sm[i, :c.shape[0]] = 1
    ls = model(torch.from_numpy(sc), torch.from_numpy(sm)).cpu()
    rs = _oracle(hp, sd, sc, sm)
    assert float((ls - rs).abs().max()) < (1e-4 if precision == "f32" else 3e-2)


# ----------------------------------------------------------------------------------------------- F1 loader
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_packed_loader_forward_matches_oracle(c3_dir, precision):
    """F1: directory -> length buckets -> PackedLoader (pinned, async H2D on a side stream) -> forward_packed, checked
    against the ORACLE run on the padded batch (not against the padded HIP path)."""
    from rnampnn.utils.data import PackedLoader, bucket_batches, load_rna_dir
    items = [it for it in load_rna_dir(c3_dir, max_len=200)]
    assert len(items) == 58
    hp = _hp(num_res_neighbours=30, padding_len=200, num_res_mpnn_layers=3)
    model, sd = _model(hp, precision)
    batches = bucket_batches([it[1].shape[0] for it in items], 16, 2048, seed=4)
    assert sorted(sum(batches, [])) == list(range(58))
    n_checked = 0
    for packed, cu, max_len, idx in PackedLoader(items, batches, device="cuda:0"):
        logits = model.forward_packed(packed, cu, max_len).cpu()
        assert logits.shape[0] == sum(items[i][1].shape[0] for i in idx)
        if n_checked < 2:                                # oracle on two of the buckets
            T = max_len
            pc = np.zeros((len(idx), T, 7, 3), np.float32); pm = np.zeros((len(idx), T), np.float32)
            for r, i in enumerate(idx):
                n = items[i][1].shape[0]
                pc[r, :n] = items[i][1]; pm[r, :n] = 1
            ref = _oracle(hp, sd, pc, pm)
            flat = ref[torch.from_numpy(pm).bool()]
            err = float((logits - flat).abs().max())
            assert err < (1e-4 if precision == "f32" else bf16_tol(ref, pm)), err
            n_checked += 1
    with pytest.raises(ValueError):                      # out-of-contract cu_seqlens are rejected on the host
        model.forward_packed(torch.zeros(10, 7, 3), torch.tensor([0, 4, 3, 10], dtype=torch.int32), 8)
    with pytest.raises(ValueError):
        model.forward_packed(torch.zeros(10, 7, 3), torch.tensor([0, 9, 10], dtype=torch.int32), 8)


# ----------------------------------------------------------------------------------------------- longest RNA of the data set
@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_longest_length_of_the_reference_data_4417_nt(precision):
    """F4 (long-RNA half): the longest RNA of data/train_data.csv has 4,417 nt and the reference pads attention to 4,500
    (functional.py:153-159).  A synthetic 4,417-nt backbone batched with a 50-nt one (T = 4,417, P = 4,500): k-NN with
    the 64-lane LDS row, attention in key chunks of 2,048 (three chunks), GraphNorm with T_tot = T and P - against the
    oracle run on the long RNA alone (same T, so the same normalisation)."""
    from rnampnn.utils import synth
    n = 4417
    coords, mask, _ = synth.synth_batch([n, 50], first_index=7000)
    hp = _hp(num_res_neighbours=30, padding_len=4500, num_res_mpnn_layers=4)
    model, sd = _model(hp, precision)
    lg = model(torch.from_numpy(coords), torch.from_numpy(mask)).cpu()
    assert torch.isfinite(lg).all() and (lg[1, 50:] == 0).all()
    ref = _oracle(hp, sd, coords[:1], mask[:1])
    err = float((lg[:1] - ref).abs().max())
    assert err < (1e-4 if precision == "f32" else bf16_tol(ref, mask[:1])), err
    with pytest.raises(RuntimeError):                      # beyond padding_len: the reference's negative-size error (functional.py:155)
        big_c, big_m, _ = synth.synth_batch([4501], first_index=1)
        model(torch.from_numpy(big_c), torch.from_numpy(big_m))
