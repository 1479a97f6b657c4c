"""Round-4 GPU tests: RCCL at world size 1 through the chunked side-stream gradient exchange (eager and hipGraph-captured producers),
and the loader -> trainer path after the epoch-plan change."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _small(precision="bf16", **kw):
    from rnampnn.model.rnampnn import RNAMPNN
    from rnampnn.utils import synth
    hp = dict(num_res_neighbours=30, num_res_mpnn_layers=3, padding_len=64)
    hp.update(kw)
    torch.manual_seed(5)
    model = RNAMPNN(precision=precision, **hp)
    sd = synth.closed_form_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model.to("cuda:0").train()


@pytest.fixture
def nccl_world1():
    """The `nccl` backend of torch.distributed IS RCCL on ROCm: one rank on the one GPU of the box."""
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_rccl_world1_chunked_side_stream_allreduce_equals_the_gradient(nccl_world1):
    """before the driver's 8-GPU run, RCCL itself executes once - `librccl` loads, `ProcessGroupNCCL` orders its
    internal stream against the library-recorded chunk events and the join before Adam, and the result of the chunked side-stream
    exchange is, at world size 1, the un-exchanged gradient bit for bit (sum over one rank, x 1.0).  Three producers: the eager
    backward (events fresh), a second exchange without a new backward (events spent: stream ordering), and a hipGraph replay of a
    captured step (no events recorded: stream ordering; ADVICE r3: a stale event must not release a chunk early).  Also prints the
    exposed all-reduce time of the eager step - the first on real RCCL (one rank: launch + kernel latency only, no xGMI traffic)."""
    from rnampnn.model.rnampnn import CapturedTrainStep
    from rnampnn.utils import synth
    dist = nccl_world1
    model = _small("bf16")
    c, m, y = (torch.from_numpy(x).cuda() for x in synth.synth_batch([24, 17, 30, 12], first_index=70))
    # reference gradient: no exchange at all
    model.loss_and_grad(y, c, m, seed=11)
    g_ref = model.flat_grad.clone()
    # flat (un-chunked) exchange on the caller's stream
    model.loss_and_grad(y, c, m, seed=11)
    model.allreduce_gradients(force=True)
    assert torch.equal(model.flat_grad, g_ref)
    # chunked exchange on the side stream, ordered by the events the backward records
    model.enable_allreduce_overlap(True)
    t = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    for _ in range(3):          # (first call: RCCL communicator set-up on the side stream)
        model.loss_and_grad(y, c, m, seed=11)
        assert model._ar["fresh"]
        model.allreduce_gradients(timing=t, force=True)
        assert not model._ar["fresh"]
        torch.cuda.synchronize()
        assert torch.equal(model.flat_grad, g_ref)
    print(f"allreduce_exposed_ms at world size 1 over RCCL: {t[0].elapsed_time(t[1]):.3f}")
    # events spent: a second exchange of the same buffer orders on the stream and still gives the same numbers
    model.allreduce_gradients(force=True)
    torch.cuda.synchronize()
    assert torch.equal(model.flat_grad, g_ref)
    # hipGraph-captured producer: the replay records no chunk events
    cap = CapturedTrainStep(model, 4, 30)
    for _ in range(3):
        cap(y, c, m, seed=11)
        assert not model._ar["fresh"]
        g_replay = None
        model.allreduce_gradients(force=True)
        torch.cuda.synchronize()
        g_replay = model.flat_grad.clone()
        assert torch.equal(g_replay, g_ref)
    # a barrier and a scalar all-reduce (the trainer's loss / recovery counters) over the same group
    s = torch.ones(3, device="cuda:0")
    dist.all_reduce(s)
    dist.barrier()
    assert s.tolist() == [1.0, 1.0, 1.0]


def test_trainer_epoch_sees_every_item_with_the_jittered_plan():
    """the epoch plan drops nothing and changes its batch composition from epoch to epoch; the trainer built on it still
    runs an epoch end to end (loader -> loss_and_grad -> fused Adam) and counts every nucleotide exactly once at world size 1."""
    from rnampnn.utils import synth
    from rnampnn.utils.train import Trainer, plan_epoch
    model = _small("bf16")
    lens = [int(n) for n in synth.synth_lengths(40, 10, 48, seed=4)]
    items = [(synth.synth_rna(n, 900 + i, seed=2), synth.synth_labels(n, 900 + i, seed=2)) for i, n in enumerate(lens)]
    p0, _ = plan_epoch(lens, 0, 1, 8, 512, seed=0)
    p1, _ = plan_epoch(lens, 0, 1, 8, 512, seed=1)
    assert sorted(sum(p0, [])) == list(range(40)) == sorted(sum(p1, []))
    assert {tuple(sorted(b)) for b in p0} != {tuple(sorted(b)) for b in p1}
    (opt,), (sched,) = model.configure_optimizers(fused=True)
    tr = Trainer(model, opt, sched, world=1, rank=0, seed=0)
    out = tr.run_epoch(items, lens, 0, 8, 512)
    assert out["nt"] == sum(lens) and np.isfinite(out["train_loss"])


def test_two_ranks_on_one_gpu_overlap_equals_flat_allreduce():
    """the overlapped (chunked, side-stream) gradient exchange with MORE THAN ONE rank, on GPU tensors, against the flat
    all-reduce - eager backward (chunk events) and hipGraph replay (no events).  Two ranks share the one GPU of the box, so the
    transport is gloo; RCCL itself runs in the world-size-1 test above.  RCCL overlap across GPUs stays unverified on hardware until
    a SCALE record with N > 1 exists."""
    import subprocess, sys
    from conftest import REPO
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(REPO, "tests", "_ddp_overlap_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DDP_OVERLAP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_embed_fused_first_launch_is_bit_identical_to_the_two_launch_form(monkeypatch):
    """at k > 16 the edge embedding (feature.py:386-571) runs inside layer 1's message launch (k_resmpnn<false, true, true>:
    e0 goes from the embedding's accumulators to the message MLP's operand registers and is stored once); RNAMPNN_EMBED_FUSED=0 runs
    k_edge_embed_bf16 + the plain message launch.  Same arithmetic in the same order: logits, the layer-1 node tap and the layer-1 edge tap
    (which re-reads the stored e0) are equal bit for bit on a ragged batch with a 1-residue and a 64-residue (= padding_len) RNA."""
    from rnampnn.utils import synth
    model = _small("bf16", num_res_mpnn_layers=4).eval()
    coords, mask, _ = synth.synth_batch([64, 20, 47, 33, 5, 58, 31, 1], first_index=77)
    c, m = torch.from_numpy(coords), torch.from_numpy(mask)
    model.profile_enable(True)
    fused = model(c, m).clone()
    n_fused = model.profile_read()[1]
    taps_f = {k: v.clone() for k, v in model.forward_taps(c, m, ["h_layer", "e_layer"], tap_layer=1).items() if torch.is_tensor(v)}
    monkeypatch.setenv("RNAMPNN_EMBED_FUSED", "0")
    two = model(c, m).clone()
    taps_t = {k: v.clone() for k, v in model.forward_taps(c, m, ["h_layer", "e_layer"], tap_layer=1).items() if torch.is_tensor(v)}
    monkeypatch.delenv("RNAMPNN_EMBED_FUSED")
    model.profile_enable(False)
    assert n_fused == 4                                     # (one timed fused-step launch per layer either way)
    assert torch.isfinite(fused).all() and torch.equal(fused, two)
    assert set(taps_f) == set(taps_t) and {"h_layer", "e_layer"} <= set(taps_f)
    for name in taps_f:
        assert torch.equal(taps_f[name], taps_t[name]), name


def test_node_update_with_in_kernel_graphnorm_statistics_matches_the_two_launch_form(monkeypatch):
    """Between two ResMPNN steps the residual + GraphNormalization (functional.py:33-46) + [P | Q] projections run as ONE kernel per RNA
    when the padded length is 48..256 (k_node_update_rna: the per-RNA statistics from a fixed-order lane tree, two passes);
    RNAMPNN_NODE_UPDATE_RNA=0 runs k_gn_coef + k_node_update.  Same arithmetic up to the f32 summation order of the statistics:
    layer taps and logits agree to 2e-3 of their scale; padded lengths 64 (4 row blocks, two waves each), 150 (5 row blocks, two waves each:
    the C2 shape) and 200 (8 row blocks, one wave each), RNAs of 1 and T residues."""
    from rnampnn.utils import synth
    for T, lens in ((64, [64, 20, 47, 33, 5, 58, 31, 1]), (150, [150, 97, 1, 129, 33, 140]), (200, [200, 161, 1, 77])):
        model = _small("bf16", num_res_mpnn_layers=3, padding_len=T).eval()
        coords, mask, _ = synth.synth_batch(lens, first_index=11)
        c, m = torch.from_numpy(coords), torch.from_numpy(mask)
        new = model(c, m).clone()
        taps_n = {k: v.clone() for k, v in model.forward_taps(c, m, ["h_layer", "e_layer"], tap_layer=2).items() if torch.is_tensor(v)}
        monkeypatch.setenv("RNAMPNN_NODE_UPDATE_RNA", "0")
        old = model(c, m).clone()
        taps_o = {k: v.clone() for k, v in model.forward_taps(c, m, ["h_layer", "e_layer"], tap_layer=2).items() if torch.is_tensor(v)}
        monkeypatch.delenv("RNAMPNN_NODE_UPDATE_RNA")
        assert torch.isfinite(new).all() and (new[mask == 0] == 0).all()
        assert not torch.equal(new, old)                          # (a different kernel really ran)
        for a, b in [(new, old)] + [(taps_n[k], taps_o[k]) for k in ("h_layer", "e_layer")]:
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-3 * scale, (T, float((a - b).abs().max()), scale)


def test_batched_reductions_give_the_gradients_of_the_launch_per_reduction_form(monkeypatch):
    """The backward's ordered reductions (weight / bias / GraphNorm-parameter gradients from split partial tiles) are recorded and run up to 48 per
    launch (kernels_train.hip: k_reduce_batch); RNAMPNN_NO_RED_BATCH=1 launches each one on its own as before.  Same partials, another fixed
    association order: same loss bit for bit, every gradient equal to f32 rounding; both forms are bit-reproducible run to run.  bf16-mixed and
    f32 trainers, dropout on."""
    from rnampnn.utils import synth
    coords, mask, labels = synth.synth_batch([40, 33, 21, 48, 7, 64], first_index=300)
    c, m, y = (torch.from_numpy(a) for a in (coords, mask, labels))
    for prec in ("bf16", "f32"):
        model = _small(prec, num_res_mpnn_layers=3)
        l_b = float(model.loss_and_grad(y, c, m, seed=9)); g_b = model.flat_grad.clone()
        l_b2 = float(model.loss_and_grad(y, c, m, seed=9)); g_b2 = model.flat_grad.clone()
        monkeypatch.setenv("RNAMPNN_NO_RED_BATCH", "1")
        l_s = float(model.loss_and_grad(y, c, m, seed=9)); g_s = model.flat_grad.clone()
        monkeypatch.delenv("RNAMPNN_NO_RED_BATCH")
        assert l_b == l_b2 and torch.equal(g_b, g_b2)                      # bit-reproducible
        assert l_b == l_s and torch.isfinite(g_b).all()
        rel = float((g_b - g_s).norm() / g_s.norm())
        assert rel < 1e-5, (prec, rel)
        assert float((g_b - g_s).abs().max()) <= 1e-4 * float(g_s.abs().max()), prec
