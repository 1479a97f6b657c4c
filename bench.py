#!/usr/bin/env python3
"""Benchmark of the RNA-MPNN forward hot path on MI355X (contract: see the task brief).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of ``RNAMPNN.forward`` over one batch of synthetic k-NN RNA graphs that is already
resident in HBM.

* N = 1: BASELINE.json configs[1] ("C2"): 256 RNAs, lengths ~ U[100,140] (mean 120), k = 30, default
  10-layer stack, bf16 MFMA kernels.
* N > 1: BASELINE.json configs[3] ("C4"): 10,000 synthetic RNAs x 200 nt, k = 30, split over the ranks as whole
  RNAs (``rnampnn.utils.shard.balanced_shards``); a step is one forward pass of every rank over its shard (in
  micro-batches of <= 2,500 RNAs): FIXED total work, ``"scaling": "strong"``.  The forward needs NO data-path
  collective (RNAs are independent units); the one exchange of the path is the gradient all-reduce of a training
  step, measured as a separate leg (``"train"``: taped forward + HIP backward + the flat 14.15 MB gradient
  all-reduced over RCCL in three chunks on a side stream under the tail of the backward + Adam, as Lightning DDP's
  bucketed overlap does for the reference, rnampnn/utils/train.py:106-117; ``allreduce_exposed_ms`` = the part of it
  the backward did not hide).
  ``value`` = (valid nucleotides all ranks pushed through forward) * K / max-over-ranks time.
  ``weak_c2`` (N > 1 only): every rank also runs the N = 1 workload (its own C2 batch of 256 RNAs) for K steps;
  ``weak_c2.value`` = sum of nucleotides * K / max time is directly comparable with the N = 1 line of this script:
  weak-scaling efficiency = weak_c2.value / (N x the N = 1 value).

Launch: with ``--gpus N > 1`` and no RANK in the environment this script spawns its own N ranks
(``python -m torch.distributed.run --nproc-per-node N ... bench.py``, as CHILD processes, before anything touches
the GPU); under an external ``torch.distributed.run`` it reads RANK / LOCAL_RANK / WORLD_SIZE and checks
WORLD_SIZE == --gpus.  One process per GPU, RCCL ("nccl") for the collectives.

Extra objects on the JSON line:
  roofline     - the dominant kernel (fused ResMPNN edge kernel): algorithmic HBM bytes per launch / its mean launch
                 duration, timed live with HIP events on the launch stream (rnampnn_profile_*), vs 8 TB/s.
  cpu_baseline - the CPU oracle (a port of the reference's PyTorch-CPU path, kind "port") timed on this box's host
                 cores on a bounded sample (16 RNAs of the same workload), rank 0 at N = 1 only.
  train        - the training-step leg (N > 1 by default, ``--train-steps`` elsewhere).
  train_epoch  - BASELINE.json configs[2] (N = 1): one epoch of the trainer (``rnampnn.utils.train.Trainer``: PaddedLoader,
                 no host syncs) over seeded synthetic RNAs of exactly the 2,083 lengths of the reference's
                 data/train_data.csv (tests/data/c3_train_lengths.npy: 1 ... 4,417 nt, 1,205,038 nt, P = 4,500): epoch
                 seconds, end-to-end nt/s, and the same batch plan with resident inputs (kernel-only) beside it.
  recovery     - closed-form weights (flat logits: chance) AND weights trained here for ``--recovery-steps`` seeded steps
                 (bf16-mixed, dropout 0.4) so the logits separate: f32 CPU oracle vs bf16 HIP on the CPU sample.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # dense bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3
C4_RNAS, C4_LEN = 10000, 200
C4_MICRO = int(os.environ.get("RNAMPNN_C4_MICRO", "2500"))      # RNAs per forward of a rank's share (tuning knob)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["c2", "c4"], help="default: c2 at N = 1, c4 at N > 1")
    ap.add_argument("--precision", default=os.environ.get("RNAMPNN_BENCH_PRECISION", "bf16"), choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=0, help="c2: RNAs per step per rank; c4: total RNAs of the job")
    ap.add_argument("--neighbours", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=16)
    ap.add_argument("--train-steps", type=int, default=-1, help="training-step leg (default: 5 at N > 1, 0 at N = 1)")
    ap.add_argument("--train-epoch", type=int, default=-1, help="config-3 epoch leg: 1 on, 0 off (default: on at N = 1 with the default workload)")
    ap.add_argument("--recovery-steps", type=int, default=-1, help="training steps before the trained-weights recovery check (default 300 at N = 1; 0 = off)")
    ap.add_argument("--train-batch", type=int, default=64, help="RNAs per rank per training step")
    ap.add_argument("--no-build", action="store_true",
                    help="load the prebuilt library only (required under rocprofv3: a hipcc child of a profiled process is a forbidden exec hop)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / reduction plumbing only: no model, no kernels, no GPU (CPU test of the N > 1 path)")
    return ap.parse_args(argv)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args) -> int:
    """--gpus N > 1 without RANK: start N ranks as children of this (GPU-untouched) process and relay their output."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def flops_per_nt(k, n, L=10, skip_dead=True, factored=False):
    """SURVEY.md section 8d model FLOPs (2*MACs) per nucleotide."""
    edge = 55808 + L * 262144 - (131072 if skip_dead else 0)
    if factored:   # executed: first Linear of each MLP acts on e only (P/Q precomputed per node)
        edge = 55808 + L * 2 * 2 * (128 * 128 * 2) - (2 * 128 * 128 * 2 if skip_dead else 0)
    node = 4365312 + 1024 * n
    if factored:
        node += (2 * L - 1) * 2 * 128 * 256
    return k * edge + node


def build_workload(args, rank, world):
    """-> (name, list of (coords, mask, labels) numpy micro-batches of this rank, lengths of this rank's RNAs)."""
    import numpy as np
    from rnampnn.utils import shard, synth
    k = args.neighbours
    if args.workload == "c2":      # BASELINE.json configs[1]; with N ranks: each its own batch of that shape
        B = args.batch or 256
        lens = synth.synth_lengths(B, 100, 140, seed=0, first_index=rank * B)
        name = f"C2: batch={B} RNAs, n~U[100,140], k={k}, default 10-layer RNAMPNN, P=T"
        return name, [synth.synth_batch(lens, first_index=rank * B, seed=0)], lens, "weak"
    total = args.batch or C4_RNAS   # BASELINE.json configs[3]: fixed job, split over the ranks
    all_lens = [C4_LEN] * total
    mine = shard.balanced_shards(all_lens, world)[rank]
    name = (f"C4: {total} RNAs x {C4_LEN} nt, k={k}, default 10-layer RNAMPNN, P=T, whole RNAs sharded over {world} rank(s) "
            f"(balanced_shards), micro-batches of <= {C4_MICRO} RNAs")
    batches = []
    for i in range(0, len(mine), C4_MICRO):
        ids = mine[i:i + C4_MICRO]
        coords = np.stack([synth.synth_rna(C4_LEN, j, 0) for j in ids]).astype(np.float32)
        labels = np.stack([synth.synth_labels(C4_LEN, j, 0) for j in ids])
        batches.append((coords, np.ones((len(ids), C4_LEN), np.float32), labels))
    return name, batches, np.full(len(mine), C4_LEN, dtype=np.int64), "strong"


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if args.workload is None:
        args.workload = "c2" if world == 1 else "c4"
    # rehearsal knobs (not used by the driver): RNAMPNN_BENCH_BACKEND=gloo and RNAMPNN_BENCH_ONE_GPU=1 let the
    # N>1 path run with every rank on GPU 0 of a one-GPU box
    backend = os.environ.get("RNAMPNN_BENCH_BACKEND", "nccl")
    one_gpu = os.environ.get("RNAMPNN_BENCH_ONE_GPU") == "1"
    builder = (rank == 0) if one_gpu else (local == 0)     # ONE process per node compiles a stale library (decided before `local` is overridden)
    if one_gpu:
        local = 0
    dry = args.dry_run
    if dry:
        backend = "gloo"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    red_dev = dev if backend == "nccl" else torch.device("cpu")     # where the scalar reductions live

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        if not dry:
            torch.cuda.synchronize(dev)

    import __graft_entry__ as g
    no_build = args.no_build or os.environ.get("RNAMPNN_NO_BUILD") == "1"
    if world > 1 and not no_build:      # one builder per node; the others load the finished library
        if builder:
            g.build()
        import torch.distributed as dist
        dist.barrier()
    g.load_only() if (no_build or world > 1) else g.build()
    from rnampnn.utils import synth, shard

    name, batches, lens, scaling = build_workload(args, rank, world)
    T = int(batches[0][1].shape[1])
    k = args.neighbours
    hp = dict(num_res_neighbours=k, padding_len=T)
    nt_rank = int(lens.sum())
    model = None
    if not dry:
        from rnampnn.model.rnampnn import RNAMPNN, argmax_recovery
        model = RNAMPNN(precision=args.precision, **hp)
        sd = synth.closed_form_state_dict({kk: tuple(v.shape) for kk, v in model.state_dict().items()})
        model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()})
        model = model.to(dev).eval()
    dev_batches = [(torch.from_numpy(c).to(dev), torch.from_numpy(m).to(dev), torch.from_numpy(y).to(dev)) for c, m, y in batches]

    def run_forward(batch_list, steps, warmup, profile):
        """W untimed + K timed passes over the resident batches -> (elapsed seconds of this rank, last logits, timed fused launches by kind)."""
        def one():
            o = None
            for c, m, _ in batch_list:
                o = model(c, m) if model is not None else None
            return o
        o = None
        for _ in range(warmup):
            o = one()
        if model is not None and profile:
            model.profile_enable(True, every=7)     # sampled: 7 is coprime to the 10 fused launches of a forward, so every layer gets timed
            model.profile_read(reset=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            o = one()
        barrier()
        el = time.perf_counter() - t0
        km, nl = ((0.0, 0), (0.0, 0))
        if model is not None and profile:
            km, nl = model.profile_read_kinds(reset=True)      # ((ms, n) of the <edge, message> launches, (ms, n) of the message-only launch)
            model.profile_enable(False)
        return el, o, km, nl

    elapsed, logits, prof_em, prof_first = run_forward(dev_batches, args.steps, args.warmup, True)
    t_max, nt_total = shard.reduce_job(elapsed, float(nt_rank), red_dev)
    value = nt_total * args.steps / t_max

    # N > 1: the N = 1 workload on every rank (weak scaling), directly comparable with this script's N = 1 line
    weak = None
    if world > 1 and args.workload == "c4":
        wargs = argparse.Namespace(**vars(args))
        wargs.workload, wargs.batch = "c2", 0
        wname, wb, wlens, _ = build_workload(wargs, rank, world)
        w_dev = [(torch.from_numpy(c).to(dev), torch.from_numpy(m).to(dev), torch.from_numpy(y).to(dev)) for c, m, y in wb]
        wmodel = None
        if not dry:     # its own module: padding_len follows the C2 batch (P = T, as the N = 1 line)
            from rnampnn.model.rnampnn import RNAMPNN as _M
            wmodel = _M(precision=args.precision, num_res_neighbours=k, padding_len=int(wb[0][1].shape[1]))
            wmodel.load_state_dict(model.state_dict())
            wmodel = wmodel.to(dev).eval()
        keep, model = model, wmodel
        w_el, _, _, _ = run_forward(w_dev, args.steps, args.warmup, False)
        model = keep
        w_t, w_nt = shard.reduce_job(w_el, float(wlens.sum()), red_dev)
        weak = {"value": w_nt * args.steps / w_t, "unit": "nucleotides/s", "ms_per_step": w_t / args.steps * 1e3, "steps": args.steps,
                "scaling": "weak", "workload": wname + " on EVERY rank (its own RNAs)", "nucleotides_per_step_all_ranks": w_nt,
                "note": "compare with the N = 1 line of this script: efficiency = weak_c2.value / (n_gpus x value at N = 1)"}
        del w_dev, wmodel

    rec_gpu = None
    if model is not None:       # recovery of the GPU path on its last micro-batch (synthetic labels, closed-form weights)
        _, correct, nvalid = argmax_recovery(logits, dev_batches[-1][1], dev_batches[-1][2])
        rec_gpu = float(correct.sum()) / float(nvalid.sum())

    train = None
    n_train = args.train_steps if args.train_steps >= 0 else (5 if world > 1 else 0)
    if n_train > 0:
        train = train_leg(args, model, dev_batches, n_train, world, dev, red_dev, barrier, dry)

    if rank == 0:
        n_mean = float(lens.mean())
        w = 2 if args.precision == "bf16" else 4
        # Roofline of the dominant kernel = the <edge update of layer l, message of layer l + 1> launch of the fused ResMPNN kernel (L - 1 = 9 per
        # forward), timed live with HIP events on the launch stream, separately from the message-only launch of layer 1.
        #   achieved = SURVEY section 8(d)'s algorithmic bytes of one ResMPNN layer per nucleotide - e read + write 2 k 128 w, h read / write 4 * 128 w,
        #   neighbour index 2 * 4 k (16,624 B at k = 30, w = 2) - x the nucleotides of a launch / the launch's duration.
        #   kernel_bytes_per_nt = what THIS kernel moves per nucleotide at its stored widths: e read + write (f16), index once, its own P rows of both
        #   MLPs (f16), the gathered Q rows counted once per table row (f16, L2-resident), the residue's h (f32) and the h + mean row it writes (f32).
        L = 10
        survey_bytes_nt = k * (2 * 128 * w + 2 * 4) + 4 * 128 * w
        kernel_bytes_nt = k * (2 * 128 * w + 4) + 4 * 128 * 2 + 2 * 128 * 4
        # layer 1: message only - one P and one Q table; e0 is WRITTEN once by the same launch when the edge embedding is fused into it (k > 16,
        # default; + the residue's 256-byte geometry record), READ once in the two-launch form (RNAMPNN_EMBED_FUSED=0)
        embed_fused = args.precision == "bf16" and k > 16 and os.environ.get("RNAMPNN_EMBED_FUSED") != "0" and os.environ.get("RNAMPNN_MPNN_V3") != "1"
        first_bytes_nt = k * (128 * w + 4) + 2 * 128 * 2 + 2 * 128 * 4 + (256 if embed_fused else 0)
        first_name = ("edge embedding + message launch of layer 1 (k_resmpnn<false, true, true>)" if embed_fused
                      else "message-only launch of layer 1 (k_resmpnn<false, true>)")
        (ms_em, n_em), (ms_first, n_first) = prof_em, prof_first
        launch_ms = ms_em / max(n_em, 1)
        first_ms = ms_first / max(n_first, 1)
        nt_call = nt_rank / len(batches)                                    # nucleotides one launch works on
        measured_cfg = args.workload == "c2" and args.precision == "bf16" and k == 30 and not args.batch and world == 1
        traffic, traffic_src, first_traffic = None, None, None   # HBM bytes per launch from PMC counters, when a profile of this workload is committed
        for cand in ("r04_pmc_traffic.json", "r03_pmc_traffic.json"):
            try:
                prof = json.load(open(os.path.join(REPO, "profiles", cand)))
                if measured_cfg:
                    traffic, traffic_src = prof["traffic_bytes_per_launch"], cand
                    first_traffic = prof.get("first_launch_embed_fused", {}).get("traffic_bytes_per_launch")
                break
            except Exception:
                continue
        achieved_gbs = survey_bytes_nt * nt_call / (launch_ms * 1e-3) / 1e9 if n_em else 0.0
        exec_flops_nt = flops_per_nt(k, n_mean, factored=True)
        model_flops_nt = flops_per_nt(k, n_mean)
        per_rank_rate = nt_rank * args.steps / elapsed
        # matrix-pipe share of the same launch: one 32-slot block per residue (k > 16), 128 v_mfma_f32_32x32x16 per block (160 in the round-3
        # kernel, RNAMPNN_MPNN_V3=1: 32 helper MFMAs), 32,768 FLOP each; useful = the k real edge slots of the 32
        mfma_per_block = 160 if os.environ.get("RNAMPNN_MPNN_V3") == "1" else 128
        peak_tf = MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16" else MFMA_F32_PEAK_TFLOPS
        issued_tf = nt_call * mfma_per_block * 32768 / (launch_ms * 1e-3) / 1e12 if (n_em and args.precision == "bf16" and k > 16) else None
        roof = {"bound": "hbm", "kernel": "fused ResMPNN kernel, <edge update, message> launch (k_resmpnn<true, true>)",
                "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "launch_ms": launch_ms, "launches_timed": n_em,
                "algorithmic_bytes_per_nt_per_launch": survey_bytes_nt, "kernel_bytes_per_nt": kernel_bytes_nt,
                "nucleotides_per_launch": nt_call,
                "mfma_issued_frac": issued_tf / peak_tf if issued_tf else None,
                "mfma_useful_frac": issued_tf * min(k, 32) / 32 * 128 / mfma_per_block / peak_tf if issued_tf else None,
                "first_launch": {"kernel": first_name, "launch_ms": first_ms, "launches_timed": n_first,
                                 "algorithmic_bytes_per_nt": first_bytes_nt,
                                 "achieved": first_bytes_nt * nt_call / (first_ms * 1e-3) / 1e9 if n_first else None,
                                 "traffic": first_traffic if embed_fused else None}}
        if measured_cfg and traffic_src:    # measured on exactly this configuration (profiles/)
            roof["traffic_note"] = f"bytes per launch, rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, profiles/{traffic_src}"
        out = {
            "metric": "nucleotides/sec (forward), k=30 RNA graphs",
            "value": value, "unit": "nucleotides/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_max / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic" if not dry else "dry-run: no kernels launched (launcher test)",
            "config": {"workload": name, "max_len": T, "nucleotides_per_step_per_gpu": nt_rank,
                       "weights": "closed-form deterministic init (3,536,900 params)", "parallelism": f"dp{world}"},
            "roofline": roof,
            "mfma": {"model_flops_per_nt": model_flops_nt, "executed_flops_per_nt": exec_flops_nt,
                     "model_tflops": model_flops_nt * per_rank_rate / 1e12,
                     "executed_tflops": exec_flops_nt * per_rank_rate / 1e12,
                     "peak_tflops": MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16" else MFMA_F32_PEAK_TFLOPS},
            "recovery": {"gpu_micro": rec_gpu},
        }
        if train is not None:
            out["train"] = train
        if weak is not None:
            out["weak_c2"] = weak
        default_cfg = args.workload == "c2" and not args.batch and k == 30
        if world == 1 and not args.no_cpu_baseline and not dry:
            c0, m0, y0 = batches[0]
            out["cpu_baseline"] = cpu_baseline(args, hp, sd, c0, m0, y0, lens, logits, out)
            n_rec = args.recovery_steps if args.recovery_steps >= 0 else (300 if default_cfg and args.precision == "bf16" else 0)
            if n_rec > 0:
                out["recovery"]["trained"] = guarded(lambda: trained_recovery(args, hp, sd, c0, m0, y0, lens, dev, n_rec))
        want_epoch = args.train_epoch if args.train_epoch >= 0 else int(world == 1 and default_cfg and not dry and args.precision == "bf16")
        if world == 1 and want_epoch and not dry:
            del dev_batches[:]
            out["train_epoch"] = guarded(lambda: train_epoch_leg(args, dev))
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def guarded(fn):
    """Extra legs never take the headline line down with them: an exception becomes {"error": ...} in that leg."""
    try:
        return fn()
    except Exception as exc:    # noqa: BLE001
        import traceback
        return {"error": f"{type(exc).__name__}: {exc}", "traceback": traceback.format_exc()[-1500:]}


def train_leg(args, model, dev_batches, n_steps, world, dev, red_dev, barrier, dry):
    """Training step of the path on ``--train-batch`` RNAs per rank: taped forward + HIP backward -> the flat gradient all-reduced
    (RCCL) in the three chunks of ``RNAMPNN.grad_chunks`` on a side stream, each as soon as the backward has finished it -> fused
    Adam (reference optimiser, rnampnn.py:156-159).  Aggregate nt/s; ``allreduce_exposed_ms`` = time the step's stream waited
    for the side stream (the all-reduce the backward did not hide)."""
    import torch
    from rnampnn.utils import shard
    c, m, y = dev_batches[0]
    nb = min(args.train_batch, int(c.shape[0]))
    c, m, y = c[:nb].contiguous(), m[:nb].contiguous(), y[:nb].contiguous()
    nt = float(m.sum())
    exposed, chunks = [], None
    if dry:
        flat = torch.zeros(3536900)
        chunks = [(3000000, 3536900), (1500000, 3000000), (0, 1500000)]
        tr = None
    else:
        from rnampnn.model.rnampnn import RNAMPNN
        from rnampnn.utils.train import Trainer
        tr = RNAMPNN(precision=args.precision, **{kk: model.hparams[kk] for kk in ("num_res_neighbours", "padding_len")}).to(dev)
        tr.load_state_dict(model.state_dict())
        tr.train()                                  # dropout 0.4 active, as the reference trains
        (opt,), _ = tr.configure_optimizers(fused=True)
        trainer = Trainer(tr, opt, None, world=world, rank=int(os.environ.get("RANK", "0")), overlap_allreduce=True)
        chunks = tr.grad_chunks()
        y = y.to(torch.int32)

    def one(timed):
        if dry:
            if world > 1:
                shard.allreduce_mean_chunks(flat, chunks)
            return
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if (timed and world > 1) else None
        trainer.step(y, c, m, timing=ev)
        if ev is not None:
            exposed.append(ev)

    one(False)                              # warm-up (workspace, RCCL channels)
    one(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        one(True)
    barrier()
    el = time.perf_counter() - t0
    t_max, nt_total = shard.reduce_job(el, nt, red_dev)
    exp_ms = sum(a.elapsed_time(b) for a, b in exposed) / len(exposed) if exposed else (0.0 if world > 1 else None)
    return {"value": nt_total * n_steps / t_max, "unit": "nucleotides/s (training step: fwd + bwd + all-reduce + Adam)",
            "steps": n_steps, "ms_per_step": t_max / n_steps * 1e3, "rnas_per_rank_per_step": nb,
            "dtype": "bf16-mixed (MFMA GEMMs, f32 accumulate; per-edge tape and gradients stored as bf16)" if args.precision == "bf16" else "f32",
            "allreduce": "3 chunks in backward order on a side stream (events recorded by the HIP backward), joined before Adam",
            "allreduce_chunks_floats": [e - b for b, e in chunks] if chunks else None,
            "allreduce_exposed_ms": exp_ms, "allreduce_bytes": 3536900 * 4}


def train_epoch_leg(args, dev):
    """BASELINE.json configs[2] shape: ONE epoch of the trainer over seeded synthetic RNAs of exactly the lengths of the
    reference's data/train_data.csv (tests/data/c3_train_lengths.npy), bf16-mixed, dropout 0.4, fused Adam, P = 4,500,
    length-bucketed steps of <= 32,768 padded rows.  End to end = loader (pad into pinned memory + H2D on a side stream)
    + kernels; beside it the same batch plan with every batch resident in HBM (kernel-only)."""
    import numpy as np
    import torch
    from rnampnn.model.rnampnn import RNAMPNN
    from rnampnn.utils import synth
    from rnampnn.utils.data import pad_batch
    from rnampnn.utils.train import Trainer, plan_epoch
    lens = [int(n) for n in np.load(os.path.join(REPO, "tests", "data", "c3_train_lengths.npy"), allow_pickle=False)]
    t0 = time.perf_counter()
    items = [(synth.synth_rna(n, 100000 + i, seed=3), synth.synth_labels(n, 100000 + i, seed=3)) for i, n in enumerate(lens)]
    t_gen = time.perf_counter() - t0
    model = RNAMPNN(precision="bf16", num_res_neighbours=args.neighbours, padding_len=4500).to(dev)
    sd = synth.closed_form_state_dict({kk: tuple(v.shape) for kk, v in model.state_dict().items()})
    model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()})
    model.train_precision = args.precision
    (opt,), (sched,) = model.configure_optimizers(fused=True)
    tr = Trainer(model, opt, sched, world=1, rank=0, seed=0)
    bs, rows = 512, 32768
    warm = tr.run_epoch(items, lens, 0, bs, rows)          # first epoch: workspace growth, one-time kernel attributes
    rec = tr.run_epoch(items, lens, 1, bs, rows)
    # kernel-only: the same plan of epoch 1 with resident inputs
    plan, _ = plan_epoch(lens, 0, 1, bs, rows, 1)
    res = []
    for b in plan:
        y, c, m, _ = pad_batch([items[i] for i in b], pin=False)
        res.append((y.to(dev), c.to(dev), m.to(dev)))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for it, (y, c, m) in enumerate(res):
        tr.step(y, c, m, seed=1000 + it)
    torch.cuda.synchronize(dev)
    t_res = time.perf_counter() - t0
    nt = int(sum(lens))
    return {"workload": f"C3: {len(lens)} synthetic RNAs with the lengths of the reference's data/train_data.csv ({nt} nt, {min(lens)}..{max(lens)} nt), "
                        f"k={args.neighbours}, P=4500, dropout 0.4, bf16-mixed, fused Adam, steps of <= {bs} RNAs / {rows} padded rows",
            "epoch_seconds": rec["seconds"], "value": rec["nt_per_s"], "unit": "nucleotides/s (end to end: loader + fwd + bwd + Adam)",
            "steps": rec["steps"], "train_loss": rec["train_loss"], "first_epoch_seconds": warm["seconds"],
            "first_epoch_train_loss": warm["train_loss"],
            "kernel_only_seconds": t_res, "kernel_only_nt_per_s": nt / t_res, "end_to_end_over_kernel_only": t_res / rec["seconds"],
            "synthesis_seconds_untimed": t_gen}


def trained_recovery(args, hp, sd, coords, mask, labels, lens, dev, n_steps):
    """Matched recovery on SEPARATED logits: train ``n_steps`` seeded steps (bf16-mixed, dropout 0.4, Adam as the reference) on the
    first 64 RNAs of the workload, then score the first ``--cpu-sample`` of them with the f32 CPU oracle and with the bf16 HIP
    path on the same weights.  Outside every timed region."""
    import numpy as np
    import torch
    from oracle import rnampnn_oracle as O
    from rnampnn.model._schema import DEFAULT_HPARAMS
    from rnampnn.model.rnampnn import RNAMPNN
    NB = min(64, len(lens))
    T = int(lens[:NB].max())
    c = torch.from_numpy(coords[:NB, :T].copy()).to(dev)
    m = torch.from_numpy(mask[:NB, :T].copy()).to(dev)
    y = torch.from_numpy(labels[:NB, :T].copy()).to(dev).to(torch.int32)
    tr = RNAMPNN(precision="bf16", **dict(hp, padding_len=T))
    tr.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()})
    tr = tr.to(dev).train()
    (opt,), _ = tr.configure_optimizers(fused=True)
    losses = []
    for it in range(n_steps):
        loss = tr.loss_and_grad(y, c, m, seed=7000 + it)
        opt.step()
        if it in (0, n_steps - 1):
            losses.append(float(loss))
    sd_t = {kk: v.detach().cpu().numpy() for kk, v in tr.state_dict().items()}
    S = min(args.cpu_sample, NB)
    Ts = int(lens[:S].max())
    full = dict(DEFAULT_HPARAMS, **hp)
    cfg = O.OracleConfig(**{kk: v for kk, v in full.items() if kk in O.OracleConfig.__dataclass_fields__})
    cfg.padding_len = Ts
    cs, ms = torch.from_numpy(coords[:S, :Ts].copy()), torch.from_numpy(mask[:S, :Ts].copy())
    lab = torch.from_numpy(labels[:S, :Ts].copy())
    with torch.no_grad():
        ref, _ = O.forward(cs, ms, O.state_dict_from_numpy(sd_t), cfg)
    ev = RNAMPNN(precision="bf16", **dict(hp, padding_len=Ts))
    ev.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd_t.items()})
    ev = ev.to(dev).eval()
    lg = ev(cs, ms).cpu()
    valid = ms.bool()
    rec_ref, _, _ = O.recovery(ref, ms, lab)
    rec_hip, _, _ = O.recovery(lg, ms, lab)
    return {"train_steps": n_steps, "train_rnas": NB, "train_dropout": float(full["dropout"]), "loss_first": losses[0], "loss_last": losses[-1],
            "sample_rnas": S, "logit_std": float(ref[valid].std()), "max_abs_dlogit": float((lg - ref).abs().max()),
            "recovery_f32_oracle": rec_ref, "recovery_bf16_hip": rec_hip,
            "argmax_agreement": float((lg.argmax(-1) == ref.argmax(-1))[valid].float().mean())}


def cpu_baseline(args, hp, sd, coords, mask, labels, lens, gpu_logits, out):
    """Time the CPU oracle (port of the reference's PyTorch-CPU path) on the first ``cpu_sample`` RNAs."""
    import numpy as np
    import torch
    from oracle import rnampnn_oracle as O
    from rnampnn.model._schema import DEFAULT_HPARAMS
    S = min(args.cpu_sample, len(lens))
    Ts = int(lens[:S].max())
    full = dict(DEFAULT_HPARAMS, **hp)
    cfg = O.OracleConfig(**{kk: v for kk, v in full.items() if kk in O.OracleConfig.__dataclass_fields__})
    cfg.padding_len = Ts
    # host cores this process may use: the 1-GPU box exposes its whole host (256 logical CPUs) but the
    # job's share is 16; more intra-op threads than that only add synchronisation overhead
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("RNAMPNN_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    cs = torch.from_numpy(coords[:S, :Ts].copy())
    ms = torch.from_numpy(mask[:S, :Ts].copy())
    osd = O.state_dict_from_numpy(sd)
    with torch.no_grad():
        O.forward(cs, ms, osd, cfg)                       # warm-up
        times = []
        t_budget = time.perf_counter() + 20.0
        while len(times) < 5 and (len(times) < 2 or time.perf_counter() < t_budget):
            t0 = time.perf_counter()
            ref, _ = O.forward(cs, ms, osd, cfg)
            times.append(time.perf_counter() - t0)
    nt = int(lens[:S].sum())
    rate = nt / float(np.median(times))
    lab = torch.from_numpy(labels[:S, :Ts])
    micro, _, _ = O.recovery(ref, ms, lab)
    out["recovery"]["cpu_sample_micro"] = micro
    # same RNAs through the GPU path, with the sample's own padding (T_norm) for a like-for-like check
    from rnampnn.model.rnampnn import RNAMPNN, argmax_recovery
    dev = gpu_logits.device
    model = RNAMPNN(precision=args.precision, **dict(hp, padding_len=Ts))
    model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()})
    model = model.to(dev).eval()
    lg = model(cs, ms)
    _, correct, nvalid = argmax_recovery(lg, ms.to(dev), lab.to(dev))
    out["recovery"]["gpu_sample_micro"] = float(correct.sum()) / float(nvalid.sum())
    out["recovery"]["max_abs_dlogit_gpu_vs_cpu_sample"] = float((lg.cpu() - ref).abs().max())
    return {"value": rate, "unit": "nucleotides/s", "cores": cores, "kind": "port",
            "sample": f"first {S} RNAs of the workload ({nt} nt, padded to {Ts}), f32 PyTorch-CPU oracle, "
                      f"median of {len(times)} forwards", "speedup_gpu_over_cpu": out["value"] / rate}


if __name__ == "__main__":
    main()
