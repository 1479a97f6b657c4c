#!/usr/bin/env python3
"""Benchmark of the RNA-MPNN forward hot path on MI355X (contract: see the task brief).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one pass of ``RNAMPNN.forward`` over one batch of synthetic k-NN RNA graphs that
is already resident in HBM.  At N = 1 the workload is BASELINE.json configs[1]:
256 RNAs, lengths ~ U[100,140] (mean 120), k = 30, default 10-layer stack, bf16 MFMA kernels.
With N ranks every rank runs its own batch of that shape (different RNAs; weak scaling, no
data-path collective: the forward shards by independent RNAs) and ``value`` is the whole-job
valid-nucleotide throughput: (sum over ranks of valid nt per step) * K / max-over-ranks time.

Extra objects on the JSON line:
  roofline     - the dominant kernel (fused ResMPNN edge kernel): algorithmic HBM bytes per
                 launch / its mean launch duration, timed live with HIP events on the launch
                 stream (rnampnn_profile_*), against the 8 TB/s HBM3E peak.
  cpu_baseline - the CPU oracle (a port of the reference's PyTorch-CPU path, kind "port")
                 timed on this box's host cores on a bounded sample (16 RNAs of the same
                 workload), rank 0 at N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "rna-mpnn_amd"))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MFMA_BF16_PEAK_TFLOPS = 2500.0 # dense bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=["c2", "c4"])
    ap.add_argument("--precision", default=os.environ.get("RNAMPNN_BENCH_PRECISION", "bf16"), choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=0, help="RNAs per step per rank (default: workload's)")
    ap.add_argument("--neighbours", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=16)
    return ap.parse_args()


def workload(args, rank):
    from rnampnn.utils import synth
    if args.workload == "c2":      # BASELINE.json configs[1]
        B = args.batch or 256
        lens = synth.synth_lengths(B, 100, 140, seed=0, first_index=rank * B)
        name = f"C2: batch={B} RNAs, n~U[100,140], k={args.neighbours}, default 10-layer RNAMPNN, P=T"
    else:                          # configs[3] shape: 200-nt RNAs
        B = args.batch or 1024
        lens = np.full(B, 200, dtype=np.int64)
        name = f"C4-shaped: batch={B} RNAs x 200 nt, k={args.neighbours}, default 10-layer RNAMPNN, P=T"
    coords, mask, labels = synth.synth_batch(lens, first_index=rank * B, seed=0)
    return name, lens, coords, mask, labels


def flops_per_nt(k, n, L=10, skip_dead=True, factored=False):
    """SURVEY.md section 8d model FLOPs (2*MACs) per nucleotide."""
    edge = 55808 + L * 262144 - (131072 if skip_dead else 0)
    if factored:   # executed: first Linear of each MLP acts on e only (P/Q precomputed per node)
        edge = 55808 + L * 2 * 2 * (128 * 128 * 2) - (2 * 128 * 128 * 2 if skip_dead else 0)
    node = 4365312 + 1024 * n
    if factored:
        node += (2 * L - 1) * 2 * 128 * 256
    return k * edge + node


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (not used by the driver): RNAMPNN_BENCH_BACKEND=gloo and RNAMPNN_BENCH_ONE_GPU=1 let the
    # N>1 path run with every rank on GPU 0 of a one-GPU box
    backend = os.environ.get("RNAMPNN_BENCH_BACKEND", "nccl")
    if os.environ.get("RNAMPNN_BENCH_ONE_GPU") == "1":
        local = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    red_dev = dev if backend == "nccl" else torch.device("cpu")     # where the two scalar reductions live

    import __graft_entry__ as g
    if world > 1:                       # one builder per node; the others load the finished library
        import torch.distributed as dist
        if local == 0:
            g.build()
        dist.barrier()
    g.build()
    from rnampnn.model.rnampnn import RNAMPNN, argmax_recovery
    from rnampnn.utils import synth

    name, lens, coords, mask, labels = workload(args, rank)
    T = int(mask.shape[1])
    k = args.neighbours
    hp = dict(num_res_neighbours=k, padding_len=T)
    model = RNAMPNN(precision=args.precision, **hp)
    sd = synth.closed_form_state_dict({kk: tuple(v.shape) for kk, v in model.state_dict().items()})
    model.load_state_dict({kk: torch.from_numpy(v) for kk, v in sd.items()})
    model = model.to(dev).eval()
    c = torch.from_numpy(coords).to(dev)
    m = torch.from_numpy(mask).to(dev)
    lab = torch.from_numpy(labels).to(dev)
    nt_rank = int(lens.sum())

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        logits = model(c, m)
    model.profile_enable(True, every=7)     # sampled: 7 is coprime to the 10 fused launches of a forward, so every layer gets timed
    model.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = model(c, m)
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = model.profile_read(reset=True)
    model.profile_enable(False)

    nt_total, t_max = float(nt_rank), elapsed
    if world > 1:
        import torch.distributed as dist
        buf = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(buf, op=dist.ReduceOp.MAX)
        t_max = float(buf[0])
        cnt = torch.tensor([nt_rank], device=red_dev, dtype=torch.float64)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        nt_total = float(cnt[0])
    value = nt_total * args.steps / t_max

    # recovery of the GPU path on its batch (synthetic labels, closed-form weights)
    _, correct, nvalid = argmax_recovery(logits, m, lab)
    rec_gpu = float(correct.sum()) / float(nvalid.sum())

    if rank == 0:
        n_mean = float(lens.mean())
        w = 2 if args.precision == "bf16" else 4
        # algorithmic HBM bytes of ONE launch of the fused ResMPNN edge kernel, per nucleotide:
        #   e read + e write (k*128*w each; the last layer's launch only reads) + neighbour index (k*4)
        #   + P|Q rows of both MLPs (2*256*4) + h read + h_pre write (2*128*4)
        L = 10
        bytes_mid = k * (2 * 128 * w + 4) + 2 * 256 * 4 + 2 * 128 * 4
        bytes_first = k * (128 * w + 4) + 256 * 4 + 2 * 128 * 4            # layer 1: message only, reads e once
        bytes_launch_nt = (bytes_first + (L - 1) * bytes_mid) / L           # mean over the L launches of a forward
        launch_ms = kern_ms / max(launches, 1)
        traffic = None                  # HBM bytes per launch from PMC counters, when a profile of this workload is committed
        try:
            prof = json.load(open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")))
            if args.workload == "c2" and args.precision == "bf16" and k == 30 and not args.batch:
                traffic = prof["traffic_bytes_per_launch"]
        except Exception:
            traffic = None
        achieved_gbs = bytes_launch_nt * nt_rank / (launch_ms * 1e-3) / 1e9 if launches else 0.0
        exec_flops_nt = flops_per_nt(k, n_mean, factored=True)
        model_flops_nt = flops_per_nt(k, n_mean)
        per_rank_rate = nt_rank * args.steps / elapsed
        out = {
            "metric": "nucleotides/sec (forward), k=30 RNA graphs",
            "value": value, "unit": "nucleotides/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": name, "max_len": T, "nucleotides_per_step_per_gpu": nt_rank,
                       "weights": "closed-form deterministic init (3,536,900 params)", "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "kernel": "fused ResMPNN edge kernel (k_mpnn_*)",
                         "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_note": "bytes per launch, rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, profiles/r01_pmc_traffic.json",
                         "limiter": "power: shader clock 1.84 GHz of 2.4 under MFMA + HBM/L2/LDS traffic; cycle count set by vector issue (matrix pipe 47 %, vector issue 54 %, TA 68 % busy) - see DESIGN.md section 4",
                         "launch_ms": launch_ms, "launches_timed": launches,
                         "algorithmic_bytes_per_nt_per_launch": bytes_launch_nt},
            "mfma": {"model_flops_per_nt": model_flops_nt, "executed_flops_per_nt": exec_flops_nt,
                     "model_tflops": model_flops_nt * per_rank_rate / 1e12,
                     "executed_tflops": exec_flops_nt * per_rank_rate / 1e12,
                     "peak_tflops": MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16" else MFMA_F32_PEAK_TFLOPS},
            "recovery": {"gpu_micro": rec_gpu},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, hp, sd, coords, mask, labels, lens, logits, out)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, hp, sd, coords, mask, labels, lens, gpu_logits, out):
    """Time the CPU oracle (port of the reference's PyTorch-CPU path) on the first ``cpu_sample`` RNAs."""
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
