"""CPU test of bench.py's own N > 1 launcher: ``--gpus 2`` without RANK spawns two ranks (torch.distributed.run
children, gloo), which shard the config-4 workload, reduce time (MAX) and units (SUM) and all-reduce the flat gradient
buffer of the training leg.  ``--dry-run`` launches no kernel (there is no GPU here); what is checked is the plumbing the
driver relies on: n_gpus, the workload named in config, the JSON contract."""
import json
import os
import subprocess
import sys

from conftest import REPO


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, cwd=REPO,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_spawns_its_own_ranks_and_reports_n_gpus():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--batch", "64", "--no-build"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["parallelism"] == "dp2" and out["config"]["workload"].startswith("C4: 64 RNAs x 200 nt")
    assert out["config"]["nucleotides_per_step_per_gpu"] == 32 * 200          # whole RNAs, balanced
    assert out["scaling"] == "strong" and out["unit"] == "nucleotides/s" and out["higher_is_better"] is True
    assert out["vs_baseline"] is None and "cpu_baseline" not in out
    assert out["train"]["allreduce_bytes"] == 3536900 * 4 and out["train"]["steps"] == 5
    assert "allreduce_exposed_ms" in out["train"] and sum(out["train"]["allreduce_chunks_floats"]) == 3536900
    # the weak-scaling leg: the N = 1 workload (C2, 256 RNAs) on every rank, comparable with the N = 1 line
    w = out["weak_c2"]
    assert w["scaling"] == "weak" and w["workload"].startswith("C2: batch=256 RNAs") and w["steps"] == 2
    assert w["nucleotides_per_step_all_ranks"] > 2 * 256 * 100 and w["value"] > 0 and w["unit"] == "nucleotides/s"
    for key in ("metric", "value", "ms_per_step", "dtype", "data", "roofline"):
        assert key in out


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _run(["--gpus", "2", "--dry-run", "--no-build"], env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
