#!/bin/bash
# per-kernel time of the config-3-shaped training epoch leg of bench.py under rocprofv3 --kernel-trace.  usage: tools/kstats_epoch.sh <outdir>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; shift; mkdir -p $OUT
python3 $ROOT/__graft_entry__.py || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-build --steps 1 --warmup 0 --no-cpu-baseline --recovery-steps 0 --train-epoch 1 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:30]:
    print(f"{r['Name'][:70]:70s} n {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):5.1f}%")
print('sum ms', round(tot/1e6, 1))
PY
