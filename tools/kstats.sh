#!/bin/bash
# per-kernel time of one bench step under rocprofv3 --kernel-trace.  usage: tools/kstats.sh <outdir> [bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; shift; mkdir -p $OUT
python3 $ROOT/__graft_entry__.py || exit 1    # build OUTSIDE the profiler (a hipcc child of a profiled process is a forbidden exec hop)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-build --steps 10 --warmup 3 --no-cpu-baseline --train-epoch 0 "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/*/*kernel_stats.csv"))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:18]:
    print(f"{r['Name'][:64]:64s} n {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us  /step {float(r['TotalDurationNs'])/13/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
print('sum per step us', round(tot/13/1e3, 1))
PY
