#!/bin/bash
# shader clock of the fused kernel = GRBM_GUI_ACTIVE (cycles, per XCD) / kernel duration, for a variant library.
# usage: tools/pmc_clock.sh <outdir> [variant-name]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
[ -n "$2" ] && [ "$2" != default ] && export RNAMPNN_LIB=$ROOT/rna-mpnn_amd/csrc/variants/$2.so
python3 $ROOT/__graft_entry__.py || exit 1    # build OUTSIDE the profiler (a hipcc child of a profiled process is a forbidden exec hop)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-build --steps 2 --warmup 1 --no-cpu-baseline --train-epoch 0 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
cc = sorted(glob.glob("$OUT/*/*counter_collection.csv"))[-1]
kt = sorted(glob.glob("$OUT/*/*kernel_trace.csv"))[-1]
dur = {}
for r in csv.DictReader(open(kt)):
    if r["Kernel_Name"].startswith("void k_mpnn_bf16<true, true"):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
cyc = {}
for r in csv.DictReader(open(cc)):
    if r["Kernel_Name"].startswith("void k_mpnn_bf16<true, true") and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cyc[r["Dispatch_Id"]] = cyc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
ghz = [cyc[d] / 8 / dur[d] for d in cyc if d in dur]
print("${2:-default}", "launches", len(ghz), "mean duration us", round(sum(dur[d] for d in cyc if d in dur) / max(len(ghz), 1) / 1e3, 1), "mean shader clock GHz", round(sum(ghz) / max(len(ghz), 1), 3))
PY
