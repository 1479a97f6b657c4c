#!/bin/bash
# memory-side PMC passes over bench.py.  usage: tools/pmc_mem.sh <outdir>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
python3 $ROOT/__graft_entry__.py || exit 1    # build OUTSIDE the profiler (a hipcc child of a profiled process is a forbidden exec hop)
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --no-build --steps 2 --warmup 1 --no-cpu-baseline --train-epoch 0 > $OUT/$name.log 2>&1; echo "$name exit $?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run grbm GRBM_GUI_ACTIVE
