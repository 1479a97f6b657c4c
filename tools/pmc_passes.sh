#!/bin/bash
# PMC passes over bench.py on the GPU box (each counter group in its own rocprofv3 run).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [bench args...]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p $OUT
python3 $ROOT/__graft_entry__.py || exit 1    # build OUTSIDE the profiler (a hipcc child of a profiled process is a forbidden exec hop)
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --no-build --steps 2 --warmup 1 --no-cpu-baseline --train-epoch 0 "${EXTRA[@]}" > $OUT/$name.log 2>&1
  echo "$name exit $?"
}
EXTRA=("$@")
run sqA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run sqB SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT
run tcp TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
