#!/bin/bash
# SQ-only PMC passes (+GRBM clock proxy) over bench.py.  usage: tools/pmc_sq.sh <outdir>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
python3 $ROOT/__graft_entry__.py || exit 1    # build OUTSIDE the profiler (a hipcc child of a profiled process is a forbidden exec hop)
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py --no-build --steps 2 --warmup 1 --no-cpu-baseline --train-epoch 0 > $OUT/$name.log 2>&1; echo "$name exit $?"; }
run sqA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sqB SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32
run grbm GRBM_GUI_ACTIVE
