#!/bin/bash
# PMC passes over two training steps (tools/train_probe.py at the C2 batch).  usage: tools/pmc_train.sh <outdir>
# --pmc only (no trace domains beside it); one counter group per run.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT
python3 $ROOT/__graft_entry__.py || exit 1
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/train_probe.py 256 bf16 1 > $OUT/$name.log 2>&1; echo "$name exit $?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sqA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
run grbm GRBM_GUI_ACTIVE
for k in "k_emm_fwd2<false, true>" "k_emm_fwd2<true, true>" "k_emm_bwd2<1>" "k_emm_bwd2<2>" "k_emm_bwd1x2" "k_eseg_mean(" "k_epq_bwd" "k_attn_fwd_m16" "k_attn_bwd_q_m16" "k_attn_bwd_kv_m16"; do
  echo "== $k"; python3 $ROOT/tools/pmc_summary.py $OUT "$k"
done
