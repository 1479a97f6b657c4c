#!/bin/bash
# A/B of library variants on one box, interleaved rounds: tools/ab_mpnn.sh <rounds> <name>...   ("default" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rounds=$1; shift
for i in $(seq $rounds); do
for n in "$@"; do
  if [ "$n" = default ]; then unset RNAMPNN_LIB; else export RNAMPNN_LIB=$ROOT/rna-mpnn_amd/csrc/variants/$n.so; fi
  python $ROOT/bench.py --no-build --steps 20 --warmup 5 --no-cpu-baseline --train-epoch 0 --recovery-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['launch_ms']*1e3,1), 'us')"
done; done
