#!/bin/bash
# per-kernel average of one kernel across variant libraries: tools/ab_kstat.sh <kernel-substring> <name>...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
pat=$1; shift
for n in "$@"; do
  if [ "$n" = default ]; then unset RNAMPNN_LIB; else export RNAMPNN_LIB=$ROOT/rna-mpnn_amd/csrc/variants/$n.so; fi
  echo "== $n"; bash $ROOT/tools/kstats.sh ab_$n 2>&1 | grep -i "$pat"
done
