#!/bin/bash
# Variant of the library that differs only in ONE source's compile flags (default kernels_mpnn.hip; RN_VARIANT_SRC=kernels_bf16.hip for the others):
#   tools/build_mpnn_variant.sh <name> <-D...>
# (the other objects come from build/obj, i.e. run `python __graft_entry__.py` first) -> rna-mpnn_amd/csrc/variants/<name>.so ; select with RNAMPNN_LIB
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/rna-mpnn_amd/csrc
SRC=${RN_VARIANT_SRC:-kernels_mpnn.hip}
name=$1; shift
mkdir -p $C/variants $ROOT/build/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DRN_EXPERIMENTS "$@" -c $C/$SRC -o $ROOT/build/var/$name.o
objs=$(ls $ROOT/build/obj/*.o | grep -v ${SRC%.*}.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/variants/$name.so $objs $ROOT/build/var/$name.o
echo built $C/variants/$name.so
