#!/bin/bash
# Build an experimental variant of the HIP library next to the default one: tools/build_variant.sh <name> <extra hipcc flags...>
# -> rna-mpnn_amd/csrc/variants/<name>.so ; select it with RNAMPNN_LIB=<path>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/rna-mpnn_amd/csrc
name=$1; shift
mkdir -p $C/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" -I$ROOT/include -I$C \
    -o $C/variants/$name.so $C/api.cpp $C/kernels_f32.hip $C/kernels_bf16.hip $C/kernels_mpnn.hip $C/kernels_train.hip $C/rdesign.hip $C/gbdt.hip
echo built $C/variants/$name.so
