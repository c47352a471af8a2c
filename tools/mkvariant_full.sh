#!/bin/bash
# tools/mkvariant_full.sh NAME [-Dflag ...]: like mkvariant.sh but with every instantiation of hk_kernels.hip (a minute)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p _ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -DHK_FIT_ONE_TU "$@" \
    -c homonim_amd/csrc/hk_kernels.hip -o _ab/hk_kernels_$name.o
objs=$(ls homonim_amd/lib/*.o | grep -v "hk_kernels.o\|hk_fit_m")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/lib_$name.so _ab/hk_kernels_$name.o $objs
echo "built _ab/lib_$name.so"
