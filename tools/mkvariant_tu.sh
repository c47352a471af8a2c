#!/bin/bash
# tools/mkvariant_tu.sh NAME MODEL R2 [-Dflag ...]: _ab/lib_NAME.so = the current objects with ONE translation unit of the fused kernel
# (homonim_amd/csrc/hk_fit_tu.hip for MODEL 0|1|2, R2 0|1) rebuilt with the given flags.  For A/B runs with tools/ab/ab_quick.sh.
set -e
cd "$(dirname "$0")/.."
name=$1; m=$2; r=$3; shift; shift; shift
mkdir -p _ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical \
    -DHK_TU_MODEL=$m -DHK_TU_R2=$r "$@" -c homonim_amd/csrc/hk_fit_tu.hip -o _ab/hk_fit_m${m}_r${r}_$name.o
objs=$(ls homonim_amd/lib/*.o | grep -v "/hk_fit_m${m}_r${r}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/lib_$name.so _ab/hk_fit_m${m}_r${r}_$name.o $objs
echo "built _ab/lib_$name.so"
