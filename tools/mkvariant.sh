#!/bin/bash
# tools/mkvariant.sh NAME [-Dflag ...]: _ab/lib_NAME.so = current objects + hk_kernels.hip rebuilt with the given flags and EVERY
# build of the fused kernel in that one translation unit (-DHK_FIT_ONE_TU; the product build has six, see mkvariant_tu.sh)
# (dev subset: RW=2, RING=1 only -> seconds).  For A/B runs with tools/ab/ab_quick.sh.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p _ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -DHK_FIT_ONE_TU -DHK_DEV_SUBSET "$@" \
    -c homonim_amd/csrc/hk_kernels.hip -o _ab/hk_kernels_$name.o -Rpass-analysis=kernel-resource-usage 2> _ab/res_$name.txt || { tail -20 _ab/res_$name.txt; exit 1; }
objs=$(ls homonim_amd/lib/*.o | grep -v "hk_kernels.o\|hk_fit_m")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/lib_$name.so _ab/hk_kernels_$name.o $objs
echo "built _ab/lib_$name.so"
