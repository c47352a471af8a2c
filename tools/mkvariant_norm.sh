#!/bin/bash
# tools/mkvariant_norm.sh NAME [-Dflag ...]: _ab/lib_NAME.so = current objects + hk_norm.hip rebuilt with the given flags
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p _ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" \
    -c homonim_amd/csrc/hk_norm.hip -o _ab/hk_norm_$name.o
objs=$(ls homonim_amd/lib/*.o | grep -v hk_norm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/lib_$name.so _ab/hk_norm_$name.o $objs
echo "built _ab/lib_$name.so"
