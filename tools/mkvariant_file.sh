#!/bin/bash
# tools/mkvariant_file.sh NAME SRC [-Dflag ...]: _ab/lib_NAME.so = current objects with homonim_amd/csrc/SRC.hip rebuilt with the flags
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift; shift
mkdir -p _ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function "$@" \
    -c homonim_amd/csrc/$src.hip -o _ab/${src}_$name.o
objs=$(ls homonim_amd/lib/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ab/lib_$name.so _ab/${src}_$name.o $objs
echo "built _ab/lib_$name.so"
