#!/bin/bash
# Tuning / diagnostic build of libct_hip with extra flags on ONE source (SRC, default linear; the other objects are reused
# from csrc/build, so run `make` first).  usage: [SRC=reinhard_persist] tools/build_variant.sh <tag> <flags...>
#   ->  color-transfer_amd/ct_hip/libct_tune_<tag>.so   (select it with CT_HIP_LIB=...; never shipped, never timed by bench.py)
set -e
cd "$(dirname "$0")/../color-transfer_amd/csrc"
tag=$1; shift
mkdir -p build_var
src=${SRC:-linear}
x="-fno-slp-vectorize"
[ $src = reinhard_persist ] && x="$x -mllvm -disable-machine-licm"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function $x "$@" -c $src.hip -o build_var/${src}_$tag.o
objs=""
for f in build/*.o; do o=$(basename $f .o); if [ $o = $src ]; then objs="$objs build_var/${src}_$tag.o"; else objs="$objs $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../ct_hip/libct_tune_$tag.so
echo built libct_tune_$tag.so
