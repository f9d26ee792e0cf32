#!/bin/bash
# Tuning / diagnostic build of libct_hip with extra flags on ONE source (SRC, default linear; the other objects are reused
# from csrc/build).  usage: [SRC=gmflow] tools/build_variant.sh <tag> <flags...>   ->  color-transfer_amd/ct_hip/libct_tune_<tag>.so
set -e
cd "$(dirname "$0")/../color-transfer_amd/csrc"
tag=$1; shift
mkdir -p build_var
src=${SRC:-linear}
noslp=""; case $src in cnn|conv_split|gmflow) noslp=-fno-slp-vectorize;; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function $noslp "$@" -c $src.hip -o build_var/${src}_$tag.o
objs=""; for o in linear idt cnn gmflow conv_split metrics regrain distort unet fsim; do
  if [ $o = $src ]; then objs="$objs build_var/${src}_$tag.o"; else objs="$objs build/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -lhipfft -o ../ct_hip/libct_tune_$tag.so
echo built libct_tune_$tag.so
