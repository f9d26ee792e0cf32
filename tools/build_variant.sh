#!/bin/bash
# Tuning / diagnostic build of libct_hip with extra flags on linear.hip only (the other objects are reused from csrc/build).
# usage: tools/build_variant.sh <tag> <flags...>   ->  color-transfer_amd/ct_hip/libct_tune_<tag>.so
set -e
cd "$(dirname "$0")/../color-transfer_amd/csrc"
tag=$1; shift
mkdir -p build_var
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -Wall -Wno-unused-function "$@" -c linear.hip -o build_var/linear_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_var/linear_$tag.o build/idt.o build/cnn.o build/gmflow.o build/conv_split.o build/metrics.o build/regrain.o build/distort.o -o ../ct_hip/libct_tune_$tag.so
echo built libct_tune_$tag.so
