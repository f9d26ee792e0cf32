#!/bin/bash
# Tuning / diagnostic build of libct_hip with extra flags on ONE source (SRC, default linear); the other objects come from csrc/build
# (built first if stale).  A wrapper around `make variant`: the per-file flags and the object list are the Makefile's, so an A/B
# number from a variant carries over to libct_hip.so.    usage: [SRC=reinhard_persist] tools/build_variant.sh <tag> <flags...>
#   ->  color-transfer_amd/ct_hip/libct_tune_<tag>.so   (select it with CT_HIP_LIB=...; never shipped, never timed by bench.py)
set -e
cd "$(dirname "$0")/../color-transfer_amd/csrc"
tag=$1; shift
make -s -j8 all
make -s variant VSRC=${SRC:-linear} TAG=$tag EXTRA="$*"
