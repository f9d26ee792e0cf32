#!/bin/bash
# GPU box: time the table kernels of tuning builds (make BUILD=.. OUT=../ct_hip/libct_tune_<tag>.so EXTRA=..)
# usage: tools/sweep_lut.sh "lib [blocks]" ...
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "$@"; do
  set -- $cfg
  [ -f color-transfer_amd/ct_hip/$1 ] || continue
  echo "== $1 blocks=${2:-auto}"
  CT_HIP_LIB=$PWD/color-transfer_amd/ct_hip/$1 CT_HIP_LUT_BLOCKS=${2:-0} python tools/check_reinhard_lut.py --time-only 2>&1 | grep table
done
