#!/bin/bash
# GPU box: bench.py kernel times of tuning builds (color-transfer_amd/ct_hip/libct_tune_<tag>.so)
# usage: tools/bench_variants.sh out_dir tag[:LUT_BLOCKS]...
cd ${GRAFT_REPO_ROOT:-.}
out=$1; shift
mkdir -p $out
for cfg in "$@"; do
  v=${cfg%%:*}; blocks=0; [ "$cfg" != "$v" ] && blocks=${cfg#*:}
  lib=$PWD/color-transfer_amd/ct_hip/libct_tune_$v.so
  [ "$v" = base ] && lib=$PWD/color-transfer_amd/ct_hip/libct_hip.so
  CT_HIP_LUT_BLOCKS=$blocks CT_HIP_LIB=$lib python3 bench.py --no-extra --no-cpu-baseline --steps 100 --warmup 10 > $out/bench_${v}_$blocks.json 2>&1
  python3 -c "
import json
d=json.loads([l for l in open('$out/bench_${v}_$blocks.json') if l.startswith('{')][-1]); k=d['roofline']['kernels']
print('%-10s blocks %5s' % ('$v', '$blocks'), round(d['value']), round(d['ms_per_step'],4), {n[:14]:round(v['avg_launch_us'],1) for n,v in k.items()})"
done
