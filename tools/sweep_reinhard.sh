#!/bin/bash
# Reinhard headline under different launch geometries / occupancy builds (tuning aid; run through gpurun).
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$1', round(d['value']), [round(v['avg_launch_us'],1) for v in k.values()])"; }
for tb in 1024 1536 1792 2048 2560 3072 4096 8192; do CT_HIP_TARGET_BLOCKS=$tb run "base tb=$tb"; done
for lib in color-transfer_amd/csrc/build/var/*.so; do for tb in 1536 2048 4096; do CT_HIP_LIB=$PWD/$lib CT_HIP_TARGET_BLOCKS=$tb run "$(basename $lib) tb=$tb"; done; done
