#!/bin/bash
# GPU box: kernel trace of the GMFlow forward (960x540), per-(kernel, grid) table.  usage: gpu_trace_gmflow.sh [rows]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_gm
rm -rf $OUT; mkdir -p $OUT
python3 $ROOT/tools/bench_gmflow.py 540 960 5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/bench_gmflow.py 540 960 3 > $OUT/run.txt 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_shapes.py $f 5 ${1:-45}
find $OUT -name "*kernel_trace.csv" -size +20M -delete
