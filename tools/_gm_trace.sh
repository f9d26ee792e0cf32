#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gm_lc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ROOT/tools/bench_gmflow.py 540 960 3 > $OUT/t.txt 2>&1
f=$(find $OUT/t -name "*kernel_stats.csv" | head -1)
cp $f $OUT/stats.csv
find $OUT/t -name "*.csv" -size +1M -delete
tail -1 $OUT/t.txt
