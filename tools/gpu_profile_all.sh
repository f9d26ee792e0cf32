#!/bin/bash
# kernel-trace stats of the non-headline paths (run through gpurun): DCMCS3DI 512^2, GMFlow 960x540, IDT 1080p, MK 1080p
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_all
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in dcmcs3di gmflow idt mk; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$t -- python3 $ROOT/tools/bench_$t.py > $OUT/$t.log 2> $OUT/$t.err
  f=$(ls $OUT/$t/*/*_kernel_stats.csv | head -1)
  cp $f $OUT/${t}_kernel_stats.csv
  echo "== $t"; tail -3 $OUT/$t.log; head -14 $f | cut -c1-150
done
