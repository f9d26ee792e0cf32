#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counters>" <python script + args>   (run on the GPU box through gpurun)
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 $ROOT/$@ > $OUT/stdout.txt 2> $OUT/stderr.txt
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]; a = agg[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k in agg:
    if "ct::" not in k: continue
    print(k)
    for c in sorted(agg[k]):
        s, n = agg[k][c]; print("    %-30s %16.1f (n=%d)" % (c, s / n, n))
PY
