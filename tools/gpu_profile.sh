#!/bin/bash
# Run on the GPU box through gpurun: kernel-trace stats + separate PMC passes for bench.py.
# usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err
done
cd $ROOT
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
