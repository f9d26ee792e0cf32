#!/bin/bash
# GPU box: SQ counters of the Reinhard kernels (persistent launch and the two sweeps) under tools/bench_reinhard.py
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_reinhard
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
i=0
for pass in "$P1" "$P2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/persist_$i -- python3 $ROOT/tools/bench_reinhard.py 16 > $OUT/persist_$i.txt 2>&1
  CT_HIP_REINHARD_PERSIST=0 timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/sweeps_$i -- python3 $ROOT/tools/bench_reinhard.py 16 > $OUT/sweeps_$i.txt 2>&1
done
cd $ROOT
for d in persist_1 persist_2 sweeps_1 sweeps_2; do
  python3 tools/summarize_pmc.py $OUT/$d reinhard lab_moments > $OUT/$d.json 2>$OUT/$d.err
done
find $OUT -name "*.csv" -size +2M -delete
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_reinhard")
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    j = json.load(open(f))
    for k, v in j.items():
        if isinstance(v, dict) and "dispatches" in v:
            print(os.path.basename(f), k[:60], {a: round(b, 1) for a, b in v.items() if a not in ("dispatches",)})
PY
