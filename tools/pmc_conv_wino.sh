#!/bin/bash
# GPU box: SQ counters of the two Winograd kernels (csrc/conv_wino4.hip, conv_wino.hip) under tools/bench_conv_ws.py at 2 x 64 x 1080 x 1920.
# usage (through gpurun): bash tools/pmc_conv_wino.sh [forms...]   -> gpurun_out/pmc_conv_wino/form<f>_<pass>.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_conv_wino
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CT_HIP_CONV_WINO=1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE"
P4="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
for f in ${@:-0 1}; do
  i=0
  for pass in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    CT_HIP_WINO_FORM=$f timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/form${f}_$i -- python3 $ROOT/tools/bench_conv_ws.py 2 64 64 1080 1920 1.0 > $OUT/form${f}_$i.txt 2>&1
  done
done
cd $ROOT
for d in $OUT/form*_?; do
  python3 tools/summarize_pmc.py $d conv_wino > $d.json 2>$d.err
done
find $OUT -name "*.csv" -size +1M -delete
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_conv_wino")
for f in sorted(glob.glob(os.path.join(out, "*.json"))):
    j = json.load(open(f))
    for k, v in j.items():
        if isinstance(v, dict) and "dispatches" in v:
            print(os.path.basename(f), k[:70], {a: round(b, 1) for a, b in v.items() if a not in ("dispatches",)})
PY
