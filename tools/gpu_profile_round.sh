#!/bin/bash
# GPU box (through gpurun): profiles of one round (ROUND=r06 by default).  Kernel-trace stats of the default bench command, separate PMC passes for HBM
# traffic / instruction counts of the Reinhard kernels, ONE MFMA-busy run per CNN forward (summarised here, on the box, where only
# this call's files exist), IDT stats + traffic.  Everything carries the source stamp of the build it ran (tools/stamp.py).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r06}
OUT=$ROOT/gpurun_out/prof_$ROUND
rm -rf $OUT; mkdir -p $OUT
python3 $ROOT/tools/stamp.py > $OUT/source_stamp.txt
cd /tmp && export TMPDIR=/tmp
BARGS="--steps 200 --warmup 20 --no-cpu-baseline --no-extra"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $BARGS > $OUT/bench_under_rocprofv3.json 2> $OUT/trace.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_$name.err
done
# the one-launch form: float32 by name and the uint8 front door (kernel stats + HBM traffic)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_persist -- python3 $ROOT/tools/bench_reinhard_persist.py > $OUT/trace_persist.txt 2>&1
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/persist_$pass -- python3 $ROOT/tools/bench_reinhard_persist.py > /dev/null 2> $OUT/persist_$pass.err
done
M="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_WAVE_CYCLES"
timeout 400 rocprofv3 --kernel-trace --pmc $M --output-format csv -d $OUT/mfma_dc1080 -- python3 $ROOT/tools/bench_dcmcs3di.py 1080 1920 2 > $OUT/mfma_dc1080.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc $M --output-format csv -d $OUT/mfma_gm960 -- python3 $ROOT/tools/bench_gmflow.py 540 960 2 > $OUT/mfma_gm960.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dc1080 -- python3 $ROOT/tools/bench_dcmcs3di.py 1080 1920 3 > $OUT/trace_dc1080.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gm960 -- python3 $ROOT/tools/bench_gmflow.py 540 960 3 > $OUT/trace_gm960.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dmsct960 -- python3 $ROOT/tools/bench_dmsct.py 540 960 3 > $OUT/trace_dmsct960.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_idt -- python3 $ROOT/tools/bench_idt.py > $OUT/trace_idt.txt 2>&1
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/idt_$pass -- python3 $ROOT/tools/bench_idt.py > /dev/null 2> $OUT/idt_$pass.err
done
# HBM traffic of the CNN kernels (round 6): the same two forwards, FETCH_SIZE and WRITE_SIZE in separate passes
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/dc1080_$pass -- python3 $ROOT/tools/bench_dcmcs3di.py 1080 1920 2 > /dev/null 2> $OUT/dc1080_$pass.err
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/gm960_$pass -- python3 $ROOT/tools/bench_gmflow.py 540 960 2 > /dev/null 2> $OUT/gm960_$pass.err
done
find $OUT/dc1080_FETCH_SIZE $OUT/dc1080_WRITE_SIZE $OUT/gm960_FETCH_SIZE $OUT/gm960_WRITE_SIZE -name "*kernel_trace.csv" -delete
# the ResB convolution alone: both Winograd kernels on one box (times + error against float64), stall counters, phase stamps
{ for f in 0 1; do echo "== CT_HIP_WINO_FORM=$f (0 = conv_wino4.hip, 1 = conv_wino.hip)"; CT_HIP_WINO_FORM=$f CT_HIP_CONV_WINO=1 python3 $ROOT/tools/bench_conv_ws.py 2 64 64 1080 1920 1.0; done; } > $OUT/conv_wino_bench.txt 2>&1
bash $ROOT/tools/pmc_conv_wino.sh 0 1 > $OUT/conv_wino_pmc.txt 2>&1
cp $ROOT/gpurun_out/pmc_conv_wino/*.json $OUT/ 2>/dev/null
cd /tmp
cd $ROOT
# the MFMA-busy summaries are made HERE: this box holds exactly one run per directory
python3 tools/summarize_pmc.py $OUT/mfma_dc1080 ct:: _all > $OUT/dcmcs3di_1080p_mfma_pmc.json 2> $OUT/summ_dc.err
python3 tools/summarize_pmc.py $OUT/mfma_gm960 ct:: _all > $OUT/gmflow_960x540_mfma_pmc.json 2> $OUT/summ_gm.err
# keep what is small enough to travel back (the raw traces of the CNN runs are large)
find $OUT -name "*kernel_trace.csv" -size +20M -delete
find $OUT/mfma_dc1080 $OUT/mfma_gm960 -name "*counter_collection.csv" -size +20M -delete
timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_20_5.json 2> $OUT/bench_20_5.err
tail -c 600 $OUT/bench_default.json
