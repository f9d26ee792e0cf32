#!/bin/bash
# tuning builds of conv_ws.hip: tools/build_ws_variants.sh FLAG... -> color-transfer_amd/ct_hip/libct_tune_ws<FLAG>.so (flag = macro suffix after CT_WS_)
cd $(dirname $0)/../color-transfer_amd/csrc
for v in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fno-slp-vectorize -DCT_WS_$v -c conv_ws.hip -o /tmp/conv_ws_$v.o 2>&1 | grep -i " error"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v "conv_ws.o\|-hip-") /tmp/conv_ws_$v.o -lhipfft -o ../ct_hip/libct_tune_ws$v.so 2>&1 | grep -i error ) &
done
wait
ls ../ct_hip/
