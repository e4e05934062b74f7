#!/bin/bash
# conv_bf16_256p_kernel probe variants on ONE box, interleaved: bash tools/probes/dual_ab.sh name ...  (libraries under tools/probes/bin)
for i in 1 2; do
  for n in "" "$@"; do
    lib=${n:+tools/probes/bin/libtsm_$n.so}
    echo "== ${n:-default}"
    TSM_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 200 python tools/dual_probe.py 64 .0.conv3 conv2 2>&1 | tail -1
  done
done
