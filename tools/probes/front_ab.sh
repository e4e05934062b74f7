#!/bin/bash
# front_s2_kernel variants on ONE box, interleaved: bash tools/probes/front_ab.sh  (libraries under tools/probes/bin)
for i in 1 2; do
  for lib in "" tools/probes/bin/libtsm_fr4.so; do
    echo "== ${lib:-default}"
    TSM_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 200 python tools/front_probe.py 2>&1 | grep -E "FRONT=1|FRONT=auto"
  done
done
