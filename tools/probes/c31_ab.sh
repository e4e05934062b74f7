#!/bin/bash
# conv31_pc_kernel variants on ONE box, interleaved: bash tools/probes/c31_ab.sh  (libraries under tools/probes/bin)
for i in 1 2; do
  for lib in "" tools/probes/bin/libtsm_pcq0.so; do
    echo "== ${lib:-default}"
    TSM_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 300 python tools/c31_probe.py 2>&1 | grep -vE "^\s*$" | tail -12
  done
done
