#!/bin/bash
# bneck_ws_kernel variants on ONE box, interleaved: bash tools/probes/bneck_ab.sh name ...  (libraries under tools/probes/bin)
for i in 1 2; do
  for n in "" "$@"; do
    lib=${n:+tools/probes/bin/libtsm_$n.so}
    echo "== ${n:-default}"
    TSM_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 200 python tools/block_probe.py 2>&1 | grep -A 3 "TSM_FUSE_BLOCK=auto"
  done
done
