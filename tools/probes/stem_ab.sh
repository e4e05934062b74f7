#!/bin/bash
# bf16 stem variants on ONE box, interleaved: bash tools/probes/stem_ab.sh name ...  (libraries under tools/probes/bin)
for i in 1 2; do
  for n in "" "$@"; do
    lib=${n:+tools/probes/bin/libtsm_$n.so}
    echo "== ${n:-default}"
    TSM_LIB_PATH=${lib:+$PWD/$lib} timeout -k 10 200 python tools/probes/stem_ab.py 2>&1 | tail -1
  done
done
