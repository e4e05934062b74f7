#!/bin/bash
export TSM_TUNE_CACHE=off   # a profiler run never writes (or reads) the per-user tune cache: serialised dispatches favour the one-launch forms
# A/B of the activation stores' cache policy (csrc/tsm_device.h: TSM_OUT_AUX and the per-family TSM_AUX_*), config 5:
# throughput (bench.py, parity checked) and per-launch HBM traffic (two PMC passes) per prebuilt library variant.
#   bash tools/store_policy_ab.sh plain sc1 nt     (GPU box, repo root; libraries libtsm_hip[_<variant>].so built beforehand:
#   TSM_LIB_PATH=$PWD/workoutdetector_amd/libtsm_hip_sc1.so TSM_BUILD_DEFS='-DTSM_OUT_AUX=16' python -m workoutdetector_amd.build --force)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
for v in "$@"; do
  if [ "$v" = plain ]; then unset TSM_LIB_PATH; else export TSM_LIB_PATH=$R/workoutdetector_amd/libtsm_hip_$v.so; fi
  echo "== $v (${TSM_LIB_PATH:-default library})"
  for i in 1 2; do
    python3 $R/bench.py --config 5 --no-alt --no-config5 --no-cpu-baseline > $O/sp_${v}_bench$i.json 2> $O/sp_${v}_bench$i.log
    python3 -c "import json,sys; d=json.loads(open('$O/sp_${v}_bench$i.json').read().strip().splitlines()[-1]); print('  clips/s', d['value'], 'ms', d['ms_per_step'], 'parity', d.get('parity'))"
  done
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_sp_${v}_$c -o run -- \
      python3 $R/bench.py --steps 2 --warmup 2 --no-alt --no-config5 --no-cpu-baseline --no-parity --config 5 > $O/pmc_sp_${v}_$c.log 2>&1
  done
  cd $R
  python3 tools/traffic_per_launch.py $O/pmc_sp_${v}_FETCH_SIZE $O/pmc_sp_${v}_WRITE_SIZE 1024 256 2 > $O/sp_${v}_traffic.txt
  python3 tools/hbm_traffic.py $O/pmc_sp_${v}_FETCH_SIZE/run_counter_collection.csv $O/pmc_sp_${v}_WRITE_SIZE/run_counter_collection.csv "conv_bf16_256p_kernel<3, false, false, false>" | head -2
  tail -1 $O/sp_${v}_traffic.txt
done
