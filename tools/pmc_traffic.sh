#!/bin/bash
export TSM_TUNE_CACHE=${TSM_TUNE_CACHE:-off}   # a profiler run never writes (or reads) the per-user tune cache: serialised dispatches favour the one-launch forms; profile_all.sh hands every pass of a mode the private cache file its kernel-trace pass wrote, so that all passes run ONE schedule
# HBM traffic passes (FETCH_SIZE, WRITE_SIZE in separate runs, MI355X_MICROARCH.md "HBM") for one bench mode.
#   bash tools/pmc_traffic.sh <tag> <bench args...>      (run on the GPU box from the repo root)
#   DOMINANT="<kernel name>" UPDATE="B T H W" bash tools/pmc_traffic.sh ...   also writes profiles/traffic.json (sha-stamped)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
(cd $R && python3 -m workoutdetector_amd.build > /dev/null)   # never let bench.py compile under the profiler
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$c -o run -- \
    python3 $R/bench.py --steps 2 --warmup 2 --no-alt --no-config5 --no-cpu-baseline --no-parity "$@" > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
python3 $R/tools/hbm_traffic.py $R/gpurun_out/pmc_${tag}_FETCH_SIZE/run_counter_collection.csv \
  $R/gpurun_out/pmc_${tag}_WRITE_SIZE/run_counter_collection.csv "$DOMINANT" ${UPDATE:+--update-json $UPDATE}
