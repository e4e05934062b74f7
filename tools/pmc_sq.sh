#!/bin/bash
export TSM_TUNE_CACHE=${TSM_TUNE_CACHE:-off}   # a profiler run never writes (or reads) the per-user tune cache: serialised dispatches favour the one-launch forms; profile_all.sh hands every pass of a mode the private cache file its kernel-trace pass wrote, so that all passes run ONE schedule
# SQ counters (MFMA pipe busy, LDS activity, clock) per conv launch of the last forward:
#   bash tools/pmc_sq.sh <tag> <bench args...>     -> gpurun_out/pmc_sq_<tag>.txt
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
(cd $R && python3 -m workoutdetector_amd.build > /dev/null)   # never let bench.py compile under the profiler
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc ${PMC:-GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES} \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_$tag -o run -- \
  python3 $R/bench.py --steps 2 --warmup 2 --no-alt --no-config5 --no-cpu-baseline --no-parity "$@" > $R/gpurun_out/pmc_sq_$tag.log 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_sq_$tag/run_counter_collection.csv gpurun_out/pmc_sq_$tag/run_kernel_trace.csv > gpurun_out/pmc_sq_$tag.txt
