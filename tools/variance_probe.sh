#!/bin/bash
export TSM_TUNE_CACHE=off   # a profiler run never writes (or reads) the per-user tune cache: serialised dispatches favour the one-launch forms
# conv3x3_ws_kernel<true> placement variance: N fresh processes, each timing the fused launch (kernel
# trace) AND collecting the L2 <-> fabric counters per TCC channel in the same run, so a slow-regime process can be laid
# beside a fast one:   bash tools/variance_probe.sh <n_processes> [tag]   -> gpurun_out/variance_<tag>/p<i>/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
n=${1:-6}; tag=${2:-r03}
(cd $R && python3 -m workoutdetector_amd.build > /dev/null)
cd /tmp && export TMPDIR=/tmp
export B=64 T=16 S=256 DTYPE=bf16 ONLY=1
for i in $(seq 1 $n); do
  out=$R/gpurun_out/variance_$tag/p$i
  mkdir -p $out
  rocprofv3 --pmc ${PMC:-TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_STALL} --kernel-trace \
    --output-format csv json -d $out -o run -- python3 $R/tools/fused_probe.py > $out/probe.log 2>&1
  python3 $R/tools/variance_summary.py $out > $out/summary.txt 2>&1 || true
  grep "conv2 " $out/probe.log | head -2 >> $out/summary.txt || true
  rm -f $out/run_results.json $out/run_counter_collection.csv $out/run_kernel_trace.csv    # 50 MB per process: only the summary travels
  echo "== process $i"; head -3 $out/summary.txt
done
