#!/bin/bash
# Cold-job costs of the dataset run: BASELINE config 4 on one GPU as (a) a warm process (engine.warmup() before
# the job, the round-3 protocol), (b) a cold process without any tune cache (the job tunes its two batch sizes itself, under the
# staging of its first frames), (c) a cold process that finds the per-user tune cache of an earlier process, and the 12-video
# job (one rank's share at W = 8) the same three ways; then a kernel trace of the cold job for tools/gpu_gaps.py.
#   bash tools/config4_cold.sh <tag>     -> gpurun_out/<tag>_config4_*.json|txt
set -e
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
C=/tmp/tsm_tune_cache_$$.txt
rm -f $C
run() {  # name, env..., -- args
  local name=$1; shift
  env "$@" python3 tools/bench_configs.py --config 4 ${EXTRA} > $O/${tag}_config4_${name}.json 2> $O/${tag}_config4_${name}.err || { tail -5 $O/${tag}_config4_${name}.err; exit 1; }
  python3 -c "import json,sys; d=json.loads(open('$O/${tag}_config4_${name}.json').read().strip().splitlines()[-1]); print('$name', d['value'], 'clips/s', d['job_s'], 's; first forward after', d.get('first_forward_after_s'), 's; cold', d.get('cold'))"
}
for dtype in f32 bf16x3; do
  echo "== $dtype, 100 videos"
  EXTRA="--dtype $dtype" run ${dtype}_warm TSM_TUNE_CACHE=off
  EXTRA="--dtype $dtype --cold" run ${dtype}_cold_nocache TSM_TUNE_CACHE=off
  EXTRA="--dtype $dtype --cold" run ${dtype}_cold_fills_cache TSM_TUNE_CACHE=$C
  EXTRA="--dtype $dtype --cold" run ${dtype}_cold_cached TSM_TUNE_CACHE=$C
  echo "== $dtype, 12 videos (one rank's share at W = 8)"
  EXTRA="--dtype $dtype --videos 12" run ${dtype}_12v_warm TSM_TUNE_CACHE=off
  EXTRA="--dtype $dtype --videos 12 --cold" run ${dtype}_12v_cold_nocache TSM_TUNE_CACHE=off
  EXTRA="--dtype $dtype --videos 12 --cold" run ${dtype}_12v_cold_cached TSM_TUNE_CACHE=$C
done
echo "== kernel trace of the cold cached f32 job"
(cd /tmp && TMPDIR=/tmp TSM_TUNE_CACHE=$C rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_config4_trace -o run -- \
   python3 $R/tools/bench_configs.py --config 4 --cold > $O/${tag}_config4_traced.json 2> $O/${tag}_config4_traced.err)
python3 tools/gpu_gaps.py $(ls $O/${tag}_config4_trace/*/run_kernel_trace.csv $O/${tag}_config4_trace/run_kernel_trace.csv 2>/dev/null | head -1) 1000 > $O/${tag}_config4_gpu_gaps.txt
head -30 $O/${tag}_config4_gpu_gaps.txt
(cd /tmp && TMPDIR=/tmp TSM_TUNE_CACHE=$C rocprofv3 --kernel-trace --output-format csv -d $O/${tag}_config4_trace12 -o run -- \
   python3 $R/tools/bench_configs.py --config 4 --cold --videos 12 > $O/${tag}_config4_traced12.json 2> $O/${tag}_config4_traced12.err)
python3 tools/gpu_gaps.py $(ls $O/${tag}_config4_trace12/*/run_kernel_trace.csv $O/${tag}_config4_trace12/run_kernel_trace.csv 2>/dev/null | head -1) 1000 > $O/${tag}_config4_gpu_gaps_12videos.txt
head -12 $O/${tag}_config4_gpu_gaps_12videos.txt
rm -rf $O/${tag}_config4_trace $O/${tag}_config4_trace12
