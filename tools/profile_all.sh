#!/bin/bash
# (the per-user tune cache is never touched: every mode gets a private cache file under gpurun_out/, written by its kernel-trace pass and read by its counter passes)
# One call that refreshes every judged profile artefact of a round on the GPU box (run from the repo root):
#   [ONLY="f32 bf16c5 bf16x3"] bash tools/profile_all.sh r02        -> gpurun_out/<tag>_*  (copy the summaries you want judged into profiles/)
# Separate rocprofv3 passes for --stats / FETCH_SIZE / WRITE_SIZE / SQ counters (never combined with trace domains).
set -e
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python3 -m workoutdetector_amd.build > /dev/null
O=$R/gpurun_out
run_stats() {  # name, frames, size, bench args...
  local name=$1 frames=$2 size=$3; shift 3
  export TSM_TUNE_CACHE=$O/${tag}_${name}_tune_cache.txt    # this pass tunes (kernel tracing does not serialise dispatches); the counter passes of the mode read it
  rm -f $TSM_TUNE_CACHE
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_${name}_stats -o run -- \
     python3 $R/bench.py --steps 20 --warmup 5 --no-alt --no-config5 --no-cpu-baseline --no-parity "$@" > $O/${tag}_${name}_bench_line.json 2> $O/${tag}_${name}_stats.log)
  python3 tools/layer_times.py $O/${tag}_${name}_stats/run_kernel_trace.csv $frames $size > $O/${tag}_${name}_per_layer.txt
  cp $O/${tag}_${name}_stats/run_kernel_stats.csv $O/${tag}_${name}_kernel_stats.csv
  tail -1 $O/${tag}_${name}_per_layer.txt
}
if [[ " ${ONLY:-f32 bf16c5 bf16x3} " == *" f32 "* ]]; then
echo "== f32 headline"; run_stats f32 256 224
DOMINANT="conv_igemm<64, 64, 2, 2, 3, false, false, 0, false, true>" UPDATE="32 8 224 224" bash tools/pmc_traffic.sh ${tag}_f32 > $O/${tag}_f32_hbm_traffic.txt 2>&1; tail -3 $O/${tag}_f32_hbm_traffic.txt
python3 tools/traffic_per_launch.py $O/pmc_${tag}_f32_FETCH_SIZE $O/pmc_${tag}_f32_WRITE_SIZE 256 224 4 $O/${tag}_f32_stats/run_kernel_trace.csv > $O/${tag}_f32_traffic_per_launch.txt; tail -2 $O/${tag}_f32_traffic_per_launch.txt
bash tools/pmc_sq.sh ${tag}_f32; cp $O/pmc_sq_${tag}_f32.txt $O/${tag}_f32_pmc_sq.txt
fi
if [[ " ${ONLY:-f32 bf16c5 bf16x3} " == *" bf16c5 "* ]]; then
echo "== bf16 config 5"; run_stats bf16c5 1024 256 --config 5
DOMINANT="conv_bf16_256p_kernel<3, false, false, false>" UPDATE="64 16 256 256" bash tools/pmc_traffic.sh ${tag}_bf16c5 --config 5 > $O/${tag}_bf16c5_hbm_traffic.txt 2>&1; tail -3 $O/${tag}_bf16c5_hbm_traffic.txt
python3 tools/traffic_per_launch.py $O/pmc_${tag}_bf16c5_FETCH_SIZE $O/pmc_${tag}_bf16c5_WRITE_SIZE 1024 256 2 $O/${tag}_bf16c5_stats/run_kernel_trace.csv > $O/${tag}_bf16c5_traffic_per_launch.txt; tail -2 $O/${tag}_bf16c5_traffic_per_launch.txt
bash tools/pmc_sq.sh ${tag}_bf16c5 --config 5; cp $O/pmc_sq_${tag}_bf16c5.txt $O/${tag}_bf16c5_pmc_sq1.txt
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  bash tools/pmc_sq.sh ${tag}_bf16c5b --config 5; cp $O/pmc_sq_${tag}_bf16c5b.txt $O/${tag}_bf16c5_pmc_sq2.txt
# instruction mix per launch (the ISA budget of the instruction-bound launches: stems, whole-block kernels)
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_TEX_LOAD SQ_INSTS_TEX_STORE SQ_WAVES GRBM_GUI_ACTIVE" \
  bash tools/pmc_sq.sh ${tag}_bf16c5c --config 5; cp $O/pmc_sq_${tag}_bf16c5c.txt $O/${tag}_bf16c5_pmc_insts.txt
fi
if [[ " ${ONLY:-f32 bf16c5 bf16x3} " == *" bf16x3 "* ]]; then
echo "== bf16x3 batch 32"; run_stats bf16x3 256 224 --dtype bf16x3
DOMINANT="conv_igemm<128, 128, 4, 2, 3, false, false, 1, false, false>" UPDATE="32 8 224 224" bash tools/pmc_traffic.sh ${tag}_bf16x3 --dtype bf16x3 > $O/${tag}_bf16x3_hbm_traffic.txt 2>&1; tail -3 $O/${tag}_bf16x3_hbm_traffic.txt
python3 tools/traffic_per_launch.py $O/pmc_${tag}_bf16x3_FETCH_SIZE $O/pmc_${tag}_bf16x3_WRITE_SIZE 256 224 4 $O/${tag}_bf16x3_stats/run_kernel_trace.csv --x3 > $O/${tag}_bf16x3_traffic_per_launch.txt; tail -2 $O/${tag}_bf16x3_traffic_per_launch.txt
bash tools/pmc_sq.sh ${tag}_bf16x3 --dtype bf16x3; cp $O/pmc_sq_${tag}_bf16x3.txt $O/${tag}_bf16x3_pmc_sq.txt
fi
cp profiles/traffic.json $O/${tag}_traffic.json
echo done
