#!/bin/bash
# Copy the summaries tools/profile_all.sh left under gpurun_out/ into profiles/ (tracked):  bash tools/collect_profiles.sh r02
set -e
t=${1:-r02}; G=gpurun_out; P=profiles
cp $G/${t}_f32_kernel_stats.csv $P/${t}_f32_b32_kernel_stats.csv
cp $G/${t}_f32_bench_line.json $P/${t}_f32_b32_bench_line_profiled.json
cp $G/${t}_f32_hbm_traffic.txt $P/${t}_f32_hbm_traffic.txt
cp $G/${t}_f32_pmc_sq.txt $P/${t}_f32_pmc_sq_summary.txt
cp $G/${t}_bf16c5_kernel_stats.csv $P/${t}_bf16_config5_kernel_stats.csv
cp $G/${t}_bf16c5_bench_line.json $P/${t}_bf16_config5_bench_line_profiled.json
cp $G/${t}_bf16c5_hbm_traffic.txt $P/${t}_bf16c5_hbm_traffic.txt
cp $G/${t}_bf16c5_pmc_sq1.txt $P/${t}_bf16c5_pmc_sq1.txt
cp $G/${t}_bf16c5_pmc_sq2.txt $P/${t}_bf16c5_pmc_sq2.txt
cp $G/${t}_bf16x3_kernel_stats.csv $P/${t}_bf16x3_b32_kernel_stats.csv
python3 tools/layer_times.py $G/${t}_f32_stats/run_kernel_trace.csv 256 224 > $P/${t}_f32_b32_per_layer.txt
python3 tools/layer_times.py $G/${t}_bf16c5_stats/run_kernel_trace.csv 1024 256 > $P/${t}_bf16_config5_per_layer.txt
python3 tools/layer_times.py $G/${t}_bf16x3_stats/run_kernel_trace.csv 256 224 > $P/${t}_bf16x3_b32_per_layer.txt
[ -f $G/${t}_traffic.json ] && cp $G/${t}_traffic.json $P/traffic.json
[ -f $G/${t}_bench_default.json ] && tail -1 $G/${t}_bench_default.json > $P/${t}_bench_line.json
[ -f $G/${t}_bench_c5.json ] && tail -1 $G/${t}_bench_c5.json > $P/${t}_bench_line_config5.json
ls $P | grep ${t}_ | wc -l
[ -f $G/${t}_tile_probe.txt ] && cp $G/${t}_tile_probe.txt $P/${t}_bf16_config5_tile_probe.txt
[ -f $G/${t}_fused_probe_bf16.txt ] && cp $G/${t}_fused_probe_bf16.txt $P/${t}_bf16_config5_fused_probe.txt
