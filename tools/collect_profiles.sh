#!/bin/bash
# Copy the summaries tools/profile_all.sh left under gpurun_out/ into profiles/ (tracked):  bash tools/collect_profiles.sh r03
set -e
t=${1:-r03}; G=gpurun_out; P=profiles
for m in f32 bf16c5 bf16x3; do
  cp $G/${t}_${m}_kernel_stats.csv $P/${t}_${m}_kernel_stats.csv
  cp $G/${t}_${m}_per_layer.txt $P/${t}_${m}_per_layer.txt
  cp $G/${t}_${m}_hbm_traffic.txt $P/${t}_${m}_hbm_traffic.txt
  [ -f $G/${t}_${m}_traffic_per_launch.txt ] && cp $G/${t}_${m}_traffic_per_launch.txt $P/${t}_${m}_traffic_per_launch.txt
done
[ -f $G/${t}_bf16c5_pmc_insts.txt ] && cp $G/${t}_bf16c5_pmc_insts.txt $P/
cp $G/${t}_f32_pmc_sq.txt $G/${t}_bf16c5_pmc_sq1.txt $G/${t}_bf16c5_pmc_sq2.txt $G/${t}_bf16x3_pmc_sq.txt $P/
cp $G/${t}_f32_bench_line.json $P/${t}_f32_b32_bench_line_profiled.json
cp $G/${t}_bf16c5_bench_line.json $P/${t}_bf16_config5_bench_line_profiled.json
cp $G/${t}_bf16x3_bench_line.json $P/${t}_bf16x3_b32_bench_line_profiled.json
[ -f $G/${t}_traffic.json ] && cp $G/${t}_traffic.json $P/traffic.json
# un-profiled bench lines, if the same call produced them (python bench.py [--config 5 | --dtype bf16 ...] > gpurun_out/<tag>_bench3*.json)
[ -f $G/${t}_bench3.json ] && grep '^{' $G/${t}_bench3.json | tail -1 > $P/${t}_bench_line.json
[ -f $G/${t}_bench3_c5.json ] && grep '^{' $G/${t}_bench3_c5.json | tail -1 > $P/${t}_bench_line_config5.json
[ -f $G/${t}_bench3_bf16.json ] && grep '^{' $G/${t}_bench3_bf16.json | tail -1 > $P/${t}_bench_line_bf16_b32.json
ls $P | grep ${t}_ | wc -l
