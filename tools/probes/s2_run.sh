set -e
R=$(pwd); O=$R/gpurun_out
python3 bench.py --config 5 --no-alt --no-cpu-baseline > $O/s2_bench1.json 2> $O/s2_bench1.log
tail -1 $O/s2_bench1.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s2_stats -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-parity --config 5 > $O/s2_prof_line.json 2> $O/s2_stats.log
cd $R
python3 tools/layer_times.py $O/s2_stats/run_kernel_trace.csv 1024 256 > $O/s2_per_layer.txt
sed -n 5,8p $O/s2_per_layer.txt; tail -1 $O/s2_per_layer.txt
