set -e
export TSM_TUNE_CACHE=off   # a profiler run never writes (or reads) the per-user tune cache: serialised dispatches favour the one-launch forms
R=$(pwd); O=$R/gpurun_out
python3 bench.py --config 5 --no-alt --no-config5 --no-cpu-baseline > $O/c5p_bench1.json 2> $O/c5p_bench1.log
tail -1 $O/c5p_bench1.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5p_stats -o run -- python3 $R/bench.py --steps 20 --warmup 5 --no-alt --no-config5 --no-cpu-baseline --no-parity --config 5 > $O/c5p_prof_line.json 2> $O/c5p_stats.log
cd $R
python3 tools/layer_times.py $O/c5p_stats/run_kernel_trace.csv 1024 256 > $O/c5p_per_layer.txt
sed -n 5,8p $O/c5p_per_layer.txt; tail -1 $O/c5p_per_layer.txt
