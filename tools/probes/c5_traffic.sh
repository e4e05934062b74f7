set -e
R=$(pwd); O=$R/gpurun_out
DOMINANT="conv_bf16_256p_kernel<3, false, false, false>" bash tools/pmc_traffic.sh r04b_bf16c5 --config 5 > $O/r04b_bf16c5_hbm_traffic.txt 2>&1
tail -3 $O/r04b_bf16c5_hbm_traffic.txt
python3 tools/traffic_per_launch.py $O/pmc_r04b_bf16c5_FETCH_SIZE $O/pmc_r04b_bf16c5_WRITE_SIZE 1024 256 2 > $O/r04b_bf16c5_traffic_per_launch.txt
tail -1 $O/r04b_bf16c5_traffic_per_launch.txt
