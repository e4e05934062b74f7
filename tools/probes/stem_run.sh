set -e
R=$(pwd); O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q -k "stem" > $O/stem_test.log 2>&1 || { tail -30 $O/stem_test.log; exit 1; }
tail -2 $O/stem_test.log
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_full_size_gpu.py -x -q -k "bf16 or golden or config" > $O/stem_test2.log 2>&1 || { tail -30 $O/stem_test2.log; exit 1; }
tail -2 $O/stem_test2.log
python3 bench.py --config 5 --no-alt --no-cpu-baseline > $O/stem_c5.json 2> $O/stem_c5.log
python3 -c "import json; d=json.loads(open('$O/stem_c5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['ms_per_step'], d['parity']['ok'], d['parity']['max_err_over_scale'])"
bash tools/probes/s2_run.sh | tail -1
sed -n 1,2p $O/s2_per_layer.txt
