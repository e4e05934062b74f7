set -e
R=$(pwd); O=$R/gpurun_out
timeout -k 10 1000 python -m pytest tests/test_bf16_gpu.py -x -q > $O/xcd_test.log 2>&1 || { tail -30 $O/xcd_test.log; exit 1; }
tail -2 $O/xcd_test.log
python3 bench.py --config 5 --no-alt --no-cpu-baseline > $O/xcd_c5.json 2> $O/xcd_c5.log
python3 -c "import json; d=json.loads(open('$O/xcd_c5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['ms_per_step'], d['parity']['ok'])"
TSM_STEM_PLANAR=0 python3 bench.py --config 5 --no-alt --no-cpu-baseline > $O/xcd_c5_p0.json 2> $O/xcd_c5_p0.log
python3 -c "import json; d=json.loads(open('$O/xcd_c5_p0.json').read().strip().splitlines()[-1]); print('config5 planar=0', d['value'], d['ms_per_step'], d['parity']['ok'])"
bash tools/probes/c5_traffic.sh
