set -e
R=$(pwd); O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q -k "stem" > $O/planar_test.log 2>&1 || { tail -30 $O/planar_test.log; exit 1; }
tail -3 $O/planar_test.log
for v in 1 0; do
  export TSM_STEM_PLANAR=$v
  python3 bench.py --config 5 --no-alt --no-cpu-baseline > $O/planar_c5_$v.json 2> $O/planar_c5_$v.log
  python3 -c "import json; d=json.loads(open('$O/planar_c5_$v.json').read().strip().splitlines()[-1]); print('config5 planar=$v', d['value'], d['ms_per_step'], d['parity']['ok'], d['parity']['max_err_over_scale'])"
  python3 bench.py --no-cpu-baseline > $O/planar_f32_$v.json 2> $O/planar_f32_$v.log
  python3 -c "import json; d=json.loads(open('$O/planar_f32_$v.json').read().strip().splitlines()[-1]); print('f32 planar=$v', d['value'], d['ms_per_step'], d['parity']['ok'], 'alt', d['alt_precision']['value'])"
done
