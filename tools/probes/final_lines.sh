set -e
O=gpurun_out
python3 bench.py > $O/r04_bench3.json 2> $O/r04_bench3.log
python3 bench.py --config 5 > $O/r04_bench3_c5.json 2> $O/r04_bench3_c5.log
python3 bench.py --dtype bf16 --no-cpu-baseline > $O/r04_bench3_bf16.json 2> $O/r04_bench3_bf16.log
for f in r04_bench3 r04_bench3_c5 r04_bench3_bf16; do python3 -c "
import json; d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); r=d['roofline']
print('$f', d['value'], d['ms_per_step'], 'frac', r['frac'], 'fwd_frac', r.get('forward_frac'), 'traffic', r.get('traffic'), 'parity', d['parity']['max_err_over_scale'], 'alt', (d.get('alt_precision') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))"; done
