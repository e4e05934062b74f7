for dt in bf16 bf16x3; do for b in 2 4 8 16 32 64; do
python bench.py --dtype $dt --no-alt --no-config5 --no-cpu-baseline --batch $b --steps 30 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt', $b, d['value'], d['ms_per_step'], d['roofline']['forward_achieved'])"
done; done
