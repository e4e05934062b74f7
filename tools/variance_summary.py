"""Summarise one `tools/variance_probe.sh` process: the fused weight-stationary launch's duration next to its L2 <-> fabric
counters PER TCC CHANNEL (16 channels x 8 XCDs), from rocprofv3's JSON (the CSV only carries the sums).

    python tools/variance_summary.py <dir with run_results.json> [kernel substring]

Prints one block per counter: total, per-channel min / mean / max, the max/mean imbalance, and the 8 per-XCD totals."""
import json
import sys
from collections import defaultdict

import numpy as np


def main(path, needle='conv3x3_ws_kernel<true>'):
    d = json.load(open(path + '/run_results.json'))['rocprofiler-sdk-tool'][0]
    names = {k['kernel_id']: k.get('formatted_kernel_name') or k.get('demangled_kernel_name') or k.get('kernel_name')
             for k in d['kernel_symbols']}
    counters = {c['id']['handle']: c for c in d['counters']}
    recs = [r for r in d['callback_records']['counter_collection']
            if needle in (names.get(r['dispatch_data']['dispatch_info']['kernel_id']) or '')]
    if not recs:
        print('no dispatch of', needle)
        return
    recs = recs[-20:]                      # the last 10 forwards x 2 launches: tuned, warm
    dur = np.array([r['dispatch_data']['end_timestamp'] - r['dispatch_data']['start_timestamp'] for r in recs]) / 1e3
    print(f'{needle}: {len(recs)} dispatches, duration us: median {np.median(dur):.1f}  min {dur.min():.1f}  max {dur.max():.1f}')
    per = defaultdict(list)
    for r in recs:
        by = defaultdict(list)
        for v in r['records']:
            by[v['counter_id']['handle']].append(v['value'])
        for h, vals in by.items():
            per[h].append(vals)
    for h, rows in per.items():
        c = counters[h]
        a = np.array(rows, dtype=np.float64).mean(axis=0)          # mean over the dispatches, per instance
        dims = [dm['instance_size'] for dm in c['dimensions']]
        print(f"  {c['name']:28s} total {a.sum():.4g}  per channel min {a.min():.4g} mean {a.mean():.4g} max {a.max():.4g}  "
              f"max/mean {a.max() / max(a.mean(), 1e-30):.3f}  n={a.size} dims={dims}")
        if a.size == 128:
            # instance order in the file: the counter's `instances` list
            idx = [(i['dimensions'][0]['index'], i['dimensions'][1]['index']) for i in c['instances']]
            grid = np.zeros((16, 8))
            for (ch, xcc), v in zip(idx, a):
                grid[ch, xcc] = v
            print('      per XCD :', ' '.join(f'{v:.4g}' for v in grid.sum(axis=0)))
            print('      per chan:', ' '.join(f'{v:.4g}' for v in grid.sum(axis=1)))


if __name__ == '__main__':
    main(*sys.argv[1:])
