"""Where a job's GPU time goes, from a rocprofv3 --kernel-trace CSV: busy time per kernel family, and the idle gaps
between consecutive kernels (how many, how long, and which kernel ended each long gap).

    python tools/gpu_gaps.py <kernel_trace.csv> [min_gap_us]
"""
import csv
import sys
from collections import defaultdict


def main(path, min_gap_us=100.0):
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(path))]
    rows.sort()
    # the job proper: from the first preprocess kernel to the last kernel
    first = next((i for i, r in enumerate(rows) if 'preprocess' in r[2]), 0)
    rows = rows[first:]
    span = (rows[-1][1] - rows[0][0]) / 1e3
    fam = defaultdict(float)
    busy_end, busy = rows[0][0], 0.0
    gaps = []
    for s, e, n in rows:
        key = n.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('tsm::', '')[:90]
        fam[key] += (e - s) / 1e3
        if s > busy_end:
            gaps.append(((s - busy_end) / 1e3, key, (s - rows[0][0]) / 1e6))
            busy += (e - s) / 1e3
        else:
            busy += max(0, e - max(s, busy_end)) / 1e3
        busy_end = max(busy_end, e)
    idle = span - busy
    print(f'span {span / 1e3:.1f} ms, busy {busy / 1e3:.1f} ms ({busy / span:.3f}), idle {idle / 1e3:.1f} ms in {len(gaps)} gaps')
    big = [g for g in gaps if g[0] >= min_gap_us]
    print(f'gaps >= {min_gap_us:.0f} us: {len(big)}, {sum(g[0] for g in big) / 1e3:.1f} ms; small gaps: {len(gaps) - len(big)}, '
          f'{sum(g[0] for g in gaps if g[0] < min_gap_us) / 1e3:.1f} ms')
    after = defaultdict(lambda: [0, 0.0])
    for g, k, at in big:
        print(f'  {g / 1e3:7.1f} ms idle until t = {at:7.0f} ms, ended by {k}')
    for g, k, _ in big:
        after[k][0] += 1
        after[k][1] += g
    for k, (n, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f'  long gaps ended by {k:90s} {n:5d} x, {t / 1e3:8.1f} ms')
    print('busy time by kernel family:')
    for k, t in sorted(fam.items(), key=lambda kv: -kv[1])[:12]:
        print(f'  {k:90s} {t / 1e3:9.1f} ms')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 100.0)
