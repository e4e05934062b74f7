"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), per MI355X_MICROARCH.md
'HBM': both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read,
so the read side is doubled.  Prints per-kernel totals of the last forward and the per-launch average
for the dominant kernel.

    python tools/hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> "<dominant kernel>" \
        [--update-json clips_per_gpu num_segments height width]

``--update-json`` writes the result into profiles/traffic.json, stamped with the sha of csrc/ it was measured on
(workoutdetector_amd.build.csrc_sha16): bench.py only reports a traffic figure whose stamp matches the current source.
"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load(path, counter):
    per, names = defaultdict(float), {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            d = int(r['Dispatch_Id'])
            per[d] += float(r['Counter_Value'])
            names[d] = r['Kernel_Name']
    return per, names


def forward_starts(kernel_names):
    """Indices of the first launch of every forward in a launch-ordered list of kernel names: the pack of an NTCHW input,
    or -- since the pool-fused stem reads that layout itself -- a stem launch that no pack precedes."""
    out = []
    for i, n in enumerate(kernel_names):
        if 'pack_input' in n or (('stem_pool' in n or 'stem_direct' in n) and (i == 0 or 'pack_input' not in kernel_names[i - 1])):
            out.append(i)
    return out


def last_forward(per, names):
    ids = sorted(per)
    starts = forward_starts([names[d] for d in ids])
    return ids[starts[-2]:starts[-1]]


def main(fetch_csv, write_csv, dominant='conv_igemm<64, 64, 2, 2, 3, false, false, 0, false, true>'):
    f, fn = load(fetch_csv, 'FETCH_SIZE')
    w, wn = load(write_csv, 'WRITE_SIZE')
    # The two passes are separate processes and the engine's tile autotuner may pick a different shape for
    # a few layers in each, so the passes are summarised independently (per kernel name), not zipped.
    fi, wi = last_forward(f, fn), last_forward(w, wn)
    tot_r = sum(2.0 * f[d] * 1024 for d in fi)
    tot_w = sum(w[d] * 1024 for d in wi)
    dom_r = [2.0 * f[d] * 1024 for d in fi if dominant in fn[d]]
    dom_w = [w[d] * 1024 for d in wi if dominant in wn[d]]
    print(f'forward: read {tot_r / 1e9:.3f} GB (FETCH_SIZE x2 x1024), write {tot_w / 1e9:.3f} GB, total {(tot_r + tot_w) / 1e9:.3f} GB')
    per_launch = None
    if dom_r and dom_w:
        per_launch = sum(dom_r) / len(dom_r) + sum(dom_w) / len(dom_w)
        print(f'{dominant}: {len(dom_r)} / {len(dom_w)} launches in the read / write pass, '
              f'avg read {sum(dom_r) / len(dom_r) / 1e6:.1f} MB + write {sum(dom_w) / len(dom_w) / 1e6:.1f} MB '
              f'= {per_launch / 1e6:.1f} MB per launch')
    return (tot_r + tot_w), per_launch


def update_json(kernel, config, forward_bytes, per_launch, note):
    from workoutdetector_amd.build import csrc_sha16
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        d = {'entries': []}
    b, t, h, w = config
    cfg = dict(clips_per_gpu=b, num_segments=t, height=h, width=w)
    d['entries'] = [e for e in d['entries'] if not (e.get('config') == cfg and e.get('kernel') == kernel)]
    d['entries'].append(dict(config=cfg, kernel=kernel, hbm_bytes_per_launch=per_launch, forward_hbm_bytes=forward_bytes,
                             csrc_sha16=csrc_sha16(), method=note))
    json.dump(d, open(path, 'w'), indent=1)
    print('profiles/traffic.json updated for', kernel, cfg)


if __name__ == '__main__':
    args = sys.argv[1:]
    upd = None
    if '--update-json' in args:
        i = args.index('--update-json')
        upd = [int(v) for v in args[i + 1:i + 5]]
        args = args[:i]
    total, per_launch = main(*args[:3])
    if upd:
        update_json(args[2], upd, total, per_launch,
                    'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_traffic.sh) over '
                    '`bench.py --steps 2 --warmup 2 --no-alt --no-cpu-baseline ...`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 '
                    "(gfx950 FETCH_SIZE halves wide coalesced reads, MI355X_MICROARCH.md 'HBM'); per-launch = average over "
                    'the launches of this kernel in the last forward')
