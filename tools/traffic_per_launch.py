"""HBM traffic of EVERY launch of one forward next to its algorithmic bytes, from the two PMC passes of tools/pmc_traffic.sh.

    python tools/traffic_per_launch.py gpurun_out/pmc_<tag>_FETCH_SIZE gpurun_out/pmc_<tag>_WRITE_SIZE frames size [bytes_per_element]

Measured = (2 * FETCH_SIZE, WRITE_SIZE) * 1024 per dispatch (MI355X_MICROARCH.md 'HBM': KiB units, gfx950 halves wide reads);
the counters sit at L2's memory side, so reads served by the Infinity Cache count as well.  Algorithmic = every tensor a
launch has to read or write, once: activations in, residual in, activations out (+ the packed weights), at
``bytes_per_element`` per activation element (2 in bf16, 4 in fp32 and split-bf16).  The ratio is what a launch re-reads.
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hbm_traffic  # noqa: E402
from layer_times import match_schedule  # noqa: E402
from workoutdetector_amd.flops import layer_table  # noqa: E402


def conv_like(name):
    return 'conv' in name or 'stem_' in name or 'bneck_ws' in name


def launches(dirname, counter):
    per, names = hbm_traffic.load(os.path.join(dirname, 'run_counter_collection.csv'), counter)
    ids = hbm_traffic.last_forward(per, names)
    return [dict(Kernel_Name=names[d], value=per[d]) for d in ids if conv_like(names[d]) and 'splitk_reduce' not in names[d]]


def algorithmic(parts, byname, frames, size, elt, label, kernel=''):
    """(read, write) bytes of one launch that runs the layers ``parts``."""
    planar = 'stem_pool' in kernel and kernel.rstrip().endswith('true>')     # the stem reads [N, 3, H, W] fp32 itself
    rd = wr = 0.0
    produced = set()
    for q in parts:
        r = byname[q]
        rd += r['cout'] * r['cin'] * r['k'] * r['k'] * (2 if elt == 2 else 4) / frames       # weights: once per launch
        if q == 'conv1':                                                                      # stem + max-pool: packed input (4 channels)
            rd += size * size * (12 if planar else 4 * elt)
            hp = (size // 2 + 1) // 2
            wr += hp * hp * r['cout'] * elt
            continue
        part = q.split('.')[-1]
        m_in = r['m'] * r['s'] * r['s']
        prev = {'conv2': q.replace('conv2', 'conv1'), 'conv3': q.replace('conv3', 'conv2')}.get(part)
        if prev not in produced:                                                              # input produced inside the same launch: no bytes
            rd += (r['m'] if part == 'downsample' else m_in) * r['cin'] * elt
        produced.add(q)
    last = byname[[q for q in parts if not q.endswith('downsample')][-1]]
    if parts != ['conv1']:
        wr += last['m'] * last['cout'] * elt
    c3 = [q for q in parts if q.endswith('.conv3')]
    if c3:
        blk = c3[0][:-len('.conv3')]
        has_down = blk + '.downsample' in parts
        whole = blk + '.conv1' in parts
        if not has_down and not whole:
            rd += byname[c3[0]]['m'] * byname[c3[0]]['cout'] * elt                            # the identity
        if '+layer' in label or (len(parts) > 1 and parts[-1].endswith('.conv1') and not whole):
            # conv3 of block b + conv1 of block b + 1: the block output is written AND t1 is written; conv1's input is not read
            n1 = byname[parts[-1]]
            rd -= n1['m'] * n1['cin'] * elt
            wr += byname[c3[0]]['m'] * byname[c3[0]]['cout'] * elt
    return rd * frames, wr * frames


def main(fetch_dir, write_dir, frames, size, elt=2):
    f = launches(fetch_dir, 'FETCH_SIZE')
    w = launches(write_dir, 'WRITE_SIZE')
    rf, okf = match_schedule(f)
    rw, okw = match_schedule(w)
    if not (okf and okw) or [x[0] for x in rf] != [x[0] for x in rw]:
        print('the two passes ran different schedules (the tuner chose differently): rows are matched by position')
    byname = {r['name']: r for r in layer_table(size, size)}
    print(f"{'launch':36s} {'kernel':46s} {'read MB':>9s} {'write MB':>9s} {'alg. read':>9s} {'alg. write':>10s} {'ratio':>6s}")
    tm = ta = 0.0
    for (nm, parts, a), (_, _, b) in zip(rf, rw):
        rd, wr = 2.0 * a['value'] * 1024, b['value'] * 1024
        kn = a['Kernel_Name'].split('(')[0].split('tsm::')[-1][:46]
        ard, awr = algorithmic(parts, byname, frames, size, elt, nm, kn)
        tm += rd + wr
        ta += ard + awr
        print(f'{nm:36s} {kn:46s} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {ard / 1e6:9.1f} {awr / 1e6:10.1f} {(rd + wr) / (ard + awr):6.2f}')
    print(f'conv-like launches: measured {tm / 1e9:.2f} GB, algorithmic {ta / 1e9:.2f} GB, ratio {tm / ta:.2f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 2)
