"""HBM traffic of EVERY launch of one forward next to its algorithmic bytes, from the two PMC passes of tools/pmc_traffic.sh.

    python tools/traffic_per_launch.py gpurun_out/pmc_<tag>_FETCH_SIZE gpurun_out/pmc_<tag>_WRITE_SIZE frames size [bytes_per_element [kernel_trace.csv]]

Measured = (2 * FETCH_SIZE, WRITE_SIZE) * 1024 per dispatch (MI355X_MICROARCH.md 'HBM': KiB units, gfx950 halves wide reads);
the counters sit at L2's memory side, so reads served by the Infinity Cache count as well.  Algorithmic = every tensor a
launch has to read or write, once: activations in, residual in, activations out (+ the packed weights), at
``bytes_per_element`` per activation element (2 in bf16, 4 in fp32 and split-bf16).  The ratio is what a launch re-reads.

With a ``rocprofv3 --kernel-trace`` CSV of the same command as sixth argument every row also gets its ROOFLINE: ``us`` = the
launch's duration in that trace, ``bound_us`` = max(algorithmic bytes / 8 TB/s, algorithmic flops / the dense MFMA peak of the
mode: 2 500 TFLOP/s bf16, 157.3 fp32; split-bf16 executes three bf16 MFMAs per product), ``x_bound`` = us / bound_us and which
of the two roofs is the nearer one -- so "this launch sits on its roofline" is a column of a regenerated table, not a sentence.
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hbm_traffic  # noqa: E402
from layer_times import match_schedule  # noqa: E402
from workoutdetector_amd.flops import layer_table  # noqa: E402


HBM_PEAK = 8.0e12
MFMA_PEAK = {2: 2.5e15, 4: 157.3e12}        # by bytes per element: bf16 | fp32 (the split-bf16 table passes --x3)


def conv_like(name):
    return 'conv' in name or 'stem_' in name or 'bneck_ws' in name or 'front_s2' in name


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


def trace_durations(path):
    """us per conv-like launch of the last whole forward of a kernel trace, in launch order (a split-K reduce is added to its conv)."""
    from layer_times import read_rows
    from hbm_traffic import forward_starts
    rows = read_rows(path)
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = forward_starts([r['Kernel_Name'] for r in rows])
    out = []
    for r in rows[idx[-2]:idx[-1]]:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if conv_like(r['Kernel_Name']) and 'splitk_reduce' not in r['Kernel_Name']:
            out.append(dict(Kernel_Name=r['Kernel_Name'], us=d))
        elif 'splitk_reduce' in r['Kernel_Name'] and out:
            out[-1]['us'] += d
    return out


def main(fetch_dir, write_dir, frames, size, elt=2, trace=None, x3=False):
    f = launches(fetch_dir, 'FETCH_SIZE')
    w = launches(write_dir, 'WRITE_SIZE')
    rf, okf = match_schedule(f)
    rw, okw = match_schedule(w)
    if not (okf and okw) or [x[0] for x in rf] != [x[0] for x in rw]:
        print('the two passes ran different schedules (the tuner chose differently): rows are matched by position')
    durs = None
    if trace:
        rt, okt = match_schedule(trace_durations(trace))
        if not okt or [x[0] for x in rt] != [x[0] for x in rf]:
            print('the kernel trace ran a different schedule than the counter passes: no roofline columns')
        else:
            durs = [x[2]['us'] for x in rt]
    byname = {r['name']: r for r in layer_table(size, size)}
    peak = 2.5e15 / 3 if x3 else MFMA_PEAK[elt]
    head = f"{'launch':36s} {'kernel':46s} {'read MB':>9s} {'write MB':>9s} {'alg. read':>9s} {'alg. write':>10s} {'ratio':>6s}"
    if durs:
        head += f" {'us':>8s} {'bound_us':>9s} {'x_bound':>8s} {'roof':>5s}"
    print(head)
    tm = ta = tu = tb = 0.0
    for i, ((nm, parts, a), (_, _, b)) in enumerate(zip(rf, rw)):
        rd, wr = 2.0 * a['value'] * 1024, b['value'] * 1024
        kn = a['Kernel_Name'].split('(')[0].split('tsm::')[-1][:46]
        ard, awr = algorithmic(parts, byname, frames, size, elt, nm, kn)
        tm += rd + wr
        ta += ard + awr
        line = f'{nm:36s} {kn:46s} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {ard / 1e6:9.1f} {awr / 1e6:10.1f} {(rd + wr) / (ard + awr):6.2f}'
        if durs:
            fl = 2.0 * frames * sum(byname[q]['macs'] for q in parts)
            t_hbm, t_mfma = (ard + awr) / HBM_PEAK * 1e6, fl / peak * 1e6
            bound = max(t_hbm, t_mfma)
            tu += durs[i]
            tb += bound
            line += f" {durs[i]:8.1f} {bound:9.1f} {durs[i] / bound:8.2f} {'hbm' if t_hbm >= t_mfma else 'mfma':>5s}"
        print(line)
    print(f'conv-like launches: measured {tm / 1e9:.2f} GB, algorithmic {ta / 1e9:.2f} GB, ratio {tm / ta:.2f}')
    if durs:
        print(f'conv-like launches: {tu:.0f} us against {tb:.0f} us of per-launch roofline (HBM 8 TB/s | dense MFMA peak): x {tu / tb:.2f}')


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if a != '--x3']
    main(args[0], args[1], int(args[2]), int(args[3]), int(args[4]) if len(args) > 4 else 2, args[5] if len(args) > 5 else None,
         x3='--x3' in sys.argv)
