"""Per-layer time and TFLOP/s of one forward from a rocprofv3 --kernel-trace CSV.

    python tools/layer_times.py gpurun_out/prof1/runc/700_kernel_trace.csv|run_results.db [frames [size]]
"""
import csv
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workoutdetector_amd.flops import layer_table  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hbm_traffic import forward_starts  # noqa: E402


def read_rows(path):
    """Rows with the CSV column names, from a kernel-trace CSV or a rocpd ``*_results.db``."""
    if not path.endswith('.db'):
        return list(csv.DictReader(open(path)))
    import sqlite3
    db = sqlite3.connect(path)
    q = ('select name, start, end, grid_x, workgroup_x, vgpr_count, lds_size from kernels')
    return [dict(Kernel_Name=n, Start_Timestamp=s, End_Timestamp=e, Grid_Size_X=g, Workgroup_Size_X=w, VGPR_Count=v,
                 LDS=l) for n, s, e, g, w, v, l in db.execute(q)]


def match_schedule(convs):
    """[(label, [layer names priced with it], launch record)] for the conv-like launches of one forward, in launch order,
    and whether the launch count matched the expected schedule."""
    # Expected launch order, consumed against the trace: the stem; per block [downsample,] conv1, then either conv2 and
    # conv3 (the latter fused with the downsample branch in a stage's first block) or ONE conv23_fused /
    # conv3x3_ws_kernel<true> launch.
    def _dual(name):      # conv_igemm<BM, BN, WGM, WGN, KS, SHIFT, RES, PREC, DUAL, SEG> / conv_bf16_256[p]_kernel<KS, SHIFT, RES, DUAL>
        if '<' not in name:
            return False
        args = [a.strip() for a in name.split('<')[1].split('>')[0].split(',')]
        if 'conv_igemm<' in name:
            return len(args) > 8 and args[8] == 'true'
        return ('conv_bf16_256_kernel<' in name or 'conv_bf16_256p_kernel<' in name) and len(args) > 3 and args[3] == 'true'
    n_down = sum(_dual(r['Kernel_Name']) for r in convs)
    rows, it = [], iter(convs)

    def take(names):
        r = next(it, None)
        if r is not None:
            rows.append(('+'.join(n.split('.')[-1] if i else n for i, n in enumerate(names)), names, r))
        return r

    take(['conv1'])
    blocks = [f'layer{li}.{b}' for li, nb in enumerate((3, 4, 6, 3), 1) for b in range(nb)]
    have_t1 = False       # the previous block's conv3 launch (conv31_fused_kernel) ran this block's shift + conv1 as well
    for k, p in enumerate(blocks):
        if True:
            b = int(p.split('.')[1])
            separate_down = b == 0 and n_down == 0
            if have_t1:
                have_t1 = False
            else:
                first = next(it, None)
                if first is None:
                    break
                if 'front_s2' in first['Kernel_Name']:      # shift + conv1 + the stride-2 conv2 in one launch
                    rows.append((p + '.conv1+conv2', [p + '.conv1', p + '.conv2'], first))
                    r3 = take([p + '.conv3'] + ([p + '.downsample'] if b == 0 else []))
                    continue
                if 'bneck_ws' in first['Kernel_Name']:      # the whole Bottleneck in one launch
                    rows.append((p + ' (block)', [p + '.conv1', p + '.conv2', p + '.conv3'] + ([p + '.downsample'] if b == 0 else []), first))
                    continue
                if separate_down:
                    rows.append((p + '.downsample', [p + '.downsample'], first))
                    take([p + '.conv1'])
                else:
                    rows.append((p + '.conv1', [p + '.conv1'], first))
            nxt = next(it, None)
            if nxt is None:
                break
            if 'conv23_fused' in nxt['Kernel_Name'] or 'conv3x3_ws_kernel<true>' in nxt['Kernel_Name']:
                rows.append((p + '.conv2+conv3', [p + '.conv2', p + '.conv3'], nxt))
                continue
            rows.append((p + '.conv2', [p + '.conv2'], nxt))
            r3 = take([p + '.conv3'] + ([p + '.downsample'] if (b == 0 and not separate_down) else []))
            if r3 is not None and 'conv31_' in r3['Kernel_Name'] and k + 1 < len(blocks):
                # conv3 + residual of this block and shift + conv1 of the next one in ONE launch: priced with both
                rows[-1] = (p + '.conv3+' + blocks[k + 1] + '.conv1', rows[-1][1] + [blocks[k + 1] + '.conv1'], r3)
                have_t1 = True
    return rows, not (next(it, None) is not None or len(rows) != len(convs))


def main(path, frames=256, size=224):
    rows = read_rows(path)
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    idx = forward_starts(names)
    fw = rows[idx[-2]:idx[-1]]
    # a split-K layer is two launches: the conv and its splitk_reduce, whose time is added to the conv's row
    convs = []
    for r in fw:
        if 'conv' in r['Kernel_Name'] or 'stem_' in r['Kernel_Name'] or 'bneck_ws' in r['Kernel_Name'] or 'front_s2' in r['Kernel_Name']:   # stem_direct / stem_pool = conv1 (+ max-pool); bneck_ws = a whole block
            convs.append(dict(r))
        elif 'splitk_reduce' in r['Kernel_Name'] and convs:
            convs[-1]['End_Timestamp'] = int(convs[-1]['End_Timestamp']) + int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            convs[-1]['Kernel_Name'] = convs[-1]['Kernel_Name'].replace('>', ' +reduce>', 1)
    byname = {r['name']: r for r in layer_table(size, size)}
    rows, ok = match_schedule(convs)
    if not ok:
        print('launch count', len(convs), 'does not match the expected schedule: rows below may be misaligned')
    tot = totf = 0.0
    for nm, parts, r in rows:
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        fl = 2 * frames * sum(byname[q]['macs'] for q in parts)
        tot += dur
        totf += fl
        kn = r['Kernel_Name'].split('<')[1].split('>')[0] if '<' in r['Kernel_Name'] else r['Kernel_Name'][:30]
        if any(k in r['Kernel_Name'] for k in ('conv23_fused', 'conv_bf16_256', 'conv3x3_ws', 'conv1x1_ws', 'bneck_ws', 'conv31_fused', 'conv31_pc', 'front_s2', 'stem_')):
            kn = r['Kernel_Name'].split('tsm::')[1].split('(')[0]
        grid = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
        print(f"{nm:34s} {kn:44s} wgs={grid:6d} {dur:8.1f}us {fl / dur / 1e6:7.1f} TF/s  vgpr={r.get('VGPR_Count','?')}")
    span = (int(fw[-1]['End_Timestamp']) - int(fw[0]['Start_Timestamp'])) / 1e3
    print(f'sum conv {tot:.1f} us = {totf / tot / 1e6:.1f} TF/s; forward span {span:.1f} us; '
          f'other kernels {sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in fw if "conv" not in r["Kernel_Name"] and "stem_" not in r["Kernel_Name"] and "bneck_ws" not in r["Kernel_Name"] and "front_s2" not in r["Kernel_Name"] and "splitk_reduce" not in r["Kernel_Name"]):.1f} us')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 224)
