"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel dispatch of the last forward.

    python tools/pmc_summary.py gpurun_out/pmc1/runc/*_counter_collection.csv gpurun_out/pmc1/runc/*_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hbm_traffic import forward_starts  # noqa: E402


def main(cc_path, kt_path):
    per = defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(cc_path)):
        d = int(r['Dispatch_Id'])
        per[d][r['Counter_Name']] = per[d].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        names[d] = r['Kernel_Name']
    dur = {}
    for r in csv.DictReader(open(kt_path)):
        dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    ids = sorted(per)
    starts = forward_starts([names[d] for d in ids])
    lo, hi = ids[starts[-2]], ids[starts[-1]]
    cols = None
    for d in ids:
        if not (lo <= d < hi) or not ('conv' in names[d] or 'stem_' in names[d] or 'bneck' in names[d] or 'front_s2' in names[d]):
            continue
        c = per[d]
        if cols is None:
            cols = sorted(c)
            print('kernel'.ljust(34), 'us'.rjust(8), ' '.join(x.replace('SQ_', '').replace('GRBM_', '')[:14].rjust(14) for x in cols),
                  'mfma_util'.rjust(9), 'clkGHz'.rjust(7))
        us = dur.get(d, 0.0)
        kn = names[d].split('<')[1].split('>')[0].replace(' ', '') if '<' in names[d] else names[d][:30]
        if 'conv_igemm' not in names[d]:      # conv23_fused_kernel<..>, conv_bf16_256_kernel<..>: keep the kernel's name
            kn = names[d].split('tsm::')[1].split('(')[0].replace(' ', '').replace('_kernel', '')
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs? normalise by BUSY_CU_CYCLES*... report ratio to GUI_ACTIVE
        gui = c.get('GRBM_GUI_ACTIVE', 0.0)
        clk = gui / 8.0 / (us * 1e3) if us else 0.0   # sum over 8 XCDs; cycles per ns
        mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
        util = mf / (gui / 8.0 * 256 * 4) if gui else 0.0
        print(kn.ljust(34), f'{us:8.1f}', ' '.join(f'{c[x]:14.4g}' for x in cols), f'{util:9.3f}', f'{clk:7.3f}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
