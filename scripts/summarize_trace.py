#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): calls, mean/min/max duration.
Usage: python scripts/summarize_trace.py <dir-with-*_kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

# --alternate SUBSTR: kernels whose name contains SUBSTR are launched alternately for the coarse (64 samples/ray) and the fine (128) pass with
# the same grid (persistent workgroups); their calls are split by position (even = coarse, odd = fine) so that the fine launch has its own row
alt = sys.argv[sys.argv.index('--alternate') + 1] if '--alternate' in sys.argv else None
acc = defaultdict(list)
meta = {}
seen = defaultdict(int)
for path in glob.glob(os.path.join(sys.argv[1], '**', '*_kernel_trace.csv'), recursive=True):
    for r in sorted(csv.DictReader(open(path)), key=lambda r_: int(r_['Start_Timestamp'])):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        if alt and alt in name:
            seen[name] += 1
            name += ' [coarse pass]' if seen[name] % 2 == 1 else ' [fine pass]'
        k = (name, int(r['Grid_Size_X']), int(r['Workgroup_Size_X']))
        acc[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        meta[k] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
tot = sum(sum(v) for v in acc.values())
print('| kernel | grid (threads) | wg | calls | mean us | min us | max us | % time | VGPR | AGPR | SGPR | LDS B | scratch |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    m = meta[k]
    print(f'| {k[0]} | {k[1]} | {k[2]} | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | {100*sum(v)/tot:.2f} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[4]} |')
