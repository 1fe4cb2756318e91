#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): calls, mean/min/max duration.
Usage: python scripts/summarize_trace.py <dir-with-*_kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(list)
meta = {}
for path in glob.glob(os.path.join(sys.argv[1], '**', '*_kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(path)):
        k = (r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', ''), int(r['Grid_Size_X']), int(r['Workgroup_Size_X']))
        acc[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        meta[k] = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
tot = sum(sum(v) for v in acc.values())
print('| kernel | grid (threads) | wg | calls | mean us | min us | max us | % time | VGPR | AGPR | SGPR | LDS B | scratch |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    m = meta[k]
    print(f'| {k[0]} | {k[1]} | {k[2]} | {len(v)} | {sum(v)/len(v)/1e3:.2f} | {min(v)/1e3:.2f} | {max(v)/1e3:.2f} | {100*sum(v)/tot:.2f} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[4]} |')
