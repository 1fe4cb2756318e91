#!/usr/bin/env python3
"""Timeline of the LAST complete training step in a rocprofv3 --kernel-trace CSV (steps are delimited by adam_clip_kernel):
start offset, gap to the previous kernel, duration, kernel name.   Usage: python scripts/step_timeline.py <trace dir>"""
import csv
import glob
import os
import sys

path = glob.glob(os.path.join(sys.argv[1], '**', '*_kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('mvnerf::', '') for r in rows]
adam = [i for i, n in enumerate(names) if n.startswith('adam_clip')]
start, end = adam[-3] + 1, adam[-1]
t0 = int(rows[start]['Start_Timestamp'])
total = 0.0
for i in range(start, end + 1):
    d = (int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3
    gap = (int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3
    total += d
    print(f"{(int(rows[i]['Start_Timestamp']) - t0) / 1e3:9.1f} +{gap:6.1f}  {d:8.1f} us  {names[i][:70]}")
print(f'sum of kernels {total:.1f} us, span {(int(rows[end]["End_Timestamp"]) - t0) / 1e3:.1f} us')
