#!/usr/bin/env python3
"""Timeline of the LAST complete training step in a rocprofv3 --kernel-trace CSV (steps are delimited by adam_clip_kernel):
start offset, gap to the previous kernel, duration, kernel name.   Usage: python scripts/step_timeline.py <trace dir> [delimiter [quiet]]
With a delimiter kernel-name prefix that occurs once per step (e.g. field_jvp_kernel for the LanguageNeRF step) the window is one period
between its last two occurrences; `quiet` prints only the sums."""
import csv
import glob
import os
import sys

path = glob.glob(os.path.join(sys.argv[1], '**', '*_kernel_trace.csv'), recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('mvnerf::', '') for r in rows]
if len(sys.argv) > 2:
    occ = [i for i, n in enumerate(names) if n.startswith(sys.argv[2])]
    start, end = occ[-2], occ[-1] - 1
else:
    adam = [i for i, n in enumerate(names) if n.startswith('adam_clip')]
    start, end = adam[-3] + 1, adam[-1]
quiet = len(sys.argv) > 3
t0 = int(rows[start]['Start_Timestamp'])
total = 0.0
for i in range(start, end + 1):
    d = (int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3
    gap = (int(rows[i]['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3
    total += d
    if not quiet:
        print(f"{(int(rows[i]['Start_Timestamp']) - t0) / 1e3:9.1f} +{gap:6.1f}  {d:8.1f} us  {names[i][:70]}")
span_end = int(rows[end + 1]['Start_Timestamp']) if len(sys.argv) > 2 else int(rows[end]['End_Timestamp'])
print(f'{end - start + 1} kernels, sum of kernels {total:.1f} us, span {(span_end - t0) / 1e3:.1f} us')
