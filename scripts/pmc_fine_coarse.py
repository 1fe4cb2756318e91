#!/usr/bin/env python3
"""Per-launch counters of the two field passes of a bench step from the separate --pmc passes of scripts/pmc_split.sh: the dispatches
of one kernel alternate coarse (8192 tiles), fine (16384 tiles) in time order, so the fine pass is every second dispatch.
Usage: python scripts/pmc_fine_coarse.py gpurun_out/prof_r03/pmc_split field_eval_split16 > profiles/r03_pmc_fine_vs_coarse.md"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, pattern = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [[], []])
for path in sorted(glob.glob(os.path.join(root, '*', '*', '*_counter_collection.csv'))):
    rows = [r for r in csv.DictReader(open(path)) if pattern in r['Kernel_Name']]
    by_counter = defaultdict(list)
    for r in rows:
        by_counter[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for c, v in by_counter.items():
        v.sort()
        for i, (_, val) in enumerate(v):
            acc[c][i % 2].append(val)
print('| counter | fine launch (16384 tiles) | coarse launch (8192 tiles) |')
print('|---|---|---|')
for c in sorted(acc):
    coarse, fine = acc[c]
    print(f'| {c} | {sum(fine)/len(fine):.6g} | {sum(coarse)/len(coarse):.6g} |')
