#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per (kernel, grid size): mean per dispatch.
Usage: python scripts/summarize_pmc.py gpurun_out/pmc_r01a > profiles/r01_pmc_summary.md"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in sorted(glob.glob(os.path.join(root, '*', '*', '*_counter_collection.csv'))):
    for row in csv.DictReader(open(path)):
        k = row['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        key = (k, int(row['Grid_Size']) if 'Grid_Size' in row else int(row.get('Grid_Size_X', 0)))
        acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
print('| kernel | grid | counter | mean/dispatch | dispatches |')
print('|---|---|---|---|---|')
for key in sorted(acc):
    for c, v in sorted(acc[key].items()):
        print(f'| {key[0]} | {key[1]} | {c} | {sum(v)/len(v):.6g} | {len(v)} |')
