#!/usr/bin/env python
"""Summarise rocprofv3 --output-format csv counter_collection files: average counter value per dispatch for pnp:: kernels."""
import csv
import glob
import re
import sys
from collections import defaultdict

agg = defaultdict(lambda: [0.0, 0])
meta = {}
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get('Kernel_Name', '')
            if 'pnp::' not in name:
                continue
            k = re.sub(r'\(.*$', '', name.replace('void ', ''))
            a = agg[(k, row['Counter_Name'])]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
            meta[k] = (row.get('VGPR_Count'), row.get('SGPR_Count'), row.get('Scratch_Size'), row.get('LDS_Block_Size'),
                       row.get('Workgroup_Size'), row.get('Grid_Size'))
for k in sorted(meta):
    print(f'{k}: vgpr={meta[k][0]} sgpr={meta[k][1]} scratch={meta[k][2]} lds={meta[k][3]} wg={meta[k][4]} grid={meta[k][5]}')
    for (kk, c), (s, n) in sorted(agg.items()):
        if kk == k:
            print(f'    {c:28s} {s / n:20.1f}   (avg of {n} dispatches)')
