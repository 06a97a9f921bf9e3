#!/usr/bin/env python
"""HBM bytes per launch from rocprofv3 PMC passes -> profiles/r03_pmc_traffic.json (read by bench.py's `roofline.traffic`).

    tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <shape key> [out.json]

FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units are KiB.
gfx950 correction (same guide, "HBM"): FETCH_SIZE reports half the bytes of a wide coalesced streaming read, so
hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; both raw values are kept in the file.
Shape key = the one bench.py builds: '<config>:B<objects/GPU>:N<points>:S<samples>:K<amis iters>:L<lm iters>'."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def averages(d, counter):
    agg = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row['Counter_Name'] != counter or 'pnp::' not in row.get('Kernel_Name', ''):
                    continue
                k = re.sub(r'[<(].*$', '', row['Kernel_Name'].replace('void ', '')).replace('pnp::', '')
                agg[k][0] += float(row['Counter_Value'])
                agg[k][1] += 1
    return {k: (s / n, n) for k, (s, n) in agg.items()}


def main():
    fetch, write, key = averages(sys.argv[1], 'FETCH_SIZE'), averages(sys.argv[2], 'WRITE_SIZE'), sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             'profiles', 'r03_pmc_traffic.json')
    table = json.load(open(out)) if os.path.exists(out) else {}
    rec = {}
    for k in sorted(set(fetch) & set(write)):
        f, w = fetch[k][0], write[k][0]
        rec[k] = {'FETCH_SIZE_KiB': round(f, 1), 'WRITE_SIZE_KiB': round(w, 1), 'dispatches': fetch[k][1],
                  'hbm_bytes_per_launch': round((2 * f + w) * 1024)}
    table[key] = rec
    json.dump(table, open(out, 'w'), indent=1, sort_keys=True)
    for k, v in rec.items():
        print(f'{key} {k:36s} {v}')


if __name__ == '__main__':
    main()
