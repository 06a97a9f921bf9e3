#!/usr/bin/env python
"""HBM bytes per launch from rocprofv3 PMC passes -> profiles/rNN_pmc_traffic.json (read by bench.py's `roofline.traffic`).

    tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <shape key> [out.json]

FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units are KiB.
gfx950 correction (same guide, "HBM"): FETCH_SIZE reports half the bytes of a wide coalesced streaming read, so
hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; both raw values are kept in the file.
Shape key = the one bench.py builds: '<config>:B<objects/GPU>:N<points>:S<samples>:K<amis iters>:L<lm iters>'.

Dispatches are grouped by kernel name AND launch size (Grid_Size x Workgroup_Size of the counter CSV): one process may launch
the same kernel at several sizes (round 4: bench.py's `roofline.large` leg launched the C5-size Jacobian sweep next to the
121 C2-size ones, and the per-name average -- 135 MB -- was attached to the C2 kernel whose own traffic is 60 MB).  The
record of a kernel is the group with the most dispatches (the shape's own launches); every group is kept under `by_launch`,
and a kernel launched at more than one size says so in `mixed_launch_sizes`."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def kernel_base(name):
    return re.sub(r'[<(].*$', '', name.replace('void ', '')).replace('pnp::', '')


def groups(d, counter):
    """{kernel: {'<grid>x<workgroup>': (mean counter value, dispatches)}} for the pnp:: kernels of one pass."""
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row['Counter_Name'] != counter or 'pnp::' not in row.get('Kernel_Name', ''):
                    continue
                launch = f"{row.get('Grid_Size', '?')}x{row.get('Workgroup_Size', '?')}"
                a = agg[kernel_base(row['Kernel_Name'])][launch]
                a[0] += float(row['Counter_Value'])
                a[1] += 1
    return {k: {g: (s / n, n) for g, (s, n) in v.items()} for k, v in agg.items()}


def records(fetch, write):
    rec = {}
    for k in sorted(set(fetch) & set(write)):
        by = {}
        for g in sorted(set(fetch[k]) & set(write[k])):
            f, w = fetch[k][g][0], write[k][g][0]
            by[g] = {'FETCH_SIZE_KiB': round(f, 1), 'WRITE_SIZE_KiB': round(w, 1), 'dispatches': fetch[k][g][1],
                     'hbm_bytes_per_launch': round((2 * f + w) * 1024)}
        if not by:
            continue
        main = max(by, key=lambda g: (by[g]['dispatches'], -by[g]['hbm_bytes_per_launch']))
        rec[k] = dict(by[main], launch=main)
        if len(by) > 1:
            rec[k]['mixed_launch_sizes'] = True
            rec[k]['by_launch'] = by
    return rec


def main():
    fetch, write, key = groups(sys.argv[1], 'FETCH_SIZE'), groups(sys.argv[2], 'WRITE_SIZE'), sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             'profiles', 'r05_pmc_traffic.json')
    table = json.load(open(out)) if os.path.exists(out) else {}
    rec = records(fetch, write)
    table[key] = rec
    json.dump(table, open(out, 'w'), indent=1, sort_keys=True)
    for k, v in rec.items():
        print(f'{key} {k:36s} { {a: b for a, b in v.items() if a != "by_launch"} }')


if __name__ == '__main__':
    main()
