#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (--kernel-trace [--pmc ...]) into a small text/CSV table for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:110]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    # launch size (work-items of the grid's x dimension), where the view has it: a kernel launched at several sizes in one
    # process gets one row per size (round 4's normal_equations_kernel row mixed 121 C2-size launches with 27 C5-size ones)
    gcol = next((x for x in cols if re.fullmatch(r'grid(_size)?_?x', x)), None)
    rows = c.execute(f'select name, start, end{", " + gcol if gcol else ""} from kernels').fetchall()
    sizes = {}
    for r in rows:
        if gcol:
            sizes.setdefault(short(r[0]), set()).add(r[3])
    agg = {}
    for r in rows:
        name, s, e = r[0], r[1], r[2]
        key = short(name)
        if gcol and len(sizes[key]) > 1:
            key = f'{key[:96]} [grid {r[3]}]'
        a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f'# kernel-trace summary of {path}  (durations in us; {len(rows)} dispatches, total {tot / 1e3:.3f} ms)')
    if gcol is None:
        print('# (no grid-size column in this database\'s kernels view: ' + ' '.join(cols) + ')')
    print(f'{"kernel":112s} {"calls":>6s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f'{k:112s} {a[0]:6d} {a[1]:12.1f} {a[1] / a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100 * a[1] / tot:6.2f}')
    # counters, if any
    try:
        pm = c.execute('select k.name, p.counter_name, avg(p.value), count(*) from pmc_events p '
                       'join kernels k on k.event_id = p.event_id group by k.name, p.counter_name').fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print('\n# PMC counters (average per dispatch)')
        for name, cn, v, n in pm:
            if 'pnp::' in name:
                print(f'{short(name):112s} {cn:24s} {v:18.1f}  (n={n})')


if __name__ == '__main__':
    main(sys.argv[1])
