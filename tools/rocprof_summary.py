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
    rows = c.execute('select name, start, end from kernels').fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f'# kernel-trace summary of {path}  (durations in us; {len(rows)} dispatches, total {tot / 1e3:.3f} ms)')
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
