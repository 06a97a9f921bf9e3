#!/usr/bin/env python
"""Kernel sequence of ONE step out of a rocprofv3 rocpd database: the dispatches between two consecutive occurrences of a marker
kernel, in start order, with start offset, duration and the gap to the previous kernel's end (us).

    python tools/rocprof_sequence.py /tmp/prof/c4_results.db adaptive_delta_kernel [which=-3]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name[:100]


def main(path, marker, which=-3):
    c = sqlite3.connect(path)
    rows = sorted(c.execute('select name, start, end from kernels').fetchall(), key=lambda r: r[1])
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    a, b = idx[which], idx[which + 1]
    t0, prev_end = rows[a][1], None
    print(f'# dispatches {a}..{b - 1} of {len(rows)}: one step = {(rows[b][1] - t0) / 1e3:.1f} us from marker to marker')
    busy = 0.0
    for name, s, e in rows[a:b]:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        busy += (e - s) / 1e3
        print(f'{(s - t0) / 1e3:9.2f}  dur {(e - s) / 1e3:7.2f}  gap {gap:6.2f}  {short(name)}')
        prev_end = max(e, prev_end or e)
    print(f'# {b - a} kernels, {busy:.1f} us busy')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -3)
