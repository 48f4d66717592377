#!/usr/bin/env python3
"""Diagnostic: the ORDERED kernel sequence of the last captured-step replay in a rocprofv3 rocpd database (name, duration, gap to the
previous kernel's end), to see which launches of a small-batch step are launch-floor bound and where they sit.
    python tools/probe/step_sequence.py gpurun_out/prof/stats_results.db [marker-kernel-substring] > sequence.txt"""
import re
import sqlite3
import sys

db = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else 'agc_adamw_kernel'
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
rows = c.execute('select name, start, end, duration from kernels order by start').fetchall()
ends = [i for i, r in enumerate(rows) if marker in r[0]]
if len(ends) < 2:
    print('marker not found twice; columns:', cols)
    sys.exit(1)
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0 = step[0][1]
print(f'# {len(step)} launches, {(step[-1][2] - t0) / 1e3:.1f} us from first start to last end; sum of durations {sum(r[3] for r in step) / 1e3:.1f} us')
prev = t0
for n, s, e, d in step:
    n = re.sub(r'\(.*\)$', '', n).replace('void ', '')
    print(f'{(s - t0) / 1e3:9.1f} {d / 1e3:7.1f} {(s - prev) / 1e3:6.1f}  {n[:120]}')
    prev = e
