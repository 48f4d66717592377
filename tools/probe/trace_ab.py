#!/usr/bin/env python3
"""Align the kernel traces of two runs of the same graphed step (rocprofv3 --kernel-trace csv) dispatch by dispatch: where do the runs
differ -- in the kernels that changed, in the kernels that did not, or in the gaps between them?   trace_ab.py default.csv other.csv"""
import collections
import csv
import re
import sys


def last_step(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', ''),
                     int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z']))))
    rows.sort()
    first = next(r[2] for r in rows if 'ce_dice' in r[2] and 'fwd' in r[2])
    marks = [i for i, r in enumerate(rows) if r[2] == first]
    # one loss forward per step: two consecutive marks bracket a step (loss forward of step n .. of step n + 1); the LAST bracket of the usual
    # length (warm-up passes and the end of the trace give shorter / longer ones)
    assert len(marks) >= 3, 'no step marks'
    lens = [marks[i + 1] - marks[i] for i in range(len(marks) - 1)]
    usual = collections.Counter(lens).most_common(1)[0][0]
    i = max(i for i, n in enumerate(lens) if n == usual)
    return rows[marks[i]:marks[i + 1]]


a, b = last_step(sys.argv[1]), last_step(sys.argv[2])
print(f'dispatches per step: {len(a)} / {len(b)}')
span = lambda s: (s[-1][1] - s[0][0]) / 1e6
dur = lambda s: sum(e - st for st, e, _, _ in s) / 1e6
print(f'step span      {span(a):9.3f} ms / {span(b):9.3f} ms')
print(f'kernel time    {dur(a):9.3f} ms / {dur(b):9.3f} ms')
print(f'gaps           {span(a) - dur(a):9.3f} ms / {span(b) - dur(b):9.3f} ms')
if len(a) != len(b):
    sys.exit('different dispatch counts: not aligned')
moved = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
same_a = same_b = 0.0
for i, (x, y) in enumerate(zip(a, b)):
    if x[2] != y[2]:
        k = (x[2][:52], y[2][:52], x[3], y[3])
        m = moved[k]
        m[0] += 1
        m[1] += (x[1] - x[0]) / 1e3
        m[2] += (y[1] - y[0]) / 1e3
        if i + 1 < len(a):       # the kernel after it: gap in front of it and its own duration
            m[3] += (a[i + 1][0] - x[1]) / 1e3
            m[4] += (b[i + 1][0] - y[1]) / 1e3
            m[5] += (a[i + 1][1] - a[i + 1][0]) / 1e3
            m[6] += (b[i + 1][1] - b[i + 1][0]) / 1e3
    else:
        same_a += (x[1] - x[0]) / 1e6
        same_b += (y[1] - y[0]) / 1e6
print(f'kernels with the same name in both runs: {same_a:9.3f} ms / {same_b:9.3f} ms')
print('kernels that differ (per step): count | us in run 1 -> run 2 | gap after | the next kernel')
ta = tb = 0.0
for k, m in sorted(moved.items(), key=lambda kv: kv[1][2] - kv[1][1]):
    ta += m[1]; tb += m[2]
    print(f'  {k[0]:52s} wgs {k[2]:6d} -> {k[1]:52s} wgs {k[3]:6d} x{m[0]:3d}: {m[1] / m[0]:8.1f} -> {m[2] / m[0]:8.1f} us | gap {m[3] / m[0]:6.1f} -> {m[4] / m[0]:6.1f} | next {m[5] / m[0]:8.1f} -> {m[6] / m[0]:8.1f}')
print(f'differing kernels total: {ta / 1e3:9.3f} ms -> {tb / 1e3:9.3f} ms')
