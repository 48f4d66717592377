#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (``*_results.db``) into the per-kernel table that
``rocprofv3 --kernel-trace --stats`` reports: calls, total / average / min / max duration and share of GPU time.
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--csv out.csv] [--top N]"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r'\(.*\)$', '', name)          # drop the argument list
    name = name.replace('void ', '')
    return name if len(name) <= 110 else name[:107] + '...'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--csv')
    ap.add_argument('--top', type=int, default=60)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                     'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    print(f'{"kernel":<112} {"calls":>6} {"total ms":>10} {"avg us":>10} {"%":>6}')
    for i, (n, k, t, avg, mn, mx) in enumerate(rows):
        lines.append(f'"{n}",{k},{t},{avg:.1f},{100 * t / total:.2f},{mn},{mx}')
        if i < a.top:
            print(f'{short(n):<112} {k:>6} {t / 1e6:>10.3f} {avg / 1e3:>10.2f} {100 * t / total:>6.2f}')
    print(f'total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches')
    if a.csv:
        open(a.csv, 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
