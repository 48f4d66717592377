#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and WRITE_SIZE
do not fit one pass; both are reported in KB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B, so it is doubled).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --batch 256 --out profiles/pmc_traffic_b256.json

Per kernel name the LARGEST launch is kept (the full-size in-graph one).  The JSON is stamped with the hash of the kernel
sources, one hash per reported kernel (bench.kernel_source_hash); bench.py reports a kernel's `traffic` only when ITS hash matches the
library it is running.  (bench.py started under rocprofv3 must not spawn its child legs: --no-extra-legs; it also skips them by itself
when it sees the profiler's environment.)"""
import argparse
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_counter(dirname, counter):
    best = {}
    files = glob.glob(os.path.join(dirname, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no *counter_collection.csv under {dirname}')
    for f in files:
        per_dispatch = {}
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                key = (row.get('Dispatch_Id'), row['Kernel_Name'])
                per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row['Counter_Value'])     # summed over XCDs / instances
        for (_, name), v in per_dispatch.items():
            name = re.sub(r'\(.*\)$', '', name).replace('void ', '').strip()
            best[name] = max(best.get(name, 0.0), v)
    return best


def main():
    from bench import kernel_source_hash
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir')
    ap.add_argument('write_dir')
    ap.add_argument('--batch', type=int, required=True)
    ap.add_argument('--out', required=True)
    a = ap.parse_args()
    fetch, write = read_counter(a.fetch_dir, 'FETCH_SIZE'), read_counter(a.write_dir, 'WRITE_SIZE')
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        rd = int(fetch.get(name, 0.0) * 1024 * 2)           # KB -> bytes, x2 (gfx950 correction)
        wr = int(write.get(name, 0.0) * 1024)
        if rd + wr >= 64 << 20:                              # launches that move >= 64 MB
            kernels[name] = {'read_bytes': rd, 'write_bytes': wr, 'total_bytes': rd + wr}
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py defaults (cfg2); KB -> bytes; FETCH_SIZE doubled '
                   '(gfx950 correction, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are included; per kernel name the largest '
                   'launch (= the full-size in-graph one)',
           'batch': a.batch, 'source_hashes': kernel_source_hash(), 'kernels': kernels}
    with open(a.out, 'w') as fh:
        json.dump(out, fh, indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['total_bytes'])[:25]:
        print(f"{k[:100]:<100} {v['read_bytes'] / 1e9:8.3f} GB read {v['write_bytes'] / 1e9:8.3f} GB written")


if __name__ == '__main__':
    main()
