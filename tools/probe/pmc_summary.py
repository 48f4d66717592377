#!/usr/bin/env python3
"""One line per kernel from the counter sets of tools/probe/pmc_mfma.sh / pmc_g8.sh: python tools/probe/pmc_summary.py gpurun_out/<tag>
(reads <tag>/summary.txt = 'kernel counter average n=..' lines and <tag>/time.txt = the unprofiled probe lines).
MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); VALU-busy = 4 * SQ_ACTIVE_INST_VALU / the same; LDS-busy =
16 * SQ_ACTIVE_INST_LDS / the same; wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; VALU per MFMA = (SQ_INSTS_VALU - SQ_INSTS_MFMA) /
SQ_INSTS_MFMA; FETCH = 2 * 1024 * FETCH_SIZE bytes (the gfx950 correction of MI355X_MICROARCH.md); cycles per launch = GRBM_GUI_ACTIVE / 8
(the counter sums the eight XCDs)."""
import collections
import os
import re
import sys

d = sys.argv[1]
acc = collections.defaultdict(dict)
for ln in open(os.path.join(d, 'summary.txt')):
    m = re.match(r'(.+?)\s{2,}(\w+)\s+(\d+)\s+n=(\d+)', ln.rstrip())
    if m:
        acc[m.group(1).strip()][m.group(2)] = float(m.group(3))
for ln in open(os.path.join(d, 'time.txt')):
    if 'TF/s' in ln or 'ms' in ln:
        print('# ' + ln.strip())
for k, c in sorted(acc.items()):
    if 'GRBM_GUI_ACTIVE' not in c:
        continue
    cyc = c['GRBM_GUI_ACTIVE'] / 8
    simd = cyc * 1024
    f = lambda name: c.get(name, float('nan'))
    hit = f('TCC_HIT_sum') / max(f('TCC_HIT_sum') + f('TCC_MISS_sum'), 1)
    vm = (f('SQ_INSTS_VALU') - f('SQ_INSTS_MFMA')) / f('SQ_INSTS_MFMA') if f('SQ_INSTS_MFMA') else float('nan')
    print(f'{k[:58]:58s} cycles/launch {cyc / 1e6:7.3f} M | MFMA-busy {100 * f("SQ_VALU_MFMA_BUSY_CYCLES") / simd:5.1f} % | VALU-busy '
          f'{100 * 4 * f("SQ_ACTIVE_INST_VALU") / simd:5.1f} % | LDS-busy {100 * 16 * f("SQ_ACTIVE_INST_LDS") / simd:5.1f} % | wait_inst '
          f'{100 * f("SQ_WAIT_INST_ANY") / f("SQ_WAVE_CYCLES"):5.1f} % | VALU per MFMA {vm:5.2f} | FETCH {2 * 1024 * f("FETCH_SIZE") / 1e9:5.2f} GB | L2 hit {100 * hit:3.0f} %')
