#!/usr/bin/env python3
"""Instruction mix of the hottest loop of each kernel in a hipcc -S dump: python tools/probe/isa_loop.py file.s [name filter]
(the loop = the backward branch whose body holds the most MFMAs; mnemonics grouped into MFMA / VALU / transcendental / SALU / LDS / VMEM / waits)."""
import collections
import re
import sys


def classify(m):
    if m.startswith('v_mfma'):
        return 'mfma'
    if m.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')):
        return 'trans'
    if m.startswith('v_'):
        return 'valu'
    if m.startswith('ds_'):
        return 'lds'
    if m.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if m.startswith(('s_waitcnt', 's_barrier', 's_nop', 's_setprio', 's_sleep')):
        return m.split()[0]
    if m.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    src = open(sys.argv[1]).read().split('\n')
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    funcs, cur = {}, None
    for ln in src:
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif cur is not None:
            if ln.startswith('\t.amdhsa_kernel') or ln.startswith('.Lfunc_end'):
                cur = None
            else:
                funcs[cur].append(ln)
    for name, body in funcs.items():
        if filt and filt not in name:
            continue
        labels = {}
        insts = []
        for ln in body:
            lm = re.match(r'^(\.LBB\d+_\d+):', ln)
            if lm:
                labels[lm.group(1)] = len(insts)
                continue
            t = ln.strip()
            if not t or t.startswith((';', '.', '//')):
                continue
            insts.append(t.split(';')[0].strip())
        best = None
        for i, ins in enumerate(insts):
            bm = re.match(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', ins) or re.match(r's_branch\s+(\.LBB\d+_\d+)', ins)
            if bm and bm.group(1) in labels and labels[bm.group(1)] <= i:
                seg = insts[labels[bm.group(1)]:i + 1]
                nm = sum(s.startswith('v_mfma') for s in seg)
                if best is None or nm > best[0]:
                    best = (nm, seg)
        if best is None:
            continue
        nm, seg = best
        c = collections.Counter(classify(s) for s in seg)
        moves = sum(s.startswith(('v_mov', 'v_accvgpr')) for s in seg)
        top = collections.Counter(s.split()[0] for s in seg if s.startswith('v_') and not s.startswith('v_mfma')).most_common(8)
        print(f'{name[:90]}\n  loop of {len(seg)} instructions: ' + ', '.join(f'{k} {v}' for k, v in sorted(c.items())) +
              f' | v_mov/accvgpr {moves} | VALU per MFMA {(c["valu"] + c["trans"]) / max(nm, 1):.2f}\n  top VALU: {top}')


if __name__ == '__main__':
    main()
