#!/usr/bin/env python3
"""One-off (round 5): DESIGN.md = tools/design_front.md (current state) + Appendix A (the engineering notes of rounds 1-4, moved verbatim
from the previous DESIGN.md with their section numbers mapped to A.x).  python tools/assemble_design.py <old DESIGN.md> > DESIGN.md"""
import re
import sys

old = open(sys.argv[1]).read().split('\n')
front = open(sys.argv[2]).read().rstrip('\n')


def section(start_pat, end_pat):
    a = next(i for i, l in enumerate(old) if l.startswith(start_pat))
    b = next((i for i, l in enumerate(old) if i > a and any(l.startswith(e) for e in end_pat)), len(old))
    return old[a:b]


parts = [
    ('A.0 Kernel notes by round (the kernel table of rounds 1–4)', section('## 4. Kernels', ['### 4.1'])[1:]),
    ('A.1 One step = one hipGraph, gradients written in place, no framework kernels inside', section('### 4.1', ['### 4.2'])[1:]),
    ('A.2 SegFormerHead folded algebraically', section('### 4.2', ['### 4.3'])[1:]),
    ('A.3 The fused loss', section('### 4.3', ['### 4.5'])[1:]),
    ('A.4 Rules that came out of the profiles (rounds 1–3)', section('### 4.4', ['## 5.'])[1:]),
    ('A.5 Device-side input pipeline', section('### 4.5', ['### 4.6'])[1:]),
    ('A.6 Bilinear resizing as a matrix product (r03)', section('### 4.6', ['### 4.7'])[1:]),
    ('A.7 The eight-phase GEMM, round 4: stagger, and what the clock does with it', section('### 4.7', ['### 4.8'])[1:]),
    ('A.8 Head-dim-64 attention, round 4', section('### 4.8', ['### 4.9'])[1:]),
    ('A.9 The reference\'s default batch (4 per GPU): a step of 423 launches, second pass (r04)', section('### 4.9', ['### 4.4'])[1:]),
    ('A.10 Measurement history, rounds 1–4', section('## 5. Measurement', ['## 6.'])[1:]),
    ('A.11 What the profiles of rounds 3 and 4 said was next', section('## 9. What the profiles say is next', ['## 99'])[1:]),
]
SUBS = [(r'§\s?4\.([1-9])', r'A.\1'), (r'[Ss]ection 4\.([1-9])', r'A.\1'), (r'§\s?9\b', 'A.11'), (r'[Ss]ection 9\b', 'A.11'), (r'§\s?5\b', 'A.10'),
        (r'[Ss]ection 5\b', 'A.10'), (r'[Ss]ection 6\b', '§8'), (r'§\s?6\b', '§8'), (r'DESIGN §\s?4\.([1-9])', r'DESIGN A.\1')]
out = [front, '', '---', '', '# Appendix A — engineering notes of rounds 1–4',
       '', 'Moved here verbatim from the round-4 DESIGN.md (section numbers mapped: old §4.x → A.x, old §5 → A.10, old §9 → A.11). The figures in',
       'this appendix are those of the round that wrote them; sections 1–11 above hold the current ones.', '']
for title, body in parts:
    out.append('## ' + title)
    txt = '\n'.join(body)
    for pat, rep in SUBS:
        txt = re.sub(pat, rep, txt)
    txt = re.sub(r'^### \(round 3\)', '### (round 3)', txt, flags=re.M)
    out.append(txt.rstrip('\n'))
    out.append('')
sys.stdout.write('\n'.join(out).rstrip('\n') + '\n')
