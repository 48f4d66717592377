#!/usr/bin/env python3
"""Rewrite the switch list inside include/segfac.h's "Dispatch policy" comment from csrc/policy.h (the single table): python tools/gen_policy_doc.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pol = open(os.path.join(ROOT, 'segmentation_factory_amd', 'csrc', 'policy.h')).read()
rows = re.findall(r'X\((\w+), "(SEGFAC_\w+)", (-?\d+), "((?:[^"\\]|\\.)*)"\)', pol)
doc = '\n'.join(f' *   {env:<28} {text}' + (f'  (default {d})' if d != '0' else '') for f, env, d, text in rows)
p = os.path.join(ROOT, 'include', 'segfac.h')
s = open(p).read()
a = s.index(' *   SEGFAC_GEMM_NO_BIG')
b = s.index(' *\n * segf_policy_count / segf_policy_describe enumerate the table')
s = s[:a] + doc + '\n' + s[b:]
open(p, 'w').write(s)
print(len(rows), 'switches documented')
