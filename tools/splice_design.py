#!/usr/bin/env python3
"""DESIGN.md = the filled front (tools/fill_design.py <tag>: sections 1-11 with the figures of profiles/<tag>_*.json) + Appendix A as it
stands in the current DESIGN.md (everything from the '---' rule in front of '# Appendix A' on):
    python tools/fill_design.py r05b > /tmp/front.md && python tools/splice_design.py /tmp/front.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cur = open(os.path.join(ROOT, 'DESIGN.md')).read()
front = open(sys.argv[1]).read().rstrip('\n')
k = cur.index('\n---\n\n# Appendix A')
open(os.path.join(ROOT, 'DESIGN.md'), 'w').write(front + '\n' + cur[k:])
print('DESIGN.md:', len(front.split('\n')), 'front lines +', len(cur[k:].split('\n')), 'appendix lines')
