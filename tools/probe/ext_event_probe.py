#!/usr/bin/env python3
"""Probe (GPU box): does an EXTERNAL event recorded inside a captured hipGraph order a wait issued on another stream AFTER
graph.replay()?  (the mechanism behind the bucketed gradient exchange overlapped with backward, graph.py)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

dev = 'cuda'
n = 1 << 24
x = torch.zeros(n, device=dev)
a = torch.zeros(n, device=dev)
b = torch.zeros(n, device=dev)
ev = hip.GraphEvent()
side = torch.cuda.Stream()


def body(capturing=True):
    a.copy_(x * 2 + 1)            # "bucket" a is final here
    if capturing:
        ev.record_external()      # external event node
    t = a
    for _ in range(200):          # long tail that the side stream may overlap with
        t = t * 1.0001 + 0.5
    b.copy_(t)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body(False)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
ok = True
for it in range(1, 6):
    x.fill_(float(it))
    g.replay()
    with torch.cuda.stream(side):
        ev.wait()
        y = a.clone()
        e_side = torch.cuda.Event(enable_timing=True); e_side.record()
    e_main = torch.cuda.Event(enable_timing=True); e_main.record()
    torch.cuda.synchronize()
    want = 2.0 * it + 1
    got = y[0].item(), y[-1].item()
    print(f'replay {it}: side stream saw a = {got} (want {want}); side finished {"before" if e_side.elapsed_time(e_main) > 0 else "after"} the graph tail ({e_side.elapsed_time(e_main):.3f} ms earlier)')
    ok &= got == (want, want)
print('EXTERNAL_EVENT_OK' if ok else 'EXTERNAL_EVENT_BROKEN')
