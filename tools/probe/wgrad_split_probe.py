#!/usr/bin/env python3
"""GPU probe: the 3x3 weight gradient (gemm8 reduction-major form) of one shape over split-K counts.
python tools/probe/wgrad_split_probe.py B H W Cin Cout"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip
B, H, W, Cin, Cout = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (8, 80, 80, 768, 768)
P = B * H * W
x = torch.randn(P, Cin, device='cuda').to(torch.bfloat16)
dy = (torch.randn(P, Cout, device='cuda') * 1e-3).to(torch.bfloat16)
wm = (torch.randn(Cout, 9 * Cin, device='cuda') * 0.01).to(torch.bfloat16)
fl = 2.0 * P * 9 * Cin * Cout / 1e12
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t = timed(lambda: hip.conv3x3(0, x, wm, B, H, W, Cin, Cout))
print(f"[{B}x{H}x{W} {Cin}->{Cout}] fwd {t:.3f} ms ({fl / t * 1e3:.0f} TF/s); pick_splitk = {hip.pick_splitk_conv3x3(Cin, Cout, P)}")
for s in (1, 2, 3, 4, 6, 9, 12):
    t = timed(lambda: hip.conv3x3(2, x, dy, B, H, W, Cin, Cout, split_k=s))
    print(f'  wgrad split_k {s:2d}: {t:.3f} ms ({fl / t * 1e3:.0f} TF/s)')
