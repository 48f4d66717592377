#!/usr/bin/env python3
"""GPU probe: the UPerHead 3x3 convolutions of BASELINE cfg3 / cfg5 as bf16 and as fp8 implicit GEMMs (forward, data gradient) and the
bf16 weight gradient: time per launch and TFLOP/s.  python tools/probe/fp8_conv_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (B, H, W, Cin, Cout) in [(8, 160, 160, 3072, 768), (8, 160, 160, 768, 768), (8, 80, 80, 768, 768), (32, 128, 128, 3072, 768), (32, 128, 128, 768, 768)]:
    P = B * H * W
    x = torch.randn(P, Cin, device='cuda').to(torch.bfloat16)
    dy = (torch.randn(P, Cout, device='cuda') * 1e-3).to(torch.bfloat16)
    wm = (torch.randn(Cout, 9 * Cin, device='cuda') * 0.01).to(torch.bfloat16)
    wt = (torch.randn(Cin, 9 * Cout, device='cuda') * 0.01).to(torch.bfloat16)
    fl = 2.0 * P * 9 * Cin * Cout / 1e12
    t0 = timed(lambda: hip.conv3x3(0, x, wm, B, H, W, Cin, Cout))
    t1 = timed(lambda: hip.conv3x3(1, dy, wt, B, H, W, Cin, Cout))
    t2 = timed(lambda: hip.conv3x3(2, x, dy, B, H, W, Cin, Cout, split_k=hip.pick_splitk_conv3x3(Cin, Cout, P)))
    xq, sx = hip.quant_tensor_fp8(x); wq, sw = hip.quant_rows_fp8(wm)
    gq, sg = hip.quant_tensor_fp8(dy, e5m2=True); wtq, swt = hip.quant_rows_fp8(wt)
    f0 = timed(lambda: hip.conv3x3_fp8(0, xq, sx, wq, sw, B, H, W, Cin, Cout))
    f1 = timed(lambda: hip.conv3x3_fp8(1, gq, sg, wtq, swt, B, H, W, Cin, Cout))
    f2 = timed(lambda: hip.conv3x3_fp8_wgrad(xq, sx, gq, sg, B, H, W, Cin, Cout)) if hip.conv3x3_fp8_wgrad_supported(B, H, W, Cin, Cout) else float('nan')
    q0 = timed(lambda: hip.quant_tensor_fp8(x))
    q1 = timed(lambda: hip.quant_rows_fp8(wm))
    print(f'[{B}x{H}x{W} {Cin}->{Cout}] {fl:.2f} TFLOP | bf16 fwd {t0:.2f} ms ({fl / t0 * 1e3:.0f} TF/s) dgrad {t1:.2f} ({fl / t1 * 1e3:.0f}) wgrad {t2:.2f} ({fl / t2 * 1e3:.0f}) | '
          f'fp8 fwd {f0:.2f} ({fl / f0 * 1e3:.0f}) dgrad {f1:.2f} ({fl / f1 * 1e3:.0f}) wgrad {f2:.2f} ({fl / f2 * 1e3:.0f}) | quant x {q0:.2f} ms, quant w {q1:.3f} ms')
