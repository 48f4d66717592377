"""Depthwise 7x7 (ConvNeXt) forward / backward at the stage shapes of cfg5 (convnextv2_large, 640^2, batch 8) and cfg3
(ConvNeXt-T, 512^2, batch 32): microseconds and the rate against the tensors' bytes (x in + y out; bwd: x, dy in, dx out)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, shapes in (('cfg5', [(8, 160, 160, 192), (8, 80, 80, 384), (8, 40, 40, 768), (8, 20, 20, 1536)]),
                     ('cfg3', [(32, 128, 128, 96), (32, 64, 64, 192), (32, 32, 32, 384), (32, 16, 16, 768)])):
    for B, H, W, C in shapes:
        x = torch.randn(B * H * W, C, device='cuda').to(torch.bfloat16)
        dy = torch.randn_like(x)
        wt = torch.randn(49, C, device='cuda')
        b = torch.randn(C, device='cuda')
        f = t(lambda: hip.dwconv7x7_fwd(x, wt, b, B, H, W, C))
        g = t(lambda: hip.dwconv7x7_bwd(x, wt, dy, B, H, W, C))
        mb = x.numel() * 2 / 1e6
        macs = x.numel() * 49
        print(f'{name} [{B},{H},{W},{C}] {mb:6.1f} MB/tensor  fwd {f:7.1f} us = {2 * mb / f:5.2f} TB/s, {macs / f / 1e6:5.2f} TMAC/s   bwd (dx + dw) {g:7.1f} us', flush=True)
