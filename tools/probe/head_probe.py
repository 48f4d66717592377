#!/usr/bin/env python3
"""GPU probe: the kernels of the folded SegFormerHead's 768-wide stride-4 map at cfg2 shapes (batch 128 by default), one by one
through the C ABI: ms per launch and GB/s of algorithmic bytes.  SEGFAC_HIP_LIB=<other .so> selects a build variant.
    python tools/probe/head_probe.py [batch]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = W = 128
C, C1, NC = 768, 32, 160
M = B * H * W
bf = torch.bfloat16
dev = 'cuda'
x1 = torch.randn(M, C1, device=dev).to(bf)
G1 = (torch.randn(C, C1, device=dev) * 0.1).to(bf)
ts = [torch.randn(B * (H >> i) * (W >> i), C, device=dev).to(bf) for i in (1, 2, 3)]
t = timed(lambda: hip.fuse_map_248(x1, G1, *ts, B, H, W))
alg = 2 * (M * C + M * C1 + sum(x.numel() for x in ts))
print(f'fuse_map_248            {t:7.3f} ms  {alg / t / 1e6:7.0f} GB/s')
fused, sums = hip.fuse_map_248(x1, G1, *ts, B, H, W)
dy = (torch.randn(M, C, device=dev) * 1e-3).to(bf)
t = timed(lambda: hip.bilinear_bwd_248(dy, B, H, W, C))
print(f'bilinear_bwd_248        {t:7.3f} ms  {alg / t / 1e6:7.0f} GB/s')
dyc = torch.zeros(M, NC, device=dev, dtype=bf); dyc[:, :150] = (torch.randn(M, 150, device=dev) * 1e-3).to(bf)
wc = torch.zeros(NC, C, device=dev, dtype=bf); wc[:150] = (torch.randn(150, C, device=dev) * 0.03).to(bf)
mean, rstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
drop = torch.ones(B, C, device=dev)
t = timed(lambda: hip.bn_cls_bwd_full(dyc, wc, fused, mean, rstd, gam, bet, 1, drop, H * W, False, x1=x1))
print(f'bn_cls_bwd_full (2 pass){t:7.3f} ms  {(3 * M * C * 2 + 2 * M * NC * 2) / t / 1e6:7.0f} GB/s')
sc, sh = torch.ones(B, C, device=dev), torch.zeros(B, C, device=dev)
bias = torch.zeros(NC, device=dev)
t = timed(lambda: hip.gemm_pro(0, fused, wc, M, NC, C, sc, sh, H * W, 1, bias=bias))
print(f'classifier fwd (pro)    {t:7.3f} ms  {(M * C * 2 + M * NC * 2) / t / 1e6:7.0f} GB/s')
t = timed(lambda: hip.gemm(1, dy, G1, M, C1, C))
print(f'stage-1 dgrad 768->32   {t:7.3f} ms  {(M * C * 2 + M * C1 * 2) / t / 1e6:7.0f} GB/s')
