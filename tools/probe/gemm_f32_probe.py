#!/usr/bin/env python3
"""fp32 products (exact-parity mode / evaluate): the f32 matrix-instruction kernel against the vector FMA kernel, same process.
python tools/probe/gemm_f32_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from segmentation_factory_amd import hip


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(0, 524288, 160, 768), (0, 524288, 768, 32), (0, 16384, 160, 768), (0, 16384, 768, 32), (0, 16384, 128, 32), (0, 32768, 640, 160),
          (0, 8192, 4096, 4096), (1, 8192, 4096, 4096), (2, 4096, 4096, 8192), (1, 768, 40, 768), (0, 256, 256, 1440)]
for lay, M, N, K in shapes:
    if lay == 0:
        A, B = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda')
    elif lay == 1:
        A, B = torch.randn(M, K, device='cuda'), torch.randn(K, N, device='cuda')
    else:
        A, B = torch.randn(K, M, device='cuda'), torch.randn(K, N, device='cuda')
    res = []
    for off in (0, 1):
        hip.policy_set('gemm_f32_no_mfma', off)
        ms = timed(lambda: hip.gemm(lay, A, B, M, N, K, out_dtype=torch.float32))
        res.append(ms)
    hip.policy_set('gemm_f32_no_mfma', 0)
    fl = 2.0 * M * N * K
    print(f'layout {lay} [{M} x {K}] -> {N}: matrix pipe {res[0]:.3f} ms ({fl / res[0] / 1e9:.1f} TF/s) | vector FMA {res[1]:.3f} ms ({fl / res[1] / 1e9:.1f} TF/s)', flush=True)
