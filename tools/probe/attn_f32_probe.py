"""fp32 attention forward (what `evaluate` runs) on the f32 matrix instruction against the vector kernel (policy attn_f32_no_mfma), at the
shapes of SegFormer-B0 512^2 (batch 1 and 32) and MiT-B2 1024 x 2048 (batch 1): microseconds per call, same process, same operands, and
the largest difference between the two outputs.   python tools/probe/attn_f32_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip, functional as Fh

SHAPES = [(1, 1, 16384, 256, 32), (1, 2, 4096, 256, 32), (1, 5, 1024, 256, 32), (1, 8, 256, 256, 32),
          (32, 1, 16384, 256, 32), (32, 2, 4096, 256, 32), (32, 5, 1024, 256, 32), (32, 8, 256, 256, 32),
          (1, 1, 131072, 2048, 64), (1, 2, 32768, 2048, 64), (1, 5, 8192, 2048, 64), (1, 8, 2048, 2048, 64)]
for (B, heads, N, Nkv, hd) in SHAPES:
    C = heads * hd
    q = torch.randn(B * N, C, device='cuda')
    kv = torch.randn(B * Nkv, 2 * C, device='cuda')
    res = []
    with torch.no_grad():
        for v in (1, 0):
            hip.policy_set('attn_f32_no_mfma', v)
            for _ in range(2):
                o = Fh.attention(q, kv, B, N, Nkv, heads)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                o = Fh.attention(q, kv, B, N, Nkv, heads)
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 10 * 1e3, o.clone()))
    hip.policy_set('attn_f32_no_mfma', 0)
    fl = 4.0 * B * heads * N * Nkv * hd
    d = (res[0][1] - res[1][1]).abs().max().item()
    print(f'[{B} x {heads} x {N} x {Nkv} x {hd}] vector {res[0][0]:9.1f} us ({fl / res[0][0] / 1e6:6.1f} TF/s) | f32 MFMA {res[1][0]:9.1f} us ({fl / res[1][0] / 1e6:6.1f} TF/s) | max |diff| {d:.2e}')
