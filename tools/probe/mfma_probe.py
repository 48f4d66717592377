#!/usr/bin/env python3
"""GPU probe for the two MFMA-bound kernels of the multi-GPU BASELINE configs: the UPerHead 3 x 3 implicit GEMM (gemm8_kernel, cfg3 / cfg5)
and the head-dim-64 attention of MiT-B2 at 1024 x 2048 (cfg4 stage 1: 131072 queries x 2048 keys).  Random operands.
python tools/probe/mfma_probe.py [conv|attn|all] [iters]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip


def timed(fn, n):
    # warm up for >= 60 ms: the first loop of a fresh process runs while the shader clock is still ramping (r05: the first figure of a
    # process read 0.77 ms for a kernel whose every later figure was 0.62 -- the round's earlier "held clock 1.59 GHz" was that ramp)
    fn(); torch.cuda.synchronize()
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record()
    for _ in range(200):
        fn()
        w1.record(); w1.synchronize()
        if w0.elapsed_time(w1) >= 60.0:
            break
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def ab_settings():
    """PROBE_AB='g8_stagger=0;g8_stagger=1' -> the dispatch-policy settings (csrc/policy.h) to alternate between, in one process."""
    v = os.environ.get('PROBE_AB')
    return v.split(';') if v else [None]


def apply_setting(setting):
    if setting:
        for kv in setting.split(','):
            k, val = kv.split('=')
            hip.policy_set(k.strip(), int(val))


def conv(n):
    B, H, W, Cin, Cout = 32, 128, 128, 3072, 768
    P = B * H * W
    x = torch.randn(P, Cin, device='cuda').to(torch.bfloat16)
    dy = (torch.randn(P, Cout, device='cuda') * 1e-3).to(torch.bfloat16)
    wm = (torch.randn(Cout, 9 * Cin, device='cuda') * 0.01).to(torch.bfloat16)
    wt = (torch.randn(Cin, 9 * Cout, device='cuda') * 0.01).to(torch.bfloat16)
    fl = 2.0 * P * 9 * Cin * Cout / 1e12
    xq, sx = hip.quant_tensor_fp8(x); wq, sw = hip.quant_rows_fp8(wm)
    orders = ab_settings()
    for rnd in range(2 if len(orders) > 1 else 1):
        for od in orders:
            apply_setting(od)
            t0 = timed(lambda: hip.conv3x3(0, x, wm, B, H, W, Cin, Cout), n)
            t1 = timed(lambda: hip.conv3x3(1, dy, wt, B, H, W, Cin, Cout), n)
            t2 = timed(lambda: hip.conv3x3(2, x, dy, B, H, W, Cin, Cout, split_k=hip.pick_splitk_conv3x3(Cin, Cout, P)), n)
            f0 = timed(lambda: hip.conv3x3_fp8(0, xq, sx, wq, sw, B, H, W, Cin, Cout), n)
            print(f'conv3x3 [{B}x{H}x{W} {Cin}->{Cout}] order {od} {fl:.2f} TFLOP | bf16 fwd {t0:.2f} ms ({fl / t0 * 1e3:.0f} TF/s) dgrad {t1:.2f} ({fl / t1 * 1e3:.0f}) '
                  f'wgrad {t2:.2f} ({fl / t2 * 1e3:.0f}) | fp8 fwd {f0:.2f} ({fl / f0 * 1e3:.0f})', flush=True)
    # plain product of the same size class through the same kernel (no gather): [P x 3072] x [3072 -> 3072]^T
    M, N, K = 65536, 3072, 3072
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * 0.02).to(torch.bfloat16)
    hip.policy_set('gemm8_linear', 1)
    try:
        tl = timed(lambda: hip.gemm(0, a, w, M, N, K), n)
        print(f'linear  [{M} x {K}] -> {N}: {tl:.3f} ms ({2.0 * M * N * K / tl / 1e9:.0f} TF/s)', flush=True)
    except Exception as e:                                   # the probe must not die on an API difference
        print('linear probe skipped:', e)


def attn(n):
    B, heads, N, Nkv, hd = 8, 1, 131072, 2048, 64
    scale = hd ** -0.5
    q = torch.randn(B * N, heads * hd, device='cuda').to(torch.bfloat16)
    k = torch.randn(B * Nkv, heads * hd, device='cuda').to(torch.bfloat16)
    v = torch.randn(B * Nkv, heads * hd, device='cuda').to(torch.bfloat16)
    do = torch.randn(B * N, heads * hd, device='cuda').to(torch.bfloat16)
    o, lse = hip.attention_fwd(q, k, v, B, heads, N, Nkv, hd, scale)
    dk = torch.empty_like(k); dv = torch.empty_like(v)
    ff = 4.0 * B * heads * N * Nkv * hd / 1e12
    variants = ab_settings()
    for rnd in range(3 if len(variants) > 1 else 1):              # interleaved rounds in one process (same device, same clocks)
        for var in variants:
            apply_setting(var)
            t0 = timed(lambda: hip.attention_fwd(q, k, v, B, heads, N, Nkv, hd, scale), n)
            t1 = timed(lambda: hip.attention_bwd(q, k, v, o, do, lse, B, heads, N, Nkv, hd, scale, dk, dv), n)
            print(f'attention [{B} x {heads} x {N} x {Nkv} x {hd}] variant {var} fwd {ff:.2f} TFLOP {t0:.3f} ms ({ff / t0 * 1e3:.0f} TF/s) | bwd {2.5 * ff:.2f} TFLOP '
                  f'{t1:.3f} ms ({2.5 * ff / t1 * 1e3:.0f} TF/s)', flush=True)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    if what in ('conv', 'all'):
        conv(n)
    if what in ('attn', 'all'):
        attn(n)
