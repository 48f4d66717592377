#!/usr/bin/env python3
"""Per-kernel microbenchmarks at the cfg2 (SegFormer-B0, 512x512, 150 classes) shapes, through the C ABI.
Prints achieved GB/s (algorithmic bytes) or TFLOP/s per op.  Usage: python tools/kbench.py [--batch 16] [--only name,...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from segmentation_factory_amd import hip  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters   # ms


def report(name, ms, nbytes=None, flops=None):
    s = f'{name:<44} {ms * 1e3:9.1f} us'
    if nbytes:
        s += f'  {nbytes / ms / 1e6:8.0f} GB/s'
    if flops:
        s += f'  {flops / ms / 1e9:8.1f} TFLOP/s'
    print(s, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--only', default='')
    a = ap.parse_args()
    only = set(a.only.split(',')) if a.only else None
    B, dev, bf = a.batch, 'cuda', torch.bfloat16
    want = lambda n: only is None or n in only   # noqa: E731

    def rnd(*shape, dtype=bf):
        return torch.randn(*shape, device=dev, dtype=torch.float32).to(dtype)

    if want('gemm'):
        # (name, layout, M, N, K)
        M1 = B * 128 * 128
        shapes = [('fuse fwd  y=xW^T', 0, M1, 768, 3072), ('fuse dx   dy W', 1, M1, 3072, 768), ('fuse dW   dy^T x', 2, 768, 3072, M1),
                  ('pred fwd', 0, M1, 150, 768), ('pred dx', 1, M1, 768, 150), ('pred dW', 2, 150, 768, M1),
                  ('c1 fwd 32->768', 0, M1, 768, 32), ('c1 dW', 2, 768, 32, M1),
                  ('fc1 s1 32->128', 0, M1, 128, 32), ('fc2 s1 128->32', 0, M1, 32, 128), ('q s1 32->32', 0, M1, 32, 32),
                  ('fc1 s2 64->256', 0, M1 // 4, 256, 64), ('fc1 s3 160->640', 0, M1 // 16, 640, 160),
                  ('fc1 s4 256->1024', 0, M1 // 64, 1024, 256), ('fc1 s1 dW', 2, 128, 32, M1), ('fc2 s1 dW', 2, 32, 128, M1)]
        for name, L, M, N, K in shapes:
            if L == 0:
                A, Bm = rnd(M, K), rnd(N, K)
            elif L == 1:
                A, Bm = rnd(M, K), rnd(K, N)
            else:
                A, Bm = rnd(K, M), rnd(K, N)
            odt = torch.float32 if L == 2 else bf
            sk = hip.pick_splitk(M, N, K) if L == 2 else 1
            out = torch.empty(M, N, device=dev, dtype=odt)
            ms = timeit(lambda: hip.gemm(L, A, Bm, M, N, K, out=out, split_k=sk))
            report(f'gemm {name} [{M}x{N}x{K}] sk{sk}', ms, nbytes=2 * (M * K + N * K) + out.element_size() * M * N, flops=2.0 * M * N * K)
    if want('bn'):
        rows, C = B * 128 * 128, 768
        x, dy = rnd(rows, C), rnd(rows, C)
        g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        cs = torch.ones(B, C, device=dev)
        ms = timeit(lambda: hip.bn_stats(x, rm, rv, 0.1, 1e-5))
        report('bn_stats [rows,768]', ms, nbytes=2 * rows * C)
        mean, rstd = hip.bn_stats(x, rm, rv, 0.1, 1e-5)
        ms = timeit(lambda: hip.bn_apply(x, mean, rstd, g, b_, 1, cs, 128 * 128))
        report('bn_apply+relu+drop', ms, nbytes=4 * rows * C)
        ms = timeit(lambda: hip.bn_bwd(x, dy, mean, rstd, g, b_, 1, cs, 128 * 128, False))
        report('bn_bwd (sums + apply)', ms, nbytes=10 * rows * C)
    if want('ln'):
        for rows, C in ((B * 16384, 32), (B * 4096, 64), (B * 1024, 160), (B * 256, 256)):
            x, dy = rnd(rows, C), rnd(rows, C)
            g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            ms = timeit(lambda: hip.layernorm_fwd(x, g, b_, 1e-5))
            report(f'ln_fwd [{rows},{C}]', ms, nbytes=4 * rows * C)
            _, mean, rstd = hip.layernorm_fwd(x, g, b_, 1e-5)
            ms = timeit(lambda: hip.layernorm_bwd(x, dy, g, mean, rstd))
            report(f'ln_bwd [{rows},{C}]', ms, nbytes=6 * rows * C)
    if want('dw'):
        for H, C in ((128, 128), (64, 256), (32, 640), (16, 1024)):
            x, dy = rnd(B * H * H, C), rnd(B * H * H, C)
            w9, bias = torch.randn(C, 9, device=dev), torch.randn(C, device=dev)
            ms = timeit(lambda: hip.dwconv3x3_gelu_fwd(x, w9, bias, B, H, H, C, True))
            report(f'dwconv3x3+gelu fwd [{B},{H},{H},{C}]', ms, nbytes=4 * B * H * H * C)
            ms = timeit(lambda: hip.dwconv3x3_gelu_bwd(x, w9, bias, dy, B, H, H, C, True))
            report(f'dwconv3x3+gelu bwd [{B},{H},{H},{C}]', ms, nbytes=14 * B * H * H * C)
            ms = timeit(lambda: hip.dwconv3x3_gelu_fwd(x, w9, bias, B, H, H, C, False))
            report(f'dwconv3x3 (no gelu) fwd [{B},{H},{H},{C}]', ms, nbytes=4 * B * H * H * C)
            ms = timeit(lambda: hip.dwconv3x3_gelu_bwd(x, w9, bias, dy, B, H, H, C, False))
            report(f'dwconv3x3 (no gelu) bwd [{B},{H},{H},{C}]', ms, nbytes=14 * B * H * H * C)
    if want('attn'):
        for N, heads in ((16384, 1), (4096, 2), (1024, 5), (256, 8)):
            C = heads * 32
            q, kv, do = rnd(B * N, C), rnd(B * 256, 2 * C), rnd(B * N, C)
            ms = timeit(lambda: hip.attention_fwd(q, kv[:, :C], kv[:, C:], B, heads, N, 256, 32, 32 ** -0.5))
            fl = 4.0 * B * heads * N * 256 * 32
            report(f'attn fwd N={N} heads={heads}', ms, flops=fl)
            o, lse = hip.attention_fwd(q, kv[:, :C], kv[:, C:], B, heads, N, 256, 32, 32 ** -0.5)
            dkv = torch.empty_like(kv)
            ms = timeit(lambda: hip.attention_bwd(q, kv[:, :C], kv[:, C:], o, do, lse, B, heads, N, 256, 32, 32 ** -0.5, dkv[:, :C], dkv[:, C:]))
            report(f'attn bwd N={N} heads={heads}', ms, flops=2.5 * fl)
    if want('loss'):
        nc, h, H = 150, 128, 512
        ld = 152
        lo = rnd(B * h * h, ld)[:, :nc]
        tgt = torch.randint(0, nc, (B, H, H), device=dev)
        tgt[:, :8] = 255
        ms = timeit(lambda: hip.ce_dice_fwd(lo, B, nc, h, h, H, H, tgt, 255, None, True))
        report('ce_dice_fwd (fused upsample)', ms, nbytes=B * (h * h * ld * 2 + H * H * 8))
        loss, stats = hip.ce_dice_fwd(lo, B, nc, h, h, H, H, tgt, 255, None, True)
        go = torch.ones(1, device=dev)

        def bwd():
            return hip.ce_dice_bwd(lo, B, nc, h, h, H, H, tgt, 255, None, True, stats, go)
        ms = timeit(bwd, iters=10)
        report('ce_dice_bwd (fused transposed upsample)', ms, nbytes=B * (2 * h * h * ld * 2 + H * H * 8))
        mat = torch.zeros(nc, nc, dtype=torch.int64, device=dev)
        hist = torch.zeros(nc, nc, dtype=torch.int64, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        ms = timeit(lambda: hip.argmax_confmat(lo, B, nc, h, h, H, H, tgt, 255, mat, hist, flag))
        report('argmax_confmat (fused upsample)', ms, nbytes=B * (h * h * ld * 2 + H * H * 8))
    if want('bncls'):
        rows, C, K = B * 128 * 128, 768, 160
        x = rnd(rows, C)
        dyc = rnd(rows, K) * 1e-3
        wc = rnd(K, C) * 0.03
        g, b_ = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        mean, rstd = hip.bn_stats(x, rm, rv, 0.1, 1e-5)
        cs = torch.ones(B, C, device=dev)
        ms = timeit(lambda: hip.bn_cls_bwd(dyc, wc, x, mean, rstd, g, b_, 1, cs, 128 * 128, False))
        report('bn_cls_bwd (both passes, da recomputed)', ms, nbytes=2 * rows * (3 * C + 2 * K))
    if want('upadd'):
        E, Hh = 768, 128
        base = rnd(B * Hh * Hh, E)
        srcs = [(rnd(B * (Hh // r) * (Hh // r), E), Hh // r, Hh // r) for r in (2, 4, 8)]
        nb = int(2 * B * Hh * Hh * E * (2 + 1 / 4 + 1 / 16 + 1 / 64))
        ms = timeit(lambda: hip.upsample_add_stats(base, srcs, B, Hh, Hh, E))
        report('upsample_add_248 + BN statistics', ms, nbytes=nb)
        ms = timeit(lambda: hip.upsample_add(base, srcs, B, Hh, Hh, E))
        report('upsample_add_248', ms, nbytes=nb)
    if want('bwd248'):
        E, Hh = 768, 128
        dy = rnd(B * Hh * Hh, E)
        ms = timeit(lambda: hip.bilinear_bwd_248(dy, B, Hh, Hh, E))
        report('bilinear_bwd_248 (x2, x4, x8 in one pass)', ms, nbytes=int(2 * B * Hh * Hh * E * (1 + 1 / 4 + 1 / 16 + 1 / 64)))
        tot = 0.0
        for r in (2, 4, 8):
            ms1 = timeit(lambda: hip.bilinear_bwd(dy, B, Hh // r, Hh // r, E, Hh, Hh))
            report(f'  separate bilinear_bwd x{r}', ms1, nbytes=int(2 * B * Hh * Hh * E * (1 + 1 / r / r)))
            tot += ms1
        print(f'  three separate launches: {tot * 1e3:.1f} us')
    if want('resize'):
        E = 768
        for h in (64, 32, 16):
            t = rnd(B * h * h, E)
            cat = torch.empty(B * 128 * 128, 4 * E, device=dev, dtype=bf)
            ms = timeit(lambda: hip.bilinear_fwd(t, B, h, h, E, 128, 128, cat[:, :E]))
            report(f'bilinear_fwd {h}->128 C=768', ms, nbytes=2 * E * B * (h * h + 128 * 128))
            d = cat[:, :E]
            ms = timeit(lambda: hip.bilinear_bwd(d, B, h, h, E, 128, 128))
            report(f'bilinear_bwd 128->{h} C=768', ms, nbytes=2 * E * B * (h * h + 128 * 128))
    if want('misc'):
        x = rnd(B * 16384, 3072)
        ms = timeit(lambda: hip.colsum(x[:, :768]))
        report('colsum [rows,768] ld 3072', ms, nbytes=2 * B * 16384 * 768)
        img = torch.randn(B, 3, 512, 512, device=dev)
        ms = timeit(lambda: hip.im2col(img, bf, True, B, 512, 512, 3, 7, 7, 4, 3, 128, 128, 152))
        report('im2col image k7s4', ms, nbytes=B * (3 * 512 * 512 * 4 + 16384 * 152 * 2))


if __name__ == '__main__':
    main()
