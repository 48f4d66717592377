#!/usr/bin/env python3
"""Dump / compare the head-dim-64 attention results of two builds or policy settings bit for bit:
  python tools/probe/attn64_dump.py dump /tmp/a.pt     (SEGFAC_HIP_LIB / SEGFAC_* select the build and the policy)
  python tools/probe/attn64_dump.py cmp /tmp/a.pt /tmp/b.pt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def dump(path):
    from segmentation_factory_amd import hip
    out = {}
    for (B, heads, N, Nkv, hd) in ((1, 2, 8192, 2048, 64), (1, 1, 8200, 2048, 64), (1, 2, 700, 300, 64), (2, 1, 1000, 130, 64)):
        g = torch.Generator().manual_seed(5)
        C = heads * hd
        q = torch.randn(B * N, C, generator=g).bfloat16().cuda()
        k = torch.randn(B * Nkv, C, generator=g).bfloat16().cuda()
        v = torch.randn(B * Nkv, C, generator=g).bfloat16().cuda()
        do = torch.randn(B * N, C, generator=g).bfloat16().cuda()
        for rep in range(4):
            o, lse = hip.attention_fwd(q, k, v, B, heads, N, Nkv, hd, hd ** -0.5)
            dk = torch.empty_like(k); dv = torch.empty_like(v)
            dq = hip.attention_bwd(q, k, v, o, do, lse, B, heads, N, Nkv, hd, hd ** -0.5, dk, dv)
            torch.cuda.synchronize()
            out[(B, heads, N, Nkv, hd, rep)] = [t.cpu() if torch.is_tensor(t) else t for t in (o, lse, dq, dk, dv)]
    torch.save(out, path)


def cmp(a, b):
    A, Bm = torch.load(a), torch.load(b)
    for key in A:
        names = ('o', 'lse', 'dq', 'dk', 'dv')
        res = []
        for nm, x, y in zip(names, A[key], Bm[key]):
            if not torch.is_tensor(x):
                continue
            res.append(f'{nm} {"==" if torch.equal(x, y) else "!= (%d of %d, max %.3e)" % ((x != y).sum().item(), x.numel(), (x.float() - y.float()).abs().max().item())}')
        print(key, ' '.join(res))


def selfcmp(a):
    A = torch.load(a)
    for key in A:
        if key[-1] == 0:
            continue
        base = A[key[:-1] + (0,)]
        res = []
        for nm, x, y in zip(('o', 'lse', 'dq', 'dk', 'dv'), base, A[key]):
            if torch.is_tensor(x):
                res.append(f'{nm} {"==" if torch.equal(x, y) else "!= (%d)" % (x != y).sum().item()}')
        print('rep', key, ' '.join(res))


if __name__ == '__main__':
    if sys.argv[1] == 'self':
        selfcmp(sys.argv[2])
    elif sys.argv[1] == 'dump':
        dump(sys.argv[2])
    else:
        cmp(sys.argv[2], sys.argv[3])
