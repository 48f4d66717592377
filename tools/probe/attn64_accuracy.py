#!/usr/bin/env python3
"""Error of the head-dim-64 attention kernels against an fp32 CPU reference on the same bf16 inputs, for the dispatch-policy settings in
PROBE_AB (e.g. 'attn64_prescale=0;attn64_prescale=1'): python tools/probe/attn64_accuracy.py
Prints, per setting, the RMS and the maximum error of O, dQ, dK/dV relative to the RMS of the reference tensor."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from segmentation_factory_amd import functional as Fh, hip  # noqa: E402


def main():
    B, heads, N, Nkv, hd = 1, 2, 8192, 2048, 64
    C = heads * hd
    g = torch.Generator().manual_seed(5)
    spread = float(os.environ.get('ATTN_SPREAD', '1.0'))          # > 1: peakier softmax rows (larger scores)
    q = (spread * torch.randn(B * N, C, generator=g)).bfloat16()
    kv = torch.randn(B * Nkv, 2 * C, generator=g).bfloat16()
    do = torch.randn(B * N, C, generator=g).bfloat16()
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    qh = qr.reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    k, v = kvr.reshape(B, Nkv, 2, heads, hd).permute(2, 0, 3, 1, 4)
    a = ((qh @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    ref = (a @ v).transpose(1, 2).reshape(B * N, C)
    ref.backward(do.float())
    settings = (os.environ.get('PROBE_AB') or 'attn64_prescale=0;attn64_prescale=1').split(';')
    for s in settings:
        for kvs in s.split(','):
            name, val = kvs.split('=')
            hip.policy_set(name.strip(), int(val))
        qd, kvd = q.cuda().requires_grad_(True), kv.cuda().requires_grad_(True)
        o = Fh.attention(qd, kvd, B, N, Nkv, heads)
        o.backward(do.cuda())
        torch.cuda.synchronize()
        row = []
        for nm, got, want in (('O', o, ref), ('dQ', qd.grad, qr.grad), ('dKV', kvd.grad, kvr.grad)):
            e = got.detach().float().cpu() - want.detach()
            sc = want.detach().pow(2).mean().sqrt().item()
            row.append(f'{nm}: rms {e.pow(2).mean().sqrt().item() / sc:.3e} max {e.abs().max().item() / sc:.3e}')
        print(f'spread {spread} {s:28s} ' + ' | '.join(row), flush=True)


if __name__ == '__main__':
    main()
