#!/usr/bin/env python3
"""Eager eval forwards of SegFormer-B0 512x512 (cfg2) or another config in one storage type, for a kernel-trace profile:
python tools/probe/eval_dtype_prof.py fp32|bf16 [batch] [cfg]   (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_legs as BL

dt = {'fp32': torch.float32, 'bf16': torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else 'fp32']
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = sys.argv[3] if len(sys.argv) > 3 else 'cfg2'
core, opt, nc, H, W = BL.build(cfg)
core.eval().set_compute_dtype(dt)
x, y = BL.synthetic_batch(B, nc, H, W, 0)
x = x.cuda()
with torch.inference_mode():
    core.forward_lowres(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        core.forward_lowres(x)
    torch.cuda.synchronize()
    print(f'{cfg} {sys.argv[1] if len(sys.argv) > 1 else "fp32"} batch {B}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per forward')
