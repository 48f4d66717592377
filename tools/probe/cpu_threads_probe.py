#!/usr/bin/env python3
"""bench.py's cpu_baseline caps torch at 16 threads on the GPU box (256 logical CPUs, a 1-GPU job's share is 16 cores): this measures
what 16 / 32 / 64 threads give on the same step (oracle forward + closed-form CE/Dice + backward, batch 2, 512 x 512, 150 classes) so
that the cap is a measurement, not a comment.  python tools/probe/cpu_threads_probe.py > profiles/r05_cpu_threads.txt"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import loss as OL, nets as ON, weights as OW

sd = OW.make_state_dict('MiT-B0', 'SegFormerHead', 150, 0, lively=False)
sd = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))) for k, v in sd.items()}
x, y = OW.synthetic_batch(2, 512, 512, 150, 0)


def step():
    for v in sd.values():
        v.grad = None
    o, _ = ON.model_forward(sd, x, 'MiT-B0', 'SegFormerHead', training=True)
    OL.criterion_closed_form(o, y, None, num_classes=150, dice=True, ignore_index=255).backward()


print(f'os.cpu_count() {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}')
for n in (16, 32, 64, 16):
    torch.set_num_threads(n)
    step()
    t0 = time.time()
    for _ in range(3):
        step()
    dt = (time.time() - t0) / 3
    print(f'{n:3d} threads: {dt:.2f} s / step = {2 / dt:.3f} images/s (vectorised Dice; the reference-structured loop adds a single-threaded Python loop on top)', flush=True)
