#!/usr/bin/env python3
"""Record which kernel every C-ABI call of one train step takes, for every BASELINE configuration (cfg2 at per-GPU batch 4 / 16 / 128 / 256,
cfg3, cfg4, cfg5, and the fp8 option of cfg3 / cfg5), on the MI355X:

    python tools/make_dispatch_table.py [--out tests/golden/dispatch_table.json] [--only cfg2_b4 ...]

One entry per distinct (entry point, arguments) of the step as the product runs it -- the function GraphedTrainStep captures, with the
grouped weight gradients and derived-weight launches of the captured step -- with the kernels it launched and how often the step
makes the call.  tests/test_host_cpu.py::test_dispatch_of_baseline_shapes replays the table as dry runs WITHOUT a GPU; a dispatch edit
that moves a BASELINE shape shows up as a diff of this file (shapes: models/build_models.py:43-54's width rule of the reference)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

CASES = {'cfg2_b4': ('cfg2', 4, False), 'cfg2_b16': ('cfg2', 16, False), 'cfg2_b128': ('cfg2', 128, False), 'cfg2_b256': ('cfg2', 256, False), 'cfg3_b32': ('cfg3', 32, False),
         'cfg4_b16': ('cfg4', 16, False), 'cfg5_b8': ('cfg5', 8, False), 'cfg3_b64': ('cfg3', 64, False), 'cfg4_b32': ('cfg4', 32, False),
         'cfg5_b32': ('cfg5', 32, False), 'cfg3_b32_fp8': ('cfg3', 32, True), 'cfg5_b8_fp8': ('cfg5', 8, True)}
# launches that are not dispatch decisions (one kernel whatever the shape): left out to keep the table readable
ELEMENTWISE = {'segf_cast', 'segf_cast2d', 'segf_permute021', 'segf_add', 'segf_zero', 'segf_scale_rows', 'segf_add_i64', 'segf_gelu',
               'segf_bn_affine_table', 'segf_bn_stats_from_sums', 'segf_hist_accum', 'segf_rowdot'}


def record_case(cfg, batch, fp8):
    import bench_legs
    from segmentation_factory_amd import criterion_lowres, dispatch
    from segmentation_factory_amd.graph import GraphedTrainStep
    core, opt, nc, H, W = bench_legs.build(cfg, fp8)
    x, y = bench_legs.synthetic_batch(batch, nc, H, W, 0)
    x, y = x.cuda(), y.cuda()

    def loss_fn(model, img, lbl):
        return criterion_lowres(model.forward_lowres(img), lbl, (H, W), None, num_classes=nc, dice=True, ignore_index=255)
    gs = GraphedTrainStep(core, opt, loss_fn, (x, y), clip_grad=0.02, clip_mode='agc', warmup=1)
    with dispatch.record() as calls:
        gs._forward_backward_eager()             # the captured function, once, launched eagerly with the trace on
    torch.cuda.synchronize()
    entries = [e for e in dispatch.unique(calls) if e['fn'] not in ELEMENTWISE]
    del gs, core, opt
    torch.cuda.empty_cache()
    return entries


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden', 'dispatch_table.json'))
    ap.add_argument('--only', nargs='*', default=None)
    a = ap.parse_args()
    assert torch.cuda.is_available(), 'recording needs the MI355X (the replay does not)'
    table = {}
    if a.only and os.path.exists(a.out):
        with open(a.out) as fh:
            table = json.load(fh)
    for key, (cfg, batch, fp8) in CASES.items():
        if a.only and key not in a.only:
            continue
        table[key] = record_case(cfg, batch, fp8)
        nk = sum(len(e['kernels']) * e['count'] for e in table[key])
        print(f'{key}: {len(table[key])} distinct launching calls, {nk} kernel launches per step', flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as fh:
        json.dump(table, fh, indent=0, separators=(',', ':'))
    print('written', a.out, os.path.getsize(a.out), 'bytes')


if __name__ == '__main__':
    main()
