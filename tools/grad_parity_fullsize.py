#!/usr/bin/env python3
"""Diagnostic (GPU box): per-tensor gradient error of the HIP path at BASELINE's full sizes against the CPU oracle -- the data
behind the per-config gradient bars of tests/test_model_gpu.py::test_full_size_fp32_and_bf16_vs_oracle.  For every config and
dtype it lists the WORST tensors (error as a fraction of max|reference gradient| + 5 % of the global gradient scale), so that a
bar can be read as "tensor X sets it", and how many tensors are above a tenth of the worst.
    python tools/grad_parity_fullsize.py cfg3 cfg5 > profiles/r03_grad_parity_fullsize.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loss as OL, nets as ON, weights as OW                     # noqa: E402
from segmentation_factory_amd import SegmentationModel, criterion_lowres     # noqa: E402

FULL_SIZE = {'cfg1': ('MobileNetV2', 'FPNHead', 21, 2, 256, 256), 'cfg2': ('MiT-B0', 'SegFormerHead', 150, 2, 512, 512),
             'cfg3': ('ConvNeXt', 'UPerHead', 150, 4, 512, 512), 'cfg4': ('MiT-B2', 'SegFormerHead', 19, 1, 1024, 2048),
             'cfg5': ('convnextv2_large', 'UPerHead', 171, 4, 640, 640)}

torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
for cfg in (sys.argv[1:] or ['cfg3', 'cfg5']):
    backbone, head, nc, B, H, W = FULL_SIZE[cfg]
    sd = OW.make_state_dict(backbone, head, nc, 0)
    x, y = OW.synthetic_batch(B, H, W, nc, 0)
    t0 = time.time()
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))) for k, v in sd.items()}
    o, _ = ON.model_forward(sdg, x, backbone, head, training=True, lowres=True)
    up = torch.nn.functional.interpolate(o, size=(H, W), mode='bilinear', align_corners=False)
    OL.criterion_closed_form(up, y, None, num_classes=nc, dice=True, ignore_index=255).backward()
    ref = {k: v.grad for k, v in sdg.items() if v.grad is not None}
    gmax = max(r.abs().max().item() for r in ref.values())
    print(f'== {cfg}: {backbone} + {head}, {nc} classes, batch {B}, {H}x{W}; oracle {time.time() - t0:.0f} s; global max|grad| {gmax:.3e}')
    # COMPARATOR: the same oracle under torch.autocast(bfloat16) on the CPU -- what autocast arithmetic (bf16 matmul / conv operands, fp32
    # accumulation, fp32 normalisation and softmax; the reference trains under torch.cuda.amp.autocast, /root/reference/engine.py:40)
    # costs against the fp32 oracle, tensor by tensor.  The HIP bf16 column is judged against THIS, not against zero.
    t1 = time.time()
    sdc = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))) for k, v in sd.items()}
    with torch.autocast('cpu', dtype=torch.bfloat16):
        oc, _ = ON.model_forward(sdc, x, backbone, head, training=True, lowres=True)
        upc = torch.nn.functional.interpolate(oc.float(), size=(H, W), mode='bilinear', align_corners=False)
        lc = OL.criterion_closed_form(upc, y, None, num_classes=nc, dice=True, ignore_index=255)
    lc.backward()
    cmp_err = {k: (sdc[k].grad.float() - r).abs().max().item() / (r.abs().max().item() + 0.05 * gmax) for k, r in ref.items() if sdc[k].grad is not None}
    cs = sorted(cmp_err.values())
    print(f'-- comparator (CPU autocast bf16 vs fp32 oracle, {time.time() - t1:.0f} s): worst {cs[-1]:.3e}, median {cs[len(cs) // 2]:.3e}; '
          f'logits {((oc.float() - o).abs().max() / o.abs().max()).item():.2e}')
    for dtype in (torch.float32, torch.bfloat16):
        m = SegmentationModel(backbone, num_classes=nc, seg_head=head, compute_dtype=dtype)
        m.load_state_dict(sd)
        m = m.cuda().train()
        for mod in m.backbone.modules():
            if hasattr(mod, 'drop_prob'):
                mod.drop_prob = 0.0
        m.decode_head.dropout.p = 0.0
        lo = m.forward_lowres(x.cuda())
        criterion_lowres(lo, y.cuda(), (H, W), None, num_classes=nc, dice=True, ignore_index=255).backward()
        rows = []
        for k, p in m.named_parameters():
            if k in ref and p.grad is not None:
                r = ref[k]
                d = (p.grad.float().cpu() - r).abs().max().item()
                rows.append((d / (r.abs().max().item() + 0.05 * gmax), k, d, r.abs().max().item(), tuple(r.shape)))
        rows.sort(reverse=True)
        worst = rows[0][0]
        print(f'-- {str(dtype)[6:]}: {len(rows)} tensors, worst {worst:.3e}; {sum(r[0] > 0.1 * worst for r in rows)} tensors above a tenth of it; '
              f'median {rows[len(rows) // 2][0]:.3e}')
        for e, k, d, rm, shp in rows[:8]:
            print(f'   {e:9.3e}  {k:<58} max|diff| {d:.3e}  max|ref| {rm:.3e}  {shp}   comparator {cmp_err.get(k, float("nan")):.3e}')
        if dtype == torch.bfloat16:
            ratios = sorted((e / max(cmp_err[k], 1e-3), k) for e, k, *_ in rows if k in cmp_err)
            print(f'   HIP bf16 error / comparator error (comparator floored at 1e-3): median {ratios[len(ratios) // 2][0]:.2f}, 90th percentile '
                  f'{ratios[int(0.9 * len(ratios))][0]:.2f}, worst {ratios[-1][0]:.2f} ({ratios[-1][1]}); tensors above 1.5: {sum(r[0] > 1.5 for r in ratios)} of {len(ratios)}')
        del m, lo
        torch.cuda.empty_cache()
