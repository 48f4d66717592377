#!/usr/bin/env python3
"""Diagnostic (GPU box): error statistics of the HIP path against the reference goldens, used to set the tolerances written
in tests/test_model_gpu.py::test_e2e_against_reference_golden.  python tools/grad_parity_report.py [tag ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights as OW                      # noqa: E402
from oracle.make_goldens import sample_indices        # noqa: E402
from segmentation_factory_amd import SegmentationModel, criterion_lowres   # noqa: E402

tags = sys.argv[1:] or ['segformer_b0_64', 'mbv2_fpn_64', 'convnext_uper_64', 'convnext_uper_128', 'convnextv2_tiny_uper_128', 'mbv2_fpn_128']
for tag in tags:
    g = np.load(os.path.join(ROOT, 'tests', 'golden', f'e2e_{tag}.npz'))
    backbone, head, nc = str(g['backbone']), str(g['head']), int(g['nc'])
    B, H, W, seed = int(g['B']), int(g['H']), int(g['W']), int(g['seed'])
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    compact = 'lowres_eval' in g.files
    for dtype in (torch.float32, torch.bfloat16):
        m = SegmentationModel(backbone, num_classes=nc, seg_head=head, compute_dtype=dtype)
        m.load_state_dict(sd)
        m = m.cuda()
        for mod in m.backbone.modules():
            if hasattr(mod, 'drop_prob'):
                mod.drop_prob = 0.0
        m.decode_head.dropout.p = 0.0
        m.eval()
        with torch.no_grad():
            if compact:
                ev = m.forward_lowres(x.cuda()).nchw().float().cpu().numpy()
                ref = g['lowres_eval']
            else:
                ev = m(x.cuda()).cpu().numpy()
                ref = g['logits_eval']
        e_eval = np.abs(ev - ref).max() / np.abs(ref).max()
        m.train()
        lo = m.forward_lowres(x.cuda())
        loss = criterion_lowres(lo, y.cuda(), (H, W), None, num_classes=nc, dice=True, ignore_index=255)
        loss.backward()
        if compact:
            tr, ref = lo.nchw().float().detach().cpu().numpy(), g['lowres_train']
        else:
            with torch.no_grad():
                tr = m(x.cuda()).cpu().numpy()
            ref = g['logits_train']
        e_tr = np.abs(tr - ref).max() / np.abs(ref).max()
        gmax = float(g['grad_global_max'])
        params = dict(m.named_parameters())
        rows = []
        for i, name in enumerate(g['grad_names']):
            name = str(name)
            gr = params[name].grad
            gr = torch.zeros_like(params[name]) if gr is None else gr
            gr = gr.detach().float().cpu()
            ref_norm = float(g['grad_norms'][i])
            got = gr.flatten()[sample_indices(name, gr.numel())].numpy()
            scale = np.abs(g['grad_samples'][i]).max() + ref_norm / max(1.0, np.sqrt(gr.numel())) + 1e-2 * gmax
            rows.append((float(np.abs(got - g['grad_samples'][i]).max() / scale),
                         abs(gr.double().norm().item() - ref_norm) / (ref_norm + 1e-1 * gmax), name))
        se = np.array([r[0] for r in rows]); ne = np.array([r[1] for r in rows])
        print(f'[{tag} {str(dtype)[6:]}] eval logits {e_eval:.2e} train logits {e_tr:.2e} loss {abs(loss.item() - float(g["loss"])) / float(g["loss"]):.2e} | '
              f'grad sample err/scale: median {np.median(se):.3f} p90 {np.quantile(se, .9):.3f} max {se.max():.3f} | norm rel: median {np.median(ne):.3f} max {ne.max():.3f}')
        for r in sorted(rows, reverse=True)[:4]:
            print('      worst', f'{r[0]:.3f} {r[1]:.3f}', r[2])
        del m
