"""Conditioning probe (test infrastructure): bf16 vs fp32 HIP gradients of ConvNeXt-T + UPerHead at a size where the
BatchNorm statistics are not degenerate (B=4, 160x160), printing the worst per-parameter relative norm errors."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import weights as OW
from segmentation_factory_amd import SegmentationModel, criterion_lowres
backbone, head, nc, B, H, W, seed = 'ConvNeXt', 'UPerHead', 19, 4, 160, 160, 2
sd = OW.make_state_dict(backbone, head, nc, seed)
x, y = OW.synthetic_batch(B, H, W, nc, seed)
grads = {}
for dtype in (torch.float32, torch.bfloat16):
    m = SegmentationModel(backbone, num_classes=nc, seg_head=head, compute_dtype=dtype); m.load_state_dict(sd); m = m.cuda().train()
    for mod in m.backbone.modules():
        if hasattr(mod, 'drop_prob'): mod.drop_prob = 0.0
    m.decode_head.dropout.p = 0.0
    lo = m.forward_lowres(x.cuda())
    loss = criterion_lowres(lo, y.cuda(), (H, W), None, num_classes=nc, dice=True, ignore_index=255)
    loss.backward()
    print(dtype, 'loss', loss.item())
    grads[dtype] = {k: p.grad.float().cpu() for k, p in m.named_parameters()}
rows = []
for k, g32 in grads[torch.float32].items():
    gb = grads[torch.bfloat16][k]
    rows.append(((gb - g32).norm().item() / max(g32.norm().item(), 1e-12), k, g32.norm().item()))
rows.sort(reverse=True)
for r in rows[:10]: print('  rel err %.3e  %s (norm %.3e)' % r)
import statistics
print('median rel err', statistics.median(r[0] for r in rows))
