"""The loss forward + backward of the cfg2 batch (128 x 150 classes x 128^2 taps -> 512^2), a few launches each, for rocprofv3
(--kernel-trace --stats, or --pmc passes).  argv: [n_launches] [lse|nolse]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
use_lse = (sys.argv[2] if len(sys.argv) > 2 else 'lse') == 'lse'
dev = 'cuda'
B, nc, h, ld = 128, 150, 128, 160
H = 4 * h
g = torch.Generator(device=dev).manual_seed(0)
buf = (torch.randn(B * h * h, ld, device=dev, generator=g) * 2).to(torch.bfloat16)
lo = buf[:, :nc]
tgt = torch.randint(0, nc, (B, H, H), device=dev, generator=g)
if 'blocks' in sys.argv:      # piecewise-constant labels (what a segmentation map looks like): colliding label-term adds
    coarse = torch.randint(0, nc, (B, H // 32, H // 32), device=dev, generator=g)
    tgt = coarse.repeat_interleave(32, 1).repeat_interleave(32, 2).contiguous()
tgt[:, :8] = 255
go = torch.ones(1, device=dev)
for _ in range(n):
    if use_lse:
        loss, stats, lse = hip.ce_dice_fwd(lo, B, nc, h, h, H, H, tgt, 255, None, True, want_lse=True)
    else:
        (loss, stats), lse = hip.ce_dice_fwd(lo, B, nc, h, h, H, H, tgt, 255, None, True), None
    d = hip.ce_dice_bwd(lo, B, nc, h, h, H, H, tgt, 255, None, True, stats, go, lse=lse)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record()
for _ in range(5):
    r = hip.ce_dice_fwd(lo, B, nc, h, h, H, H, tgt, 255, None, True, want_lse=use_lse)
e1.record()
for _ in range(5):
    d = hip.ce_dice_bwd(lo, B, nc, h, h, H, H, tgt, 255, None, True, stats, go, lse=lse)
e2.record()
torch.cuda.synchronize()
print(f'fwd {e0.elapsed_time(e1) / 5:.3f} ms  bwd {e1.elapsed_time(e2) / 5:.3f} ms  loss {loss[0].item():.5f}')
