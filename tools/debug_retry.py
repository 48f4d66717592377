"""Probe (test infrastructure): retry flag of the batched loss forward under eager / side-stream / graph replay."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from segmentation_factory_amd import SegmentationModel, functional as Fh
from segmentation_factory_amd.graph import GraphedTrainStep
from segmentation_factory_amd.optim import FusedAGCAdamW, param_groups_weight_decay
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NC, H, W = 150, 512, 512
torch.manual_seed(1234)
m = SegmentationModel('MiT-B0', num_classes=NC, seg_head='SegFormerHead').cuda().train()
opt = FusedAGCAdamW(param_groups_weight_decay(m, 0.025), lr=2e-4)
x, y = bench.synthetic_batch(B, 0)
x, y = x.cuda(), y.cuda()
keep = {}
def loss_fn(model, img, lbl):
    lo = model.forward_lowres(img)
    loss, parts, stats = Fh.upsample_ce_dice(lo.data, lbl, (B, NC, lo.H, lo.W, H, W), 255, None, True)
    keep['stats'] = stats
    return loss
gs = GraphedTrainStep(m, opt, loss_fn, (x, y), clip_grad=0.02, clip_mode='agc')
for i in range(3):
    l = gs.step(x, y)
    torch.cuda.synchronize()
    print('step', i, 'loss', l.item(), 'retry flags', keep['stats'][-4:].view(torch.int32).tolist())
import time
for _ in range(3): gs.forward_backward(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): gs.forward_backward(x, y)
torch.cuda.synchronize(); print('ms/replay', (time.perf_counter() - t0) * 100)
