#!/usr/bin/env python3
"""GPU probe: where the time of engine.evaluate goes (eager vs graphed eval forward), SegFormer-B0 512x512.  python tools/probe/eval_probe.py [batch]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_legs as BL
from segmentation_factory_amd.graph import GraphedEvalForward
from segmentation_factory_amd.metrics import Metrics
from segmentation_factory_amd.utils import ConfusionMatrix

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
core, opt, nc, H, W = BL.build('cfg2')
core.eval()
x, y = BL.synthetic_batch(B, nc, H, W, 0)
x, y = x.cuda(), y.cuda()
n = 20


def wall(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.inference_mode():
    g = GraphedEvalForward(core, x)
    metric, confmat = Metrics(nc, 255, 'cuda'), ConfusionMatrix(nc)
    t_eager = wall(lambda: core.forward_lowres(x))
    t_graph = wall(lambda: g(x))
    t_graph_nocopy = wall(lambda: g(g.x))
    lo = g(x)
    t_metric = wall(lambda: metric.update_lowres(lo, y, (H, W), confmat=confmat))
    print(f'batch {B}: eager forward {t_eager:.3f} ms, graph replay (with input copy) {t_graph:.3f} ms, replay only {t_graph_nocopy:.3f} ms, '
          f'metric update {t_metric:.3f} ms')
    host0 = time.perf_counter()
    for _ in range(n):
        core.forward_lowres(x)
    host = (time.perf_counter() - host0) / n * 1e3
    torch.cuda.synchronize()
    print(f'eager host enqueue time per forward {host:.3f} ms')
