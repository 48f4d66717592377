#!/usr/bin/env python3
"""GPU probe: engine.evaluate, graph vs eager, with a breakdown of the graph path.  python tools/probe/eval_loop_probe.py [batch] [nb]"""
import contextlib, io, os, sys, time
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_legs as BL
from segmentation_factory_amd.engine import evaluate
from segmentation_factory_amd.graph import GraphedEvalSession

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 10
core, opt, nc, H, W = BL.build('cfg2')
core.eval()
x, y = BL.synthetic_batch(B, nc, H, W, 0)
x, y = x.cuda(), y.cuda()
data = [(x, y)] * nb
for mode in ('graph', 'eager', 'graph'):
    args = SimpleNamespace(nb_classes=nc, ignore_label=255, hip_graph=(mode == 'graph'))
    with contextlib.redirect_stdout(io.StringIO()):
        evaluate(args, core, data[:3], torch.device('cuda'), 100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        confmat, metric = evaluate(args, core, data, torch.device('cuda'), 100)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f'{mode}: evaluate returned after {1e3 * (t1 - t0) / nb:.3f} ms / batch (host), GPU drained after {1e3 * (t2 - t0) / nb:.3f} ms / batch')
with torch.inference_mode():
    sess = GraphedEvalSession(core)
    for _ in range(3):
        sess(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nb):
        sess(x)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'session(x) x {nb}: host {1e3 * (t1 - t0) / nb:.3f} ms, drained {1e3 * (t2 - t0) / nb:.3f} ms / batch')
    g = sess.cache[(tuple(x.shape), x.dtype)]
    t0 = time.perf_counter()
    for _ in range(nb):
        g.graph.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'replay only x {nb}: host {1e3 * (t1 - t0) / nb:.3f} ms, drained {1e3 * (t2 - t0) / nb:.3f} ms / batch')
    t0 = time.perf_counter()
    for _ in range(nb):
        g.x.copy_(x, non_blocking=True)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'input copy x {nb}: host {1e3 * (t1 - t0) / nb:.3f} ms, drained {1e3 * (t2 - t0) / nb:.3f} ms / batch')
