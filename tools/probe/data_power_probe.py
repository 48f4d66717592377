"""Does the eight-phase GEMM's rate depend on WHAT it multiplies?  The UPerHead bottleneck conv 3072 -> 768 @ 128^2, batch 32 (22.3 TFLOP),
forward, on operands of different bit activity: zeros, a constant, unit normal (the microbenchmarks' operands), a ReLU'd normal (what the
layer sees in a network: half the activations are exact zeros) and a normal scaled to 1e-3 (gradient-like magnitudes).  Same kernel, same
launch; rocm-smi samples power and shader clock while each variant loops.   python tools/probe/data_power_probe.py"""
import os, subprocess, sys, threading, time, re
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

B, H, W, Cin, Cout = 32, 128, 128, 3072, 768
P = B * H * W
fl = 2.0 * P * 9 * Cin * Cout / 1e12
g = torch.Generator(device='cuda').manual_seed(0)
wm = (torch.randn(Cout, 9 * Cin, device='cuda', generator=g) * 0.01).to(torch.bfloat16)
base = torch.randn(P, Cin, device='cuda', generator=g)
variants = {'zeros': torch.zeros(P, Cin, device='cuda').bfloat16(), 'constant 1.0': torch.ones(P, Cin, device='cuda').bfloat16(),
            'unit normal': base.bfloat16(), 'ReLU(normal)': base.clamp_min(0).bfloat16(), 'normal x 1e-3': (base * 1e-3).bfloat16()}
del base


def sample(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r'Power \(W\): ([\d.]+)', t); c = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', t)
            if p and c:
                out.append((float(p.group(1)), int(c.group(1))))
        except Exception:
            pass
        time.sleep(0.3)


for name, x in variants.items():
    for wname, w in (('random weights', wm), ('zero weights', torch.zeros_like(wm))) if name in ('unit normal', 'zeros') else (('random weights', wm),):
        for _ in range(3):
            hip.conv3x3(0, x, w, B, H, W, Cin, Cout)
        torch.cuda.synchronize()
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out)); th.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 150
        e0.record()
        for _ in range(n):
            hip.conv3x3(0, x, w, B, H, W, Cin, Cout)
        e1.record(); torch.cuda.synchronize()
        stop.set(); th.join()
        ms = e0.elapsed_time(e1) / n
        hot = out[len(out) // 3:] or out or [(0.0, 0)]
        print(f'{name:14s} x {wname:14s}: {ms:6.2f} ms  {fl / ms * 1e3:5.0f} TFLOP/s | power {sum(p for p, _ in hot) / len(hot):5.0f} W, sclk {sum(c for _, c in hot) / len(hot):5.0f} MHz ({len(hot)} samples)', flush=True)
