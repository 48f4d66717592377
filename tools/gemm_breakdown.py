"""Per-shape GEMM time of one eager SegFormer-B0 train step (HIP events around every segf_gemm launch), against each
shape's own roofline (max of HBM time at 6 TB/s and MFMA time at 2.5 PFLOP/s).  Usage: python tools/gemm_breakdown.py [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from segmentation_factory_amd import SegmentationModel, functional as Fh, hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NC, H, W = 150, 512, 512
torch.manual_seed(0)
m = SegmentationModel('MiT-B0', num_classes=NC, seg_head='SegFormerHead').cuda().train()
x, y = bench.synthetic_batch(B, 0)
x, y = x.cuda(), y.cuda()


def step():
    for p in m.parameters():
        p.grad = None
    lo = m.forward_lowres(x)
    loss, _, _ = Fh.upsample_ce_dice(lo.data, y, (B, NC, lo.H, lo.W, H, W), 255, None, True)
    loss.backward()


for _ in range(2):
    step()
with hip.KernelTimer(lambda k: k[0] == 'gemm') as t:
    for _ in range(3):
        step()
s = t.summary()
rows = []
for (_, layout, M, N, K), (n, avg) in s.items():
    n //= 3
    byt = 2 * (M * K + K * N) + (2 if layout != 2 else 4) * M * N
    ideal = max(byt / 6e12, 2.0 * M * N * K / 2.5e15) * 1e3
    rows.append((n * avg, n, layout, M, N, K, avg, ideal))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'total gemm ms/step {tot:.2f}  ideal {sum(r[1] * r[7] for r in rows):.2f}')
print('  ms/step calls layout        M      N      K   avg_us  ideal_us  eff')
for r in rows[:40]:
    print(f'{r[0]:8.3f} {r[1]:5d} {r[2]:6d} {r[3]:8d} {r[4]:6d} {r[5]:6d} {r[6] * 1e3:8.1f} {r[7] * 1e3:8.1f} {r[7] / r[6]:5.2f}')
