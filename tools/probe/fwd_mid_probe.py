"""Forward / data-gradient GEMMs of MiT stages 3 and 4 (mid-size, short K): 256-tile vs 128-tile kernel."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip
shapes = [(131072, 640, 160), (131072, 160, 640), (131072, 160, 160), (32768, 1024, 256), (32768, 256, 1024), (32768, 256, 256),
          (131072, 320, 160), (524288, 64, 64)]
for layout in (0, 1):
    for M, N, K in shapes:
        x = torch.randn(M, K, device='cuda').bfloat16()
        w = torch.randn((N, K) if layout == 0 else (K, N), device='cuda').bfloat16()
        out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        res = []
        for name, env in (('default', {}), ('128', {'SEGFAC_GEMM_NO_BIG': '1'})):
            os.environ.pop('SEGFAC_GEMM_NO_BIG', None); os.environ.update(env)
            for _ in range(5): hip.gemm(layout, x, w, M, N, K, out=out)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): hip.gemm(layout, x, w, M, N, K, out=out)
            torch.cuda.synchronize(); res.append((name, (time.perf_counter() - t0) / 20 * 1e6))
        ideal = (M * (K + N) * 2 + N * K * 2) / 6e12 * 1e6
        print(f'layout {layout} [{M}x{K}]->{N}: ' + '  '.join(f'{n} {t:.1f}us' for n, t in res) + f'   (HBM {ideal:.0f}us)', flush=True)
