"""Weight gradients with mid-size outputs and token-count K (MiT stages 3 / 4): time by split-K and tile kernel."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

shapes = [(256, 256, 32768), (1024, 256, 32768), (256, 1024, 32768), (512, 256, 32768), (160, 160, 131072), (640, 160, 131072),
          (160, 640, 131072), (64, 256, 524288), (256, 64, 524288)]
for M, N, K in shapes:
    dy = torch.randn(K, M, device='cuda').bfloat16()
    x = torch.randn(K, N, device='cuda').bfloat16()
    res = []
    for big in (True, False):
        if big: os.environ.pop('SEGFAC_GEMM_NO_BIG', None)
        else: os.environ['SEGFAC_GEMM_NO_BIG'] = '1'
        default = hip.pick_splitk(M, N, K)
        for split in sorted(set([default, 8, 16, 32, 64, 128, 256])):
            if K // split < 256: continue
            try:
                for _ in range(3): hip.gemm_dw_db(dy, x, M, N, K, split_k=split)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10): hip.gemm_dw_db(dy, x, M, N, K, split_k=split)
                torch.cuda.synchronize()
                res.append(((time.perf_counter() - t0) / 10 * 1e6, 'big' if big else '128', split, split == default))
            except Exception as e:
                res.append((1e9, 'big' if big else '128', split, False))
    res.sort()
    print(f'[{M}x{N}] K={K}: ' + '  '.join(f'{t:.0f}us {k}/s{s}{"*" if d else ""}' for t, k, s, d in res[:6]) +
          '   | defaults: ' + '  '.join(f'{t:.0f}us {k}/s{s}' for t, k, s, d in res if d), flush=True)
