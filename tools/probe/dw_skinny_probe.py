"""Time segf_gemm_dw_db on the small-output weight-gradient shapes (streaming kernel vs tiled kernel)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

shapes = [(32, 147, 2097152, 152), (32, 32, 2097152, 32), (32, 128, 2097152, 128), (128, 32, 2097152, 32), (64, 256, 524288, 256),
          (256, 64, 524288, 64), (64, 64, 524288, 64)]
if len(sys.argv) > 1:
    shapes = [shapes[int(sys.argv[1])]]
for M, N, K, ldx in shapes:
    dy = torch.randn(K, M, device='cuda').bfloat16()
    x = torch.randn(K, ldx, device='cuda').bfloat16()
    for name, env in (('stream', None), ('tiled', '1')):
        if env: os.environ['SEGFAC_GEMM_NO_DW_SKINNY'] = env
        else: os.environ.pop('SEGFAC_GEMM_NO_DW_SKINNY', None)
        split = hip.pick_splitk(M, N, K)
        for _ in range(3): hip.gemm_dw_db(dy, x[:, :N], M, N, K, split_k=split)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): hip.gemm_dw_db(dy, x[:, :N], M, N, K, split_k=split)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        gb = K * (M + N) * 2 / 1e9
        print(f'[{M}x{N}] K={K} {name}: split {split} {ms * 1e3:8.1f} us  {gb / ms:6.2f} TB/s', flush=True)
