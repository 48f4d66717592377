"""Streaming forward GEMM (gemm_skinny_kernel) on the stage-1 shapes: time per launch."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from segmentation_factory_amd import hip
for M, K, N in ((2097152, 32, 768), (2097152, 32, 128), (2097152, 128, 32), (2097152, 32, 32), (524288, 64, 256)):
    x = torch.randn(M, K, device='cuda').bfloat16(); w = torch.randn(N, K, device='cuda').bfloat16()
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(5): hip.gemm(0, x, w, M, N, K, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): hip.gemm(0, x, w, M, N, K, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f'[{M}x{K}]->{N}: {dt * 1e6:8.1f} us  {(M * (K + N) * 2) / dt / 1e12:5.2f} TB/s')
