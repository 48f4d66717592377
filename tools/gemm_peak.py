import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from segmentation_factory_amd import hip
for (layout, M, N, K) in [(0, 8192, 8192, 8192), (0, 262144, 768, 6912), (0, 16384, 4096, 4096), (1, 8192, 8192, 8192), (2, 4096, 4096, 65536)]:
    a = torch.randn((K, M) if layout == 2 else (M, K), device='cuda').bfloat16() * 0.1
    b = torch.randn((N, K) if layout == 0 else (K, N), device='cuda').bfloat16() * 0.1
    kw = dict(out_dtype=torch.float32, split_k=hip.pick_splitk(M, N, K)) if layout == 2 else {}
    for _ in range(2): hip.gemm(layout, a, b, M, N, K, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): hip.gemm(layout, a, b, M, N, K, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'L{layout} {M}x{N}x{K}: {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.0f} TFLOP/s')
    t0 = time.perf_counter()
    bb = b.t() if layout == 0 else b
    aa = a.t() if layout == 2 else a
    for _ in range(5): torch.matmul(aa, bb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f'   torch.matmul (hipBLASLt): {dt*1e3:.3f} ms  {2.0*M*N*K/dt/1e12:.0f} TFLOP/s')
