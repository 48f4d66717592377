"""Time individual segf_gemm shapes (bf16, the dispatch the model uses) under debug switches.
Usage: python tools/gemm_bench.py [ENV=VALUE ...]   e.g.  SEGFAC_GEMM_NO_FASTLOAD=1"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kv in sys.argv[1:]:
    k, v = kv.split('=')
    os.environ[k] = v
from segmentation_factory_amd import hip

SHAPES = [  # (layout, M, N, K, f32_out)
    (2, 152, 768, 1048576, 1), (1, 1048576, 768, 152, 0), (0, 1048576, 152, 768, 0), (2, 768, 32, 1048576, 1),
    (1, 1048576, 32, 768, 0), (2, 32, 147, 1048576, 1), (2, 32, 32, 1048576, 1), (2, 256, 256, 16384, 1),
    (2, 160, 640, 65536, 1), (2, 640, 160, 65536, 1), (2, 160, 160, 65536, 1), (0, 262144, 64, 256, 0),
    (2, 1024, 256, 16384, 1), (2, 256, 1440, 16384, 1), (0, 65536, 160, 160, 0), (0, 65536, 640, 160, 0),
    (1, 65536, 160, 640, 0), (0, 16384, 256, 256, 0), (0, 16384, 1024, 256, 0), (0, 16384, 256, 1024, 0),
]
for (layout, M, N, K, f32o) in SHAPES:
    a = torch.randn((K, M) if layout == 2 else (M, K), device='cuda').bfloat16()
    b = torch.randn((N, K) if layout == 0 else (K, N), device='cuda').bfloat16()
    sk = hip.pick_splitk(M, N, K) if layout == 2 else 1
    kw = dict(out_dtype=torch.float32 if f32o else None, split_k=sk)
    for _ in range(3):
        hip.gemm(layout, a, b, M, N, K, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        hip.gemm(layout, a, b, M, N, K, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    byt = 2 * (M * K + K * N) + (4 if f32o else 2) * M * N
    print(f'L{layout} M={M:8d} N={N:5d} K={K:8d} sk={sk:3d}  {dt * 1e6:8.1f} us  {byt / dt / 1e12:5.2f} TB/s  {2.0 * M * N * K / dt / 1e12:7.1f} TFLOP/s')
