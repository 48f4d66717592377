"""The nn.Linear products of cfg3 / cfg5 (ConvNeXt block MLPs, UPerHead 1 x 1 convs) on the kernel the dispatch picks today against the
eight-phase tile (policy gemm8_linear): per shape, microseconds and TFLOP/s of both, same process, same operands.
Usage (GPU box): python tools/probe/linear8_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

SHAPES = [  # (layout, M, N, K)  [cfg5 b8: ConvNeXtV2-L 640^2; cfg3 b32: ConvNeXt-T 512^2]
    (0, 12800, 3072, 768), (0, 12800, 768, 3072), (1, 12800, 3072, 768), (1, 12800, 768, 3072),
    (0, 51200, 1536, 384), (0, 51200, 384, 1536), (1, 51200, 1536, 384), (1, 51200, 384, 1536),
    (0, 204800, 768, 192), (0, 204800, 192, 768), (1, 204800, 768, 192), (1, 204800, 192, 768),
    (0, 3200, 6144, 1536), (0, 3200, 1536, 6144), (1, 3200, 6144, 1536), (1, 3200, 1536, 6144),
    (0, 32768, 1536, 384), (0, 32768, 384, 1536), (1, 32768, 1536, 384), (1, 32768, 384, 1536),
    (0, 131072, 768, 192), (0, 131072, 192, 768), (0, 524288, 384, 96), (0, 524288, 96, 384),
    (0, 8192, 3072, 768), (0, 8192, 768, 3072), (1, 8192, 3072, 768), (1, 8192, 768, 3072),
    # cfg4 b16: MiT-B2 1024 x 2048, stage 4 (512 wide, 32768 tokens) and the head
    (0, 32768, 512, 512), (1, 32768, 512, 512), (0, 32768, 1024, 512), (1, 32768, 512, 1024), (0, 32768, 2048, 512), (1, 32768, 2048, 512),
    (0, 32768, 512, 2048), (1, 32768, 512, 2048), (0, 32768, 512, 2880), (0, 32768, 768, 512), (1, 32768, 512, 768),
    (0, 131072, 1280, 320), (1, 131072, 1280, 320), (0, 131072, 320, 1280), (1, 131072, 320, 1280),
    # cfg2 b128 (MiT-B0: 160 / 256 wide stages)
    (0, 131072, 160, 640), (1, 131072, 160, 640), (0, 131072, 640, 160), (0, 32768, 256, 1024), (1, 32768, 256, 1024), (0, 32768, 1024, 256),
]


def time_one(layout, a, b, M, N, K, reps=20):
    for _ in range(3):
        hip.gemm(layout, a, b, M, N, K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.gemm(layout, a, b, M, N, K)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (layout, M, N, K) in SHAPES:
    a = torch.randn((M, K), device='cuda').bfloat16()
    b = torch.randn((N, K) if layout == 0 else (K, N), device='cuda').bfloat16()
    res = []
    for v in (0, 1, 2):                  # 0 = the 256 / 128 tile kernels of r04, 1 = today's rule, 2 = the eight-phase kernel wherever it can run
        hip.policy_set('gemm8_linear', 1 if v else 0)
        hip.policy_set('gemm8_linear_min_fill', 1 if v == 2 else 60)
        hip.policy_set('gemm8_linear_min_k', 256)
        hip.policy_set('gemm8_linear_min_tiles', 1 if v == 2 else 128)
        with hip.trace() as t:
            hip.gemm(layout, a, b, M, N, K)
        us = time_one(layout, a, b, M, N, K)
        res.append((us, t.kernels[0].split('<')[0] if t.kernels else '?'))
    hip.policy_reload()
    fl = 2.0 * M * N * K
    print(f'L{layout} M={M:7d} N={N:5d} K={K:5d}  r04 rule {res[0][0]:7.1f} us {fl / res[0][0] / 1e6:7.1f} TF/s ({res[0][1]})'
          f'  | today {res[1][0]:7.1f} us {fl / res[1][0] / 1e6:7.1f} TF/s ({res[1][1]})'
          f'  | forced {res[2][0]:7.1f} us {fl / res[2][0] / 1e6:7.1f} TF/s ({res[2][1]})')
