"""Times segf_im2col on the stem geometry (fp32 NCHW image -> bf16 [B*128*128, 160] matrix, k7 s4 p3) at batch 128."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip
B, H, W = 128, 512, 512
x = torch.randn(B, 3, H, W, device='cuda')
for ld in (152, 160):
    for _ in range(3):
        col = hip.im2col(x, torch.bfloat16, True, B, H, W, 3, 7, 7, 4, 3, 128, 128, ld)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        col = hip.im2col(x, torch.bfloat16, True, B, H, W, 3, 7, 7, 4, 3, 128, 128, ld)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    byt = x.numel() * 4 + col.numel() * 2
    print(f'ld {ld}: {ms * 1e3:.1f} us, {byt / ms / 1e9:.2f} TB/s (image read once + matrix written)')
