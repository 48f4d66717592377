"""Times segf_input_train at the BASELINE batch (128 x 512 x 512 crops of ~600 x 700 decoded images resident in HBM).
Algorithmic bytes per output pixel: 3 (RGB) + 1 (label) read, 12 (fp32 x 3) + 8 (int64) written = 24."""
import json
import random
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd.transforms import DeviceTrainTransform, label_table  # noqa: E402

B, S = int(os.environ.get('B', 128)), 512
gen = torch.Generator(device='cuda').manual_seed(0)
srcs = [(S + 88 + (k % 7) * 13, S + 188 + (k % 5) * 29) for k in range(B)]
imgs = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device='cuda', generator=gen) for h, w in srcs]
lbls = [torch.randint(0, 256, (h, w), dtype=torch.uint8, device='cuda', generator=gen) for h, w in srcs]
t = DeviceTrainTransform(S, label_lut=label_table({255: 0}), rng=random.Random(0))
for name, jit in (('full stack', True), ('crop + flip + normalise only', False)):
    params = [t.draw(h, w) for h, w in srcs]
    if not jit:
        params = [(a, b, [], f) for (a, b, _, f) in params]
    samples = t.pack(imgs, lbls, params)
    from segmentation_factory_amd import hip
    out = hip.input_train(samples, B, S, S, t.mean, t.std, t.label_lut)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.input_train(samples, B, S, S, t.mean, t.std, t.label_lut, out[0], out[1])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gb = B * S * S * 24 / 1e9
    print(json.dumps({'kernel': 'segf_input_train', 'case': name, 'batch': B, 'ms': round(ms, 4), 'algorithmic_GB': round(gb, 3),
                      'GB_per_s': round(gb / ms * 1e3, 1), 'img_per_s': round(B / ms * 1e3)}))
# host side: draws + record packing + upload per batch
import time
t0 = time.perf_counter()
for _ in range(20):
    t.pack(imgs, lbls, [t.draw(h, w) for h, w in srcs])
print(json.dumps({'host_draw_pack_upload_ms_per_batch': round((time.perf_counter() - t0) / 20 * 1e3, 3)}))
