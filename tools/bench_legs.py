#!/usr/bin/env python3
"""Secondary legs of bench.py, one child process per leg (a leg that fails or times out cannot take the headline line with it):

  config      one BASELINE.json config other than the headline (cfg3 / cfg4 / cfg5, cfg5 --fp8, cfg2 at the reference's default batch
              4 and at 16): images/s of the full graphed train step + its dominant kernel against that kernel's roofline
  train_loop  the loop a user runs -- `engine.train_one_epoch(..., args.hip_graph=True)` over a `DeviceBatchLoader` (the device-side
              transform stack writes every batch straight into the captured step's input buffers; /root/reference/engine.py:36-56,
              train_gpu.py:322-336) -- in steady state, next to the replay-only rate of the same captured step in the same process
  eval        `engine.evaluate` (/root/reference/engine.py:74-104) images/s in fp32 (the reference's eval precision, the default) and in
              bf16, with the eval forward replayed as a hipGraph and launched eagerly, plus the fused upsample + argmax +
              confusion-matrix kernel against the HBM roofline
  default_cli `engine.train_one_epoch` as the README command runs it (DataLoader-fed, batch 4, no flag = graph) next to --no-hip-graph

python tools/bench_legs.py config cfg3 --batch 64 [--fp8] | train_loop --batch 256 | eval --batch 1      -> one JSON line
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK, MFMA_FP8_PEAK, HBM_PEAK = 2.5e15, 5.0e15, 8.0e12          # MI355X_MICROARCH.md (dense peaks)
CONFIGS = {'cfg2': ('MiT-B0', 'SegFormerHead', 150, 512, 512), 'cfg3': ('ConvNeXt', 'UPerHead', 150, 512, 512),
           'cfg4': ('MiT-B2', 'SegFormerHead', 19, 1024, 2048), 'cfg5': ('convnextv2_large', 'UPerHead', 171, 640, 640)}


def synthetic_batch(batch, nc, H, W, seed):
    rng = np.random.default_rng(seed)                                   # numpy-seeded inputs
    img = rng.standard_normal((batch, 3, H, W), dtype=np.float32)
    lbl = rng.integers(0, nc, (batch, H, W), dtype=np.int64)
    lbl[:, :8] = 255
    lbl[rng.random((batch, H, W)) < 0.02] = 255
    return torch.from_numpy(img), torch.from_numpy(lbl)


def ev_time(fn, n, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n            # ms


def build(cfg, fp8=False):
    from segmentation_factory_amd import SegmentationModel
    from segmentation_factory_amd.optim import FusedAGCAdamW, param_groups_weight_decay
    bb, head, nc, H, W = CONFIGS[cfg]
    torch.manual_seed(1234)
    core = SegmentationModel(bb, num_classes=nc, seg_head=head, compute_dtype=torch.bfloat16).cuda().train()
    if fp8:
        core.set_fp8(True)
    from bench import numpy_seed_weights
    numpy_seed_weights(core, 0)          # the headline's initialisation (numpy default_rng(0)), not torch's RNG order
    opt = FusedAGCAdamW(param_groups_weight_decay(core, 0.025), lr=2e-4)
    return core, opt, nc, H, W


def dominant_kernel(cfg, batch, nc, H, W, fp8):
    """The kernel family that carries the config's step (profiles/r03f_cfg*_kernel_stats.csv), one representative launch timed with HIP
    events through the C ABI, against its roofline."""
    from segmentation_factory_amd import hip
    dev = 'cuda'
    if cfg in ('cfg3', 'cfg5'):
        # UPerHead bottleneck 3x3 conv 3072 -> 768 at stride 4 (heads/upernet.py:28), forward, as the eight-phase implicit GEMM
        h, w, Cin, Cout = H // 4, W // 4, 3072, 768
        P = batch * h * w
        x = torch.randn(P, Cin, device=dev).to(torch.bfloat16)
        wm = (torch.randn(Cout, 9 * Cin, device=dev) * 0.01).to(torch.bfloat16)
        fl = 2.0 * P * 9 * Cin * Cout
        if fp8:
            xq, sx = hip.quant_tensor_fp8(x)
            wq, sw = hip.quant_rows_fp8(wm)
            ms = ev_time(lambda: hip.conv3x3_fp8(0, xq, sx, wq, sw, batch, h, w, Cin, Cout), 3)
            peak, name = MFMA_FP8_PEAK, 'gemm8_kernel<0,0,CONV,FP8> (block-scaled fp8 MFMA)'
        else:
            ms = ev_time(lambda: hip.conv3x3(0, x, wm, batch, h, w, Cin, Cout), 3)
            peak, name = MFMA_BF16_PEAK, 'gemm8_kernel<0,0,CONV> (bf16 MFMA)'
        return {"kernel": f"{name}: UPerHead bottleneck 3x3 conv {Cin}->{Cout} @ {h}x{w}, batch {batch}, forward", "bound": "mfma",
                "achieved": round(fl / ms / 1e9, 1), "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(fl / (ms * 1e-3) / peak, 4),
                "avg_launch_ms": round(ms, 3), "flops_per_launch": fl, "traffic": None}
    if cfg == 'cfg4':
        # stage-1 attention of MiT-B2 at 1024 x 2048: 131072 queries x 2048 keys x head dim 64 (mit.py:43-59), forward + backward
        B, heads, N, Nkv, hd = batch, 1, (H // 4) * (W // 4), (H // 32) * (W // 32), 64
        q = torch.randn(B * N, hd, device=dev).to(torch.bfloat16)
        k = torch.randn(B * Nkv, hd, device=dev).to(torch.bfloat16)
        v = torch.randn(B * Nkv, hd, device=dev).to(torch.bfloat16)
        do = torch.randn(B * N, hd, device=dev).to(torch.bfloat16)
        o, lse = hip.attention_fwd(q, k, v, B, heads, N, Nkv, hd, hd ** -0.5)
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        ff = 4.0 * B * heads * N * Nkv * hd
        t0 = ev_time(lambda: hip.attention_fwd(q, k, v, B, heads, N, Nkv, hd, hd ** -0.5), 3)
        t1 = ev_time(lambda: hip.attention_bwd(q, k, v, o, do, lse, B, heads, N, Nkv, hd, hd ** -0.5, dk, dv), 3)
        return {"kernel": f"attn_mfma_fwd / bwd_dq + bwd_dkv <64>: {N} queries x {Nkv} keys x head dim 64, batch {batch} (MiT-B2 stage 1)",
                "bound": "mfma", "achieved": round(3.5 * ff / (t0 + t1) / 1e9, 1), "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": round(3.5 * ff / ((t0 + t1) * 1e-3) / MFMA_BF16_PEAK, 4), "fwd_ms": round(t0, 3), "fwd_TFLOPs": round(ff / t0 / 1e9, 1),
                "bwd_ms": round(t1, 3), "bwd_TFLOPs": round(2.5 * ff / t1 / 1e9, 1), "flops_per_launch": 3.5 * ff, "traffic": None}
    # cfg2: the fused loss backward (engine.py:10-15 backward), priced in bytes as SURVEY 8(d) prescribes
    hq, wq = H // 4, W // 4
    ld = (nc + 31) // 32 * 32
    lo = torch.randn(batch * hq * wq, ld, device=dev).to(torch.bfloat16)[:, :nc]
    _, y = synthetic_batch(batch, nc, H, W, 0)
    y = y.cuda()
    loss_, stats_, lse_ = hip.ce_dice_fwd(lo, batch, nc, hq, wq, H, W, y, 255, None, True, want_lse=True)
    go = torch.ones(1, device=dev)
    ms = ev_time(lambda: hip.ce_dice_bwd(lo, batch, nc, hq, wq, H, W, y, 255, None, True, stats_, go, lse=lse_), 5)
    nbytes = batch * (2 * hq * wq * nc * 2 + H * W * 8)
    return {"kernel": "ce_dice_bwd_band_kernel (fused transposed upsample + softmax + CE/Dice backward)", "bound": "hbm",
            "achieved": round(nbytes / ms / 1e6, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4),
            "avg_launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": nbytes, "traffic": None}


def leg_config(a):
    from segmentation_factory_amd import criterion_lowres
    from segmentation_factory_amd.graph import GraphedTrainStep
    core, opt, nc, H, W = build(a.config, a.fp8)
    x, y = synthetic_batch(a.batch, nc, H, W, 0)
    x, y = x.cuda(), y.cuda()

    def loss_fn(model, img, lbl):
        return criterion_lowres(model.forward_lowres(img), lbl, (H, W), None, num_classes=nc, dice=True, ignore_index=255)
    gs = GraphedTrainStep(core, opt, loss_fn, (x, y), clip_grad=0.02, clip_mode='agc')
    feed = tuple(gs.static_inputs)
    for _ in range(a.warmup):
        gs.step(*feed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = gs.step(*feed)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"config": a.config, "workload": f"{CONFIGS[a.config][0]} + {CONFIGS[a.config][1]}, {nc} classes, {H}x{W}, full train step "
                                           "(zero_grad+fwd+CE/Dice+bwd+AGC/AdamW), one hipGraph", "per_gpu_batch": a.batch, "fp8": bool(a.fp8),
           "dtype": "fp8 (e4m3 / e5m2 operands) + bf16" if a.fp8 else "bf16", "steps": a.steps, "warmup": a.warmup,
           "images_per_sec": round(a.batch * a.steps / el, 2), "ms_per_step": round(1e3 * el / a.steps, 3), "loss_after": round(loss.item(), 4),
           "peak_hbm_allocated_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}
    del gs
    out["roofline"] = dominant_kernel(a.config, a.batch, nc, H, W, a.fp8)
    return out


def leg_train_loop(a):
    """engine.train_one_epoch with --hip-graph over the device-side loader: what `train_gpu.py --hip-graph --device-input` runs per epoch."""
    import random as _random
    from segmentation_factory_amd.engine import train_one_epoch
    from segmentation_factory_amd.optim import NativeScaler
    from segmentation_factory_amd.transforms import DeviceBatchLoader, DeviceDataset, DeviceTrainTransform, label_table
    core, opt, nc, H, W = build('cfg2')
    steps_per_epoch = a.steps
    n_img = steps_per_epoch * a.batch
    n_src = min(n_img, 512)                                   # decoded uint8 sources resident in HBM (the loader re-draws crops / jitter)
    gen = torch.Generator(device='cuda').manual_seed(0)
    ds = DeviceDataset('cuda')
    srcs = [(H + 40 + (k % 7) * 13, W + 60 + (k % 5) * 29) for k in range(n_src)]
    imgs = [torch.randint(0, 256, (h_, w_, 3), dtype=torch.uint8, device='cuda', generator=gen) for h_, w_ in srcs]
    lbls = [torch.randint(0, nc, (h_, w_), dtype=torch.uint8, device='cuda', generator=gen) for h_, w_ in srcs]
    for k in range(n_img):                                    # the dataset lists every image; sources are shared storage
        ds.images.append(imgs[k % n_src])
        ds.labels.append(lbls[k % n_src])
    tf = DeviceTrainTransform((H, W), label_lut=label_table({255: 0}, 'cuda'), device='cuda', rng=_random.Random(0))
    loader = DeviceBatchLoader(ds, a.batch, tf, shuffle=True, seed=0)
    args = SimpleNamespace(nb_classes=nc, dice=True, ignore_index=255, hip_graph=True, local_rank=0)
    scaler = NativeScaler()
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        loader.set_epoch(0)
        train_one_epoch(core, opt, loader, 0, torch.device('cuda'), a.print_freq, 0.02, 'agc', scaler, None, args)     # capture + warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for ep in range(1, 1 + a.epochs):
            loader.set_epoch(ep)
            mean_loss, lr = train_one_epoch(core, opt, loader, ep, torch.device('cuda'), a.print_freq, 0.02, 'agc', scaler, None, args)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    n_steps = a.epochs * len(loader)
    gs = core._graphed_step
    feed = tuple(gs.static_inputs)
    for _ in range(3):
        gs.step(*feed)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n_steps):
        gs.step(*feed)
    torch.cuda.synchronize()
    rp = time.perf_counter() - t1
    loop_ips, replay_ips = a.batch * n_steps / el, a.batch * n_steps / rp
    return {"leg": "train_loop", "what": "engine.train_one_epoch(args.hip_graph) over transforms.DeviceBatchLoader (crop + colour jitter + flip + "
                                        "normalise on the device, zero-copy into the captured step), SegFormer-B0 512x512 150 classes, steady state "
                                        f"(epochs after the capturing one; one host sync per {a.print_freq} steps)",
            "per_gpu_batch": a.batch, "steps_timed": n_steps, "print_freq": a.print_freq, "images_per_sec": round(loop_ips, 2),
            "ms_per_step": round(1e3 * el / n_steps, 3), "replay_only_images_per_sec": round(replay_ips, 2),
            "ratio_to_replay_only": round(loop_ips / replay_ips, 4), "mean_loss_last_epoch": round(float(mean_loss), 4),
            "lines_printed": sink.getvalue().count('\n')}


def leg_eval(a):
    """engine.evaluate in the reference's precision (fp32: engine.py:86-88 switches autocast off) and in the bf16 production type
    (--eval-dtype bf16), each with the eval forward replayed as a hipGraph (the default) and launched eagerly."""
    from segmentation_factory_amd import hip
    from segmentation_factory_amd.engine import evaluate
    core, opt, nc, H, W = build('cfg2')
    core.eval()
    nb = a.steps
    x, y = synthetic_batch(a.batch, nc, H, W, 0)
    x, y = x.cuda(), y.cuda()
    data = [(x, y)] * nb                                          # batches already resident in HBM
    res = {}
    sink = io.StringIO()
    for dt in ('fp32', 'bf16'):
        res[dt] = {}
        for mode in ('graph', 'eager'):
            args = SimpleNamespace(nb_classes=nc, ignore_label=255, hip_graph=(mode == 'graph'), eval_dtype=dt)
            with contextlib.redirect_stdout(sink):
                _, mw = evaluate(args, core, data[:3], torch.device('cuda'), a.print_freq)    # warm (graph mode: eager first sighting, capture on the second)
                mw.compute_iou()                                                          # ... including the first use of the summary's own kernels
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                confmat, metric = evaluate(args, core, data, torch.device('cuda'), a.print_freq)
                miou = metric.compute_iou()[1]                    # reads the histogram: the synchronisation a user's loop ends with
                el = time.perf_counter() - t0
            res[dt][mode] = {"images_per_sec": round(a.batch * nb / el, 2), "ms_per_batch": round(1e3 * el / nb, 3), "mIoU": miou}
        res[dt]["speedup_graph_over_eager"] = round(res[dt]['graph']['images_per_sec'] / res[dt]['eager']['images_per_sec'], 3)
    hq, wq = H // 4, W // 4
    lo = core.forward_lowres(x)
    mat = torch.zeros(nc, nc, dtype=torch.int64, device='cuda')
    cnt = torch.zeros(nc, nc, dtype=torch.int64, device='cuda')
    flag = torch.zeros(1, dtype=torch.int32, device='cuda')
    ms = ev_time(lambda: hip.argmax_confmat(lo.data, a.batch, nc, hq, wq, H, W, y, 255, mat, cnt, flag), 10, warm=2)
    nbytes = a.batch * (hq * wq * nc * 2 + H * W * 8)             # low-res logits read + int64 labels read (SURVEY 8(d) definition)
    return {"leg": "eval", "what": "engine.evaluate (eval forward + fused upsample/argmax/confusion matrices), SegFormer-B0 512x512 150 classes, "
                                   f"{nb} batches resident in HBM; fp32 = the default (the reference's eval precision), bf16 = --eval-dtype bf16",
            "per_gpu_batch": a.batch, "eval_dtype_default": "fp32", "fp32": res['fp32'], "bf16": res['bf16'],
            "bf16_over_fp32": round(res['bf16']['graph']['images_per_sec'] / res['fp32']['graph']['images_per_sec'], 3),
            "roofline": {"kernel": "argmax_confmat_pix_kernel (fused bilinear upsample + argmax + int64 confusion matrices, engine.py:89-91; lane = pixel, "
                                   "counts privatised in LDS)",
                         "bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(nbytes / (ms * 1e-3) / HBM_PEAK, 4), "avg_launch_ms": round(ms, 4),
                         "algorithmic_bytes_per_launch": nbytes, "traffic": None,
                         "note": "VALU / LDS-latency-bound, not HBM-bound: 1 mul + 3 fma + compare + 2 selects per full-resolution (pixel, class) pair"}}


def leg_default_cli(a):
    """What `python train_gpu.py ...` as the reference's README launches it delivers per step (train_gpu.py:71,322-336; engine.py:36-56):
    engine.train_one_epoch fed by a torch DataLoader (host tensors, pinned, batch 4 = the reference default), args WITHOUT a hip_graph
    attribute (AUTO: the replayed graph) next to args.hip_graph=False (--no-hip-graph: per-kernel launches, one loss.item() per step)."""
    from torch.utils.data import DataLoader, TensorDataset
    from segmentation_factory_amd.engine import train_one_epoch
    from segmentation_factory_amd.optim import NativeScaler
    res = {}
    n_img = a.steps * a.batch
    sink = io.StringIO()
    for mode in ('auto', 'eager'):
        core, opt, nc, H, W = build('cfg2')
        x, y = synthetic_batch(min(n_img, 64), nc, H, W, 0)
        reps = -(-n_img // x.shape[0])
        ds = TensorDataset(x.repeat(reps, 1, 1, 1)[:n_img], y.repeat(reps, 1, 1)[:n_img])
        loader = DataLoader(ds, batch_size=a.batch, shuffle=True, drop_last=True, pin_memory=True, num_workers=0)
        args = SimpleNamespace(nb_classes=nc, dice=True, ignore_index=255, local_rank=0)
        if mode == 'eager':
            args.hip_graph = False
        scaler = NativeScaler()
        with contextlib.redirect_stdout(sink):
            train_one_epoch(core, opt, loader, 0, torch.device('cuda'), a.print_freq, 0.02, 'agc', scaler, None, args)   # capture / warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for ep in range(1, 1 + a.epochs):
                mean_loss, lr = train_one_epoch(core, opt, loader, ep, torch.device('cuda'), a.print_freq, 0.02, 'agc', scaler, None, args)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        n_steps = a.epochs * len(loader)
        res[mode] = {"images_per_sec": round(a.batch * n_steps / el, 2), "ms_per_step": round(1e3 * el / n_steps, 3),
                     "graphed": getattr(core, '_graphed_step', None) is not None, "mean_loss_last_epoch": round(float(mean_loss), 4)}
        del core, opt, loader, ds
        torch.cuda.empty_cache()
    return {"leg": "default_cli", "what": "engine.train_one_epoch over a torch DataLoader (pinned host tensors, H2D copy per step), SegFormer-B0 "
                                         "512x512 150 classes: 'auto' = the default command (hipGraph step, one host sync per logging interval), "
                                         "'eager' = --no-hip-graph", "per_gpu_batch": a.batch, "steps_timed": n_steps, "print_freq": a.print_freq,
            "auto": res['auto'], "eager": res['eager'],
            "speedup_auto_over_eager": round(res['auto']['images_per_sec'] / res['eager']['images_per_sec'], 3)}


def resolve_batch(config, batch=None):
    """The per-GPU batch a configuration is benchmarked at unless --batch says otherwise (bench.DEFAULT_BATCH)."""
    if batch is not None:
        return int(batch)
    from bench import DEFAULT_BATCH
    return DEFAULT_BATCH[config]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('leg', choices=['config', 'train_loop', 'eval', 'default_cli'])
    ap.add_argument('config', nargs='?', default='cfg2')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--fp8', action='store_true')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--print-freq', type=int, default=100, help='the reference default (train_gpu.py:74 --train_print_freq 100)')
    a = ap.parse_args()
    assert torch.cuda.is_available(), 'bench legs need the MI355X (there is no CPU fallback)'
    a.batch = resolve_batch(a.config, a.batch)
    out = {'config': leg_config, 'train_loop': leg_train_loop, 'eval': leg_eval, 'default_cli': leg_default_cli}[a.leg](a)
    print('LEG_JSON ' + json.dumps(out))


if __name__ == '__main__':
    main()
