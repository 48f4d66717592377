"""Device input pipeline (segmentation_factory_amd/transforms.py, csrc/input.hip) against the reference's transform stack.

CPU part: oracle/input_pipeline.py reproduces tests/golden/input_pipeline_cases.npz (written by oracle/make_input_goldens.py from
Pillow + torch CPU composed as datasets/build_datasets.py:14-29 composes them), agrees with Pillow itself where Pillow is
importable, and the host-side random draws follow the reference's call order.  GPU part: the HIP kernels, through the C ABI,
are BIT-EXACT against the golden outputs and the oracle (uint8 / int64 work and the four IEEE float32 operations of the tail)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import input_pipeline as IP

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'input_pipeline_cases.npz'), allow_pickle=False)


def _params(g, n):
    p = g[f'train{n}_params']
    ops = [(int(o), float(f)) for o, f in zip(p[3:], g[f'train{n}_factors'])]
    return dict(top=int(p[0]), left=int(p[1]), flip=bool(p[2]), ops=ops)


# ---- CPU ----------------------------------------------------------------------------------------------------------------------
def test_oracle_reproduces_golden_train_and_val(golden_dir):
    g = _gold(golden_dir)
    lut = g['label_lut']
    seen_ops, flips, oob = set(), 0, 0
    for n in range(int(g['train_count'])):
        p = _params(g, n)
        img, lbl = IP.train_transform(g[f'train{n}_img'], g[f'train{n}_lbl'], p, (32, 32), g['mean'], g['std'], lut)
        assert img.dtype == np.float32 and lbl.dtype == np.int64
        assert np.array_equal(img, g[f'train{n}_out_img']), n
        assert np.array_equal(lbl, g[f'train{n}_out_lbl']), n
        seen_ops.add(tuple(o for o, _ in p['ops']))
        flips += p['flip']
        oob += g[f'train{n}_img'].shape[0] < 32
    assert len(seen_ops) >= 4 and 0 < flips < int(g['train_count']) and oob >= 1       # the fixture exercises the branches
    for k in range(int(g['val_count'])):
        img, lbl = IP.val_transform(g[f'val{k}_img'], g[f'val{k}_lbl'], int(g['val_size']), g['mean'], g['std'], lut)
        assert np.array_equal(img, g[f'val{k}_out_img']), k
        assert np.array_equal(lbl, g[f'val{k}_out_lbl']), k


def test_oracle_against_pillow_when_importable():
    PIL = pytest.importorskip('PIL')
    from PIL import Image, ImageEnhance
    rng = np.random.default_rng(5)
    enh = {IP.OP_BRIGHTNESS: ImageEnhance.Brightness, IP.OP_CONTRAST: ImageEnhance.Contrast, IP.OP_SATURATION: ImageEnhance.Color}
    a = rng.integers(0, 256, (41, 29, 3), dtype=np.uint8)
    for f in (0.0, 0.5, 0.73, 1.0, 1.31, 1.5):
        for op, cls in enh.items():
            assert np.array_equal(IP._ADJUST[op](a, f), np.array(cls(Image.fromarray(a)).enhance(f))), (op, f)
    assert np.array_equal(IP.resize_bilinear(a, 32, 23), np.array(Image.fromarray(a).resize((23, 32), Image.BILINEAR)))
    assert np.array_equal(IP.resize_bilinear(a, 64, 45), np.array(Image.fromarray(a).resize((45, 64), Image.BILINEAR)))
    assert np.array_equal(IP.resize_nearest(a[..., 0], 32, 23), np.array(Image.fromarray(a[..., 0]).resize((23, 32), Image.NEAREST)))
    assert PIL.__version__


def test_host_draws_follow_the_reference_call_order():
    from segmentation_factory_amd.transforms import DeviceTrainTransform, DeviceValTransform
    t = DeviceTrainTransform(32, device='cpu', rng=random.Random(77))
    top, left, ops, flip = t.draw(50, 60)
    r = random.Random(77)
    want_top, want_left = r.randint(0, 18), r.randint(0, 28)                       # extra_transform.py:358-359
    b, c, s = r.uniform(0.5, 1.5), r.uniform(0.5, 1.5), r.uniform(0.5, 1.5)        # :477, :481, :485
    tl = [(1, b), (2, c), (3, s)]
    r.shuffle(tl)                                                                  # :492
    want_flip = r.random() < 0.5                                                   # :211
    assert (top, left, ops, flip) == (want_top, want_left, tl, want_flip)
    # the same stream as the oracle's restatement over several samples, incl. the no-draw case of an exactly fitting image
    t = DeviceTrainTransform(32, device='cpu', rng=random.Random(3))
    r = random.Random(3)
    for (h, w) in ((40, 52), (32, 32), (28, 36), (64, 33)):
        p = IP.draw_train_params(r, h, w, (32, 32))
        assert t.draw(h, w) == (p['top'], p['left'], p['ops'], p['flip'])
    v = DeviceValTransform(32, device='cpu')
    for (h, w) in ((37, 53), (20, 30), (32, 48), (90, 41), (64, 64), (500, 375)):
        assert v.output_size(h, w) == IP.resized_size(h, w, 32)
    with pytest.raises(RuntimeError):                                              # no CPU fallback
        t([torch.zeros(40, 52, 3, dtype=torch.uint8)], [torch.zeros(40, 52, dtype=torch.uint8)])


# ---- GPU ----------------------------------------------------------------------------------------------------------------------
def _dev_pair(img, lbl):
    return torch.from_numpy(np.ascontiguousarray(img)).cuda(), torch.from_numpy(np.ascontiguousarray(lbl)).cuda()


@pytest.mark.gpu
def test_train_kernel_bit_exact_on_golden(golden_dir):
    from segmentation_factory_amd.transforms import DeviceTrainTransform
    g = _gold(golden_dir)
    n = int(g['train_count'])
    t = DeviceTrainTransform(32, label_lut=torch.from_numpy(g['label_lut']))
    pairs = [_dev_pair(g[f'train{k}_img'], g[f'train{k}_lbl']) for k in range(n)]
    params = []
    for k in range(n):
        p = _params(g, k)
        params.append((p['top'], p['left'], p['ops'], p['flip']))
    img, lbl = t([a for a, _ in pairs], [b for _, b in pairs], params)              # one batch with ragged sources
    assert img.shape == (n, 3, 32, 32) and img.dtype == torch.float32 and lbl.shape == (n, 32, 32) and lbl.dtype == torch.int64
    img, lbl = img.cpu().numpy(), lbl.cpu().numpy()
    for k in range(n):
        assert np.array_equal(img[k], g[f'train{k}_out_img']), (k, np.abs(img[k] - g[f'train{k}_out_img']).max())
        assert np.array_equal(lbl[k], g[f'train{k}_out_lbl']), k


@pytest.mark.gpu
def test_val_kernel_bit_exact_on_golden(golden_dir):
    from segmentation_factory_amd.transforms import DeviceValTransform
    g = _gold(golden_dir)
    t = DeviceValTransform(int(g['val_size']), label_lut=torch.from_numpy(g['label_lut']))
    for k in range(int(g['val_count'])):
        a, b = _dev_pair(g[f'val{k}_img'], g[f'val{k}_lbl'])
        img, lbl = t(a, b)
        assert np.array_equal(img.cpu().numpy(), g[f'val{k}_out_img']), k
        assert np.array_equal(lbl.cpu().numpy(), g[f'val{k}_out_lbl']), k


@pytest.mark.gpu
@pytest.mark.parametrize('size', [(64, 64), (48, 37)])
def test_train_kernel_vs_oracle_ragged_and_unaligned(size):
    """Random draws through the product's own `draw`, sources of every alignment class (odd widths, crops hanging over the
    right / bottom edge, sub-views with padded row strides), W % 4 != 0 (one-pixel path)."""
    from segmentation_factory_amd.transforms import DeviceTrainTransform
    rng = np.random.default_rng(11)
    lut = np.arange(256, dtype=np.int64)
    lut[255] = 0
    t = DeviceTrainTransform(size, label_lut=torch.from_numpy(lut), rng=random.Random(5))
    srcs = [(70, 81), (64, 64), (50, 90), (91, 40), (size[0], size[1]), (65, 67), (130, 131), (30, 30)]
    imgs, lbls, host = [], [], []
    for k, (h, w) in enumerate(srcs):
        a = rng.integers(0, 256, (h, w + 3, 3), dtype=np.uint8)
        l = rng.integers(0, 256, (h, w + 5), dtype=np.uint8)
        da, dl = _dev_pair(a, l)
        imgs.append(da[:, (k % 3):(k % 3) + w])                     # row stride 3 * (w + 3), start offset 0 / 3 / 6 bytes
        lbls.append(dl[:, (k % 4):(k % 4) + w])
        host.append((a[:, (k % 3):(k % 3) + w], l[:, (k % 4):(k % 4) + w]))
    params = [t.draw(h, w) for (h, w) in srcs]
    img, lbl = t(imgs, lbls, params)
    img, lbl = img.cpu().numpy(), lbl.cpu().numpy()
    for k, (top, left, ops, flip) in enumerate(params):
        wi, wl = IP.train_transform(host[k][0], host[k][1], dict(top=top, left=left, ops=ops, flip=flip), size, MEAN, STD, lut)
        assert np.array_equal(lbl[k], wl), k
        assert np.array_equal(img[k], wi), (k, ops, flip)


@pytest.mark.gpu
def test_train_kernel_full_size_batch_properties():
    """BASELINE size (batch 128 x 512 x 512): (1) without jitter the output is a table look-up of the cropped (and mirrored)
    source -- checked for every pixel of the batch against a CPU-built table; (2) with jitter, three samples against the oracle;
    (3) mirroring commutes with the rest of the stack: flip=1 equals the x-reversed flip=0 batch; (4) two runs agree bitwise."""
    from segmentation_factory_amd.transforms import DeviceTrainTransform
    B, S = 128, 512
    gen = torch.Generator(device='cuda').manual_seed(0)
    srcs = [(S + (k % 7) * 13, S + (k % 5) * 29) for k in range(B)]
    imgs = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device='cuda', generator=gen) for h, w in srcs]
    lbls = [torch.randint(0, 256, (h, w), dtype=torch.uint8, device='cuda', generator=gen) for h, w in srcs]
    lut = torch.arange(256, dtype=torch.int64)
    lut[255] = 0
    r = random.Random(1)
    t = DeviceTrainTransform(S, label_lut=lut, rng=r)
    params = [t.draw(h, w) for h, w in srcs]
    plain = [(top, left, [], False) for (top, left, _, _) in params]
    img, lbl = t(imgs, lbls, plain)
    table = torch.from_numpy(np.stack([IP.to_tensor_normalize(np.full((1, 1, 3), v, np.uint8), MEAN, STD)[:, 0, 0] for v in range(256)], 1)).cuda()
    for k in range(B):
        top, left = plain[k][0], plain[k][1]
        crop = imgs[k][top:top + S, left:left + S].long()
        want = torch.stack([table[c][crop[..., c]] for c in range(3)])
        assert torch.equal(img[k], want), k
        assert torch.equal(lbl[k], lut.cuda()[lbls[k][top:top + S, left:left + S].long()]), k
    mirrored = [(top, left, [], True) for (top, left, _, _) in params]
    img_m, lbl_m = t(imgs, lbls, mirrored)
    assert torch.equal(img_m, img.flip(-1)) and torch.equal(lbl_m, lbl.flip(-1))
    out1 = t(imgs, lbls, params)
    out2 = t(imgs, lbls, params)
    assert torch.equal(out1[0], out2[0]) and torch.equal(out1[1], out2[1])
    for k in (0, 57, 127):
        top, left, ops, flip = params[k]
        wi, wl = IP.train_transform(imgs[k].cpu().numpy(), lbls[k].cpu().numpy(), dict(top=top, left=left, ops=ops, flip=flip),
                                    (S, S), MEAN, STD, lut.numpy())
        assert np.array_equal(out1[0][k].cpu().numpy(), wi), k
        assert np.array_equal(out1[1][k].cpu().numpy(), wl), k
    # mirrored jitter = x-reversed jitter (the contrast mean is a sum over the crop)
    flipped = [(top, left, ops, not flip) for (top, left, ops, flip) in params]
    out3 = t(imgs, lbls, flipped)
    assert torch.equal(out3[0], out1[0].flip(-1)) and torch.equal(out3[1], out1[1].flip(-1))


@pytest.mark.gpu
def test_val_kernel_vs_oracle_realistic_sizes():
    from segmentation_factory_amd.transforms import DeviceValTransform
    rng = np.random.default_rng(2)
    lut = np.arange(256, dtype=np.int64)
    lut[255] = 0
    t = DeviceValTransform(512, label_lut=torch.from_numpy(lut))
    for (h, w) in ((683, 512), (256, 341), (512, 600), (1024, 2048)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        l = rng.integers(0, 256, (h, w), dtype=np.uint8)
        img, lbl = t(*_dev_pair(a, l))
        wi, wl = IP.val_transform(a, l, 512, MEAN, STD, lut)
        assert img.shape == wi.shape
        assert np.array_equal(lbl.cpu().numpy(), wl), (h, w)
        assert np.array_equal(img.cpu().numpy(), wi), (h, w)


@pytest.mark.gpu
def test_device_batch_loader_feeds_train_one_epoch_shapes():
    from segmentation_factory_amd.transforms import DeviceBatchLoader, DeviceDataset, DeviceTrainTransform, label_table
    rng = np.random.default_rng(0)
    ds = DeviceDataset()
    for k in range(10):
        h, w = 70 + k, 90 - k
        ds.add(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 150, (h, w), dtype=np.uint8))
    loaders = [DeviceBatchLoader(ds, 2, DeviceTrainTransform(64, label_lut=label_table({255: 0}), rng=random.Random(r)), seed=4, rank=r,
                                 world=2) for r in range(2)]
    assert len(loaders[0]) == 2
    for ld in loaders:
        ld.set_epoch(3)
    batches = [list(ld) for ld in loaders]
    for bs in batches:
        assert len(bs) == 2
        for img, lbl in bs:
            assert img.shape == (2, 3, 64, 64) and img.dtype == torch.float32 and img.is_cuda
            assert lbl.shape == (2, 64, 64) and lbl.dtype == torch.int64 and int(lbl.max()) < 150
            assert torch.isfinite(img).all()


@pytest.mark.gpu
def test_zero_copy_feed_into_the_captured_step():
    """DeviceBatchLoader.bind_output: after the step is captured, batches are produced INSIDE its input buffers (the loader yields
    those very tensors, step() copies nothing), and what the step sees is exactly what an unbound loader with the same seeds
    yields."""
    import types
    from oracle import weights as OW
    from segmentation_factory_amd import SegmentationModel, engine
    from segmentation_factory_amd.optim import FusedAGCAdamW, NativeScaler
    from segmentation_factory_amd.transforms import DeviceBatchLoader, DeviceDataset, DeviceTrainTransform
    rng = np.random.default_rng(3)
    ds = DeviceDataset()
    for k in range(8):
        ds.add(rng.integers(0, 256, (80 + k, 90, 3), dtype=np.uint8), rng.integers(0, 5, (80 + k, 90), dtype=np.uint8))

    def make_loader():
        return DeviceBatchLoader(ds, 2, DeviceTrainTransform(64, rng=random.Random(11)), shuffle=True, seed=5)
    want = []
    ref_loader = make_loader()
    for ep in range(2):
        ref_loader.set_epoch(ep)
        want += [(a.clone(), b.clone()) for a, b in ref_loader]
    model = SegmentationModel('MiT-B0', num_classes=5, seg_head='SegFormerHead', compute_dtype=torch.bfloat16).cuda()
    model.load_state_dict(OW.make_state_dict('MiT-B0', 'SegFormerHead', 5, 7))
    model.train()
    opt = FusedAGCAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    loader = make_loader()
    seen, losses = [], []

    class Rec:
        def add_scalar(self, name, v, it=None):
            if name == 'train_loss':                       # called right after the step: what did the step read?
                gs = model._graphed_step
                seen.append((gs.static_inputs[0].clone(), gs.static_inputs[1].clone()))
                losses.append(float(v))
    args = types.SimpleNamespace(nb_classes=5, dice=True, ignore_index=255, ignore_label=255, local_rank=0, device='cuda', hip_graph=True)
    for ep in range(2):
        loader.set_epoch(ep)
        engine.train_one_epoch(model, opt, loader, ep, 'cuda', 1, 0.02, 'agc', NativeScaler(), Rec(), args)
    gs = model._graphed_step
    assert loader.out is not None and loader.out[0].data_ptr() == gs.static_inputs[0].data_ptr() \
        and loader.out[1].data_ptr() == gs.static_inputs[1].data_ptr()
    assert len(seen) == len(want) == 8 and all(np.isfinite(losses))
    for (si, sl), (wi, wl) in zip(seen, want):
        assert torch.equal(si, wi) and torch.equal(sl, wl)
    # the bound loader yields the step's own tensors: nothing left to copy
    it = iter(loader)
    bi, bl = next(it)
    assert bi.data_ptr() == gs.static_inputs[0].data_ptr() and bl.data_ptr() == gs.static_inputs[1].data_ptr()


@pytest.mark.gpu
def test_train_gpu_cli_with_device_input(tmp_path):
    """train_gpu.py --device-input --hip-graph --clip-mode norm: the captured step fed by the device pipeline (and the L2-norm
    clipping of the reference's --clip-mode norm in front of the optimizer kernel), end to end, loss falling."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'train_gpu.py'), '--dataset', 'synthetic', '--data_len', '16', '--image_size', '64',
           '--nb_classes', '5', '--backbone', 'MiT-B0', '--heads', 'SegFormerHead', '--batch-size', '4', '--val_batch_size', '2',
           '--epochs', '3', '--save_weights_dir', str(tmp_path / 'out'), '--writer_output', str(tmp_path), '--train_print_freq', '1',
           '--val_print_freq', '1', '--lr', '2e-3', '--device-input', '--hip-graph', '--clip-mode', 'norm', '--clip-grad', '1.0']
    r = subprocess.run(cmd, cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'Val_mIOU' in r.stdout
    import glob
    res = glob.glob(str(tmp_path / 'results*.txt'))                                # train_gpu.py:344-352: per-epoch results file
    assert res, r.stdout[-2000:]
    losses = [float(v) for v in re.findall(r'train_loss: ([0-9.]+)', open(res[0]).read())]
    assert len(losses) == 3 and losses[-1] < losses[0], losses
