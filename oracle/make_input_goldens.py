"""Pins oracle/input_pipeline.py and writes tests/golden/input_pipeline_cases.npz.

The expected outputs are produced by Pillow (crop / ImageEnhance / transpose / resize) and torch CPU (to_tensor / normalize
arithmetic), composed as the reference's transforms compose them (datasets/build_datasets.py:14-29,
datasets/extra_transform.py); torchvision, whose functional wrappers sit between the two, is not installed here (see the header
of oracle/input_pipeline.py).  Every uint8 stage of the restatement is asserted bit-equal to Pillow's before anything is written.
Run:  python oracle/make_input_goldens.py
"""
import os
import random
import sys

import numpy as np
import torch
from PIL import Image, ImageEnhance

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import input_pipeline as ip                                     # noqa: E402

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
_ENH = {ip.OP_BRIGHTNESS: ImageEnhance.Brightness, ip.OP_CONTRAST: ImageEnhance.Contrast, ip.OP_SATURATION: ImageEnhance.Color}


def pil_to_tensor_normalize(pic):
    t = torch.from_numpy(np.array(pic)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)     # F.to_tensor
    t = t.float()
    t /= 255                                                                                         # ExtNormalize, quirk Q11
    mean = torch.as_tensor(MEAN, dtype=torch.float32)[:, None, None]
    std = torch.as_tensor(STD, dtype=torch.float32)[:, None, None]
    return t.clone().sub_(mean).div_(std).numpy()                                                    # F.normalize


def pil_train(img, lbl, p, size):
    th, tw = size
    pi, pl = Image.fromarray(img), Image.fromarray(lbl)
    box = (p['left'], p['top'], p['left'] + tw, p['top'] + th)
    pi, pl = pi.crop(box), pl.crop(box)                                                             # F.crop
    for op, f in p['ops']:
        pi = _ENH[op](pi).enhance(f)                                                                # F.adjust_*
    if p['flip']:
        pi, pl = pi.transpose(Image.FLIP_LEFT_RIGHT), pl.transpose(Image.FLIP_LEFT_RIGHT)           # F.hflip
    return pil_to_tensor_normalize(pi), np.array(pl, dtype='uint8'), np.array(pi)


def pil_val(img, lbl, size):
    oh, ow = ip.resized_size(img.shape[0], img.shape[1], size)
    pi, pl = Image.fromarray(img), Image.fromarray(lbl)
    if (oh, ow) != img.shape[:2]:
        pi, pl = pi.resize((ow, oh), Image.BILINEAR), pl.resize((ow, oh), Image.NEAREST)            # F.resize
    return pil_to_tensor_normalize(pi), np.array(pl, dtype='uint8'), np.array(pi)


def smooth_image(rng, h, w):
    """Photo-like content: smooth gradients + noise, full 0..255 range, so the blends hit both the clipping and the truncation
    branches."""
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 255 // max(h + w - 2, 1))], -1)
    return np.clip(base + rng.integers(-40, 41, (h, w, 3)), 0, 255).astype(np.uint8)


def main():
    rng = np.random.default_rng(20260)
    out = {}
    lut = np.arange(256, dtype=np.int64)
    lut[255] = 0                                                    # datasets/ade.py:123
    # --- unit checks of the restated Pillow arithmetic on random data, far more values than the fixture holds
    for _ in range(20):
        a = rng.integers(0, 256, (33, 47, 3), dtype=np.uint8)
        assert np.array_equal(ip.rgb_to_l(a), np.array(Image.fromarray(a).convert('L')))
        for f in (0.0, 1.0, 0.5, 1.5, float(rng.uniform(0.5, 1.5)), float(rng.uniform(0.0, 2.0))):
            for op in (ip.OP_BRIGHTNESS, ip.OP_CONTRAST, ip.OP_SATURATION):
                got = ip._ADJUST[op](a, f)
                want = np.array(_ENH[op](Image.fromarray(a)).enhance(f))
                assert np.array_equal(got, want), (op, f)
    for (h, w, oh, ow) in ((37, 53, 32, 45), (20, 30, 32, 48), (64, 64, 32, 32), (50, 31, 51, 32), (200, 300, 32, 48)):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        l = rng.integers(0, 256, (h, w), dtype=np.uint8)
        assert np.array_equal(ip.resize_bilinear(a, oh, ow), np.array(Image.fromarray(a).resize((ow, oh), Image.BILINEAR)))
        assert np.array_equal(ip.resize_nearest(l, oh, ow), np.array(Image.fromarray(l).resize((ow, oh), Image.NEAREST)))
    # --- training cases: (source h, w, crop size); the third source is smaller than the crop (Image.crop pads with 0)
    size = (32, 32)
    train_src = [(40, 52), (32, 32), (28, 36), (45, 33), (64, 80), (33, 70)]
    r = random.Random(1234)
    n = 0
    for rep in range(3):
        for (h, w) in train_src:
            img = smooth_image(rng, h, w) if (n % 2 == 0) else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            lbl = rng.integers(0, 151, (h, w), dtype=np.uint8)
            lbl[rng.random((h, w)) < 0.05] = 255
            p = ip.draw_train_params(r, h, w, size)
            want_img, want_lbl_u8, want_u8 = pil_train(img, lbl, p, size)
            got_img, got_lbl = ip.train_transform(img, lbl, p, size, MEAN, STD, lut)
            assert np.array_equal(got_lbl, lut[want_lbl_u8]), n
            assert np.array_equal(got_img, want_img), (n, np.abs(got_img - want_img).max())
            out[f'train{n}_img'], out[f'train{n}_lbl'] = img, lbl
            out[f'train{n}_params'] = np.array([p['top'], p['left'], int(p['flip'])] + [o for o, _ in p['ops']], np.int64)
            out[f'train{n}_factors'] = np.array([f for _, f in p['ops']], np.float64)
            out[f'train{n}_out_img'], out[f'train{n}_out_lbl'] = want_img, lut[want_lbl_u8]
            n += 1
    out['train_count'] = np.array(n)
    # the draw order itself: the same seed through the reference's sequence of random calls, written out literally
    r1, r2 = random.Random(77), random.Random(77)
    p = ip.draw_train_params(r1, 50, 60, size)
    top, left = r2.randint(0, 18), r2.randint(0, 28)                             # extra_transform.py:358-359
    b, c, s = r2.uniform(0.5, 1.5), r2.uniform(0.5, 1.5), r2.uniform(0.5, 1.5)   # :477,481,485
    t = [(1, b), (2, c), (3, s)]
    r2.shuffle(t)                                                                # :492
    flip = r2.random() < 0.5                                                     # :211
    assert (p['top'], p['left'], p['ops'], p['flip']) == (top, left, t, flip)
    # --- validation cases
    val_src = [(37, 53), (20, 30), (32, 48), (90, 41), (64, 64)]
    for k, (h, w) in enumerate(val_src):
        img = smooth_image(rng, h, w) if k % 2 == 0 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        lbl = rng.integers(0, 151, (h, w), dtype=np.uint8)
        lbl[rng.random((h, w)) < 0.05] = 255
        want_img, want_lbl_u8, _ = pil_val(img, lbl, 32)
        got_img, got_lbl = ip.val_transform(img, lbl, 32, MEAN, STD, lut)
        assert np.array_equal(got_lbl, lut[want_lbl_u8]), k
        assert np.array_equal(got_img, want_img), k
        out[f'val{k}_img'], out[f'val{k}_lbl'] = img, lbl
        out[f'val{k}_out_img'], out[f'val{k}_out_lbl'] = want_img, lut[want_lbl_u8]
    out['val_count'] = np.array(len(val_src))
    out['val_size'] = np.array(32)
    out['label_lut'] = lut
    out['mean'], out['std'] = np.array(MEAN), np.array(STD)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'input_pipeline_cases.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', n, 'train cases,', len(val_src), 'val cases; Pillow', Image.__version__)


if __name__ == '__main__':
    main()
