"""CPU restatement of the reference's TRAIN / VAL input transforms (test infrastructure only -- never imported by the product).

Reference: datasets/build_datasets.py:14-29 composes, for training,
    ExtRandomCrop(size) -> ExtColorJitter(0.5, 0.5, 0.5) -> ExtRandomHorizontalFlip() -> ExtToTensor() -> ExtNormalize(mean, std)
(datasets/extra_transform.py:319-392, 426-509, 196-214, 259-281, 288-313) and, for validation,
    ExtResize(size) -> ExtToTensor() -> ExtNormalize(mean, std)                      (extra_transform.py:395-419).
The dataset classes then map the uint8 label through a table and widen it to int64 (datasets/ade.py:122-124: 255 -> 0;
cityscapes.py:159 id_to_train_id; coco_stuff.py:95-100 label_map; voc.py:230 identity).

Third-party arithmetic.  The transforms call torchvision.transforms.functional (reference pin torchvision==0.15.2, absent from
this image) on PIL images; for PIL inputs those functions are thin calls into Pillow (reference pin pillow==9.3.0; 12.2.0 is
installed here and on the GPU box):
    F.crop -> Image.crop                F.hflip -> Image.transpose(FLIP_LEFT_RIGHT)
    F.adjust_brightness / contrast / saturation -> ImageEnhance.Brightness / Contrast / Color (.enhance(factor))
    F.resize -> Image.resize((w, h), BILINEAR / NEAREST) with the smaller edge matched to `size`, long edge int(size * long / short)
    F.to_tensor -> torch.from_numpy(np.array(pic)).permute(2, 0, 1).to(float32).div(255)
    F.normalize -> tensor.sub_(mean[:, None, None]).div_(std[:, None, None]) with float32 mean / std
This file restates Pillow's published algorithms (libImaging/Blend.c ImagingBlend, Convert.c rgb2l = L24 >> 16, ImageStat mean,
Resample.c ImagingResample with 22-bit fixed-point coefficients, Geometry.c nearest affine scaling) and is PINNED against Pillow
itself: oracle/make_input_goldens.py asserts bit-equality on the uint8 stages and writes tests/golden/input_pipeline_cases.npz
from Pillow + torch CPU outputs; tests/test_oracle_golden.py re-checks it on every box.  The random draws follow the
reference's call order on Python's `random` (crop i, j; brightness, contrast, saturation factors; shuffle; flip).
"""
import random

import numpy as np

OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION = 1, 2, 3


# ---- Pillow arithmetic ------------------------------------------------------------------------------------------------------
def rgb_to_l(img):
    """Convert.c rgb2l: L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16.  img uint8 [..., 3] -> uint8 [...]."""
    v = img.astype(np.int64)
    return ((v[..., 0] * 19595 + v[..., 1] * 38470 + v[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg, img, alpha):
    """Blend.c ImagingBlend(im1=deg, im2=img, alpha) in C float arithmetic: interpolation truncates, extrapolation clips then
    truncates.  (ImageEnhance._Enhance.enhance: Image.blend(self.degenerate, self.image, factor).)"""
    a = np.float32(alpha)
    if a == np.float32(0.0):
        return deg.copy()
    if a == np.float32(1.0):
        return img.copy()
    d = (img.astype(np.int32) - deg.astype(np.int32)).astype(np.float32)
    t = deg.astype(np.float32) + a * d                      # float32 multiply, then float32 add (no contraction)
    assert t.dtype == np.float32
    if np.float32(0.0) <= a <= np.float32(1.0):
        return t.astype(np.int32).astype(np.uint8)          # (UINT8) of a value inside [0, 255]: truncation
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def adjust_brightness(img, f):
    """ImageEnhance.Brightness: degenerate = black."""
    return blend(np.zeros_like(img), img, f)


def contrast_mean(img):
    """ImageEnhance.Contrast: int(ImageStat.Stat(image.convert('L')).mean[0] + 0.5)."""
    l = rgb_to_l(img)
    return int(float(l.astype(np.int64).sum()) / float(l.size) + 0.5)


def adjust_contrast(img, f):
    return blend(np.full_like(img, contrast_mean(img)), img, f)


def adjust_saturation(img, f):
    """ImageEnhance.Color: degenerate = image.convert('L').convert('RGB')."""
    return blend(np.repeat(rgb_to_l(img)[..., None], 3, axis=-1), img, f)


_ADJUST = {OP_BRIGHTNESS: adjust_brightness, OP_CONTRAST: adjust_contrast, OP_SATURATION: adjust_saturation}


def crop(a, top, left, h, w):
    """Image.crop((left, top, left+w, top+h)): area outside the source reads as 0 (extra_transform.py:388 via F.crop)."""
    out = np.zeros((h, w) + a.shape[2:], a.dtype)
    sh, sw = a.shape[:2]
    hh, ww = max(0, min(h, sh - top)), max(0, min(w, sw - left))
    out[:hh, :ww] = a[top:top + hh, left:left + ww]
    return out


def to_tensor_normalize(img, mean, std):
    """ExtToTensor + ExtNormalize incl. quirk Q11 (extra_transform.py:279, 311-313): ((u8 / 255) / 255 - mean) / std in float32,
    CHW."""
    t = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    t = t / np.float32(255)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return (t - m) / s


# ---- random draws in the reference's order ----------------------------------------------------------------------------------
def draw_train_params(rng, src_h, src_w, size, brightness=0.5, contrast=0.5, saturation=0.5, flip_p=0.5):
    """One sample's draws on `rng` (a random.Random or the `random` module), in the order the composed transforms make them:
    ExtRandomCrop.get_params (extra_transform.py:342-360), ExtColorJitter.get_params (:470-497), ExtRandomHorizontalFlip (:205-213).
    Returns dict(top, left, ops=[(op, factor), ...] in application order, flip)."""
    th, tw = size
    if src_w == tw and src_h == th:
        top, left = 0, 0
    else:
        top = rng.randint(0, abs(src_h - th))
        left = rng.randint(0, abs(src_w - tw))
    ops = []
    for op, amount in ((OP_BRIGHTNESS, brightness), (OP_CONTRAST, contrast), (OP_SATURATION, saturation)):
        if amount:                                         # _check_input: [max(0, 1 - v), 1 + v]; None when v == 0
            ops.append((op, rng.uniform(max(0.0, 1.0 - amount), 1.0 + amount)))
    rng.shuffle(ops)
    flip = rng.random() < flip_p
    return dict(top=top, left=left, ops=ops, flip=bool(flip))


def train_transform(img, lbl, p, size, mean, std, label_lut=None):
    """img uint8 [h, w, 3], lbl uint8 [h, w], p from draw_train_params -> (float32 [3, H, W], int64 [H, W])."""
    th, tw = size
    x = crop(img, p['top'], p['left'], th, tw)
    y = crop(lbl, p['top'], p['left'], th, tw)
    for op, f in p['ops']:
        x = _ADJUST[op](x, f)
    if p['flip']:
        x, y = x[:, ::-1], y[:, ::-1]
    y = y.astype(np.int64) if label_lut is None else np.asarray(label_lut, np.int64)[y]
    return to_tensor_normalize(np.ascontiguousarray(x), mean, std), np.ascontiguousarray(y)


# ---- validation: Pillow's bilinear / nearest resize -------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def resized_size(src_h, src_w, size):
    """torchvision F.resize with an int size: smaller edge -> size, other edge int(size * long / short); unchanged if equal."""
    short, long_ = (src_w, src_h) if src_w <= src_h else (src_h, src_w)
    if short == size:
        return src_h, src_w
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if src_w <= src_h else (new_short, new_long)


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the whole axis.
    Returns (bounds [out, 2] = (first tap, tap count), integer coefficients [out, ksize])."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        k = kk[xx]
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            w = 1.0 - abs(a) if abs(a) < 1.0 else 0.0
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64))
    return bounds, ik.astype(np.int32)


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _resample_axis(a, out_size, axis):
    a = np.moveaxis(a, axis, 0)
    bounds, ik = resample_coeffs(a.shape[0], out_size)
    out = np.empty((out_size,) + a.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(a.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += a[xmin + x].astype(np.int64) * int(ik[xx, x])
        out[xx] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resize_bilinear(img, out_h, out_w):
    """Image.resize((out_w, out_h), BILINEAR) on an RGB image: horizontal pass, then vertical pass, each rounded to uint8
    (Resample.c ImagingResampleInner; a pass whose size does not change is skipped)."""
    x = img
    if out_w != x.shape[1]:
        x = _resample_axis(x, out_w, 1)
    if out_h != x.shape[0]:
        x = _resample_axis(x, out_h, 0)
    return x


def resize_nearest(lbl, out_h, out_w):
    """Image.resize(..., NEAREST) = Geometry.c ImagingScaleAffine: source index = int(out_index * scale + scale / 2) evaluated as
    xo = a0 + a1 * 0.5; xin = COORD(xo) per pixel with xo += a1 (double accumulation)."""
    def idx(n_in, n_out):
        a1 = float(n_in) / n_out
        xo = a1 * 0.5
        out = np.empty(n_out, np.int64)
        for i in range(n_out):
            out[i] = int(xo)
            xo += a1
        return np.clip(out, 0, n_in - 1)
    iy, ix = idx(lbl.shape[0], out_h), idx(lbl.shape[1], out_w)
    return lbl[iy][:, ix]


def val_transform(img, lbl, size, mean, std, label_lut=None):
    oh, ow = resized_size(img.shape[0], img.shape[1], size)
    if (oh, ow) != img.shape[:2]:
        img, lbl = resize_bilinear(img, oh, ow), resize_nearest(lbl, oh, ow)
    y = lbl.astype(np.int64) if label_lut is None else np.asarray(label_lut, np.int64)[lbl]
    return to_tensor_normalize(img, mean, std), y
