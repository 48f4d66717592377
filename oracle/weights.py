"""Architecture tables + numpy-seeded state_dict factory (TEST INFRASTRUCTURE).

The key names / shapes restate what the reference modules register
(reference ``models/backbones/mit.py:102-190``,
``models/heads/segformer.py:32-40``, ``models/backbones/convnext.py:26-107``,
``models/backbones/convnextv2.py:68-170``, ``models/heads/upernet.py:18-33``,
``models/modules/ppm.py:10-20``, ``models/heads/fpn.py:14-25``,
``models/backbones/mobilenetv2.py:5-70``).  ``oracle/make_goldens.py`` asserts
that the key set and shapes equal the imported reference's ``state_dict()``.

Weights come from ``numpy.random.default_rng(seed)`` so that they are
identical in this container and on the GPU box (never torch RNG: the
reference constructs the SegFormer head twice, ``models/build_models.py:43-54``).
"""
from collections import OrderedDict

import numpy as np
import torch

# --- architecture tables --------------------------------------------------------
# reference models/backbones/mit.py:149-156 (embed dims, depths); heads / sr per stage :176-188
MIT_SETTINGS = {
    'B0': ([32, 64, 160, 256], [2, 2, 2, 2]),
    'B1': ([64, 128, 320, 512], [2, 2, 2, 2]),
    'B2': ([64, 128, 320, 512], [3, 4, 6, 3]),
    'B3': ([64, 128, 320, 512], [3, 4, 18, 3]),
    'B4': ([64, 128, 320, 512], [3, 8, 27, 3]),
    'B5': ([64, 128, 320, 512], [3, 6, 40, 3]),
}
MIT_HEADS = [1, 2, 5, 8]
MIT_SR = [8, 4, 2, 1]
MIT_PATCH = [(7, 4), (3, 2), (3, 2), (3, 2)]  # (kernel, stride); padding = kernel // 2
MIT_DROP_PATH = 0.1

# reference models/backbones/convnext.py:70-76 -- only 'T' is reachable by name ("ConvNeXt()")
CONVNEXT_SETTINGS = {
    'T': ([3, 3, 9, 3], [96, 192, 384, 768], 0.1),
    'S': ([3, 3, 27, 3], [96, 192, 384, 768], 0.4),
    'B': ([3, 3, 27, 3], [128, 256, 512, 1024], 0.5),
    'L': ([3, 3, 27, 3], [192, 384, 768, 1536], 0.5),
    'XL': ([3, 3, 27, 3], [256, 512, 1024, 2048], 0.5),
}
# reference models/backbones/convnextv2.py:182-234 (factory functions)
CONVNEXTV2_SETTINGS = {
    'convnextv2_atto': ([2, 2, 6, 2], [40, 80, 160, 320], 0.0),
    'convnextv2_femto': ([2, 2, 6, 2], [48, 96, 192, 384], 0.0),
    'convnext_pico': ([2, 2, 6, 2], [64, 128, 256, 512], 0.0),
    'convnextv2_nano': ([2, 2, 8, 2], [80, 160, 320, 640], 0.0),
    'convnextv2_tiny': ([3, 3, 9, 3], [96, 192, 384, 768], 0.1),
    'convnextv2_base': ([3, 3, 27, 3], [128, 256, 512, 1024], 0.4),
    'convnextv2_large': ([3, 3, 27, 3], [192, 384, 768, 1536], 0.5),
    'convnextv2_huge': ([3, 3, 27, 3], [352, 704, 1408, 2816], 0.5),
}
# reference models/backbones/mobilenetv2.py:52-61  (t, c, n, s)
MBV2_SETTING = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
                (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]
MBV2_TAPS = [3, 6, 13, 17]
MBV2_CHANNELS = [24, 32, 96, 320]
PPM_SCALES = (1, 2, 3, 6)


def backbone_channels(backbone: str):
    if backbone.startswith('MiT'):
        return MIT_SETTINGS[backbone.split('-')[1]][0]
    if backbone == 'ConvNeXt':
        return CONVNEXT_SETTINGS['T'][1]
    if backbone in CONVNEXTV2_SETTINGS:
        return CONVNEXTV2_SETTINGS[backbone][1]
    if backbone == 'MobileNetV2':
        return MBV2_CHANNELS
    raise KeyError(backbone)


def head_width(backbone: str) -> int:
    """Quirk Q1 (reference models/build_models.py:25-27,53-54): for 'MiT-Bx' the name has
    been rebound to 'MiT' before the test, so every MiT gets 768."""
    name = 'MiT' if backbone.startswith('MiT') else backbone
    return 128 if ('tiny' in name or 'small' in name) else 768


def drop_path_rates(backbone: str):
    """linspace(0, rate, sum(depths)) -- mit.py:173, convnext.py:93, convnextv2.py:148."""
    if backbone.startswith('MiT'):
        depths, rate = MIT_SETTINGS[backbone.split('-')[1]][1], MIT_DROP_PATH
    elif backbone == 'ConvNeXt':
        depths, _, rate = CONVNEXT_SETTINGS['T']
    elif backbone in CONVNEXTV2_SETTINGS:
        depths, _, rate = CONVNEXTV2_SETTINGS[backbone]
    else:
        return []
    return [float(v) for v in torch.linspace(0, rate, sum(depths))]


# --- parameter inventories ------------------------------------------------------
def _bn(d, p, c):
    d[p + 'weight'] = ((c,), 'norm_w')
    d[p + 'bias'] = ((c,), 'norm_b')
    d[p + 'running_mean'] = ((c,), 'run_mean')
    d[p + 'running_var'] = ((c,), 'run_var')
    d[p + 'num_batches_tracked'] = ((), 'count')


def _ln(d, p, c):
    d[p + 'weight'] = ((c,), 'norm_w')
    d[p + 'bias'] = ((c,), 'norm_b')


def _lin(d, p, cin, cout, bias=True):
    d[p + 'weight'] = ((cout, cin), 'w')
    if bias:
        d[p + 'bias'] = ((cout,), 'b')


def _conv(d, p, cin, cout, k, groups=1, bias=True):
    d[p + 'weight'] = ((cout, cin // groups, k, k), 'w')
    if bias:
        d[p + 'bias'] = ((cout,), 'b')


def mit_inventory(variant, prefix='backbone.'):
    dims, depths = MIT_SETTINGS[variant]
    d = OrderedDict()
    cin = 3
    for s in range(4):
        c = dims[s]
        k, _ = MIT_PATCH[s]
        _conv(d, f'{prefix}patch_embed{s + 1}.proj.', cin, c, k)
        _ln(d, f'{prefix}patch_embed{s + 1}.norm.', c)
        cin = c
    for s in range(4):
        c = dims[s]
        for j in range(depths[s]):
            p = f'{prefix}block{s + 1}.{j}.'
            _ln(d, p + 'norm1.', c)
            _lin(d, p + 'attn.q.', c, c)
            _lin(d, p + 'attn.kv.', c, 2 * c)
            _lin(d, p + 'attn.proj.', c, c)
            if MIT_SR[s] > 1:
                _conv(d, p + 'attn.sr.', c, c, MIT_SR[s])
                _ln(d, p + 'attn.norm.', c)
            _ln(d, p + 'norm2.', c)
            _lin(d, p + 'mlp.fc1.', c, 4 * c)
            _conv(d, p + 'mlp.dwconv.dwconv.', 4 * c, 4 * c, 3, groups=4 * c)
            _lin(d, p + 'mlp.fc2.', 4 * c, c)
        _ln(d, f'{prefix}norm{s + 1}.', c)
    return d


def convnext_inventory(depths, dims, v2=False, prefix='backbone.'):
    d = OrderedDict()
    _conv(d, f'{prefix}downsample_layers.0.0.', 3, dims[0], 4)
    _ln(d, f'{prefix}downsample_layers.0.1.', dims[0])
    for i in range(3):
        _ln(d, f'{prefix}downsample_layers.{i + 1}.0.', dims[i])
        _conv(d, f'{prefix}downsample_layers.{i + 1}.1.', dims[i], dims[i + 1], 2)
    for i in range(4):
        c = dims[i]
        for j in range(depths[i]):
            p = f'{prefix}stages.{i}.{j}.'
            if not v2:
                d[p + 'gamma'] = ((c,), 'layer_scale')
            _conv(d, p + 'dwconv.', c, c, 7, groups=c)
            _ln(d, p + 'norm.', c)
            _lin(d, p + 'pwconv1.', c, 4 * c)
            if v2:
                d[p + 'grn.gamma'] = ((1, 1, 1, 4 * c), 'grn')
                d[p + 'grn.beta'] = ((1, 1, 1, 4 * c), 'grn')
            _lin(d, p + 'pwconv2.', 4 * c, c)
    for i in range(4):
        _ln(d, f'{prefix}norm{i}.', dims[i])
    return d


def mobilenetv2_inventory(prefix='backbone.'):
    d = OrderedDict()

    def conv_bn(p, cin, cout, k, groups=1):
        _conv(d, p + '0.', cin, cout, k, groups=groups, bias=False)
        _bn(d, p + '1.', cout)

    conv_bn(f'{prefix}features.0.', 3, 32, 3)
    cin, idx = 32, 1
    for t, c, n, s in MBV2_SETTING:
        for i in range(n):
            ch = int(round(cin * t))
            p = f'{prefix}features.{idx}.conv.'
            li = 0
            if t != 1:
                conv_bn(p + f'{li}.', cin, ch, 1)
                li += 1
            conv_bn(p + f'{li}.', ch, ch, 3, groups=ch)
            li += 1
            _conv(d, p + f'{li}.', ch, c, 1, bias=False)
            _bn(d, p + f'{li + 1}.', c)
            cin = c
            idx += 1
    return d


def segformer_head_inventory(dims, embed, nc, prefix='decode_head.'):
    d = OrderedDict()
    for i, c in enumerate(dims):
        _lin(d, f'{prefix}linear_c{i + 1}.proj.', c, embed)
    _conv(d, f'{prefix}linear_fuse.conv.', 4 * embed, embed, 1, bias=False)
    _bn(d, f'{prefix}linear_fuse.bn.', embed)
    _conv(d, f'{prefix}linear_pred.', embed, nc, 1)
    return d


def _convmodule(d, p, cin, cout, k):
    _conv(d, p + '0.', cin, cout, k, bias=False)
    _bn(d, p + '1.', cout)


def uper_head_inventory(dims, ch, nc, prefix='decode_head.'):
    d = OrderedDict()
    for k in range(len(PPM_SCALES)):
        _convmodule(d, f'{prefix}ppm.stages.{k}.1.', dims[-1], ch, 1)
    _convmodule(d, f'{prefix}ppm.bottleneck.', dims[-1] + ch * len(PPM_SCALES), ch, 3)
    for i, c in enumerate(dims[:-1]):
        _convmodule(d, f'{prefix}fpn_in.{i}.', c, ch, 1)
    for i in range(len(dims) - 1):
        _convmodule(d, f'{prefix}fpn_out.{i}.', ch, ch, 3)
    _convmodule(d, f'{prefix}bottleneck.', len(dims) * ch, ch, 3)
    _conv(d, f'{prefix}conv_seg.', ch, nc, 1)
    return d


def fpn_head_inventory(dims, ch, nc, prefix='decode_head.'):
    d = OrderedDict()
    for i, c in enumerate(dims[::-1]):
        _convmodule(d, f'{prefix}lateral_convs.{i}.', c, ch, 1)
    for i in range(len(dims)):
        _convmodule(d, f'{prefix}output_convs.{i}.', ch, ch, 3)
    _conv(d, f'{prefix}conv_seg.', ch, nc, 1)
    return d


def model_inventory(backbone: str, head: str, nc: int):
    """name -> (shape, kind) for SegmentationModel(backbone, seg_head=head, num_classes=nc)."""
    if backbone.startswith('MiT'):
        d = mit_inventory(backbone.split('-')[1])
    elif backbone == 'ConvNeXt':
        dep, dims, _ = CONVNEXT_SETTINGS['T']
        d = convnext_inventory(dep, dims, v2=False)
    elif backbone in CONVNEXTV2_SETTINGS:
        dep, dims, _ = CONVNEXTV2_SETTINGS[backbone]
        d = convnext_inventory(dep, dims, v2=True)
    elif backbone == 'MobileNetV2':
        d = mobilenetv2_inventory()
    else:
        raise KeyError(backbone)
    dims = backbone_channels(backbone)
    ch = head_width(backbone)
    if head == 'SegFormerHead':
        d.update(segformer_head_inventory(dims, ch, nc))
    elif head == 'UPerHead':
        d.update(uper_head_inventory(dims, ch, nc))
    elif head == 'FPNHead':
        d.update(fpn_head_inventory(dims, ch, nc))
    else:
        raise KeyError(head)
    return d


def make_state_dict(backbone: str, head: str, nc: int, seed: int = 1234, lively: bool = True):
    """Deterministic fp32 state_dict.

    lively=True draws every tensor from a distribution that keeps activations O(1) in every
    branch (weights ~ N(0, 1/fan_in), norm gains ~ 1 +- 0.1, non-trivial BN running stats), so a
    parity test is sensitive to every op.  lively=False mimics the reference initialisers
    (mit.py:27-40: linear std .02 / conv fan-out normal / norm (1, 0))."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, (shape, kind) in model_inventory(backbone, head, nc).items():
        if kind == 'w':
            fan_in = int(np.prod(shape[1:]))
            fan_out = int(shape[0] * np.prod(shape[2:])) if len(shape) == 4 else shape[0]
            if lively:
                v = rng.standard_normal(shape) / np.sqrt(fan_in)
            elif len(shape) == 4:
                v = rng.standard_normal(shape) * np.sqrt(2.0 / max(fan_out, 1))
            else:
                v = np.clip(rng.standard_normal(shape), -2, 2) * 0.02
        elif kind == 'b':
            v = rng.standard_normal(shape) * (0.1 if lively else 0.0)
        elif kind == 'norm_w':
            v = 1.0 + rng.standard_normal(shape) * (0.1 if lively else 0.0)
        elif kind == 'norm_b':
            v = rng.standard_normal(shape) * (0.1 if lively else 0.0)
        elif kind == 'run_mean':
            v = rng.standard_normal(shape) * (0.1 if lively else 0.0)
        elif kind == 'run_var':
            v = 1.0 + np.abs(rng.standard_normal(shape)) * (0.2 if lively else 0.0)
        elif kind == 'layer_scale':
            v = (0.5 + 0.5 * rng.random(shape)) if lively else np.full(shape, 1e-6)
        elif kind == 'grn':
            v = rng.standard_normal(shape) * (0.2 if lively else 0.0)
        elif kind == 'count':
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        else:
            raise KeyError(kind)
        sd[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32).reshape(shape))
    return sd


def synthetic_batch(batch: int, height: int, width: int, nc: int, seed: int = 1234,
                    ignore_index: int = 255, ignore_frac: float = 0.02):
    """images ~ N(0,1) fp32 NCHW; labels uniform in [0,nc) int64 with a
    deterministic ignore band (top 8 rows, or top 1/8 of small images) + ~2 % random ignore."""
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((batch, 3, height, width), dtype=np.float32)
    lbl = rng.integers(0, nc, (batch, height, width), dtype=np.int64)
    band = min(8, max(1, height // 8))
    lbl[:, :band] = ignore_index
    lbl[rng.random((batch, height, width)) < ignore_frac] = ignore_index
    return torch.from_numpy(img), torch.from_numpy(lbl)


def learnable_batch(batch: int, height: int, width: int, nc: int, seed: int = 1234, block: int = 16,
                    ignore_index: int = 255, noise: float = 0.15):
    """SURVEY.md section 8(d) / Appendix D `train_loop_overfit`: labels are a block pattern (block x block pixels per
    cell, classes drawn per cell) and the image colour is a fixed function of the label (one RGB code per class) plus
    noise, so a model can fit it and the logits become decisive.  Top band ignored (255) like synthetic_batch."""
    rng = np.random.default_rng(seed)
    gh, gw = (height + block - 1) // block, (width + block - 1) // block
    coarse = rng.integers(0, nc, (batch, gh, gw), dtype=np.int64)
    lbl = np.repeat(np.repeat(coarse, block, axis=1), block, axis=2)[:, :height, :width].copy()
    codes = np.random.default_rng(977).uniform(-1.5, 1.5, (nc, 3)).astype(np.float32)       # class -> RGB code
    img = codes[lbl].transpose(0, 3, 1, 2).copy()
    img += rng.standard_normal(img.shape, dtype=np.float32) * noise
    band = min(8, max(1, height // 16))
    lbl[:, :band] = ignore_index
    return torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)), torch.from_numpy(lbl)
