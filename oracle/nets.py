"""Functional fp32 restatement of the reference networks (TEST INFRASTRUCTURE ONLY).

Every function takes the reference-keyed ``state_dict`` (see ``oracle/weights.py``)
and plain NCHW fp32 tensors, and is differentiable through torch autograd, which
makes it the backward oracle as well.  Stochastic layers (DropPath, Dropout2d) take
their keep-masks as explicit inputs so product and oracle can be driven with the
same draws; ``masks=None`` means "rates forced to 0" (SURVEY.md Appendix A step 4).

Reference lines restated:
  MiT            models/backbones/mit.py:43-59 (Attention), :62-99 (DWConv/MLP),
                 :102-131 (PatchEmbed), :134-146 (Block), :192-218 (forward)
  SegFormerHead  models/heads/segformer.py:10-29, :42-58
  ConvNeXt       models/backbones/convnext.py:8-23, :26-51, :109-120
  ConvNeXtV2     models/backbones/convnextv2.py:68-80 (GRN), :83-113, :170-178
  MobileNetV2    models/backbones/mobilenetv2.py:5-37, :86-92
  UPerHead/PPM   models/heads/upernet.py:35-50, models/modules/ppm.py:21-27
  FPNHead        models/heads/fpn.py:26-38
  model forward  models/build_models.py:62-66
"""
import torch
import torch.nn.functional as F

from . import weights as W

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
DROPOUT2D_P = 0.1   # heads/segformer.py:40, upernet.py:31, fpn.py:24


class Ctx:
    """Mutable per-forward context: BN buffers (chained across repeated calls -- FPNHead quirk
    Q3), train flag, stochastic keep-masks."""

    def __init__(self, sd, training, masks=None):
        self.sd = sd
        self.training = training
        self.buffers = {k: v.clone() for k, v in sd.items()
                        if k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}
        self.masks = masks or {}
        self._dp = 0

    def next_drop_path(self):
        keep = self.masks.get('drop_path')
        i = self._dp
        self._dp += 1
        if keep is None:
            return None
        return keep[i]


def _drop_path(ctx, x, rate):
    """x / kp * floor(kp + U)  (models/layers/drop_path.py:18-25); keep = floor(kp+U) in {0,1}."""
    keep = ctx.next_drop_path()
    if (not ctx.training) or keep is None or rate == 0.0:
        return x
    kp = 1.0 - rate
    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
    return x / kp * keep.to(x.dtype).reshape(shape)


def _dropout2d(ctx, x):
    keep = ctx.masks.get('dropout2d')
    if (not ctx.training) or keep is None:
        return x
    return x * keep.to(x.dtype)[:, :, None, None] / (1.0 - DROPOUT2D_P)


def _bn(ctx, x, p):
    sd = ctx.sd
    rm, rv = ctx.buffers[p + 'running_mean'], ctx.buffers[p + 'running_var']
    y = F.batch_norm(x, rm, rv, sd[p + 'weight'], sd[p + 'bias'], ctx.training, BN_MOMENTUM, BN_EPS)
    if ctx.training:
        ctx.buffers[p + 'num_batches_tracked'] += 1
    return y


def _ln_tokens(sd, p, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[p + 'weight'], sd[p + 'bias'], eps)


def _ln_channels_first(sd, p, x, eps=1e-6):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[p + 'weight'][:, None, None] * x + sd[p + 'bias'][:, None, None]


def _linear(sd, p, x):
    return F.linear(x, sd[p + 'weight'], sd.get(p + 'bias'))


# --- MiT -------------------------------------------------------------------------
def mit_attention(sd, p, x, H, Wd, heads, sr):
    B, N, C = x.shape
    hd = C // heads
    q = _linear(sd, p + 'q.', x).reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    if sr > 1:
        xr = x.permute(0, 2, 1).reshape(B, C, H, Wd)
        xr = F.conv2d(xr, sd[p + 'sr.weight'], sd[p + 'sr.bias'], stride=sr)
        xr = xr.reshape(B, C, -1).permute(0, 2, 1)
        xr = _ln_tokens(sd, p + 'norm.', xr, 1e-5)
    else:
        xr = x
    kv = _linear(sd, p + 'kv.', xr).reshape(B, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return _linear(sd, p + 'proj.', out)


def mit_mlp(sd, p, x, H, Wd):
    B, N, _ = x.shape
    h = _linear(sd, p + 'fc1.', x)
    C4 = h.shape[-1]
    h = h.transpose(1, 2).reshape(B, C4, H, Wd)
    h = F.conv2d(h, sd[p + 'dwconv.dwconv.weight'], sd[p + 'dwconv.dwconv.bias'], padding=1, groups=C4)
    h = h.flatten(2).transpose(1, 2)
    return _linear(sd, p + 'fc2.', F.gelu(h))


def mit_forward(ctx, x, variant, prefix='backbone.'):
    sd = ctx.sd
    dims, depths = W.MIT_SETTINGS[variant]
    rates = W.drop_path_rates('MiT-' + variant)
    outs, bi = [], 0
    for s in range(4):
        k, st = W.MIT_PATCH[s]
        pe = f'{prefix}patch_embed{s + 1}.'
        x = F.conv2d(x, sd[pe + 'proj.weight'], sd[pe + 'proj.bias'], stride=st, padding=k // 2)
        B, C, H, Wd = x.shape
        t = _ln_tokens(sd, pe + 'norm.', x.flatten(2).transpose(1, 2), 1e-5)
        for j in range(depths[s]):
            p = f'{prefix}block{s + 1}.{j}.'
            rate = rates[bi]
            bi += 1
            # the reference uses nn.Identity when dpr == 0 (mit.py:139): no RNG draw for that block
            a = mit_attention(sd, p + 'attn.', _ln_tokens(sd, p + 'norm1.', t, 1e-5), H, Wd, W.MIT_HEADS[s], W.MIT_SR[s])
            t = t + (_drop_path(ctx, a, rate) if rate > 0 else a)
            m = mit_mlp(sd, p + 'mlp.', _ln_tokens(sd, p + 'norm2.', t, 1e-5), H, Wd)
            t = t + (_drop_path(ctx, m, rate) if rate > 0 else m)
        t = _ln_tokens(sd, f'{prefix}norm{s + 1}.', t, 1e-5)
        x = t.reshape(B, H, Wd, C).permute(0, 3, 1, 2)
        outs.append(x)
    return outs


# --- ConvNeXt / ConvNeXtV2 -------------------------------------------------------
def convnext_forward(ctx, x, depths, dims, rates, v2, prefix='backbone.'):
    sd = ctx.sd
    outs, bi = [], 0
    for i in range(4):
        p = f'{prefix}downsample_layers.{i}.'
        if i == 0:
            x = F.conv2d(x, sd[p + '0.weight'], sd[p + '0.bias'], stride=4)
            x = _ln_channels_first(sd, p + '1.', x)
        else:
            x = _ln_channels_first(sd, p + '0.', x)
            x = F.conv2d(x, sd[p + '1.weight'], sd[p + '1.bias'], stride=2)
        C = dims[i]
        for j in range(depths[i]):
            b = f'{prefix}stages.{i}.{j}.'
            rate = rates[bi]
            bi += 1
            h = F.conv2d(x, sd[b + 'dwconv.weight'], sd[b + 'dwconv.bias'], padding=3, groups=C)
            h = h.permute(0, 2, 3, 1)
            h = _ln_tokens(sd, b + 'norm.', h, 1e-6)
            h = F.gelu(_linear(sd, b + 'pwconv1.', h))
            if v2:
                gx = torch.norm(h, p=2, dim=(1, 2), keepdim=True)
                nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
                h = sd[b + 'grn.gamma'] * (h * nx) + sd[b + 'grn.beta'] + h
            h = _linear(sd, b + 'pwconv2.', h)
            if not v2:
                h = sd[b + 'gamma'] * h
            h = h.permute(0, 3, 1, 2)
            x = x + (_drop_path(ctx, h, rate) if rate > 0 else h)
        outs.append(_ln_channels_first(sd, f'{prefix}norm{i}.', x))
    return outs


# --- MobileNetV2 -----------------------------------------------------------------
def _conv_bn_act(ctx, x, p, stride=1, padding=0, groups=1, act='relu'):
    x = F.conv2d(x, ctx.sd[p + '0.weight'], None, stride=stride, padding=padding, groups=groups)
    x = _bn(ctx, x, p + '1.')
    if act == 'relu':
        return F.relu(x)
    if act == 'relu6':
        return F.relu6(x)
    return x


def mobilenetv2_forward(ctx, x, prefix='backbone.'):
    sd = ctx.sd
    outs = []
    x = _conv_bn_act(ctx, x, f'{prefix}features.0.', stride=2, padding=1, act='relu6')
    cin, idx = 32, 1
    for t, c, n, s in W.MBV2_SETTING:
        for i in range(n):
            stride = s if i == 0 else 1
            ch = int(round(cin * t))
            p = f'{prefix}features.{idx}.conv.'
            li, h = 0, x
            if t != 1:
                h = _conv_bn_act(ctx, h, p + f'{li}.', act='relu6')
                li += 1
            h = _conv_bn_act(ctx, h, p + f'{li}.', stride=stride, padding=1, groups=ch, act='relu6')
            li += 1
            h = F.conv2d(h, sd[p + f'{li}.weight'])
            h = _bn(ctx, h, p + f'{li + 1}.')
            x = x + h if (stride == 1 and cin == c) else h
            cin = c
            if idx in W.MBV2_TAPS:
                outs.append(x)
            idx += 1
    return outs


# --- heads -----------------------------------------------------------------------
def segformer_head(ctx, feats, prefix='decode_head.'):
    sd = ctx.sd
    B, _, H, Wd = feats[0].shape
    outs = []
    for i, f in enumerate(feats):
        t = _linear(sd, f'{prefix}linear_c{i + 1}.proj.', f.flatten(2).transpose(1, 2))
        t = t.permute(0, 2, 1).reshape(B, -1, *f.shape[-2:])
        if i > 0:
            t = F.interpolate(t, size=(H, Wd), mode='bilinear', align_corners=False)
        outs.append(t)
    cat = torch.cat(outs[::-1], dim=1)
    x = F.conv2d(cat, sd[prefix + 'linear_fuse.conv.weight'])
    x = F.relu(_bn(ctx, x, prefix + 'linear_fuse.bn.'))
    x = _dropout2d(ctx, x)
    return F.conv2d(x, sd[prefix + 'linear_pred.weight'], sd[prefix + 'linear_pred.bias'])


def ppm(ctx, x, prefix):
    outs = []
    for k, scale in enumerate(W.PPM_SCALES):
        y = F.adaptive_avg_pool2d(x, scale)
        y = _conv_bn_act(ctx, y, f'{prefix}stages.{k}.1.')
        outs.append(F.interpolate(y, size=x.shape[-2:], mode='bilinear', align_corners=True))
    cat = torch.cat([x] + outs[::-1], dim=1)
    return _conv_bn_act(ctx, cat, prefix + 'bottleneck.', padding=1)


def uper_head(ctx, feats, prefix='decode_head.'):
    f = ppm(ctx, feats[-1], prefix + 'ppm.')
    fpn = [f]
    for i in reversed(range(len(feats) - 1)):
        lat = _conv_bn_act(ctx, feats[i], f'{prefix}fpn_in.{i}.')
        f = lat + F.interpolate(f, size=lat.shape[-2:], mode='bilinear', align_corners=False)
        fpn.append(_conv_bn_act(ctx, f, f'{prefix}fpn_out.{i}.', padding=1))
    fpn.reverse()
    for i in range(1, len(feats)):
        fpn[i] = F.interpolate(fpn[i], size=fpn[0].shape[-2:], mode='bilinear', align_corners=False)
    out = _conv_bn_act(ctx, torch.cat(fpn, dim=1), prefix + 'bottleneck.', padding=1)
    out = _dropout2d(ctx, out)
    return F.conv2d(out, ctx.sd[prefix + 'conv_seg.weight'], ctx.sd[prefix + 'conv_seg.bias'])


def fpn_head(ctx, feats, prefix='decode_head.'):
    """Quirk Q3: lateral_convs[i] is *evaluated* up to three times per forward (fpn.py:30,31,34),
    which advances its BN running stats each time; only the last value is used."""
    feats = feats[::-1]
    out = _conv_bn_act(ctx, feats[0], f'{prefix}lateral_convs.0.')
    for i in range(1, len(feats)):
        lat = _conv_bn_act(ctx, feats[i], f'{prefix}lateral_convs.{i}.')          # shape probe (:30)
        if out.shape[2:] != lat.shape[2:]:
            lat = _conv_bn_act(ctx, feats[i], f'{prefix}lateral_convs.{i}.')      # size probe (:31)
            out = F.interpolate(out, size=lat.shape[2:], mode='nearest')
        lat = _conv_bn_act(ctx, feats[i], f'{prefix}lateral_convs.{i}.')          # value used (:34)
        out = out + lat
        out = F.interpolate(out, scale_factor=2.0, mode='nearest')
        out = _conv_bn_act(ctx, out, f'{prefix}output_convs.{i}.', padding=1)
    out = _dropout2d(ctx, out)
    return F.conv2d(out, ctx.sd[prefix + 'conv_seg.weight'], ctx.sd[prefix + 'conv_seg.bias'])


# --- whole model -------------------------------------------------------------------
def backbone_forward(ctx, x, backbone):
    if backbone.startswith('MiT'):
        return mit_forward(ctx, x, backbone.split('-')[1])
    if backbone == 'ConvNeXt':
        dep, dims, _ = W.CONVNEXT_SETTINGS['T']
        return convnext_forward(ctx, x, dep, dims, W.drop_path_rates(backbone), v2=False)
    if backbone in W.CONVNEXTV2_SETTINGS:
        dep, dims, _ = W.CONVNEXTV2_SETTINGS[backbone]
        return convnext_forward(ctx, x, dep, dims, W.drop_path_rates(backbone), v2=True)
    if backbone == 'MobileNetV2':
        return mobilenetv2_forward(ctx, x)
    raise KeyError(backbone)


def head_forward(ctx, feats, head):
    return {'SegFormerHead': segformer_head, 'UPerHead': uper_head, 'FPNHead': fpn_head}[head](ctx, feats)


def model_forward(sd, x, backbone, head, training=True, masks=None, lowres=False):
    """SegmentationModel.forward (build_models.py:62-66).  Returns (logits, ctx); ``ctx.buffers``
    holds the BN buffers after the call.  lowres=True returns the head output before the final
    bilinear resize."""
    ctx = Ctx(sd, training, masks)
    y = head_forward(ctx, backbone_forward(ctx, x, backbone), head)
    if not lowres:
        y = F.interpolate(y, size=x.shape[2:], mode='bilinear', align_corners=False)
    return y, ctx


def count_drop_path_draws(backbone):
    """Number of DropPath calls that consume a keep-mask per forward (2 per block with rate>0)."""
    return 2 * sum(1 for r in W.drop_path_rates(backbone) if r > 0)
