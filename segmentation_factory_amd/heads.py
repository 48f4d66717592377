"""Decode heads behind the reference's head contract (models/build_models.py:9-14,53-54):
``Head(in_channels: list, channel: int, num_classes: int)``, ``forward(features) -> [B, nc, h, w]``.
"""
import torch
from torch import nn

from . import functional as Fh
from .backbones import TokenMap, tokens_from_nchw
from .containers import BatchNormWeights, ConvWeights, LinearWeights

__all__ = ['SegFormerHead', 'UPerHead', 'FPNHead']

DROPOUT2D_P = 0.1


class MLP(nn.Module):
    """Linear embedding of one scale (heads/segformer.py:10-18)."""

    def __init__(self, dim, embed_dim):
        super().__init__()
        self.proj = LinearWeights(dim, embed_dim)


class ConvModule(nn.Module):
    """1x1 conv (no bias) + BatchNorm2d + ReLU (heads/segformer.py:21-29)."""

    def __init__(self, c1, c2):
        super().__init__()
        self.conv = ConvWeights(c1, c2, 1, bias=False)
        self.bn = BatchNormWeights(c2)


def dropout2d_scale(training, B, C, device, override, owner=None):
    """nn.Dropout2d(0.1): whole channels of a sample are zeroed, survivors scaled by 1/0.9."""
    if not training:
        return None
    if override is not None and 'dropout2d' in override:
        keep = override['dropout2d'].to(device=device, dtype=torch.float32)
        return (keep / (1.0 - DROPOUT2D_P)).contiguous()
    return Fh.stochastic_scales(owner, (1.0 - DROPOUT2D_P,), B * C, device).view(B, C)


class SegFormerHead(nn.Module):
    """All-MLP decoder (heads/segformer.py:32-58)."""

    def __init__(self, dims: list, embed_dim: int = 256, num_classes: int = 19):
        super().__init__()
        for i, dim in enumerate(dims):
            self.add_module(f"linear_c{i + 1}", MLP(dim, embed_dim))
        self.linear_fuse = ConvModule(embed_dim * 4, embed_dim)
        self.linear_pred = ConvWeights(embed_dim, num_classes, 1)
        self.dropout = nn.Dropout2d(DROPOUT2D_P)      # rate holder; the mask is applied inside the BN kernel
        self.embed_dim, self.num_classes = embed_dim, num_classes
        self.compute_dtype = torch.bfloat16
        self.stochastic_override = None               # tests: {'dropout2d': keep[B, embed_dim]}
        self.fold = True      # fold Linear -> resize -> concat -> 1x1 conv algebraically (functional.SegformerFoldedFuseFn)

    def forward_tokens(self, feats):
        """feats: 4 TokenMaps.  Returns a TokenMap of logits at stride 4 (class rows padded to a multiple of 32 columns: the
        classifier's backward then runs inside the BatchNorm backward, functional.BnActLinearFn)."""
        if len(feats) != 4:
            raise ValueError('SegFormerHead expects 4 feature maps (the 5-feature adjust_channels path of '
                             'heads/segformer.py:52-54 builds fresh random weights per call and is not supported)')
        B, H1, W1 = feats[0].B, feats[0].H, feats[0].W
        E, nc = self.embed_dim, self.num_classes
        lins = [getattr(self, f'linear_c{i + 1}').proj for i in range(4)]
        geoms = [(f.B, f.H, f.W) for f in feats]
        if self.fold and all(f.data.shape[1] % 8 == 0 for f in feats) and E % 8 == 0:
            x, pre = Fh.segformer_folded_fuse([f.data for f in feats], [l.weight for l in lins], [l.bias for l in lins],
                                              self.linear_fuse.conv.weight, geoms)
        else:                                                                  # the reference's literal op order
            cat = Fh.segformer_project_concat([f.data for f in feats], [l.weight for l in lins], [l.bias for l in lins], geoms)
            x = Fh.linear(cat, self.linear_fuse.conv.weight)                   # 1x1 conv 4E -> E, no bias
            pre = None
        bn = self.linear_fuse.bn
        drop = dropout2d_scale(self.training and self.dropout.p > 0, B, E, x.device, self.stochastic_override, self.dropout)
        logits = Fh.bn_act_linear(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, bn.momentum, bn.eps, 1,
                                  drop, H1 * W1, self.linear_pred.weight, self.linear_pred.bias, pad_to=(nc + 31) // 32 * 32,
                                  pre_sums=pre if self.training else None)
        if self.training:
            Fh.hip.add_i64_(bn.num_batches_tracked, 1)
        return TokenMap(logits, B, H1, W1)

    def forward(self, features):
        tms = [f if isinstance(f, TokenMap) else tokens_from_nchw(f, self.compute_dtype) for f in features]
        return self.forward_tokens(tms).nchw()



def _conv_module(c1, c2, k, s=1, p=0):
    """ConvModule = Conv2d(bias=False) + BatchNorm2d + ReLU (models/layers/conv_module.py:4-9) as an nn.Sequential so the
    state_dict keys stay `<name>.0.weight`, `<name>.1.{weight,bias,running_mean,running_var,num_batches_tracked}`."""
    return nn.Sequential(ConvWeights(c1, c2, k, s, p, bias=False), BatchNormWeights(c2))


def _bn_relu(x, bn, training, chan_scale=None, rows_per_sample=None):
    y = Fh.batch_norm_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.momentum, bn.eps, act=1,
                          chan_scale=chan_scale, rows_per_sample=rows_per_sample)
    if training:
        Fh.hip.add_i64_(bn.num_batches_tracked, 1)
    return y


class PPM(nn.Module):
    """Pyramid Pooling Module (models/modules/ppm.py:7-27)."""

    def __init__(self, c1, c2=128, scales=(1, 2, 3, 6)):
        super().__init__()
        self.scales = tuple(scales)
        self.stages = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d(scale), _conv_module(c1, c2, 1)) for scale in scales])
        self.bottleneck = _conv_module(c1 + c2 * len(scales), c2, 3, 1, 1)

    def tokens(self, x: TokenMap, training):
        B, H, W = x.B, x.H, x.W
        C1 = x.data.shape[1]
        outs, geoms = [], []
        xs = Fh.fork(x.data, len(self.scales) + 1)                         # the pools and the concat all read x
        for k, (S, stage) in enumerate(zip(self.scales, self.stages)):
            conv, bn = stage[1][0], stage[1][1]
            p = Fh.adaptive_avgpool(xs[k + 1], B, H, W, S)                 # [B*S*S, C1]
            p = _bn_relu(Fh.linear(p, conv.weight), bn, training)         # 1x1 conv (no bias) + BN + ReLU
            outs.append(p)
            geoms.append((S, S))
        feats = [xs[0]] + outs[::-1]                                       # ppm.py:25: [x] + outs[::-1]
        gs = [(H, W)] + geoms[::-1]
        cat = Fh.resize_concat(feats, gs, [False] + [True] * len(outs), (B, H, W))     # align_corners=True (ppm.py:23)
        conv, bn = self.bottleneck[0], self.bottleneck[1]
        return _bn_relu(Fh.conv3x3(cat, conv.weight, B, H, W, fp8=getattr(self, 'fp8', False)), bn, training)


class UPerHead(nn.Module):
    """Unified Perceptual Parsing head (models/heads/upernet.py:11-50): PPM on C5, top-down FPN with 1x1 laterals and 3x3
    output convs, all levels resized to stride 4 and concatenated, 3x3 bottleneck, Dropout2d(0.1), 1x1 classifier."""

    def __init__(self, in_channels, channel=128, num_classes: int = 19, scales=(1, 2, 3, 6)):
        super().__init__()
        self.ppm = PPM(in_channels[-1], channel, scales)
        self.fpn_in = nn.ModuleList()
        self.fpn_out = nn.ModuleList()
        for in_ch in in_channels[:-1]:
            self.fpn_in.append(_conv_module(in_ch, channel, 1))
            self.fpn_out.append(_conv_module(channel, channel, 3, 1, 1))
        self.bottleneck = _conv_module(len(in_channels) * channel, channel, 3, 1, 1)
        self.dropout = nn.Dropout2d(DROPOUT2D_P)
        self.conv_seg = ConvWeights(channel, num_classes, 1)
        self.embed_dim, self.num_classes = channel, num_classes
        self.compute_dtype = torch.bfloat16
        self.stochastic_override = None               # tests: {'dropout2d': keep[B, channel]}
        self.fp8 = False      # SegmentationModel.set_fp8: forward + data gradient of the 3x3 convs on fp8 operands (BASELINE cfg5)

    def forward_tokens(self, feats):
        tr = self.training
        B = feats[0].B
        ch, nc = self.embed_dim, self.num_classes
        self.ppm.fp8 = self.fp8
        fa, fb = Fh.fork(self.ppm.tokens(feats[-1], tr), 2)               # PPM output: a pyramid level AND the top-down input
        f = TokenMap(fb, B, feats[-1].H, feats[-1].W)
        fpn = [TokenMap(fa, B, feats[-1].H, feats[-1].W)]
        for i in reversed(range(len(feats) - 1)):
            lat_conv, lat_bn = self.fpn_in[i][0], self.fpn_in[i][1]
            fi = feats[i]
            lat = _bn_relu(Fh.linear(fi.data, lat_conv.weight), lat_bn, tr)
            fsum = Fh.upsample_add(lat, f.data, (B, fi.H, fi.W, f.H, f.W))                          # upernet.py:41
            if i > 0:
                fsum, fnext = Fh.fork(fsum, 2)                              # feeds its 3x3 output conv and the next finer level
                f = TokenMap(fnext, B, fi.H, fi.W)
            oc, obn = self.fpn_out[i][0], self.fpn_out[i][1]
            fpn.append(TokenMap(_bn_relu(Fh.conv3x3(fsum, oc.weight, B, fi.H, fi.W, fp8=self.fp8), obn, tr), B, fi.H, fi.W))
        fpn.reverse()
        H1, W1 = fpn[0].H, fpn[0].W
        cat = Fh.resize_concat([t.data for t in fpn], [(t.H, t.W) for t in fpn], [False] * len(fpn), (B, H1, W1))
        conv, bn = self.bottleneck[0], self.bottleneck[1]
        drop = dropout2d_scale(tr and self.dropout.p > 0, B, ch, cat.device, self.stochastic_override, self.dropout)
        x = _bn_relu(Fh.conv3x3(cat, conv.weight, B, H1, W1, fp8=self.fp8), bn, tr, chan_scale=drop, rows_per_sample=H1 * W1)
        logits = Fh.linear(x, self.conv_seg.weight, self.conv_seg.bias, pad_to=(nc + 7) // 8 * 8)
        return TokenMap(logits, B, H1, W1)

    def forward(self, features):
        tms = [f if isinstance(f, TokenMap) else tokens_from_nchw(f, self.compute_dtype) for f in features]
        return self.forward_tokens(tms).nchw()


class FPNHead(nn.Module):
    """Panoptic-FPN head (models/heads/fpn.py:9-38): 1x1 lateral ConvModules on the reversed features, nearest-neighbour
    top-down path, one 3x3 ConvModule per merge, Dropout2d(0.1), 1x1 classifier at stride 2 of the finest feature.

    Quirk Q3 reproduced: the reference EVALUATES `lateral_convs[i](features[i])` once more for the shape test (fpn.py:30) and
    again for the `size=` argument when the shapes differ (:31) before the value that is used (:34), so in training the
    lateral BatchNorms' running statistics advance 2 or 3 momentum steps per forward (`num_batches_tracked` likewise);
    `output_convs[0]` is constructed but never called (its parameters get no gradient)."""

    def __init__(self, in_channels, channel=128, num_classes=19):
        super().__init__()
        self.lateral_convs = nn.ModuleList([])
        self.output_convs = nn.ModuleList([])
        for ch in in_channels[::-1]:
            self.lateral_convs.append(_conv_module(ch, channel, 1))
            self.output_convs.append(_conv_module(channel, channel, 3, 1, 1))
        self.conv_seg = ConvWeights(channel, num_classes, 1)
        self.dropout = nn.Dropout2d(DROPOUT2D_P)
        self.embed_dim, self.num_classes = channel, num_classes
        self.compute_dtype = torch.bfloat16
        self.stochastic_override = None               # tests: {'dropout2d': keep[B, channel]}

    def _lateral(self, i, f: TokenMap, evaluations):
        conv, bn = self.lateral_convs[i][0], self.lateral_convs[i][1]
        z = Fh.linear(f.data, conv.weight)
        if self.training:
            with torch.no_grad():                      # the discarded evaluations: same batch statistics, extra momentum steps
                for _ in range(evaluations - 1):
                    Fh.hip.bn_stats(z.detach(), bn.running_mean, bn.running_var, bn.momentum, bn.eps)
                    Fh.hip.add_i64_(bn.num_batches_tracked, 1)
        return _bn_relu(z, bn, self.training)

    def forward_tokens(self, feats):
        feats = feats[::-1]
        B = feats[0].B
        ch, nc = self.embed_dim, self.num_classes
        out = self._lateral(0, feats[0], 1)
        H, W = feats[0].H, feats[0].W
        n = len(feats)
        x = None
        for i in range(1, n):
            f = feats[i]
            resized = (f.H, f.W) != (H, W)
            lat = self._lateral(i, f, 3 if resized else 2)
            if resized:                                                        # fpn.py:30-31: nearest to the lateral's size
                out = Fh.nearest_up(out, (B, H, W, f.H, f.W), base=lat)        # ... and fpn.py:34: out + lateral, fused
            else:
                out = Fh.add(out, lat)
            H, W = f.H, f.W
            out = Fh.nearest_up(out, (B, H, W, 2 * H, 2 * W))                  # fpn.py:35: scale_factor=2.0
            H, W = 2 * H, 2 * W
            conv, bn = self.output_convs[i][0], self.output_convs[i][1]
            y = Fh.conv3x3(out, conv.weight, B, H, W)
            if i == n - 1:                                                      # fpn.py:37: Dropout2d in front of the classifier
                drop = dropout2d_scale(self.training and self.dropout.p > 0, B, ch, y.device, self.stochastic_override, self.dropout)
                out = _bn_relu(y, bn, self.training, chan_scale=drop, rows_per_sample=H * W)
            else:
                out = _bn_relu(y, bn, self.training)
        logits = Fh.linear(out, self.conv_seg.weight, self.conv_seg.bias, pad_to=(nc + 7) // 8 * 8)
        return TokenMap(logits, B, H, W)

    def forward(self, features):
        tms = [f if isinstance(f, TokenMap) else tokens_from_nchw(f, self.compute_dtype) for f in features]
        return self.forward_tokens(tms).nchw()
