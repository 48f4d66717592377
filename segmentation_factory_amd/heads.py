"""Decode heads behind the reference's head contract (models/build_models.py:9-14,53-54):
``Head(in_channels: list, channel: int, num_classes: int)``, ``forward(features) -> [B, nc, h, w]``.
"""
import torch
from torch import nn

from . import functional as Fh
from .backbones import TokenMap, tokens_from_nchw
from .containers import BatchNormWeights, ConvWeights, LinearWeights

__all__ = ['SegFormerHead']

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


def dropout2d_scale(training, B, C, device, override):
    """nn.Dropout2d(0.1): whole channels of a sample are zeroed, survivors scaled by 1/0.9."""
    if not training:
        return None
    if override is not None and 'dropout2d' in override:
        keep = override['dropout2d'].to(device=device, dtype=torch.float32)
    else:
        keep = (torch.rand(B, C, device=device) >= DROPOUT2D_P).to(torch.float32)
    return (keep / (1.0 - DROPOUT2D_P)).contiguous()


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
        """feats: 4 TokenMaps.  Returns a TokenMap of logits at stride 4 (leading dim padded to 8)."""
        if len(feats) != 4:
            raise ValueError('SegFormerHead expects 4 feature maps (the 5-feature adjust_channels path of '
                             'heads/segformer.py:52-54 builds fresh random weights per call and is not supported)')
        B, H1, W1 = feats[0].B, feats[0].H, feats[0].W
        E, nc = self.embed_dim, self.num_classes
        lins = [getattr(self, f'linear_c{i + 1}').proj for i in range(4)]
        geoms = [(f.B, f.H, f.W) for f in feats]
        if self.fold and all(f.data.shape[1] % 8 == 0 for f in feats) and E % 8 == 0:
            x = Fh.segformer_folded_fuse([f.data for f in feats], [l.weight for l in lins], [l.bias for l in lins],
                                         self.linear_fuse.conv.weight, geoms)
        else:                                                                  # the reference's literal op order
            cat = Fh.segformer_project_concat([f.data for f in feats], [l.weight for l in lins], [l.bias for l in lins], geoms)
            x = Fh.linear(cat, self.linear_fuse.conv.weight)                   # 1x1 conv 4E -> E, no bias
        bn = self.linear_fuse.bn
        drop = dropout2d_scale(self.training and self.dropout.p > 0, B, E, x.device, self.stochastic_override)
        x = Fh.batch_norm_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training, bn.momentum, bn.eps,
                              act=1, chan_scale=drop, rows_per_sample=H1 * W1)
        if self.training:
            bn.num_batches_tracked += 1
        logits = Fh.linear(x, self.linear_pred.weight, self.linear_pred.bias, pad_to=(nc + 7) // 8 * 8)
        return TokenMap(logits, B, H1, W1)

    def forward(self, features):
        tms = [f if isinstance(f, TokenMap) else tokens_from_nchw(f, self.compute_dtype) for f in features]
        return self.forward_tokens(tms).nchw()
