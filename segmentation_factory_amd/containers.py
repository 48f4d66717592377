"""Parameter containers with the reference's state_dict keys.

The reference's modules are nn.Linear / nn.Conv2d / nn.LayerNorm / nn.BatchNorm2d instances; their
parameter names and shapes are the checkpoint contract.  The classes below reuse
torch's constructors for *storage and initialisation only*: calling them raises, because all arithmetic
runs in the HIP kernels through ``segmentation_factory_amd.functional``.
"""
import math

import torch
from torch import nn


def _no_forward(self, *a, **k):
    raise RuntimeError(f'{type(self).__name__} is a parameter container: its arithmetic is executed by the HIP '
                       f'kernels in segmentation_factory_amd.functional, not by torch')


class LinearWeights(nn.Linear):
    forward = _no_forward


class ConvWeights(nn.Conv2d):
    forward = _no_forward


class LayerNormWeights(nn.LayerNorm):
    forward = _no_forward


class BatchNormWeights(nn.BatchNorm2d):
    forward = _no_forward


class ChannelsFirstLayerNormWeights(nn.Module):
    """ConvNeXt's channels-first LayerNorm (convnext.py:8-23): weight/bias of shape [C], eps 1e-6."""

    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    forward = _no_forward


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """Truncated normal initialiser with the reference's semantics (models/layers/initialize.py:17-70):
    inverse-CDF sampling on [a, b]."""
    def cdf(v):
        return (1. + math.erf(v / math.sqrt(2.))) / 2.
    with torch.no_grad():
        lo, hi = cdf((a - mean) / std), cdf((b - mean) / std)
        tensor.uniform_(2 * lo - 1, 2 * hi - 1).erfinv_().mul_(std * math.sqrt(2.)).add_(mean).clamp_(min=a, max=b)
    return tensor


def init_mit_style(m):
    """mit.py:27-40: Linear trunc-normal(.02) / zero bias, LayerNorm (1, 0), Conv fan-out normal."""
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            m.bias.data.zero_()
