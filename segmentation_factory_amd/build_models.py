"""SegmentationModel: the reference's model factory / plugin boundary (models/build_models.py:9-66,
models/base_model.py:4-16), with the compute underneath replaced by the HIP path.

Drop-in surface kept: constructor signature, ``.backbone`` / ``.decode_head`` attribute names (state_dict
prefixes), ``head_dict``, ``__str__``, and the head-width rule (quirk Q1: 128 channels only when the backbone
*name* contains 'tiny' or 'small', else 768 -- including every MiT).
"""
import os

import torch
from torch import nn

from . import functional as Fh
from . import backbones as _backbones
from . import heads as _heads
from .backbones import TokenMap, tokens_from_nchw

# name -> constructor; the reference resolves names with eval() over models.backbones' namespace (:25-29)
backbone_registry = {'MiT': _backbones.MiT, 'ConvNeXt': _backbones.ConvNeXt, 'ConvNeXtV2': _backbones.ConvNeXtV2,
                     'MobileNetV2': _backbones.MobileNetV2}
backbone_registry.update({n: getattr(_backbones, n) for n in (
    'convnextv2_atto', 'convnextv2_femto', 'convnext_pico', 'convnextv2_nano', 'convnextv2_tiny', 'convnextv2_base',
    'convnextv2_large', 'convnextv2_huge')})
head_dict = {'SegFormerHead': _heads.SegFormerHead, 'UPerHead': _heads.UPerHead, 'FPNHead': _heads.FPNHead}


def register_backbone(name, ctor):
    backbone_registry[name] = ctor


def register_head(name, ctor):
    head_dict[name] = ctor


class BaseSegModel(nn.Module):
    def __init__(self, backbone: str = 'MiT-B0', num_classes: int = 19, seg_head: str = 'UPerHead', **kwargs):
        super().__init__()
        self.backbone_name = backbone
        self.num_classes = num_classes
        self.head_name = seg_head

    def __str__(self):
        if 'MiT' in self.backbone_name:
            return f'SegFormer-{self.backbone_name}'
        return f'{self.backbone_name}_{self.head_name}'


class SegmentationModel(BaseSegModel):
    def __init__(self, backbone: str = 'MiT-B0', pretrained_backbone='', num_classes: int = 19,
                 seg_head: str = 'UPerHead', aux_for_deeplab: bool = False, compute_dtype=torch.bfloat16, **kwargs):
        super().__init__(backbone=backbone, num_classes=num_classes, seg_head=seg_head, **kwargs)
        self.aux_for_deeplab = aux_for_deeplab
        if 'MiT' in backbone:
            family, variant = backbone.split('-')
            self.backbone = backbone_registry[family](variant)
            name_for_width = family          # quirk Q1: the name has been rebound to 'MiT' before the width test
        else:
            if backbone not in backbone_registry:
                raise KeyError(f'backbone {backbone!r} is not available in the MI355X path; have {sorted(backbone_registry)}')
            self.backbone = backbone_registry[backbone]()
            name_for_width = backbone
        if seg_head not in head_dict:
            raise KeyError(f'seg_head {seg_head!r} is not available in the MI355X path; have {sorted(head_dict)}')
        width = 128 if ('tiny' in name_for_width or 'small' in name_for_width) else 768
        self.decode_head = head_dict[seg_head](self.backbone.channels, width, num_classes)
        self.set_compute_dtype(compute_dtype)
        if pretrained_backbone:
            if os.path.exists(pretrained_backbone):
                self.backbone.load_state_dict(torch.load(pretrained_backbone, map_location='cpu'), strict=False)
            else:
                print('The pretrained weights path of backbone is wrong! File does not exists!!')

    def set_compute_dtype(self, dtype):
        """torch.bfloat16 (speed; fp32 statistics/accumulation) or torch.float32 (exact-parity mode)."""
        assert dtype in (torch.bfloat16, torch.float32)
        self.compute_dtype = dtype
        for m in (self.backbone, self.decode_head):
            if hasattr(m, 'compute_dtype'):
                m.compute_dtype = dtype
        return self

    def set_fp8(self, enabled: bool = True):
        """BASELINE cfg5 ("fp8 MFMA weights"): all three products -- forward, data gradient, weight gradient -- of (a) the ConvNeXt /
        ConvNeXtV2 block MLPs (convnextv2.py:90-95) where the shapes fill 256 x 256 tiles and (b) UPerHead's / PPM's 3x3 convolutions
        (heads/upernet.py:26-31, modules/ppm.py:19) on OCP fp8 operands: activations and weights e4m3, gradients e5m2, one dynamic scale
        per activation / gradient tensor and one per weight row, fp32 accumulate on the block-scaled fp8 matrix instruction (csrc/gemm8.hip,
        csrc/fp8.hip); everything else stays bf16.  Not a reference feature: tolerances stated in the tests.  Enabling it times the two
        schedules of the fp8 GEMM on this device once per process (hip.autotune_gemm8_fp8)."""
        if enabled and torch.cuda.is_available():
            from . import hip
            hip.autotune_gemm8_fp8()
        n = 0
        for m in self.backbone.modules():
            if hasattr(m, 'fp8') and hasattr(m, 'pwconv1'):
                m.fp8 = bool(enabled)
                n += 1
        if hasattr(self.decode_head, 'fp8'):
            self.decode_head.fp8 = bool(enabled)
            n += 1
        if enabled and n == 0:
            raise ValueError(f'set_fp8: model {self.backbone_name!r} + {self.head_name!r} has no fp8-capable layers '
                             '(ConvNeXt / ConvNeXtV2 blocks, UPerHead 3x3 convolutions)')
        return self

    def _features(self, x):
        if hasattr(self.backbone, 'forward_tokens'):
            return self.backbone.forward_tokens(x)
        return [tokens_from_nchw(f, self.compute_dtype) for f in self.backbone(x)]       # foreign plugin backbone

    def forward_lowres(self, x):
        """Head output before the final resize, as a TokenMap [B*h*w, nc] -- feed it to
        engine.criterion_lowres / Metrics.update_lowres so the full-resolution tensor is never materialised."""
        feats = self._features(x)
        if hasattr(self.decode_head, 'forward_tokens'):
            return self.decode_head.forward_tokens(feats)
        y = self.decode_head([f.nchw() for f in feats])                                 # foreign plugin head
        return tokens_from_nchw(y, self.compute_dtype)

    def forward(self, x, lowres: bool = False):
        """fp32 NCHW logits at input resolution, as the reference returns them (build_models.py:62-66).
        lowres=True (used through DistributedDataParallel, whose forward must return tensors) returns
        ``(tokens [B*h*w, nc], (B, h, w))`` = the pieces of ``forward_lowres``'s TokenMap."""
        lo = self.forward_lowres(x)
        if lowres:
            return lo.data, (lo.B, lo.H, lo.W)
        nc = lo.data.shape[1]
        return Fh.upsample_to_nchw(lo.data, (lo.B, nc, lo.H, lo.W, x.shape[2], x.shape[3]))
