"""Single-image inference: the GPU half of the reference's ``estimate_model.py`` (class SemSeg, :53-123).

    preprocess   (estimate_model.py:85-98)   short side -> img_size, ceil to a multiple of 32, /255, ImageNet normalise
    model_forward(:113-115)                  SegmentationModel under inference_mode
    postprocess  (:100-111)                  logits -> F.interpolate(orig size, bilinear, align_corners=True)
                                             -> softmax(dim=1).argmax(dim=1) -> palette

File reading, the dataset palettes / class names and ``draw_text`` stay with the reference's ``datasets`` package (PIL /
torchvision, outside the hot path: SURVEY section 8); this module takes and returns tensors.  The device work -- model, the two
bilinear resizes (head -> network input size with align_corners=False as build_models.py:65 does, then -> original size with
align_corners=True) and the class arg max -- runs in the HIP library: the low-resolution head output is the only logits tensor
that exists in bf16; both resizes run in fp32 on NHWC rows and the arg max reads the final rows directly (softmax is monotonic,
so it is skipped).
"""
import math

import torch

from . import hip
from .backbones import TokenMap

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


class SemSeg:
    def __init__(self, model, img_size=1024, palette=None, labels=None, device='cuda'):
        """model: a segmentation_factory_amd.SegmentationModel with its weights loaded (the reference loads 'model_state' /
        'state_dict' checkpoints into the same keys).  palette: optional uint8 tensor [num_classes, 3]."""
        self.device = device
        self.model = model.to(device).eval()
        self.size = [img_size, img_size]
        self.palette = palette.to(device) if palette is not None else None
        self.labels = labels

    def inference_size(self, H, W):
        """estimate_model.py:88-92: scale the short side to the target, then up to the next multiple of the model stride."""
        scale = self.size[0] / min(H, W)
        nH, nW = round(H * scale), round(W * scale)
        return int(math.ceil(nH / 32)) * 32, int(math.ceil(nW / 32)) * 32

    def preprocess(self, image: torch.Tensor) -> torch.Tensor:
        """image: uint8 CHW in [0, 255] -> normalised fp32 [1, 3, nH, nW] on the device: estimate_model.py:85-97 as ONE kernel of this
        library (segf_infer_preprocess: T.Resize of a uint8 tensor as torchvision 0.15.2 computes it -- bilinear, no antialias, rounded
        back to uint8 -- then / 255 and mean / std)."""
        H, W = image.shape[1:]
        nH, nW = self.inference_size(H, W)
        if image.dtype != torch.uint8:
            # the reference feeds torchvision.io.read_image's uint8 tensor (estimate_model.py:117-118), and T.Resize of a uint8 tensor
            # rounds back to uint8 where a float tensor would be interpolated in float: silently rounding a float image here would
            # match neither.  Only the byte path is pinned (byte-equal to torch's CPU resize), so only it is accepted
            raise TypeError(f'SemSeg.preprocess takes the uint8 CHW image torchvision.io.read_image returns, got {image.dtype}')
        img = image.to(self.device).contiguous()
        mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=self.device)
        std = torch.tensor(IMAGENET_STD, dtype=torch.float32, device=self.device)
        return hip.infer_preprocess(img, nH, nW, mean, std)

    @torch.inference_mode()
    def model_forward(self, img: torch.Tensor) -> TokenMap:
        """Head output at stride 4 as NHWC token rows (the full-resolution logits tensor is formed in postprocess)."""
        return self.model.forward_lowres(img)

    @torch.inference_mode()
    def postprocess(self, orig_hw, lo: TokenMap, in_hw, orig_img=None, overlay=False):
        """-> (seg_map int64 [H0, W0], colour image uint8/float [H0, W0, 3] or None)."""
        B, h, w = lo.B, lo.H, lo.W
        nc = self.model.num_classes if hasattr(self.model, 'num_classes') else lo.data.shape[1]
        H1, W1 = in_hw
        H0, W0 = orig_hw
        ld = (nc + 7) // 8 * 8
        low = torch.zeros((B * h * w, ld), dtype=torch.float32, device=lo.data.device)
        hip.cast2d(lo.data[:, :nc], low[:, :nc])                                   # bf16 head output -> fp32 rows
        mid = torch.empty((B * H1 * W1, ld), dtype=torch.float32, device=low.device)
        hip.bilinear_fwd(low, B, h, w, ld, H1, W1, mid, align_corners=False)        # build_models.py:65
        full = torch.empty((B * H0 * W0, ld), dtype=torch.float32, device=low.device)
        hip.bilinear_fwd(mid, B, H1, W1, ld, H0, W0, full, align_corners=True)      # estimate_model.py:102
        seg = hip.argmax_rows(full, nc).view(B, H0, W0)                             # :104
        img = None
        if self.palette is not None:
            img = self.palette[seg].squeeze(0)
            if overlay and orig_img is not None:
                img = orig_img.to(img.device).permute(1, 2, 0) * 0.4 + img * 0.6
        return seg.squeeze(0), img

    def predict(self, image: torch.Tensor, overlay: bool = True):
        """image: uint8 CHW tensor (what torchvision.io.read_image returns).  -> (seg_map, colour image or None)."""
        x = self.preprocess(image)
        lo = self.model_forward(x)
        return self.postprocess(tuple(image.shape[1:]), lo, tuple(x.shape[2:]), image, overlay)
