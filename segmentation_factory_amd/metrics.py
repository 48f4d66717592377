"""mIoU / F1 / pixel-accuracy accumulator with the reference's interface (util/metrics.py:9-49,108-114).

``hist`` stays an fp32 [n, n] tensor accumulated per batch (quirk Q5: exact only below 2**24 per cell), rows =
ground truth.  Counting happens on device: ``update`` takes materialised NCHW logits like the reference;
``update_lowres`` takes the head output and fuses upsample + argmax + histogram in one kernel.
"""
import torch
import torch.distributed as dist

from . import hip
from .backbones import TokenMap, tokens_from_nchw


class Metrics:
    def __init__(self, num_classes: int, ignore_label: int, device) -> None:
        self.ignore_label = ignore_label
        self.num_classes = num_classes
        self.hist = torch.zeros(num_classes, num_classes).to(device)
        self.device = device
        self.bad_label_seen = None

    def _scratch(self, dev):
        """Per-batch int64 counts + the bad-label flag: static buffers (zeroed once, the accumulate kernel clears the counts again), so
        that an evaluation step can be captured as a hipGraph and launches no allocator-backed fills."""
        if getattr(self, '_counts', None) is None or self._counts.device != dev:
            n = self.num_classes
            self._counts = hip.zeros((n, n), torch.int64, dev)
            self.bad_label_seen = hip.zeros((1,), torch.int32, dev)
            if self.hist.device != dev:
                self.hist = self.hist.to(dev)
        return self._counts, self.bad_label_seen

    def update(self, pred: torch.Tensor, target: torch.Tensor) -> None:
        """pred: [B, n, H, W] logits; target: flat (or [B,H,W]) int64 labels (util/metrics.py:24-27)."""
        B, n, H, W = pred.shape
        tm = tokens_from_nchw(pred, pred.dtype if pred.dtype in (torch.float32, torch.bfloat16) else torch.float32)
        self.update_lowres(tm, target, (H, W))

    def update_lowres(self, lowres: TokenMap, target: torch.Tensor, size, confmat=None):
        """Fused bilinear-upsample + argmax + counting.  Optionally also feeds a utils.ConfusionMatrix (valid iff
        0 <= t < n) from the same pass, as engine.evaluate needs both (engine.py:90-91)."""
        H, W = size
        n = self.num_classes
        dev = lowres.data.device
        counts, flag = self._scratch(dev)
        if confmat is not None:
            confmat._ensure(dev)
            mat = confmat.mat
        else:
            if getattr(self, '_mat_unused', None) is None:
                self._mat_unused = hip.zeros((n, n), torch.int64, dev)
            mat = self._mat_unused
        if target.dtype != torch.int64 or not target.is_contiguous():
            target = target.contiguous().to(torch.int64)
        hip.argmax_confmat(lowres.data, lowres.B, n, lowres.H, lowres.W, H, W, target, self.ignore_label, mat, counts, flag)
        hip.hist_accum_(self.hist, counts)        # hist (fp32) += this batch's int64 counts, as the reference's `self.hist += bincount`

    def _finish(self, v):
        m = v[~v.isnan()].mean().item()
        v = v * 100
        return v.cpu().numpy().round(2).tolist(), round(m * 100, 2)

    def compute_iou(self):
        d = self.hist.diag()
        return self._finish(d / (self.hist.sum(0) + self.hist.sum(1) - d))

    def compute_f1(self):
        return self._finish(2 * self.hist.diag() / (self.hist.sum(0) + self.hist.sum(1)))

    def compute_pixel_acc(self):
        return self._finish(self.hist.diag() / self.hist.sum(1))

    def reduce_from_all_processes(self):
        if not dist.is_available() or not dist.is_initialized():
            return
        dist.barrier()
        dist.all_reduce(self.hist)
