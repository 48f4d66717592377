"""CE + Dice criterion and mIoU metrics, restated (TEST INFRASTRUCTURE ONLY).

  criterion_loops        engine.py:10-15 + util/losses.py:126-177 with the reference's own
                         cost structure (Python loop over batch x class, boolean-mask gathers);
                         this is what ``bench.py``'s cpu_baseline times.
  criterion_closed_form  the same value as one vectorised expression:
                         loss = CE + 1 - mean_c mean_b (2 I + eps) / (P + T + eps)
  confusion_counts / metrics_from_hist   util/metrics.py:24-49, util/utils.py:99-123
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS = 1e-6


def _dice_target(target, num_classes, ignore_index):
    # util/losses.py:126-138
    t = target.clone()
    if ignore_index >= 0:
        ign = target == ignore_index
        t[ign] = 0
        oh = F.one_hot(t, num_classes).float()
        oh[ign] = ignore_index
    else:
        oh = F.one_hot(t, num_classes).float()
    return oh.permute(0, 3, 1, 2)


def _dice_coeff_loops(x, target, ignore_index, epsilon=EPS):
    # util/losses.py:141-161: x, target are ONE class's [B,H,W] slices; the batch loop runs inside
    d = 0.
    batch_size = x.shape[0]
    for i in range(batch_size):
        x_i = x[i].reshape(-1)
        t_i = target[i].reshape(-1)
        if ignore_index >= 0:
            roi = torch.ne(t_i, ignore_index)
            x_i = x_i[roi]
            t_i = t_i[roi]
        inter = torch.dot(x_i, t_i)
        sets = torch.sum(x_i) + torch.sum(t_i)
        if sets == 0:
            sets = 2 * inter
        d += (2 * inter + epsilon) / (sets + epsilon)
    return d / batch_size


def criterion_loops(inputs, target, loss_weight=None, num_classes=2, dice=True, ignore_index=-100):
    """Same autograd graph shape as the reference: the channel is sliced ONCE per class in the outer loop
    (util/losses.py:164-170: ``x[:, channel, ...]``) and the batch is indexed inside (``x[i]``, :146-159), so the
    backward materialises one full-size zero tensor per class (not per (image, class) pair)."""
    loss = F.cross_entropy(inputs, target, ignore_index=ignore_index, weight=loss_weight)
    if dice is True:
        tgt = _dice_target(target, num_classes, ignore_index)
        prob = F.softmax(inputs, dim=1)                # losses.py:175
        total = 0.
        for channel in range(prob.shape[1]):           # losses.py:167-168
            total += _dice_coeff_loops(prob[:, channel, ...], tgt[:, channel, ...], ignore_index)
        loss += 1 - total / prob.shape[1]              # losses.py:170,177; engine.py:14
    return loss


def criterion_closed_form(inputs, target, loss_weight=None, num_classes=2, dice=True, ignore_index=-100):
    loss = F.cross_entropy(inputs, target, ignore_index=ignore_index, weight=loss_weight)
    if dice is True:
        B, C = inputs.shape[:2]
        prob = F.softmax(inputs, dim=1).reshape(B, C, -1)
        t = target.reshape(B, -1)
        valid = (t != ignore_index) if ignore_index >= 0 else torch.ones_like(t, dtype=torch.bool)
        vf = valid.to(prob.dtype)
        oh = F.one_hot(torch.where(valid, t, torch.zeros_like(t)), C).to(prob.dtype).permute(0, 2, 1) * vf[:, None]
        inter = (prob * oh).sum(-1)
        psum = (prob * vf[:, None]).sum(-1)
        tsum = oh.sum(-1)
        sets = psum + tsum
        sets = torch.where(sets == 0, 2 * inter, sets)
        loss = loss + (1 - ((2 * inter + EPS) / (sets + EPS)).mean())
    return loss


def confusion_counts(pred_logits, target_flat, num_classes, ignore_label):
    """Returns (mat_int64, hist_counts_int64): ConfusionMatrix.update (util/utils.py:99-109; valid
    iff 0 <= t < n) and the per-batch bincount of Metrics.update (util/metrics.py:24-27; valid iff
    t != ignore_label).  Rows = ground truth, columns = prediction."""
    n = num_classes
    pred = pred_logits.argmax(1).flatten().cpu().numpy().astype(np.int64)
    t = target_flat.flatten().cpu().numpy().astype(np.int64)
    k = (t >= 0) & (t < n)
    mat = np.bincount(n * t[k] + pred[k], minlength=n * n).reshape(n, n)
    keep = t != ignore_label
    hist = np.bincount(t[keep] * n + pred[keep], minlength=n * n).reshape(n, n)
    return mat, hist


def metrics_from_hist(hist_f32: torch.Tensor):
    """Metrics.compute_iou / compute_f1 / compute_pixel_acc (util/metrics.py:30-49) on an fp32 hist."""
    def fin(v):
        m = v[~v.isnan()].mean().item() * 100
        return (v * 100).cpu().numpy().round(2).tolist(), round(m, 2)
    d = hist_f32.diag()
    iou = fin(d / (hist_f32.sum(0) + hist_f32.sum(1) - d))
    f1 = fin(2 * d / (hist_f32.sum(0) + hist_f32.sum(1)))
    acc = fin(d / hist_f32.sum(1))
    return iou, f1, acc


def confmat_compute(mat_i64: torch.Tensor):
    """ConfusionMatrix.compute (util/utils.py:115-123): (acc_global, acc, iu), no NaN filter."""
    h = mat_i64.float()
    return torch.diag(h).sum() / h.sum(), torch.diag(h) / h.sum(1), torch.diag(h) / (h.sum(1) + h.sum(0) - torch.diag(h))
