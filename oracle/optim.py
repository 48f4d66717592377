"""CPU restatement of the optimizer step the reference runs after backward -- TEST INFRASTRUCTURE ONLY.

    engine.py:52-53   loss_scaler(loss, optimizer, clip_grad=args.clip_grad, clip_mode=args.clip_mode, parameters=...)
    -> timm.utils.NativeScaler.__call__ -> dispatch_clip_grad(mode='agc') -> timm.utils.agc.adaptive_clip_grad
    -> torch.optim.AdamW.step   (train_gpu.py:243-247 builds it through timm.optim.create_optimizer)

PARITY UNPINNED for the AGC half: timm (pinned 0.9.2 in the reference's requirements) is not installed in this image and
is not vendored under /root/reference, so `adaptive_clip_grad_` could not be checked against the real implementation.  The
AdamW half IS pinned: tests/test_host_cpu.py checks `adamw_step_` against torch.optim.AdamW itself.  The arithmetic below restates
the published algorithm (Brock et al. 2021, "High-Performance Large-Scale Image Recognition Without Normalization", eq. 3:
unit-wise gradient clipping with the parameter norm floored at eps = 1e-3) and torch.optim.AdamW's documented update
(decoupled weight decay, bias-corrected moments, eps added to the corrected second-moment root).
"""
import torch


def unitwise_norm(x: torch.Tensor) -> torch.Tensor:
    """L2 norm per output unit: whole tensor for <= 1-D, over all but the first dimension otherwise (kept for broadcast)."""
    if x.ndim <= 1:
        return x.norm(2)
    return x.norm(2, dim=tuple(range(1, x.ndim)), keepdim=True)


def adaptive_clip_grad_(param: torch.Tensor, grad: torch.Tensor, clip_factor: float = 0.01, eps: float = 1e-3) -> torch.Tensor:
    """grad <- grad * min(1, clip_factor * max(|p|, eps) / max(|g|, 1e-6)) per unit; returns the clipped gradient."""
    p_norm, g_norm = unitwise_norm(param), unitwise_norm(grad)
    max_norm = p_norm.clamp(min=eps) * clip_factor
    return torch.where(g_norm < max_norm, grad, grad * (max_norm / g_norm.clamp(min=1e-6)))


def adamw_step_(param, grad, exp_avg, exp_avg_sq, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                weight_decay: float = 0.0):
    """One torch.optim.AdamW update, in place on (param, exp_avg, exp_avg_sq); `step` counts from 1."""
    b1, b2 = betas
    param.mul_(1 - lr * weight_decay)
    exp_avg.mul_(b1).add_(grad, alpha=1 - b1)
    exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1 - b2)
    denom = exp_avg_sq.sqrt() / (1 - b2 ** step) ** 0.5 + eps
    param.addcdiv_(exp_avg, denom, value=-lr / (1 - b1 ** step))
    return param
