"""criterion / train_one_epoch / evaluate with the reference's signatures (engine.py:10-104).

Differences that are deliberate: compute is bf16-with-fp32-accumulate inside
the kernels instead of fp16 autocast + GradScaler; when the model offers ``forward_lowres`` the loss and the
eval metrics consume the stride-4 head output directly (fused upsample), so the 157 MB/img full-resolution
logits tensor is never written.
"""
import math
import sys

import torch

from . import functional as Fh
from . import hip
from . import utils
from .backbones import TokenMap, tokens_from_nchw
from .metrics import Metrics


def _class_weight(loss_weight, device):
    if loss_weight is None:
        return None
    return torch.as_tensor(loss_weight, dtype=torch.float32, device=device).contiguous()


def criterion_lowres(lowres: TokenMap, target, size, loss_weight=None, num_classes: int = 2, dice: bool = True,
                     ignore_index: int = -100):
    """criterion(F.interpolate(head_out, size), target) without materialising the upsampled logits."""
    H, W = size
    nc = lowres.data.shape[1]
    assert nc == num_classes, (nc, num_classes)
    loss, parts, _ = Fh.upsample_ce_dice(lowres.data, target, (lowres.B, nc, lowres.H, lowres.W, H, W), ignore_index,
                                         _class_weight(loss_weight, target.device), dice is True)
    return loss


def criterion(inputs, target, loss_weight=None, num_classes: int = 2, dice: bool = True, ignore_index: int = -100):
    """engine.py:10-15: F.cross_entropy + (dice) util.losses.dice_loss on [B, C, H, W] logits."""
    if isinstance(inputs, TokenMap):
        return criterion_lowres(inputs, target, target.shape[-2:], loss_weight, num_classes, dice, ignore_index)
    B, C, H, W = inputs.shape
    dtype = inputs.dtype if inputs.dtype in (torch.float32, torch.bfloat16) else torch.float32
    t = inputs.permute(0, 2, 3, 1)
    if t.is_contiguous() and inputs.dtype == dtype:
        tm = TokenMap(t.reshape(B * H * W, C), B, H, W)              # zero-copy: already NHWC underneath
    else:
        tm = TokenMap(Fh.nchw_to_tokens(inputs.to(dtype)), B, H, W)
    return criterion_lowres(tm, target, (H, W), loss_weight, num_classes, dice, ignore_index)


class _DeferredLosses:
    """Device ring of the last steps' loss values (graph path of train_one_epoch): `push` is one 4-byte device copy behind the
    replayed step, `drain` is the one host synchronisation of a logging interval."""

    def __init__(self, device, capacity):
        self.ring = torch.zeros(capacity, 1, dtype=torch.float32, device=device)
        self.meta, self.n = [], 0

    def full(self):
        return self.n == self.ring.shape[0]

    def push(self, loss, lr, it):
        hip.cast2d(loss.detach().reshape(1, 1), self.ring[self.n:self.n + 1])
        self.meta.append((lr, it))
        self.n += 1

    def drain(self):
        vals = self.ring[:self.n, 0].cpu().tolist()          # the synchronisation point
        out = [(v, lr, it) for v, (lr, it) in zip(vals, self.meta)]
        self.meta, self.n = [], 0
        return out


def _flush_losses(pending, metric_logger, writer, print_freq, it0):
    for loss_value, lr, it in pending.drain():
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        metric_logger.update(loss=loss_value, lr=lr)
        if writer is not None and (it - it0) % print_freq == 0:
            writer.add_scalar('train_loss', loss_value, it)
            writer.add_scalar('train_lr', lr, it)


def _graph_step_wanted(args, model, core, optimizer, loss_scaler, device):
    """The step runs as a replayed hipGraph (graph.GraphedTrainStep) unless the caller opted out.  `args.hip_graph`: True = required
    (a capture failure raises), False = eager launches, absent / None = AUTO: what `python train_gpu.py ...` as the reference's
    README launches it gets (train_gpu.py:325) -- the graph whenever the pieces it drives are the product's own (a model with
    `forward_lowres`, the fused AGC/AdamW optimizer, the no-scaling NativeScaler) and the model is not already wrapped in
    DistributedDataParallel (whose hooks then carry the exchange); an earlier failed capture on this model keeps it eager."""
    from .optim import FusedAGCAdamW, NativeScaler
    pref = getattr(args, 'hip_graph', None)
    if pref is not None and not pref:
        return False
    ok = (hasattr(core, 'forward_lowres') and isinstance(optimizer, FusedAGCAdamW) and torch.device(device).type == 'cuda')
    if pref:
        return ok
    return (ok and isinstance(loss_scaler, NativeScaler) and not hasattr(model, 'module')
            and getattr(core, '_graph_disabled', None) is None)


def _fall_back_to_eager(core, model, optimizer, device, err):
    """AUTO mode only: the capture of the train step failed (a plugin module that synchronises or allocates on another stream, tied
    weights, out of memory during the warm-up passes ...).  Say why, undo what the attempt armed, and continue with per-kernel
    launches in THIS process -- never a re-exec.  Under several ranks the graph path would have carried the gradient exchange: a
    DistributedDataParallel wrapper (the reference's own, train_gpu.py:233-236) takes it over."""
    print(f'[segmentation_factory_amd] hipGraph capture of the train step failed ({type(err).__name__}: {str(err)[:300]}); '
          'continuing with eager launches (pass --no-hip-graph to skip the attempt)', flush=True)
    core._graph_disabled = f'{type(err).__name__}: {err}'
    core._graphed_step = None
    if hasattr(optimizer, 'disable_direct_grads'):
        optimizer.disable_direct_grads()
    for p in core.parameters():
        p.grad = None
    torch.cuda.synchronize()
    if utils.get_world_size() > 1 and not hasattr(model, 'module'):
        dev = torch.device(device)
        core._ddp_fallback = torch.nn.parallel.DistributedDataParallel(
            core, device_ids=[dev.index if dev.index is not None else torch.cuda.current_device()], find_unused_parameters=True)
        return core._ddp_fallback
    return model


def train_one_epoch(model, optimizer, dataloader, epoch, device, print_freq, clip_grad, clip_mode, loss_scaler,
                    writer=None, args=None):
    model.train()
    num_steps = len(dataloader)
    metric_logger = utils.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    loss_weight = torch.as_tensor([1.0, 2.0], device=device) if args.nb_classes == 2 else None   # engine.py:28-32
    core = model.module if hasattr(model, 'module') else model
    fused = hasattr(core, 'forward_lowres')
    use_graph = _graph_step_wanted(args, model, core, optimizer, loss_scaler, device)
    if getattr(core, '_ddp_fallback', None) is not None:
        model = core._ddp_fallback      # an earlier capture failed under several ranks: DistributedDataParallel carries the exchange

    pending = None
    for idx, (img, lbl) in enumerate(metric_logger.log_every(dataloader, print_freq, header)):
        img = img.to(device, non_blocking=True)
        lbl = lbl.to(device, non_blocking=True)
        if use_graph:
            # the whole step (zero_grad, forward, criterion, backward, gradient gather) replayed as one hipGraph, followed by
            # the RCCL all-reduce of the flat gradient buffer and the fused AGC/AdamW kernel (graph.py)
            from .graph import GraphedTrainStep
            key = (tuple(img.shape), tuple(lbl.shape), clip_grad, clip_mode)
            gs = getattr(core, '_graphed_step', None)
            if gs is None or gs.key != key:
                def loss_fn(m, x, y, _lw=loss_weight, _hw=tuple(img.shape[2:])):
                    return criterion_lowres(m.forward_lowres(x), y, _hw, _lw, num_classes=args.nb_classes, dice=args.dice,
                                            ignore_index=args.ignore_index)
                core._graphed_step = gs = None
                try:
                    gs = GraphedTrainStep(core, optimizer, loss_fn, (img, lbl), clip_grad=clip_grad, clip_mode=clip_mode,
                                          exchange=getattr(args, 'grad_exchange', None), payload=getattr(args, 'grad_payload', None))
                except Exception as e:      # noqa: BLE001 -- whatever stopped the capture, the eager launches below still train
                    if getattr(args, 'hip_graph', None):
                        raise               # asked for explicitly (--hip-graph): the failure is the caller's to see
                    model = _fall_back_to_eager(core, model, optimizer, device, e)
                    use_graph = False
                if gs is not None:
                    gs.key = key
                    core._graphed_step = gs
                    print(f'[segmentation_factory_amd] train step captured as one hipGraph (input {tuple(img.shape)}, '
                          f'{"bucketed gradient exchange over " + str(gs.world) + " ranks" if gs.exchanging else "single rank"})', flush=True)
        if use_graph:
            if getattr(dataloader, 'bind_output', None) is not None and getattr(dataloader, 'out', None) is None:
                dataloader.bind_output(gs.static_inputs)    # device-side input pipeline: later batches land in the step's buffers
            loss = gs.step(img, lbl)
            lr = optimizer.param_groups[0]["lr"]
            # ONE host synchronisation per logging interval instead of one per step (the reference reads
            # loss.item() and calls torch.cuda.synchronize() every step, engine.py:44,56): the step's loss is parked in a device
            # ring, and the meters / the non-finite check / the TensorBoard scalars see every value, in order, when the line is due
            if pending is None:
                pending = _DeferredLosses(loss.device, max(1, min(int(print_freq), 256)))
            pending.push(loss, lr, epoch * num_steps + idx)
            if idx % print_freq == 0 or idx == num_steps - 1 or pending.full():
                _flush_losses(pending, metric_logger, writer if getattr(args, 'local_rank', 0) == 0 else None, print_freq, epoch * num_steps)
            continue
        optimizer.zero_grad()
        if fused:
            # DDP hooks fire on backward through the wrapped module's parameters either way
            if hasattr(model, 'module'):
                data, (b_, h_, w_) = model(img, lowres=True)
                lo = TokenMap(data, b_, h_, w_)
            else:
                lo = core.forward_lowres(img)
            loss = criterion_lowres(lo, lbl, img.shape[2:], loss_weight, num_classes=args.nb_classes, dice=args.dice,
                                    ignore_index=args.ignore_index)
        else:
            loss = criterion(model(img), lbl, loss_weight, num_classes=args.nb_classes, dice=args.dice,
                             ignore_index=args.ignore_index)
        loss_value = loss.item()
        if not math.isfinite(loss_value):
            print("Loss is {}, stopping training".format(loss_value))
            sys.exit(1)
        is_second_order = hasattr(optimizer, 'is_second_order') and optimizer.is_second_order
        loss_scaler(loss, optimizer, clip_grad=clip_grad, clip_mode=clip_mode, parameters=model.parameters(),
                    create_graph=is_second_order)
        lr = optimizer.param_groups[0]["lr"]
        metric_logger.update(loss=loss_value, lr=lr)
        if writer is not None and idx % print_freq == 0 and getattr(args, 'local_rank', 0) == 0:
            it = epoch * num_steps + idx
            writer.add_scalar('train_loss', loss, it)
            writer.add_scalar('train_lr', lr, it)
    if pending is not None and pending.n:           # a loader that yielded fewer batches than len() promised
        _flush_losses(pending, metric_logger, writer if getattr(args, 'local_rank', 0) == 0 else None, print_freq, epoch * num_steps)
    metric_logger.synchronize_between_processes()
    return metric_logger.meters["loss"].global_avg, lr


EVAL_DTYPES = {'fp32': torch.float32, 'bf16': torch.bfloat16}


@torch.inference_mode()
def evaluate(args, model, dataloader, device, print_freq, writer=None):
    """engine.py:74-104.  The reference evaluates in fp32 with autocast deliberately off (engine.py:86-88), whatever precision it
    trained in -- so does this: the forward runs the exact-fp32 kernels unless `args.eval_dtype == 'bf16'` (train_gpu.py
    --eval-dtype bf16: the production storage type, ~5x the rate; tests/test_model_gpu.py measures what it flips)."""
    model.eval()
    metric = Metrics(args.nb_classes, args.ignore_label, device)
    confmat = utils.ConfusionMatrix(args.nb_classes)
    metric_logger = utils.MetricLogger(delimiter="  ")
    header = 'Test:'
    core = model.module if hasattr(model, 'module') else model
    fused = hasattr(core, 'forward_lowres')
    eval_dtype = EVAL_DTYPES[str(getattr(args, 'eval_dtype', None) or 'fp32')]
    trained_dtype = getattr(core, 'compute_dtype', None)
    switch = hasattr(core, 'set_compute_dtype') and trained_dtype is not None and trained_dtype != eval_dtype
    if switch:
        core.set_compute_dtype(eval_dtype)
    try:
        return _evaluate(args, model, core, fused, dataloader, device, print_freq, writer, metric, confmat, metric_logger, header)
    finally:
        if switch:
            core.set_compute_dtype(trained_dtype)


def _evaluate(args, model, core, fused, dataloader, device, print_freq, writer, metric, confmat, metric_logger, header):
    if utils.get_world_size() > 1:
        # collective C2: the reference evaluates the DDP-wrapped model, whose forward first hands every rank rank 0's BatchNorm
        # buffers (train_gpu.py:233-236, broadcast_buffers=True).  The forwards below call the core module directly (graph path: there
        # is no wrapper at all), so the broadcast is issued here, once, in front of the forwards that read the running statistics
        from .graph import broadcast_buffers_
        broadcast_buffers_(core)
    session = None
    graph_pref = getattr(args, 'hip_graph', None)            # None = auto (on), as in train_one_epoch
    if (graph_pref is None or bool(graph_pref)) and fused and torch.device(device).type == 'cuda':
        from .graph import GraphedEvalSession
        session = GraphedEvalSession(core)                   # the eval forward as a replayed hipGraph per RECURRING input shape
    for idx, (images, labels) in enumerate(metric_logger.log_every(dataloader, print_freq, header)):
        images = images.to(device, non_blocking=True)
        labels = labels.to(device, non_blocking=True)
        if fused:
            lo = session(images) if (session is not None and images.is_cuda) else core.forward_lowres(images)
            metric.update_lowres(lo, labels, images.shape[2:], confmat=confmat)     # one pass feeds both matrices
        else:
            outputs = model(images)
            confmat.update(labels.flatten(), outputs.argmax(1).flatten())
            metric.update(outputs, labels.flatten())
        if writer and idx % print_freq == 0:
            writer.add_scalar('valid_mf1', metric.compute_f1()[1])
            writer.add_scalar('valid_acc', metric.compute_pixel_acc()[1])
            writer.add_scalar('valid_mIOU', metric.compute_iou()[1])
    confmat.reduce_from_all_processes()
    metric.reduce_from_all_processes()
    return confmat, metric
