"""One training step as a replayed hipGraph.

The reference's step (engine.py:36-56) issues ~700 small kernels from Python; on MI355X the kernels of
SegFormer-B0 finish faster than the host can enqueue them.  ``GraphedTrainStep`` captures
zero_grad + forward + fused CE/Dice + backward + the gather of all parameter gradients into one flat fp32
buffer as a single HIP graph (torch.cuda.CUDAGraph == hipGraph on ROCm) and replays it per step; the two
things that stay outside the graph are the data-parallel exchange (ONE RCCL all-reduce of the flat gradient
buffer, averaged) and the fused AGC + AdamW kernel, whose bias-correction scalars change every step.

Data parallelism (train_gpu.py:233-236 wraps the model in DistributedDataParallel): here each rank replays
its own graph on its shard of the minibatch and the flat gradient buffer is all-reduced -- same arithmetic
(mean of per-rank gradients), one collective per step instead of per-bucket hooks.  BatchNorm stays
per-rank, as in the reference (plain nn.BatchNorm2d, no SyncBN).
"""
import torch
import torch.distributed as dist

from .optim import FusedAGCAdamW


def allreduce_mean_(flat: torch.Tensor, group=None):
    """In-place mean over ranks of one flat buffer: the whole data-parallel exchange of a step (collective C1 of SURVEY.md
    section 2.3).  RCCL ('nccl') averages inside the collective; gloo (CPU tests) sums and divides."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return flat
    if dist.get_backend(group) == 'nccl':
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(dist.get_world_size(group))
    return flat


def broadcast_flat_(flat: torch.Tensor, src: int = 0, group=None):
    """Rank `src`'s parameters to everyone (what DistributedDataParallel does at construction, train_gpu.py:233-236)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)
    return flat


class GraphedTrainStep:
    def __init__(self, model, optimizer: FusedAGCAdamW, loss_fn, example_inputs, clip_grad=None, clip_mode='agc',
                 warmup: int = 2, process_group=None):
        """loss_fn(model, *inputs) -> scalar loss tensor.  ``example_inputs`` fix the shapes; their storage becomes
        the static input buffers (``step(*new_inputs)`` copies into them)."""
        assert isinstance(optimizer, FusedAGCAdamW), 'the graphed step drives the fused AGC/AdamW kernel'
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.static_inputs = [t.clone() for t in example_inputs]
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.group = process_group
        if clip_grad is not None and clip_mode != 'agc':
            raise NotImplementedError("clip_mode='agc' is the fused mode (engine.py:52-53 default)")
        self.opt.agc_clip = float(clip_grad) if clip_grad is not None else 0.0
        self.opt.ensure_built()                 # parameters are re-homed into the flat buffer BEFORE capture
        broadcast_flat_(self.opt.flat_params, 0, self.group)     # DDP's initial parameter broadcast from rank 0
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        # the warm-up passes run real train-mode forwards: snapshot every buffer (BatchNorm running_mean / running_var /
        # num_batches_tracked) and put it back afterwards, so that the graphed path leaves the same eval-time statistics and
        # checkpoint contents as the eager path and the reference (one momentum update per optimisation step, engine.py:41)
        with torch.cuda.stream(s):
            saved_buffers = [(b, b.detach().clone()) for b in self.model.buffers()]
            for _ in range(warmup):             # allocator / lazy-init warm-up on a side stream (no optimizer step)
                self.opt.zero_grad(set_to_none=True)
                self.loss_fn(self.model, *self.static_inputs).backward()
            with torch.no_grad():
                for b, keep in saved_buffers:
                    b.copy_(keep)
            del saved_buffers
        torch.cuda.current_stream().wait_stream(s)
        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread polls events while we capture; in the default global mode that aborts the capture
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            self.loss = self.loss_fn(self.model, *self.static_inputs)
            self.loss.backward()
            self.opt.gather_grads()

    def step(self, *inputs):
        """One optimisation step; returns the (device) loss tensor of this step."""
        for dst, src in zip(self.static_inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        allreduce_mean_(self.opt.flat_grads, self.group)
        self.opt.apply_flat()
        return self.loss

    def forward_backward(self, *inputs):
        """Replay without the optimizer (forward + loss + backward only)."""
        for dst, src in zip(self.static_inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss
