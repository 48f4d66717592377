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
        if self.world > 1:                      # DDP's initial parameter broadcast from rank 0
            dist.broadcast(self.opt.flat_params, src=0, group=self.group)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):             # allocator / lazy-init warm-up on a side stream (no optimizer step)
                self.opt.zero_grad(set_to_none=True)
                self.loss_fn(self.model, *self.static_inputs).backward()
        torch.cuda.current_stream().wait_stream(s)
        self.opt.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self.loss_fn(self.model, *self.static_inputs)
            self.loss.backward()
            self.opt.gather_grads()

    def step(self, *inputs):
        """One optimisation step; returns the (device) loss tensor of this step."""
        for dst, src in zip(self.static_inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if self.world > 1:
            dist.all_reduce(self.opt.flat_grads, op=dist.ReduceOp.AVG, group=self.group)
        self.opt.apply_flat()
        return self.loss

    def forward_backward(self, *inputs):
        """Replay without the optimizer (forward + loss + backward only)."""
        for dst, src in zip(self.static_inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.loss
