"""One training step as a replayed hipGraph, with the data-parallel gradient exchange overlapped with backward.

The reference's step (engine.py:36-56) issues ~700 small kernels from Python; on MI355X the kernels of
SegFormer-B0 finish faster than the host can enqueue them.  ``GraphedTrainStep`` captures
forward + fused CE/Dice + backward as a single HIP graph (torch.cuda.CUDAGraph == hipGraph on ROCm) and replays it per
step.  Parameter gradients are written by the backward kernels straight into the optimizer's flat fp32 gradient buffer
(functional.direct_grads): no ``.grad`` tensors, no gather pass, no zero-fill.  Outside the graph remain the data-parallel
exchange and the fused AGC + AdamW kernel, whose bias-correction scalars change every step.

Data parallelism (train_gpu.py:233-236 wraps the model in DistributedDataParallel, whose bucket hooks all-reduce gradients
while backward is still running): here the flat gradient buffer is laid out in parameter-registration order, i.e. the
REVERSE of the order in which backward completes gradients, and cut into buckets of ~``bucket_mb``.  Inside the captured
graph an EXTERNAL event (hipEventRecordExternal) is recorded the moment the last gradient of a bucket has been written; at
replay the communication stream waits on that event and launches the bucket's all-reduce (RCCL over xGMI) while the graph
keeps executing the rest of backward on the compute stream.  Same arithmetic as the reference (mean of per-rank gradients:
the loss gradient is pre-scaled by 1/world and the collective sums), same bucket idea, no per-parameter Python hooks.
BatchNorm stays per-rank, as in the reference (plain nn.BatchNorm2d, no SyncBN).
"""
import collections
import contextlib
import gc
import os

import torch
import torch.distributed as dist

from . import functional as Fh
from . import hip
from .optim import FusedAGCAdamW


def allreduce_mean_(flat: torch.Tensor, group=None):
    """In-place mean over ranks of one flat buffer (collective C1 of SURVEY.md section 2.3).  RCCL ('nccl') averages inside
    the collective; gloo (CPU tests) sums and divides."""
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


def broadcast_buffers_(module, src: int = 0, group=None):
    """Collective C2: DistributedDataParallel(broadcast_buffers=True, the default at train_gpu.py:233-236) hands every rank
    rank `src`'s buffers -- BatchNorm running_mean / running_var / num_batches_tracked -- before EVERY forward, training and
    `evaluate` alike (engine.py:74-104 runs the wrapped model).  Training forwards never read the running statistics, and rank 0's own
    sequence of momentum updates is not changed by receiving its own values back, so one broadcast in front of the forwards that DO read
    them (evaluate) leaves exactly the state the reference's per-forward broadcasts leave.  One flat broadcast per dtype."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    by_dtype = {}
    for b in module.buffers():
        by_dtype.setdefault(b.dtype, []).append(b)
    n = 0
    with torch.no_grad():
        for dtype in sorted(by_dtype, key=str):
            bufs = by_dtype[dtype]
            flat = torch.cat([b.detach().reshape(-1) for b in bufs])
            dist.broadcast(flat, src=src, group=group)
            off = 0
            for b in bufs:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            n += len(bufs)
    return n


@contextlib.contextmanager
def _capture(graph):
    """torch.cuda.graph(graph) with the cyclic garbage collector held off for the length of the capture.  On ROCm the destructor of
    a torch CUDAGraph synchronises the DEVICE, which is illegal while this thread captures: an older graph (a previous model's
    train step, an evicted eval graph) that the collector happens to free between two captured launches throws out of a
    destructor and the process aborts ("Fatal Python error: Aborted ... Garbage-collecting", seen once in four full test runs).
    Collect BEFORE the capture, at a point where a device synchronise is harmless, then keep the collector off until it ends."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        # thread_local: RCCL's watchdog thread polls events while we capture; in the default global mode that aborts the capture
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            yield
    finally:
        if was_enabled:
            gc.enable()


def _storage_signature(module):
    """Changes when any parameter / buffer moves (the optimizer re-homing parameters into its flat buffer, .to(), load into new
    tensors): captured graphs hold raw addresses."""
    h = 0
    for t in list(module.parameters()) + list(module.buffers()):
        h = (h * 1000003 + t.data_ptr()) & 0xFFFFFFFFFFFFFFFF
    return h


class GraphedEvalForward:
    """`forward_lowres` of an eval-mode model as ONE replayed hipGraph (the forward of engine.evaluate, engine.py:86-88).  At the
    reference's default --val_batch_size 1 (train_gpu.py:72) an eager forward is ~300 launches of a few microseconds each and
    the host cannot enqueue them as fast as the GPU retires them.  The graph is a pure function of the static input buffer; the
    fused upsample + argmax + confusion-matrix kernel consumes its (static) output outside the graph."""

    def __init__(self, core, images, warmup: int = 1):
        assert not core.training, 'GraphedEvalForward captures the eval-mode forward'
        self.x = images.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                core.forward_lowres(self.x)
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with _capture(self.graph):
            self.out = core.forward_lowres(self.x)

    def __call__(self, images):
        if images.data_ptr() != self.x.data_ptr():
            self.x.copy_(images, non_blocking=True)
        self.graph.replay()
        return self.out


class GraphedEvalSession:
    """Per-`evaluate` handle on the cached eval graphs of `core`.  The reference's validation transform keeps the aspect ratio
    (ExtResize(int), datasets/build_datasets.py:24-29) and --val_batch_size is 1 (train_gpu.py:72): ADE20K / VOC / COCO-Stuff hand
    `evaluate` MANY distinct (H, W), Cityscapes a single one.  A capture costs a clone, an eager warm-up forward, the capture and an
    instantiation -- several eager forwards' worth -- so:

      * a shape is captured on its SECOND sighting; the first runs eagerly (a shape seen once never pays for a graph),
      * at most `max_captures` captures per session, after which unseen shapes stay eager (`core.forward_lowres`),
      * the cache holds `max_graphs` graphs and evicts the LEAST RECENTLY USED,
      * sighting counts and graphs persist on the model across sessions (the next epoch's `evaluate` replays from image one).

    The key includes the compute dtype (`evaluate` runs fp32 by default, --eval-dtype bf16 on request).  The storage signature is
    checked ONCE per session -- walking a module tree costs milliseconds, more than a batch-1 forward -- so the parameters must not
    move while a session is open (they do not inside engine.evaluate)."""

    def __init__(self, core, max_graphs: int = 8, max_captures: int = 8):
        self.core, self.max_graphs, self.max_captures = core, max_graphs, max_captures
        cache = core.__dict__.setdefault('_graphed_eval', {})
        sig = _storage_signature(core)
        if cache.get('sig') != sig:
            cache.clear()
            cache['sig'] = sig
        self.graphs = cache.setdefault('graphs', collections.OrderedDict())      # key -> GraphedEvalForward, LRU order
        self.seen = cache.setdefault('seen', {})                                  # key -> sightings without a graph
        self.captures = self.replays = self.eager = 0

    def __call__(self, images):
        key = (tuple(images.shape), images.dtype, getattr(self.core, 'compute_dtype', None))
        g = self.graphs.get(key)
        if g is not None:
            self.graphs.move_to_end(key)
            self.replays += 1
            return g(images)
        n = self.seen.get(key, 0) + 1
        if n < 2 or self.captures >= self.max_captures:
            if len(self.seen) > 4096:
                self.seen.clear()
            self.seen[key] = n
            self.eager += 1
            return self.core.forward_lowres(images)
        while len(self.graphs) >= self.max_graphs:
            self.graphs.popitem(last=False)
        self.seen.pop(key, None)
        g = self.graphs[key] = GraphedEvalForward(self.core, images)
        self.captures += 1
        return g(images)


def plan_buckets(numels, bucket_elems):
    """Cut the flat buffer (parameters in registration order, sizes `numels`) into contiguous buckets of >= bucket_elems
    elements, walking from the END (the gradients that backward finishes first).  Returns [(lo, hi, first_param, last_param)]
    in the order the buckets complete (last bucket of the buffer first)."""
    buckets, hi_p = [], len(numels)
    offs = [0]
    for n in numels:
        offs.append(offs[-1] + n)
    acc = 0
    for i in range(len(numels) - 1, -1, -1):
        acc += numels[i]
        if acc >= bucket_elems or i == 0:
            buckets.append((offs[i], offs[hi_p], i, hi_p - 1))
            hi_p, acc = i, 0
    return buckets


def comm_ranges(buckets, total, align):
    """Element ranges the collectives run over, one per bucket, in completion order.  Bucket k = [L_k, L_(k-1)) of the flat
    buffer completes when its last gradient is written; its collective covers [up(L_k), up(L_(k-1))) with up() = round up to a
    multiple of `align` (the top end: `total` rounded up -- the flat buffers carry that much slack).  The ranges tile the padded
    buffer exactly once; the few elements [L_k, up(L_k)) of a bucket travel with the NEXT (later-completing) bucket's range, and a
    range only reaches into gradients that were complete earlier.  Every range is a multiple of `align` = world x 16 elements,
    so reduce-scatter / all-gather shards are whole and 64-byte aligned."""
    up = lambda v: -(-v // align) * align
    return [(up(lo), up(hi) if hi < total else up(total)) for lo, hi in buckets]


EXCHANGE_MODES = ('all_reduce', 'rs_ag')


def sum_over_ranks_(buf, mode='all_reduce', group=None):
    """Sum the flat tensor `buf` (a whole number of world-size elements) over the ranks, in place, asynchronously on the current
    stream; returns the work handles.  'all_reduce': one collective, RCCL picks the algorithm.  'rs_ag': in-place reduce-scatter
    (this rank's shard is reduced where it lies) followed by an in-place all-gather."""
    if mode == 'all_reduce':
        return [dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert buf.numel() % world == 0, (buf.numel(), world)
    n = buf.numel() // world
    shard = buf[rank * n:(rank + 1) * n]
    w1 = dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
    if dist.get_backend(group) != 'nccl':
        w1.wait()            # RCCL orders the two on its stream; gloo's worker threads do not
    w2 = dist.all_gather_into_tensor(buf, shard, group=group, async_op=True)
    return [w1, w2]


class GraphedTrainStep:
    def __init__(self, model, optimizer: FusedAGCAdamW, loss_fn, example_inputs, clip_grad=None, clip_mode='agc',
                 warmup: int = 2, process_group=None, bucket_mb: float = 25.0, overlap: bool = True,
                 exchange: str = None, payload: str = None, force_exchange: bool = None):
        """loss_fn(model, *inputs) -> scalar loss tensor.  ``example_inputs`` fix the shapes; their storage becomes
        the static input buffers (``step(*new_inputs)`` copies into them).

        exchange: 'all_reduce' (default; RCCL picks the algorithm) or 'rs_ag' = in-place reduce-scatter + all-gather per bucket
        (on a fully connected xGMI node both halves use all 7 links of every GPU directly).  payload: 'fp32'
        (default) or 'bf16' (the bucket is cast to a bf16 staging buffer, exchanged, and cast back: half the bytes per link;
        changes the arithmetic, see the loss-curve test).  force_exchange: run the exchange even in a 1-rank process group
        (exercises the whole event -> communication stream -> collective -> optimizer chain on one GPU).  Environment overrides:
        SEGFAC_EXCHANGE, SEGFAC_GRAD_PAYLOAD, SEGFAC_FORCE_EXCHANGE."""
        if not isinstance(optimizer, FusedAGCAdamW):
            raise TypeError('GraphedTrainStep drives the fused AGC/AdamW kernel over flat buffers: pass a FusedAGCAdamW '
                            '(create_optimizer builds one); other optimizers run on the eager path')
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.static_inputs = [t.clone() for t in example_inputs]
        have_pg = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if have_pg else 1
        self.group = process_group
        self.exchange_mode = exchange or os.environ.get('SEGFAC_EXCHANGE', 'all_reduce')
        self.payload = payload or os.environ.get('SEGFAC_GRAD_PAYLOAD', 'fp32')
        if self.exchange_mode not in EXCHANGE_MODES or self.payload not in ('fp32', 'bf16'):
            raise ValueError(f'exchange {self.exchange_mode!r} / payload {self.payload!r}: expected {EXCHANGE_MODES} / fp32 | bf16')
        if force_exchange is None:
            force_exchange = bool(os.environ.get('SEGFAC_FORCE_EXCHANGE'))
        self.exchanging = have_pg and (self.world > 1 or force_exchange)
        self.opt.set_clipping(clip_grad, clip_mode)      # 'agc' in the AdamW kernel; 'norm' / 'value' as kernels right before it
        # parameters are re-homed into the flat buffer BEFORE capture, laid out in registration order (bucket = suffix)
        self.opt.ensure_built(order=list(model.parameters()))
        broadcast_flat_(self.opt.flat_params, 0, self.group)     # DDP's initial parameter broadcast from rank 0
        # ---- buckets of the flat gradient buffer, in completion order -------------------------------------------------
        self.buckets, self.events, self._pending = [], [], {}
        self._capturing = False
        self._spin_us = int(os.environ.get('SEGFAC_TEST_SPIN_US', '0'))
        if self.exchanging:
            numels = list(self.opt._spans)          # aligned extents of the parameters in the flat buffers
            total = sum(numels)
            plan = plan_buckets(numels, int(bucket_mb * (1 << 20) / 4)) if overlap else [(0, total, 0, len(numels) - 1)]
            for k, (lo, hi, p0, p1) in enumerate(plan):
                self.buckets.append((lo, hi))
                self.events.append(hip.GraphEvent() if overlap else None)
                for i in range(p0, p1 + 1):
                    self._pending[self.opt._grad_views[i].data_ptr()] = k
            align = self.world * 16
            if align > self.opt.FLAT_SLACK:
                raise ValueError(f'world size {self.world}: the flat buffers carry {self.opt.FLAT_SLACK} elements of slack')
            self.ranges = comm_ranges(self.buckets, total, align)
            self._left = [0] * len(self.buckets)
            self.comm = torch.cuda.Stream()
            if self.payload == 'bf16':
                self._stage = torch.empty(max(hi - lo for lo, hi in self.ranges), dtype=torch.bfloat16,
                                          device=self.opt.flat_grads_padded.device)
        # bf16 shadow of the flat parameter buffer, refreshed by one cast at the start of every step (functional.shadow_scope)
        self._shadow, self._shadow_map = None, None
        if any(p.dtype == torch.float32 for p in self.opt._params) and not hip.policy('no_weight_shadow'):
            flat = self.opt.flat_params
            self._shadow = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
            self._shadow_map = {}
            for p, o in zip(self.opt._params, self.opt._offsets):
                self._shadow_map[p.data_ptr()] = self._shadow[o:o + p.numel()]
        self.opt.enable_direct_grads(self._on_grad_written if (self.exchanging and overlap) else None)
        self._seed = torch.full((), 1.0 / self.world, dtype=torch.float32, device=self.static_inputs[0].device)
        self._derived = None if hip.policy('no_derived_weights') else Fh.DerivedWeights()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        # the warm-up passes run real train-mode forwards: snapshot every buffer (BatchNorm running_mean / running_var /
        # num_batches_tracked) and put it back afterwards, so that the graphed path leaves the same eval-time statistics and
        # checkpoint contents as the eager path and the reference (one momentum update per optimisation step, engine.py:41)
        with torch.cuda.stream(s):
            saved_buffers = [(b, b.detach().clone()) for b in self.model.buffers()]
            for _ in range(warmup):             # allocator / lazy-init warm-up on a side stream (no optimizer step)
                self._forward_backward_eager()
            with torch.no_grad():
                for b, keep in saved_buffers:
                    b.copy_(keep)
            del saved_buffers
        torch.cuda.current_stream().wait_stream(s)
        # A gradient that still arrives as `.grad` (plugin backbone / head on plain autograd, the case gather_grads exists for) must
        # be captured as an ASSIGNMENT: with the warm-up passes' .grad tensors left in place autograd would record `grad += new`,
        # nothing inside the graph zeroes it, and every replay would add to the running sum
        for p in self.model.parameters():
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        self._capturing = True
        self._reset_pending()
        with _capture(self.graph):
            self.loss = self._forward_backward_eager()
            self.opt.gather_grads()          # only gradients that arrived as .grad (foreign plugin modules); normally nothing
        self._capturing = False
        self.opt.finish_backward()           # parameters the captured backward never wrote are skipped by the optimizer kernel
        if self.exchanging and overlap:
            missing = [k for k, n in enumerate(self._left) if n > 0]
            if missing:       # parameters without a gradient (frozen / unused): their buckets are complete when the graph ends
                self.events = [None if k in missing else e for k, e in enumerate(self.events)]

    # ---- pieces --------------------------------------------------------------------------------------------------------
    def _forward_backward_eager(self):
        if self._shadow is not None:
            hip.cast_into(self.opt.flat_params, self._shadow)
            scope = Fh.shadow_scope(self._shadow_map)
        else:
            scope = contextlib.nullcontext()
        self.opt.begin_backward()
        # (the re-laid-out weight copies of the step in one launch; the weight gradients queued and issued in grouped launches)
        with scope, Fh.derived_scope(self._derived), Fh.defer_weight_grads():      # the Linear layers' weight gradients are queued and issued in grouped launches
            loss = self.loss_fn(self.model, *self.static_inputs)
            # d(mean over ranks of the per-rank losses) / d(this rank's loss) = 1 / world: the collective then only SUMS
            loss.backward(gradient=self._seed)          # (always passed: autograd's implicit ones_like would be a fill launch per step)
        return loss

    def _reset_pending(self):
        if self.exchanging:
            counts = [0] * len(self.buckets)
            have = {v.data_ptr() for p, v in zip(self.opt._params, self.opt._grad_views)}
            for ptr, k in self._pending.items():
                if ptr in have:
                    counts[k] += 1
            self._left = counts

    def _on_grad_written(self, view):
        """functional.direct_grads callback (runs while backward is being captured): when the last gradient of a bucket
        has been enqueued, record the bucket's external event on the capture stream."""
        if not self._capturing:
            return
        k = self._pending.get(view.data_ptr())
        if k is None:
            return
        self._left[k] -= 1
        if self._left[k] == 1 and self._spin_us > 0:
            # TEST HOOK (tests/test_model_gpu.py, ordering test): hold the stream in front of this bucket's LAST gradient, so that a
            # communication stream that did not wait for THIS replay's event record would exchange the previous step's values
            hip.debug_spin(self._spin_us)
        if self._left[k] == 0 and self.events[k] is not None:
            self.events[k].record_external()

    def _collective(self, buf):
        return sum_over_ranks_(buf, self.exchange_mode, self.group)

    def _exchange(self):
        """Sum (the gradients carry the 1/world factor) bucket by bucket on the communication stream, each one as soon as its
        event inside the running graph has fired; the compute stream joins before the optimizer."""
        works = []
        grads = self.opt.flat_grads_padded
        with torch.cuda.stream(self.comm):
            for (lo, hi), ev in zip(self.ranges, self.events):
                if hi <= lo:
                    continue
                if ev is not None:
                    ev.wait()                                  # this replay's record node (the graph was launched above)
                else:
                    self.comm.wait_stream(self._main)          # bucket without an in-graph event: after the whole graph
                if self.payload == 'bf16':
                    st = self._stage[:hi - lo]
                    hip.cast_into(grads[lo:hi], st)            # our own cast kernels, on the communication stream
                    for w in self._collective(st):
                        w.wait()                               # stream-orders the cast back behind the collective (no host block)
                    hip.cast_into(st, grads[lo:hi])
                else:
                    works += self._collective(grads[lo:hi])
        for w in works:
            w.wait()                                           # the compute stream waits for the collective's result
        self._main.wait_stream(self.comm)

    def _feed(self, inputs):
        # zero-copy feed: a producer that writes the batch straight into `static_inputs` (transforms.DeviceBatchLoader.bind_output,
        # bench.py) passes those very tensors back and nothing is copied; anything else is copied into them
        for dst, src in zip(self.static_inputs, inputs):
            if src.data_ptr() != dst.data_ptr() or src.shape != dst.shape:
                dst.copy_(src, non_blocking=True)

    def step(self, *inputs):
        """One optimisation step; returns the (device) loss tensor of this step."""
        self._feed(inputs)
        self._main = torch.cuda.current_stream()
        self.graph.replay()
        if self.exchanging:
            self._exchange()
        self.opt.apply_flat()
        return self.loss

    def forward_backward(self, *inputs):
        """Replay without the exchange / optimizer (forward + loss + backward only)."""
        self._feed(inputs)
        self.graph.replay()
        return self.loss
