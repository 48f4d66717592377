"""Autograd operators over the HIP kernels (segmentation_factory_amd.hip).

Activations are token-major 2-D tensors ``[B*H*W, C]`` (NHWC flattened) in the compute dtype
(bf16 for speed, fp32 for the exact-parity mode); parameters and their gradients stay fp32.
Every forward/backward below is a hand-written formula over C-ABI kernel calls -- torch autograd
only sequences them.  Nothing here falls back to eager PyTorch math.
"""
import contextlib

import torch
from torch.autograd import Function

from . import hip


def _rowmajor(t):
    return t if t.stride(-1) == 1 else t.contiguous()


# bf16 shadow of the optimizer's flat parameter buffer (graph.GraphedTrainStep): inside `shadow_scope` a parameter that lives in
# the flat buffer is not cast per use -- the step casts the whole buffer with ONE launch at its start and `_w` hands out views
# (45 five-microsecond cast launches per SegFormer-B0 step otherwise).  Only active while the step function runs (warm-up,
# capture): outside, e.g. in evaluate() between epochs, the shadow would be one optimizer update behind.
_SHADOW = None


class shadow_scope:
    def __init__(self, table):
        self.table = table

    def __enter__(self):
        global _SHADOW
        self.prev, _SHADOW = _SHADOW, self.table
        return self

    def __exit__(self, *exc):
        global _SHADOW
        _SHADOW = self.prev
        return False


def _w(param, dtype):
    """fp32 master parameter -> compute dtype copy (bf16 cast kernel; identity in fp32 mode)."""
    p = param.detach()
    if _SHADOW is not None and dtype == torch.bfloat16 and p.is_contiguous():
        v = _SHADOW.get(p.data_ptr())
        if v is not None and v.numel() == p.numel():
            return v.view(p.shape)
    p = p if p.is_contiguous() else p.contiguous()
    return hip.cast(p, dtype) if dtype != torch.float32 else p


def _splitk(n_out, k_in, tokens):
    return hip.pick_splitk(n_out, k_in, tokens)


# ---- derived weight layouts ------------------------------------------------------------------------------------------------------
# Several layers consume a re-laid-out copy of a parameter (the [O][(ky,kx)][ci] matrix of a patch / spatial-reduction convolution,
# mit.py:105,47; the [49][C] table of a depthwise 7 x 7, convnext.py:29).  The copy depends on the weights alone, so inside a train
# step (graph.GraphedTrainStep) all of them are produced by ONE hip.prep_grouped launch at the start of the step instead of one launch
# in front of each layer: the registry learns the jobs during the first pass (where each still runs on its own) and replays them.
_DERIVED = None


class DerivedWeights:
    def __init__(self):
        self.entries = {}          # key -> (tensor handed to the layer, [prep jobs that (re)build it])
        self.fresh = False

    def refresh(self):
        jobs = [j for _, js in self.entries.values() for j in js]
        if jobs:
            hip.prep_grouped(jobs)
        self.fresh = True


@contextlib.contextmanager
def derived_scope(registry):
    """Inside: derived_weight() hands out the registry's copies, rebuilt here in one launch."""
    global _DERIVED
    prev = _DERIVED
    _DERIVED = registry
    try:
        if registry is not None:
            registry.refresh()
        yield
    finally:
        if registry is not None:
            registry.fresh = False
        _DERIVED = prev


def derived_weight(param, kind, make):
    """make() -> (tensor, jobs): allocate the derived copy and list the hip.prep_grouped jobs that fill it from `param`."""
    reg = _DERIVED
    key = (param.data_ptr(), kind)
    if reg is not None and reg.fresh:
        e = reg.entries.get(key)
        if e is not None:
            return e[0]
    t, jobs = make()
    hip.prep_grouped(jobs)
    if reg is not None:
        reg.entries[key] = (t, jobs)
    return t


# ---- direct gradient placement ---------------------------------------------------------------------------------------------
# When the fused optimizer has re-homed the parameters into one flat buffer and `enable_direct_grads()` was called (the
# graphed train step does), every parameter carries `_segf_grad` = its fp32 view in the flat GRADIENT buffer.  The backward
# formulas below then write (or copy with segf_cast2d) the parameter gradients straight into those views and hand autograd
# `None`: no `.grad` tensors, no gather pass, no ATen accumulate / copy kernels inside the captured step, and the optimizer's
# `_segf_grad_cb` learns the moment a gradient is final (bucketed all-reduce overlapped with the rest of backward, graph.py).
def _slot(p):
    return getattr(p, '_segf_grad', None) if isinstance(p, torch.Tensor) else None


def _deliver_job(slot, value):
    """The copy job (hip.prep_grouped) that puts `value` into its flat-gradient view, or None when it already lives there."""
    if value is None or value.data_ptr() == slot.data_ptr():
        return None
    v = value.detach()
    assert v.dtype == torch.float32 and v.numel() == slot.numel(), (v.dtype, v.shape, slot.shape)
    if slot.ndim <= 1:                      # vectors (possibly a strided column of a wider buffer): copy as an [n, 1] matrix
        v, d = v.reshape(-1).unsqueeze(1), slot.reshape(-1).unsqueeze(1)
    else:
        v, d = v.reshape(slot.shape[0], -1), slot.reshape(slot.shape[0], -1)
        if v.stride(-1) != 1:
            v = v.contiguous()
    return ('cast', v, d)


def _deliver(slot_info, value):
    slot, cb = slot_info
    job = _deliver_job(slot, value)
    if job is not None:
        hip.cast2d(job[1], job[2])
    if cb is not None:
        cb(slot)


def _deliver_many(pairs):
    """[(slot_info, value)]: the copies of one backward formula's parameter gradients in ONE launch, then the delivery callbacks."""
    jobs = [j for j in (_deliver_job(info[0], v) for info, v in pairs) if j is not None]
    if len(jobs) == 1:
        hip.cast2d(jobs[0][1], jobs[0][2])
    elif jobs:
        hip.prep_grouped(jobs)
    for info, _ in pairs:
        if info[1] is not None:
            info[1](info[0])


# ---- deferred, grouped weight gradients -----------------------------------------------------------------------------------------
# The weight / bias gradients of the nn.Linear layers do not feed anything inside the backward pass, so under `defer_weight_grads()`
# (the captured train step) their dy^T x products are QUEUED by the backward formulas and issued together: hip.gemm_dw_db_grouped runs
# up to 12 layers per launch pair (csrc/gemm.hip: gemm_bf16_dw_group_kernel + splitk_reduce_group_kernel) instead of two launches per
# layer.  At the reference's default batch of 4 (train_gpu.py:71) those ~100 launches of 5 - 16 us were a quarter of the step.  Results
# are bitwise those of the per-layer launches; the queue keeps dy and x alive until the flush, and the direct-placement delivery
# callbacks (bucket events of the data-parallel exchange) fire at the flush.
_DW_QUEUE = None
DW_QUEUE_MAX = 12


_FIN_QUEUE = None          # deferred finalizes of two-stage column reductions (LayerNorm dgamma / dbeta), same scope
FIN_QUEUE_MAX = 32


def _flush_finalizes():
    q = _FIN_QUEUE
    if not q:
        return
    hip.colreduce_finalize_grouped([e[0] for e in q])
    for _, infos in q:
        for info in infos:
            _deliver(info, info[0])
    q.clear()


# dx of a LayerNorm whose input came out of a residual branch `x + DropPath(f(.))` (mit.py:143-146): the branch's backward starts by scaling
# that very gradient per sample (drop_path.py:18-25).  The LayerNorm backward kernel writes the scaled copy along with dx (one store more
# instead of a launch that reads dx again); it waits here, keyed by dx's address, for LinearFn.backward to pick it up.
_SCALED_DY = {}


def _take_scaled(dy, rscale, rpg):
    hit = _SCALED_DY.pop(dy.data_ptr(), None)
    # hit[3]: dx's version when the entry was made.  The hand-over is only right while the tagged tensor has ONE consumer: a second
    # one makes autograd accumulate into dx in place (same address, version bumped) and the parked copy would be that of a partial sum
    if (hit is not None and hit[1] == rscale.data_ptr() and hit[2] == rpg and hit[0].shape == dy.shape and hit[0].dtype == dy.dtype
            and hit[3] == dy._version):
        return hit[0]
    return None


def _ln_bwd(ctx, x, dy, g, mean, rstd, dy2=None, dres=None, dy2_patch=None):
    """LayerNorm backward with the (dgamma, dbeta) finalize deferred into the scope's grouped launch when both parameters carry adjacent
    flat-gradient slots; returns (dx, dgamma, dbeta) with None for gradients that will be delivered at the flush."""
    gg, gb = gslot(ctx, 1), gslot(ctx, 2)
    dgb = (gg, gb) if (gg is not None and gb is not None and gb.data_ptr() == gg.data_ptr() + 4 * gg.numel()) else None
    dp = getattr(ctx, 'dp', None)
    rscale, rpg = dp if dp is not None else (None, 1)
    q = _FIN_QUEUE
    if q is not None and dgb is not None:
        slots = ctx._gslots
        dx, item = hip.layernorm_bwd(x, dy, g, mean, rstd, dgb_out=dgb, dy2=dy2, dres=dres, defer=True, rscale=rscale, rows_per_group=rpg,
                                     dy2_patch=dy2_patch)
        q.append((item, (slots[1], slots[2])))
        if len(q) >= FIN_QUEUE_MAX:
            _flush_finalizes()
        out = (dx, None, None)
    else:
        out = hip.layernorm_bwd(x, dy, g, mean, rstd, dgb_out=dgb, dy2=dy2, dres=dres, rscale=rscale, rows_per_group=rpg, dy2_patch=dy2_patch)
    dx = out[0]
    if getattr(dx, 'scaled', None) is not None:
        if len(_SCALED_DY) > 64:
            _SCALED_DY.clear()                   # (entries nobody came for: a consumer that did not need its input gradient)
        _SCALED_DY[dx.data_ptr()] = (dx.scaled, rscale.data_ptr(), rpg, dx._version)
        dx.scaled = None
    return out


def flush_weight_grads():
    _flush_finalizes()
    q = _DW_QUEUE
    if not q:
        return
    # (the members' split-K counts were chosen per layer; inside a group they may share one smaller count: hip / segfac.h shared_split)
    hip.gemm_dw_db_grouped([e[0] for e in q], shared_split=not hip.policy('dw_no_shared_split'))
    # the layout passes behind the products (the [O][k k][Cin] -> OIHW permutes of the patch convolutions' weight gradients): one launch
    hip.prep_grouped([post for _, _, post in q if post is not None])
    for _, infos, post in q:
        for info in infos:
            _deliver(info, info[0])                   # the slot IS the result: no copy, only the delivery callback
    q.clear()


@contextlib.contextmanager
def defer_weight_grads():
    global _DW_QUEUE, _FIN_QUEUE
    prev, prevf = _DW_QUEUE, _FIN_QUEUE
    _DW_QUEUE = [] if not hip.policy('no_deferred_dw') else None
    _FIN_QUEUE = [] if not hip.policy('no_deferred_finalize') else None
    try:
        yield
        flush_weight_grads()
    finally:
        _DW_QUEUE, _FIN_QUEUE = prev, prevf
        _SCALED_DY.clear()          # entries nobody collected must not outlive the backward they belong to (their addresses get reused)


def _queue_dw(ctx, dy, x, n_out, n_in, tokens):
    """Queue dW [n_out, n_in] = dy^T x and db = colsum(dy) of a Linear whose parameters both carry flat-gradient slots; False when the
    product has to be issued by the caller (no scope, no slots)."""
    q = _DW_QUEUE
    slots = getattr(ctx, '_gslots', None)
    if q is None or not slots or 1 not in slots or 2 not in slots:
        return False
    gw, gb = slots[1][0].view(n_out, n_in), slots[2][0]
    if gb.numel() != n_out or not gb.is_contiguous():
        return False
    q.append(((dy, x, n_out, n_in, tokens, _splitk(n_out, n_in, tokens), gw, gb), (slots[1], slots[2]), None))
    if len(q) >= DW_QUEUE_MAX:
        flush_weight_grads()
    return True


def _queue_dw_post(ctx, dy, x, n_out, n_in, tokens, post_of):
    """As _queue_dw for a layer whose weight gradient needs a layout pass behind the product: the product lands in a temporary
    [n_out, n_in] fp32 matrix and the hip.prep_grouped job post_of(temp, weight_slot) runs after the grouped launch."""
    q = _DW_QUEUE
    slots = getattr(ctx, '_gslots', None)
    if q is None or not slots or 1 not in slots or 2 not in slots:
        return False
    gb = slots[2][0]
    if gb.numel() != n_out or not gb.is_contiguous():
        return False
    tmp = torch.empty((n_out, n_in), dtype=torch.float32, device=dy.device)
    gw = slots[1][0]
    q.append(((dy, x, n_out, n_in, tokens, _splitk(n_out, n_in, tokens), tmp, gb), (slots[1], slots[2]), post_of(tmp, gw)))
    if len(q) >= DW_QUEUE_MAX:
        flush_weight_grads()
    return True


def direct_grads(*param_idx):
    """Class decorator for the autograd Functions: the gradients of the forward arguments at positions `param_idx` (parameters)
    go to the optimizer's flat gradient buffer when the parameter carries a `_segf_grad` view; a backward that already wrote
    into `ctx.gslot(i)` (e.g. as the `out=` of its GEMM) returns that view and no copy is made."""
    def deco(cls):
        fwd, bwd = cls.forward, cls.backward

        def forward(ctx, *args):
            slots = None
            for i in param_idx:
                if i < len(args):
                    sl = _slot(args[i])
                    if sl is not None:
                        if slots is None:
                            slots = {}
                        slots[i] = (sl, getattr(args[i], '_segf_grad_cb', None))
            ctx._gslots = slots
            return fwd(ctx, *args)

        def backward(ctx, *grads):
            out = bwd(ctx, *grads)
            slots = getattr(ctx, '_gslots', None)
            if not slots:
                return out
            out = list(out) if isinstance(out, tuple) else [out]
            pairs = [(info, out[i]) for i, info in slots.items() if out[i] is not None]
            _deliver_many(pairs)
            for i in slots:
                out[i] = None
            return tuple(out)
        cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
        return cls
    return deco


def gslot(ctx, i, shape=None):
    """The flat-buffer view for forward argument i (or None): pass it as `out=` to write the gradient in place."""
    slots = getattr(ctx, '_gslots', None)
    if not slots or i not in slots:
        return None
    v = slots[i][0]
    return v.view(shape) if shape is not None else v


def _padded_rows(weight, bias, N, Np, K, dtype, device):
    """(w [Np, K] in `dtype`, b [Np] fp32 or None): the layer's parameters with zero rows up to Np (a column-padded output, e.g. 150 classes
    -> 152).  Weights-only work: inside a train step it comes from the step's one derived-weights launch."""
    wsrc = weight.detach().reshape(N, -1)

    def make_w():
        t = torch.empty((Np, K), dtype=dtype, device=device)
        return t, [('cast', wsrc, t[:N]), ('zero', t[N:])]
    w = derived_weight(weight.detach(), ('padrows', dtype, Np), make_w)
    b = None
    if bias is not None:
        bsrc = bias.detach()

        def make_b():
            t = torch.empty((Np,), dtype=torch.float32, device=device)
            return t, [('cast', bsrc.unsqueeze(1), t[:N].unsqueeze(1)), ('zero', t[N:])]
        b = derived_weight(bsrc, ('padrows', Np), make_b)
    return w, b


@direct_grads(1, 2)
class LinearFn(Function):
    """y = [residual + rscale[b] *] (x W^T + bias)   (nn.Linear / 1x1 conv on tokens).
    reference: mit.py:45,52,58,98-99 (q/kv/proj/fc1/fc2), heads/segformer.py:13,24,39."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, rscale, rows_per_group, pad_to, fp8=False):
        x = _rowmajor(x)
        M, K = x.shape
        N = weight.shape[0]
        Np = pad_to if (pad_to and pad_to > N) else N
        if fp8 and Np == N and x.dtype == torch.bfloat16 and hip.linear_fp8_supported(0, M, N, K):
            # FP8 on the 256 x 256 tile kernel (BASELINE cfg5): the activation tensor quantised to e4m3 with ONE dynamic scale, the
            # weight rows with one each.  The quantised input is kept for the weight gradient (instead of the bf16 tensor, when that
            # product has an fp8 form too); the backward quantises the gradient once (e5m2) for its two products.
            w = _w(weight.reshape(N, -1), x.dtype)
            xq, sx = hip.quant_tensor_fp8(x)
            wq, sw = hip.quant_rows_fp8(weight.detach().reshape(N, -1))
            y = hip.linear_fp8(0, xq, sx, wq, sw, bias=bias.detach() if bias is not None else None, residual=residual, rscale=rscale,
                               rows_per_group=rows_per_group or 1)
            keep8 = hip.linear_fp8_supported(2, M, N, K)
            ctx.save_for_backward(None if keep8 else x, w, rscale, xq if keep8 else None, sx if keep8 else None)
            ctx.meta = (M, N, K, bias is not None, residual is not None, rows_per_group or 1, weight.shape, Np)
            ctx.fp8t = True
            return y
        if fp8 and Np == N and x.dtype == torch.bfloat16 and hip.gemm_fp8_supported(M, N, K):
            # FP8 forward (BASELINE cfg5): token rows and output-channel rows quantised to e4m3 with dynamic amax scales, product
            # on the block-scaled fp8 matrix instruction; the backward products below stay bf16 on the saved operands
            w = _w(weight.reshape(N, -1), x.dtype)
            xq, sx = hip.quant_rows_fp8(x)
            wq, sw = hip.quant_rows_fp8(weight.detach().reshape(N, -1))
            y = hip.gemm_fp8(xq, sx, wq, sw, bias=bias.detach() if bias is not None else None, residual=residual, rscale=rscale,
                             rows_per_group=rows_per_group or 1)
            ctx.save_for_backward(x, w, rscale)
            ctx.meta = (M, N, K, bias is not None, residual is not None, rows_per_group or 1, weight.shape, Np)
            return y
        if Np > N:
            # column-padded output (e.g. 150 classes -> 152): the layer is run as an Np-wide linear whose extra weight rows
            # and biases are zero, so the pad columns of y are exact zeros, every 16-byte chunk of a row is either fully
            # valid or fully padding, and the backward products take the padded gradient as it stands
            assert residual is None
            w, b = _padded_rows(weight, bias, N, Np, K, x.dtype, x.device)
            y = hip.gemm(0, x, w, M, Np, K, bias=b)[:, :N]
        else:
            w = _w(weight.reshape(N, -1), x.dtype)
            b = bias.detach() if bias is not None else None
            y = hip.gemm(0, x, w, M, N, K, bias=b, residual=residual, rscale=rscale, rows_per_group=rows_per_group or 1)
        ctx.save_for_backward(x, w, rscale)
        ctx.meta = (M, N, K, bias is not None, residual is not None, rows_per_group or 1, weight.shape, Np)
        return y

    @staticmethod
    def backward(ctx, dy):
        if getattr(ctx, 'fp8t', False):
            return LinearFn._backward_fp8(ctx, dy)
        x, w, rscale = ctx.saved_tensors
        M, N, K, has_bias, has_res, rpg, wshape, Np = ctx.meta
        dy = _rowmajor(dy)
        if Np > N and dy.stride(0) == Np:
            # the gradient arrives in a column-padded buffer; every kernel of this library that writes padded rows zero-fills
            # the pad (segfac.h), so the products run Np wide and the zero rows / columns are sliced away afterwards
            dyp = dy.as_strided((M, Np), (Np, 1))
            dx = hip.gemm(1, dyp, w, M, K, Np) if ctx.needs_input_grad[0] else None
            dw = db = None
            if ctx.needs_input_grad[1] and has_bias and ctx.needs_input_grad[2]:
                dw, db = hip.gemm_dw_db(dyp, x, Np, K, M, split_k=_splitk(Np, K, M))
                dw, db = dw[:N].view(wshape), db[:N]
            else:
                if ctx.needs_input_grad[1]:
                    dw = hip.gemm(2, dyp, x, Np, K, M, out_dtype=torch.float32, split_k=_splitk(Np, K, M))[:N].view(wshape)
                if has_bias and ctx.needs_input_grad[2]:
                    db = hip.colsum(dyp)[:N]
            return dx, dw, db, None, None, None, None, None
        wv = w[:N] if Np > N else w
        dys = dy
        if rscale is not None:
            dys = _take_scaled(dy, rscale, rpg)              # written by the LayerNorm backward that produced dy, when there was one
            if dys is None:
                dys = hip.scale_rows(dy, rscale, rpg)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = hip.gemm(1, dys, wv, M, K, N)
        gw, gb = gslot(ctx, 1, (N, K)), gslot(ctx, 2)       # flat-gradient views (direct placement) or None
        if ctx.needs_input_grad[1] and has_bias and ctx.needs_input_grad[2]:
            if not _queue_dw(ctx, dys, x, N, K, M):           # deferred + grouped under defer_weight_grads(); else right here
                dw, db = hip.gemm_dw_db(dys, x, N, K, M, split_k=_splitk(N, K, M), out=gw, db_out=gb)   # one pass over dy for both
                dw = dw.view(wshape)
        else:
            if ctx.needs_input_grad[1]:
                dw = hip.gemm(2, dys, x, N, K, M, out=gw, out_dtype=torch.float32, split_k=_splitk(N, K, M)).view(wshape)
            if has_bias and ctx.needs_input_grad[2]:
                db = hip.colsum(dys, out=gb)
        dres = dy if (has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None, None, None, None


def _linear_backward_fp8(ctx, dy):
    """Backward of the tensor-scaled fp8 Linear: dx = dy W and dW = dy^T x on fp8 operands where the shapes have the tile kernel
    (the gradient quantised once, e5m2, one scale), bf16 otherwise; the bias gradient is a column sum of the bf16 gradient."""
    x, w, rscale, xq, sx = ctx.saved_tensors
    M, N, K, has_bias, has_res, rpg, wshape, Np = ctx.meta
    dy = _rowmajor(dy)
    dys = hip.scale_rows(dy, rscale, rpg) if rscale is not None else dy
    dx8 = ctx.needs_input_grad[0] and hip.linear_fp8_supported(1, M, K, N)
    dw8 = ctx.needs_input_grad[1] and xq is not None
    gq = sg = None
    if dx8 or dw8:
        gq, sg = hip.quant_tensor_fp8(dys, e5m2=True)
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        if dx8:
            wt = hip.permute021(w.reshape(1, N, K), 1, N, K, w.dtype).view(K, N)       # W^T, rows = input features
            wtq, swt = hip.quant_rows_fp8(wt)
            dx = hip.linear_fp8(1, gq, sg, wtq, swt)
        else:
            dx = hip.gemm(1, dys, w, M, K, N)
    gw, gb = gslot(ctx, 1, (N, K)), gslot(ctx, 2)
    if ctx.needs_input_grad[1]:
        if dw8:
            dw = hip.linear_fp8_wgrad(gq, sg, xq, sx, out=gw).view(wshape)
        else:
            dw = hip.gemm(2, dys, x, N, K, M, out=gw, out_dtype=torch.float32, split_k=_splitk(N, K, M)).view(wshape)
    if has_bias and ctx.needs_input_grad[2]:
        db = hip.colsum(dys, out=gb)
    dres = dy if (has_res and ctx.needs_input_grad[3]) else None
    return dx, dw, db, dres, None, None, None, None


LinearFn._backward_fp8 = staticmethod(_linear_backward_fp8)


@direct_grads(1, 2)
class LinearForkFn(Function):
    """(y, x') = (x W^T + bias, x): a Linear whose input has a second consumer (MiT attention: q = Linear(h) while h also feeds
    the key / value path, mit.py:52-53).  The second consumer's gradient arrives as an argument of this backward and rides into
    the data-gradient product as its residual operand, dx = dx' + dy W, instead of a separate add launch per block."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _rowmajor(x)
        M, K = x.shape
        N = weight.shape[0]
        w = _w(weight.reshape(N, -1), x.dtype)
        y = hip.gemm(0, x, w, M, N, K, bias=bias.detach() if bias is not None else None)
        ctx.save_for_backward(x, w)
        ctx.meta = (M, N, K, bias is not None, weight.shape)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dx2):
        x, w = ctx.saved_tensors
        M, N, K, has_bias, wshape = ctx.meta
        dy = _rowmajor(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = hip.gemm(1, dy, w, M, K, N, residual=_rowmajor(dx2) if dx2 is not None else None)
        gw, gb = gslot(ctx, 1, (N, K)), gslot(ctx, 2)
        if ctx.needs_input_grad[1] and has_bias and ctx.needs_input_grad[2]:
            if not _queue_dw(ctx, dy, x, N, K, M):
                dw, db = hip.gemm_dw_db(dy, x, N, K, M, split_k=_splitk(N, K, M), out=gw, db_out=gb)
                dw = dw.view(wshape)
        else:
            if ctx.needs_input_grad[1]:
                dw = hip.gemm(2, dy, x, N, K, M, out=gw, out_dtype=torch.float32, split_k=_splitk(N, K, M)).view(wshape)
            if has_bias and ctx.needs_input_grad[2]:
                db = hip.colsum(dy, out=gb)
        return dx, dw, db


def linear_fork(x, weight, bias=None):
    """(Linear(x), alias of x for a second consumer) -- see LinearForkFn"""
    return LinearForkFn.apply(x, weight, bias)


def linear(x, weight, bias=None, residual=None, rscale=None, rows_per_group=None, pad_to=None, fp8=False):
    y = LinearFn.apply(x, weight, bias, residual, rscale, rows_per_group, pad_to, fp8)
    if rscale is not None and not fp8:
        y._segf_dp = (rscale, rows_per_group or 1)         # a LayerNorm that consumes y can prepare this layer's scaled gradient (_ln_bwd)
    return y


@direct_grads(1, 2)
class LayerNormFn(Function):
    """nn.LayerNorm over the channel dim of token rows (mit.py:107,136-140; convnext.py:8-23 in NHWC)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, dp_scale=None, dp_rpg=1):
        ctx.dp = (dp_scale, int(dp_rpg)) if dp_scale is not None else None
        x = x if x.is_contiguous() else x.contiguous()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y, mean, rstd = hip.layernorm_fwd(x, g, b, eps)
        ctx.save_for_backward(x, g, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, mean, rstd = ctx.saved_tensors
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx, dg, db = _ln_bwd(ctx, x, dy, g, mean, rstd)       # adjacent flat-gradient views are one [2][C] target, written in place
        return dx, dg, db, None, None, None


def _dp_of(x):
    """(rscale, rows_per_group) when x is the output of a `residual + DropPath-scaled branch` product (functional.linear tags it)"""
    dp = getattr(x, '_segf_dp', None)
    return dp if dp is not None and not hip.policy('no_scaled_ln_bwd') else (None, 1)


def layer_norm(x, gamma, beta, eps):
    return LayerNormFn.apply(x, gamma, beta, eps, *_dp_of(x))


@direct_grads(1, 2)
class LayerNormResFn(Function):
    """The pre-norm residual pattern `x + f(norm(x))` (mit.py:143-146, convnext.py:36-50): returns (x, norm(x)) -- x passes
    through for the residual add -- so that the backward sees BOTH gradients of x at once and the LayerNorm backward kernel adds
    the residual-path gradient while it stores dx (no separate gradient-accumulation kernel)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, dp_scale=None, dp_rpg=1):
        ctx.dp = (dp_scale, int(dp_rpg)) if dp_scale is not None else None
        x = x if x.is_contiguous() else x.contiguous()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y, mean, rstd = hip.layernorm_fwd(x, g, b, eps)
        ctx.save_for_backward(x, g, mean, rstd)
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        x, g, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None, None
        dy = dy if dy.is_contiguous() else dy.contiguous()
        if dres is not None and not dres.is_contiguous():
            dres = dres.contiguous()
        dx, dg, db = _ln_bwd(ctx, x, dy, g, mean, rstd, dres=dres)
        return dx, dg, db, None, None, None


def layer_norm_res(x, gamma, beta, eps):
    """-> (x, LayerNorm(x)): use the returned x for the residual connection."""
    return LayerNormResFn.apply(x, gamma, beta, eps, *_dp_of(x))


@direct_grads(1, 2)
class LayerNormResPatchFn(Function):
    """LayerNormResFn for the norm in front of a MiT attention with spatial reduction (mit.py:143, 47): besides (x, norm(x)) it returns
    norm(x) a second time in the patch-major row order of the k = s = sr convolution's im2col matrix, [B Ho Wo, sr sr C] -- written by the same
    kernel (no im2col pass); the backward reads that output's gradient in the same order (no col2im pass) and sums it with the other
    consumer's on load, as LayerNormForkFn does for two consumers in token order."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, lw, ls, dp_scale=None, dp_rpg=1):
        ctx.dp = (dp_scale, int(dp_rpg)) if dp_scale is not None else None
        ctx.patch = (int(lw), int(ls))
        x = x if x.is_contiguous() else x.contiguous()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y, mean, rstd, col = hip.layernorm_fwd(x, g, b, eps, patch=ctx.patch)
        ctx.save_for_backward(x, g, mean, rstd)
        return x.view_as(x), y, col

    @staticmethod
    def backward(ctx, dres, dy, dcol):
        x, g, mean, rstd = ctx.saved_tensors
        if dy is None and dcol is None:
            return dres, None, None, None, None, None, None, None
        if dy is None:                       # (only the convolution consumed the norm: its gradient back in token order first)
            raise RuntimeError('layer_norm_res_patch: the token-order output received no gradient')
        dy = dy if dy.is_contiguous() else dy.contiguous()
        if dres is not None and not dres.is_contiguous():
            dres = dres.contiguous()
        if dcol is not None and not dcol.is_contiguous():
            dcol = dcol.contiguous()
        dx, dg, db = _ln_bwd(ctx, x, dy, g, mean, rstd, dy2=dcol, dres=dres, dy2_patch=ctx.patch if dcol is not None else None)
        return dx, dg, db, None, None, None, None, None


def patch_layout_ok(W, H, sr):
    """layer_norm_res_patch covers maps whose width and reduction ratio are powers of two (every BASELINE geometry)"""
    return (sr > 1 and (sr & (sr - 1)) == 0 and (W & (W - 1)) == 0 and W >= sr and H % sr == 0
            and not hip.policy('no_ln_patch'))


def layer_norm_res_patch(x, gamma, beta, eps, W, sr):
    """-> (x, LayerNorm(x), im2col matrix of LayerNorm(x) for a k = s = sr convolution [rows / sr^2, sr^2 C])"""
    return LayerNormResPatchFn.apply(x, gamma, beta, eps, W.bit_length() - 1, sr.bit_length() - 1, *_dp_of(x))


@direct_grads(1, 2)
class ConvFromColFn(Function):
    """The k = s = sr convolution of MiT's spatial reduction (mit.py:21,48) on an im2col matrix that already exists
    (layer_norm_res_patch): y = col W^T + b; the data gradient is returned in the matrix's own layout."""

    @staticmethod
    def forward(ctx, col, weight, bias, k):
        col = _rowmajor(col)
        M, K = col.shape
        O, Cin = weight.shape[0], weight.shape[1]
        assert K == k * k * Cin
        dtype = col.dtype
        wsrc = weight.detach().contiguous()

        def make_wmat():
            t = torch.empty((O, k * k, Cin), dtype=dtype, device=col.device)
            return t, [('perm', wsrc, t, O, Cin, k * k, Cin)]
        wmat = derived_weight(wsrc, ('patch', dtype, K), make_wmat).view(O, K)                 # [O][(ky,kx)][ci]
        y = hip.gemm(0, col, wmat, M, O, K, bias=bias.detach() if bias is not None else None)
        ctx.save_for_backward(col, wmat)
        ctx.meta = (M, O, K, k, Cin, bias is not None, weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        col, wmat = ctx.saved_tensors
        M, O, K, k, Cin, has_bias, wshape = ctx.meta
        dy = _rowmajor(dy)
        dcol = dw = db = None
        gw, gb = gslot(ctx, 1), gslot(ctx, 2)
        if ctx.needs_input_grad[1] and has_bias and ctx.needs_input_grad[2]:
            if not (gw is not None and _queue_dw_post(ctx, dy, col, O, K, M, lambda t, g: ('perm', t, g, O, k * k, Cin, k * k))):
                dwm, db = hip.gemm_dw_db(dy, col, O, K, M, split_k=_splitk(O, K, M), db_out=gb)           # [O][(ky,kx)][ci]
                dw = hip.permute021(dwm, O, k * k, Cin, torch.float32, out=gw).view(wshape)
        else:
            if ctx.needs_input_grad[1]:
                dwm = hip.gemm(2, dy, col, O, K, M, out_dtype=torch.float32, split_k=_splitk(O, K, M))
                dw = hip.permute021(dwm, O, k * k, Cin, torch.float32, out=gw).view(wshape)
            if has_bias and ctx.needs_input_grad[2]:
                db = hip.colsum(dy, out=gb)
        if ctx.needs_input_grad[0]:
            dcol = hip.gemm(1, dy, wmat, M, K, O)
        return dcol, dw, db, None


def conv_from_col(col, weight, bias, k):
    return ConvFromColFn.apply(col, weight, bias, k)


@direct_grads(1, 2)
class LayerNormForkFn(Function):
    """LayerNorm whose output has TWO consumers (a MiT / ConvNeXt stage output feeds the decode head and the next stage,
    mit.py:196-216): returns the normalised map twice; the backward kernel sums the two incoming gradients on load."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, dp_scale=None, dp_rpg=1):
        ctx.dp = (dp_scale, int(dp_rpg)) if dp_scale is not None else None
        x = x if x.is_contiguous() else x.contiguous()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        y, mean, rstd = hip.layernorm_fwd(x, g, b, eps)
        ctx.save_for_backward(x, g, mean, rstd)
        return y, y.view_as(y)

    @staticmethod
    def backward(ctx, dy1, dy2):
        x, g, mean, rstd = ctx.saved_tensors
        if dy1 is None:
            dy1, dy2 = dy2, None
        dy1 = dy1 if dy1.is_contiguous() else dy1.contiguous()
        if dy2 is not None and not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx, dg, db = _ln_bwd(ctx, x, dy1, g, mean, rstd, dy2=dy2)
        return dx, dg, db, None, None, None


def layer_norm_fork(x, gamma, beta, eps):
    return LayerNormForkFn.apply(x, gamma, beta, eps, *_dp_of(x))


class ForkFn(Function):
    """x -> n aliases of x for n consumers; the backward adds the n gradients with the library's own add kernel (autograd's
    implicit accumulation would run a framework kernel inside the captured step)."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        for g in gs[1:]:
            acc = hip.add(_rowmajor(acc), _rowmajor(g))
        return acc, None


def fork(x, n=2):
    return ForkFn.apply(x, n)


_RNG_SALT = [0]


def stochastic_scales(owner, keep_probs, row_len, device):
    """fp32 [len(keep_probs), row_len]: per entry 1 / kp with probability kp else 0 (DropPath: one row per draw, row_len = batch;
    Dropout2d: one row of B * C entries).  The generator state {seed, launch counter} lives on the device and is advanced by
    the kernel (segf_bernoulli_scale), so a replayed hipGraph draws fresh numbers; seeded from torch.initial_seed() and the
    order in which the layers first draw.  First call per layer = outside graph capture (the warm-up pass)."""
    key = (str(device), tuple(keep_probs), int(row_len))
    st = getattr(owner, '_segf_rng', None)
    if st is None or st[0] != key:
        _RNG_SALT[0] += 1
        seed = (torch.initial_seed() * 6364136223846793005 + _RNG_SALT[0] * 1442695040888963407) & 0x7FFFFFFFFFFFFFFF
        st = (key, torch.tensor([seed, 0], dtype=torch.int64, device=device),
              torch.tensor(list(keep_probs), dtype=torch.float32, device=device))
        owner._segf_rng = st
    return hip.bernoulli_scale(st[1], st[2], len(keep_probs) * int(row_len), int(row_len)).view(len(keep_probs), int(row_len))


@direct_grads(1, 2)
class ConvPatchFn(Function):
    """Strided conv as im2col + MFMA GEMM: PatchEmbed (mit.py:105,127: k7 s4 p3 / k3 s2 p1), the
    spatial-reduction conv (mit.py:21,48: k = s = sr) and ConvNeXt's stem / downsample convs.
    x is either the fp32 NCHW image (image=True) or NHWC tokens [B*H*W, Cin]."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom, image, dtype):
        B, H, W, Cin, k, stride, pad = geom
        O = weight.shape[0]
        Ho = (H + 2 * pad - k) // stride + 1
        Wo = (W + 2 * pad - k) // stride + 1
        K = k * k * Cin
        ld = (K + 7) // 8 * 8
        # the stem (7 x 7 x 3 = 147 columns, 2 M tokens at cfg2): columns padded to whole 32-steps (160) so that the forward
        # product takes the streaming kernel (rows are its MFMA operand as they lie in memory); the pad columns are zeros on
        # both sides.  The tiled kernel spent 392 us on this [2M x 147] -> 32 product, 75 % of its tile columns empty
        stream_form = (image and dtype == torch.bfloat16 and K % 32 != 0 and (K + 31) // 32 * 32 == 160 and O % 32 == 0
                       and B * Ho * Wo >= 16384)
        if stream_form:
            ld = 160
        x = x if x.is_contiguous() else x.contiguous()
        col = hip.im2col(x, dtype, image, B, H, W, Cin, k, k, stride, pad, Ho, Wo, ld)
        wsrc = weight.detach().contiguous()
        b = bias.detach() if bias is not None else None
        if stream_form:
            def make_wpad():                             # rows of [(ky,kx)][ci] at a row stride of `ld`, the pad columns zero
                t = torch.empty((O, ld), dtype=dtype, device=x.device)
                return t, [('perm', wsrc, t, O, Cin, k * k, Cin, ld), ('zero', t[:, K:])]
            wpad = derived_weight(wsrc, ('patch', dtype, ld), make_wpad)
            wmat = wpad[:, :K]
            y = hip.gemm(0, col, wpad, B * Ho * Wo, O, ld, bias=b)
        else:
            def make_wmat():
                t = torch.empty((O, k * k, Cin), dtype=dtype, device=x.device)
                return t, [('perm', wsrc, t, O, Cin, k * k, Cin)]
            wmat = derived_weight(wsrc, ('patch', dtype, K), make_wmat).view(O, K)             # [O][(ky,kx)][ci]
            y = hip.gemm(0, col, wmat, B * Ho * Wo, O, K, bias=b)
        # the im2col matrix itself is kept for the weight gradient (k*k/stride^2 <= 3.1x the conv input, < 1 GB in total
        # for SegFormer-B0 at batch 64 out of 288 GB): rebuilding it in backward costs a second pass over the image
        ctx.save_for_backward(col, wmat)
        ctx.meta = (geom, image, dtype, O, Ho, Wo, K, ld, bias is not None, weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        col, wmat = ctx.saved_tensors
        geom, image, dtype, O, Ho, Wo, K, ld, has_bias, wshape = ctx.meta
        B, H, W, Cin, k, stride, pad = geom
        M = B * Ho * Wo
        dy = _rowmajor(dy)
        dx = dw = db = None
        gw, gb = gslot(ctx, 1), gslot(ctx, 2)                # flat-gradient views: the permute / column sum write in place
        if ctx.needs_input_grad[1] and has_bias and ctx.needs_input_grad[2]:
            if not (gw is not None and _queue_dw_post(ctx, dy, col, O, K, M, lambda t, g: ('perm', t, g, O, k * k, Cin, k * k))):
                dwm, db = hip.gemm_dw_db(dy, col, O, K, M, split_k=_splitk(O, K, M), db_out=gb)           # [O][(ky,kx)][ci]
                dw = hip.permute021(dwm, O, k * k, Cin, torch.float32, out=gw).view(wshape)
        else:
            if ctx.needs_input_grad[1]:
                dwm = hip.gemm(2, dy, col, O, K, M, out_dtype=torch.float32, split_k=_splitk(O, K, M))
                dw = hip.permute021(dwm, O, k * k, Cin, torch.float32, out=gw).view(wshape)
            if has_bias and ctx.needs_input_grad[2]:
                db = hip.colsum(dy, out=gb)
        if ctx.needs_input_grad[0] and not image:
            dcol = hip.gemm(1, dy, wmat, M, K, O)
            dx = hip.col2im(dcol, B, H, W, Cin, k, k, stride, pad, Ho, Wo)
        return dx, dw, db, None, None, None


def conv_patch(x, weight, bias, geom, image=False, dtype=None):
    return ConvPatchFn.apply(x, weight, bias, geom, image, dtype or x.dtype)


class AttentionFn(Function):
    """softmax(Q K^T * scale) V per head (mit.py:52-57).  q: [B*N, C]; kv: [B*Nkv, 2C] = [k | v]."""

    @staticmethod
    def forward(ctx, q, kv, B, N, Nkv, heads):
        q, kv = _rowmajor(q), _rowmajor(kv)
        Cc = q.shape[1]
        hd = Cc // heads
        scale = hd ** -0.5
        o, lse = hip.attention_fwd(q, kv[:, :Cc], kv[:, Cc:], B, heads, N, Nkv, hd, scale)
        ctx.save_for_backward(q, kv, o, lse)
        ctx.meta = (B, N, Nkv, heads, hd, scale, Cc)
        return o

    @staticmethod
    def backward(ctx, do):
        q, kv, o, lse = ctx.saved_tensors
        B, N, Nkv, heads, hd, scale, Cc = ctx.meta
        do = _rowmajor(do)
        dkv = torch.empty_like(kv)
        dq = hip.attention_bwd(q, kv[:, :Cc], kv[:, Cc:], o, do, lse, B, heads, N, Nkv, hd, scale, dkv[:, :Cc], dkv[:, Cc:])
        return dq, dkv, None, None, None, None


def attention(q, kv, B, N, Nkv, heads):
    return AttentionFn.apply(q, kv, B, N, Nkv, heads)


@direct_grads(1, 2)
class DWConvGeluFn(Function):
    """gelu(depthwise3x3(x) + b) on NHWC tokens (mit.py:62-71 + F.gelu at :99)."""

    @staticmethod
    def forward(ctx, x, weight, bias, B, H, W, apply_gelu):
        x = x if x.is_contiguous() else x.contiguous()
        Cc = x.shape[1]
        w9 = weight.detach().reshape(Cc, 9).contiguous()
        b = bias.detach().contiguous() if bias is not None else None
        y = hip.dwconv3x3_gelu_fwd(x, w9, b, B, H, W, Cc, apply_gelu)
        ctx.save_for_backward(x, w9, b)
        ctx.meta = (B, H, W, Cc, apply_gelu, weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w9, b = ctx.saved_tensors
        B, H, W, Cc, apply_gelu, wshape = ctx.meta
        dy = dy if dy.is_contiguous() else dy.contiguous()
        gw, gb = gslot(ctx, 1), (gslot(ctx, 2) if b is not None else None)       # flat-gradient views: written in place, no copy
        q = _FIN_QUEUE
        if q is not None and gw is not None and gb is not None and gb.data_ptr() == gw.data_ptr() + 4 * gw.numel() and gw.is_contiguous():
            # the finalize of the (dw, db) partial sums joins the step's grouped finalize launches (one instead of two launches per block)
            slots = ctx._gslots
            dx, item = hip.dwconv3x3_gelu_bwd(x, w9, b, dy, B, H, W, Cc, apply_gelu, dw_out=gw, db_out=gb, defer=True)
            q.append((item, (slots[1], slots[2])))
            if len(q) >= FIN_QUEUE_MAX:
                _flush_finalizes()
            return dx, None, None, None, None, None, None
        dx, dw, db = hip.dwconv3x3_gelu_bwd(x, w9, b, dy, B, H, W, Cc, apply_gelu, dw_out=gw, db_out=gb)
        return dx, dw.view(wshape), (db if b is not None else None), None, None, None, None


def dwconv3x3_gelu(x, weight, bias, B, H, W, apply_gelu=True):
    return DWConvGeluFn.apply(x, weight, bias, B, H, W, apply_gelu)


@direct_grads(1, 2)
class BatchNormActFn(Function):
    """BatchNorm2d (+ReLU/ReLU6) (+Dropout2d channel scale) on NHWC rows.
    ConvModule of heads/segformer.py:21-29, layers/conv_module.py:4-9, mobilenetv2.py:5-11; Dropout2d of
    heads/segformer.py:40,57.  Training uses batch statistics and updates the running buffers in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, act, chan_scale, rows_per_sample,
                pre_sums=None):
        x = x if x.is_contiguous() else x.contiguous()
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        if training and pre_sums is not None:
            mean, rstd = hip.bn_stats_from_sums(pre_sums, x.shape[0], running_mean, running_var, momentum, eps)
        elif training:
            mean, rstd = hip.bn_stats(x, running_mean, running_var, momentum, eps)
        else:
            mean = running_mean.detach().clone()
            rstd = torch.rsqrt(running_var.detach() + eps)
        y = hip.bn_apply(x, mean, rstd, g, b, act, chan_scale, rows_per_sample or 1)
        ctx.save_for_backward(x, mean, rstd, g, b, chan_scale)
        ctx.meta = (act, rows_per_sample or 1, not training)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, g, b, chan_scale = ctx.saved_tensors
        act, rps, eval_mode = ctx.meta
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx, dg, db = hip.bn_bwd(x, dy, mean, rstd, g, b, act, chan_scale, rps, eval_mode)
        return dx, dg, db, None, None, None, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5, act=1, chan_scale=None,
                   rows_per_sample=None, pre_sums=None):
    return BatchNormActFn.apply(x, gamma, beta, running_mean, running_var, training, momentum, eps, act, chan_scale,
                                rows_per_sample, pre_sums)


@direct_grads(1, 2, 11, 12)
class BnActLinearFn(Function):
    """ConvModule's BatchNorm2d + ReLU, the Dropout2d channel scale and the following 1x1 conv (heads/segformer.py:21-29,40,
    57-58: linear_fuse.bn/activate -> dropout -> linear_pred) as ONE product: the normalisation is folded into per-(sample,
    channel) scale / shift tables and applied while the GEMM stages its activation operand, in the forward product and again in
    the weight-gradient product, so the normalised [B*H*W, E] tensor is never written.  Same arithmetic as
    BatchNormActFn + LinearFn (bf16 rounding of the normalised value included)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, act, chan_scale, rps, weight, bias, Np,
                pre_sums, fold_slot=None):
        ctx.fold_slot = fold_slot
        x = x if x.is_contiguous() else x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        if training and pre_sums is not None:       # the producer already summed x and x^2 per channel
            mean, rstd = hip.bn_stats_from_sums(pre_sums, M, running_mean, running_var, momentum, eps)
        elif training:
            mean, rstd = hip.bn_stats(x, running_mean, running_var, momentum, eps)
        else:
            mean = running_mean.detach().clone()
            rstd = torch.rsqrt(running_var.detach() + eps)
        scale, shift = hip.bn_affine_table(mean, rstd, g, b, chan_scale, M // rps, K)
        w, bp = _padded_rows(weight, bias, N, Np, K, x.dtype, x.device)
        y = hip.gemm_pro(0, x, w, M, Np, K, scale, shift, rps, act, bias=bp)[:, :N]
        ctx.save_for_backward(x, mean, rstd, g, b, chan_scale, scale, shift, w)
        ctx.meta = (M, N, K, Np, act, rps, not training, bias is not None, weight.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, g, b, chan_scale, scale, shift, w = ctx.saved_tensors
        M, N, K, Np, act, rps, eval_mode, has_bias, wshape = ctx.meta
        if dy.stride(-1) == 1 and dy.stride(0) == Np:
            dyp = dy.as_strided((M, Np), (Np, 1))           # the producer zero-fills the pad columns (segfac.h)
        else:
            dyp = hip.zeros((M, Np), x.dtype, x.device)
            hip.cast2d(_rowmajor(dy), dyp[:, :N])
        db = hip.colsum(dyp)[:N] if has_bias else None
        dw = None
        if hip.bn_cls_bwd_supported(x.dtype, M, K, Np, rps):
            # da = dyp W is recomputed inside both BatchNorm passes (K = #classes terms per element) instead of being written
            # once and read twice: 19.8 -> 10.9 GB at cfg2, batch 128 (head_fused.hip)
            slot = ctx.fold_slot
            x1 = slot.get('x1') if slot is not None else None
            ride_dg = x1 is not None and x1.shape[0] == M and hip.bn_cls_bwd_dw_supported(x.dtype, M, K, Np, rps, x1.shape[1])
            if act in (0, 1) and not hip.policy('no_head_fused_cw'):
                # ... and the classifier's own weight gradient rides on the first pass (it has the x and dy tiles on chip), as the
                # stage-1 weight gradient of the folded head rides on the second: no separate pass over x for either
                dx, dg, dbeta, dG, dwc = hip.bn_cls_bwd_full(dyp, w, x, mean, rstd, g, b, act, chan_scale, rps, eval_mode,
                                                             x1=x1 if ride_dg else None)
                dw = dwc[:N].view(wshape)
                if ride_dg:
                    slot['dG'] = dG
            elif ride_dg:
                # x is the folded SegFormerHead's stride-4 map: the weight-gradient product of its stage-1 term (dx^T x1 and the
                # column sums of dx) rides on the second BatchNorm pass, which has the dx tile on chip (one 3.2 GB pass less)
                dx, dg, dbeta, slot['dG'] = hip.bn_cls_bwd_dw(dyp, w, x, mean, rstd, g, b, act, chan_scale, rps, eval_mode, x1)
            else:
                dx, dg, dbeta = hip.bn_cls_bwd(dyp, w, x, mean, rstd, g, b, act, chan_scale, rps, eval_mode)
        else:
            da = hip.gemm(1, dyp, w, M, K, Np)               # gradient w.r.t. the (never materialised) normalised tensor
            dx, dg, dbeta = hip.bn_bwd(x, da, mean, rstd, g, b, act, chan_scale, rps, eval_mode)
        if dw is None:
            dw = hip.gemm_pro(2, dyp, x, Np, K, M, scale, shift, rps, act, split_k=_splitk(Np, K, M))[:N].view(wshape)
        return dx, dg, dbeta, None, None, None, None, None, None, None, None, dw, db, None, None, None


def bn_act_linear(x, gamma, beta, running_mean, running_var, training, momentum, eps, act, chan_scale, rows_per_sample, weight,
                  bias, pad_to=None, pre_sums=None):
    """batch_norm_act followed by linear; fused into the GEMM operand load when the shape takes the 256-tile kernel.
    pre_sums: optional fp32 [2, C] (sum, sum of squares) of x per channel from its producer (replaces the statistics pass)."""
    M, K = x.shape
    N = weight.shape[0]
    Np = pad_to if (pad_to and pad_to > N) else N
    rps = rows_per_sample or M
    if (x.dtype == torch.bfloat16 and act in (0, 1) and M % rps == 0 and hip.gemm_pro_supported(x.dtype, 0, M, Np, K, rps)
            and hip.gemm_pro_supported(x.dtype, 2, Np, K, M, rps)):
        return BnActLinearFn.apply(x, gamma, beta, running_mean, running_var, training, momentum, eps, act, chan_scale, rps,
                                   weight, bias, Np, pre_sums, getattr(x, '_segf_fold', None))
    y = batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum, eps, act, chan_scale, rows_per_sample,
                       pre_sums=pre_sums)
    return linear(y, weight, bias, pad_to=pad_to)


@direct_grads(5, 6, 7, 8, 9, 10, 11, 12)
class SegformerProjectConcatFn(Function):
    """The front half of SegFormerHead.forward (heads/segformer.py:42-50): per-scale Linear(C_i -> E),
    bilinear resize to the stride-4 grid, channel concat in the order [c4, c3, c2, c1].  Every branch writes
    straight into its column slice of one [B*H1*W1, 4E] buffer -- no torch.cat copy."""

    @staticmethod
    def forward(ctx, geoms, *args):
        feats, weights, biases = args[0:4], args[4:8], args[8:12]
        B = geoms[0][0]
        H1, W1 = geoms[0][1], geoms[0][2]
        E = weights[0].shape[0]
        dtype = feats[0].dtype
        M1 = B * H1 * W1
        cat = torch.empty((M1, 4 * E), dtype=dtype, device=feats[0].device)
        saved = []
        for i in range(4):
            f = _rowmajor(feats[i])
            w = _w(weights[i], dtype)
            _, h, wd = geoms[i]
            sl = cat[:, (3 - i) * E:(4 - i) * E]
            if i == 0:
                hip.gemm(0, f, w, M1, E, f.shape[1], out=sl, bias=biases[i].detach())
            else:
                t = hip.gemm(0, f, w, B * h * wd, E, f.shape[1], bias=biases[i].detach())
                hip.bilinear_fwd(t, B, h, wd, E, H1, W1, sl, align_corners=False)
            saved += [f, w]
        ctx.save_for_backward(*saved)
        ctx.meta = (geoms, E)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        geoms, E = ctx.meta
        saved = ctx.saved_tensors
        dcat = _rowmajor(dcat)
        B, H1, W1 = geoms[0]
        dfs, dws, dbs = [], [], []
        for i in range(4):
            f, w = saved[2 * i], saved[2 * i + 1]
            _, h, wd = geoms[i]
            sl = dcat[:, (3 - i) * E:(4 - i) * E]
            dy = sl if i == 0 else hip.bilinear_bwd(sl, B, h, wd, E, H1, W1, align_corners=False)
            M, Ci = f.shape
            dfs.append(hip.gemm(1, dy, w, M, Ci, E) if ctx.needs_input_grad[1 + i] else None)
            dws.append(hip.gemm(2, dy, f, E, Ci, M, out_dtype=torch.float32, split_k=_splitk(E, Ci, M)))
            dbs.append(hip.colsum(dy))
        return (None, *dfs, *dws, *dbs)


def segformer_project_concat(feats, weights, biases, geoms):
    return SegformerProjectConcatFn.apply(tuple(geoms), *feats, *weights, *biases)


@direct_grads(5, 6, 7, 8, 9, 10, 11, 12, 13)
class SegformerFoldedFuseFn(Function):
    """linear_fuse.conv( cat_i( resize( linear_c{i}(x_i) ) ) ) of SegFormerHead.forward (heads/segformer.py:42-56) with
    the algebra folded: Linear -> bilinear resize -> concat -> 1x1 conv (no bias) is affine in x_i, and bilinear resizing
    commutes with affine maps (its weights sum to one), so
        y = sum_i resize_i( x_i G_i^T ) + beta,   G_i = F_i W_i  [E, C_i],   beta = sum_i F_i b_i,
    where F_i is the column block of the fuse weight that multiplies scale i in the reversed concat [c4, c3, c2, c1].
    The [B*H1*W1, 4E] concat buffer and the 4E -> E GEMM over it (77 of the model's 89 GFLOP/img at cfg2) never exist;
    the per-scale products run at native resolution (1.6 GFLOP/img).  The backward returns gradients for the ORIGINAL
    parameters (linear_c{i}.proj.{weight,bias}, linear_fuse.conv.weight) by the chain rule through G_i and beta."""

    @staticmethod
    def forward(ctx, geoms, *args):
        feats, weights, biases, wf = args[0:4], args[4:8], args[8:12], args[12]
        ctx.fold_slot = args[13] if len(args) > 13 else None
        B, H1, W1 = geoms[0]
        E = weights[0].shape[0]
        dtype = feats[0].dtype
        dev = feats[0].device
        wf32 = wf.detach().reshape(E, 4 * E)
        wfc = _w(wf32 if wf32.is_contiguous() else wf32.contiguous(), dtype)
        ts, saved = [], []
        # one launch for the whole stride-4 map (x1 G1^T and the three resizes as one accumulated matrix product per 8 x 8 pixel
        # block, csrc/fuse_map.hip) when the geometry is the 1/2-1/4-1/8 pyramid; the stage-1 bias then rides in the 1/2 map
        one_pass = (dtype == torch.bfloat16 and all(geoms[i] == (B, H1 >> i, W1 >> i) for i in (1, 2, 3))
                    and hip.fuse_map_248_supported(dtype, B, H1, W1, E, feats[0].shape[1]))
        G1, beta1 = None, None
        # W'_i = [W_i | b_i | 0] (bias as one extra input column): one GEMM yields G_i = F_i W_i and beta_i = F_i b_i.  The packing of
        # the four W'_i is ONE launch, the unpacking of the four products another (hip.prep_grouped) -- at batch 4 these 24 jobs were
        # 24 launches of the launch floor each
        xs, Wps, Gps, Gs, betas, jobs = [], [], [], [], [], []
        for i in range(4):
            x = _rowmajor(feats[i])
            Ci = x.shape[1]
            Wp = torch.empty((E, Ci + 8), dtype=dtype, device=dev)
            jobs += [('cast', weights[i].detach(), Wp[:, :Ci]), ('cast', biases[i].detach().unsqueeze(1), Wp[:, Ci:Ci + 1]),
                     ('zero', Wp[:, Ci + 1:])]
            xs.append(x)
            Wps.append(Wp)
        hip.prep_grouped(jobs)
        jobs = []
        for i in range(4):
            Ci = xs[i].shape[1]
            Fi = wfc[:, (3 - i) * E:(4 - i) * E]
            Gp = hip.gemm(1, Fi, Wps[i], E, Ci + 8, E, out_dtype=torch.float32)                # [E, Ci+8] fp32
            G = torch.empty((E, Ci), dtype=dtype, device=dev)
            beta_i = torch.empty(E, dtype=torch.float32, device=dev)
            jobs += [('cast', Gp[:, :Ci], G), ('cast', Gp[:, Ci:Ci + 1], beta_i.unsqueeze(1))]
            Gps.append(Gp)
            Gs.append(G)
            betas.append(beta_i)
        hip.prep_grouped(jobs)
        for i in range(4):
            x, G, beta_i, Wp = xs[i], Gs[i], betas[i], Wps[i]
            Ci = x.shape[1]
            _, h, w = geoms[i]
            if one_pass and i == 0:
                G1, beta1 = G, beta_i
                ts.append(None)
            else:
                if one_pass and i == 1:
                    beta_i = hip.add(beta_i.unsqueeze(0), beta1.unsqueeze(0)).squeeze(0)    # a constant passes through the resize
                ts.append(hip.gemm(0, x, G, B * h * w, E, Ci, bias=beta_i))   # a constant row passes through the resize unchanged
            saved += [x, G, Wp]
            if i == 0 and ctx.fold_slot is not None and dtype == torch.bfloat16:
                ctx.fold_slot['x1'] = x           # BnActLinearFn.backward may compute dG_1 while it has the gradient tile on chip
        if one_pass:
            y, sums = hip.fuse_map_248(saved[0], G1, ts[1], ts[2], ts[3], B, H1, W1)
        else:
            y, sums = hip.upsample_add_stats(ts[0], [(ts[i], geoms[i][1], geoms[i][2]) for i in range(1, 4)], B, H1, W1, E)
        if sums is None:
            sums = torch.empty(0, device=y.device)          # geometry without the fused statistics: the consumer runs its own pass
        ctx.mark_non_differentiable(sums)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(wfc, *saved)
        ctx.meta = (geoms, E, dtype)
        return y, sums

    @staticmethod
    def backward(ctx, dy, _dsums=None):
        geoms, E, dtype = ctx.meta
        sv = ctx.saved_tensors
        wfc = sv[0]
        B, H1, W1 = geoms[0]
        dy = _rowmajor(dy)
        dev = dy.device
        dbeta = None                         # [E] fp32 = colsum(dy) (colsum(resize^T(dy)) == colsum(dy): the resize weights sum
                                             # to one); it rides along with the stage-1 weight-gradient product below
        dwf = torch.empty((E, 4 * E), dtype=torch.float32, device=dev)
        dxs, dws, dbs = [], [], []
        dts = None
        if (H1 % 8 == 0 and W1 % 8 == 0 and E % 8 == 0 and not hip.policy('no_bwd248')
                and all(geoms[i][1:] == (H1 >> i, W1 >> i) for i in (1, 2, 3))):
            dts = hip.bilinear_bwd_248(dy, B, H1, W1, E)          # the three transposed resizes in ONE pass over dy
        dGps = []
        for i in range(4):
            x, G, Wp = sv[1 + 3 * i], sv[2 + 3 * i], sv[3 + 3 * i]
            _, h, w = geoms[i]
            M, Ci = x.shape
            dt = dy if i == 0 else (dts[i - 1] if dts is not None else hip.bilinear_bwd(dy, B, h, w, E, H1, W1, align_corners=False))
            dxs.append(hip.gemm(1, dt, G, M, Ci, E) if ctx.needs_input_grad[1 + i] else None)
            pre = ctx.fold_slot.pop('dG', None) if (i == 0 and ctx.fold_slot is not None) else None
            if pre is not None and tuple(pre.shape) == (E, Ci + 8):
                dGp = pre                          # [dy^T x_1 | colsum(dy) | 0] arrived with dy (segf_bn_cls_bwd_dw)
                dbeta = dGp[:, Ci]
            else:
                dGp = torch.empty((E, Ci + 8), dtype=torch.float32, device=dev)             # d [G_i | . ]: the product fills [:, :Ci]
                if i == 0:
                    _, dbeta = hip.gemm_dw_db(dt, x, E, Ci, M, split_k=_splitk(E, Ci, M), out=dGp[:, :Ci])
                else:
                    hip.gemm(2, dt, x, E, Ci, M, out=dGp[:, :Ci], split_k=_splitk(E, Ci, M))
            dGps.append(dGp)
        # d [G_i | beta_i | 0] in the compute dtype, all four in one launch
        dGcs, jobs = [], []
        for i in range(4):
            Ci = sv[1 + 3 * i].shape[1]
            if dtype == torch.float32:
                dGc = dGps[i]
                jobs += [('zero', dGc[:, Ci + 1:])] + ([('cast', dbeta.unsqueeze(1), dGc[:, Ci:Ci + 1])] if dbeta.data_ptr() != dGc[:, Ci].data_ptr() else [])
            else:
                dGc = torch.empty((E, Ci + 8), dtype=dtype, device=dev)
                jobs += [('cast', dGps[i][:, :Ci], dGc[:, :Ci]), ('cast', dbeta.unsqueeze(1), dGc[:, Ci:Ci + 1]), ('zero', dGc[:, Ci + 1:])]
            dGcs.append(dGc)
        hip.prep_grouped(jobs)
        for i in range(4):
            Wp, dGc = sv[3 + 3 * i], dGcs[i]
            Ci = sv[1 + 3 * i].shape[1]
            Fi = wfc[:, (3 - i) * E:(4 - i) * E]
            hip.gemm(0, dGc, Wp, E, E, Ci + 8, out=dwf[:, (3 - i) * E:(4 - i) * E])          # dF_i = dG_i W_i^T + dbeta b_i^T
            dWp = hip.gemm(2, Fi, dGc, E, Ci + 8, E, out_dtype=torch.float32)               # [dW_i | db_i | 0] = F_i^T dG'_i
            dws.append(dWp[:, :Ci])
            dbs.append(dWp[:, Ci])
        if ctx.fold_slot is not None:
            ctx.fold_slot.clear()
        return (None, *dxs, *dws, *dbs, dwf.view(E, 4 * E, 1, 1), None)


def segformer_folded_fuse(feats, weights, biases, fuse_weight, geoms):
    """-> (y [B*H1*W1, E], sums): sums = fp32 [2, E] per-channel (sum, sum of squares) of y for the BatchNorm that follows, or
    None when the geometry did not take the fused kernel."""
    slot = {}
    y, sums = SegformerFoldedFuseFn.apply(tuple(geoms), *feats, *weights, *biases, fuse_weight, slot)
    y._segf_fold = slot          # read by bn_act_linear: its backward can compute this function's stage-1 weight gradient in passing
    return y, (sums if sums.numel() else None)


@direct_grads(1, 2)
class DWConv7Fn(Function):
    """Depthwise 7x7 conv + bias on NHWC tokens (ConvNeXt Block.dwconv, convnext.py:29,39; convnextv2.py:88,101)."""

    @staticmethod
    def forward(ctx, x, weight, bias, B, H, W):
        x = x if x.is_contiguous() else x.contiguous()
        Cc = x.shape[1]
        wsrc = weight.detach().contiguous()

        def make_wt():
            t = torch.empty((49, Cc), dtype=torch.float32, device=x.device)
            return t, [('perm', wsrc, t, 1, Cc, 49, Cc)]
        wt = derived_weight(wsrc, ('dw7',), make_wt)                                              # [49][C] fp32
        y = hip.dwconv7x7_fwd(x, wt, bias.detach() if bias is not None else None, B, H, W, Cc)
        ctx.save_for_backward(x, wt)
        ctx.meta = (B, H, W, Cc, weight.shape, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wt = ctx.saved_tensors
        B, H, W, Cc, wshape, has_bias = ctx.meta
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx, dw, db = hip.dwconv7x7_bwd(x, wt, dy, B, H, W, Cc, need_dx=ctx.needs_input_grad[0], need_db=has_bias)
        return dx, dw.view(wshape), db, None, None, None


def dwconv7x7(x, weight, bias, B, H, W):
    return DWConv7Fn.apply(x, weight, bias, B, H, W)


@direct_grads(1)
class Conv3x3Fn(Function):
    """3x3 / stride 1 / pad 1 convolution without bias on NHWC tokens as an implicit MFMA GEMM (ConvModule(c1, c2, 3, 1, 1):
    heads/upernet.py:26,28, modules/ppm.py:19, heads/fpn.py:19).  x may be a column slice of a wider concat buffer."""

    @staticmethod
    def forward(ctx, x, weight, B, H, W, fp8=False):
        x = _rowmajor(x)
        O, I = weight.shape[0], weight.shape[1]
        wsrc = weight.detach().contiguous()
        cdt = x.dtype

        def make_wm():
            t = torch.empty((O, 9, I), dtype=cdt, device=x.device)
            return t, [('perm', wsrc, t, O, I, 9, I)]
        wm = derived_weight(wsrc, ('conv3x3', cdt), make_wm).view(O, 9 * I)                         # [O][(ky,kx)][ci]
        use8 = bool(fp8) and x.dtype == torch.bfloat16 and hip.conv3x3_fp8_supported(0, B, H, W, I, O)
        xq = sx = None
        if use8:
            # fp8 forward (BASELINE cfg5): activations e4m3 with one dynamic scale for the tensor, weights e4m3 per output channel
            xq, sx = hip.quant_tensor_fp8(x)
            wq, sw = hip.quant_rows_fp8(wm)
            y = hip.conv3x3_fp8(0, xq, sx, wq, sw, B, H, W, I, O)
        else:
            y = hip.conv3x3(0, x, wm, B, H, W, I, O)
        # the quantised input is kept for the fp8 weight gradient (one byte per element next to the bf16 tensor autograd keeps anyway)
        keep8 = use8 and hip.conv3x3_fp8_wgrad_supported(B, H, W, I, O)
        ctx.save_for_backward(x, weight.detach(), xq if keep8 else None, sx if keep8 else None)
        ctx.meta = (B, H, W, I, O, weight.shape, bool(fp8))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, xq, sx = ctx.saved_tensors
        B, H, W, I, O, wshape, fp8 = ctx.meta
        dy = _rowmajor(dy)
        dx = dw = None
        gq = sg = None
        if fp8 and dy.dtype == torch.bfloat16 and (xq is not None or (ctx.needs_input_grad[0] and hip.conv3x3_fp8_supported(1, B, H, W, I, O))):
            gq, sg = hip.quant_tensor_fp8(dy, e5m2=True)          # the gradient in e5m2, one dynamic scale: data and weight gradient share it
        if ctx.needs_input_grad[0]:
            wt = hip.permute021(w.reshape(1, O, I * 9), 1, O, I * 9, dy.dtype).view(I, 9 * O)             # [ci][(ky,kx)][co]
            if gq is not None and hip.conv3x3_fp8_supported(1, B, H, W, I, O):
                wq, sw = hip.quant_rows_fp8(wt)                   # transposed weights e4m3 per input channel
                dx = hip.conv3x3_fp8(1, gq, sg, wq, sw, B, H, W, I, O)
            else:
                dx = hip.conv3x3(1, dy, wt, B, H, W, I, O)
        if ctx.needs_input_grad[1]:
            if gq is not None and xq is not None:
                dwm = hip.conv3x3_fp8_wgrad(xq, sx, gq, sg, B, H, W, I, O)
            else:
                dwm = hip.conv3x3(2, x, dy, B, H, W, I, O, split_k=hip.pick_splitk_conv3x3(I, O, B * H * W))           # [O][(ky,kx)][ci] fp32
            dw = hip.permute021(dwm.view(O, 9, I), O, 9, I, torch.float32).view(wshape)
        return dx, dw, None, None, None, None


def conv3x3(x, weight, B, H, W, fp8=False):
    """bf16: implicit GEMM (fp8=True: forward and data gradient on fp8 operands where the shape takes the 256-tile kernel); fp32
    parity mode: im2col + exact-fp32 GEMM (ConvPatchFn)."""
    if x.dtype == torch.bfloat16 and weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0:
        return Conv3x3Fn.apply(x, weight, B, H, W, fp8)
    return ConvPatchFn.apply(x, weight, None, (B, H, W, weight.shape[1], 3, 1, 1), False, x.dtype)


class GeluFn(Function):
    """nn.GELU (erf) between ConvNeXt's pointwise linears (convnext.py:32,43)."""

    @staticmethod
    def forward(ctx, u):
        u = u if u.is_contiguous() else u.contiguous()
        ctx.save_for_backward(u)
        return hip.gelu_fwd(u)

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        return hip.gelu_bwd(u, dy if dy.is_contiguous() else dy.contiguous())


def gelu(u):
    return GeluFn.apply(u)


@direct_grads(1, 2, 3)
class LinearLayerScaleFn(Function):
    """x_in + drop_path( gamma * (x W^T + b) )  (convnext.py:44-49).  The layer scale is folded into the parameters
    (W' = diag(gamma) W, b' = gamma o b) so the activation is never touched; gradients for W, b and gamma follow by the
    chain rule: dW = gamma o dW', db = gamma o db', dgamma[c] = <dW'[c], W[c]> + db'[c] b[c]."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, residual, rscale, rows_per_group):
        x = _rowmajor(x)
        M, K = x.shape
        N = weight.shape[0]
        w32, b32, g32 = weight.detach(), bias.detach(), gamma.detach().contiguous()
        ws = hip.scale_rows(w32 if w32.is_contiguous() else w32.contiguous(), g32, 1)              # fp32 [N, K]
        bs = hip.scale_rows(b32.view(N, 1), g32, 1).view(N)
        wc = _w(ws, x.dtype)
        y = hip.gemm(0, x, wc, M, N, K, bias=bs, residual=residual, rscale=rscale, rows_per_group=rows_per_group or 1)
        ctx.save_for_backward(x, wc, w32, b32, g32, rscale)
        ctx.meta = (M, N, K, residual is not None, rows_per_group or 1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wc, w32, b32, g32, rscale = ctx.saved_tensors
        M, N, K, has_res, rpg = ctx.meta
        dy = _rowmajor(dy)
        dys = hip.scale_rows(dy, rscale, rpg) if rscale is not None else dy
        dx = hip.gemm(1, dys, wc, M, K, N) if ctx.needs_input_grad[0] else None
        dws, dbs = hip.gemm_dw_db(dys, x, N, K, M, split_k=_splitk(N, K, M))                         # d W', d b'
        dw = hip.scale_rows(dws, g32, 1)
        db = hip.scale_rows(dbs.view(N, 1), g32, 1).view(N)
        dg = hip.rowdot(dws, w32 if w32.is_contiguous() else w32.contiguous(), dbs, b32)
        return dx, dw, db, dg, (dy if has_res else None), None, None


def linear_layer_scale(x, weight, bias, gamma, residual=None, rscale=None, rows_per_group=None):
    return LinearLayerScaleFn.apply(x, weight, bias, gamma, residual, rscale, rows_per_group)


@direct_grads(1, 2)
class GRNFn(Function):
    """Global Response Normalization of ConvNeXtV2 (convnextv2.py:68-80) on NHWC tokens; gamma / beta are [1,1,1,C].
    pre_gelu: GRN(gelu(x)) with the activation applied inside the GRN kernels (convnextv2.py:92-94: act then grn) -- gelu(x) is
    never materialised and the backward returns the gradient of the pre-activation."""

    @staticmethod
    def forward(ctx, x, gamma, beta, B, rows_per_sample, pre_gelu=False):
        x = x if x.is_contiguous() else x.contiguous()
        g = gamma.detach().reshape(-1).contiguous()
        b = beta.detach().reshape(-1).contiguous()
        y, sq = hip.grn_fwd(x, g, b, B, rows_per_sample, pre_gelu)
        ctx.save_for_backward(x, g, sq)
        ctx.meta = (B, rows_per_sample, gamma.shape, bool(pre_gelu))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, sq = ctx.saved_tensors
        B, rps, pshape, pre_gelu = ctx.meta
        dx, dg, db = hip.grn_bwd(x, dy if dy.is_contiguous() else dy.contiguous(), g, sq, B, rps, pre_gelu)
        return dx, dg.view(pshape), db.view(pshape), None, None, None


def grn(x, gamma, beta, B, rows_per_sample, pre_gelu=False):
    return GRNFn.apply(x, gamma, beta, B, rows_per_sample, pre_gelu)


class AdaptiveAvgPoolFn(Function):
    """nn.AdaptiveAvgPool2d(S) on NHWC tokens (modules/ppm.py:13)."""

    @staticmethod
    def forward(ctx, x, B, H, W, S):
        x = x if x.is_contiguous() else x.contiguous()
        ctx.meta = (B, H, W, x.shape[1], S)
        return hip.adaptive_avgpool(x, B, H, W, x.shape[1], S)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, Cc, S = ctx.meta
        return hip.adaptive_avgpool(dy if dy.is_contiguous() else dy.contiguous(), B, H, W, Cc, S, bwd=True), None, None, None, None


def adaptive_avgpool(x, B, H, W, S):
    return AdaptiveAvgPoolFn.apply(x, B, H, W, S)


class UpsampleAddFn(Function):
    """lateral + F.interpolate(top, size=lateral.shape[-2:], bilinear, align_corners=False)  (heads/upernet.py:41)."""

    @staticmethod
    def forward(ctx, base, top, geom):
        B, H, W, h, w = geom
        base, top = _rowmajor(base), _rowmajor(top)
        ctx.meta = (geom, base.shape[1])
        return hip.upsample_add(base, [(top, h, w)], B, H, W, base.shape[1])

    @staticmethod
    def backward(ctx, dy):
        (B, H, W, h, w), Cc = ctx.meta
        dy = _rowmajor(dy)
        dtop = hip.bilinear_bwd(dy, B, h, w, Cc, H, W, align_corners=False) if ctx.needs_input_grad[1] else None
        return dy, dtop, None


def upsample_add(base, top, geom):
    return UpsampleAddFn.apply(base, top, tuple(geom))


class ResizeConcatFn(Function):
    """torch.cat([F.interpolate(f_i, size=(H, W), bilinear, align_corners=ac_i) ...], dim=1) on NHWC tokens: every branch is
    written straight into its column slice of one [B*H*W, sum C_i] buffer (modules/ppm.py:23-26, heads/upernet.py:45-49).
    A branch that already has the target size is copied (F.interpolate to the same size is the identity)."""

    @staticmethod
    def forward(ctx, geoms, aligns, size, *feats):
        B, H, W = size
        Cs = [f.shape[1] for f in feats]
        cat = torch.empty((B * H * W, sum(Cs)), dtype=feats[0].dtype, device=feats[0].device)
        off = 0
        for f, (h, w), ac, Cc in zip(feats, geoms, aligns, Cs):
            f = _rowmajor(f)
            sl = cat[:, off:off + Cc]
            if (h, w) == (H, W):
                hip.cast2d(f, sl)
            else:
                hip.bilinear_fwd(f, B, h, w, Cc, H, W, sl, align_corners=ac)
            off += Cc
        ctx.meta = (geoms, aligns, size, Cs)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        geoms, aligns, (B, H, W), Cs = ctx.meta
        dcat = _rowmajor(dcat)
        grads, off = [], 0
        for i, ((h, w), ac, Cc) in enumerate(zip(geoms, aligns, Cs)):
            sl = dcat[:, off:off + Cc]
            if not ctx.needs_input_grad[3 + i]:
                grads.append(None)
            elif (h, w) == (H, W):
                grads.append(sl)
            else:
                grads.append(hip.bilinear_bwd(sl, B, h, w, Cc, H, W, align_corners=ac))
            off += Cc
        return (None, None, None, *grads)


def resize_concat(feats, geoms, aligns, size):
    return ResizeConcatFn.apply(tuple(geoms), tuple(aligns), tuple(size), *feats)


class AddFn(Function):
    """x + y on token rows (the residual of mobilenetv2.InvertedResidual, mobilenetv2.py:33-35)."""

    @staticmethod
    def forward(ctx, a, b):
        return hip.add(_rowmajor(a), _rowmajor(b))

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return AddFn.apply(a, b)


class SubsampleFn(Function):
    """x[:, ::s, ::s, :] on NHWC tokens, rows (oy, ox) <- (s*oy, s*ox): a stride-s convolution with padding 1 and kernel 3 equals
    the stride-1 convolution sampled at every s-th position, which is how MobileNetV2's four stride-2 depthwise layers
    (mobilenetv2.py:27, ConvModule(ch, ch, 3, s, 1, g=ch)) run on the stride-1 depthwise kernel.  Forward = im2col with a 1x1
    window, backward = its col2im."""

    @staticmethod
    def forward(ctx, x, B, H, W, stride):
        x = x if x.is_contiguous() else x.contiguous()
        Cc = x.shape[1]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        ctx.meta = (B, H, W, Cc, stride, Ho, Wo)
        return hip.im2col(x, x.dtype, False, B, H, W, Cc, 1, 1, stride, 0, Ho, Wo, Cc)

    @staticmethod
    def backward(ctx, dy):
        B, H, W, Cc, stride, Ho, Wo = ctx.meta
        dy = dy if dy.is_contiguous() else dy.contiguous()
        return hip.col2im(dy, B, H, W, Cc, 1, 1, stride, 0, Ho, Wo), None, None, None, None


def subsample(x, B, H, W, stride):
    return SubsampleFn.apply(x, B, H, W, stride)


class NearestUpFn(Function):
    """F.interpolate(x, mode='nearest') by integer factors (+ base): heads/fpn.py:31 (`size=`), :34-35 (`out + lateral`, then
    `scale_factor=2.0`)."""

    @staticmethod
    def forward(ctx, x, base, geom):
        B, h, w, H, W = geom
        x = x if x.is_contiguous() else x.contiguous()
        if base is not None:
            base = base if base.is_contiguous() else base.contiguous()
        ctx.meta = (geom, x.shape[1], base is not None)
        return hip.nearest_up(x, B, h, w, x.shape[1], H, W, base=base)

    @staticmethod
    def backward(ctx, dy):
        (B, h, w, H, W), Cc, has_base = ctx.meta
        dy = dy if dy.is_contiguous() else dy.contiguous()
        dx = hip.nearest_up(dy, B, h, w, Cc, H, W, bwd=True) if ctx.needs_input_grad[0] else None
        return dx, (dy if has_base else None), None


def nearest_up(x, geom, base=None):
    """geom = (B, h, w, H, W)."""
    return NearestUpFn.apply(x, base, tuple(geom))


class UpsampleCEDiceFn(Function):
    """criterion(F.interpolate(logits, size), target): build_models.py:65 + engine.py:10-15 +
    util/losses.py:126-177, without materialising full-resolution logits in the forward."""

    @staticmethod
    def forward(ctx, logits, target, geom, ignore_index, class_weight, dice):
        B, Cc, h, w, H, W = geom
        logits = _rowmajor(logits)
        # the kernel reads `target` as int64 [B, H, W] on the device: anything else would be silently reinterpreted
        # (F.cross_entropy raises for non-Long targets, engine.py:12)
        if not target.is_cuda or target.device != logits.device:
            raise RuntimeError('criterion: target must be a device tensor on the logits\' device (no CPU fallback)')
        if target.dtype != torch.int64:
            raise RuntimeError(f'criterion: expected an int64 (Long) target, got {target.dtype}')
        if target.numel() != B * H * W:
            raise RuntimeError(f'criterion: target has {target.numel()} elements, expected B*H*W = {B * H * W}')
        if logits.shape[0] != B * h * w or logits.shape[1] < Cc:
            raise RuntimeError(f'criterion: logits {tuple(logits.shape)} do not match geometry {geom}')
        target = target.contiguous()
        # lse: the per-pixel log-sums the forward leaves for the backward (None where the configuration has no such path)
        # (grad mode is always off inside Function.forward and _rowmajor may have copied: the autograd context knows whether a backward
        # can follow)
        loss, stats, lse = hip.ce_dice_fwd(logits, B, Cc, h, w, H, W, target, ignore_index, class_weight, dice,
                                           want_lse=bool(ctx.needs_input_grad[0]))
        ctx.save_for_backward(logits, target, stats, class_weight, lse)
        ctx.meta = (geom, ignore_index, dice)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)          # (no zero-fill launches for the gradients of the two auxiliary outputs)
        return loss[0], loss.detach(), stats

    @staticmethod
    def backward(ctx, gloss, _gparts, _gstats):
        logits, target, stats, cw, lse = ctx.saved_tensors
        (B, Cc, h, w, H, W), ignore_index, dice = ctx.meta
        if gloss is None:                           # only the detached parts were used downstream
            return None, None, None, None, None, None
        go = gloss.reshape(1).to(torch.float32).contiguous()
        dl = hip.ce_dice_bwd(logits, B, Cc, h, w, H, W, target, ignore_index, cw, dice, stats, go, lse=lse)[:, :Cc]
        return dl, None, None, None, None, None


def upsample_ce_dice(logits, target, geom, ignore_index=255, class_weight=None, dice=True):
    """Returns (loss, parts[3] = {total, ce, dice_loss}, stats)."""
    return UpsampleCEDiceFn.apply(logits, target, tuple(geom), int(ignore_index), class_weight, bool(dice))


class UpsampleToNCHWFn(Function):
    """F.interpolate(head_out, size=input, bilinear, align_corners=False) -> fp32 NCHW logits
    (build_models.py:65), for callers that want the reference's materialised tensor."""

    @staticmethod
    def forward(ctx, logits, geom):
        B, Cc, h, w, H, W = geom
        logits = _rowmajor(logits)
        ctx.meta = (geom, logits.dtype, logits.stride(0))
        return hip.bilinear_to_nchw_f32(logits, B, h, w, Cc, H, W)

    @staticmethod
    def backward(ctx, dout):
        (B, Cc, h, w, H, W), dtype, ld = ctx.meta
        dn = hip.permute021(dout.contiguous(), B, Cc, H * W, dtype).view(B * H * W, Cc)   # NCHW -> NHWC
        dl = hip.bilinear_bwd(dn, B, h, w, Cc, H, W, align_corners=False, ld_in=ld)[:, :Cc]
        return dl, None


def upsample_to_nchw(logits, geom):
    return UpsampleToNCHWFn.apply(logits, tuple(geom))


class NCHWToTokensFn(Function):
    """[B, C, H, W] -> token rows [B*H*W, C] through the re-layout kernel (both directions)."""

    @staticmethod
    def forward(ctx, x):
        B, Cc, H, W = x.shape
        ctx.meta = (B, Cc, H, W, x.dtype)
        return hip.permute021(x.contiguous(), B, Cc, H * W, x.dtype).view(B * H * W, Cc)

    @staticmethod
    def backward(ctx, dt):
        B, Cc, H, W, dtype = ctx.meta
        return hip.permute021(_rowmajor(dt).contiguous(), B, H * W, Cc, dtype).view(B, Cc, H, W)


def nchw_to_tokens(x):
    return NCHWToTokensFn.apply(x)
