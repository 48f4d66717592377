"""Optimizer side of the train step: what engine.py:52-53 calls through timm 0.9.2 in the reference
(``loss_scaler(loss, optimizer, clip_grad=0.02, clip_mode='agc', parameters=..., create_graph=...)``).

timm is not available in the build image and no reference test covers this arithmetic, so AGC / AdamW are
restated from timm 0.9.2 + torch.optim.AdamW semantics ("parity unpinned") and checked against
the CPU restatement oracle/optim.py (tests/test_kernels_gpu.py::test_agc_adamw_known_answers).  The step itself is one HIP kernel over flat buffers.
"""
import torch

from . import hip


class FusedAGCAdamW(torch.optim.Optimizer):
    FLAT_SLACK = 1024        # elements of zero padding behind the flat buffers (collective ranges round up to world x 16, graph.py)
    PARAM_ALIGN = 8          # every parameter starts at a multiple of this many elements in the flat buffers (_build)

    """AdamW whose step (optionally preceded by unit-wise adaptive gradient clipping) runs as a single
    multi-tensor kernel (segf_agc_adamw).  Parameters are re-homed into one flat fp32 buffer (each
    ``p.data`` becomes a view), gradients are gathered into a flat buffer of the same layout."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = None
        self.direct = False
        self._step = 0
        self.agc_clip = 0.0          # set per step by NativeScaler when clip_mode == 'agc'

    def _build(self, order=None):
        """order: optional sequence of parameters (e.g. ``list(model.parameters())``) that fixes the LAYOUT of the flat buffers
        (registration order = reverse of the order in which backward finishes the gradients, so a suffix of the buffer is a
        bucket that completes early); default: parameter-group order.  The layout does not affect the arithmetic."""
        ps, decay = [], {}
        for g in self.param_groups:
            for p in g['params']:
                if p.requires_grad:
                    ps.append(p)
                    decay[id(p)] = g['weight_decay'] > 0
        if order is not None:
            rank = {id(p): i for i, p in enumerate(order)}
            ps.sort(key=lambda p: rank.get(id(p), len(rank)))
        dev = ps[0].device
        # every parameter starts on a PARAM_ALIGN-element boundary of the flat buffers (32 bytes in fp32, 16 bytes in the bf16 weight
        # shadow), whatever the sizes before it (a 19- or 171-class bias): the vector loads of the kernels that read weights in
        # place rely on it.  The gap elements stay zero in all four buffers and belong to no optimizer unit.
        self._spans = [-(-p.numel() // self.PARAM_ALIGN) * self.PARAM_ALIGN for p in ps]
        total = sum(self._spans)
        store = torch.zeros(total + self.FLAT_SLACK, dtype=torch.float32, device=dev)
        flat = store[:total]
        offs, lens, flags = [], [], []
        o = 0
        for p, span in zip(ps, self._spans):
            d = decay[id(p)]
            n = p.numel()
            flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = flat[o:o + n].view(p.shape)
            rows = p.shape[0] if p.ndim > 1 else 1
            cols = n // rows
            for r in range(rows):
                offs.append(o + r * cols)
                lens.append(cols)
                flags.append(1 if d else 0)
            o += span
        self._params = ps
        self._flat = flat
        self._grad_store = torch.zeros(total + self.FLAT_SLACK, dtype=torch.float32, device=dev)
        self._grad = self._grad_store[:total]
        self._grad_views, self._offsets, o = [], [], 0
        for p, span in zip(ps, self._spans):
            self._grad_views.append(self._grad[o:o + p.numel()].view(p.shape))
            self._offsets.append(o)
            o += span
        self._m = torch.zeros_like(flat)
        self._v = torch.zeros_like(flat)
        self._off = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._len = torch.tensor(lens, dtype=torch.int32, device=dev)
        self._flags = torch.tensor(flags, dtype=torch.uint8, device=dev)
        # units (rows) of every parameter, for the per-step "received no gradient" bit (flag bit 1): torch.optim.AdamW -- the
        # reference's optimizer, train_gpu.py:269 -- skips a parameter whose .grad is None entirely (no decay, no moment update)
        self._unit_range, u = [], 0
        for p in ps:
            rows = p.shape[0] if p.ndim > 1 else 1
            self._unit_range.append((u, u + rows))
            u += rows
        self._base_flags = list(flags)
        self._nograd = frozenset()          # indices (into self._params) of parameters skipped by the current flags
        self._ustep = torch.zeros(len(flags), dtype=torch.int32, device=dev)   # per-unit step counts (torch's per-parameter state['step'])
        self._written = {}                  # direct placement: gradient-slot pointer -> deliveries in the current backward
        self.direct = False

    def _set_nograd(self, idx):
        """Mark the parameters `idx` (indices into the flat layout) as 'no gradient this step': the kernel leaves their
        parameters and moments untouched.  The flag tensor is rewritten only when the set changes."""
        idx = frozenset(idx)
        if idx == self._nograd:
            return
        flags = list(self._base_flags)
        for i in idx:
            lo, hi = self._unit_range[i]
            for u in range(lo, hi):
                flags[u] |= 2
        self._flags.copy_(torch.tensor(flags, dtype=torch.uint8), non_blocking=False)
        self._nograd = idx

    def set_clipping(self, clip_grad, clip_mode):
        """timm.utils.dispatch_clip_grad(parameters, value=clip_grad, mode=clip_mode) as part of the optimizer step: 'agc' inside
        the AdamW kernel, 'norm' / 'value' as kernels over the flat gradient buffer right before it (segf_clip_grad)."""
        if clip_grad is not None and clip_mode not in ('agc', 'norm', 'value'):
            raise AssertionError(f"Unknown clip mode ({clip_mode}).")          # timm's wording
        self.clip_mode = clip_mode if clip_grad is not None else 'agc'
        self.clip_value = None if clip_grad is None else float(clip_grad)
        self.agc_clip = float(clip_grad) if (clip_grad is not None and clip_mode == 'agc') else 0.0

    def ensure_built(self, order=None):
        if self._flat is None:
            self._build(order)

    @property
    def flat_params(self):
        self.ensure_built()
        return self._flat

    @property
    def flat_grads(self):
        self.ensure_built()
        return self._grad

    @property
    def flat_grads_padded(self):
        """The gradient buffer including its FLAT_SLACK zero elements (collectives run over aligned ranges of this)."""
        self.ensure_built()
        return self._grad_store

    def enable_direct_grads(self, callback=None):
        """Hand every parameter its view of the flat gradient buffer (``p._segf_grad``): the backward formulas of
        segmentation_factory_amd.functional then write parameter gradients in place and ``.grad`` stays None
        (functional.direct_grads).  callback(view) is invoked, in backward order, whenever one gradient is final."""
        self.ensure_built()
        self._grad.zero_()
        self._user_cb = callback
        for p, view in zip(self._params, self._grad_views):
            p._segf_grad = view
            p._segf_grad_cb = self._note_delivery
            p.grad = None
        self.direct = True

    def begin_backward(self):
        """Direct placement: call before every forward+backward that is run from Python (warm-up, capture, eager steps)."""
        self._written = {}

    def _note_delivery(self, view):
        # a slot is an assignment target: a parameter consumed by two Functions in one step (tied weights, a module applied
        # twice) would keep only the last contribution and release its bucket early -- refuse instead of training on it
        k = view.data_ptr()
        if k in self._written:
            raise RuntimeError('direct gradient placement: a parameter received two gradients in one backward (tied weights / '
                               'a module applied twice); use the eager path (plain autograd accumulation) for such a model')
        self._written[k] = 1
        if self._user_cb is not None:
            self._user_cb(view)

    def finish_backward(self):
        """Direct placement: parameters that received neither an in-place gradient nor a .grad in the backward that just ran
        (e.g. FPNHead.output_convs[0], quirk Q3; frozen-by-construction plugin parts) are skipped by the optimizer kernel."""
        self._set_nograd(i for i, (p, v) in enumerate(zip(self._params, self._grad_views))
                         if v.data_ptr() not in self._written and p.grad is None)

    def disable_direct_grads(self):
        for p in getattr(self, '_params', []):
            for a in ('_segf_grad', '_segf_grad_cb'):
                if hasattr(p, a):
                    delattr(p, a)
        self.direct = False

    @torch.no_grad()
    def gather_grads(self):
        """Copy every parameter's .grad into the flat gradient buffer (multi-tensor copy; graph-capturable).  With direct
        placement only gradients that still arrived as ``.grad`` (foreign plugin modules on plain autograd) are copied."""
        self.ensure_built()
        dst, src, missing = [], [], []
        for i, (p, view) in enumerate(zip(self._params, self._grad_views)):
            if p.grad is not None:
                dst.append(view)
                src.append(p.grad)
            elif not self.direct:
                missing.append(i)
        if dst:
            torch._foreach_copy_(dst, src)
        if not self.direct:
            self._set_nograd(missing)          # torch.optim.AdamW: `if p.grad is None: continue`
            if missing and getattr(self, 'clip_mode', 'agc') in ('norm', 'value'):
                # clip_grad_norm_ / clip_grad_value_ ignore parameters without a gradient; the flat-buffer kernels see every slot, so a
                # gradient left over from an earlier step must not enter the global norm
                torch._foreach_zero_([self._grad_views[i] for i in missing])

    @torch.no_grad()
    def apply_flat(self):
        """AGC + AdamW over the flat buffers: one kernel launch (bias corrections are host scalars of this step)."""
        g = self.param_groups[0]
        # one kernel over the flat buffer: lr / betas / eps are launch scalars and weight decay is an on/off flag per unit, so
        # every group must agree on them (timm's param_groups_weight_decay gives exactly {0, wd}); anything else would be
        # silently ignored
        for pg in self.param_groups[1:]:
            if (pg['lr'], tuple(pg['betas']), pg['eps']) != (g['lr'], tuple(g['betas']), g['eps']):
                raise NotImplementedError('FusedAGCAdamW: all parameter groups must share lr / betas / eps')
        wds = {pg['weight_decay'] for pg in self.param_groups if pg['weight_decay'] > 0}
        if len(wds) > 1:
            raise NotImplementedError(f'FusedAGCAdamW: one non-zero weight decay for all decayed groups, got {sorted(wds)}')
        wd = max(pg['weight_decay'] for pg in self.param_groups)
        self._step += 1
        mode, value = getattr(self, 'clip_mode', 'agc'), getattr(self, 'clip_value', None)
        if mode in ('norm', 'value') and value is not None:      # timm dispatch_clip_grad's other modes, on the flat buffer
            if getattr(self, '_clip_ws', None) is None:
                self._clip_ws = torch.empty(int(hip.lib().segf_clip_grad_ws()), dtype=torch.float32, device=self._grad.device)
            hip.clip_grad(self._grad, mode, value, self._clip_ws)
        hip.agc_adamw(self._flat, self._grad, self._m, self._v, self._off, self._len, self._flags, g['lr'], g['betas'][0],
                      g['betas'][1], g['eps'], wd, self._step, float(self.agc_clip) if mode == 'agc' else 0.0, unit_step=self._ustep)

    @torch.no_grad()
    def step(self, closure=None):
        self.gather_grads()
        self.apply_flat()

    def _packed_ids(self):
        ids, i = {}, 0
        for g in self.param_groups:
            for p in g['params']:
                ids[id(p)] = i
                i += 1
        return ids

    def state_dict(self):
        """torch.optim.AdamW's state_dict layout (per-parameter 'step' / 'exp_avg' / 'exp_avg_sq' under packed indices), so the
        checkpoint's 'optimizer_state' (train_gpu.py:354-362) can be read back by torch.optim.AdamW / timm's AdamW and vice versa."""
        sd = super().state_dict()
        if self._flat is not None and self._step > 0:
            ids = self._packed_ids()
            state = {}
            usteps = self._ustep.cpu().tolist()
            for i, p in enumerate(self._params):
                n, o = p.numel(), self._offsets[i]
                t = usteps[self._unit_range[i][0]]
                if t > 0:                       # torch lists state only for parameters that have been stepped
                    state[ids[id(p)]] = {'step': torch.tensor(float(t)),
                                         'exp_avg': self._m[o:o + n].view(p.shape).detach().cpu().clone(),
                                         'exp_avg_sq': self._v[o:o + n].view(p.shape).detach().cpu().clone()}
            sd['state'] = state
        return sd

    def load_state_dict(self, sd):
        """Accepts this class's own state_dict and a torch.optim.AdamW / timm AdamW one (the reference's checkpoints)."""
        sd = dict(sd)
        fused = sd.pop('fused', None)          # round-1 layout of this class
        state = sd.get('state', {}) or {}
        super().load_state_dict({'state': {}, 'param_groups': sd['param_groups']})
        if fused and fused.get('exp_avg') is not None:
            self.ensure_built()
            self._step = int(fused['step'])
            self._ustep.fill_(self._step)
            # the round-1 layout was UNPADDED (parameters back to back); the flat buffers now align every parameter: scatter by offset
            m_old, v_old = fused['exp_avg'].reshape(-1), fused['exp_avg_sq'].reshape(-1)
            total = sum(p.numel() for p in self._params)
            if m_old.numel() == self._m.numel():               # written by a build with the same (padded) layout
                self._m.copy_(m_old)
                self._v.copy_(v_old)
            elif m_old.numel() == total:
                pos = 0
                for i, p in enumerate(self._params):
                    n, o = p.numel(), self._offsets[i]
                    self._m[o:o + n].copy_(m_old[pos:pos + n])
                    self._v[o:o + n].copy_(v_old[pos:pos + n])
                    pos += n
            else:
                raise ValueError(f"FusedAGCAdamW.load_state_dict: legacy 'fused' moments hold {m_old.numel()} values, the model has "
                                 f'{total} parameters ({self._m.numel()} with alignment padding)')
            return
        if not state:
            return
        self.ensure_built()
        by_index = {}
        for g in self.param_groups:
            for p in g['params']:
                by_index[len(by_index)] = p
        offs, pos = {}, {}
        for i, p in enumerate(self._params):
            offs[id(p)] = self._offsets[i]
            pos[id(p)] = i
        for idx, st in state.items():
            p = by_index[int(idx)]
            if id(p) not in offs:
                continue                        # frozen parameter: not part of the flat buffers
            o, n = offs[id(p)], p.numel()
            lo, hi = self._unit_range[pos[id(p)]]
            self._ustep[lo:hi] = int(float(st['step']))
            self._m[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self._v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            self._step = max(self._step, int(float(st['step'])))


class NativeScaler:
    """Call-compatible stand-in for timm.utils.NativeScaler.  bf16 needs no loss scaling, so this is:
    backward -> (optional) clipping -> optimizer.step().  ``state_dict`` keeps the checkpoint key 'scaler'."""
    state_dict_key = 'amp_scaler'

    def __call__(self, loss, optimizer, clip_grad=None, clip_mode='norm', parameters=None, create_graph=False,
                 need_update=True):
        loss.backward(create_graph=create_graph)
        if not need_update:
            return
        if isinstance(optimizer, FusedAGCAdamW):
            optimizer.set_clipping(clip_grad, clip_mode)
        elif clip_grad is not None:
            raise NotImplementedError('gradient clipping is fused into FusedAGCAdamW; use it or pass clip_grad=None')
        optimizer.step()

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


def param_groups_weight_decay(model, weight_decay):
    """timm.optim.optim_factory.param_groups_weight_decay: no decay for 1-D tensors and biases."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or name.endswith('.bias')) else decay).append(p)
    return [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': weight_decay}]


def create_optimizer(args, model):
    """--opt adamw path of timm.optim.create_optimizer (train_gpu.py:269) on the fused kernel."""
    if getattr(args, 'opt', 'adamw').lower() != 'adamw':
        raise NotImplementedError("only --opt adamw is implemented on the MI355X path")
    wd = getattr(args, 'weight_decay', 0.025)
    kw = dict(lr=getattr(args, 'lr', 1e-3), weight_decay=wd, eps=getattr(args, 'opt_eps', None) or 1e-8)
    betas = getattr(args, 'opt_betas', None)
    if betas:
        kw['betas'] = tuple(betas)
    return FusedAGCAdamW(param_groups_weight_decay(model, wd), **kw)
