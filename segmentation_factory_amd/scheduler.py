"""Learning-rate schedules behind `create_scheduler(args, optimizer)` (scheduler/scheduler_factory.py:12-110).

Host-side scalar arithmetic, once per epoch / update -- outside the accelerated path.  Every
`--sched` value of the reference is restated: cosine, tanh, step, multistep, plateau, poly, with linear warm-up, restart cycles,
k-decay and LR noise, the reference's time units (for cosine t_initial / warmup_t are counted in ITERATIONS,
scheduler_factory.py:14-16,41-51) and its default inertness (quirk Q9: cosine is built with `t_in_epochs=args.lr_ep`,
train_gpu.py:336 only ever calls `step(epoch)`, so without `--lr-ep` the rate stays at the warm-up start value the constructor
writes into the optimizer).  Pinned by tests/golden/scheduler_cases.json (LR sequences captured from the reference's classes).

One class hierarchy instead of the reference's six near-identical files: `Schedule` owns warm-up, noise, the two clocks
(epochs / updates) and the state dict; a subclass only states its decay curve.  Attribute names follow the reference
(scheduler/scheduler_main.py:24-57) because `state_dict()` -- every attribute but the optimizer -- is the checkpoint's
'scheduler_state' (train_gpu.py:354-362).
"""
import bisect
import math

import torch


class Schedule:
    """Warm-up + decay curve + optional noise on one optimizer field, per parameter group (scheduler_main.py:6-116)."""

    def __init__(self, optimizer, warmup_t=0, warmup_lr_init=0., t_in_epochs=True, noise_range_t=None, noise_type='normal',
                 noise_pct=0.67, noise_std=1.0, noise_seed=None, param_group_field='lr'):
        self.optimizer = optimizer
        self.param_group_field = param_group_field
        self._initial_param_group_field = f'initial_{param_group_field}'
        for i, group in enumerate(optimizer.param_groups):
            if param_group_field not in group:
                raise KeyError(f'{param_group_field} missing from param_groups[{i}]')
            group.setdefault(self._initial_param_group_field, group[param_group_field])
        self.base_values = [g[self._initial_param_group_field] for g in optimizer.param_groups]
        self.metric = None
        self.noise_range_t, self.noise_pct, self.noise_type, self.noise_std = noise_range_t, noise_pct, noise_type, noise_std
        self.noise_seed = noise_seed if noise_seed is not None else 42
        self.update_groups(self.base_values)
        self.warmup_t, self.warmup_lr_init, self.t_in_epochs = warmup_t, warmup_lr_init, t_in_epochs

    def _init_warmup(self, targets=None):
        """Linear ramp from warmup_lr_init to `targets` (default: the base values) over warmup_t ticks; the constructor writes the
        start value into the optimizer (cosine_lr.py:66-68)."""
        if self.warmup_t:
            self.warmup_steps = [(v - self.warmup_lr_init) / self.warmup_t for v in (targets or self.base_values)]
            self.update_groups(self.warmup_lr_init)
        else:
            self.warmup_steps = [1 for _ in self.base_values]

    # ---- the decay curve (after warm-up) ------------------------------------------------------------------------------
    def curve(self, t):
        raise NotImplementedError

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        return self.curve(t)

    # ---- the two clocks ------------------------------------------------------------------------------------------------
    def get_epoch_values(self, epoch):
        return self._get_lr(epoch) if self.t_in_epochs else None

    def get_update_values(self, num_updates):
        return self._get_lr(num_updates) if not self.t_in_epochs else None

    def step(self, epoch, metric=None):
        self.metric = metric
        self._apply(self.get_epoch_values(epoch), epoch)

    def step_update(self, num_updates, metric=None):
        self.metric = metric
        self._apply(self.get_update_values(num_updates), num_updates)

    def _apply(self, values, t):
        if values is None:
            return
        if self._is_apply_noise(t):
            noise = self._calculate_noise(t)
            values = [v + v * noise for v in values]
        self.update_groups(values)

    def update_groups(self, values):
        if not isinstance(values, (list, tuple)):
            values = [values] * len(self.optimizer.param_groups)
        for group, value in zip(self.optimizer.param_groups, values):
            group[self.param_group_field] = value * group['lr_scale'] if 'lr_scale' in group else value

    # ---- noise (scheduler_main.py:88-116) ------------------------------------------------------------------------------
    def _is_apply_noise(self, t):
        r = self.noise_range_t
        if r is None:
            return False
        return r[0] <= t < r[1] if isinstance(r, (list, tuple)) else t >= r

    def _calculate_noise(self, t):
        g = torch.Generator()
        g.manual_seed(self.noise_seed + t)
        if self.noise_type == 'normal':
            while True:                                     # resample until inside the percent limit
                noise = torch.randn(1, generator=g).item()
                if abs(noise) < self.noise_pct:
                    return noise
        return 2 * (torch.rand(1, generator=g).item() - 0.5) * self.noise_pct

    # ---- checkpoint 'scheduler_state' ---------------------------------------------------------------------------------
    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class _Cyclic(Schedule):
    """Schedules with restart cycles: t -> (cycle index i, cycle length t_i, position t_curr) (cosine_lr.py:74-84)."""

    def __init__(self, optimizer, t_initial, lr_min=0., cycle_mul=1., cycle_decay=1., cycle_limit=1, warmup_prefix=False, **kw):
        super().__init__(optimizer, **kw)
        assert t_initial > 0 and lr_min >= 0
        self.t_initial, self.lr_min = t_initial, lr_min
        self.cycle_mul, self.cycle_decay, self.cycle_limit, self.warmup_prefix = cycle_mul, cycle_decay, cycle_limit, warmup_prefix

    def shape(self, lr_max, t_curr, t_i):
        raise NotImplementedError

    def curve(self, t):
        if self.warmup_prefix:
            t = t - self.warmup_t
        if self.cycle_mul != 1:
            i = math.floor(math.log(1 - t / self.t_initial * (1 - self.cycle_mul), self.cycle_mul))
            t_i = self.cycle_mul ** i * self.t_initial
            t_curr = t - (1 - self.cycle_mul ** i) / (1 - self.cycle_mul) * self.t_initial
        else:
            i = t // self.t_initial
            t_i = self.t_initial
            t_curr = t - self.t_initial * i
        if i >= self.cycle_limit:
            return [self.lr_min for _ in self.base_values]
        gamma = self.cycle_decay ** i
        return [self.shape(v * gamma, t_curr, t_i) for v in self.base_values]

    def get_cycle_length(self, cycles=0):
        cycles = max(1, cycles or self.cycle_limit)
        if self.cycle_mul == 1.0:
            return self.t_initial * cycles
        return int(math.floor(-self.t_initial * (self.cycle_mul ** cycles - 1) / (1 - self.cycle_mul)))


class CosineLRScheduler(_Cyclic):
    """scheduler/cosine_lr.py:17-118 (SGDR cosine with restarts, k-decay)."""

    def __init__(self, optimizer, t_initial, lr_min=0., cycle_mul=1., cycle_decay=1., cycle_limit=1, warmup_t=0, warmup_lr_init=0,
                 warmup_prefix=False, t_in_epochs=True, noise_range_t=None, noise_pct=0.67, noise_std=1.0, noise_seed=42,
                 k_decay=1.0):
        super().__init__(optimizer, t_initial, lr_min, cycle_mul, cycle_decay, cycle_limit, warmup_prefix, warmup_t=warmup_t,
                         warmup_lr_init=warmup_lr_init, t_in_epochs=t_in_epochs, noise_range_t=noise_range_t, noise_pct=noise_pct,
                         noise_std=noise_std, noise_seed=noise_seed)
        self.k_decay = k_decay
        self._init_warmup()

    def shape(self, lr_max, t_curr, t_i):
        k = self.k_decay
        return self.lr_min + 0.5 * (lr_max - self.lr_min) * (1 + math.cos(math.pi * t_curr ** k / t_i ** k))


class PolyLRScheduler(_Cyclic):
    """scheduler/poly_lr.py:17-115."""

    def __init__(self, optimizer, t_initial, power=0.5, lr_min=0., cycle_mul=1., cycle_decay=1., cycle_limit=1, warmup_t=0,
                 warmup_lr_init=0, warmup_prefix=False, t_in_epochs=True, noise_range_t=None, noise_pct=0.67, noise_std=1.0,
                 noise_seed=42, k_decay=1.0):
        super().__init__(optimizer, t_initial, lr_min, cycle_mul, cycle_decay, cycle_limit, warmup_prefix, warmup_t=warmup_t,
                         warmup_lr_init=warmup_lr_init, t_in_epochs=t_in_epochs, noise_range_t=noise_range_t, noise_pct=noise_pct,
                         noise_std=noise_std, noise_seed=noise_seed)
        self.power, self.k_decay = power, k_decay
        self._init_warmup()

    def shape(self, lr_max, t_curr, t_i):
        k = self.k_decay
        return self.lr_min + (lr_max - self.lr_min) * (1 - t_curr ** k / t_i ** k) ** self.power


class TanhLRScheduler(_Cyclic):
    """scheduler/tanh_lr.py:17-116.  Without warmup_prefix the warm-up ramps to the curve's value AT warmup_t (:66-68)."""

    def __init__(self, optimizer, t_initial, lb=-7., ub=3., lr_min=0., cycle_mul=1., cycle_decay=1., cycle_limit=1, warmup_t=0,
                 warmup_lr_init=0, warmup_prefix=False, t_in_epochs=True, noise_range_t=None, noise_pct=0.67, noise_std=1.0,
                 noise_seed=42):
        # warm-up fields are set after the curve can be evaluated (the ramp target needs it)
        super().__init__(optimizer, t_initial, lr_min, cycle_mul, cycle_decay, cycle_limit, warmup_prefix, warmup_t=0,
                         warmup_lr_init=warmup_lr_init, t_in_epochs=t_in_epochs, noise_range_t=noise_range_t, noise_pct=noise_pct,
                         noise_std=noise_std, noise_seed=noise_seed)
        assert lb < ub and cycle_limit >= 0 and warmup_t >= 0 and warmup_lr_init >= 0
        self.lb, self.ub = lb, ub
        targets = None
        if warmup_t and not warmup_prefix:
            targets = self.curve(warmup_t)
        self.warmup_t = warmup_t
        self._init_warmup(targets)

    def shape(self, lr_max, t_curr, t_i):
        tr = t_curr / t_i
        return self.lr_min + 0.5 * (lr_max - self.lr_min) * (1 - math.tanh(self.lb * (1. - tr) + self.ub * tr))


class StepLRScheduler(Schedule):
    """scheduler/step_lr.py:13-62."""

    def __init__(self, optimizer, decay_t, decay_rate=1., warmup_t=0, warmup_lr_init=0, t_in_epochs=True, noise_range_t=None,
                 noise_pct=0.67, noise_std=1.0, noise_seed=42):
        super().__init__(optimizer, warmup_t=warmup_t, warmup_lr_init=warmup_lr_init, t_in_epochs=t_in_epochs,
                         noise_range_t=noise_range_t, noise_pct=noise_pct, noise_std=noise_std, noise_seed=noise_seed)
        self.decay_t, self.decay_rate = decay_t, decay_rate
        self._init_warmup()

    def curve(self, t):
        return [v * (self.decay_rate ** (t // self.decay_t)) for v in self.base_values]


class MultiStepLRScheduler(Schedule):
    """scheduler/multistep_lr.py:10-65 (milestones compared against t + 1, :44-47)."""

    def __init__(self, optimizer, decay_t, decay_rate=1., warmup_t=0, warmup_lr_init=0, t_in_epochs=True, noise_range_t=None,
                 noise_pct=0.67, noise_std=1.0, noise_seed=42):
        super().__init__(optimizer, warmup_t=warmup_t, warmup_lr_init=warmup_lr_init, t_in_epochs=t_in_epochs,
                         noise_range_t=noise_range_t, noise_pct=noise_pct, noise_std=noise_std, noise_seed=noise_seed)
        self.decay_t, self.decay_rate = decay_t, decay_rate
        self._init_warmup()

    def curve(self, t):
        n = bisect.bisect_right(self.decay_t, t + 1)
        return [v * (self.decay_rate ** n) for v in self.base_values]


class PlateauLRScheduler(Schedule):
    """scheduler/plateau_lr.py:10-102: torch's ReduceLROnPlateau behind the warm-up / noise wrapper.  Its state_dict is
    {'best', 'last_epoch'} (:64-73), not the attribute dump."""

    def __init__(self, optimizer, decay_rate=0.1, patience_t=10, verbose=True, threshold=1e-4, cooldown_t=0, warmup_t=0,
                 warmup_lr_init=0, lr_min=0, mode='max', noise_range_t=None, noise_type='normal', noise_pct=0.67, noise_std=1.0,
                 noise_seed=None):
        super().__init__(optimizer, warmup_t=warmup_t, warmup_lr_init=warmup_lr_init, noise_range_t=noise_range_t,
                         noise_type=noise_type, noise_pct=noise_pct, noise_std=noise_std, noise_seed=noise_seed)
        self.lr_scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, patience=patience_t, factor=decay_rate,
                                                                       threshold=threshold, cooldown=cooldown_t, mode=mode,
                                                                       min_lr=lr_min)
        self._init_warmup()
        self.restore_lr = None

    def state_dict(self):
        return {'best': self.lr_scheduler.best, 'last_epoch': self.lr_scheduler.last_epoch}

    def load_state_dict(self, sd):
        self.lr_scheduler.best = sd['best']
        if 'last_epoch' in sd:
            self.lr_scheduler.last_epoch = sd['last_epoch']

    def step(self, epoch, metric=None):
        if epoch <= self.warmup_t:
            self.update_groups([self.warmup_lr_init + epoch * s for s in self.warmup_steps])
            return
        if self.restore_lr is not None:                      # undo last epoch's noise before the plateau logic looks at the rate
            for group, lr in zip(self.optimizer.param_groups, self.restore_lr):
                group['lr'] = lr
            self.restore_lr = None
        self.lr_scheduler.step(metric, epoch)
        if self._is_apply_noise(epoch):
            noise = self._calculate_noise(epoch)
            self.restore_lr = [float(g['lr']) for g in self.optimizer.param_groups]
            for group, old in zip(self.optimizer.param_groups, self.restore_lr):
                group['lr'] = old + old * noise

    def step_update(self, num_updates, metric=None):
        pass


def create_scheduler(args, optimizer):
    """scheduler/scheduler_factory.py:12-110: (scheduler, number of epochs incl. cool-down)."""
    num_epochs = args.epochs
    n_iter = max(args.data_len // (args.batch_size * max(getattr(args, 'world_size', 1), 1)), 1)
    tot_iter, warmup_iters = num_epochs * n_iter, args.warmup_epochs * n_iter
    lr_noise = getattr(args, 'lr_noise', None)
    if lr_noise is not None:
        if isinstance(lr_noise, (list, tuple)):
            noise_range = [n * num_epochs for n in lr_noise]
            if len(noise_range) == 1:
                noise_range = noise_range[0]
        else:
            noise_range = lr_noise * num_epochs
    else:
        noise_range = None
    noise = dict(noise_range_t=noise_range, noise_pct=getattr(args, 'lr_noise_pct', 0.67), noise_std=getattr(args, 'lr_noise_std', 1.),
                 noise_seed=getattr(args, 'seed', 42))
    cycle = dict(cycle_mul=getattr(args, 'lr_cycle_mul', 1.), cycle_decay=getattr(args, 'lr_cycle_decay', 0.1),
                 cycle_limit=getattr(args, 'lr_cycle_limit', 1))
    sched = getattr(args, 'sched', 'cosine')
    if sched == 'cosine':      # the only schedule counted in iterations, and only live with --lr-ep (quirk Q9)
        s = CosineLRScheduler(optimizer, t_initial=tot_iter, lr_min=args.min_lr, warmup_lr_init=args.warmup_lr, warmup_t=warmup_iters,
                              k_decay=getattr(args, 'lr_k_decay', 1.0), t_in_epochs=args.lr_ep, **cycle, **noise)
        return s, s.get_cycle_length() // n_iter + args.cooldown_epochs
    if sched == 'tanh':
        s = TanhLRScheduler(optimizer, t_initial=num_epochs, lr_min=args.min_lr, warmup_lr_init=args.warmup_lr,
                            warmup_t=args.warmup_epochs, t_in_epochs=True, **cycle, **noise)
        return s, s.get_cycle_length() + args.cooldown_epochs
    if sched == 'step':
        return StepLRScheduler(optimizer, decay_t=args.decay_epochs, decay_rate=args.decay_rate, warmup_lr_init=args.warmup_lr,
                               warmup_t=args.warmup_epochs, **noise), num_epochs
    if sched == 'multistep':
        return MultiStepLRScheduler(optimizer, decay_t=args.decay_milestones, decay_rate=args.decay_rate,
                                    warmup_lr_init=args.warmup_lr, warmup_t=args.warmup_epochs, **noise), num_epochs
    if sched == 'plateau':
        mode = 'min' if 'loss' in getattr(args, 'eval_metric', '') else 'max'
        return PlateauLRScheduler(optimizer, decay_rate=args.decay_rate, patience_t=args.patience_epochs, lr_min=args.min_lr,
                                  mode=mode, warmup_lr_init=args.warmup_lr, warmup_t=args.warmup_epochs, cooldown_t=0,
                                  **noise), num_epochs
    if sched == 'poly':
        s = PolyLRScheduler(optimizer, power=args.decay_rate, t_initial=num_epochs, lr_min=args.min_lr, warmup_lr_init=args.warmup_lr,
                            warmup_t=args.warmup_epochs, k_decay=getattr(args, 'lr_k_decay', 1.0), **cycle, **noise)
        return s, s.get_cycle_length() + args.cooldown_epochs
    return None, num_epochs             # the reference returns (None, epochs) for an unknown name (and then fails on .step)
