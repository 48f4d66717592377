"""LR schedule behind `create_scheduler(args, optimizer)` (scheduler/scheduler_factory.py:12-110).

Host-side scalar arithmetic, once per epoch -- outside the accelerated path.  Restated is
what the training loop can observe: the cosine schedule with linear warm-up, with the reference's time units
(t_initial / warmup_t are counted in ITERATIONS, scheduler_factory.py:14-16,41-51) and its default inertness (quirk Q9):
the scheduler is built with `t_in_epochs=args.lr_ep`, train_gpu.py:336 only ever calls `step(epoch)`, so unless `--lr-ep` is
given the learning rate stays at the warm-up start value that the constructor writes into the optimizer.
Other `--sched` values fall back to a constant rate and say so.
"""
import math


class CosineLRScheduler:
    def __init__(self, optimizer, t_initial, lr_min=0., warmup_t=0, warmup_lr_init=0., t_in_epochs=True, cycle_mul=1.,
                 cycle_decay=1., cycle_limit=1, k_decay=1.0):
        self.optimizer = optimizer
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_values = [g['initial_lr'] for g in optimizer.param_groups]
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = max(int(t_initial), 1), lr_min, warmup_t, warmup_lr_init
        self.t_in_epochs, self.cycle_mul, self.cycle_decay, self.cycle_limit, self.k_decay = t_in_epochs, cycle_mul, cycle_decay, cycle_limit, k_decay
        if self.warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / self.warmup_t for v in self.base_values]
            self._update([self.warmup_lr_init] * len(self.base_values))      # cosine_lr.py:66-68
        else:
            self.warmup_steps = [1 for _ in self.base_values]

    def _update(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g['lr'] = v

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        if self.cycle_mul != 1:
            i = math.floor(math.log(1 - t / self.t_initial * (1 - self.cycle_mul), self.cycle_mul))
            t_i = self.cycle_mul ** i * self.t_initial
            t_curr = t - (1 - self.cycle_mul ** i) / (1 - self.cycle_mul) * self.t_initial
        else:
            i = t // self.t_initial
            t_i = self.t_initial
            t_curr = t - (self.t_initial * i)
        gamma = self.cycle_decay ** i
        if i < self.cycle_limit:
            k = self.k_decay
            return [self.lr_min + 0.5 * (v * gamma - self.lr_min) * (1 + math.cos(math.pi * t_curr ** k / t_i ** k)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def step(self, epoch, metric=None):
        if self.t_in_epochs:                      # cosine_lr.py:102-106; otherwise step(epoch) is a no-op (quirk Q9)
            self._update(self._get_lr(epoch))

    def step_update(self, num_updates, metric=None):
        if not self.t_in_epochs:
            self._update(self._get_lr(num_updates))

    def get_cycle_length(self, cycles=0):
        cycles = max(1, cycles or self.cycle_limit)
        if self.cycle_mul == 1.0:
            return self.t_initial * cycles
        return int(math.floor(-self.t_initial * (self.cycle_mul ** cycles - 1) / (1 - self.cycle_mul)))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class ConstantLR:
    def __init__(self, optimizer):
        self.optimizer = optimizer

    def step(self, epoch, metric=None):
        pass

    def step_update(self, num_updates, metric=None):
        pass

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


def create_scheduler(args, optimizer):
    num_epochs = args.epochs
    n_iter = max(args.data_len // (args.batch_size * max(getattr(args, 'world_size', 1), 1)), 1)
    if getattr(args, 'sched', 'cosine') == 'cosine':
        sch = CosineLRScheduler(optimizer, t_initial=num_epochs * n_iter, lr_min=args.min_lr, warmup_lr_init=args.warmup_lr,
                                warmup_t=args.warmup_epochs * n_iter, k_decay=getattr(args, 'lr_k_decay', 1.0),
                                t_in_epochs=args.lr_ep, cycle_mul=getattr(args, 'lr_cycle_mul', 1.),
                                cycle_decay=getattr(args, 'lr_cycle_decay', 0.1), cycle_limit=getattr(args, 'lr_cycle_limit', 1))
        return sch, sch.get_cycle_length() // n_iter + args.cooldown_epochs
    print(f"--sched {args.sched}: only 'cosine' is restated on the MI355X path; keeping a constant learning rate")
    return ConstantLR(optimizer), num_epochs
