"""Host-side helpers with the reference's interfaces (util/utils.py): SmoothedValue / MetricLogger
(:32-232), ConfusionMatrix (:94-143), distributed helpers (:243-310), checkpoint helpers (:313-331).
"""
import datetime
import os
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist

from . import hip


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


class SmoothedValue:
    """Windowed + global statistics of a scalar series (util/utils.py:32-91)."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total, self.count = 0.0, 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """All-reduce count/total (collective C6 of SURVEY.md section 2.3); the window is left local."""
        if not is_dist_avail_and_initialized():
            return
        dev = 'cuda' if torch.cuda.is_available() else 'cpu'
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


class MetricLogger:
    """Console meter with the reference's line format (util/utils.py:146-232)."""

    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("{}: {}".format(name, str(meter)) for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None):
        header = header or ''
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt='{avg:.4f}'), SmoothedValue(fmt='{avg:.4f}')
        n = len(iterable)
        fields = [header, '[{0:' + str(len(str(n))) + 'd}/{1}]', 'eta: {eta}', '{meters}', 'time: {time}', 'data: {data}']
        cuda = torch.cuda.is_available()
        if cuda:
            fields.append('max mem: {memory:.0f}')
        msg = self.delimiter.join(fields)
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                kw = dict(eta=eta, meters=str(self), time=str(iter_time), data=str(data_time))
                if cuda:
                    kw['memory'] = torch.cuda.max_memory_allocated() / (1024.0 * 1024.0)
                print(msg.format(i, n, **kw))
            end = time.time()
        print('{} Total time: {}'.format(header, str(datetime.timedelta(seconds=int(time.time() - start)))))


class ConfusionMatrix:
    """int64 [n, n] confusion matrix, rows = ground truth (util/utils.py:94-143)."""

    def __init__(self, num_classes):
        self.num_classes = num_classes
        self.mat = None

    def _ensure(self, device):
        if self.mat is None:
            self.mat = torch.zeros((self.num_classes, self.num_classes), dtype=torch.int64, device=device)

    def update(self, a, b):
        """a: flat ground truth, b: flat predictions (int64).  Counted on device by segf_confmat_pairs."""
        self._ensure(a.device)
        flag = torch.zeros(1, dtype=torch.int32, device=a.device)
        hip.confmat_pairs(a.contiguous().to(torch.int64), b.contiguous().to(torch.int64), self.num_classes, -1,
                          self.mat, None, flag)

    def reset(self):
        if self.mat is not None:
            self.mat.zero_()

    def compute(self):
        h = self.mat.float()
        acc_global = torch.diag(h).sum() / h.sum()
        acc = torch.diag(h) / h.sum(1)
        iu = torch.diag(h) / (h.sum(1) + h.sum(0) - torch.diag(h))
        return acc_global, acc, iu

    def reduce_from_all_processes(self):
        if not is_dist_avail_and_initialized():
            return
        if self.mat is None:          # a rank whose shard held no complete batch still takes part in the collective
            self._ensure(torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu'))
        dist.barrier()
        dist.all_reduce(self.mat)

    def __str__(self):
        acc_global, acc, iu = self.compute()
        return ('global correct: {:.1f}\naverage row correct: {}\nIoU: {}\nmean IoU: {:.1f}').format(
            acc_global.item() * 100, ['{:.1f}'.format(i) for i in (acc * 100).tolist()],
            ['{:.1f}'.format(i) for i in (iu * 100).tolist()], iu.mean().item() * 100)


def setup_for_distributed(is_master):
    """Silence print() on non-master ranks unless force=True (util/utils.py:243-255)."""
    import builtins
    builtin_print = builtins.print

    def print_(*args, **kwargs):
        if is_master or kwargs.pop('force', False):
            builtin_print(*args, **kwargs)
        else:
            kwargs.pop('force', None)
    builtins.print = print_


def init_distributed_mode(args):
    """env:// rendezvous, one process per GPU (util/utils.py:287-310).  backend 'nccl' is RCCL on ROCm."""
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        args.rank = int(os.environ['RANK'])
        args.world_size = int(os.environ['WORLD_SIZE'])
        args.gpu = int(os.environ.get('LOCAL_RANK', 0))
    else:
        print('Not using distributed mode')
        args.distributed = False
        return
    args.distributed = True
    # SEGFAC_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses duplicate devices): used by the tests to drive the
    # world_size > 1 code paths on a 1-GPU box; production runs use 'nccl' (= RCCL on ROCm), one rank per GPU
    backend = os.environ.get('SEGFAC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if torch.cuda.is_available():
        args.gpu = args.gpu % torch.cuda.device_count()
        torch.cuda.set_device(args.gpu)
    args.dist_backend = backend
    dist.init_process_group(backend=backend, init_method=getattr(args, 'dist_url', 'env://'),
                            world_size=args.world_size, rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0 or bool(os.environ.get('SEGFAC_PRINT_ALL_RANKS')))     # the override: multi-rank tests read every rank's lines


def load_model(modelpath, model=None):
    """Checkpoint reader with the reference's signature and key handling (util/utils.py:313-324; called as
    `utils.load_model(args.finetune, model)` at train_gpu.py:243 -- the reference never touches `model` either): unwrap
    'state_dict', and for NVIDIA SegFormer files drop decode_head.conv_seg.*"""
    ckpt = torch.load(modelpath, map_location='cpu')
    if isinstance(ckpt, dict) and 'state_dict' in ckpt:
        ckpt = ckpt['state_dict']
    if 'segformer' in os.path.basename(modelpath):
        for k in ('decode_head.conv_seg.weight', 'decode_head.conv_seg.bias'):
            ckpt.pop(k, None)
    return ckpt


def get_pth_file(folder):
    """First *.pth in a folder (auto-resume rule, util/utils.py:327-331)."""
    if not os.path.isdir(folder):
        return None
    for f in sorted(os.listdir(folder)):
        if f.endswith('.pth'):
            return f
    return None


def get_model_size(model):
    """Size of the serialised state_dict in MB (util/utils.py:334-342)."""
    import io
    buf = io.BytesIO()
    if isinstance(model, torch.jit.ScriptModule):
        torch.jit.save(model, buf)
    else:
        torch.save(model.state_dict(), buf)
    return buf.getbuffer().nbytes / 1e6


def cleanup_ddp():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def time_sync() -> float:
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return time.time()


@torch.no_grad()
def throughput(dataloader, model, times: int = 30, lowres: bool = False):
    """util/utils.py:356-367: eval-mode forwards of the first batch, `times` times, images/s printed in the reference's wording (and
    returned).  `model(images)` is the reference's call: full-resolution fp32 logits are produced.  lowres=True times what
    `evaluate` actually runs on this path -- `forward_lowres`, the head output the fused argmax / confusion-matrix kernel consumes."""
    model.eval()
    images, _ = next(iter(dataloader))
    images = images.cuda(non_blocking=True)
    B = images.shape[0]
    core = model.module if hasattr(model, 'module') else model
    fwd = core.forward_lowres if (lowres and hasattr(core, 'forward_lowres')) else model
    fwd(images)                                             # first call: lazy initialisation is not throughput
    print(f"Throughput averaged with {times} times")
    start = time_sync()
    for _ in range(times):
        fwd(images)
    end = time_sync()
    ips = times * B / (end - start)
    print(f"Batch Size {B} throughput {ips} images/s")
    return ips


@torch.no_grad()
def test_model_latency(model, inputs, use_cuda: bool = False) -> float:
    """util/utils.py:370-374: one profiled forward, total self CPU time in ms (torch.autograd.profiler).  use_cuda=True returns the
    DEVICE time of the forward instead, measured with events on the launch stream (the kernels of this path are launched through the
    C ABI and are invisible to torch's profiler)."""
    if use_cuda:
        model(inputs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        model(inputs)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)
    with torch.autograd.profiler.profile(use_cuda=False) as prof:
        _ = model(inputs)
    return prof.self_cpu_time_total / 1000  # ms


test_model_latency.__test__ = False          # (the reference's name; not a pytest case)
