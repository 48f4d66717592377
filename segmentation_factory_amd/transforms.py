"""Device-side input pipeline: the reference's train / val transform stacks over decoded uint8 images resident in HBM.

Reference (datasets/build_datasets.py:14-29):
    train = ExtCompose([ExtRandomCrop((S, S)), ExtColorJitter(0.5, 0.5, 0.5), ExtRandomHorizontalFlip(), ExtToTensor(),
                        ExtNormalize(mean, std)])
    val   = ExtCompose([ExtResize(S), ExtToTensor(), ExtNormalize(mean, std)])
followed by the dataset class' label table and ``.long()`` (datasets/ade.py:122-124, cityscapes.py:159, coco_stuff.py:95-100,
voc.py:230).  There the stack runs per sample on PIL images inside DataLoader workers (about a hundred images per second and
core); here JPEG / PNG decoding stays wherever the caller does it, the decoded bytes are uploaded once (all of ADE20K is ~15 GB
of the 288 GB), and a batch is produced by two kernel launches (csrc/input.hip, `segf_input_train`) with Pillow's uint8
arithmetic reproduced bit for bit, including the second ``/ 255`` of ExtNormalize (quirk Q11).

The RANDOM DRAWS stay on the host and follow the reference's call order on Python's ``random`` module -- ExtRandomCrop.get_params
(extra_transform.py:342-360), ExtColorJitter.get_params (:470-497: brightness, contrast, saturation factor, then
``random.shuffle``), ExtRandomHorizontalFlip (:205-213) -- so that, with the same seed and the same decoded images, a batch equals
what the reference's single-process loader yields.  No CPU fallback: the tensors must be device tensors.
"""
import random as _random

import torch

from . import hip

OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION = 1, 2, 3
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def label_table(mapping=None, device='cuda'):
    """int64[256] label table: identity, overridden by `mapping` {uint8 value: class id} (ADE20K: {255: 0}, datasets/ade.py:123;
    Cityscapes: id_to_train_id, cityscapes.py:159; COCO-Stuff: label_map, coco_stuff.py:98-100)."""
    t = torch.arange(256, dtype=torch.int64)
    for k, v in (mapping or {}).items():
        t[int(k)] = int(v)
    return t.to(device)


def _jitter_range(v):
    """ExtColorJitter._check_input (extra_transform.py:450-467) for brightness / contrast / saturation given as a number."""
    if v is None or v == 0:
        return None
    if isinstance(v, (tuple, list)):
        lo, hi = float(v[0]), float(v[1])
        if not 0 <= lo <= hi:
            raise ValueError(f'jitter range {v} should be between (0, inf)')
        return None if lo == hi == 1 else (lo, hi)
    if v < 0:
        raise ValueError('If a jitter amount is a single number, it must be non negative.')
    return (max(1.0 - v, 0.0), 1.0 + v)


class _Base:
    def __init__(self, mean, std, label_lut, device):
        self.device = torch.device(device)
        self.mean = torch.tensor(mean, dtype=torch.float32, device=self.device)
        self.std = torch.tensor(std, dtype=torch.float32, device=self.device)
        self.label_lut = None if label_lut is None else label_lut.to(self.device, torch.int64).contiguous()

    @staticmethod
    def _check(img, lbl):
        if not (img.is_cuda and lbl.is_cuda):
            raise RuntimeError('the device input pipeline needs device tensors (no CPU fallback)')
        if img.dtype != torch.uint8 or lbl.dtype != torch.uint8:
            raise TypeError('decoded images and labels must be uint8 (PIL RGB / L arrays)')
        if img.dim() != 3 or img.shape[2] != 3 or img.stride(2) != 1 or img.stride(1) != 3 or lbl.stride(-1) != 1 \
                or tuple(lbl.shape) != tuple(img.shape[:2]):
            raise ValueError(f'expected packed [h, w, 3] image and [h, w] label, got {tuple(img.shape)} / {tuple(lbl.shape)}')


class DeviceTrainTransform(_Base):
    """The training stack of build_datasets.py:14-22 for a whole batch.  ``rng``: a ``random.Random`` (default: the ``random``
    module itself, as the reference uses it)."""

    def __init__(self, image_size, brightness=0.5, contrast=0.5, saturation=0.5, flip_p=0.5, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                 label_lut=None, device='cuda', rng=None):
        super().__init__(mean, std, label_lut, device)
        self.size = (int(image_size), int(image_size)) if isinstance(image_size, (int, float)) else tuple(int(v) for v in image_size)
        self.ranges = [(OP_BRIGHTNESS, _jitter_range(brightness)), (OP_CONTRAST, _jitter_range(contrast)),
                       (OP_SATURATION, _jitter_range(saturation))]
        self.flip_p = flip_p
        self.rng = rng if rng is not None else _random

    def draw(self, src_h, src_w):
        """One sample's random values, drawn in the reference's order.  -> (top, left, [(op, factor), ...], flip)."""
        th, tw = self.size
        if src_w == tw and src_h == th:                     # ExtRandomCrop.get_params: no draw when the size already matches
            top, left = 0, 0
        else:
            top = self.rng.randint(0, abs(src_h - th))
            left = self.rng.randint(0, abs(src_w - tw))
        ops = [(op, self.rng.uniform(r[0], r[1])) for op, r in self.ranges if r is not None]
        self.rng.shuffle(ops)
        flip = self.rng.random() < self.flip_p
        return top, left, ops, bool(flip)

    def pack(self, images, labels, params):
        recs = (hip.InputSample * len(images))()
        for k, (img, lbl, (top, left, ops, flip)) in enumerate(zip(images, labels, params)):
            self._check(img, lbl)
            r = recs[k]
            r.img, r.lbl = img.data_ptr(), lbl.data_ptr()
            r.img_stride, r.lbl_stride = img.stride(0), lbl.stride(0)
            r.src_h, r.src_w, r.top, r.left = int(img.shape[0]), int(img.shape[1]), int(top), int(left)
            r.flip = int(flip)
            order = 0
            for j, (op, f) in enumerate(ops):
                order |= op << (2 * j)
                r.factor[j] = f                              # Python float -> C float, as Image.blend's (float)alpha
            r.order = order
        host = torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8)
        return host.to(self.device, non_blocking=False)

    def __call__(self, images, labels, params=None, out=None):
        """images: list of uint8 [h, w, 3] device tensors, labels: list of uint8 [h, w] -> (fp32 [B, 3, S, S], int64 [B, S, S]).
        ``params`` overrides the draws (one (top, left, ops, flip) per sample); ``out`` = (image, label) tensors to fill in place
        (e.g. the input buffers of a captured train step: zero-copy feed)."""
        if params is None:
            params = [self.draw(int(i.shape[0]), int(i.shape[1])) for i in images]
        samples = self.pack(images, labels, params)
        oi, ol = out if out is not None else (None, None)
        return hip.input_train(samples, len(images), self.size[0], self.size[1], self.mean, self.std, self.label_lut, oi, ol)


class DeviceValTransform(_Base):
    """The validation stack of build_datasets.py:24-29 for one image (the resized size depends on the image)."""

    def __init__(self, image_size, mean=IMAGENET_MEAN, std=IMAGENET_STD, label_lut=None, device='cuda'):
        super().__init__(mean, std, label_lut, device)
        self.size = int(image_size)

    def output_size(self, src_h, src_w):
        """torchvision F.resize with an int: smaller edge -> size, the other int(size * long / short); unchanged when equal."""
        short, long_ = (src_w, src_h) if src_w <= src_h else (src_h, src_w)
        if short == self.size:
            return src_h, src_w
        new_long = int(self.size * long_ / short)
        return (new_long, self.size) if src_w <= src_h else (self.size, new_long)

    def __call__(self, img, lbl):
        self._check(img, lbl)
        oh, ow = self.output_size(int(img.shape[0]), int(img.shape[1]))
        return hip.input_val(img, lbl, oh, ow, self.mean, self.std, self.label_lut)


class DeviceDataset:
    """Decoded (image, label) pairs kept in HBM.  ``add`` takes what ``np.array(PIL image)`` gives (uint8 [h, w, 3] / [h, w])."""

    def __init__(self, device='cuda'):
        self.device = torch.device(device)
        self.images, self.labels = [], []

    def add(self, image, label):
        img = torch.as_tensor(image).contiguous()
        lbl = torch.as_tensor(label).contiguous()
        if img.dtype != torch.uint8 or lbl.dtype != torch.uint8:
            raise TypeError('decoded images and labels must be uint8')
        self.images.append(img.to(self.device))
        self.labels.append(lbl.to(self.device))

    def __len__(self):
        return len(self.images)


class DeviceBatchLoader:
    """Iterable of training batches for `engine.train_one_epoch`: what DataLoader(train_set, batch_size, drop_last=True,
    sampler=DistributedSampler / RandomSampler) over the transformed dataset yields (train_gpu.py:211-224), produced on the device.
    The index order is DistributedSampler's (torch/utils/data/distributed.py, drop_last=False): `randperm(n)` from a generator
    seeded `seed + epoch`, PADDED with its own head to `ceil(n / world) * world` entries, rank r taking entries r, r + world, ...;
    `set_epoch` reseeds as `sampler.set_epoch` does (train_gpu.py:323).  train_gpu.py's DistributedSampler is built without a
    seed, i.e. seed 0: pass seed=0 for its exact order.  (The single-process RandomSampler of the reference draws from torch's
    global generator instead; the same function with world=1 is used there -- same distribution, not the same sequence.)"""

    def __init__(self, dataset, batch_size, transform, shuffle=True, seed=0, rank=0, world=1):
        self.dataset, self.batch_size, self.transform = dataset, int(batch_size), transform
        self.shuffle, self.seed, self.rank, self.world, self.epoch = shuffle, seed, rank, world, 0
        self.out = None

    def bind_output(self, buffers):
        """From now on every batch is written into `buffers` = (image fp32 [B, 3, S, S], label int64 [B, S, S]) -- the input
        buffers of a captured train step (graph.GraphedTrainStep.static_inputs) -- and those tensors are what the loader yields:
        the step then has nothing to copy.  The consumer must be done with a batch before asking for the next (stream order
        guarantees it for the replayed step)."""
        img, lbl = buffers
        S = self.transform.size
        if tuple(img.shape) != (self.batch_size, 3, S[0], S[1]) or tuple(lbl.shape) != (self.batch_size, S[0], S[1]) \
                or img.dtype != torch.float32 or lbl.dtype != torch.int64 or not img.is_contiguous() or not lbl.is_contiguous():
            raise ValueError('bind_output: buffers do not match the batches of this loader')
        self.out = (img, lbl)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def num_samples(self):
        return -(-len(self.dataset) // self.world)             # ceil: DistributedSampler.num_samples with drop_last=False

    def __len__(self):
        return self.num_samples() // self.batch_size           # DataLoader(drop_last=True)

    def indices(self):
        """This rank's sample indices for the current epoch (DistributedSampler.__iter__)."""
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        total = self.num_samples() * self.world
        pad = total - len(order)
        if pad > 0:
            order += order[:pad] if pad <= len(order) else (order * (-(-pad // len(order))))[:pad]
        return order[self.rank:total:self.world]

    def __iter__(self):
        order = self.indices()
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield self.transform([self.dataset.images[i] for i in idx], [self.dataset.labels[i] for i in idx], out=self.out)

