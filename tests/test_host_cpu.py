"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol (no compute),
model containers carry the reference's state_dict keys, the product refuses to run without a GPU (no
fallback), and the multi-process reductions work (gloo, world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from segmentation_factory_amd import hip
    if not os.path.isfile(hip.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    header = open(os.path.join(ROOT, 'include', 'segfac.h')).read()
    declared = sorted(set(re.findall(r'\b(segf_\w+)\s*\(', header)))
    assert len(declared) >= 30
    lib = hip.lib()
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/segfac.h but not exported'
    assert set(hip.exported_symbols()) == set(declared)
    assert b'gfx950' in lib.segf_version()


@pytest.mark.parametrize('nc', [19, 150])
def test_state_dict_keys_match_reference_inventory(nc):
    from oracle import weights as OW
    from segmentation_factory_amd import SegmentationModel
    m = SegmentationModel('MiT-B0', num_classes=nc, seg_head='SegFormerHead')
    inv = OW.model_inventory('MiT-B0', 'SegFormerHead', nc)
    sd = m.state_dict()
    assert list(sd.keys()) == list(inv.keys())
    for k, (shape, _) in inv.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert str(m) == 'SegFormer-MiT-B0'
    assert m.decode_head.embed_dim == 768          # quirk Q1
    m2 = SegmentationModel('MiT-B2', num_classes=19, seg_head='SegFormerHead')
    assert list(m2.state_dict().keys()) == list(OW.model_inventory('MiT-B2', 'SegFormerHead', 19).keys())


def test_no_cpu_fallback():
    from segmentation_factory_amd import SegmentationModel
    m = SegmentationModel('MiT-B0', num_classes=19, seg_head='SegFormerHead')
    with pytest.raises(RuntimeError, match='no CPU fallback|parameter container'):
        m(torch.randn(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match='parameter container'):
        m.backbone.patch_embed1.proj(torch.randn(1, 3, 64, 64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'segmentation_factory_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                assert '/root/reference' not in src, f


def test_weight_decay_groups():
    from segmentation_factory_amd import SegmentationModel
    from segmentation_factory_amd.optim import param_groups_weight_decay
    m = SegmentationModel('MiT-B0', num_classes=19, seg_head='SegFormerHead')
    no_decay, decay = param_groups_weight_decay(m, 0.025)
    assert all(p.ndim > 1 for p in decay['params']) and decay['weight_decay'] == 0.025
    assert all(p.ndim <= 1 for p in no_decay['params']) and no_decay['weight_decay'] == 0.
    assert len(decay['params']) + len(no_decay['params']) == len(list(m.parameters()))


def test_metric_logger_line_format(capsys):
    from segmentation_factory_amd import utils
    ml = utils.MetricLogger(delimiter="  ")
    ml.add_meter('lr', utils.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    for _ in ml.log_every(list(range(3)), 1, 'Epoch: [0]'):
        ml.update(loss=1.5, lr=0.001)
    out = capsys.readouterr().out.splitlines()
    # same layout as the reference's util/utils.py:190-208 line, e.g.
    # "Epoch: [0]  [0/6]  eta: 0:00:00  lr: 0.050000  loss: 3.1562 (3.1562)  time: 0.0725  data: 0.0000"
    assert re.match(r'Epoch: \[0\]  \[0/3\]  eta: 0:00:00  lr: 0\.001000  loss: 1\.5000 \(1\.5000\)  time: \d+\.\d{4}  data: \d+\.\d{4}$', out[0])
    assert out[-1].startswith('Epoch: [0] Total time:')


def test_metrics_compute_matches_golden(golden_dir):
    from segmentation_factory_amd.metrics import Metrics
    g = np.load(os.path.join(golden_dir, 'metrics_case.npz'))
    m = Metrics(int(g['nc']), 255, 'cpu')
    m.hist = torch.from_numpy(g['hist'])
    iou, f1, acc = m.compute_iou(), m.compute_f1(), m.compute_pixel_acc()
    assert iou[1] == float(g['miou']) and f1[1] == float(g['mf1']) and acc[1] == float(g['macc'])
    assert np.allclose(np.array(f1[0]), g['f1'], equal_nan=True) and np.allclose(np.array(acc[0]), g['acc'], equal_nan=True)


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SEGFAC_ROOT"])
from segmentation_factory_amd import utils
from segmentation_factory_amd.metrics import Metrics
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="env://", rank=rank, world_size=world)
n = 5
cm = utils.ConfusionMatrix(n); cm.mat = torch.full((n, n), rank + 1, dtype=torch.int64)
m = Metrics(n, 255, "cpu"); m.hist = torch.full((n, n), float(rank + 1))
cm.reduce_from_all_processes(); m.reduce_from_all_processes()
assert int(cm.mat[0, 0]) == 3 and float(m.hist[0, 0]) == 3.0
sv = utils.SmoothedValue(); sv.update(float(rank + 1)); sv.synchronize_between_processes()
assert sv.count == 2 and abs(sv.total - 3.0) < 1e-12
# data-parallel gradient averaging (collective C1): DDP over a parameter container module
lin = torch.nn.Linear(4, 3); torch.manual_seed(0)
for p in lin.parameters(): torch.nn.init.constant_(p, 0.5)
ddp = torch.nn.parallel.DistributedDataParallel(lin)
x = torch.full((2, 4), float(rank + 1))
ddp(x).sum().backward()
assert torch.allclose(lin.weight.grad, torch.full((3, 4), 3.0)), lin.weight.grad   # mean of (2*1, 2*2) = 3
# the graphed step's exchange: ONE all-reduce (mean) of the flat gradient buffer + rank-0 parameter broadcast
from segmentation_factory_amd.graph import allreduce_mean_, broadcast_flat_
flat = torch.arange(6, dtype=torch.float32) * (rank + 1)
allreduce_mean_(flat)
assert torch.allclose(flat, torch.arange(6, dtype=torch.float32) * 1.5), flat
params = torch.full((5,), float(rank + 7))
broadcast_flat_(params, 0)
assert torch.all(params == 7.0)
# the bucketed exchange of the graphed step: aligned collective ranges + both exchange algorithms, fp32 and bf16 payloads
from segmentation_factory_amd.graph import comm_ranges, plan_buckets, sum_over_ranks_
numels = [7, 33, 5, 64, 1, 90, 13]
total = sum(numels)
plan = plan_buckets(numels, 60)
buckets = [(lo, hi) for lo, hi, _, _ in plan]
rng_ = comm_ranges(buckets, total, world * 16)
assert rng_[0][1] >= total and rng_[-1][0] == 0 and all((hi - lo) % (world * 16) == 0 for lo, hi in rng_)
assert all(rng_[k][0] == rng_[k + 1][1] for k in range(len(rng_) - 1))          # exact tiling, completion order
assert all(r[0] >= b[0] for r, b in zip(rng_, buckets))                          # a range never reaches into a later-completing bucket
for mode in ("all_reduce", "rs_ag"):
    for dt in (torch.float32, torch.bfloat16):
        buf = torch.zeros(rng_[0][1], dtype=dt)
        buf[:total] = (torch.arange(total) % 17).to(dt) * (rank + 1)
        for lo, hi in rng_:
            for w in sum_over_ranks_(buf[lo:hi], mode):
                w.wait()
        want = torch.zeros_like(buf); want[:total] = (torch.arange(total) % 17).to(dt) * 3
        assert torch.equal(buf, want), (mode, dt)
# collective C2: rank 0's BatchNorm buffers to every rank before evaluate (train_gpu.py:233-236 broadcast_buffers=True), one flat
# broadcast per dtype; parameters are NOT touched
from segmentation_factory_amd.graph import broadcast_buffers_
bn = torch.nn.Sequential(torch.nn.BatchNorm1d(3), torch.nn.Linear(3, 2), torch.nn.BatchNorm1d(2))
with torch.no_grad():
    for i, bnm in enumerate((bn[0], bn[2])):
        bnm.running_mean.fill_(10.0 * rank + i); bnm.running_var.fill_(2.0 + rank); bnm.num_batches_tracked.fill_(5 + 3 * rank)
    bn[1].weight.fill_(float(rank))
assert broadcast_buffers_(bn) == 6
assert float(bn[0].running_mean[0]) == 0.0 and float(bn[2].running_mean[0]) == 1.0 and float(bn[2].running_var[1]) == 2.0
assert int(bn[0].num_batches_tracked) == 5 and bn[0].num_batches_tracked.dtype == torch.int64
assert float(bn[1].weight[0, 0]) == float(rank)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_process_gloo_reductions(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    env = dict(os.environ, SEGFAC_ROOT=ROOT, MASTER_ADDR='127.0.0.1', MASTER_PORT='29561', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_train_gpu_cli_flags_and_scheduler_quirk():
    """The CLI keeps the reference's flags and defaults (train_gpu.py:33-184); the default schedule is inert (quirk Q9)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('train_gpu_cli', os.path.join(ROOT, 'train_gpu.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import argparse
    args = argparse.ArgumentParser(parents=[mod.get_args_parser()]).parse_args([])
    assert (args.batch_size, args.epochs, args.clip_grad, args.clip_mode, args.opt, args.lr, args.weight_decay) == (4, 5, 0.02, 'agc', 'adamw', 1e-3, 0.025)
    assert (args.backbone, args.heads, args.nb_classes, args.image_size, args.ignore_index, args.dice) == ('MiT-B2', 'SegFormerHead', 19, 1024, 255, True)
    assert (args.warmup_lr, args.min_lr, args.sched, args.lr_ep, args.device, args.dist_url) == (2e-4, 1e-4, 'cosine', False, 'cuda', 'env://')
    from segmentation_factory_amd.scheduler import create_scheduler
    import torch
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([p], lr=args.lr)
    sch, _ = create_scheduler(args, opt)
    assert opt.param_groups[0]['lr'] == args.warmup_lr            # constructor writes the warm-up start value
    for e in range(3):
        sch.step(e)
    assert opt.param_groups[0]['lr'] == args.warmup_lr            # ... and step(epoch) never moves it without --lr-ep
    ds = mod.SyntheticSegDataset(4, 32, 7)
    img, lbl = ds[1]
    assert img.shape == (3, 32, 32) and img.dtype == torch.float32 and lbl.dtype == torch.int64 and int(lbl[0, 0]) == 255
    assert int(lbl[2:].max()) < 7


def test_oracle_optimizer_restatement_properties():
    """oracle/optim.py (AGC + AdamW, parity unpinned): unit-wise clipping bound, no-op below the bound, eps floor for tiny
    parameters, and agreement of the AdamW restatement with torch.optim.AdamW itself (which IS available here)."""
    import torch
    from oracle import optim as OO
    g = torch.Generator().manual_seed(5)
    p, gr = torch.randn(7, 3, 3, generator=g), torch.randn(7, 3, 3, generator=g) * 10
    c = OO.adaptive_clip_grad_(p, gr, 0.02)
    pn, cn = p.flatten(1).norm(dim=1), c.flatten(1).norm(dim=1)
    assert torch.all(cn <= 0.02 * pn.clamp(min=1e-3) * (1 + 1e-5))
    small = gr * 1e-6
    assert torch.equal(OO.adaptive_clip_grad_(p, small, 0.02), small)
    tiny = torch.zeros(4, 5)
    ct = OO.adaptive_clip_grad_(tiny, torch.ones(4, 5), 0.02)
    assert torch.allclose(ct.norm(dim=1), torch.full((4,), 0.02 * 1e-3), rtol=1e-5)
    w = torch.nn.Parameter(torch.randn(5, 4, generator=g))
    ref = torch.optim.AdamW([w], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    pw, m, v = w.detach().clone(), torch.zeros(5, 4), torch.zeros(5, 4)
    for step in range(1, 5):
        grad = torch.randn(5, 4, generator=g)
        w.grad = grad.clone()
        ref.step()
        OO.adamw_step_(pw, grad, m, v, step, 3e-3, weight_decay=0.05)
        assert torch.allclose(w.detach(), pw, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('backbone,head,nc', [('MobileNetV2', 'FPNHead', 21), ('ConvNeXt', 'UPerHead', 150),
                                              ('convnextv2_tiny', 'UPerHead', 19), ('convnextv2_large', 'UPerHead', 171)])
def test_state_dict_keys_of_every_baseline_family(backbone, head, nc):
    """BASELINE cfg1 / cfg3 / cfg5 models: key order, shapes, head width (quirk Q1) and __str__ equal the reference's inventory
    (oracle/weights.py, itself asserted against the imported reference in oracle/make_goldens.py)."""
    from oracle import weights as OW
    from segmentation_factory_amd import SegmentationModel
    m = SegmentationModel(backbone, num_classes=nc, seg_head=head)
    inv = OW.model_inventory(backbone, head, nc)
    sd = m.state_dict()
    assert list(sd.keys()) == list(inv.keys())
    for k, (shape, _) in inv.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert str(m) == f'{backbone}_{head}'
    assert m.decode_head.embed_dim == OW.head_width(backbone)


def test_load_model_and_pretrained_backbone_key_handling(tmp_path):
    """util/utils.py:313-324 (load_model: unwrap 'state_dict'; for NVIDIA SegFormer files drop decode_head.conv_seg.*),
    train_gpu.py:246-252 (drop linear_pred, load strict=False), build_models.py:56-60 (pretrained_backbone)."""
    from oracle import weights as OW
    from segmentation_factory_amd import SegmentationModel, utils
    sd = OW.make_state_dict('MiT-B0', 'SegFormerHead', 19, 5)
    nvidia = dict(sd)
    nvidia['decode_head.conv_seg.weight'] = torch.zeros(150, 768, 1, 1)      # the extra classifier of the released checkpoints
    nvidia['decode_head.conv_seg.bias'] = torch.zeros(150)
    f1 = tmp_path / 'segformer.b0.512x512.ade.160k.pth'
    torch.save({'state_dict': nvidia, 'meta': {'note': 'NVIDIA-style file'}}, str(f1))
    got = utils.load_model(str(f1))
    assert 'decode_head.conv_seg.weight' not in got and 'decode_head.conv_seg.bias' not in got
    assert set(got) == set(sd)
    f2 = tmp_path / 'plain_checkpoint.pth'                                    # no 'state_dict' wrapper, no 'segformer' in the name
    torch.save(dict(sd), str(f2))
    assert set(utils.load_model(str(f2))) == set(sd)
    # the finetune load of train_gpu.py: classifier dropped, everything else restored
    m = SegmentationModel('MiT-B0', num_classes=7, seg_head='SegFormerHead')        # another class count than the checkpoint
    ck = {k: v for k, v in got.items() if 'linear_pred' not in k}
    msg = m.load_state_dict(ck, strict=False)
    assert sorted(msg.missing_keys) == ['decode_head.linear_pred.bias', 'decode_head.linear_pred.weight'] and not msg.unexpected_keys
    assert torch.equal(m.state_dict()['backbone.block1.0.attn.q.weight'], sd['backbone.block1.0.attn.q.weight'])
    # pretrained_backbone: a file of backbone keys WITHOUT the 'backbone.' prefix, loaded strict=False
    bb = {k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}
    f3 = tmp_path / 'mit_b0.pth'
    torch.save(bb, str(f3))
    m2 = SegmentationModel('MiT-B0', pretrained_backbone=str(f3), num_classes=19, seg_head='SegFormerHead')
    assert torch.equal(m2.backbone.state_dict()['patch_embed1.proj.weight'], bb['patch_embed1.proj.weight'])
    m3 = SegmentationModel('MiT-B0', pretrained_backbone=str(tmp_path / 'missing.pth'), num_classes=19, seg_head='SegFormerHead')
    assert not torch.equal(m3.backbone.state_dict()['patch_embed1.proj.weight'], bb['patch_embed1.proj.weight'])


def test_fused_optimizer_state_dict_is_torch_adamw_layout():
    """The checkpoint's 'optimizer_state' (train_gpu.py:354-362): FusedAGCAdamW reads a torch.optim.AdamW state_dict (what the
    reference's finetune path saves) into its flat buffers and writes the same layout back (host-side plumbing only)."""
    from segmentation_factory_amd.optim import FusedAGCAdamW
    g = torch.Generator().manual_seed(9)
    ps = [torch.nn.Parameter(torch.randn(4, 3, generator=g)), torch.nn.Parameter(torch.randn(5, generator=g)),
          torch.nn.Parameter(torch.randn(2, 2, 3, generator=g))]
    ps[1].requires_grad_(False)                                  # a frozen parameter keeps its packed index but has no state
    ref = torch.optim.AdamW(ps, lr=2e-4, weight_decay=0.025)
    for _ in range(3):
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g) if p.requires_grad else None
        ref.step()
    sd = ref.state_dict()
    fused = FusedAGCAdamW([torch.nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad) for p in ps], lr=1.0, weight_decay=0.5)
    fused.load_state_dict(sd)
    assert fused.param_groups[0]['lr'] == 2e-4 and fused.param_groups[0]['weight_decay'] == 0.025 and fused._step == 3
    assert torch.equal(fused._m[:12].view(4, 3), sd['state'][0]['exp_avg'])
    assert fused._offsets == [0, 16] and fused._m.numel() == 32        # every parameter starts on a PARAM_ALIGN (8) boundary
    assert torch.equal(fused._v[16:28].view(2, 2, 3), sd['state'][2]['exp_avg_sq']) and not fused._v[12:16].any()
    out = fused.state_dict()
    assert set(out['state']) == {0, 2} and float(out['state'][2]['step']) == 3.0
    back = torch.optim.AdamW(ps, lr=1.0)
    back.load_state_dict(out)                                    # torch accepts what we wrote
    assert torch.equal(back.state[ps[0]]['exp_avg'], sd['state'][0]['exp_avg'])


def test_schedulers_match_reference_lr_sequences(golden_dir):
    """(f4) every --sched value: create_scheduler driven like train_gpu.py (step_update per iteration, step(epoch) per epoch)
    reproduces the LR sequences captured from the reference's scheduler package (tests/golden/scheduler_cases.json,
    oracle/make_goldens.py::scheduler_cases), both parameter groups, including restart cycles, k-decay and seeded LR noise."""
    import json
    import types
    from segmentation_factory_amd.scheduler import create_scheduler
    cases = json.load(open(os.path.join(golden_dir, 'scheduler_cases.json')))
    assert {c['name'] for c in cases} >= {'cosine_default_inert', 'cosine_epochs', 'tanh', 'step', 'multistep', 'poly'}
    for c in cases:
        a = c['args']
        args = types.SimpleNamespace(**a)
        p, q = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))
        opt = torch.optim.SGD([{'params': [p]}, {'params': [q], 'lr': a['lr'] * 0.5}], lr=a['lr'])
        sch, n_epochs = create_scheduler(args, opt)
        assert n_epochs == c['num_epochs'], c['name']
        assert [g['lr'] for g in opt.param_groups] == pytest.approx(c['seq']['init'], rel=1e-12), c['name']
        n_iter = a['data_len'] // (a['batch_size'] * a['world_size'])
        ep_seq, up_seq, upd = [], [], 0
        for ep in range(n_epochs + 2):
            for _ in range(n_iter):
                upd += 1
                sch.step_update(upd)
                if upd % 5 == 0:
                    up_seq.append([g['lr'] for g in opt.param_groups])
            sch.step(ep)
            ep_seq.append([g['lr'] for g in opt.param_groups])
        assert np.allclose(ep_seq, c['seq']['epoch'], rtol=1e-10, atol=0), c['name']
        assert np.allclose(up_seq, c['seq']['update'], rtol=1e-10, atol=0), c['name']
        assert sorted(sch.state_dict().keys()) == c['state_keys'], c['name']      # same 'scheduler_state' keys in the checkpoint
    # quirk Q9: the reference's loop only ever calls step(epoch) (train_gpu.py:336), and without --lr-ep that never moves the rate;
    # the step_update calls above are what WOULD drive the default cosine
    inert = next(c for c in cases if c['name'] == 'cosine_default_inert')
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([p], lr=inert['args']['lr'])
    sch, n_epochs = create_scheduler(types.SimpleNamespace(**inert['args']), opt)
    for ep in range(n_epochs):
        sch.step(ep)
        assert opt.param_groups[0]['lr'] == inert['args']['warmup_lr']
    assert len({tuple(v) for v in inert['seq']['update']}) > 1


def test_plateau_scheduler_wraps_torch_reduce_on_plateau():
    """--sched plateau (scheduler/plateau_lr.py:10-102; the reference class itself does not construct on torch >= 2.4): warm-up,
    then torch.optim.lr_scheduler.ReduceLROnPlateau on the epoch metric; state_dict = {'best', 'last_epoch'}."""
    import types
    from segmentation_factory_amd.scheduler import create_scheduler
    args = types.SimpleNamespace(epochs=12, data_len=40, batch_size=4, world_size=1, warmup_epochs=2, cooldown_epochs=0, min_lr=1e-5,
                                 warmup_lr=1e-4, lr=1e-2, lr_ep=False, sched='plateau', decay_rate=0.5, patience_epochs=2, seed=0)
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([p], lr=args.lr)
    sch, _ = create_scheduler(args, opt)
    assert opt.param_groups[0]['lr'] == 1e-4
    ref_p = torch.nn.Parameter(torch.zeros(3))
    ref_opt = torch.optim.SGD([ref_p], lr=args.lr)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(ref_opt, patience=2, factor=0.5, threshold=1e-4, cooldown=0, mode='max', min_lr=1e-5)
    metrics = [10, 20, 30, 30, 30, 30, 30, 31, 31, 31, 31, 31]
    for ep, m in enumerate(metrics):
        sch.step(ep, m)
        if ep <= 2:
            assert opt.param_groups[0]['lr'] == pytest.approx(1e-4 + ep * (1e-2 - 1e-4) / 2)
        else:
            ref.step(m, ep)
            assert opt.param_groups[0]['lr'] == ref_opt.param_groups[0]['lr']
    assert opt.param_groups[0]['lr'] < 1e-2 and set(sch.state_dict()) == {'best', 'last_epoch'}


def test_device_batch_loader_index_order_is_distributed_samplers():
    """transforms.DeviceBatchLoader.indices() == list(DistributedSampler(...)) (train_gpu.py:212-214: shuffle=True, default seed 0,
    drop_last=False => padded with the head of the permutation), for dataset sizes that do and do not divide by the world size."""
    from torch.utils.data.distributed import DistributedSampler
    from segmentation_factory_amd.transforms import DeviceBatchLoader
    for n, world in ((10, 2), (11, 2), (7, 3), (2, 3), (64, 8)):
        data = list(range(n))
        for rank in range(world):
            ld = DeviceBatchLoader(data, 2, transform=None, shuffle=True, seed=0, rank=rank, world=world)
            sm = DistributedSampler(data, num_replicas=world, rank=rank, shuffle=True)
            for epoch in (0, 3):
                ld.set_epoch(epoch)
                sm.set_epoch(epoch)
                assert ld.indices() == list(sm), (n, world, rank, epoch)
                assert ld.num_samples() == len(sm) and len(ld) == len(sm) // 2


def test_dispatch_policy_table_is_one_place():
    """Every dispatch switch lives in csrc/policy.h, is read from the environment once and is documented in include/segfac.h; nothing
    else in csrc/ calls getenv().  segf_policy_get / set / reload work without a GPU."""
    import re
    from segmentation_factory_amd import hip
    table = hip.policy_table()
    assert len(table) >= 40 and len({t[0] for t in table}) == len(table) and len({t[1] for t in table}) == len(table)
    header = open(os.path.join(ROOT, 'include', 'segfac.h')).read()
    for field, env, value, default, doc in table:
        assert env.startswith('SEGFAC_') and env in header, env          # documented where the C ABI is declared
        assert doc
    csrc = os.path.join(ROOT, 'segmentation_factory_amd', 'csrc')
    for fn in os.listdir(csrc):
        if fn.endswith(('.hip', '.h')) and fn != 'policy.hip':
            assert 'getenv' not in re.sub(r'//.*', '', open(os.path.join(csrc, fn)).read()), fn
    pkg = os.path.join(ROOT, 'segmentation_factory_amd')
    known = {t[1] for t in table} | {'SEGFAC_HIP_LIB', 'SEGFAC_VERBOSE', 'SEGFAC_EXCHANGE', 'SEGFAC_GRAD_PAYLOAD', 'SEGFAC_FORCE_EXCHANGE',
                                      'SEGFAC_TEST_SPIN_US', 'SEGFAC_DIST_BACKEND', 'SEGFAC_PRINT_ALL_RANKS'}       # run-time plumbing, not dispatch
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            for name in re.findall(r'SEGFAC_[A-Z0-9_]+', open(os.path.join(pkg, fn)).read()):
                assert name in known, (fn, name)
    assert hip.policy('gemm_no_narrow') == 0 and hip.policy('SEGFAC_G8_STAGGER') == -1
    with hip.policy_override(gemm_no_narrow=1):
        assert hip.policy('SEGFAC_GEMM_NO_NARROW') == 1
    assert hip.policy('gemm_no_narrow') == 0
    os.environ['SEGFAC_GEMM_NO_NARROW'] = 'yes'                               # (conftest re-reads the policy after SEGFAC_* changes)
    try:
        assert hip.policy('gemm_no_narrow') == 1
    finally:
        del os.environ['SEGFAC_GEMM_NO_NARROW']
    assert hip.policy('gemm_no_narrow') == 0
    with pytest.raises(KeyError):
        hip.policy('no_such_switch')


def test_dispatch_of_baseline_shapes(golden_dir):
    """The kernel every launching C-ABI call of one train step takes, for cfg2 at per-GPU batch 4 / 16 / 128 and cfg3 / cfg4 / cfg5
    (+ the fp8 option), recorded on the MI355X by tools/make_dispatch_table.py (shapes from the reference's width rule,
    models/build_models.py:43-54) -- replayed here as DRY RUNS through the C ABI: same entry point, same arguments, placeholder
    pointers, nothing launched.  A dispatch edit that moves a BASELINE shape to another kernel fails this test and shows up as a diff
    of tests/golden/dispatch_table.json when the table is regenerated."""
    import json
    from segmentation_factory_amd import dispatch
    with open(os.path.join(golden_dir, 'dispatch_table.json')) as fh:
        table = json.load(fh)
    assert {'cfg2_b4', 'cfg2_b16', 'cfg2_b128', 'cfg2_b256', 'cfg3_b64', 'cfg4_b32', 'cfg5_b32', 'cfg3_b32', 'cfg4_b16', 'cfg5_b8'} <= set(table)
    moved, n = [], 0
    for case, entries in table.items():
        fns = {e['fn'] for e in entries}
        assert 'segf_gemm' in fns and 'segf_ce_dice_fwd' in fns, case
        assert ('segf_attention_fwd' in fns) == case.startswith(('cfg2', 'cfg4')) and ('segf_conv3x3' in fns) == case.startswith(('cfg3', 'cfg5')), case
        for e in entries:
            got = dispatch.replay(e)
            n += 1
            if got != e['kernels']:
                moved.append((case, e['fn'], e['args'], e['kernels'], got))
    assert n > 500
    assert not moved, moved[:5]


def test_bench_default_batches_resolve():
    """bench.py / tools/bench_legs.py: a configuration without --batch runs at bench.DEFAULT_BATCH (r05: a misplaced comment once made the
    legs' default the whole table, and the cfg3 / cfg4 / cfg5 legs of the default line failed on the GPU box only)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench
    import bench_legs
    assert bench.DEFAULT_BATCH == {'cfg2': 256, 'cfg3': 64, 'cfg4': 32, 'cfg5': 32}
    for cfg, b in bench.DEFAULT_BATCH.items():
        assert bench_legs.resolve_batch(cfg) == b and isinstance(bench_legs.resolve_batch(cfg), int)
        assert bench_legs.resolve_batch(cfg, 7) == 7
    assert set(bench.DEFAULT_BATCH) == set(bench_legs.CONFIGS) | {'cfg2'} or set(bench.DEFAULT_BATCH) >= set(bench_legs.CONFIGS)


def test_attention_backward_query_chunks_fill_whole_rounds():
    """csrc/attention.hip: attn_chunks (through segf_attention_bwd_ws, host arithmetic only).  The key-side backward runs in rounds of 512
    resident workgroups; r04's rule stopped at ">= 512 workgroups" and gave 192 images x 1 head 3 chunks = 576 workgroups (a full round
    and one of 64: that kernel cost 28 % more per image at batch 192 than at 256).  The count now minimises rounds x (queries per chunk +
    a fixed cost): 8 chunks = three full rounds there, the r04 choices at the benchmarked power-of-two batches."""
    from segmentation_factory_amd import hip
    lib = hip.lib()

    def nchunk(B, heads, N, Nkv, hd):
        ws = lib.segf_attention_bwd_ws(B, heads, N, Nkv, hd)
        per = B * Nkv * 2 * heads * hd
        n, rem = divmod(ws - B * heads * N, per)
        assert rem == 0 and n >= 1
        return n
    assert nchunk(192, 1, 16384, 256, 32) == 8          # 1536 workgroups = 3 rounds
    assert nchunk(256, 1, 16384, 256, 32) == 2          # 512: as before
    assert nchunk(128, 1, 16384, 256, 32) == 4
    assert nchunk(4, 1, 16384, 256, 32) == 64           # far below one round: as many chunks as the rule allows
    assert nchunk(32, 1, 131072, 2048, 64) == 1         # cfg4 stage 1 at batch 32: 16 key blocks x 32 images = 512
    assert nchunk(16, 1, 131072, 2048, 64) == 2
    for B in (3, 7, 24, 48, 100, 192, 200, 256, 384):
        for heads, N in ((1, 16384), (2, 4096), (5, 1024), (8, 256)):
            n = nchunk(B, heads, N, 256, 32)
            wg = B * heads * n
            if wg >= 512:
                assert wg / (-(-wg // 512) * 512) >= 0.74, (B, heads, N, n, wg)


def test_weight_gradient_slices_fill_whole_rounds_of_the_eight_phase_tile():
    """segf_gemm_pick_splitk (host arithmetic): nn.Linear weight gradients that csrc/gemm.hip:dw_on_gemm8 sends to the eight-phase kernel
    (both feature counts multiples of 256, >= 100 GFLOP, >= 256 FLOP per operand byte) get the smallest slice count whose last round of
    256 tiles is >= 90 % full (else the fullest); smaller or thinner products keep the 128-tile kernel's rule (SEGFAC_GEMM8_DW=0: all do)."""
    from segmentation_factory_amd import hip
    assert hip.pick_splitk(3072, 768, 51200) == 7          # 36 tiles x 7 = 252 of 256
    assert hip.pick_splitk(768, 3072, 51200) == 7
    assert hip.pick_splitk(6144, 1536, 12800) == 5         # 144 tiles x 5 = 720 = 2.8 rounds
    assert hip.pick_splitk(2048, 512, 65472) == 15         # 16 tiles x 15 = 240 of 256 (MiT-B2 stage 4 at batch 32)
    with hip.policy_override(gemm8_dw=0):
        assert hip.pick_splitk(3072, 768, 51200) != 7      # the 128-tile kernel's count (144 tiles of 128 x 128)
    # below the rule: ConvNeXt-T stage 4 at batch 64 (77 GFLOP), a 384-wide operand (not a multiple of 256)
    for shape in ((3072, 768, 16384), (1536, 384, 204800)):
        with hip.policy_override(gemm8_dw=0):
            want = hip.pick_splitk(*shape)
        assert hip.pick_splitk(*shape) == want
