"""GPU end-to-end parity of the HIP SegFormer path against (a) the reference's captured outputs
(tests/golden/e2e_*.npz) and (b) the CPU oracle on the same seeded inputs, in the exact-fp32 mode (north-star
tolerance 1e-3 relative) and in the bf16 production mode (bf16-rounding tolerance, stated below)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import loss as OL            # noqa: E402  (checker only)
from oracle import nets as ON            # noqa: E402
from oracle import weights as OW         # noqa: E402
from oracle.make_goldens import sample_indices   # noqa: E402


def _build(backbone, head, nc, sd, dtype, B, deterministic=True):
    from segmentation_factory_amd import SegmentationModel
    m = SegmentationModel(backbone, num_classes=nc, seg_head=head, compute_dtype=dtype)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    m = m.cuda()
    if deterministic:
        # DropPath / Dropout2d rates forced to 0 on the instance
        for mod in m.backbone.modules():
            if hasattr(mod, 'drop_prob'):
                mod.drop_prob = 0.0
        m.decode_head.dropout.p = 0.0
    return m


E2E_TAGS = ['segformer_b0_64', 'segformer_b0_96x128', 'convnext_uper_64', 'convnextv2_tiny_uper_64', 'mbv2_fpn_64',
            'convnext_uper_128', 'convnextv2_tiny_uper_128', 'mbv2_fpn_128',
            # H, W that are not multiples of 32 (datasets/build_datasets.py:24-29 keeps the aspect ratio, train_gpu.py:72 evaluates at
            # batch 1): 19 x 25 -> 10 x 13 -> 5 x 7 -> 3 x 4 maps, remainder-dropping spatial-reduction convs, non-2/4/8 head resizes
            'segformer_b0_75x100', 'convnext_uper_90x123', 'mbv2_fpn_70x94']
# Gradient tolerances (fraction `rt` of a parameter's gradient scale, see `tol` below), set from tools/grad_parity_report.py on
# the MI355X with ~2x margin.  Measured worst sample error / scale: fp32 segformer 1e-4, convnext_64 3e-3, convnext*_128 2e-3,
# mbv2_64 8e-3, mbv2_128 1.3e-2; bf16 segformer 0.064, convnext_uper_128 0.115, convnextv2_tiny_uper_128 0.133.
GRAD_RT = {'segformer': (3e-3, 0.12), 'convnext_64': (2e-2, None), 'convnext_128': (5e-3, 0.25), 'mbv2': (3e-2, None)}


def _kind(tag):
    if tag.startswith('segformer'):
        return 'segformer'
    if tag.startswith('mbv2'):
        return 'mbv2'
    # convnext_uper_90x123 (batch 2, 22 x 30 ... 2 x 3 maps) is as BatchNorm-ill-conditioned as the 64 x 64 fixture
    return 'convnext_128' if tag.endswith('_128') else 'convnext_64'


@pytest.mark.parametrize('tag', E2E_TAGS)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_e2e_against_reference_golden(golden_dir, tag, dtype):
    """Every BASELINE model family against the REFERENCE's captured outputs: eval logits, train-mode logits, loss, sampled
    parameter gradients + gradient norms, BatchNorm step counters.  The *_128 fixtures (128 x 128, batch 4 / 2) are the
    BatchNorm-well-conditioned ones: there the bf16 production mode is held to SAMPLED-GRADIENT checks as well.  At the 64 x 64,
    batch-2 fixtures ConvNeXt/UPer's and MobileNetV2/FPN's BatchNorms see 2..32 samples and bf16 rounding is amplified by
    1/sigma of near-constant channels (fp32 stays within 1e-3 / 3e-2 there): bf16 is then held to logits, loss and
    gradient norms only, and the same composition is pinned tightly by the *_128 fixture."""
    from segmentation_factory_amd import criterion_lowres
    g = np.load(os.path.join(golden_dir, f'e2e_{tag}.npz'))
    backbone, head, nc = str(g['backbone']), str(g['head']), int(g['nc'])
    B, H, W, seed = int(g['B']), int(g['H']), int(g['W']), int(g['seed'])
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    fp32 = dtype == torch.float32
    kind = _kind(tag)
    compact = 'lowres_eval' in g.files      # head output + strided sample of the full-size logits (oracle/make_goldens.py)
    rel = 1e-3 if fp32 else 5e-2          # north star: 1e-3 (fp32 path); bf16 storage: ~2^-8 per op, ~40 ops deep
    # MobileNetV2 in bf16 TRAIN mode: 52 BatchNorms, several over (near-)dead ReLU6 channels of the random-weight fixture,
    # whose 1/sigma amplifies the storage rounding (measured 0.22-0.31 of the logit scale; eval mode 7e-3..1e-2, fp32 3e-5).
    # cfg1 is the reference's fp32 CPU configuration: fp32 is the pinned mode, bf16 is only sanity-checked here.
    rel_train = rel if (fp32 or kind != 'mbv2') else 0.6
    model = _build(backbone, head, nc, sd, dtype, B)
    # eval-mode logits (SegmentationModel.forward, build_models.py:62-66)
    model.eval()
    with torch.no_grad():
        ev = model(x.cuda()).cpu().numpy()
        if compact:
            lo = model.forward_lowres(x.cuda()).nchw().float().cpu().numpy()
            assert lo.shape == g['lowres_eval'].shape
            assert np.abs(lo - g['lowres_eval']).max() <= rel * np.abs(g['lowres_eval']).max()
            assert np.abs(ev[:, :, 1::4, 2::4] - g['logits_eval_sub']).max() <= rel * np.abs(g['logits_eval_sub']).max()
        else:
            assert ev.shape == g['logits_eval'].shape
            assert np.abs(ev - g['logits_eval']).max() <= rel * np.abs(g['logits_eval']).max()
    # train-mode forward + fused loss + backward
    model.train()
    lo = model.forward_lowres(x.cuda())
    loss = criterion_lowres(lo, y.cuda(), (H, W), None, num_classes=nc, dice=True, ignore_index=255)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= (2e-4 if fp32 else 2e-2) * abs(float(g['loss']))
    with torch.no_grad():
        if compact:
            tr = model.forward_lowres(x.cuda()).nchw().float().cpu().numpy()        # second train-mode forward: same batch statistics
            assert np.abs(tr - g['lowres_train']).max() <= rel_train * np.abs(g['lowres_train']).max()
        else:
            tr = model(x.cuda()).cpu().numpy()
            assert np.abs(tr - g['logits_train']).max() <= rel_train * np.abs(g['logits_train']).max()
    gmax = float(g['grad_global_max'])
    params = dict(model.named_parameters())
    rt = GRAD_RT[kind][0 if fp32 else 1]
    norms_only = rt is None                 # bf16 at the ill-conditioned fixtures: per-parameter gradient NORMS only
    bad = []
    for i, name in enumerate(g['grad_names']):
        name = str(name)
        gr = params[name].grad
        ref_norm = float(g['grad_norms'][i])
        if gr is None:
            assert ref_norm == 0.0, name      # FPNHead.output_convs[0] is never called (quirk Q3): no gradient in the reference either
            continue
        gr = gr.detach().float().cpu()
        got = gr.flatten()[sample_indices(name, gr.numel())].numpy()
        if norms_only:
            if 'ppm.stages.0.' in name:      # PPM scale 1: BatchNorm over the 2 samples of a 1x1 map, a (near-)singular Jacobian
                continue
            if abs(gr.double().norm().item() - ref_norm) > 0.6 * ref_norm + 1e-1 * gmax:
                bad.append((name, gr.double().norm().item(), ref_norm))
            continue
        tol = rt * (np.abs(g['grad_samples'][i]).max() + ref_norm / max(1.0, np.sqrt(gr.numel()))) + rt * 1e-2 * gmax
        if np.abs(got - g['grad_samples'][i]).max() > tol or abs(gr.double().norm().item() - ref_norm) > rt * ref_norm + rt * 1e-1 * gmax:
            bad.append((name, float(np.abs(got - g['grad_samples'][i]).max()), tol, gr.double().norm().item(), ref_norm))
    assert not bad, bad[:8]
    if fp32:
        sdn = model.state_dict()
        for i, name in enumerate(g['bn_names']):
            v = sdn[str(name)]
            got = v.float().double().norm().item() if v.ndim else float(v)
            # two train-mode forwards ran above -> compare only the first-forward quantity we can: num_batches_tracked
            # (FPNHead's lateral BatchNorms count 3 / 2 / 2 per forward, quirk Q3)
            if str(name).endswith('num_batches_tracked'):
                assert got == 2 * float(g['bn_norms'][i]), str(name)


@pytest.mark.parametrize('tag', ['segformer_b0_64', 'mbv2_fpn_64', 'mbv2_fpn_128', 'convnext_uper_128'])
def test_bn_running_stats_after_one_forward(golden_dir, tag):
    """BatchNorm running_mean / running_var / num_batches_tracked after ONE train-mode forward vs the reference's buffers
    (captures quirk Q3: FPNHead evaluates its lateral ConvModules 1 / 3 / 2 / 2 times per forward, fpn.py:29-36)."""
    g = np.load(os.path.join(golden_dir, f'e2e_{tag}.npz'))
    backbone, head, nc = str(g['backbone']), str(g['head']), int(g['nc'])
    B, H, W, seed = int(g['B']), int(g['H']), int(g['W']), int(g['seed'])
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, _ = OW.synthetic_batch(B, H, W, nc, seed)
    model = _build(backbone, head, nc, sd, torch.float32, B).train()
    model.forward_lowres(x.cuda())
    sdn = model.state_dict()
    assert len(g['bn_names']) > 0
    for i, name in enumerate(g['bn_names']):
        v = sdn[str(name)]
        got = v.float().double().norm().item() if v.ndim else float(v)
        assert abs(got - float(g['bn_norms'][i])) <= 1e-4 * max(1.0, abs(float(g['bn_norms'][i]))), str(name)


def test_stochastic_layers_match_oracle_with_shared_masks():
    """DropPath / Dropout2d with explicit keep-masks: product (fp32) vs oracle given the same draws."""
    from segmentation_factory_amd import criterion_lowres
    backbone, head, nc, B, H, W, seed = 'MiT-B0', 'SegFormerHead', 7, 3, 64, 64, 5
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    gen = torch.Generator().manual_seed(0)
    dp = (torch.rand(ON.count_drop_path_draws(backbone), B, generator=gen) > 0.3).float()
    d2 = (torch.rand(B, 768, generator=gen) > 0.1).float()
    o, _ = ON.model_forward(sd, x, backbone, head, training=True, masks={'drop_path': list(dp), 'dropout2d': d2}, lowres=True)
    model = _build(backbone, head, nc, sd, torch.float32, B, deterministic=False).train()
    model.backbone.stochastic_override = {'drop_path': dp}
    model.decode_head.stochastic_override = {'dropout2d': d2}
    lo = model.forward_lowres(x.cuda())
    got = lo.nchw().float().cpu()
    assert (got - o).abs().max() <= 1e-3 * o.abs().max()
    # and the random path runs (statistical smoke): different draws give different outputs
    model.backbone.stochastic_override = None
    model.decode_head.stochastic_override = None
    a = model.forward_lowres(x.cuda()).data.float()
    b = model.forward_lowres(x.cuda()).data.float()
    assert (a - b).abs().max() > 0


def test_train_loop_golden(golden_dir):
    """engine.train_one_epoch + evaluate against the reference's captured loss curve / confusion matrix."""
    import types
    from segmentation_factory_amd import engine
    g = np.load(os.path.join(golden_dir, 'train_loop_segformer_b0.npz'))
    backbone, head, nc = str(g['backbone']), str(g['head']), int(g['nc'])
    B, H, W, seed, steps, lr = int(g['B']), int(g['H']), int(g['W']), int(g['seed']), int(g['steps']), float(g['lr'])
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    model = _build(backbone, head, nc, sd, torch.float32, B)
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.0)   # torch's SGD update: plumbing, as in the golden run
    losses = []

    class Rec:
        def add_scalar(self, name, v, it=None):
            if name == 'train_loss':
                losses.append(float(v))

    class Scaler:
        def __call__(self, loss, optimizer, clip_grad=None, clip_mode='norm', parameters=None, create_graph=False):
            loss.backward()
            optimizer.step()
    args = types.SimpleNamespace(nb_classes=nc, dice=True, ignore_index=255, ignore_label=255, local_rank=0, device='cuda')
    mean_loss, _ = engine.train_one_epoch(model, opt, [(x, y)] * steps, 0, 'cuda', 1, None, None, Scaler(), Rec(), args)
    ref = g['losses']
    assert np.abs(np.array(losses) - ref).max() <= 2e-3 * np.abs(ref).max(), (losses, ref)
    confmat, metric = engine.evaluate(args, model, [(x, y)], 'cuda', 1, None)
    # the logits after 6 SGD steps differ from the reference's by ~1e-6 relative (other summation orders inside the network), so
    # pixels whose top-2 margin is below that may flip: at most 0.1 % of the 8192 pixels
    assert np.abs(confmat.mat.cpu().numpy() - g['mat']).sum() <= 0.002 * g['mat'].sum()
    assert abs(metric.compute_iou()[1] - float(g['miou'])) <= 0.1 + 1e-9      # north star: mIoU within +-0.1


@pytest.mark.parametrize('mode', ['bf16-eager', 'bf16-graph', 'fp32-eager'])
def test_train_overfit_miou_parity(golden_dir, mode):
    """The metric's second clause ("mIoU parity vs CPU ref", north star: within +-0.1) on DECISIVE logits (SURVEY Appendix D
    `train_loop_overfit`): the reference's engine.train_one_epoch ran 6 epochs x 10 AdamW steps on a learnable batch (labels =
    block pattern, colour = function of the label) and reached mIoU 99.7 on it / 98.2 on a held-out batch
    (tests/golden/train_overfit_segformer_b0.npz, oracle/make_goldens.py::train_overfit_case).  The product walks the same
    protocol through its own train_one_epoch (eager launches and --hip-graph) + evaluate in the bf16 PRODUCTION mode."""
    import types
    from segmentation_factory_amd import engine
    from segmentation_factory_amd.optim import FusedAGCAdamW, NativeScaler
    g = np.load(os.path.join(golden_dir, 'train_overfit_segformer_b0.npz'))
    backbone, head, nc = str(g['backbone']), str(g['head']), int(g['nc'])
    B, H, W, seed = int(g['B']), int(g['H']), int(g['W']), int(g['seed'])
    epochs, per_epoch, lr, wd = int(g['epochs']), int(g['per_epoch']), float(g['lr']), float(g['wd'])
    dtype = torch.bfloat16 if mode.startswith('bf16') else torch.float32
    sd = OW.make_state_dict(backbone, head, nc, seed, lively=True)
    x, y = OW.learnable_batch(B, H, W, nc, seed)
    xv, yv = OW.learnable_batch(B, H, W, nc, seed + 1)
    model = _build(backbone, head, nc, sd, dtype, B)
    opt = FusedAGCAdamW(model.parameters(), lr=lr, weight_decay=wd)     # torch.optim.AdamW(model.parameters()) arithmetic, no clipping
    losses = []

    class Rec:
        def add_scalar(self, name, v, it=None):
            if name == 'train_loss':
                losses.append(float(v))
    args = types.SimpleNamespace(nb_classes=nc, dice=True, ignore_index=255, ignore_label=255, local_rank=0, device='cuda',
                                 hip_graph=mode.endswith('graph'))
    for ep in range(epochs):
        engine.train_one_epoch(model, opt, [(x, y)] * per_epoch, ep, 'cuda', 1, None, None, NativeScaler(), Rec(), args)
    ref = g['losses']
    losses = np.array(losses)
    assert losses.shape == ref.shape
    fp32 = dtype == torch.float32
    # early steps (before rounding differences compound through 60 AdamW updates): 1e-2 relative in bf16, 2e-3 in fp32
    assert np.abs(losses[:5] - ref[:5]).max() <= (2e-3 if fp32 else 1e-2) * np.abs(ref[:5]).max(), (losses[:5], ref[:5])
    assert abs(losses[-1] - ref[-1]) <= 0.25 * ref[-1] + 5e-3, (losses[-1], ref[-1])
    for tag, (xe, ye) in (('train', (x, y)), ('heldout', (xv, yv))):
        confmat, metric = engine.evaluate(args, model, [(xe, ye)], 'cuda', 1, None)
        miou, ref_miou = metric.compute_iou()[1], float(g[f'miou_{tag}'])
        print(f'[{mode}] {tag}: mIoU {miou} (reference {ref_miou}), final loss {losses[-1]:.4f} (reference {ref[-1]:.4f})')
        assert abs(miou - ref_miou) <= 0.1 + 1e-9, (tag, miou, ref_miou)                    # north star: mIoU within +-0.1
        assert confmat.mat.sum().item() == int(g[f'mat_{tag}'].sum())                        # same valid-pixel count


def _oracle_fwd_bwd(backbone, head, nc, x, y, sd, H, W):
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('running_mean', 'running_var')))
           for k, v in sd.items()}
    o, _ = ON.model_forward(sdg, x, backbone, head, training=True, lowres=True)
    up = torch.nn.functional.interpolate(o, size=(H, W), mode='bilinear', align_corners=False)
    ref_loss = OL.criterion_closed_form(up, y, None, num_classes=nc, dice=True, ignore_index=255)
    ref_loss.backward()
    return o.detach(), ref_loss.item(), {k: v.grad for k, v in sdg.items() if v.grad is not None}


def _autocast_comparator(backbone, head, nc, x, y, sd, H, W, ref_grads):
    """Per-tensor gradient error of the CPU oracle run under torch.autocast(bfloat16) against the fp32 oracle: what autocast arithmetic
    (bf16 matmul / conv operands, fp32 accumulation, fp32 normalisation / softmax) -- the regime the reference trains in,
    /root/reference/engine.py:40 -- costs on this graph.  The yardstick for the HIP bf16 gradient errors."""
    sdc = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('running_mean', 'running_var'))) for k, v in sd.items()}
    with torch.autocast('cpu', dtype=torch.bfloat16):
        o, _ = ON.model_forward(sdc, x, backbone, head, training=True, lowres=True)
        up = torch.nn.functional.interpolate(o.float(), size=(H, W), mode='bilinear', align_corners=False)
        loss = OL.criterion_closed_form(up, y, None, num_classes=nc, dice=True, ignore_index=255)
    loss.backward()
    gmax = max(r.abs().max().item() for r in ref_grads.values())
    return {k: (sdc[k].grad.float() - r).abs().max().item() / (r.abs().max().item() + 0.05 * gmax)
            for k, r in ref_grads.items() if sdc[k].grad is not None}


def _grad_errors(model, ref_grads):
    params = dict(model.named_parameters())
    gmax = max(r.abs().max().item() for r in ref_grads.values())
    return {k: (params[k].grad.float().cpu() - r).abs().max().item() / (r.abs().max().item() + 0.05 * gmax)
            for k, r in ref_grads.items() if k in params and params[k].grad is not None}


def _grad_report(model, ref_grads, skip=()):
    """worst over parameter tensors of max|g - r| / (max|r| + 0.05 * global max|r|); returns (worst, name, n_compared)."""
    params = dict(model.named_parameters())
    gmax = max(r.abs().max().item() for r in ref_grads.values())
    worst, wname, n = 0.0, '', 0
    for k, r in ref_grads.items():
        if any(sk in k for sk in skip):
            continue
        g = params[k].grad
        assert g is not None, k
        e = (g.float().cpu() - r).abs().max().item() / (r.abs().max().item() + 0.05 * gmax)
        n += 1
        if e > worst:
            worst, wname = e, k
    return worst, wname, n


FULL_SIZE = {   # BASELINE.json configs as the reference builds them, at their full spatial size (batch: what the CPU oracle finishes in ~1 min)
    'cfg1': ('MobileNetV2', 'FPNHead', 21, 2, 256, 256),     # BASELINE cfg1 exactly: VOC 21 classes, 256 x 256, batch 2 (the reference's CPU / fp32 configuration)
    'cfg2': ('MiT-B0', 'SegFormerHead', 150, 2, 512, 512),
    'cfg3': ('ConvNeXt', 'UPerHead', 150, 4, 512, 512),      # batch 4: PPM's scale-1 BatchNorm sees 4 values per channel (2 is degenerate)
    'cfg4': ('MiT-B2', 'SegFormerHead', 19, 1, 1024, 2048),
    'cfg5': ('convnextv2_large', 'UPerHead', 171, 4, 640, 640),    # BASELINE cfg5's model / classes / size (its fp8 option: test_fp8_*)
}


@pytest.mark.parametrize('cfg', sorted(FULL_SIZE))
def test_full_size_fp32_and_bf16_vs_oracle(cfg):
    """BASELINE shapes at full size: exact-fp32 HIP path vs the (reference-pinned) CPU oracle -- low-res logits 1e-3, loss, and
    EVERY parameter gradient; then the bf16 production mode vs the same oracle (bf16 tolerance).  cfg4 is where MiT's attention
    sees 2048 keys x head_dim 64 (KV-tiled online softmax), cfg3 where the UPerHead 3x3 convs run at 3072 -> 768 channels."""
    import time
    from segmentation_factory_amd import criterion_lowres
    backbone, head, nc, B, H, W = FULL_SIZE[cfg]
    seed = 0
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    t0 = time.time()
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    o, ref_loss, ref_grads = _oracle_fwd_bwd(backbone, head, nc, x, y, sd, H, W)
    t_oracle = time.time() - t0
    cmp_err = _autocast_comparator(backbone, head, nc, x, y, sd, H, W, ref_grads) if cfg != 'cfg1' else None
    scale = o.abs().max()
    for dtype in (torch.float32, torch.bfloat16):
        fp32 = dtype == torch.float32
        model = _build(backbone, head, nc, sd, dtype, B).train()
        lo = model.forward_lowres(x.cuda())
        loss = criterion_lowres(lo, y.cuda(), (H, W), None, num_classes=nc, dice=True, ignore_index=255)
        loss.backward()
        got = lo.nchw().float().cpu()
        e_log = ((got - o).abs().max() / scale).item()
        e_loss = abs(loss.item() - ref_loss) / abs(ref_loss)
        # bf16, cfg3: PPM's scale-1 branch is a BatchNorm over B values per channel (quirk Q16): its own parameters' gradients are
        # differences of nearly equal numbers divided by a small sigma -- checked in fp32 only
        skip = ('ppm.stages.0.',) if (cfg in ('cfg3', 'cfg5') and not fp32) else ()
        worst, wname, n = _grad_report(model, ref_grads, skip)
        print(f'[{cfg} {str(dtype)[6:]}] oracle {t_oracle:.0f} s; logits {e_log:.2e}, loss {e_loss:.2e}, worst gradient error {worst:.3e} ({wname}), {n} tensors')
        assert n >= 50
        # cfg1 is the reference's fp32 configuration: fp32 is the pinned mode.  In bf16 MobileNetV2's 52 train-mode BatchNorms over
        # (near-)dead ReLU6 channels of the random-weight fixture amplify the storage rounding by 1/sigma (see
        # test_e2e_against_reference_golden): bf16 is a sanity bound there, not a parity claim
        assert e_log <= (1e-3 if fp32 else (0.6 if cfg == 'cfg1' else 6e-2))
        assert e_loss <= (1e-4 if fp32 else (6e-2 if cfg == 'cfg1' else 2e-2))
        if cfg == 'cfg1':
            assert not fp32 or worst <= 3e-2, (wname, worst)
            # quirk Q3: FPNHead.output_convs[0] is never evaluated (heads/fpn.py:29-36): no gradient reaches it
            assert all(p.grad is None for k, p in model.named_parameters() if 'output_convs.0.' in k)
            del model, lo, loss
            torch.cuda.empty_cache()
            continue
        # Bars = the worst tensor of tools/grad_parity_fullsize.py on the MI355X (profiles/r03b_grad_parity_fullsize.txt) x ~1.5, named:
        #   fp32: cfg2 2.0e-3; cfg3 1.4e-2 = decode_head.ppm.bottleneck.0.weight (PPM's scale-1 branch feeds a BatchNorm with B samples
        #         per channel, quirk Q16, whose Jacobian is ill-conditioned: its neighbours move at the 1e-2 level with the summation
        #         order; the next tensor is at 2.7e-3); cfg4 4.4e-4; cfg5 6.8e-3 = the same tensor
        #   bf16: cfg2 3.1e-2; cfg3 0.135 = decode_head.ppm.stages.1.1.0.weight (the 2 x 2 pooled branch: BatchNorm over 16 values
        #         per channel), then 0.134 backbone.downsample_layers.1.0.weight, median tensor 0.048 (ppm.stages.0.* = 0.29 is the
        #         skipped scale-1 branch); cfg4 7.5e-3; cfg5 0.275 = backbone.stages.3.1.grn.gamma (a [1,1,1,6144] GRN scale whose
        #         gradient is a sum of 400 products of bf16 activations), then 0.25 = the pwconv weights of stages 1 / 2, median 0.065
        #         (ppm.stages.0.* = 0.51 skipped)
        assert worst <= {'cfg2': (5e-3, 6e-2), 'cfg3': (3e-2, 0.2), 'cfg4': (2e-3, 2e-2), 'cfg5': (1.5e-2, 0.4)}[cfg][0 if fp32 else 1], (wname, worst)
        if not fp32:
            # CALIBRATION of the bf16 bars: every tensor's error against the fp32 oracle is compared with the error the
            # oracle ITSELF shows under CPU autocast(bfloat16), same tensor, same normalisation (comparator floored at 1e-3 so that tensors
            # autocast happens to get exactly do not divide by ~0).  Measured on the MI355X (profiles/r04_grad_parity_calibrated.txt):
            # median ratio 0.79 / 0.88 / 0.59 / 0.81 (cfg2 / 3 / 4 / 5) -- the HIP path is the more accurate of the two -- 90th percentile
            # 1.07 - 1.13, worst single tensor 1.9 (a bias / BatchNorm-bias vector), 0 - 7 tensors of 189 - 422 above 1.5
            errs = _grad_errors(model, ref_grads)
            ratios = sorted((e / max(cmp_err[k], 1e-3), k) for k, e in errs.items() if k in cmp_err)
            med, p90, top = ratios[len(ratios) // 2][0], ratios[int(0.9 * len(ratios))][0], ratios[-1]
            above = sum(r[0] > 1.5 for r in ratios)
            print(f'[{cfg} bf16 vs CPU autocast comparator] ratio median {med:.2f}, 90th percentile {p90:.2f}, worst {top[0]:.2f} ({top[1]}), '
                  f'{above} of {len(ratios)} tensors above 1.5')
            assert med <= 1.0 and p90 <= 1.3 and top[0] <= 2.5 and above <= 0.03 * len(ratios), (med, p90, top, above)
        del model, lo, loss
        torch.cuda.empty_cache()


def test_graphed_step_matches_eager_step():
    """segmentation_factory_amd.graph.GraphedTrainStep (one hipGraph per step + fused AGC/AdamW) must walk the same
    loss curve as the eager engine.py:36-53 sequence (zero_grad, forward, criterion, backward, clip + step)."""
    from segmentation_factory_amd import criterion_lowres
    from segmentation_factory_amd.graph import GraphedTrainStep
    from segmentation_factory_amd.optim import FusedAGCAdamW, NativeScaler, param_groups_weight_decay
    backbone, head, nc, B, H, W, seed = 'MiT-B0', 'SegFormerHead', 19, 2, 64, 64, 3
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    x, y = x.cuda(), y.cuda()

    def loss_fn(model, img, lbl):
        return criterion_lowres(model.forward_lowres(img), lbl, (H, W), None, num_classes=nc, dice=True, ignore_index=255)

    curves = []
    for graphed in (False, True):
        model = _build(backbone, head, nc, sd, torch.float32, B).train()
        opt = FusedAGCAdamW(param_groups_weight_decay(model, 0.025), lr=1e-3)
        losses = []
        if graphed:
            gs = GraphedTrainStep(model, opt, loss_fn, (x, y), clip_grad=0.02, clip_mode='agc', warmup=1)
            for _ in range(4):
                losses.append(gs.step(x, y).item())
        else:
            scaler = NativeScaler()
            for _ in range(4):
                opt.zero_grad(set_to_none=True)
                loss = loss_fn(model, x, y)
                losses.append(loss.item())
                scaler(loss, opt, clip_grad=0.02, clip_mode='agc', parameters=model.parameters())
        curves.append(losses)
    assert curves[0][0] != curves[0][-1]                       # the optimizer actually moved the loss
    np.testing.assert_allclose(curves[1], curves[0], rtol=2e-5)


@pytest.mark.parametrize('graphed', [False, True])
def test_parameters_without_gradient_are_not_stepped(graphed):
    """torch.optim.AdamW -- the reference's optimizer (train_gpu.py:269) -- skips a parameter whose .grad is None: no weight
    decay, no moments, no entry in state_dict()['state'].  FPNHead.output_convs[0] (quirk Q3, heads/fpn.py:29-36: built, never
    evaluated) is such a parameter in BASELINE cfg1's model; the fused kernel must leave it alone, eager and graphed."""
    from segmentation_factory_amd import criterion_lowres
    from segmentation_factory_amd.graph import GraphedTrainStep
    from segmentation_factory_amd.optim import FusedAGCAdamW, NativeScaler, param_groups_weight_decay
    backbone, head, nc, B, H, W, seed = 'MobileNetV2', 'FPNHead', 21, 2, 64, 64, 5
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    x, y = x.cuda(), y.cuda()
    model = _build(backbone, head, nc, sd, torch.float32, B).train()
    dead = {k: p.detach().clone() for k, p in model.named_parameters() if 'output_convs.0.' in k}
    live_name = next(k for k, p in model.named_parameters() if 'output_convs.1.' in k and p.ndim == 4)
    live0 = dict(model.named_parameters())[live_name].detach().clone()
    assert len(dead) >= 2

    def loss_fn(m, img, lbl):
        return criterion_lowres(m.forward_lowres(img), lbl, (H, W), None, num_classes=nc, dice=True, ignore_index=255)

    opt = FusedAGCAdamW(param_groups_weight_decay(model, 0.1), lr=1e-2)
    if graphed:
        gs = GraphedTrainStep(model, opt, loss_fn, (x, y), clip_grad=0.02, clip_mode='agc', warmup=1)
        for _ in range(3):
            gs.step(x, y)
    else:
        scaler = NativeScaler()
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            scaler(loss_fn(model, x, y), opt, clip_grad=0.02, clip_mode='agc', parameters=model.parameters())
    torch.cuda.synchronize()
    params = dict(model.named_parameters())
    for k, v in dead.items():
        assert torch.equal(params[k].detach(), v), k                 # bit-identical: no decay, no update
    assert not torch.equal(params[live_name].detach(), live0)
    st = opt.state_dict()
    n_params = sum(len(g['params']) for g in st['param_groups'])
    assert len(st['state']) == n_params - len(dead)                  # torch lists state only for stepped parameters
    # and the same state dict loads into torch.optim.AdamW over the same parameter groups
    ref = torch.optim.AdamW(param_groups_weight_decay(model, 0.1), lr=1e-2)
    ref.load_state_dict(st)


def test_graphed_step_with_plain_autograd_plugin_head():
    """A head registered through register_head that is an ordinary nn.Module on plain autograd (its gradients arrive as .grad and
    are gathered inside the captured step): three graphed steps must equal three eager steps.  With the warm-up passes' .grad
    tensors left in place the capture recorded `grad += new` and every replay added to the running sum."""
    from segmentation_factory_amd import SegmentationModel, criterion_lowres, register_head, head_dict
    from segmentation_factory_amd.graph import GraphedTrainStep
    from segmentation_factory_amd.optim import FusedAGCAdamW, NativeScaler, param_groups_weight_decay

    class PlainHead(torch.nn.Module):
        def __init__(self, in_channels, channel, num_classes):
            super().__init__()
            self.proj = torch.nn.Linear(in_channels[0], num_classes)

        def forward(self, feats):
            f = feats[0].float().permute(0, 2, 3, 1)                 # NCHW view -> NHWC
            return self.proj(f).permute(0, 3, 1, 2)

    register_head('PlainHead', PlainHead)
    try:
        nc, B, H, W = 5, 2, 64, 64
        x, y = OW.synthetic_batch(B, H, W, nc, 9)
        x, y = x.cuda(), y.cuda()

        def loss_fn(m, img, lbl):
            return criterion_lowres(m.forward_lowres(img), lbl, (H, W), None, num_classes=nc, dice=True, ignore_index=255)

        curves, finals = [], []
        for graphed in (False, True):
            torch.manual_seed(0)
            model = SegmentationModel('MiT-B0', num_classes=nc, seg_head='PlainHead', compute_dtype=torch.float32).cuda().train()
            for mod in model.backbone.modules():
                if hasattr(mod, 'drop_prob'):
                    mod.drop_prob = 0.0
            opt = FusedAGCAdamW(param_groups_weight_decay(model, 0.025), lr=1e-3)
            losses = []
            if graphed:
                gs = GraphedTrainStep(model, opt, loss_fn, (x, y), clip_grad=0.02, clip_mode='agc', warmup=2)
                for _ in range(4):
                    losses.append(gs.step(x, y).item())
            else:
                scaler = NativeScaler()
                for _ in range(4):
                    opt.zero_grad(set_to_none=True)
                    loss = loss_fn(model, x, y)
                    losses.append(loss.item())
                    scaler(loss, opt, clip_grad=0.02, clip_mode='agc', parameters=model.parameters())
            curves.append(losses)
            finals.append(model.decode_head.proj.weight.detach().cpu().clone())
        np.testing.assert_allclose(curves[1], curves[0], rtol=2e-5)
        assert (finals[0] - finals[1]).abs().max() <= 1e-5 * finals[0].abs().max() + 1e-7
    finally:
        head_dict.pop('PlainHead', None)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_folded_head_equals_literal_head(dtype):
    """SegformerFoldedFuseFn (algebraically folded Linear->resize->concat->1x1 conv) vs the literal op order of
    heads/segformer.py:42-56 on the same HIP kernels: outputs and every parameter gradient."""
    from segmentation_factory_amd import criterion_lowres
    backbone, head, nc, B, H, W, seed = 'MiT-B0', 'SegFormerHead', 19, 2, 96, 128, 11
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    res = {}
    for fold in (False, True):
        m = _build(backbone, head, nc, sd, dtype, B).train()
        m.decode_head.fold = fold
        lo = m.forward_lowres(x.cuda())
        loss = criterion_lowres(lo, y.cuda(), (H, W), None, num_classes=nc, dice=True, ignore_index=255)
        loss.backward()
        res[fold] = (lo.data.float().cpu(), {k: p.grad.float().cpu() for k, p in m.named_parameters()})
    # fp32: the two orders differ by ~1e-6 before BatchNorm, whose 1/sigma (random-init channels have tiny variance) amplifies it
    tol = 1e-2 if dtype == torch.float32 else 4e-2
    assert (res[True][0] - res[False][0]).abs().max() <= tol * res[False][0].abs().max()
    gmax = max(g.abs().max().item() for g in res[False][1].values())
    for k, g in res[False][1].items():
        err = (res[True][1][k] - g).abs().max().item()
        assert err <= tol * (g.abs().max().item() + 0.05 * gmax), (k, err, g.abs().max().item())


@pytest.mark.parametrize('graph', [False, True])
def test_train_gpu_cli_synthetic(tmp_path, graph):
    """train_gpu.py end to end on generated data: two epochs of MiT-B0 + SegFormerHead at 64x64, checkpoint written with
    the reference's keys (train_gpu.py:354-362), auto-resume picks it up (train_gpu.py:281-307).  The command as the reference's
    README launches it -- NO extra flag -- takes the replayed-hipGraph step and the graphed eval forward; --no-hip-graph the eager launches."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / 'out'
    cmd = [sys.executable, os.path.join(root, 'train_gpu.py'), '--dataset', 'synthetic', '--data_len', '8', '--image_size', '64',
           '--nb_classes', '5', '--backbone', 'MiT-B0', '--heads', 'SegFormerHead', '--batch-size', '2', '--val_batch_size', '2',
           '--epochs', '2', '--save_weights_dir', str(out), '--writer_output', str(tmp_path), '--train_print_freq', '1',
           '--val_print_freq', '1', '--lr', '1e-3'] + ([] if graph else ['--no-hip-graph'])
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'Start training for 2 epochs' in r.stdout and 'Val_mIOU' in r.stdout
    assert ('train step captured as one hipGraph' in r.stdout) == graph, r.stdout[-2000:]
    ck = torch.load(str(out / 'MiT-B0_SegFormerHead_best_model.pth'), map_location='cpu', weights_only=False)
    assert {'model_state', 'optimizer_state', 'scheduler_state', 'best_mIoU', 'F1_Score', 'Acc', 'scaler'} <= set(ck)
    assert (out / 'model.txt').exists() and (out / 'args.txt').exists()
    r2 = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0 and 'Loading local checkpoint' in r2.stdout, r2.stdout[-2000:] + r2.stderr[-2000:]


@pytest.mark.parametrize('backbone,head', [('MiT-B2', 'SegFormerHead'), ('convnextv2_nano', 'UPerHead'), ('MiT-B1', 'UPerHead')])
def test_other_variants_fp32_vs_oracle(backbone, head):
    """Variants without a committed golden (BASELINE cfg4's MiT-B2: head_dim 64; a ConvNeXtV2 width that is not T; a MiT +
    UPerHead pairing): exact-fp32 HIP path vs the (reference-pinned) CPU oracle -- low-res logits, loss, a few gradients."""
    from segmentation_factory_amd import criterion_lowres
    nc, B, H, W, seed = 11, 2, 64, 96, 21
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('running_mean', 'running_var')))
           for k, v in sd.items()}
    o, _ = ON.model_forward(sdg, x, backbone, head, training=True, lowres=True)
    up = torch.nn.functional.interpolate(o, size=(H, W), mode='bilinear', align_corners=False)
    ref_loss = OL.criterion_closed_form(up, y, None, num_classes=nc, dice=True, ignore_index=255)
    ref_loss.backward()
    model = _build(backbone, head, nc, sd, torch.float32, B).train()
    lo = model.forward_lowres(x.cuda())
    loss = criterion_lowres(lo, y.cuda(), (H, W), None, num_classes=nc, dice=True, ignore_index=255)
    loss.backward()
    got = lo.nchw().float().cpu()
    assert (got - o.detach()).abs().max() <= 1e-3 * o.detach().abs().max()
    assert abs(loss.item() - ref_loss.item()) <= 2e-4 * abs(ref_loss.item())
    params = dict(model.named_parameters())
    gmax = max(v.grad.abs().max().item() for v in sdg.values() if v.grad is not None)
    checked = 0
    for k, v in sdg.items():
        if v.grad is None or k not in params:
            continue
        g, r = params[k].grad.float().cpu(), v.grad
        assert (g - r).abs().max().item() <= 5e-2 * (r.abs().max().item() + 0.1 * gmax), (k, (g - r).abs().max().item(), r.abs().max().item())
        checked += 1
    assert checked > 50


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_nb_classes_151_as_the_reference_cli_admits(dtype):
    """ADE20K through the reference's CLI is --nb_classes 151 (datasets/build_datasets.py:32; SURVEY 8(d) 'nc 150 and 151'): 151
    classes pad to 160 columns in the head, run the 10-tile band loss kernels with one live column in the last tile, and the fused
    BatchNorm backward with K = 160.  MiT-B0 + SegFormerHead at 128 x 128 against the CPU oracle: logits, loss, every gradient."""
    from segmentation_factory_amd import criterion_lowres
    backbone, head, nc, B, H, W, seed = 'MiT-B0', 'SegFormerHead', 151, 2, 128, 128, 33
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    assert int(y[y != 255].max()) == nc - 1                      # the last class occurs
    sdg = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith(('running_mean', 'running_var')))
           for k, v in sd.items()}
    o, _ = ON.model_forward(sdg, x, backbone, head, training=True, lowres=True)
    up = torch.nn.functional.interpolate(o, size=(H, W), mode='bilinear', align_corners=False)
    ref_loss = OL.criterion_closed_form(up, y, None, num_classes=nc, dice=True, ignore_index=255)
    ref_loss.backward()
    model = _build(backbone, head, nc, sd, dtype, B).train()
    lo = model.forward_lowres(x.cuda())
    loss = criterion_lowres(lo, y.cuda(), (H, W), None, num_classes=nc, dice=True, ignore_index=255)
    loss.backward()
    got = lo.nchw().float().cpu()
    fp32 = dtype == torch.float32
    assert got.shape == o.shape
    assert (got - o.detach()).abs().max() <= (1e-3 if fp32 else 4e-2) * o.detach().abs().max()
    assert abs(loss.item() - ref_loss.item()) <= (2e-4 if fp32 else 1e-2) * abs(ref_loss.item())
    params = dict(model.named_parameters())
    gmax = max(v.grad.abs().max().item() for v in sdg.values() if v.grad is not None)
    checked = 0
    for k, v in sdg.items():
        if v.grad is None or k not in params:
            continue
        g, r = params[k].grad.float().cpu(), v.grad
        tol = (5e-2 if fp32 else 2.5e-1) * (r.abs().max().item() + 0.1 * gmax)
        assert (g - r).abs().max().item() <= tol, (k, (g - r).abs().max().item(), r.abs().max().item())
        checked += 1
    assert checked > 50


@pytest.mark.gpu
def test_single_image_inference_matches_reference_pipeline():
    """inference.SemSeg (estimate_model.py:53-123 restated for tensors): HIP resizes + arg max against the reference's op
    sequence on this library's own full-resolution logits: F.interpolate(align_corners=True) -> softmax -> argmax."""
    import torch.nn.functional as F
    from segmentation_factory_amd import SegmentationModel
    from segmentation_factory_amd.inference import SemSeg
    torch.manual_seed(3)
    m = SegmentationModel('MiT-B0', num_classes=19, seg_head='SegFormerHead').cuda().eval()
    pal = torch.randint(0, 255, (19, 3), dtype=torch.uint8)
    ss = SemSeg(m, img_size=128, palette=pal)
    img = torch.randint(0, 256, (3, 150, 200), dtype=torch.uint8)
    assert ss.inference_size(150, 200) == (128, 192)
    seg, col = ss.predict(img, overlay=False)
    assert seg.shape == (150, 200) and seg.dtype == torch.int64 and col.shape == (150, 200, 3)
    x = ss.preprocess(img)
    with torch.inference_mode():
        logits = m(x)                                                    # fp32 NCHW at the network input size
        ref = F.interpolate(logits, size=(150, 200), mode='bilinear', align_corners=True).softmax(dim=1).argmax(dim=1)[0]
    agree = (seg == ref).float().mean().item()
    assert agree >= 0.999, agree
    assert torch.equal(col, pal.cuda()[seg])


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [((150, 200), (128, 192)), ((37, 91), (64, 160)), ((512, 683), (512, 704)), ((1024, 2048), (512, 1024))])
def test_inference_preprocess_kernel_matches_torchvision_tensor_resize(shape):
    """segf_infer_preprocess = estimate_model.py:85-97 with the pinned torchvision 0.15.2 (environment.yml:22): T.Resize of a uint8 CHW tensor
    is F.interpolate(bilinear, align_corners=False, antialias=False) on the float32 cast, rounded back to uint8; then x / 255 and
    Normalize.  Against torch's own CPU kernels, exactly (the uint8 stage) and to float32 round-off (the normalisation)."""
    import torch.nn.functional as F
    from segmentation_factory_amd import hip
    from segmentation_factory_amd.inference import IMAGENET_MEAN, IMAGENET_STD
    (h, w), (nH, nW) = shape
    g = torch.Generator().manual_seed(h * 7 + w)
    img = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)
    mean, std = torch.tensor(IMAGENET_MEAN), torch.tensor(IMAGENET_STD)
    u8 = torch.round(F.interpolate(img.float()[None], size=(nH, nW), mode='bilinear', align_corners=False, antialias=False)).to(torch.uint8)
    want = (u8.float() / 255 - mean.view(1, 3, 1, 1)) / std.view(1, 3, 1, 1)
    got = hip.infer_preprocess(img.cuda(), nH, nW, mean.cuda(), std.cuda()).cpu()
    back = torch.round((got * std.view(1, 3, 1, 1) + mean.view(1, 3, 1, 1)) * 255).to(torch.uint8)
    assert torch.equal(back, u8)                                      # every resized byte
    assert (got - want).abs().max().item() <= 1e-6


@pytest.mark.gpu
def test_argmax_rows_kernel():
    from segmentation_factory_amd import hip
    g = torch.Generator().manual_seed(4)
    for C, ld, dt in ((19, 24, torch.float32), (150, 152, torch.bfloat16), (171, 176, torch.float32), (2, 8, torch.bfloat16)):
        x = torch.randn(1000, ld, generator=g).to(dt).cuda()
        x[5, :C] = 1.0                                                   # ties -> lowest index
        out = hip.argmax_rows(x, C)
        assert torch.equal(out, x[:, :C].float().argmax(dim=1))


def test_fused_adamw_continues_a_torch_adamw_checkpoint():
    """Resuming from the REFERENCE's checkpoint dict (train_gpu.py:286-298): its 'optimizer_state' is a torch.optim.AdamW
    state_dict.  After loading it, the next fused step on the GPU must equal torch.optim.AdamW's own next step on the CPU."""
    from segmentation_factory_amd.optim import FusedAGCAdamW
    g = torch.Generator().manual_seed(21)
    shapes = [(6, 5), (7,), (3, 2, 3, 3), (4,)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    ref = torch.optim.AdamW(ps, lr=2e-4, weight_decay=0.025)
    for _ in range(3):
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g)
        ref.step()
    sd = ref.state_dict()
    mine = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ps]
    fused = FusedAGCAdamW(mine, lr=1.0, weight_decay=0.0)
    fused.load_state_dict(sd)
    grads = [torch.randn(p.shape, generator=g) for p in ps]
    for p, q, gr in zip(ps, mine, grads):
        p.grad = gr.clone()
        q.grad = gr.clone().cuda()
    ref.step()
    fused.agc_clip = 0.0
    fused.step()
    for p, q in zip(ps, mine):
        assert torch.allclose(q.detach().cpu(), p.detach(), rtol=2e-6, atol=1e-7)


def _run_cli(args, cwd, env=None, timeout=600):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, 'train_gpu.py')] + args
    e = dict(os.environ, PYTHONPATH=root)
    e.update(env or {})
    return subprocess.run(cmd, cwd=str(cwd), env=e, capture_output=True, text=True, timeout=timeout)


def test_train_gpu_cli_finetune_freeze_and_reference_checkpoint(tmp_path):
    """(f3) drop-in files: --finetune from an NVIDIA-style {'state_dict': ...} file with decode_head.conv_seg.* + --freeze_layers
    (train_gpu.py:238-260, util/utils.py:313-324): only linear_pred trains; then auto-resume from a checkpoint in the
    REFERENCE's layout (train_gpu.py:281-307,354-362: torch.optim.AdamW 'optimizer_state', the scheduler's attribute dict,
    best_mIoU / F1_Score / Acc, GradScaler 'scaler')."""
    nc = 5
    sd = OW.make_state_dict('MiT-B0', 'SegFormerHead', nc, 31)
    nvidia = dict(sd)
    nvidia['decode_head.conv_seg.weight'] = torch.zeros(150, 768, 1, 1)
    nvidia['decode_head.conv_seg.bias'] = torch.zeros(150)
    ft = tmp_path / 'segformer.b0.fake.pth'
    torch.save({'state_dict': nvidia}, str(ft))
    out = tmp_path / 'out'
    base = ['--dataset', 'synthetic', '--data_len', '8', '--image_size', '64', '--nb_classes', str(nc), '--backbone', 'MiT-B0',
            '--heads', 'SegFormerHead', '--batch-size', '2', '--val_batch_size', '2', '--epochs', '1', '--save_weights_dir', str(out),
            '--writer_output', str(tmp_path), '--train_print_freq', '1', '--val_print_freq', '1']
    r = _run_cli(base + ['--finetune', str(ft), '--hip-graph'], tmp_path)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'Removing key decode_head.linear_pred.weight from pretrained checkpoint' in r.stdout
    assert 'training decode_head.linear_pred.weight' in r.stdout and 'Number of parameters: %d' % (768 * nc + nc) in r.stdout
    ck = torch.load(str(out / 'MiT-B0_SegFormerHead_best_model.pth'), map_location='cpu', weights_only=False)
    ms = ck['model_state']
    changed = [k for k in sd if sd[k].is_floating_point() and not torch.equal(ms[k], sd[k]) and 'running_' not in k]
    assert changed and all('linear_pred' in k for k in changed), changed[:5]       # frozen everywhere else
    assert len(ck['optimizer_state']['state']) == 2                                # AdamW moments exist for the two trained tensors only
    # ---- a checkpoint in the reference's layout, written with torch's own AdamW on the CPU ----
    from segmentation_factory_amd import SegmentationModel
    from segmentation_factory_amd.optim import param_groups_weight_decay
    cpu_model = SegmentationModel('MiT-B0', num_classes=nc, seg_head='SegFormerHead')       # parameter container (same keys)
    cpu_model.load_state_dict(sd)
    opt = torch.optim.AdamW(param_groups_weight_decay(cpu_model, 0.025), lr=1e-3)           # timm create_optimizer's grouping
    g = torch.Generator().manual_seed(1)
    for p in cpu_model.parameters():
        p.grad = torch.randn(p.shape, generator=g) * 1e-3
    opt.step()
    sched_state = {'param_group_field': 'lr', '_initial_param_group_field': 'initial_lr', 'base_values': [1e-3, 1e-3], 'metric': None,
                   't_in_epochs': False, 'noise_range_t': None, 'noise_pct': 0.67, 'noise_type': 'normal', 'noise_std': 1.0,
                   'noise_seed': 0, 't_initial': 20, 'lr_min': 1e-4, 'cycle_mul': 1.0, 'cycle_decay': 1.0, 'cycle_limit': 1,
                   'warmup_t': 20, 'warmup_lr_init': 2e-4, 'warmup_prefix': False, 'k_decay': 1.0, 'warmup_steps': [4e-5, 4e-5]}
    out2 = tmp_path / 'out2'
    out2.mkdir()
    torch.save({'model_state': cpu_model.state_dict(), 'optimizer_state': opt.state_dict(), 'scheduler_state': sched_state,
                'best_mIoU': 12.34, 'F1_Score': 23.45, 'Acc': 34.56,
                'scaler': {'scale': 65536.0, 'growth_factor': 2.0, 'backoff_factor': 0.5, 'growth_interval': 2000, '_growth_tracker': 0}},
               str(out2 / 'reference_format.pth'))
    base2 = [a if a != str(out) else str(out2) for a in base]
    r2 = _run_cli(base2 + ['--hip-graph'], tmp_path)
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    assert 'Loading local checkpoint' in r2.stdout and 'Now max mIOU is 12.34' in r2.stdout and '<All keys matched successfully>' in r2.stdout


def test_two_rank_eager_ddp_finetune_freeze(tmp_path):
    """DistributedDataParallel must be built AFTER the --finetune load / --freeze_layers and with
    find_unused_parameters=True (train_gpu.py:233-260), or the second iteration raises 'Expected to have finished reduction'.
    Two ranks (gloo, sharing this box's one GPU; fresh child processes), eager launches, two iterations per rank."""
    import subprocess
    import sys
    nc = 5
    sd = OW.make_state_dict('MiT-B0', 'SegFormerHead', nc, 32)
    ft = tmp_path / 'pretrained.pth'
    torch.save(dict(sd), str(ft))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ['--dataset', 'synthetic', '--data_len', '8', '--image_size', '64', '--nb_classes', str(nc), '--backbone', 'MiT-B0',
            '--heads', 'SegFormerHead', '--batch-size', '2', '--val_batch_size', '2', '--epochs', '1', '--save_weights_dir', '',
            '--writer_output', str(tmp_path), '--train_print_freq', '1', '--val_print_freq', '1', '--finetune', str(ft), '--no-hip-graph']
    env = dict(os.environ, PYTHONPATH=root, MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', WORLD_SIZE='2',
               SEGFAC_DIST_BACKEND='gloo')
    procs = [subprocess.Popen([sys.executable, os.path.join(root, 'train_gpu.py')] + args, cwd=str(tmp_path),
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-3000:] + outs[1][-2000:]
    assert 'Epoch: [0]  [1/2]' in outs[0] and 'Val_mIOU' in outs[0]


def test_two_rank_graph_mode_cli(tmp_path):
    """`train_gpu.py` with NO extra flag under two ranks (gloo, sharing this box's GPU): one epoch of the graphed step with the bucketed
    exchange, then `evaluate` with the C2 buffer broadcast, the graphed eval forward and the C4 - C6 reductions
    (train_gpu.py:211-236,322-336; util/utils.py:125-131; util/metrics.py:108-114).  Both ranks must print the same validation line."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ['--dataset', 'synthetic', '--data_len', '8', '--image_size', '64', '--nb_classes', '5', '--backbone', 'MiT-B0',
            '--heads', 'SegFormerHead', '--batch-size', '2', '--val_batch_size', '2', '--epochs', '1', '--save_weights_dir', '',
            '--writer_output', str(tmp_path), '--train_print_freq', '1', '--val_print_freq', '1', '--lr', '1e-3']
    env = dict(os.environ, PYTHONPATH=root, MASTER_ADDR='127.0.0.1', MASTER_PORT='29579', WORLD_SIZE='2',
               SEGFAC_DIST_BACKEND='gloo', SEGFAC_PRINT_ALL_RANKS='1')
    procs = [subprocess.Popen([sys.executable, os.path.join(root, 'train_gpu.py')] + args, cwd=str(tmp_path),
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-3000:] + outs[1][-2000:]
    assert 'Epoch: [0]  [1/2]' in outs[0] and 'Val_mIOU' in outs[0]
    assert 'train step captured as one hipGraph' in outs[0]          # the default command took the graph path
    vals = [re.findall(r'Val_mIOU[^\n]*', o) for o in outs]
    assert vals[0] and vals[0] == vals[1], vals                     # the reduced matrices (hence every printed figure) agree


@pytest.mark.parametrize('exchange,payload', [('all_reduce', 'fp32'), ('rs_ag', 'fp32'), ('all_reduce', 'bf16')])
def test_one_rank_rccl_exchange_leg(tmp_path, exchange, payload):
    """The RCCL leg of the data-parallel step on ONE GPU: a fresh child process with a 1-rank `nccl` process group and the
    exchange forced on drives GraphedTrainStep for three steps -- graph replay, in-graph external event nodes, the communication
    stream waiting on them, the RCCL collective(s) on each aligned bucket range, the optimizer behind the communication stream
    (train_gpu.py:211-236's DDP exchange).  A sum over one rank is the identity, so the fp32 modes must reproduce the
    no-exchange run of the same step in this process exactly; the bf16 payload rounds every gradient once (rel. 2^-9), which the
    loss curve must tolerate at 1e-3."""
    import subprocess
    import sys
    from segmentation_factory_amd import criterion_lowres
    from segmentation_factory_amd.graph import GraphedTrainStep
    from segmentation_factory_amd.optim import FusedAGCAdamW, param_groups_weight_decay
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    steps = 3
    out = tmp_path / 'rank0.pt'
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29591', WORLD_SIZE='1', RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'dp_worker.py'), str(out), str(steps), '4.0', 'nccl', exchange, payload],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = torch.load(str(out), map_location='cpu', weights_only=False)
    assert got['backend'] == 'nccl' and got['exchanging'] and got['n_buckets'] >= 4 and got['events'] == got['n_buckets']
    assert all((hi - lo) % 16 == 0 for lo, hi in got['ranges'])
    backbone, head, nc, B, H, W, seed = 'MiT-B0', 'SegFormerHead', 19, 2, 64, 64, 17
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.synthetic_batch(B, H, W, nc, seed)
    x, y = x.cuda(), y.cuda()
    model = _build(backbone, head, nc, sd, torch.float32, B).train()
    opt = FusedAGCAdamW(param_groups_weight_decay(model, 0.025), lr=1e-3)

    def loss_fn(m, img, lbl):
        return criterion_lowres(m.forward_lowres(img), lbl, (H, W), None, num_classes=nc, dice=True, ignore_index=255)
    gs = GraphedTrainStep(model, opt, loss_fn, (x, y), clip_grad=0.02, clip_mode='agc', warmup=1)
    assert not gs.exchanging
    losses = [gs.step(x, y).item() for _ in range(steps)]
    ref = model.state_dict()
    if payload == 'fp32':
        assert got['losses'] == losses
        for k, v in got['state'].items():
            assert torch.equal(v, ref[k].detach().cpu()), k
    else:
        np.testing.assert_allclose(got['losses'], losses, rtol=1e-3)


@pytest.mark.parametrize('bucket_mb,exchange,mode', [(4.0, 'all_reduce', ''), (1000.0, 'all_reduce', ''), (4.0, 'rs_ag', ''),
                                                     (4.0, 'all_reduce', 'spin'), (4.0, 'all_reduce', 'evalsync')])
def test_two_rank_graphed_step_matches_manual_data_parallel(tmp_path, bucket_mb, exchange, mode):
    """D1 / collective C1 on the PRODUCT step: two ranks (fresh child processes, gloo, sharing this box's GPU) each drive
    GraphedTrainStep on their shard -- hipGraph replay, per-bucket external events, all-reduce on the communication stream,
    fused AGC/AdamW -- for three steps.  Expected values: data parallelism by hand in this process = per-shard forward /
    backward on the SAME weights (BatchNorm statistics per shard, as per rank: the reference uses plain BatchNorm2d), mean of
    the two gradients, one optimizer step (train_gpu.py:233-236 DistributedDataParallel semantics).  bucket_mb=4: six buckets
    with in-graph events; 1000: a single bucket.

    mode 'spin' = the DETERMINISTIC ordering test: a 20 ms spin kernel is captured in front of the LAST gradient of every bucket
    (SEGFAC_TEST_SPIN_US, graph.py), so a communication stream whose `ev.wait()` saw an older record of the bucket's event than THIS
    replay's would start its collective 20 ms before that gradient exists and exchange the previous step's values -- the result must
    still equal the manual average.  mode 'evalsync' = collective C2 on the graph path: after the three steps the ranks hold
    different BatchNorm statistics; `engine.evaluate` (with its forward replayed as a hipGraph) must leave every rank with rank 0's
    and return the matrices of rank 0's model over both shards."""
    import subprocess
    import sys
    from segmentation_factory_amd import criterion_lowres
    from segmentation_factory_amd.optim import FusedAGCAdamW, param_groups_weight_decay
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    steps, world, per_rank = 3, 2, 2
    out = tmp_path / 'rank0.pt'
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29583', WORLD_SIZE=str(world))
    if mode == 'spin':
        env['SEGFAC_TEST_SPIN_US'] = '20000'
    elif mode:
        env['DP_MODE'] = mode
    procs = [subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'dp_worker.py'), str(out), str(steps), str(bucket_mb), 'gloo', exchange],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][-3000:] + outs[1][-3000:]
    got = torch.load(str(out), map_location='cpu', weights_only=False)
    if bucket_mb < 100:
        assert got['n_buckets'] >= 4 and got['events'] == got['n_buckets']
    else:
        assert got['n_buckets'] == 1
    # ---- the same three steps by hand -------------------------------------------------------------------------------
    backbone, head, nc, H, W, seed = 'MiT-B0', 'SegFormerHead', 19, 64, 64, 17
    sd = OW.make_state_dict(backbone, head, nc, seed)                       # rank 0's initial weights
    x, y = OW.synthetic_batch(per_rank * world, H, W, nc, seed)
    models = [_build(backbone, head, nc, sd, torch.float32, per_rank).train() for _ in range(world)]
    opt = FusedAGCAdamW(param_groups_weight_decay(models[0], 0.025), lr=1e-3)
    opt.agc_clip = 0.02
    losses = []
    for _ in range(steps):
        grads = []
        for r, m in enumerate(models):
            m.load_state_dict({k: v for k, v in models[0].state_dict().items() if 'running_' not in k and 'num_batches' not in k},
                              strict=False)                                  # same weights, own BatchNorm buffers
            for p in m.parameters():
                p.grad = None
            xs, ys = x[r * per_rank:(r + 1) * per_rank].cuda(), y[r * per_rank:(r + 1) * per_rank].cuda()
            loss = criterion_lowres(m.forward_lowres(xs), ys, (H, W), None, num_classes=nc, dice=True, ignore_index=255)
            loss.backward()
            grads.append([p.grad.clone() for p in m.parameters()])
            if r == 0:
                losses.append(loss.item())
        for p, g0, g1 in zip(models[0].parameters(), grads[0], grads[1]):
            p.grad = (g0 + g1) / world
        opt.step()
    np.testing.assert_allclose(got['losses'], losses, rtol=2e-5)
    ref = models[0].state_dict()
    worst = 0.0
    for k, v in got['state'].items():
        r = ref[k].detach().cpu()
        if v.is_floating_point():
            worst = max(worst, ((v - r).abs().max() / (r.abs().max() + 1e-6)).item())
        else:
            assert torch.equal(v, r), k
    assert worst <= 2e-5, worst
    if mode == 'evalsync':
        from segmentation_factory_amd.metrics import Metrics
        from segmentation_factory_amd.utils import ConfusionMatrix
        assert got['buffers_differed_before'] and got['buffers_equal_after'] and got['eval_graphs'] == 1
        # rank 0's model (weights AND BatchNorm statistics) over both ranks' validation batches
        m0 = models[0].eval()
        metric, confmat = Metrics(nc, 255, 'cuda'), ConfusionMatrix(nc)
        with torch.inference_mode():
            for r in range(world):
                xs, ys = x[r * per_rank:(r + 1) * per_rank].cuda(), y[r * per_rank:(r + 1) * per_rank].cuda()
                for xv, yv in ((xs, ys), (xs.flip(3).contiguous(), ys.flip(2).contiguous())):
                    metric.update_lowres(m0.forward_lowres(xv), yv, (H, W), confmat=confmat)
        total = metric.hist.sum().item()
        assert got['hist'].sum().item() == total and got['mat'].sum().item() == confmat.mat.sum().item()
        # weights agree to 2e-5, so a handful of near-tie pixels may fall the other way; rank-local statistics would move thousands
        assert (got['hist'] - metric.hist.cpu()).abs().sum().item() <= 0.002 * total
        m1 = models[1].eval()                      # control: rank 1's own statistics give a visibly different matrix
        metric1 = Metrics(nc, 255, 'cuda')
        with torch.inference_mode():
            for r in range(world):
                xs, ys = x[r * per_rank:(r + 1) * per_rank].cuda(), y[r * per_rank:(r + 1) * per_rank].cuda()
                for xv, yv in ((xs, ys), (xs.flip(3).contiguous(), ys.flip(2).contiguous())):
                    metric1.update_lowres(m1.forward_lowres(xv), yv, (H, W))
        print('evalsync: |hist - rank0-stat hist| =', (got['hist'] - metric.hist.cpu()).abs().sum().item(),
              ' |rank1-stat hist - rank0-stat hist| =', (metric1.hist - metric.hist).abs().sum().item())


FP8_GRAD_BAR = 0.8          # measured 0.551 (backbone.stages.0.0.grn.gamma; bf16 on the same tensor family: 0.335) x 1.45


def test_fp8_forward_of_cfg5_model_within_stated_tolerance():
    """BASELINE cfg5 "fp8 MFMA weights": convnextv2_large + UPerHead, 171 classes, 640 x 640, batch 4 with set_fp8() -- all three
    products of the UPerHead / PPM 3x3 convolutions and of the stage-3 block MLPs on fp8 operands (forward e4m3 x e4m3, data and weight
    gradients with the gradient in e5m2) -- against the fp32 CPU oracle.  The reference has no fp8: the bars are stated tolerances --
    logits within 0.25 of their scale, loss within 3 % (measured 0.14 / 5e-5; bf16 alone: 4e-2 / 3e-5,
    test_full_size_fp32_and_bf16_vs_oracle), gradients finite, and every parameter gradient within FP8_GRAD_BAR of its tensor's scale
    (the bf16 bar of the same model is 0.4; ppm.stages.0.* -- a BatchNorm over 4 values per channel -- skipped as there)."""
    from segmentation_factory_amd import criterion_lowres
    backbone, head, nc, B, H, W = 'convnextv2_large', 'UPerHead', 171, 4, 640, 640
    sd = OW.make_state_dict(backbone, head, nc, 0)
    x, y = OW.synthetic_batch(B, H, W, nc, 0)
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    o, ref_loss, ref_grads = _oracle_fwd_bwd(backbone, head, nc, x, y, sd, H, W)
    model = _build(backbone, head, nc, sd, torch.bfloat16, B).train()
    model.set_fp8(True)
    lo = model.forward_lowres(x.cuda())
    loss = criterion_lowres(lo, y.cuda(), (H, W), None, num_classes=nc, dice=True, ignore_index=255)
    loss.backward()
    got = lo.nchw().float().cpu()
    e_log = ((got - o).abs().max() / o.abs().max()).item()
    e_loss = abs(loss.item() - ref_loss) / abs(ref_loss)
    worst, wname, n = _grad_report(model, ref_grads, ('ppm.stages.0.',))
    print(f'[cfg5 fp8] logits {e_log:.2e}, loss {e_loss:.2e}, worst gradient error {worst:.3e} ({wname}), {n} tensors')
    assert e_log <= 0.25 and e_loss <= 3e-2
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())
    assert worst <= FP8_GRAD_BAR, (wname, worst)
    with pytest.raises(ValueError):
        _build('MiT-B0', 'SegFormerHead', 19, OW.make_state_dict('MiT-B0', 'SegFormerHead', 19, 0), torch.bfloat16, 1).set_fp8(True)


@pytest.mark.parametrize('geom', [('ConvNeXt', 19, 4, 256, 256, 30), ('convnextv2_large', 171, 4, 640, 640, 20)], ids=['convnext_t_256', 'cfg5_width_640'])
def test_fp8_training_curve_tracks_bf16(geom):
    """Overfit-style check of the fp8 option (BASELINE cfg5 'fp8 MFMA weights'; no counterpart in the reference): ConvNeXt-T + UPerHead,
    19 classes, 256 x 256, batch 4 -- large enough for every UPerHead 3x3 convolution (all three directions) to take its fp8 kernel --
    and BASELINE cfg5's own model (convnextv2_large + UPerHead, 171 classes, 640 x 640, batch 4: 6400 stage-3 tokens, so the block MLPs
    run on fp8 operands in all three products too), trained on one batch with the fused AGC / AdamW step, once in bf16 and once with
    set_fp8().  Both must learn, and the fp8 curve must track the bf16 curve step by step: quantisation noise, not a different
    optimisation trajectory."""
    from segmentation_factory_amd import criterion_lowres
    from segmentation_factory_amd.optim import FusedAGCAdamW, NativeScaler, param_groups_weight_decay
    backbone, nc, B, H, W, nsteps = geom
    head, seed = 'UPerHead', 23
    sd = OW.make_state_dict(backbone, head, nc, seed)
    x, y = OW.learnable_batch(B, H, W, nc, seed, block=32)          # labels a function of the colours: fittable in a few steps
    x, y = x.cuda(), y.cuda()
    curves = {}
    for fp8 in (False, True):
        model = _build(backbone, head, nc, sd, torch.bfloat16, B).train()
        if fp8:
            model.set_fp8(True)
        opt = FusedAGCAdamW(param_groups_weight_decay(model, 0.025), lr=2e-4)
        scaler = NativeScaler()
        losses = []
        for _ in range(nsteps):
            opt.zero_grad(set_to_none=True)
            loss = criterion_lowres(model.forward_lowres(x), y, (H, W), None, num_classes=nc, dice=True, ignore_index=255)
            losses.append(loss.item())
            scaler(loss, opt, clip_grad=0.02, clip_mode='agc', parameters=model.parameters())
        curves[fp8] = losses
        del model, opt
        torch.cuda.empty_cache()
    print('bf16', [round(v, 4) for v in curves[False]])
    print('fp8 ', [round(v, 4) for v in curves[True]])
    dev = [abs(a - b) / b for a, b in zip(curves[True], curves[False])]
    print('max deviation', max(dev), 'last ten', max(dev[-10:]), 'bf16 loss ratio end / start', curves[False][-1] / curves[False][0])
    for c in curves.values():
        assert all(np.isfinite(c)) and c[-1] < (0.05 if backbone == 'ConvNeXt' else 0.6) * c[0], c
    # ConvNeXt-T: measured 8.5 % at step 4 (loss falling 4x per 2 steps), <= 0.6 % at the end
    assert max(dev) <= 0.12 and max(dev[-10:]) <= (0.015 if backbone == 'ConvNeXt' else 0.05), dev


def test_default_step_is_the_graph_and_a_failed_capture_falls_back_to_eager(capsys):
    """engine.train_one_epoch with args that carry NO hip_graph attribute (what a caller written against the reference passes,
    engine.py:18-70): the step is the replayed hipGraph.  A model whose forward cannot be captured (stand-in: a plugin op that
    refuses to run while the stream is capturing) makes the same call print the reason and train with eager launches -- in this
    process, with the optimizer's direct gradient placement disarmed -- while args.hip_graph=True turns the failure into the error."""
    import types
    from segmentation_factory_amd import engine
    from segmentation_factory_amd.optim import FusedAGCAdamW, NativeScaler
    backbone, head, nc, B, H, W, seed = 'MiT-B0', 'SegFormerHead', 5, 2, 64, 64, 8
    sd = OW.make_state_dict(backbone, head, nc, seed, lively=True)
    x, y = OW.learnable_batch(B, H, W, nc, seed)
    args = types.SimpleNamespace(nb_classes=nc, dice=True, ignore_index=255, ignore_label=255, local_rank=0, device='cuda')

    def run(model, a, steps=6):
        opt = FusedAGCAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
        losses = []

        class Rec:
            def add_scalar(self, name, v, it=None):
                if name == 'train_loss':
                    losses.append(float(v))
        engine.train_one_epoch(model, opt, [(x, y)] * steps, 0, 'cuda', 1, None, None, NativeScaler(), Rec(), a)
        return losses, opt

    good = _build(backbone, head, nc, sd, torch.float32, B)
    l_graph, _ = run(good, args)
    assert getattr(good, '_graphed_step', None) is not None and 'captured as one hipGraph' in capsys.readouterr().out

    def refuse_capture(model):
        real = model.forward_lowres

        def fwd(img):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('plugin op: not capturable')
            return real(img)
        model.forward_lowres = fwd
        return model

    bad = refuse_capture(_build(backbone, head, nc, sd, torch.float32, B))
    l_eager, opt = run(bad, args)
    out = capsys.readouterr().out
    assert 'hipGraph capture of the train step failed' in out and 'not capturable' in out and 'eager launches' in out
    assert getattr(bad, '_graphed_step', None) is None and bad._graph_disabled and not opt.direct
    assert all(not hasattr(p, '_segf_grad') for p in bad.parameters())
    # same arithmetic either way (fp32 mode): the fallback trained the model the graph would have trained
    np.testing.assert_allclose(l_eager, l_graph, rtol=2e-4)
    assert l_eager[-1] < l_eager[0]
    l_again, _ = run(bad, args, steps=2)                    # later epochs stay eager without a second attempt
    assert 'capture of the train step failed' not in capsys.readouterr().out and len(l_again) == 2

    bad2 = refuse_capture(_build(backbone, head, nc, sd, torch.float32, B))
    with pytest.raises(RuntimeError, match='not capturable'):
        run(bad2, types.SimpleNamespace(**vars(args), hip_graph=True))


def test_graphed_eval_session_capture_policy():
    """validation images keep their aspect ratio (datasets/build_datasets.py:24-29) at --val_batch_size 1
    (train_gpu.py:72), so `evaluate` meets MANY shapes.  A shape is captured on its SECOND sighting, a session captures at most
    `max_captures` graphs and holds `max_graphs` (least recently used goes first); 24 distinct shapes, each seen once, run eagerly
    without a single capture.  Results equal the eager forward bit for bit throughout."""
    from segmentation_factory_amd.graph import GraphedEvalSession
    backbone, head, nc, seed = 'MiT-B0', 'SegFormerHead', 5, 9
    sd = OW.make_state_dict(backbone, head, nc, seed)
    model = _build(backbone, head, nc, sd, torch.bfloat16, 1).eval()
    g = torch.Generator().manual_seed(0)
    shapes = [(64 + 4 * i, 96 + 8 * (i % 5)) for i in range(24)]
    assert len(set(shapes)) == 24
    with torch.inference_mode():
        sess = GraphedEvalSession(model, max_graphs=3, max_captures=4)
        for h, w in shapes:                                   # every shape once: nothing is captured
            sess(torch.randn(1, 3, h, w, generator=g).cuda())
        assert (sess.captures, sess.replays, sess.eager) == (0, 0, 24)
        xs = [torch.randn(1, 3, h, w, generator=g).cuda() for h, w in shapes[:6]]
        for x in xs[:4]:                                      # second sighting of four shapes: four captures (the session's cap) ...
            assert torch.equal(sess(x).data, model.forward_lowres(x).data)
        assert sess.captures == 4 and len(sess.graphs) == 3   # ... of which the three most recent are kept
        assert torch.equal(sess(xs[3]).data, model.forward_lowres(xs[3]).data) and sess.replays == 1
        n_eager = sess.eager
        assert torch.equal(sess(xs[4]).data, model.forward_lowres(xs[4]).data)     # cap reached: a recurring shape stays eager
        assert sess.captures == 4 and sess.eager == n_eager + 1
        # the next session (next epoch's evaluate) finds graphs and sighting counts on the model
        sess2 = GraphedEvalSession(model, max_graphs=3, max_captures=4)
        assert torch.equal(sess2(xs[2]).data, model.forward_lowres(xs[2]).data) and sess2.replays == 1
        assert torch.equal(sess2(xs[5]).data, model.forward_lowres(xs[5]).data) and sess2.captures == 1
        # another compute dtype is another graph (evaluate runs fp32 by default, --eval-dtype bf16 on request)
        model.set_compute_dtype(torch.float32)
        a = sess2(xs[2])
        assert a.data.dtype == torch.float32 and sess2.replays == 1
        model.set_compute_dtype(torch.bfloat16)


def test_capture_collects_dead_graphs_first_and_holds_the_collector_off():
    """A torch CUDAGraph's destructor synchronises the device on ROCm -- illegal inside a capture, and an exception out of a
    destructor aborts the process.  A dead graph kept alive only by a reference cycle (a dropped model <-> its graphed step) must
    therefore be freed BEFORE the next capture starts, and the cyclic collector must not run during it (graph._capture)."""
    import gc
    import weakref
    from segmentation_factory_amd.graph import GraphedEvalForward
    model = _build('MiT-B0', 'SegFormerHead', 5, OW.make_state_dict('MiT-B0', 'SegFormerHead', 5, 9), torch.bfloat16, 1).eval()
    x = torch.randn(1, 3, 64, 96).cuda()
    seen = []
    orig = model.forward_lowres

    def spying(inp):
        seen.append(gc.isenabled())
        return orig(inp)
    with torch.inference_mode():
        old = GraphedEvalForward(model, x)
        ring = [old]
        ring.append(ring)                       # only the cyclic collector can free `old` now
        dead = weakref.ref(old)
        del old, ring
        model.forward_lowres = spying
        try:
            new = GraphedEvalForward(model, x)   # warm-up call (collector on), captured call (collector off)
        finally:
            del model.forward_lowres
        assert dead() is None and seen == [True, False] and gc.isenabled()
        assert torch.equal(new(x).data, model.forward_lowres(x).data)


@pytest.mark.parametrize('family', ['segformer', 'convnext_uper', 'mbv2_fpn'])
def test_evaluate_batch1_over_odd_sizes_against_oracle(family):
    """What `evaluate` sees in a real run (datasets/build_datasets.py:24-29: ExtResize keeps the aspect ratio; train_gpu.py:72:
    --val_batch_size 1): three images of three different sizes, none a multiple of 32, each size twice, in ONE session (default
    arguments: fp32 forward, graph per recurring shape).  Confusion matrix and histogram against the CPU oracle's on the same
    weights and inputs: logits agree to ~1e-6, so at most a handful of near-tie pixels may flip."""
    import types
    from segmentation_factory_amd import engine
    backbone, head, nc = {'segformer': ('MiT-B0', 'SegFormerHead', 7), 'convnext_uper': ('ConvNeXt', 'UPerHead', 9),
                          'mbv2_fpn': ('MobileNetV2', 'FPNHead', 6)}[family]
    seed = 41
    sd = OW.make_state_dict(backbone, head, nc, seed, lively=True)
    sizes = [(75, 100), (90, 123), (67, 131)]
    batches = []
    for i, (h, w) in enumerate(sizes * 2):
        x, y = OW.learnable_batch(1, h, w, nc, seed + i)
        batches.append((x, y))
    model = _build(backbone, head, nc, sd, torch.bfloat16, 1)          # trained-in-bf16 model; evaluate switches to fp32 like the reference
    args = types.SimpleNamespace(nb_classes=nc, ignore_label=255)
    confmat, metric = engine.evaluate(args, model, batches, 'cuda', 10, None)
    assert model.compute_dtype == torch.bfloat16                       # restored
    sess = model.__dict__['_graphed_eval']
    assert len(sess['graphs']) == 3                                     # each size captured at its second sighting
    mat = np.zeros((nc, nc), np.int64)
    hist = np.zeros((nc, nc), np.int64)
    with torch.no_grad():
        for x, y in batches:
            o, _ = ON.model_forward(sd, x, backbone, head, training=False)
            m, h_ = OL.confusion_counts(o, y, nc, 255)
            mat += m
            hist += h_
    total = mat.sum()
    got = confmat.mat.cpu().numpy()
    assert got.sum() == total
    flips = np.abs(got - mat).sum() / 2
    print(f'[{family}] {int(total)} valid pixels over {len(batches)} images, {flips:.0f} arg-max decisions differ from the oracle')
    assert flips <= 2e-4 * total + 2
    o_iou, _, _ = OL.metrics_from_hist(torch.from_numpy(hist).float())
    assert abs(metric.compute_iou()[1] - o_iou[1]) <= 0.1 + 1e-9


EVAL_PRECISION = {'cfg2': ('MiT-B0', 'SegFormerHead', 150, 2, 512, 512, 60), 'cfg4': ('MiT-B2', 'SegFormerHead', 19, 1, 1024, 2048, 40)}


@pytest.mark.parametrize('cfg', sorted(EVAL_PRECISION))
def test_evaluate_in_fp32_like_the_reference_and_what_bf16_flips(cfg):
    """The reference evaluates in fp32 with autocast off (engine.py:86-88); so does engine.evaluate by default.  At BASELINE's cfg2 /
    cfg4 full size on TRAINED-LIKE logits with small top-2 margins (a model trained for a few dozen steps in bf16 on a learnable batch,
    evaluated on a held-out batch with extra input noise): (a) the default (fp32) confusion matrix against the CPU oracle's on the
    trained weights -- mIoU within the north star's +-0.1; (b) --eval-dtype bf16 against the same oracle: how many arg-max decisions
    the production storage type flips (reported), mIoU within +-0.1 as well on this fixture."""
    import types
    from segmentation_factory_amd import engine
    from segmentation_factory_amd.optim import FusedAGCAdamW, NativeScaler
    backbone, head, nc, B, H, W, steps = EVAL_PRECISION[cfg]
    seed = 97
    sd = OW.make_state_dict(backbone, head, nc, seed, lively=True)
    x, y = OW.learnable_batch(B, H, W, nc, seed, block=32)
    model = _build(backbone, head, nc, sd, torch.bfloat16, B)
    opt = FusedAGCAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    args = types.SimpleNamespace(nb_classes=nc, dice=True, ignore_index=255, ignore_label=255, local_rank=0, device='cuda')
    engine.train_one_epoch(model, opt, [(x, y)] * steps, 0, 'cuda', 1000, None, None, NativeScaler(), None, args)
    xv, yv = OW.learnable_batch(B, H, W, nc, seed + 1, block=32, noise=0.6)      # held out, 4x the training noise: small margins
    res = {}
    for dt in ('fp32', 'bf16'):
        a = types.SimpleNamespace(**vars(args), eval_dtype=dt)
        confmat, metric = engine.evaluate(a, model, [(xv, yv)], 'cuda', 10, None)
        res[dt] = (confmat.mat.cpu().numpy(), metric.compute_iou()[1])
    trained = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    trained = {k: (v.long() if k.endswith('num_batches_tracked') else v) for k, v in trained.items()}
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    with torch.no_grad():
        o, _ = ON.model_forward(trained, xv, backbone, head, training=False)
    mat, hist = OL.confusion_counts(o, yv, nc, 255)
    o_miou = OL.metrics_from_hist(torch.from_numpy(hist).float())[0][1]
    top2 = o.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).flatten()
    total = mat.sum()
    acc = np.trace(mat) / total
    line = [f'[{cfg}] oracle mIoU {o_miou}, pixel accuracy {acc:.3f}, median top-2 margin {margin.median().item():.3f} '
            f'(logit scale {o.abs().max().item():.1f}), {int(total)} valid pixels']
    for dt in ('fp32', 'bf16'):
        flips = np.abs(res[dt][0] - mat).sum() / 2
        line.append(f'{dt}: mIoU {res[dt][1]} ({res[dt][1] - o_miou:+.2f}), {flips:.0f} flipped decisions = {100 * flips / total:.3f} %')
        res[dt] += (flips,)
    print('; '.join(line))
    assert 0.05 < acc < 0.999                                             # trained-like, not decisive: the fixture measures something
    assert res['fp32'][0].sum() == total and res['bf16'][0].sum() == total
    assert abs(res['fp32'][1] - o_miou) <= 0.1 + 1e-9 and res['fp32'][2] <= 1e-4 * total + 2
    assert abs(res['bf16'][1] - o_miou) <= 0.1 + 1e-9, line


def test_dispatch_table_is_what_the_step_launches(golden_dir):
    """tests/golden/dispatch_table.json (replayed on the CPU by tests/test_host_cpu.py::test_dispatch_of_baseline_shapes) against a live
    recording on this GPU: the train step of cfg2 at the reference's default batch 4 (train_gpu.py:71) makes exactly the calls the
    table lists, and each launches exactly the kernels the table names."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import make_dispatch_table as mk
    with open(os.path.join(golden_dir, 'dispatch_table.json')) as fh:
        want = json.load(fh)['cfg2_b4']
    got = mk.record_case('cfg2', 4, False)
    key = lambda e: (e['fn'], json.dumps(e['args']))
    w, g = {key(e): e for e in want}, {key(e): e for e in got}
    assert set(w) == set(g), (sorted(set(w) - set(g))[:3], sorted(set(g) - set(w))[:3])
    for k in w:
        assert w[k]['kernels'] == g[k]['kernels'] and w[k]['count'] == g[k]['count'], (k, w[k], g[k])


def test_headline_batch_equals_its_halves():
    """bench.py's per-GPU batch of the headline configuration (256 x 512 x 512: the 768-wide stride-4 map has 3.2 x 10^9 elements, past
    2^31) against its 128-image halves in eval mode (tools/check_large_batch.py): bit-equal logits, equal mean loss, gradients within
    the split-K round-off -- an index that overflows 32 bits reads other memory and is off by the scale of the tensor."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d in (root, os.path.join(root, 'tools')):
        if d not in sys.path:
            sys.path.insert(0, d)
    import check_large_batch
    e_fwd, scale, l1, l2, e_g = check_large_batch.check(128, 'cfg2', verbose=False)
    assert e_fwd == 0.0, (e_fwd, scale)
    assert abs(l1 - l2) <= 1e-5 * abs(l1) and e_g < 2e-2
    torch.cuda.empty_cache()
