"""A/B of the band-sweep loss backward against the tile kernel (same library, env switch), then timing."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_factory_amd import hip

dev = 'cuda'


def run(B, nc, h, w, ld, seed=0, ign_rows=2, weighted=False):
    g = torch.Generator(device=dev).manual_seed(seed)
    buf = torch.zeros(B * h * w, ld, dtype=torch.bfloat16, device=dev)
    buf[:, :nc] = (torch.randn(B * h * w, nc, generator=g, device=dev) * 2).to(torch.bfloat16)
    lo = buf[:, :nc]
    H, W = 4 * h, 4 * w
    tgt = torch.randint(0, nc, (B, H, W), device=dev, generator=g)
    tgt[:, :ign_rows] = 255
    cw = (torch.rand(nc, device=dev, generator=g) + 0.5) if weighted else None
    os.environ['SEGFAC_LOSS_NO_BAND_FWD'] = '1'
    loss0, stats0 = hip.ce_dice_fwd(lo, B, nc, h, w, H, W, tgt, 255, cw, True)
    os.environ.pop('SEGFAC_LOSS_NO_BAND_FWD')
    loss, stats = hip.ce_dice_fwd(lo, B, nc, h, w, H, W, tgt, 255, cw, True)
    loss_b, stats_b = hip.ce_dice_fwd(lo, B, nc, h, w, H, W, tgt, 255, cw, True)
    torch.cuda.synchronize()
    n = B * (3 * nc + 4) + 4
    assert torch.equal(stats[:n], stats_b[:n]) and torch.equal(loss, loss_b), 'band forward not reproducible'
    st0, st1 = stats0[:n - 4].view(B, 3 * nc + 4), stats[:n - 4].view(B, 3 * nc + 4)
    relP = ((st0[:, nc:2 * nc] - st1[:, nc:2 * nc]).abs().max() / st0[:, nc:2 * nc].abs().max()).item()
    relI = ((st0[:, :nc] - st1[:, :nc]).abs().max() / st0[:, :nc].abs().max().clamp_min(1e-30)).item()
    eqT = torch.equal(st0[:, 2 * nc:3 * nc], st1[:, 2 * nc:3 * nc])
    print(f'   fwd: loss tile {loss0[0].item():.7f} band {loss[0].item():.7f} relP {relP:.2e} relI {relI:.2e} T equal {eqT} tails {st0[:, 3 * nc:].sum(0).tolist()} {st1[:, 3 * nc:].sum(0).tolist()}')
    go = torch.full((1,), 1.7, device=dev)
    outs = {}
    for name, env in (('tile', '1'), ('band', None)):
        if env: os.environ['SEGFAC_LOSS_NO_BAND'] = env
        else: os.environ.pop('SEGFAC_LOSS_NO_BAND', None)
        d = hip.ce_dice_bwd(lo, B, nc, h, w, H, W, tgt, 255, cw, True, stats, go)
        torch.cuda.synchronize()
        outs[name] = d.float().clone()
        if name == 'band':
            d2 = hip.ce_dice_bwd(lo, B, nc, h, w, H, W, tgt, 255, cw, True, stats, go)
            torch.cuda.synchronize()
            assert torch.equal(d2.float(), outs['band']), 'band kernel not reproducible'
    a, b_ = outs['tile'], outs['band']
    scale = a.abs().max().item()
    err = (a - b_).abs().max().item()
    print(f'B={B} nc={nc} {h}x{w} ld={ld} w={weighted}: max|tile|={scale:.3e} max|diff|={err:.3e} rel={err / max(scale, 1e-30):.3e} '
          f'nan={bool(torch.isnan(b_).any())} shape={tuple(b_.shape)}', flush=True)
    return err / max(scale, 1e-30)


if __name__ == '__main__':
    worst = 0.0
    for cfg in [] if '--time-only' in sys.argv else [(2, 19, 8, 8, 24), (2, 150, 6, 10, 152), (1, 2, 5, 5, 8), (2, 171, 9, 7, 176), (2, 150, 16, 23, 160),
                (2, 19, 32, 64, 32), (3, 150, 128, 128, 160), (2, 40, 9, 30, 40), (1, 64, 7, 15, 64), (1, 130, 8, 8, 136)]:
        worst = max(worst, run(*cfg))
        worst = max(worst, run(*cfg, weighted=True, seed=3))
    print('worst rel', worst)
    B, nc, h, ld = 128, 150, 128, 160
    buf = (torch.randn(B * h * h, ld, device=dev) * 2).to(torch.bfloat16)
    buf[:, nc:] = 0
    lo = buf[:, :nc]
    tgt = torch.randint(0, nc, (B, 512, 512), device=dev)
    tgt[:, :8] = 255
    loss, stats = hip.ce_dice_fwd(lo, B, nc, h, h, 512, 512, tgt, 255, None, True)
    go = torch.ones(1, device=dev)
    for name, envs in (('fwd tile', {'SEGFAC_LOSS_NO_BAND_FWD': '1'}), ('fwd band', {})):
        os.environ.pop('SEGFAC_LOSS_NO_BAND_FWD', None)
        os.environ.update(envs)
        for _ in range(3): hip.ce_dice_fwd(lo, B, nc, h, h, 512, 512, tgt, 255, None, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): hip.ce_dice_fwd(lo, B, nc, h, h, 512, 512, tgt, 255, None, True)
        torch.cuda.synchronize()
        print(f'{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms (incl. finalize kernels)', flush=True)
    os.environ.pop('SEGFAC_LOSS_NO_BAND_FWD', None)
    for name, envs in (('tile', {'SEGFAC_LOSS_NO_BAND': '1'}), ('band', {}), ('band rows16', {'SEGFAC_LOSS_BAND_ROWS': '16'}),
                       ('band rows32', {'SEGFAC_LOSS_BAND_ROWS': '32'}), ('band rows64', {'SEGFAC_LOSS_BAND_ROWS': '64'}),
                       ('band rows128', {'SEGFAC_LOSS_BAND_ROWS': '128'})):
        for k in ('SEGFAC_LOSS_NO_BAND', 'SEGFAC_LOSS_BAND_ROWS'): os.environ.pop(k, None)
        os.environ.update(envs)
        for _ in range(3): hip.ce_dice_bwd(lo, B, nc, h, h, 512, 512, tgt, 255, None, True, stats, go)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): hip.ce_dice_bwd(lo, B, nc, h, h, 512, 512, tgt, 255, None, True, stats, go)
        torch.cuda.synchronize()
        print(f'{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms', flush=True)
