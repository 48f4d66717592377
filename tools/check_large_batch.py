"""Index-width check at large per-GPU batches: in eval mode (BatchNorm on running statistics, no dropout / drop-path) samples
are independent, so a batch of 2n images made of two copies of n images must give the same low-res logits per sample and the
same parameter gradients (mean-reduced loss) as the n-image batch -- any 32-bit offset overflow at > 2^31 elements / 2^32 bytes per
tensor would show up here.  Usage: python tools/check_large_batch.py [n=64] [cfg2|cfg3|cfg4|cfg5]   (checks the 2n-image batch)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_legs
from segmentation_factory_amd import functional as Fh


def check(n, cfg='cfg2', verbose=True):
    """-> (logit diff, scale, loss n, loss 2n, worst relative gradient diff); raises AssertionError on a mismatch."""
    m, _opt, NC, H, W = bench_legs.build(cfg)          # the bench's model and initialisation (bf16 compute)
    m.eval()
    x, y = bench_legs.synthetic_batch(n, NC, H, W, 0)
    x, y = x.cuda(), y.cuda()

    def run(xx, yy):
        for p in m.parameters():
            p.grad = None
        B = xx.shape[0]
        lo = m.forward_lowres(xx)
        loss, _, _ = Fh.upsample_ce_dice(lo.data, yy, (B, NC, lo.H, lo.W, H, W), 255, None, False)     # CE only: mean over pixels
        loss.backward()
        torch.cuda.synchronize()
        return lo.data.detach().float().clone(), loss.item(), {k: p.grad.detach().float().clone() for k, p in m.named_parameters() if p.grad is not None}

    lo1, l1, g1 = run(x, y)
    lo2, l2, g2 = run(torch.cat([x, x]), torch.cat([y, y]))
    rows = lo1.shape[0]
    e_fwd = max((lo2[:rows] - lo1).abs().max().item(), (lo2[rows:] - lo1).abs().max().item())
    e_g = max(((g2[k] - g1[k]).abs().max() / (g1[k].abs().max() + 1e-12)).item() for k in g1)
    scale = lo1.abs().max().item()
    if verbose:
        print(f'{cfg} n={n} (batch {2 * n} against {n}): logits max |diff| {e_fwd:.3e} (scale {scale:.2f}), loss {l1:.6f} vs {l2:.6f}, '
              f'worst relative grad diff {e_g:.3e}')
    # (a few-tile product may take its split-K form at one batch and not at the other: one bf16 unit of the logit scale is allowed;
    # an index overflow reads other memory and is off by the scale itself)
    assert e_fwd <= 2.0 ** -6 * max(1.0, scale) and abs(l1 - l2) <= 1e-5 * abs(l1) and e_g < 2e-2, 'MISMATCH'
    return e_fwd, scale, l1, l2, e_g


if __name__ == '__main__':
    check(int(sys.argv[1]) if len(sys.argv) > 1 else 64, sys.argv[2] if len(sys.argv) > 2 else 'cfg2')
    print('OK')
