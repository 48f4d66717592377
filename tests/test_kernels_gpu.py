"""GPU parity tests of every HIP kernel (through the C ABI via segmentation_factory_amd.hip / functional)
against a plain PyTorch fp32 CPU statement of the same op.  fp32 storage must match to ~1e-5; bf16 storage to
bf16 rounding of inputs/outputs (fp32 accumulation inside)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _tol(dtype):
    return (2e-5, 2e-5) if dtype == torch.float32 else (3e-2, 3e-2)


def _close(got, ref, dtype, scale=None, fac=1.0):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    rt, at = _tol(dtype)
    s = ref.abs().max().item() if scale is None else scale
    err = (got - ref).abs().max().item()
    assert err <= fac * (rt * s + at * 1e-2), f'max err {err:.3e} vs scale {s:.3e} ({dtype})'


def _dev(t, dtype=None):
    t = t.detach().cuda()
    return t.to(dtype) if dtype is not None else t


def _q(t, dtype):
    """quantise a CPU fp32 tensor to the storage dtype and back (what the kernel actually sees)."""
    return t.detach().to(dtype).float().clone()


@pytest.fixture(scope='module')
def hipmod():
    from segmentation_factory_amd import hip
    hip.lib()
    return hip


def test_plumbing(hipmod):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(5, 37, 29, generator=g)
    for dt in DTYPES:
        out = hipmod.permute021(_dev(x), 5, 37, 29, dt, ld_out=40)
        ref = torch.zeros(5, 29, 40)
        ref[:, :, :37] = x.permute(0, 2, 1)
        _close(out, ref, dt)
    a = torch.randn(70, 150, generator=g)
    b = torch.randn(70, 150, generator=g)
    s = torch.rand(7, generator=g)
    for dt in DTYPES:
        _close(hipmod.cast(_dev(a), dt), a, dt)
        _close(hipmod.scale_rows(_dev(a, dt), _dev(s), 10), _q(a, dt) * s.repeat_interleave(10)[:, None], dt)
        _close(hipmod.add(_dev(a, dt), _dev(b, dt)), _q(a, dt) + _q(b, dt), dt)
        _close(hipmod.colsum(_dev(a, dt)), _q(a, dt).sum(0), torch.float32, fac=4)
    big = torch.randn(5000, 768, generator=g)
    _close(hipmod.colsum(_dev(big)), big.sum(0), torch.float32, fac=20)
    wide = torch.randn(300, 3072, generator=g)
    _close(hipmod.colsum(_dev(wide, torch.bfloat16)), _q(wide, torch.bfloat16).sum(0), torch.float32, fac=20)


GEMM_SHAPES = [(300, 150, 147), (257, 32, 32), (128, 128, 64), (1000, 768, 3072 // 4), (64, 768, 32), (513, 259, 1031)]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('no_tr', ['0', '1'])
def test_gemm_layouts(hipmod, dtype, no_tr):
    os.environ['SEGFAC_GEMM_NO_TR'] = no_tr
    try:
        g = torch.Generator().manual_seed(1)
        for (M, N, K) in GEMM_SHAPES:
            x = torch.randn(M, K, generator=g)
            w = torch.randn(N, K, generator=g) / K ** 0.5
            dy = torch.randn(M, N, generator=g)
            xq, wq, dyq = _q(x, dtype), _q(w, dtype), _q(dy, dtype)
            y = hipmod.gemm(0, _dev(x, dtype), _dev(w, dtype), M, N, K)
            _close(y, xq @ wq.t(), dtype)
            dx = hipmod.gemm(1, _dev(dy, dtype), _dev(w, dtype), M, K, N)
            _close(dx, dyq @ wq, dtype)
            dw = hipmod.gemm(2, _dev(dy, dtype), _dev(x, dtype), N, K, M, out_dtype=torch.float32)
            _close(dw, dyq.t() @ xq, torch.float32 if dtype == torch.float32 else dtype, fac=4)
            sk = 3
            dw2 = hipmod.gemm(2, _dev(dy, dtype), _dev(x, dtype), N, K, M, out_dtype=torch.float32, split_k=sk)
            _close(dw2, dyq.t() @ xq, torch.float32 if dtype == torch.float32 else dtype, fac=4)
    finally:
        os.environ['SEGFAC_GEMM_NO_TR'] = '0'


@pytest.mark.parametrize('dtype', DTYPES)
def test_gemm_epilogue_and_strides(hipmod, dtype):
    g = torch.Generator().manual_seed(2)
    M, N, K, rpg = 96, 150, 64, 32
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / 8
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    s = torch.rand(M // rpg, generator=g) * 2
    xq, wq, rq = _q(x, dtype), _q(w, dtype), _q(r, dtype)
    ref = rq + s.repeat_interleave(rpg)[:, None] * (xq @ wq.t() + b)
    y = hipmod.gemm(0, _dev(x, dtype), _dev(w, dtype), M, N, K, bias=_dev(b), residual=_dev(r, dtype), rscale=_dev(s),
                    rows_per_group=rpg)
    _close(y, ref, dtype)
    # strided A (column slice of a wider buffer) and strided output slice
    wide = torch.randn(M, 3 * K, generator=g)
    out = torch.zeros(M, 2 * N + 10, dtype=dtype, device='cuda')
    hipmod.gemm(0, _dev(wide, dtype)[:, K:2 * K], _dev(w, dtype), M, N, K, out=out[:, N:2 * N], bias=_dev(b))
    _close(out[:, N:2 * N], _q(wide, dtype)[:, K:2 * K] @ wq.t() + b, dtype)
    assert out[:, :N].abs().max().item() == 0 and out[:, 2 * N:].abs().max().item() == 0


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('C', [32, 64, 160, 256, 768, 1536, 2816])     # 2816: convnextv2_huge's last stage (six chunks per lane)
def test_layernorm(dtype, C):
    from segmentation_factory_amd import functional as Fh
    g = torch.Generator().manual_seed(3)
    rows = 333
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5)
    gam = 1 + 0.1 * torch.randn(C, generator=g)
    bet = 0.1 * torch.randn(C, generator=g)
    dy = torch.randn(rows, C, generator=g)
    xr = _q(x, dtype).requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), gr, br, 1e-5)
    ref.backward(_q(dy, dtype))
    xd = _dev(x, dtype).requires_grad_(True)
    gd, bd = _dev(gam).requires_grad_(True), _dev(bet).requires_grad_(True)
    y = Fh.layer_norm(xd, gd, bd, 1e-5)
    y.backward(_dev(dy, dtype))
    _close(y, ref, dtype)
    _close(xd.grad, xr.grad, dtype)
    _close(gd.grad, gr.grad, dtype, fac=8)
    _close(bd.grad, br.grad, dtype, fac=8)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('act', [0, 1, 2])
def test_batchnorm_act(dtype, act):
    from segmentation_factory_amd import functional as Fh
    g = torch.Generator().manual_seed(4)
    B, HW, C = 3, 50, 72
    x = torch.randn(B * HW, C, generator=g) * 3 + 1
    gam = 1 + 0.2 * torch.randn(C, generator=g)
    bet = 0.5 * torch.randn(C, generator=g)
    dy = torch.randn(B * HW, C, generator=g)
    keep = (torch.rand(B, C, generator=g) > 0.2).float() / 0.9
    rm0, rv0 = 0.1 * torch.randn(C, generator=g), 1 + 0.1 * torch.rand(C, generator=g)
    for training in (True, False):
        xr = _q(x, dtype).requires_grad_(True)
        gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
        rm, rv = rm0.clone(), rv0.clone()
        t = F.batch_norm(xr.t().reshape(1, C, -1), rm, rv, gr, br, training, 0.1, 1e-5)[0].t()
        t = F.relu(t) if act == 1 else (F.relu6(t) if act == 2 else t)
        ref = t * keep.repeat_interleave(HW, 0)
        ref.backward(_q(dy, dtype))
        xd = _dev(x, dtype).requires_grad_(True)
        gd, bd = _dev(gam).requires_grad_(True), _dev(bet).requires_grad_(True)
        rmd, rvd = _dev(rm0.clone()), _dev(rv0.clone())
        y = Fh.batch_norm_act(xd, gd, bd, rmd, rvd, training, 0.1, 1e-5, act, _dev(keep), HW)
        y.backward(_dev(dy, dtype))
        _close(y, ref, dtype)
        _close(xd.grad, xr.grad, dtype, fac=2)
        _close(gd.grad, gr.grad, dtype, fac=8)
        _close(bd.grad, br.grad, dtype, fac=8)
        _close(rmd, rm, torch.float32, fac=4)
        _close(rvd, rv, torch.float32, fac=4)


def _attn_ref(q, kv, B, N, Nkv, heads):
    C = q.shape[1]
    hd = C // heads
    qh = q.reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    k, v = kv.reshape(B, Nkv, 2, heads, hd).permute(2, 0, 3, 1, 4)
    a = ((qh @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1)
    return (a @ v).transpose(1, 2).reshape(B * N, C)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 2, 100, 37, 32), (1, 1, 300, 256, 64), (2, 5, 64, 4, 32), (1, 8, 16, 16, 32), (1, 2, 700, 300, 64),
                                   (1, 2, 8192, 2048, 64), (1, 1, 8200, 2048, 64), (2, 1, 4096, 256, 32)])   # cfg4: 2048 keys x head_dim 64; cfg2 stage 2
def test_attention(dtype, shape):
    from segmentation_factory_amd import functional as Fh
    B, heads, N, Nkv, hd = shape
    C = heads * hd
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B * N, C, generator=g)
    kv = torch.randn(B * Nkv, 2 * C, generator=g)
    do = torch.randn(B * N, C, generator=g)
    qr, kvr = _q(q, dtype).requires_grad_(True), _q(kv, dtype).requires_grad_(True)
    ref = _attn_ref(qr, kvr, B, N, Nkv, heads)
    ref.backward(_q(do, dtype))
    qd, kvd = _dev(q, dtype).requires_grad_(True), _dev(kv, dtype).requires_grad_(True)
    o = Fh.attention(qd, kvd, B, N, Nkv, heads)
    o.backward(_dev(do, dtype))
    _close(o, ref, dtype)
    _close(qd.grad, qr.grad, dtype, fac=2)
    _close(kvd.grad, kvr.grad, dtype, fac=4)


def test_attention_head_dim_64_long_keys_is_bit_reproducible():
    """Head dim 64 with >= 128 keys (MiT-B2 and up, csrc/attention_mfma.hip: attn_fwd64p / attn_bwd_dq64p / attn_mfma_bwd_dkv): repeated
    runs on the same inputs agree bit for bit.  r05 found r04's forward NOT reproducible on a full chip (a few of 16384 log-sum-exp rows
    per dozen runs): its inline-asm v_max3 read score accumulators behind the matrix instructions without the wait states the compiler
    only pads in front of instructions it knows to be vector ones -- a stale (similar-sized) score moved the running maximum.  The
    shape is the one that showed it: 2 heads x 128 query blocks = one workgroup per CU."""
    from segmentation_factory_amd import hip
    B, heads, N, Nkv, hd = 1, 2, 8192, 2048, 64
    g = torch.Generator().manual_seed(5)
    C = heads * hd
    q, k, v, do = (torch.randn(n, C, generator=g).bfloat16().cuda() for n in (B * N, B * Nkv, B * Nkv, B * N))
    first = None
    for rep in range(12):
        o, lse = hip.attention_fwd(q, k, v, B, heads, N, Nkv, hd, hd ** -0.5)
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        dq = hip.attention_bwd(q, k, v, o, do, lse, B, heads, N, Nkv, hd, hd ** -0.5, dk, dv)
        torch.cuda.synchronize()
        cur = [t.clone() for t in (o, lse, dq, dk, dv)]
        if first is None:
            first = cur
        else:
            for name, a, b in zip(('o', 'lse', 'dq', 'dk', 'dv'), first, cur):
                assert torch.equal(a, b), (rep, name, (a != b).sum().item())


@pytest.mark.parametrize('shape', [(1, 2, 8192, 2048, 64), (1, 1, 8200, 2048, 64), (1, 2, 700, 300, 64), (2, 1, 1000, 130, 64), (1, 1, 40, 128, 64), (1, 1, 100, 128, 64)])
def test_attention_head_dim_64_key_side_backward_tile_rows(shape, monkeypatch):
    """SEGFAC_ATTN64_DKV_ROWS (csrc/policy.h; default 128): the key-side backward stages Q / dO tiles of 128 or 64 queries per barrier
    instead of 32 and walks them as 32-query parts -- the same products in the same order: dK / dV (and dQ, untouched) bit for bit, for
    query counts that end inside a tile's first part, inside a later one and on a tile boundary."""
    from segmentation_factory_amd import hip
    B, heads, N, Nkv, hd = shape
    C = heads * hd
    g = torch.Generator().manual_seed(9)
    q, k, v, do = (torch.randn(n, C, generator=g).bfloat16().cuda() for n in (B * N, B * Nkv, B * Nkv, B * N))
    o, lse = hip.attention_fwd(q, k, v, B, heads, N, Nkv, hd, hd ** -0.5)
    outs = []
    for rows in ('32', '64', '128'):
        monkeypatch.setenv('SEGFAC_ATTN64_DKV_ROWS', rows)
        dk, dv = torch.empty_like(k), torch.empty_like(v)
        dq = hip.attention_bwd(q, k, v, o, do, lse, B, heads, N, Nkv, hd, hd ** -0.5, dk, dv)
        torch.cuda.synchronize()
        outs.append((dq.clone(), dk.clone(), dv.clone()))
    for other in outs[1:]:
        for name, a, b in zip(('dq', 'dk', 'dv'), outs[0], other):
            assert torch.equal(a, b), (name, (a != b).sum().item())


@pytest.mark.parametrize('shape', [(1, 2, 8192, 2048, 64), (1, 1, 8200, 2048, 64), (1, 2, 700, 300, 64), (2, 1, 1000, 130, 64)])
def test_attention_head_dim_64_prescale_option(shape, monkeypatch):
    """SEGFAC_ATTN64_PRESCALE=1 (csrc/policy.h: scale log2 e on the Q fragments, -max / -lse as the score accumulators' initial values;
    off by default because it rounds q once more): same tolerance against the fp32 reference as the default path, and two runs agree
    bitwise.  Key counts that are not multiples of 64 / 32 take the clamped last stage and the masked last step."""
    from segmentation_factory_amd import functional as Fh
    B, heads, N, Nkv, hd = shape
    C = heads * hd
    g = torch.Generator().manual_seed(5)
    q = torch.randn(B * N, C, generator=g)
    kv = torch.randn(B * Nkv, 2 * C, generator=g)
    do = torch.randn(B * N, C, generator=g)
    dtype = torch.bfloat16
    qr, kvr = _q(q, dtype).requires_grad_(True), _q(kv, dtype).requires_grad_(True)
    ref = _attn_ref(qr, kvr, B, N, Nkv, heads)
    ref.backward(_q(do, dtype))
    monkeypatch.setenv('SEGFAC_ATTN64_PRESCALE', '1')
    outs = []
    for rep in range(2):
        qd, kvd = _dev(q, dtype).requires_grad_(True), _dev(kv, dtype).requires_grad_(True)
        o = Fh.attention(qd, kvd, B, N, Nkv, heads)
        o.backward(_dev(do, dtype))
        torch.cuda.synchronize()
        outs.append((o.detach().clone(), qd.grad.clone(), kvd.grad.clone()))
    _close(outs[0][0], ref, dtype)
    _close(outs[0][1], qr.grad, dtype, fac=2)
    _close(outs[0][2], kvr.grad, dtype, fac=4)
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('shape', [(2, 1, 4096, 256, 32), (3, 2, 1000, 200, 32), (2, 5, 70, 64, 32), (1, 8, 256, 256, 32)])
def test_attention_backward_one_kernel_form_equals_two_kernel_form(shape, monkeypatch):
    """Head dim 32 with at most 256 keys (every stage of the 512^2 configs): the backward runs as ONE kernel (dQ through a transposed
    pass of dS over the wave's LDS slab, the waves' partial dQ tiles summed in fixed order) -- against the two-kernel form of the same
    library (env switch) and, through test_attention, the oracle.  dK / dV take the same arithmetic in both: bitwise equal; dQ differs
    by the bf16 rounding of dS (it is the B operand of an MFMA in both forms, summed in another order): within bf16 of the gradient scale.
    Two runs agree bitwise."""
    from segmentation_factory_amd import functional as Fh
    B, heads, N, Nkv, hd = shape
    C = heads * hd
    g = torch.Generator().manual_seed(7)
    q = torch.randn(B * N, C, generator=g)
    kv = torch.randn(B * Nkv, 2 * C, generator=g)
    do = torch.randn(B * N, C, generator=g)
    outs = []
    for mode in ('fused', 'fused', 'split'):
        if mode == 'split':
            monkeypatch.setenv('SEGFAC_ATTN_NO_FUSED_BWD', '1')
        qd, kvd = _dev(q, torch.bfloat16).requires_grad_(True), _dev(kv, torch.bfloat16).requires_grad_(True)
        Fh.attention(qd, kvd, B, N, Nkv, heads).backward(_dev(do, torch.bfloat16))
        torch.cuda.synchronize()
        outs.append((qd.grad.clone(), kvd.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][1], outs[2][1])
    sc = outs[2][0].float().abs().max().item()
    assert (outs[0][0].float() - outs[2][0].float()).abs().max().item() <= 2.0 ** -7 * sc


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('geom', [(2, 9, 13, 32), (1, 16, 16, 128), (2, 5, 3, 8), (1, 1, 1, 64)])
def test_dwconv3x3_gelu(dtype, geom):
    from segmentation_factory_amd import functional as Fh
    B, H, W, C = geom
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B * H * W, C, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) / 3
    b = torch.randn(C, generator=g) * 0.2
    dy = torch.randn(B * H * W, C, generator=g)
    xr = _q(x, dtype).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    t = F.conv2d(xr.reshape(B, H, W, C).permute(0, 3, 1, 2), wr, br, padding=1, groups=C)
    ref = F.gelu(t).permute(0, 2, 3, 1).reshape(B * H * W, C)
    ref.backward(_q(dy, dtype))
    xd = _dev(x, dtype).requires_grad_(True)
    wd, bd = _dev(w).requires_grad_(True), _dev(b).requires_grad_(True)
    y = Fh.dwconv3x3_gelu(xd, wd, bd, B, H, W, True)
    y.backward(_dev(dy, dtype))
    _close(y, ref, dtype)
    _close(xd.grad, xr.grad, dtype, fac=2)
    _close(wd.grad, wr.grad, dtype, fac=8)
    _close(bd.grad, br.grad, dtype, fac=8)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', [(2, 3, 32, 40, 32, 7, 4, 3, True), (2, 32, 16, 12, 64, 3, 2, 1, False),
                                 (1, 64, 16, 16, 64, 4, 4, 0, False), (2, 160, 7, 9, 256, 3, 2, 1, False),
                                 (1, 32, 16, 24, 32, 8, 8, 0, False), (1, 16, 9, 11, 24, 5, 2, 2, False)])
def test_conv_patch(dtype, cfg):
    from segmentation_factory_amd import functional as Fh
    B, Cin, H, W, O, k, s, p, image = cfg
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(O, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(O, generator=g) * 0.1
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(B * Ho * Wo, O, generator=g)
    xq = x if image else _q(x, dtype)
    xr = xq.clone().requires_grad_(True)
    wr, br = _q(w, dtype).requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr if not image else _q(xr, dtype), wr, br, stride=s, padding=p).permute(0, 2, 3, 1).reshape(B * Ho * Wo, O)
    ref.backward(_q(dy, dtype))
    if image:
        xd = _dev(x)
    else:
        xd = _dev(x.permute(0, 2, 3, 1).reshape(B * H * W, Cin).contiguous(), dtype).requires_grad_(True)
    wd, bd = _dev(w).requires_grad_(True), _dev(b).requires_grad_(True)
    y = Fh.conv_patch(xd, wd, bd, (B, H, W, Cin, k, s, p), image=image, dtype=dtype)
    y.backward(_dev(dy, dtype))
    _close(y, ref, dtype)
    _close(wd.grad, wr.grad, dtype, fac=8)
    _close(bd.grad, br.grad, dtype, fac=8)
    if not image:
        _close(xd.grad, xr.grad.permute(0, 2, 3, 1).reshape(B * H * W, Cin), dtype, fac=2)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', [(2, 4, 5, 16, 20, 24, False), (1, 6, 6, 16, 16, 8, True), (2, 1, 1, 7, 9, 16, True),
                                 (1, 16, 16, 16, 16, 8, False), (2, 3, 7, 12, 9, 150, False), (1, 8, 8, 4, 4, 16, False)])
def test_bilinear(hipmod, dtype, cfg):
    B, h, w, H, W, C, ac = cfg
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, C, h, w, generator=g)
    dy = torch.randn(B, C, H, W, generator=g)
    xr = _q(x, dtype).requires_grad_(True)
    ref = F.interpolate(xr, size=(H, W), mode='bilinear', align_corners=ac)
    ref.backward(_q(dy, dtype))
    xt = _dev(x.permute(0, 2, 3, 1).reshape(B * h * w, C).contiguous(), dtype)
    buf = torch.zeros(B * H * W, C + 16, dtype=dtype, device='cuda')
    hipmod.bilinear_fwd(xt, B, h, w, C, H, W, buf[:, 8:8 + C], align_corners=ac)
    _close(buf[:, 8:8 + C], ref.permute(0, 2, 3, 1).reshape(B * H * W, C), dtype)
    assert buf[:, :8].abs().max().item() == 0
    dyt = _dev(dy.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous(), dtype)
    din = hipmod.bilinear_bwd(dyt, B, h, w, C, H, W, align_corners=ac)
    _close(din, xr.grad.permute(0, 2, 3, 1).reshape(B * h * w, C), dtype, fac=4)
    full = hipmod.bilinear_to_nchw_f32(xt, B, h, w, C, H, W)
    if not ac:
        _close(full, ref, dtype)


def test_loss_golden_cases(golden_dir):
    """engine.criterion captured from the reference (tests/golden/loss_cases.npz): fp32 kernels."""
    from segmentation_factory_amd import engine
    g = np.load(os.path.join(golden_dir, 'loss_cases.npz'))
    for i in range(int(g['n'])):
        C, dice, weighted, ign = [int(v) for v in g[f'meta_{i}']]
        logits = torch.from_numpy(g[f'logits_{i}']).cuda().requires_grad_(True)
        t = torch.from_numpy(g[f'target_{i}']).cuda()
        w = torch.tensor([1.0, 2.0]).cuda() if weighted else None
        loss = engine.criterion(logits, t, w, num_classes=C, dice=bool(dice), ignore_index=ign)
        loss.backward()
        name = str(g[f'name_{i}'])
        assert abs(loss.item() - float(g[f'loss_{i}'])) < 5e-6 * max(1, abs(float(g[f'loss_{i}']))), name
        err = (logits.grad.cpu().numpy() - g[f'grad_{i}'])
        assert np.abs(err).max() < 2e-6 * max(1e-3, np.abs(g[f'grad_{i}']).max()) + 1e-9, name


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', [(2, 19, 8, 8, 32, 32), (2, 150, 6, 10, 24, 40), (1, 2, 5, 5, 20, 20), (3, 65, 4, 4, 4, 4),
                                 (2, 171, 9, 7, 36, 28), (2, 21, 8, 8, 16, 16), (1, 40, 4, 6, 32, 48)])
def test_upsample_ce_dice_vs_oracle(dtype, cfg):
    from oracle import loss as OL
    from segmentation_factory_amd import functional as Fh
    B, C, h, w, H, W = cfg
    g = torch.Generator().manual_seed(9)
    lo = torch.randn(B, C, h, w, generator=g) * 2
    t = torch.randint(0, C, (B, H, W), generator=g)
    t[:, :2] = 255
    if B > 1:
        t[1] = 255 if cfg[1] == 65 else t[1]
    lr = _q(lo, dtype).requires_grad_(True)
    up = F.interpolate(lr, size=(H, W), mode='bilinear', align_corners=False)
    ref = OL.criterion_closed_form(up, t, None, num_classes=C, dice=True, ignore_index=255)
    ref.backward()
    ld = (C + 7) // 8 * 8
    buf = torch.zeros(B * h * w, ld, dtype=dtype, device='cuda')
    buf[:, :C] = lo.permute(0, 2, 3, 1).reshape(B * h * w, C).to(dtype)
    tok = buf[:, :C].requires_grad_(True)
    loss, parts, _ = Fh.upsample_ce_dice(tok, t.cuda(), (B, C, h, w, H, W), 255, None, True)
    loss.backward()
    assert abs(loss.item() - ref.item()) < (1e-5 if dtype == torch.float32 else 2e-3) * max(1, abs(ref.item()))
    _close(tok.grad, lr.grad.permute(0, 2, 3, 1).reshape(B * h * w, C), dtype, fac=2 if dtype == torch.float32 else 6)


def test_upsample_ce_dice_retry_and_reproducible():
    """Logit spreads far beyond exp's range make the cell bound underflow: the batched / MFMA kernels must hand over to
    the exact-maximum pass (retry flag) and still match the oracle; two runs must agree bitwise."""
    from oracle import loss as OL
    from segmentation_factory_amd import functional as Fh
    B, C, h, w, H, W = 2, 150, 8, 8, 32, 32
    g = torch.Generator().manual_seed(12)
    t = torch.randint(0, C, (B, H, W), generator=g)
    t[:, :1] = 255
    for scale, expect_retry in ((2.0, 0), (150.0, 1)):
        lo = torch.randn(B, C, h, w, generator=g) * scale
        lr = lo.clone().requires_grad_(True)
        up = F.interpolate(lr, size=(H, W), mode='bilinear', align_corners=False)
        ref = OL.criterion_closed_form(up, t, None, num_classes=C, dice=True, ignore_index=255)
        ref.backward()
        outs = []
        for _ in range(2):
            tok = lo.permute(0, 2, 3, 1).reshape(B * h * w, C).contiguous().cuda().requires_grad_(True)
            loss, parts, stats = Fh.upsample_ce_dice(tok, t.cuda(), (B, C, h, w, H, W), 255, None, True)
            loss.backward()
            torch.cuda.synchronize()
            outs.append((loss.detach().clone(), tok.grad.clone()))
        assert int(stats[-4:].view(torch.int32)[0].item() != 0) == expect_retry
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        assert abs(outs[0][0].item() - ref.item()) < 1e-5 * max(1, abs(ref.item()))
        _close(outs[0][1], lr.grad.permute(0, 2, 3, 1).reshape(B * h * w, C), torch.float32, fac=2)


@pytest.mark.parametrize('cfg', [(2, 150, 16, 23, 160, False), (2, 19, 32, 64, 32, True), (1, 171, 9, 7, 176, False), (2, 40, 9, 30, 40, True),
                                 (1, 2, 5, 5, 8, False), (2, 150, 40, 40, 152, True), (1, 64, 8, 8, 64, False), (1, 32, 8, 13, 32, True),
                                 (1, 192, 8, 8, 192, False), (2, 128, 10, 9, 128, True), (1, 16, 3, 50, 16, False)])
def test_loss_backward_band_kernel_vs_tile_kernel_and_oracle(cfg, monkeypatch):
    """bf16, ratio 4: the band-sweep backward (loss_band.hip) against the tile kernel it replaces (same library, env switch) and
    against the fp32 oracle on the same bf16-rounded logits; two runs bitwise equal; pad columns exactly zero.  Shapes cover
    bands that end inside the image, one-band images, several segments, class counts in every tile bucket and padded rows."""
    from oracle import loss as OL
    from segmentation_factory_amd import hip
    B, C, h, w, ld, weighted = cfg
    H, W = 4 * h, 4 * w
    g = torch.Generator().manual_seed(21)
    lo = (torch.randn(B, C, h, w, generator=g) * 2).to(torch.bfloat16)
    t = torch.randint(0, C, (B, H, W), generator=g)
    t[:, :3] = 255
    t[:, :, -5:] = 255
    cw = (torch.rand(C, generator=g) + 0.5) if weighted else None
    lr = lo.float().requires_grad_(True)
    up = F.interpolate(lr, size=(H, W), mode='bilinear', align_corners=False)
    ref = OL.criterion_closed_form(up, t, cw, num_classes=C, dice=True, ignore_index=255)
    (1.7 * ref).backward()
    buf = torch.zeros(B * h * w, ld, dtype=torch.bfloat16, device='cuda')
    buf[:, :C] = lo.permute(0, 2, 3, 1).reshape(B * h * w, C).cuda()
    tok, tg = buf[:, :C], t.cuda()
    cwd = cw.cuda() if weighted else None
    loss, stats = hip.ce_dice_fwd(tok, B, C, h, w, H, W, tg, 255, cwd, True)
    # forward: the strided-cell kernel of loss_band.hip (per-class P sums on the matrix pipe from bf16 exp tiles) against the
    # oracle and against the tile kernel; reproducible; label counts exact
    loss_again, stats_again = hip.ce_dice_fwd(tok, B, C, h, w, H, W, tg, 255, cwd, True)
    monkeypatch.setenv('SEGFAC_LOSS_NO_BAND_FWD', '1')
    loss_tile, stats_tile = hip.ce_dice_fwd(tok, B, C, h, w, H, W, tg, 255, cwd, True)
    monkeypatch.delenv('SEGFAC_LOSS_NO_BAND_FWD')
    n = B * (3 * C + 4)
    assert torch.equal(loss, loss_again) and torch.equal(stats[:n], stats_again[:n])
    assert abs(loss[0].item() - ref.item()) < 1e-4 * max(1, abs(ref.item())), (loss[0].item(), ref.item())
    assert abs(loss_tile[0].item() - ref.item()) < 2e-5 * max(1, abs(ref.item()))
    sb, st = stats[:n].view(B, 3 * C + 4), stats_tile[:n].view(B, 3 * C + 4)
    assert torch.equal(sb[:, 2 * C:3 * C], st[:, 2 * C:3 * C]) and torch.equal(sb[:, 3 * C + 1:], st[:, 3 * C + 1:])    # T, weights, counts
    assert (sb[:, :C] - st[:, :C]).abs().max().item() <= 1e-5 * st[:, :C].abs().max().item() + 1e-6                    # I: fp32 both
    assert (sb[:, C:2 * C] - st[:, C:2 * C]).abs().max().item() <= 2e-3 * st[:, C:2 * C].abs().max().item()          # P: bf16 tiles
    go = torch.full((1,), 1.7, device='cuda')
    monkeypatch.setenv('SEGFAC_LOSS_BAND_ROWS', '8')           # several segments also on the small maps
    band = hip.ce_dice_bwd(tok, B, C, h, w, H, W, tg, 255, cwd, True, stats, go)
    band2 = hip.ce_dice_bwd(tok, B, C, h, w, H, W, tg, 255, cwd, True, stats, go)
    monkeypatch.delenv('SEGFAC_LOSS_BAND_ROWS')
    band3 = hip.ce_dice_bwd(tok, B, C, h, w, H, W, tg, 255, cwd, True, stats, go)
    monkeypatch.setenv('SEGFAC_LOSS_NO_BAND', '1')
    tile = hip.ce_dice_bwd(tok, B, C, h, w, H, W, tg, 255, cwd, True, stats, go)
    torch.cuda.synchronize()
    assert torch.equal(band, band2)
    assert band.shape == (B * h * w, ld) and not band[:, C:].any() and not band3[:, C:].any()
    want = lr.grad.permute(0, 2, 3, 1).reshape(B * h * w, C)
    scale = want.abs().max().item()
    for got in (band, band3, tile):
        err = (got[:, :C].float().cpu() - want).abs().max().item()
        assert err < 1e-2 * scale, (err, scale)                  # bf16 storage of the gradient: 2^-8 relative
    # the segment length only changes which wave computes a tap, never the arithmetic of a tap
    assert torch.equal(band, band3)
    # the variant the training step runs: the forward leaves -log2(sum exp) per pixel, the backward's exp2 yields probabilities
    # (no class maximum, no sum, no reciprocal).  Same forward statistics bit for bit, gradient within the same bar, reproducible,
    # independent of the segment length, pad columns zero
    monkeypatch.delenv('SEGFAC_LOSS_NO_BAND')
    loss_l, stats_l, lse = hip.ce_dice_fwd(tok, B, C, h, w, H, W, tg, 255, cwd, True, want_lse=True)
    assert lse is not None and lse.numel() == B * (h + 1) * (w + 1) * 16
    assert torch.equal(loss_l, loss) and torch.equal(stats_l[:n], stats[:n]) and stats_l[-4:].view(torch.int32)[1].item() == 0
    # the buffer against the oracle's full-resolution logits: pixel (y, x) lives in cell ((y + 2) // 4, (x + 2) // 4)
    ys, xs = torch.arange(H), torch.arange(W)
    cell = ((ys + 2) // 4)[:, None] * (w + 1) + ((xs + 2) // 4)[None, :]
    slot = (((ys + 2) % 4) * 4)[:, None] + ((xs + 2) % 4)[None, :]
    got_lse = lse.view(B, (h + 1) * (w + 1), 16).cpu()[:, cell, slot]
    want_lse = -torch.logsumexp(up.detach(), dim=1) / math.log(2.0)
    assert (got_lse - want_lse).abs().max().item() < 1e-4 * max(1.0, want_lse.abs().max().item())
    l1 = hip.ce_dice_bwd(tok, B, C, h, w, H, W, tg, 255, cwd, True, stats_l, go, lse=lse)
    monkeypatch.setenv('SEGFAC_LOSS_BAND_ROWS', '8')
    l2 = hip.ce_dice_bwd(tok, B, C, h, w, H, W, tg, 255, cwd, True, stats_l, go, lse=lse)
    l3 = hip.ce_dice_bwd(tok, B, C, h, w, H, W, tg, 255, cwd, True, stats_l, go, lse=lse)
    torch.cuda.synchronize()
    assert stats_l[-4:].view(torch.int32)[0].item() == 0                        # no hand-over to the exact-maximum pass
    assert torch.equal(l1, l2) and torch.equal(l2, l3) and not l1[:, C:].any()
    err = (l1[:, :C].float().cpu() - want).abs().max().item()
    assert err < 1e-2 * scale, (err, scale)
    assert (l1.float() - band.float()).abs().max().item() <= 2.0 ** -7 * scale   # two bf16 roundings of the same value


def test_loss_backward_band_kernel_hands_over_on_underflow():
    """Logit spreads beyond exp2's range: the band kernel must raise the retry flag and the exact-maximum pass must deliver."""
    from oracle import loss as OL
    from segmentation_factory_amd import hip
    B, C, h, w = 2, 150, 8, 8
    H, W = 4 * h, 4 * w
    g = torch.Generator().manual_seed(12)
    t = torch.randint(0, C, (B, H, W), generator=g)
    lo = (torch.randn(B, C, h, w, generator=g) * 150).to(torch.bfloat16)
    lr = lo.float().requires_grad_(True)
    up = F.interpolate(lr, size=(H, W), mode='bilinear', align_corners=False)
    OL.criterion_closed_form(up, t, None, num_classes=C, dice=True, ignore_index=255).backward()
    buf = torch.zeros(B * h * w, 160, dtype=torch.bfloat16, device='cuda')
    buf[:, :C] = lo.permute(0, 2, 3, 1).reshape(B * h * w, C).cuda()
    loss, stats = hip.ce_dice_fwd(buf[:, :C], B, C, h, w, H, W, t.cuda(), 255, None, True)
    d = hip.ce_dice_bwd(buf[:, :C], B, C, h, w, H, W, t.cuda(), 255, None, True, stats, torch.ones(1, device='cuda'))
    torch.cuda.synchronize()
    assert stats[-4:].view(torch.int32)[0].item() != 0
    want = lr.grad.permute(0, 2, 3, 1).reshape(B * h * w, C)
    assert (d[:, :C].float().cpu() - want).abs().max().item() < 1e-2 * want.abs().max().item()
    # with the per-pixel log-sum hand-over: the forward marks its buffer unusable (retry[1]), the backward's band kernel steps
    # aside without touching it and the exact-maximum pass delivers the same gradient
    loss_l, stats_l, lse = hip.ce_dice_fwd(buf[:, :C], B, C, h, w, H, W, t.cuda(), 255, None, True, want_lse=True)
    assert stats_l[-4:].view(torch.int32)[1].item() != 0 and torch.equal(loss_l, loss)
    dl = hip.ce_dice_bwd(buf[:, :C], B, C, h, w, H, W, t.cuda(), 255, None, True, stats_l, torch.ones(1, device='cuda'), lse=lse)
    torch.cuda.synchronize()
    assert stats_l[-4:].view(torch.int32)[0].item() != 0
    assert torch.equal(dl, d)


def test_argmax_confmat_and_metrics_golden(golden_dir):
    from segmentation_factory_amd import utils
    from segmentation_factory_amd.metrics import Metrics
    g = np.load(os.path.join(golden_dir, 'metrics_case.npz'))
    nc = int(g['nc'])
    m = Metrics(nc, 255, 'cuda')
    cm = utils.ConfusionMatrix(nc)
    for b in range(2):
        logits = torch.from_numpy(g[f'logits_{b}']).cuda()
        t = torch.from_numpy(g[f'target_{b}']).cuda()
        cm.update(t.flatten(), logits.argmax(1).flatten())
        m.update(logits, t.flatten())
    assert np.array_equal(cm.mat.cpu().numpy(), g['mat'])
    assert np.array_equal(m.hist.cpu().numpy(), g['hist'])
    iou, f1, acc = m.compute_iou(), m.compute_f1(), m.compute_pixel_acc()
    assert iou[1] == float(g['miou']) and f1[1] == float(g['mf1']) and acc[1] == float(g['macc'])
    assert np.allclose(np.array(iou[0]), g['iou'], equal_nan=True)
    assert str(cm) == str(g['confmat_str'])


def _fused_counts(lo, t, C, H, W):
    from segmentation_factory_amd.backbones import TokenMap
    from segmentation_factory_amd.metrics import Metrics
    from segmentation_factory_amd import utils
    B, _, h, w = lo.shape
    m, cm = Metrics(C, 255, 'cuda'), utils.ConfusionMatrix(C)
    tm = TokenMap(lo.permute(0, 2, 3, 1).reshape(B * h * w, C).contiguous().cuda(), B, h, w)
    m.update_lowres(tm, t.cuda(), (H, W), confmat=cm)
    return cm.mat.cpu().numpy(), m.hist.cpu().numpy()


@pytest.mark.parametrize('geom', [(2, 150, 8, 8, 32, 32), (2, 19, 16, 16, 64, 64), (1, 7, 4, 6, 16, 24), (2, 21, 16, 16, 32, 32),
                                  (1, 5, 4, 4, 32, 32)])
def test_argmax_confmat_fused_upsample(geom):
    """evaluate()'s default path (fused upsample + arg max + counting, Metrics.update_lowres) against the reference's op
    sequence F.interpolate -> argmax -> bincount on the CPU: EXACT (integer work).  The kernel interpolates in ATen's
    operation order (common.h bilinear_aten), which is bit-identical to F.interpolate at these map sizes."""
    from oracle import loss as OL
    B, C, h, w, H, W = geom
    g = torch.Generator().manual_seed(10)
    lo = torch.randn(B, C, h, w, generator=g)
    t = torch.randint(0, C, (B, H, W), generator=g)
    t[:, :3] = 255
    up = F.interpolate(lo, size=(H, W), mode='bilinear', align_corners=False)
    mat, hist = OL.confusion_counts(up, t, C, 255)
    got_mat, got_hist = _fused_counts(lo, t, C, H, W)
    assert np.array_equal(got_mat, mat)
    assert np.array_equal(got_hist, hist)


def test_argmax_tie_policy(hipmod):
    """Tie rule of the evaluation kernels, on constructed ties and near-ties.
    (1) exact ties (a class channel duplicated at a higher index) resolve to the LOWEST index, like torch.argmax;
    (2) the materialised logits (segf_bilinear_to_nchw_f32 = SegmentationModel.forward's resize) are bit-identical to
        F.interpolate where ATen uses its scalar loop (small maps, power-of-two ratio);
    (3) at BASELINE size (128 -> 512, where ATen's vectorised loops associate the four products differently) every pixel whose
        prediction differs from torch's is a NEAR-TIE: the reference's own top-2 margin there is below 4 ulp of the maximum."""
    hip = hipmod
    g = torch.Generator().manual_seed(12)
    # (1) + (2)
    B, C, h, w, H, W = 2, 12, 8, 8, 32, 32
    lo = torch.randn(B, C, h, w, generator=g)
    lo[:, 7] = lo[:, 2]                                   # exact tie between classes 2 and 7 wherever they are the maximum
    lo[:, 2] += 3.0                                       # ... and make that pair the maximum almost everywhere
    lo[:, 7] += 3.0
    up = F.interpolate(lo, size=(H, W), mode='bilinear', align_corners=False)
    tok = lo.permute(0, 2, 3, 1).reshape(B * h * w, C).contiguous().cuda()
    mine = hip.bilinear_to_nchw_f32(tok, B, h, w, C, H, W).cpu()
    # same operation order as ATen's scalar loop: bit-identical on the build container's CPU; ATen picks its loop by CPU
    # capability and thread count, so on other hosts allow the last rounding (1 ulp of the largest tap) and report
    nbits = int((mine != up).sum())
    print(f'bilinear_to_nchw vs F.interpolate on this host: {nbits} of {up.numel()} values differ in the last bit')
    assert (mine - up).abs().max() <= 2.0 ** -22 * up.abs().max()
    t = torch.randint(0, C, (B, H, W), generator=g)
    pred = torch.empty((B, H, W), dtype=torch.int64, device='cuda')
    z = torch.zeros((C, C), dtype=torch.int64, device='cuda')
    hip.argmax_confmat(tok, B, C, h, w, H, W, t.cuda(), 255, z, z.clone(), torch.zeros(1, dtype=torch.int32, device='cuda'), pred)
    # the fused kernel and the materialised logits are the SAME numbers: identical predictions, ties included
    assert torch.equal(pred.cpu(), mine.argmax(1))
    ref = up.argmax(1)
    assert (pred.cpu() == 7).sum() == 0 and (ref == 7).sum() == 0           # the duplicate at the higher index never wins
    assert (pred.cpu() == 2).sum() > 0.9 * ref.numel()
    assert torch.equal(pred.cpu(), ref)                                      # classes 2 / 7 tie exactly under any association
    # (3)
    B, C, h, w, H, W = 2, 150, 128, 128, 512, 512
    lo = torch.randn(B, C, h, w, generator=g)
    lo[:, 100] = lo[:, 3] + 1e-7 * torch.randn(B, h, w, generator=g)      # a near-tie pair on top of the random field
    up = F.interpolate(lo, size=(H, W), mode='bilinear', align_corners=False)
    tok = lo.permute(0, 2, 3, 1).reshape(B * h * w, C).contiguous().cuda()
    t = torch.randint(0, C, (B, H, W), generator=g)
    pred = torch.empty((B, H, W), dtype=torch.int64, device='cuda')
    z = torch.zeros((C, C), dtype=torch.int64, device='cuda')
    hip.argmax_confmat(tok, B, C, h, w, H, W, t.cuda(), 255, z, z.clone(), torch.zeros(1, dtype=torch.int32, device='cuda'), pred)
    ref = up.argmax(1)
    diff = pred.cpu() != ref
    top2 = up.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]) / top2[:, 0].abs().clamp_min(1e-30)
    ulp = 2.0 ** -23
    assert (margin[diff] <= 4 * ulp).all(), margin[diff].max()
    # the predictions at those pixels are the runner-up of a near-tie, not something else
    idx2 = up.topk(2, dim=1).indices
    pd = pred.cpu()[diff]
    assert ((pd == idx2[:, 0][diff]) | (pd == idx2[:, 1][diff])).all()
    print(f'128 -> 512: {int(diff.sum())} of {diff.numel()} predictions differ from torch (all near-ties, margin <= 4 ulp)')
    assert diff.float().mean().item() < 1e-3


def test_agc_adamw_known_answers(hipmod):
    """Fused AGC + AdamW kernel against the CPU restatement in oracle/optim.py (timm-0.9.2 adaptive_clip_grad + torch AdamW;
    parity unpinned: timm is not installed in this image)."""
    from oracle import optim as OO
    from segmentation_factory_amd.optim import FusedAGCAdamW
    g = torch.Generator().manual_seed(11)
    w = torch.nn.Parameter(torch.randn(6, 10, generator=g).cuda())
    b = torch.nn.Parameter(torch.randn(6, generator=g).cuda())
    pw, pb = w.detach().cpu().clone(), b.detach().cpu().clone()
    opt = FusedAGCAdamW([{'params': [b], 'weight_decay': 0.}, {'params': [w], 'weight_decay': 0.05}], lr=1e-2)
    mw, vw, mb, vb = torch.zeros_like(pw), torch.zeros_like(pw), torch.zeros_like(pb), torch.zeros_like(pb)
    for step in range(1, 4):
        gw, gb = torch.randn(6, 10, generator=g) * 3, torch.randn(6, generator=g) * 3
        w.grad, b.grad = gw.cuda(), gb.cuda()
        opt.agc_clip = 0.02
        opt.step()
        for (p, gr, m_, v_, wd) in ((pw, gw, mw, vw, 0.05), (pb, gb, mb, vb, 0.)):
            OO.adamw_step_(p, OO.adaptive_clip_grad_(p, gr, 0.02), m_, v_, step, 1e-2, weight_decay=wd)
        assert (w.detach().cpu() - pw).abs().max() < 1e-5
        assert (b.detach().cpu() - pb).abs().max() < 1e-5


def test_fused_adamw_skips_grad_none_like_torch_adamw(hipmod):
    """torch.optim.AdamW (the reference's optimizer, train_gpu.py:269) does `if p.grad is None: continue`: such a parameter gets
    no weight decay and no moment update in that step.  Eager path of FusedAGCAdamW (gather_grads sets the kernel's per-unit skip
    bit), with the set of gradient-less parameters CHANGING between steps."""
    from segmentation_factory_amd.optim import FusedAGCAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(5, 7), (5,), (4, 3, 2, 2), (9,)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    groups = lambda l: [{'params': [p for p in l if p.ndim <= 1], 'weight_decay': 0.}, {'params': [p for p in l if p.ndim > 1], 'weight_decay': 0.1}]
    opt, ropt = FusedAGCAdamW(groups(ps), lr=1e-2), torch.optim.AdamW(groups(ref), lr=1e-2)
    opt.set_clipping(None, 'agc')
    # parameter 2 never gets a gradient; parameter 1 misses one step and then continues with ITS OWN step count (torch keeps
    # state['step'] per parameter: the kernel keeps per-unit counts on the device and derives the bias corrections from them)
    for step, have in enumerate([(0, 1, 3), (0, 1, 3), (0, 3), (0, 1, 3)]):
        for i, (p, r) in enumerate(zip(ps, ref)):
            if i in have:
                gr = torch.randn(*shapes[i], generator=g)
                p.grad, r.grad = gr.cuda(), gr.clone()
            else:
                p.grad, r.grad = None, None
        opt.step()
        ropt.step()
        for i, (p, r) in enumerate(zip(ps, ref)):
            assert (p.detach().cpu() - r.detach()).abs().max() < 2e-6, (step, i)
    assert torch.equal(ps[2].detach().cpu(), ref[2].detach())         # untouched, bit for bit
    assert len(opt.state_dict()['state']) == 3


@pytest.mark.parametrize('mode,value', [('norm', 0.5), ('norm', 1e4), ('value', 0.3)])
def test_clip_modes_norm_and_value_vs_torch(hipmod, mode, value):
    """The reference's other --clip-mode values (train_gpu.py:99-102 -> timm dispatch_clip_grad = torch.nn.utils.clip_grad_norm_ /
    clip_grad_value_) through NativeScaler + FusedAGCAdamW, against torch's own clipping + torch.optim.AdamW on the CPU; and the
    flat-buffer kernel alone on a buffer large enough to use every workgroup slice (bitwise reproducible)."""
    from segmentation_factory_amd.optim import FusedAGCAdamW, NativeScaler
    g = torch.Generator().manual_seed(5)
    shapes = [(6, 10), (6,), (3, 4, 2, 2), (17,)]
    ps = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt = FusedAGCAdamW([{'params': [p for p in ps if p.ndim <= 1], 'weight_decay': 0.},
                         {'params': [p for p in ps if p.ndim > 1], 'weight_decay': 0.05}], lr=1e-2)
    ropt = torch.optim.AdamW([{'params': [p for p in ref if p.ndim <= 1], 'weight_decay': 0.},
                              {'params': [p for p in ref if p.ndim > 1], 'weight_decay': 0.05}], lr=1e-2)
    scaler = NativeScaler()
    for step in range(3):
        coefs = [torch.randn(*s, generator=g) * 2 for s in shapes]
        opt.zero_grad()
        loss = sum((p * c.cuda()).sum() for p, c in zip(ps, coefs))
        scaler(loss, opt, clip_grad=value, clip_mode=mode, parameters=ps)
        rloss = sum((p * c).sum() for p, c in zip(ref, coefs))
        ropt.zero_grad()
        rloss.backward()
        if mode == 'norm':
            torch.nn.utils.clip_grad_norm_(ref, value, norm_type=2.0)
        else:
            torch.nn.utils.clip_grad_value_(ref, value)
        ropt.step()
        for p, r in zip(ps, ref):
            assert (p.detach().cpu() - r.detach()).abs().max() < 2e-6, (mode, step)
    big = torch.randn(5_000_011, generator=g)
    want = big.clone()
    if mode == 'norm':
        want *= min(1.0, value / (float(torch.linalg.vector_norm(big.double())) + 1e-6))
    else:
        want.clamp_(-value, value)
    a = hipmod.clip_grad(big.cuda(), mode, value).cpu()
    b = hipmod.clip_grad(big.cuda(), mode, value).cpu()
    assert torch.equal(a, b)
    assert (a - want).abs().max() <= 1e-6 * want.abs().max()


# ---- ConvNeXt / UPerNet kernels ------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('geom', [(2, 9, 13, 32), (1, 16, 16, 96), (2, 5, 3, 8), (1, 1, 1, 64)])
def test_dwconv7x7(dtype, geom):
    from segmentation_factory_amd import functional as Fh
    B, H, W, C = geom
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, 7, 7, generator=g) * 0.2
    b = torch.randn(C, generator=g) * 0.1
    dy = torch.randn(B, C, H, W, generator=g)
    xr = _q(x, dtype).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=3, groups=C)
    ref.backward(_q(dy, dtype))
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(B * H * W, C)   # noqa: E731
    xd = _dev(tok(x), dtype).requires_grad_(True)
    wd, bd = _dev(w).requires_grad_(True), _dev(b).requires_grad_(True)
    y = Fh.dwconv7x7(xd, wd, bd, B, H, W)
    y.backward(_dev(tok(dy), dtype))
    _close(y, tok(ref), dtype)
    _close(xd.grad, tok(xr.grad), dtype)
    _close(wd.grad, wr.grad, dtype, fac=4)
    _close(bd.grad, br.grad, dtype, fac=4)


@pytest.mark.parametrize('cfg', [(2, 8, 8, 16, 24), (1, 5, 7, 64, 8), (2, 16, 12, 128, 136), (1, 1, 1, 8, 8),
                                 (3, 128, 128, 136, 264),       # runs on the 256x256 tile kernel in all three modes
                                 (4, 96, 128, 256, 512),        # forward and data gradient on the eight-phase kernel (gemm8.hip): whole
                                 (2, 128, 128, 512, 256),       # 256^2 tiles, channels % 128 == 0; borders on all four sides of every image
                                 (8, 96, 128, 256, 256)])       # >= 65536 pixels: the weight gradient on the eight-phase kernel too (28 K slices)
def test_conv3x3_implicit_gemm(cfg):
    """bf16 implicit-GEMM 3x3 conv (forward, data gradient, weight gradient) vs F.conv2d on bf16-rounded inputs; the input
    is a column slice of a wider buffer as in the UPerNet concat."""
    from segmentation_factory_amd import functional as Fh
    B, H, W, I, O = cfg
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, I, H, W, generator=g)
    w = torch.randn(O, I, 3, 3, generator=g) * (2.0 / (9 * I)) ** 0.5
    dy = torch.randn(B, O, H, W, generator=g)
    xr = _q(x, dtype).requires_grad_(True)
    wr = _q(w, dtype).requires_grad_(True)
    ref = F.conv2d(xr, wr, None, padding=1)
    ref.backward(_q(dy, dtype))
    tok = lambda t, c: t.permute(0, 2, 3, 1).reshape(B * H * W, c)   # noqa: E731
    buf = torch.zeros(B * H * W, I + 16, dtype=dtype, device='cuda')
    buf[:, 8:8 + I] = tok(x, I).to(dtype).cuda()
    xd = buf[:, 8:8 + I].detach().requires_grad_(True)
    wd = _dev(w).requires_grad_(True)
    y = Fh.conv3x3(xd, wd, B, H, W)
    y.backward(_dev(tok(dy, O), dtype))
    _close(y, tok(ref, O), dtype)
    _close(xd.grad, tok(xr.grad, I), dtype)
    _close(wd.grad, wr.grad, dtype, fac=2)


@pytest.mark.parametrize('dtype', DTYPES)
def test_gelu_avgpool_layerscale(dtype):
    from segmentation_factory_amd import functional as Fh
    g = torch.Generator().manual_seed(8)
    u = torch.randn(64, 48, generator=g) * 2
    dy = torch.randn(64, 48, generator=g)
    ur = _q(u, dtype).requires_grad_(True)
    F.gelu(ur).backward(_q(dy, dtype))
    ud = _dev(u, dtype).requires_grad_(True)
    y = Fh.gelu(ud)
    y.backward(_dev(dy, dtype))
    _close(y, F.gelu(_q(u, dtype)), dtype)
    _close(ud.grad, ur.grad, dtype)
    # adaptive average pooling, every PPM scale, non-divisible sizes
    B, H, W, C = 2, 7, 5, 16
    x = torch.randn(B, C, H, W, generator=g)
    for S in (1, 2, 3, 6):
        xr = _q(x, dtype).requires_grad_(True)
        ref = F.adaptive_avg_pool2d(xr, S)
        d = torch.randn(B, C, S, S, generator=g)
        ref.backward(_q(d, dtype))
        xd = _dev(x.permute(0, 2, 3, 1).reshape(B * H * W, C), dtype).requires_grad_(True)
        yp = Fh.adaptive_avgpool(xd, B, H, W, S)
        yp.backward(_dev(d.permute(0, 2, 3, 1).reshape(B * S * S, C), dtype))
        _close(yp, ref.permute(0, 2, 3, 1).reshape(B * S * S, C), dtype)
        _close(xd.grad, xr.grad.permute(0, 2, 3, 1).reshape(B * H * W, C), dtype)
    # linear + layer scale + residual + per-sample DropPath scale (ConvNeXt block tail)
    M, K, N, Bn = 96, 64, 24, 3
    x = torch.randn(M, K, generator=g)
    res = torch.randn(M, N, generator=g)
    w = torch.randn(N, K, generator=g) * 0.1
    b = torch.randn(N, generator=g) * 0.1
    gam = torch.rand(N, generator=g) + 0.5
    rs = torch.tensor([0.0, 1.25, 1.25])
    d = torch.randn(M, N, generator=g)
    xr, rr = _q(x, dtype).requires_grad_(True), _q(res, dtype).requires_grad_(True)
    wr, br, gr = w.clone().requires_grad_(True), b.clone().requires_grad_(True), gam.clone().requires_grad_(True)
    ref = rr + rs.repeat_interleave(M // Bn)[:, None] * (gr * F.linear(xr, wr, br))
    ref.backward(_q(d, dtype))
    xd, rd = _dev(x, dtype).requires_grad_(True), _dev(res, dtype).requires_grad_(True)
    wd, bd, gd = _dev(w).requires_grad_(True), _dev(b).requires_grad_(True), _dev(gam).requires_grad_(True)
    yl = Fh.linear_layer_scale(xd, wd, bd, gd, residual=rd, rscale=_dev(rs), rows_per_group=M // Bn)
    yl.backward(_dev(d, dtype))
    _close(yl, ref, dtype, fac=2)
    _close(xd.grad, xr.grad, dtype, fac=2)
    _close(rd.grad, rr.grad, dtype)
    _close(wd.grad, wr.grad, dtype, fac=4)
    _close(bd.grad, br.grad, dtype, fac=4)
    _close(gd.grad, gr.grad, dtype, fac=4)


@pytest.mark.parametrize('dtype', DTYPES)
def test_grn(dtype):
    """ConvNeXtV2 GRN (convnextv2.py:68-80) forward/backward incl. gamma / beta gradients and an all-zero channel."""
    from segmentation_factory_amd import functional as Fh
    g = torch.Generator().manual_seed(12)
    B, H, W, C = 3, 5, 7, 48
    x = torch.randn(B, H, W, C, generator=g)
    x[..., 5] = 0
    gam = torch.randn(1, 1, 1, C, generator=g) * 0.5
    bet = torch.randn(1, 1, 1, C, generator=g) * 0.1
    dy = torch.randn(B, H, W, C, generator=g)
    xr = _q(x, dtype).requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    Gx = torch.norm(xr, p=2, dim=(1, 2), keepdim=True)
    Nx = Gx / (Gx.mean(dim=-1, keepdim=True) + 1e-6)
    ref = gr * (xr * Nx) + br + xr
    ref.backward(_q(dy, dtype))
    xd = _dev(x.reshape(-1, C), dtype).requires_grad_(True)
    gd, bd = _dev(gam).requires_grad_(True), _dev(bet).requires_grad_(True)
    y = Fh.grn(xd, gd, bd, B, H * W)
    y.backward(_dev(dy.reshape(-1, C), dtype))
    _close(y, ref.reshape(-1, C), dtype)
    _close(xd.grad, xr.grad.reshape(-1, C), dtype, fac=2)
    _close(gd.grad, gr.grad, dtype, fac=4)
    _close(bd.grad, br.grad, dtype, fac=4)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_grn_with_gelu_applied_inside(dtype):
    """act -> grn of the ConvNeXtV2 block (convnextv2.py:92-94) as ONE op: Fh.grn(u, ..., pre_gelu=True) = GRN(gelu(u)) with the
    GELU applied inside the GRN kernels (gelu(u) never materialised) against torch's nn.GELU + the GRN formula, forward and all
    three gradients (the gradient of the PRE-activation u); and against the two-op path of this library."""
    from segmentation_factory_amd import functional as Fh
    g = torch.Generator().manual_seed(14)
    B, H, W, C = 3, 6, 5, 64
    u = torch.randn(B, H, W, C, generator=g) * 1.5
    u[..., 7] = 0
    gam = torch.randn(1, 1, 1, C, generator=g) * 0.5
    bet = torch.randn(1, 1, 1, C, generator=g) * 0.1
    dy = torch.randn(B, H, W, C, generator=g)
    ur = _q(u, dtype).requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    a = F.gelu(ur)
    Gx = torch.norm(a, p=2, dim=(1, 2), keepdim=True)
    Nx = Gx / (Gx.mean(dim=-1, keepdim=True) + 1e-6)
    ref = gr * (a * Nx) + br + a
    ref.backward(_q(dy, dtype))
    outs = []
    for fused in (True, False):
        ud = _dev(u.reshape(-1, C), dtype).requires_grad_(True)
        gd, bd = _dev(gam).requires_grad_(True), _dev(bet).requires_grad_(True)
        y = Fh.grn(ud, gd, bd, B, H * W, pre_gelu=True) if fused else Fh.grn(Fh.gelu(ud), gd, bd, B, H * W)
        y.backward(_dev(dy.reshape(-1, C), dtype))
        _close(y, ref.reshape(-1, C), dtype, fac=2)
        _close(ud.grad, ur.grad.reshape(-1, C), dtype, fac=3)
        _close(gd.grad, gr.grad, dtype, fac=4)
        _close(bd.grad, br.grad, dtype, fac=4)
        outs.append((y.detach().float(), ud.grad.float()))
    if dtype == torch.float32:       # same arithmetic, other rounding points: the two paths of this library agree to fp32 round-off
        assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-5 * outs[1][0].abs().max().item()
        assert (outs[0][1] - outs[1][1]).abs().max().item() <= 1e-5 * outs[1][1].abs().max().item()


def test_wave_reduce16_transposing_reduction(hipmod):
    """The permlane32/16-swap + DPP reduction the loss kernels use for 16 pixels at a time (exact integer data)."""
    x = torch.randint(-50, 50, (64, 16)).float()
    out = torch.empty(16, device='cuda')
    xd = x.cuda()
    rc = hipmod.lib().segf_debug_wave_reduce16(xd.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.equal(out.cpu(), x.sum(0))


@pytest.mark.parametrize('layout', [0, 1, 2])
def test_gemm_big_tile_variant(hipmod, layout):
    """256x256 tile kernel (chosen for M, N > 128 with many tiles / token-count K): odd sizes, bias + residual epilogue,
    split-K, against a float64 product of the bf16-rounded operands."""
    g = torch.Generator().manual_seed(30 + layout)
    if layout == 2:
        M, N, K, sk = 200, 392, 20000, 5
    else:
        M, N, K, sk = 256 * 12 + 72, 256 * 17 - 40, 200, 1
    a = torch.randn((K, M) if layout == 2 else (M, K), generator=g)
    b = torch.randn((N, K) if layout == 0 else (K, N), generator=g)
    aq, bq = a.bfloat16().double(), b.bfloat16().double()
    A = aq.t() if layout == 2 else aq
    Bm = bq.t() if layout == 0 else bq
    ref = A @ Bm
    ad, bd = a.bfloat16().cuda(), b.bfloat16().cuda()
    if layout == 2:
        out = hipmod.gemm(2, ad, bd, M, N, K, out_dtype=torch.float32, split_k=sk)
        err = (out.double().cpu() - ref).abs().max().item()
        assert err <= 2e-3 * ref.abs().max().item()
    else:
        bias = torch.randn(N, generator=g)
        res = torch.randn(M, N, generator=g)
        out = hipmod.gemm(layout, ad, bd, M, N, K, bias=bias.cuda(), residual=res.bfloat16().cuda())
        full = ref + bias.double() + res.bfloat16().double()
        err = (out.double().cpu() - full).abs().max().item()
        assert err <= 1.2e-2 * full.abs().max().item()       # one bf16 rounding of the output


@pytest.mark.parametrize('ragged', [False, True])
@pytest.mark.parametrize('layout', [0, 1, 2])
def test_gemm_eight_phase_tile(hipmod, layout, ragged):
    """gemm8.hip (256 x 256 tile in eight phases: LDS-DMA staging, counted waits) on shapes of the ConvNeXt / MiT linears -- since r05
    the kernel plain products with K >= 512 take by default: layout 0 with bias + residual + per-sample DropPath scale, layout 1 with a
    residual, layout 2 with split-K; whole tiles and (layouts 0 / 1) a RAGGED last row and column tile ([M x N] = [50 tiles + 72, 4 tiles
    - 40]: rows / column chunks past the end are clamped on the load side and not stored).  Against a float64 product of the bf16-rounded
    operands and against the two-phase 256-tile kernel (policy gemm8_linear = 0)."""
    if layout == 2 and ragged:
        pytest.skip('weight gradients: whole tiles only')
    g = torch.Generator().manual_seed(90 + layout)
    if layout == 2:
        M, N, K = 768, 3072, 65536 + 64 * 5
        sk = hipmod.pick_splitk(M, N, K)
    else:
        M, N, K, sk = 256 * 50 + (72 if ragged else 0), 256 * 4 - (40 if ragged else 0), 64 * 8 if layout == 0 else 64 * 12, 1
    a = torch.randn((K, M) if layout == 2 else (M, K), generator=g)
    b = torch.randn((N, K) if layout == 0 else (K, N), generator=g)
    aq, bq = a.bfloat16().double(), b.bfloat16().double()
    A = aq.t() if layout == 2 else aq
    Bm = bq.t() if layout == 0 else bq
    ref = A @ Bm
    ad, bd = a.bfloat16().cuda(), b.bfloat16().cuda()
    groups = (M + 255) // 256

    def run():
        if layout == 2:
            return hipmod.gemm(2, ad, bd, M, N, K, out_dtype=torch.float32, split_k=sk)
        bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).cuda() if layout == 0 else None
        res = torch.randn(M, N, generator=torch.Generator().manual_seed(6)).bfloat16().cuda()
        rs = (torch.rand(groups, generator=torch.Generator().manual_seed(7)) + 0.5).cuda() if layout == 0 else None
        return hipmod.gemm(layout, ad, bd, M, N, K, bias=bias, residual=res, rscale=rs, rows_per_group=256 if layout == 0 else 1)
    with hipmod.policy_override(gemm8_linear_min_gflop=0), hipmod.trace() as tr:          # (the dispatch rule wants >= 36 GFLOP; this is 13)
        out = run()
    assert any('gemm8_kernel' in k for k in tr.kernels), tr.kernels
    with hipmod.policy_override(gemm8_linear=0), hipmod.trace() as tr0:
        old = run()
    assert not any('gemm8_kernel' in k for k in tr0.kernels), tr0.kernels
    if layout == 2:
        assert (out.double().cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
        assert (out - old).abs().max().item() <= 1e-3 * old.abs().max().item()
    else:
        bias = torch.randn(N, generator=torch.Generator().manual_seed(5)).double() if layout == 0 else 0.0
        res = torch.randn(M, N, generator=torch.Generator().manual_seed(6)).bfloat16().double()
        rs = (torch.rand(groups, generator=torch.Generator().manual_seed(7)) + 0.5).double().repeat_interleave(256)[:M, None] if layout == 0 else 1.0
        full = res + rs * (ref + bias)
        assert (out.double().cpu() - full).abs().max().item() <= 1.2e-2 * full.abs().max().item()
        assert (out.float() - old.float()).abs().max().item() <= 2 ** -7 * old.float().abs().max().item()
    if ragged:           # nothing is written past the ragged edges: the same product into a view of a larger buffer filled with a sentinel
        big = torch.full((M + 256, N + 264), 7.0, dtype=torch.bfloat16, device='cuda')
        with hipmod.policy_override(gemm8_linear_min_gflop=0), hipmod.trace() as tr:
            hipmod.gemm(layout, ad, bd, M, N, K, out=big[:M, :N])
        assert any('gemm8_kernel' in k for k in tr.kernels), tr.kernels
        assert bool((big[M:] == 7.0).all()) and bool((big[:, N:] == 7.0).all())
        plain = ref.float().bfloat16()
        assert (big[:M, :N].double().cpu() - ref).abs().max().item() <= 1.2e-2 * ref.abs().max().item(), plain.shape


def test_gemm_streaming_whole_rows(hipmod, monkeypatch):
    """[M x 32] -> 768 with M >= 65536 (the folded head's stage-1 projection): the whole-row form (weights in LDS, a wave writes
    complete 1536-byte rows) against fp64 and against the column-chunk form (switch): identical products, identical rounding."""
    M, K, N = 65536 + 16 * 37 + 5, 32, 768
    g = torch.Generator().manual_seed(91)
    a = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    out = hipmod.gemm(0, ad, wd, M, N, K, bias=bd)
    monkeypatch.setenv('SEGFAC_GEMM_NO_SKINNY_ROWS', '1')
    out2 = hipmod.gemm(0, ad, wd, M, N, K, bias=bd)
    monkeypatch.delenv('SEGFAC_GEMM_NO_SKINNY_ROWS')
    nob = hipmod.gemm(0, ad, wd, M, N, K)
    torch.cuda.synchronize()
    assert (out.double().cpu() - ref).abs().max().item() <= 1.2e-2 * ref.abs().max().item()
    assert torch.equal(out, out2)
    assert (nob.double().cpu() - (ref - bias.double())).abs().max().item() <= 1.2e-2 * ref.abs().max().item()


@pytest.mark.parametrize('cfg', [(1, 256, 32, False), (1, 768, 32, False), (1, 1024, 32, False), (1, 768, 64, False),
                                 (1, 256, 64, True), (0, 256, 64, True), (0, 512, 32, True), (0, 768, 64, False)])
def test_gemm_streaming_wide_k(hipmod, monkeypatch, cfg):
    """y[M][N] = x[M][K] W with N = 32 / 64, K = 256 .. 1024, M >= 65536 (the folded head's stage-1 / stage-2 data gradients, the
    stage-2 MLP's fc2 forward with bias + residual + DropPath scale): gemm_skinny_k_kernel (four waves split K, partial tiles
    added in fixed order) against fp64, against the tiled kernel (switch), ragged last group, twice (bitwise reproducible)."""
    layout, K, N, epi = cfg
    M, rpg = 65536 + 16 * 11 + 7, 4096
    g = torch.Generator().manual_seed(92)
    x = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(*((N, K) if layout == 0 else (K, N)), generator=g) / K ** 0.5).bfloat16()
    ref = x.double() @ (W.double().t() if layout == 0 else W.double())
    kw = {}
    if epi:
        bias = torch.randn(N, generator=g)
        res = torch.randn(M, N, generator=g).bfloat16()
        rsc = torch.rand((M + rpg - 1) // rpg, generator=g) * 2
        ref = res.double() + rsc.double().repeat_interleave(rpg)[:M, None] * (ref + bias.double())
        kw = dict(bias=bias.cuda(), residual=res.cuda(), rscale=rsc.cuda(), rows_per_group=rpg)
    xd, Wd = x.cuda(), W.cuda()
    out = hipmod.gemm(layout, xd, Wd, M, N, K, **kw)
    out_b = hipmod.gemm(layout, xd, Wd, M, N, K, **kw)
    monkeypatch.setenv('SEGFAC_GEMM_NO_SKINNY_K', '1')
    out2 = hipmod.gemm(layout, xd, Wd, M, N, K, **kw)
    monkeypatch.delenv('SEGFAC_GEMM_NO_SKINNY_K')
    torch.cuda.synchronize()
    assert torch.equal(out, out_b)
    tol = 1.2e-2 * ref.abs().max().item()                                          # one bf16 rounding of the output
    assert (out.double().cpu() - ref).abs().max().item() <= tol
    assert (out2.double().cpu() - ref).abs().max().item() <= tol
    assert (out.float() - out2.float()).abs().max().item() <= tol


def test_stem_conv_streaming_form(hipmod, monkeypatch):
    """PatchEmbed's k7 s4 p3 conv of the fp32 NCHW image (mit.py:105) at a token count that takes the streaming forward product:
    im2col columns padded 147 -> 160 (zeros on both sides), gemm_skinny_kernel<0, 5, 2>.  Against F.conv2d on the bf16-rounded
    operands, and against the tiled kernel on the same padded operands (switch): forward, weight and bias gradients."""
    from segmentation_factory_amd import functional as Fh
    g = torch.Generator().manual_seed(17)
    B, H, W, O = 2, 384, 400, 32                                     # 2 * 96 * 100 = 19200 tokens (ragged last 16-token group: none)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(O, 3, 7, 7, generator=g) * 0.1
    b = torch.randn(O, generator=g) * 0.1
    dy = torch.randn(B, O, H // 4, W // 4, generator=g)
    wr, br = _q(w, torch.bfloat16).requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(_q(x, torch.bfloat16), wr, br, stride=4, padding=3)
    ref.backward(_q(dy, torch.bfloat16))
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, O)             # noqa: E731

    def run():
        wd, bd = _dev(w).requires_grad_(True), _dev(b).requires_grad_(True)
        y = Fh.conv_patch(_dev(x), wd, bd, (B, H, W, 3, 7, 4, 3), image=True, dtype=torch.bfloat16)
        y.backward(_dev(tok(dy), torch.bfloat16))
        return y.detach(), wd.grad, bd.grad
    y, dw, db = run()
    _close(y, tok(ref), torch.bfloat16)
    _close(dw, wr.grad, torch.bfloat16, fac=4)
    _close(db, br.grad, torch.bfloat16, fac=4)
    monkeypatch.setenv('SEGFAC_GEMM_NO_SKINNY', '1')
    y2, dw2, db2 = run()
    monkeypatch.delenv('SEGFAC_GEMM_NO_SKINNY')
    assert (y.float() - y2.float()).abs().max().item() <= 2e-2 * y2.float().abs().max().item()
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 70, 9, 8, True), (1, 5, 131, 16, True), (3, 16, 16, 40, False), (1, 130, 6, 136, True)])
def test_dwconv3x3_walk_equals_strip_form(dtype, shape, monkeypatch):
    """Depthwise 3x3 (+GELU) forward and backward: the vertical-walk kernels against the strip kernels they replace (switch), on
    heights that do not divide into the row segments, widths that are no multiple of the 4-pixel column and few / many channels.
    Same multiply-adds in another order: fp32 agrees to rounding, bf16 to one ulp of the stored result."""
    from segmentation_factory_amd import hip
    B, H, W, C, gelu = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B * H * W, C, generator=g).to(dtype).cuda()
    dy = torch.randn(B * H * W, C, generator=g).to(dtype).cuda()
    w9 = (torch.randn(C, 9, generator=g) * 0.3).cuda()
    bias = torch.randn(C, generator=g).cuda()
    outs = {}
    for name, env, rows in (('walk', None, None), ('walk5', None, '5'), ('strip', '1', None)):
        if env: monkeypatch.setenv('SEGFAC_DW_NO_WALK', env)
        if rows: monkeypatch.setenv('SEGFAC_DW_WALK_ROWS', rows)
        y = hip.dwconv3x3_gelu_fwd(x, w9, bias, B, H, W, C, gelu)
        dx, dw, db = hip.dwconv3x3_gelu_bwd(x, w9, bias, dy, B, H, W, C, gelu)
        torch.cuda.synchronize()
        outs[name] = [t.float().cpu().clone() for t in (y, dx, dw, db)]
        monkeypatch.delenv('SEGFAC_DW_NO_WALK', raising=False)
        monkeypatch.delenv('SEGFAC_DW_WALK_ROWS', raising=False)
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2
    for name in ('walk', 'walk5'):
        for a, b_, what in zip(outs[name], outs['strip'], ('y', 'dx', 'dw', 'db')):
            scale = b_.abs().max().item() + 1e-6
            assert (a - b_).abs().max().item() <= tol * scale, (name, what, (a - b_).abs().max().item(), scale)


@pytest.mark.parametrize('case', [(0, 256 * 40 + 72, 150, 768, 160), (0, 256 * 33, 160, 200, 160), (0, 256 * 34 + 8, 137, 136, 144),
                                  (0, 256 * 20 + 8, 150, 64 * 15, 160), (0, 256 * 9 + 200, 152, 64 * 19, 152), (0, 256 * 12, 144, 64 * 5, 144),   # (the three-deep pipeline: 12 straight-line steps + the loop; an early exit)
                                  (1, 256 * 50 + 24, 160, 640, 160), (1, 256 * 49, 152, 328, 152),
                                  (2, 150, 768, 70000, 160), (2, 160, 392, 66000, 160), (2, 131, 520, 65536 + 8, 136), (2, 200, 392, 70000, 200)])
def test_gemm_big_tile_narrow_shapes(hipmod, case, monkeypatch):
    """The 256-tile kernel's narrow wave shapes (N <= 160 in layout 0: 4 x 2 waves of 64 x 80; M <= 160 in layout 2: 2 x 4 waves of
    80 x 64 -- the classifier's 150 -> 160 classes) against fp64 and against the full-shape kernel (switch); padded row strides,
    ragged M / K, bias + residual epilogue (layout 0) and split-K partials (layout 2)."""
    layout, M, N, K, ld = case
    g = torch.Generator().manual_seed(5 + layout)
    if layout == 1:
        # dx = dy W with W [K][N] (reduction-major second operand), N <= 160
        a = torch.randn(M, K, generator=g).bfloat16()
        wb = (torch.randn(K, ld, generator=g) / K ** 0.5).bfloat16()
        ref = a.double() @ wb[:, :N].double()
        ad, wd = a.cuda(), wb.cuda()
        outs = []
        for env in (None, '1'):
            if env: monkeypatch.setenv('SEGFAC_GEMM_NO_NARROW', env)
            outs.append(hipmod.gemm(1, ad, wd[:, :N], M, N, K).clone())
        monkeypatch.delenv('SEGFAC_GEMM_NO_NARROW')
        for out in outs:
            assert (out.double().cpu() - ref).abs().max().item() <= 1.2e-2 * ref.abs().max().item()
        assert torch.equal(outs[0], outs[1])
    elif layout == 0:
        a = torch.randn(M, K, generator=g).bfloat16()
        wb = torch.zeros(ld, K).bfloat16()
        wb[:N] = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
        bias, res = torch.randn(N, generator=g), torch.randn(M, ld, generator=g).bfloat16()
        ref = a.double() @ wb[:N].double().t() + bias.double() + res[:, :N].double()
        ad, wd = a.cuda(), wb.cuda()
        outs = []
        for env in (None, '1'):
            if env: monkeypatch.setenv('SEGFAC_GEMM_NO_NARROW', env)
            out = torch.full((M, ld), 3.0, dtype=torch.bfloat16, device='cuda')
            hipmod.gemm(0, ad, wd[:N], M, N, K, out=out[:, :N], bias=bias.cuda(), residual=res.cuda()[:, :N])
            outs.append(out.clone())
        monkeypatch.delenv('SEGFAC_GEMM_NO_NARROW')
        for out in outs:
            assert (out[:, :N].double().cpu() - ref).abs().max().item() <= 1.2e-2 * ref.abs().max().item()
            assert (out[:, N:] == 3.0).all()                      # columns past N are not the kernel's to touch
        assert torch.equal(outs[0], outs[1])                      # same products in the same order, only dealt to other waves
    else:
        dy = torch.zeros(K, ld).bfloat16()
        dy[:, :M] = torch.randn(K, M, generator=g).bfloat16()
        dy[:, M:] = 5.0
        x = torch.randn(K, N, generator=g).bfloat16()
        ref = dy[:, :M].double().t() @ x.double()
        dyd, xd = dy.cuda(), x.cuda()
        split = hipmod.pick_splitk(M, N, K)
        outs = []
        for env in (None, '1'):
            if env: monkeypatch.setenv('SEGFAC_GEMM_NO_NARROW', env)
            outs.append(hipmod.gemm(2, dyd[:, :M], xd, M, N, K, out_dtype=torch.float32, split_k=split).clone())
        monkeypatch.delenv('SEGFAC_GEMM_NO_NARROW')
        for out in outs:
            assert (out.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-3
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('shape', [(32, 147, 131072, 152), (32, 32, 65536, 32), (32, 128, 65536, 128), (128, 32, 98304, 32),
                                   (64, 256, 65536, 256), (256, 64, 65536, 64), (64, 64, 65536 + 4096, 72), (40, 100, 70000, 104)])
def test_gemm_streaming_weight_gradient(hipmod, shape, monkeypatch):
    """dW = dy^T x and db = column sums of dy for small outputs over many tokens (gemm_dw_skinny_kernel: a wave per K slice, both
    operands transposed through ds_read_b64_tr_b16, bias gradient as an all-ones column) against fp64 on the bf16-rounded
    operands; ragged K, padded rows, row / column blocking, and bitwise agreement of two runs.  The last shape's K does not
    divide into the kernel's slices evenly; whichever kernel takes it must still be right."""
    M, N, K, ldx = shape
    g = torch.Generator().manual_seed(77)
    dy = (torch.randn(K, M, generator=g) * 0.5).bfloat16()
    xb = torch.zeros(K, ldx).bfloat16()
    xb[:, :N] = torch.randn(K, N, generator=g).bfloat16()
    if ldx > N:
        xb[:, N:] = 7.0                      # padding columns must not leak into the product
    ref = dy.double().t() @ xb[:, :N].double()
    refb = dy.double().sum(0)
    dyd, xd = dy.cuda(), xb.cuda()
    split = hipmod.pick_splitk(M, N, K)
    dw, db = hipmod.gemm_dw_db(dyd, xd[:, :N], M, N, K, split_k=split)
    dw2, db2 = hipmod.gemm_dw_db(dyd, xd[:, :N], M, N, K, split_k=split)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    scale = ref.abs().max().item()
    assert (dw.double().cpu() - ref).abs().max().item() <= 2e-5 * scale + 1e-3, (dw.double().cpu() - ref).abs().max().item()
    assert (db.double().cpu() - refb).abs().max().item() <= 2e-5 * refb.abs().max().item() + 1e-3
    # the same through the tiled kernel (switch) and through plain segf_gemm layout 2
    monkeypatch.setenv('SEGFAC_GEMM_NO_DW_SKINNY', '1')
    split_t = hipmod.pick_splitk(M, N, K)
    dwt, dbt = hipmod.gemm_dw_db(dyd, xd[:, :N], M, N, K, split_k=split_t)
    monkeypatch.delenv('SEGFAC_GEMM_NO_DW_SKINNY')
    assert (dwt.double().cpu() - ref).abs().max().item() <= 2e-5 * scale + 1e-3
    # (tiled kernel: bias gradient from an all-ones column when N leaves one free, from extra MFMAs when N % 128 == 0)
    assert (dbt.double().cpu() - refb).abs().max().item() <= 2e-5 * refb.abs().max().item() + 1e-3
    plain = hipmod.gemm(2, dyd, xd[:, :N], M, N, K, split_k=split, out_dtype=torch.float32)
    assert (plain.double().cpu() - ref).abs().max().item() <= 2e-5 * scale + 1e-3


@pytest.mark.parametrize('layout', [0, 1])
@pytest.mark.parametrize('shape', [(16391, 32, 32), (20000, 128, 32), (16400, 32, 128), (16384, 64, 64), (16390, 768, 32),
                                   (16384, 256, 64), (16392, 64, 128)])
def test_gemm_streaming_variant(hipmod, layout, shape):
    """Register-resident-weight streaming kernel (huge token count, K and N small): ragged token counts, bias + residual +
    per-row-group scale epilogue, strided operands, against a float64 product of the bf16-rounded operands; and the same
    call with the variant switched off must agree to bf16 rounding."""
    M, N, K = shape
    g = torch.Generator().manual_seed(40 + layout)
    a = torch.randn(M, K + 8, generator=g)[:, :K]                  # row stride K + 8
    b = torch.randn((N, K) if layout == 0 else (K, N), generator=g) / K ** 0.5
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    rpg = 4099
    rs = torch.rand((M + rpg - 1) // rpg, generator=g) * 2
    aq, bq = a.bfloat16().double(), b.bfloat16().double()
    ref = res.bfloat16().double() + rs.double().repeat_interleave(rpg)[:M, None] * (aq @ (bq.t() if layout == 0 else bq) + bias.double())
    ad = torch.randn(M, K + 8).bfloat16().cuda()
    ad[:, :K] = a.bfloat16().cuda()
    outs = []
    for off in ('0', '1'):
        os.environ['SEGFAC_GEMM_NO_SKINNY'] = off
        try:
            out = torch.zeros(M, N + 8, dtype=torch.bfloat16, device='cuda')
            hipmod.gemm(layout, ad[:, :K], b.bfloat16().cuda(), M, N, K, out=out[:, :N], bias=bias.cuda(),
                        residual=res.bfloat16().cuda(), rscale=rs.cuda(), rows_per_group=rpg)
        finally:
            os.environ.pop('SEGFAC_GEMM_NO_SKINNY', None)
        assert out[:, N:].abs().max().item() == 0
        err = (out[:, :N].double().cpu() - ref).abs().max().item()
        assert err <= 1.2e-2 * ref.abs().max().item()
        outs.append(out)
    assert (outs[0].float() - outs[1].float()).abs().max().item() <= 2e-2 * ref.abs().max().item()
    plain = hipmod.gemm(layout, ad[:, :K], b.bfloat16().cuda(), M, N, K)
    assert (plain.double().cpu() - aq @ (bq.t() if layout == 0 else bq)).abs().max().item() <= 1.2e-2 * ref.abs().max().item()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('geom', [(2, 16, 24, 16), (1, 8, 8, 40), (1, 12, 20, 8)])
def test_upsample_add_multi_scale(hipmod, dtype, geom):
    """out = base + sum_k bilinear_up(src_k) on the stride-4 grid (folded SegFormer head, heads/segformer.py:44-56): the
    specialised 1/2-1/4-1/8 strip kernel and the generic kernel against F.interpolate; (12, 20) is not a multiple of 8 and
    takes the generic kernel either way."""
    B, H, W, C = geom
    g = torch.Generator().manual_seed(50)
    base = torch.randn(B, C, H, W, generator=g)
    sizes = [(max(H // r, 1), max(W // r, 1)) for r in (2, 4, 8)]
    srcs = [torch.randn(B, C, h, w, generator=g) for (h, w) in sizes]
    ref = _q(base, dtype) + sum(F.interpolate(_q(s, dtype), size=(H, W), mode='bilinear', align_corners=False) for s in srcs)
    tok = lambda t: _dev(t.permute(0, 2, 3, 1).reshape(-1, C).contiguous(), dtype)
    for generic in ('', '1'):
        if generic:
            os.environ['SEGFAC_UPADD_GENERIC'] = '1'
        try:
            out = hipmod.upsample_add(tok(base), [(tok(s), h, w) for s, (h, w) in zip(srcs, sizes)], B, H, W, C)
        finally:
            os.environ.pop('SEGFAC_UPADD_GENERIC', None)
        _close(out, ref.permute(0, 2, 3, 1).reshape(-1, C), dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(32, 32, 20000), (150, 64, 4097), (768, 40, 30000), (200, 392, 20000), (128, 130, 777)])
def test_gemm_dw_db(hipmod, dtype, shape):
    """Weight gradient + bias gradient in one pass (column sums on the matrix pipe inside the layout-2 GEMM; the 256-tile and
    fp32 kernels fall back to a column reduction behind the product): against float64 on the rounded operands, with and
    without split-K, and with the fused path switched off."""
    M, N, K = shape
    g = torch.Generator().manual_seed(60)
    dy, x = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    dyq, xq = _q(dy, dtype).double(), _q(x, dtype).double()
    rw, rb = dyq.t() @ xq, dyq.sum(0)
    for sk in (1, hipmod.pick_splitk(M, N, K)):
        for off in ('', '1'):
            if off:
                os.environ['SEGFAC_GEMM_NO_FUSED_DB'] = '1'
            try:
                dw, db = hipmod.gemm_dw_db(_dev(dy, dtype), _dev(x, dtype), M, N, K, split_k=sk)
            finally:
                os.environ.pop('SEGFAC_GEMM_NO_FUSED_DB', None)
            assert (dw.double().cpu() - rw).abs().max().item() <= 2e-3 * rw.abs().max().item()
            assert (db.double().cpu() - rb).abs().max().item() <= 1e-4 * max(1.0, rb.abs().max().item())


def test_bn_act_linear_fused_into_gemm(hipmod):
    """BatchNorm + ReLU + Dropout2d scale applied while the GEMM stages its activation operand (segf_gemm_pro, forward and
    weight-gradient products) against the unfused batch_norm_act -> linear pair of this library, and against fp64 torch."""
    from segmentation_factory_amd import functional as Fh
    B, hw, K, N = 5, 16384, 256, 150
    M, Np = B * hw, 152
    g = torch.Generator().manual_seed(70)
    x = (torch.randn(M, K, generator=g) * 1.5 + 0.3).bfloat16().cuda()
    gam, bet = (1 + 0.2 * torch.randn(K, generator=g)).cuda(), (0.1 * torch.randn(K, generator=g)).cuda()
    w, bb = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(), torch.randn(N, generator=g).cuda()
    cs = ((torch.rand(B, K, generator=g) > 0.1).float() / 0.9).cuda()
    dy = torch.randn(M, N, generator=g).bfloat16().cuda()
    assert hipmod.gemm_pro_supported(torch.bfloat16, 0, M, Np, K, hw) and hipmod.gemm_pro_supported(torch.bfloat16, 2, Np, K, M, hw)
    outs = []
    for fused in (True, False):
        if not fused:
            os.environ['SEGFAC_GEMM_NO_PRO'] = '1'
        try:
            xs = x.clone().requires_grad_(True)
            ps = [t.clone().requires_grad_(True) for t in (gam, bet, w, bb)]
            rm, rv = torch.zeros(K, device='cuda'), torch.ones(K, device='cuda')
            y = Fh.bn_act_linear(xs, ps[0], ps[1], rm, rv, True, 0.1, 1e-5, 1, cs, hw, ps[2], ps[3], pad_to=Np)
            y.backward(dy)
        finally:
            os.environ.pop('SEGFAC_GEMM_NO_PRO', None)
        outs.append((y.detach().float(), xs.grad.float(), [p.grad.float() for p in ps], rm, rv))
    (y1, dx1, gp1, rm1, rv1), (y2, dx2, gp2, rm2, rv2) = outs
    assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
    assert (y1 - y2).abs().max().item() <= 2e-2 * y2.abs().max().item()
    assert (dx1 - dx2).abs().max().item() <= 2e-2 * dx2.abs().max().item()
    for a, b in zip(gp1, gp2):
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
    # float64 reference of the forward on the bf16-rounded input
    xd = x.double()
    mu, var = xd.mean(0), xd.var(0, unbiased=False)
    a = torch.relu((xd - mu) / torch.sqrt(var + 1e-5) * gam.double() + bet.double()) * cs.double().repeat_interleave(hw, 0)
    ref = a.bfloat16().double() @ w.bfloat16().double().t() + bb.double()
    assert (y1.double() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()


@pytest.mark.parametrize('dtype', DTYPES)
def test_upsample_add_with_batchnorm_statistics(hipmod, dtype):
    """segf_upsample_add_stats: same result as segf_upsample_add plus per-channel (sum, sum of squares) of the STORED values,
    against torch on the kernel's own output; segf_bn_stats_from_sums against segf_bn_stats."""
    B, H, W, C = 3, 16, 24, 104
    g = torch.Generator().manual_seed(80)
    base = torch.randn(B * H * W, C, generator=g)
    sizes = [(H // r, W // r) for r in (2, 4, 8)]
    srcs = [torch.randn(B * h * w, C, generator=g) for (h, w) in sizes]
    args = (_dev(base, dtype), [(_dev(s, dtype), h, w) for s, (h, w) in zip(srcs, sizes)], B, H, W, C)
    ref = hipmod.upsample_add(*args)
    out, sums = hipmod.upsample_add_stats(*args)
    assert sums is not None and (out.float() - ref.float()).abs().max().item() <= 1e-5 * ref.float().abs().max().item()
    o64 = out.double()
    assert torch.allclose(sums[0].double(), o64.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(sums[1].double(), (o64 * o64).sum(0), rtol=1e-5, atol=1e-3)
    rm1, rv1, rm2, rv2 = (torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros(C, device='cuda'),
                          torch.ones(C, device='cuda'))
    m1, r1 = hipmod.bn_stats(out, rm1, rv1, 0.1, 1e-5)
    m2, r2 = hipmod.bn_stats_from_sums(sums, B * H * W, rm2, rv2, 0.1, 1e-5)
    assert torch.allclose(m1, m2, rtol=1e-5, atol=1e-6) and torch.allclose(r1, r2, rtol=1e-4)
    assert torch.allclose(rm1, rm2, rtol=1e-5, atol=1e-6) and torch.allclose(rv1, rv2, rtol=1e-4)
    # a geometry outside the fused case reports "no sums"
    out2, sums2 = hipmod.upsample_add_stats(_dev(base, dtype), [(_dev(srcs[0], dtype), *sizes[0])], B, H, W, C)
    assert sums2 is None and out2.shape == out.shape


@pytest.mark.parametrize('geom', [(2, 8, 8, 128, 32), (1, 16, 24, 256, 64), (3, 32, 40, 768, 32), (2, 128, 128, 128, 32)])
def test_fuse_map_248_one_pass(hipmod, geom):
    """segf_fuse_map_248 (the folded SegFormerHead's stride-4 map, heads/segformer.py:42-56: x1 G1^T and the three bilinear resizes
    as ONE accumulated matrix product per 8 x 8 pixel block, with the BatchNorm statistics) against float64 torch on the same bf16
    operands (x1 @ G1^T + F.interpolate of the 1/2, 1/4, 1/8 maps, align_corners=False), and against the two launches it replaces.
    8 x 8 maps: every block is a border block (clamped source addresses on every side)."""
    B, H, W, C, C1 = geom
    g = torch.Generator().manual_seed(91)
    bf = torch.bfloat16
    x1 = torch.randn(B * H * W, C1, generator=g).to(bf)
    G1 = (torch.randn(C, C1, generator=g) * 0.3).to(bf)
    sizes = [(H // r, W // r) for r in (2, 4, 8)]
    ts = [(torch.randn(B * h * w, C, generator=g) * (1 + k)).to(bf) for k, (h, w) in enumerate(sizes)]
    assert hipmod.fuse_map_248_supported(bf, B, H, W, C, C1)
    out, sums = hipmod.fuse_map_248(x1.cuda(), G1.cuda(), *[t.cuda() for t in ts], B, H, W)
    ref = x1.double() @ G1.double().t()
    for t, (h, w) in zip(ts, sizes):
        up = F.interpolate(t.double().view(B, h, w, C).permute(0, 3, 1, 2), size=(H, W), mode='bilinear', align_corners=False)
        ref = ref + up.permute(0, 2, 3, 1).reshape(-1, C)
    scale = ref.abs().max().item()
    err = (out.double().cpu() - ref).abs().max().item()
    assert err <= 2 ** -8 * scale, (err, scale)                      # one bf16 rounding of an fp32-accumulated exact product
    # statistics: of the fp32 results (before the rounding to bf16)
    assert torch.allclose(sums[0].double().cpu(), ref.sum(0), rtol=1e-4, atol=1e-3 * scale)
    assert torch.allclose(sums[1].double().cpu(), (ref * ref).sum(0), rtol=1e-4, atol=1e-3 * scale * scale)
    # the two launches it replaces (bf16 round trip of x1 G1^T through memory in between)
    base = hipmod.gemm(0, x1.cuda(), G1.cuda(), B * H * W, C, C1)
    two, _ = hipmod.upsample_add_stats(base, [(t.cuda(), h, w) for t, (h, w) in zip(ts, sizes)], B, H, W, C)
    assert (two.double().cpu() - out.double().cpu()).abs().max().item() <= 2 ** -6 * scale
    # deterministic
    out2, sums2 = hipmod.fuse_map_248(x1.cuda(), G1.cuda(), *[t.cuda() for t in ts], B, H, W)
    assert torch.equal(out, out2) and torch.equal(sums, sums2)
    out3, none = hipmod.fuse_map_248(x1.cuda(), G1.cuda(), *[t.cuda() for t in ts], B, H, W, with_sums=False)
    assert none is None and torch.equal(out, out3)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('geom', [(2, 16, 24, 16), (1, 8, 8, 40), (1, 32, 16, 8), (2, 128, 128, 64), (2, 32, 40, 128), (1, 24, 24, 256),
                                  (2, 128, 128, 128), (1, 24, 64, 768), (1, 8, 8, 128), (2, 8, 16, 128), (1, 16, 16, 256)])
def test_bilinear_bwd_248_equals_three_transposed_resizes(hipmod, dtype, geom):
    """segf_bilinear_bwd_248 (one pass over the gradient, nested windows) against the three segf_bilinear_bwd launches it
    replaces and against autograd through F.interpolate on the CPU; includes the clamped borders (8-wide maps: every output
    pixel is a border pixel).  bf16 with C % 128 == 0 runs on the matrix pipe (fuse_map.hip: fuse_map_bwd_kernel, ATen's index
    clamping folded into the 1-D weight factors of the border blocks): 8 x 8 = one block that is first and last on both axes,
    (8, 16) / (16, 16) = border blocks only, (24, 24) = exactly one interior block."""
    B, H, W, C = geom
    g = torch.Generator().manual_seed(33)
    dy = torch.randn(B * H * W, C, generator=g)
    d = _dev(dy, dtype)
    outs = hipmod.bilinear_bwd_248(d, B, H, W, C)
    for r, got in zip((2, 4, 8), outs):
        sep = hipmod.bilinear_bwd(d, B, H // r, W // r, C, H, W, align_corners=False)
        _close(got, sep, dtype, fac=2)
        x = torch.zeros(B, C, H // r, W // r, requires_grad=True)
        up = F.interpolate(x, size=(H, W), mode='bilinear', align_corners=False)
        up.backward(_q(dy, dtype).reshape(B, H, W, C).permute(0, 3, 1, 2))
        ref = x.grad.permute(0, 2, 3, 1).reshape(-1, C)
        _close(got, ref, dtype, fac=4)


@pytest.mark.parametrize('cfg', [(2, 128, 128, 768, 150, 1, True, False), (2, 64, 128, 256, 19, 1, False, False),
                                 (1, 128, 128, 512, 21, 0, True, True), (3, 96, 64, 768, 171, 2, False, False),
                                 (5, 72, 72, 768, 150, 1, True, False)])
def test_bn_backward_with_classifier_dx_folded_in(hipmod, cfg):
    """segf_bn_cls_bwd (head_fused.hip: both BatchNorm-backward passes recompute da = dy W on the matrix pipe, da is never
    materialised) against fp32 autograd of  a = act(bn(x)) * drop;  y = a W^T  on the CPU, and against the two-launch path
    segf_gemm(layout 1) + segf_bn_bwd it replaces."""
    B, h, w, C, nc, act, with_drop, eval_mode = cfg
    hip = hipmod
    g = torch.Generator().manual_seed(41)
    M, K = B * h * w, (nc + 31) // 32 * 32
    x = torch.randn(M, C, generator=g) * 1.5 + 0.3
    dy = torch.zeros(M, K)
    dy[:, :nc] = torch.randn(M, nc, generator=g) * 1e-3
    W = torch.zeros(K, C)
    W[:nc] = torch.randn(nc, C, generator=g) / C ** 0.5
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    drop = ((torch.rand(B, C, generator=g) > 0.1).float() / 0.9) if with_drop else None
    xq, dyq, Wq = _q(x, torch.bfloat16), _q(dy, torch.bfloat16), _q(W, torch.bfloat16)
    # reference: fp32 autograd on the CPU
    xr = xq.clone().requires_grad_(True)
    gr, br = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    if eval_mode:
        mean, var = torch.randn(C, generator=g) * 0.1 + 0.3, 2.0 + torch.rand(C, generator=g)
        xh = (xr - mean) / torch.sqrt(var + 1e-5)
    else:
        mean, var = xq.mean(0), xq.var(0, unbiased=False)
        xh = (xr - xr.mean(0)) / torch.sqrt(xr.var(0, unbiased=False) + 1e-5)
    z = xh * gr + br
    a = z if act == 0 else (torch.relu(z) if act == 1 else torch.clamp(z, 0, 6))
    if drop is not None:
        a = a * drop.repeat_interleave(h * w, dim=0)
    (a @ Wq.t()).backward(dyq)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    args = (_dev(dyq, torch.bfloat16), _dev(Wq, torch.bfloat16), _dev(xq, torch.bfloat16), _dev(mean), _dev(rstd), _dev(gam), _dev(bet),
            act, _dev(drop) if drop is not None else None, h * w, eval_mode)
    assert hip.bn_cls_bwd_supported(torch.bfloat16, M, C, K, h * w)
    dx, dg, db = hip.bn_cls_bwd(*args)
    scale = xr.grad.abs().max().item()
    assert (dx.float().cpu() - xr.grad).abs().max().item() <= 2e-2 * scale                 # bf16 output rounding
    assert (dg.cpu() - gr.grad).abs().max().item() <= 2e-3 * gr.grad.abs().max().item() + 1e-6
    assert (db.cpu() - br.grad).abs().max().item() <= 2e-3 * br.grad.abs().max().item() + 1e-6
    # the two-launch path it replaces (da rounded to bf16 in between)
    da = hip.gemm(1, args[0], args[1], M, C, K)
    dx2, dg2, db2 = hip.bn_bwd(args[2], da, args[3], args[4], args[5], args[6], act, args[8], h * w, eval_mode)
    assert (dx.float() - dx2.float()).abs().max().item() <= 3e-2 * scale
    assert (dg - dg2).abs().max().item() <= 2e-2 * dg2.abs().max().item() + 1e-6
    # segf_bn_cls_bwd_dw: the same launches with the consumer's weight-gradient product riding on pass 2:
    # dG = [dx^T x1 | colsum(dx) | 0] from the bf16 dx tile on chip -- dx / dgamma / dbeta must not change by a bit, dG must equal
    # the fp32 product over the STORED dx (fp32 accumulation order aside), and two runs must agree bitwise
    C1 = 32
    if hip.bn_cls_bwd_dw_supported(torch.bfloat16, M, C, K, h * w, C1):
        x1 = (torch.randn(M, C1 + 8, generator=g) * 0.7).to(torch.bfloat16)
        x1d = x1.cuda()[:, :C1]                                                   # row stride 40: a column slice of a wider buffer
        dxw, dgw, dbw, dG = hip.bn_cls_bwd_dw(*args, x1d)
        assert torch.equal(dxw, dx) and torch.equal(dgw, dg) and torch.equal(dbw, db)
        assert dG.shape == (C, C1 + 8) and dG.dtype == torch.float32
        dxf = dx.float().cpu().double()
        want = dxf.t() @ x1[:, :C1].double()
        tol = 1e-5 * (dxf.abs().t() @ x1[:, :C1].double().abs()).max().item() + 1e-12
        assert (dG[:, :C1].cpu().double() - want).abs().max().item() <= tol
        assert (dG[:, C1].cpu().double() - dxf.sum(0)).abs().max().item() <= 1e-5 * dxf.abs().sum(0).max().item() + 1e-12
        assert float(dG[:, C1 + 1:].abs().max()) == 0.0
        dG2 = hip.bn_cls_bwd_dw(*args, x1d)[3]
        assert torch.equal(dG, dG2)
    # segf_bn_cls_bwd_full: the classifier's weight gradient dW = dy^T a rides on pass 1 (a = act(bn(x)) * drop rounded to bf16 on
    # chip): dx must not change by a bit (other workgroup chunking: dgamma / dbeta to fp32 round-off), dW against fp32 autograd
    # (Wq's gradient above) within the bf16 rounding of a, and two runs bitwise equal
    if act in (0, 1):
        x1f = None
        if hip.bn_cls_bwd_dw_supported(torch.bfloat16, M, C, K, h * w, C1):
            x1f = x1d
        dxf_, dgf, dbf, dGf, dwc = hip.bn_cls_bwd_full(*args, x1=x1f)
        assert torch.equal(dxf_, dx)
        assert (dgf - dg).abs().max().item() <= 1e-5 * dg.abs().max().item() + 1e-9
        assert (dbf - db).abs().max().item() <= 1e-5 * db.abs().max().item() + 1e-9
        if x1f is not None:
            assert (dGf - dG).abs().max().item() <= 1e-5 * dG.abs().max().item() + 1e-9
        assert dwc.shape == (K, C) and dwc.dtype == torch.float32
        # reference: the layout-2 operand-prologue GEMM it replaces is dy^T bf16(a); here against exact fp32 a
        a32 = a.detach()
        want_w = dyq.double().t() @ a32.double()
        bound = 2 ** -8 * (dyq.double().abs().t() @ a32.double().abs()).max().item()
        assert (dwc.cpu().double() - want_w).abs().max().item() <= bound, ((dwc.cpu().double() - want_w).abs().max().item(), bound)
        assert float(dwc[nc:].abs().max()) == 0.0 if nc < K else True
        dwc2 = hip.bn_cls_bwd_full(*args, x1=x1f)[4]
        assert torch.equal(dwc, dwc2)
        # the finalize of the [K x C] partial slabs takes the wide form (four adjacent outputs per thread) when the output is large: the
        # same sums in the same order as the 16-outputs-per-block form -- bitwise
        os.environ['SEGFAC_NO_WIDE_FINALIZE'] = '1'
        try:
            narrow = hip.bn_cls_bwd_full(*args, x1=x1f)
        finally:
            del os.environ['SEGFAC_NO_WIDE_FINALIZE']
        assert torch.equal(dwc, narrow[4]) and torch.equal(dgf, narrow[1]) and (x1f is None or torch.equal(dGf, narrow[3]))


@pytest.mark.parametrize('geom', [(2, 128, 128, 512, 512), (1, 160, 160, 272, 384), (4, 96, 128, 144, 768)])
def test_fp8_conv3x3_forward_and_data_gradient(hipmod, geom):
    """segf_conv3x3_fp8 (BASELINE cfg5's UPerHead / PPM 3x3 convolutions on fp8 operands: the 256-tile implicit GEMM of gemm.hip
    with 2-byte-unit addressing, two v_mfma_f32_16x16x32_fp8 per 16-byte fragment) and segf_quant_tensor_fp8, against
    torch.conv2d on the DEQUANTISED operands in float64 -- fp8 x fp8 products are exact in fp32, so only accumulation order and the
    bf16 output rounding remain -- for the forward (e4m3 x e4m3) and the data gradient (e5m2 x e4m3, transposed weights)."""
    hip = hipmod
    B, H, W, Cin, Cout = geom
    g = torch.Generator().manual_seed(7)
    P = B * H * W
    x = (torch.randn(P, Cin, generator=g) * 2).to(torch.bfloat16)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5)
    dy = (torch.randn(P, Cout, generator=g) * 1e-3).to(torch.bfloat16)
    assert hip.conv3x3_fp8_supported(0, B, H, W, Cin, Cout) and hip.conv3x3_fp8_supported(1, B, H, W, Cin, Cout)
    # quantisation: tensor scale = amax / 448 (e4m3) or / 57344 (e5m2); bytes decode with torch's own fp8 dtypes
    xq, sx = hip.quant_tensor_fp8(x.cuda())
    assert abs(sx.item() - x.float().abs().max().item() / 448.0) <= 1e-6 * sx.item()
    xd = xq.cpu().view(torch.float8_e4m3fn).double() * sx.item()
    assert (xd - x.double()).abs().max().item() <= 2 ** -3 * x.float().abs().max().item() / 1.75        # half an e4m3 step at the top binade
    gq, sg = hip.quant_tensor_fp8(dy.cuda(), e5m2=True)
    gd = gq.cpu().view(torch.float8_e5m2).double() * sg.item()
    assert (gd - dy.double()).abs().max().item() <= 2 ** -2 * dy.float().abs().max().item() / 1.5
    # forward
    wm = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(torch.bfloat16)                    # [O][(ky,kx)][ci]
    wq, sw = hip.quant_rows_fp8(wm.cuda())
    wd = (wq.cpu().view(torch.float8_e4m3fn).double() * sw.cpu().double()[:, None]).view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    y = hip.conv3x3_fp8(0, xq, sx, wq, sw, B, H, W, Cin, Cout)
    ref = F.conv2d(xd.view(B, H, W, Cin).permute(0, 3, 1, 2), wd, padding=1).permute(0, 2, 3, 1).reshape(P, Cout)
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 2 ** -8 * ref.abs().max().item() + 1e-9, (err, ref.abs().max().item())
    # ... and the fp8 result is a sensible approximation of the bf16 convolution it stands in for
    yb = hip.conv3x3(0, x.cuda(), wm.cuda(), B, H, W, Cin, Cout)
    assert (y.float() - yb.float()).abs().max().item() <= 0.12 * yb.float().abs().max().item()
    # data gradient: dx = conv_transpose(dy, w): gq e5m2 x transposed weights e4m3 (rows = input channels)
    wt = w.permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous().to(torch.bfloat16)                    # [ci][(ky,kx)][co]
    wtq, swt = hip.quant_rows_fp8(wt.cuda())
    wtd = (wtq.cpu().view(torch.float8_e4m3fn).double() * swt.cpu().double()[:, None]).view(Cin, 3, 3, Cout)   # [ci][ky][kx][co]
    dx = hip.conv3x3_fp8(1, gq, sg, wtq, swt, B, H, W, Cin, Cout)
    wfull = wtd.permute(3, 0, 1, 2)                                                                           # [co][ci][ky][kx]
    refdx = F.conv_transpose2d(gd.view(B, H, W, Cout).permute(0, 3, 1, 2), wfull, padding=1).permute(0, 2, 3, 1).reshape(P, Cin)
    err = (dx.double().cpu() - refdx).abs().max().item()
    assert err <= 2 ** -8 * refdx.abs().max().item() + 1e-12, (err, refdx.abs().max().item())


def test_fp8_conv3x3_weight_gradient(hipmod):
    """segf_conv3x3_fp8_wgrad: the conv weight gradient on the quantised tensors the forward / data-gradient calls already made (x e4m3,
    dy e5m2, one scale per tensor) -- reduction-major fp8 operands through ds_read_b64_tr_b8, block-scaled K = 128 MFMA, split-K --
    against autograd of F.conv2d on the DEQUANTISED tensors (fp8 x fp8 products are exact in fp32: only the summation order differs)."""
    hip = hipmod
    B, H, W, Cin, Cout = 8, 96, 128, 256, 256
    P = B * H * W
    g = torch.Generator().manual_seed(17)
    x = (torch.randn(P, Cin, generator=g) * 2).to(torch.bfloat16)
    dy = (torch.randn(P, Cout, generator=g) * 1e-3).to(torch.bfloat16)
    assert hip.conv3x3_fp8_wgrad_supported(B, H, W, Cin, Cout)
    xq, sx = hip.quant_tensor_fp8(x.cuda())
    gq, sg = hip.quant_tensor_fp8(dy.cuda(), e5m2=True)
    dw = hip.conv3x3_fp8_wgrad(xq, sx, gq, sg, B, H, W, Cin, Cout)
    xd = (xq.cpu().view(torch.float8_e4m3fn).float() * sx.item()).view(B, H, W, Cin).permute(0, 3, 1, 2)
    gd = (gq.cpu().view(torch.float8_e5m2).float() * sg.item()).view(B, H, W, Cout).permute(0, 3, 1, 2)
    w0 = torch.zeros(Cout, Cin, 3, 3, requires_grad=True)
    F.conv2d(xd, w0, padding=1).backward(gd)
    ref = w0.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)                           # [co][(ky,kx)][ci]
    err = (dw.cpu() - ref).abs().max().item()
    assert err <= 2e-4 * ref.abs().max().item(), (err, ref.abs().max().item())
    assert torch.equal(dw, hip.conv3x3_fp8_wgrad(xq, sx, gq, sg, B, H, W, Cin, Cout))       # deterministic
    # and it approximates the bf16 weight gradient it stands in for
    dwb = hip.conv3x3(2, x.cuda(), dy.cuda(), B, H, W, Cin, Cout, split_k=hip.pick_splitk(Cout, 9 * Cin, P))
    assert (dw - dwb).abs().max().item() <= 0.1 * dwb.abs().max().item()


@pytest.mark.parametrize('shape', [(8192, 1024, 512), (12800, 3072, 768), (12800, 768, 3072)])
def test_fp8_linear_tensor_scaled_products(hipmod, shape):
    """segf_linear_fp8 / segf_linear_fp8_wgrad (the ConvNeXt block MLPs with set_fp8, convnextv2.py:83-113): y = x W^T + b with the
    residual / per-sample scale epilogue, dx = dy W, dW = dy^T x on fp8 operands with ONE scale per activation / gradient tensor and one
    per weight row, against fp32 matmuls of the DEQUANTISED operands (fp8 x fp8 products are exact in fp32: only the summation order and
    the bf16 output rounding remain); deterministic; and close to the bf16 products they stand in for."""
    hip = hipmod
    M, N, K = shape
    g = torch.Generator().manual_seed(67)
    x = (torch.randn(M, K, generator=g) * 1.5).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = (torch.randn(N, generator=g) * 0.1).cuda()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    rpg = M // 8
    rs = (torch.rand(8, generator=g) + 0.5).cuda()
    dy = (torch.randn(M, N, generator=g) * 1e-3).to(torch.bfloat16).cuda()
    assert hip.linear_fp8_supported(0, M, N, K)
    xq, sx = hip.quant_tensor_fp8(x)
    wq, sw = hip.quant_rows_fp8(w)
    xd = xq.view(torch.float8_e4m3fn).float() * sx
    wd = wq.view(torch.float8_e4m3fn).float() * sw[:, None]
    y = hip.linear_fp8(0, xq, sx, wq, sw, bias=bias, residual=res, rscale=rs, rows_per_group=rpg)
    lin = xd @ wd.t() + bias
    ref = res.float() + rs.repeat_interleave(rpg)[:, None] * lin
    assert (y.float() - ref).abs().max().item() <= 2 ** -8 * ref.abs().max().item() + 1e-6
    y0 = hip.linear_fp8(0, xq, sx, wq, sw)
    assert (y0.float() - xd @ wd.t()).abs().max().item() <= 2 ** -8 * lin.abs().max().item() + 1e-6
    assert torch.equal(y0, hip.linear_fp8(0, xq, sx, wq, sw))
    yb = hip.gemm(0, x, w.to(torch.bfloat16), M, N, K)
    assert (y0.float() - yb.float()).abs().max().item() <= 0.06 * yb.float().abs().max().item()
    # data gradient: dy e5m2 (one scale), W^T e4m3 per row
    gq, sg = hip.quant_tensor_fp8(dy, e5m2=True)
    gd = gq.view(torch.float8_e5m2).float() * sg
    if hip.linear_fp8_supported(1, M, K, N):
        wt = w.t().contiguous()
        wtq, swt = hip.quant_rows_fp8(wt)
        wtd = wtq.view(torch.float8_e4m3fn).float() * swt[:, None]
        dx = hip.linear_fp8(1, gq, sg, wtq, swt)
        refdx = gd @ wtd.t()
        assert (dx.float() - refdx).abs().max().item() <= 2 ** -8 * refdx.abs().max().item() + 1e-12
        dxb = dy.float() @ w
        assert (dx.float() - dxb).abs().max().item() <= 0.1 * dxb.abs().max().item()
    else:
        assert (M // 256) * (K // 256) < 128
    # weight gradient on the same quantised tensors
    if hip.linear_fp8_supported(2, M, N, K):
        dw = hip.linear_fp8_wgrad(gq, sg, xq, sx)
        refdw = gd.t() @ xd
        assert (dw - refdw).abs().max().item() <= 2e-4 * refdw.abs().max().item()
        assert torch.equal(dw, hip.linear_fp8_wgrad(gq, sg, xq, sx))
        dwb = dy.float().t() @ x.float()
        assert (dw - dwb).abs().max().item() <= 0.1 * dwb.abs().max().item()
    else:
        assert (N // 256) * (K // 256) * (M // 1024) < 128


@pytest.mark.parametrize('shape', [(300, 256, 128), (1000, 768, 3072), (4096, 1536, 384), (129, 40, 256)])
def test_fp8_quantise_and_gemm(hipmod, shape):
    """csrc/fp8.hip: row-wise e4m3fn quantisation (decode with torch.float8_e4m3fn) and the block-scaled fp8 MFMA product against
    an fp32 matmul of the DEQUANTISED operands (the fp8 products are exact in fp32, so only the bf16 output rounding remains)."""
    hip = hipmod
    M, N, K = shape
    g = torch.Generator().manual_seed(61)
    x = (torch.randn(M, K, generator=g) * torch.rand(M, 1, generator=g) * 3).to(torch.bfloat16)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g) * 0.1
    xq, sx = hip.quant_rows_fp8(x.cuda())
    wq, sw = hip.quant_rows_fp8(w.cuda())
    xd = xq.view(torch.float8_e4m3fn).float() * sx[:, None]
    wd = wq.view(torch.float8_e4m3fn).float() * sw[:, None]
    assert torch.isfinite(xd).all() and torch.isfinite(wd).all()
    assert torch.allclose(sx.cpu(), x.float().abs().amax(1) / 448, rtol=1e-6)
    assert (xd.cpu() - x.float()).abs().max() <= 0.0625 * x.float().abs().amax() + 1e-6     # e4m3: 3 mantissa bits, RNE
    assert ((xd.cpu() - x.float()).abs() <= 0.0626 * x.float().abs().clamp_min(sx.cpu()[:, None] * 2 ** -6) + 1e-12).all()
    ref = xd @ wd.t() + bias.cuda()
    out = hip.gemm_fp8(xq, sx, wq, sw, bias=bias.cuda())
    assert (out.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).cuda()
    rs = torch.rand(4, generator=g).cuda()
    rpg = (M + 3) // 4
    out2 = hip.gemm_fp8(xq, sx, wq, sw, bias=bias.cuda(), residual=res, rscale=rs, rows_per_group=rpg)
    ref2 = res.float() + rs.repeat_interleave(rpg)[:M, None] * ref
    assert (out2.float() - ref2).abs().max().item() <= 2e-2 * ref2.abs().max().item()
    # against the unquantised product: the fp8 error itself (about 2^-4 per operand element, averaged over K)
    full = x.float().cuda() @ w.cuda().t() + bias.cuda()
    assert (out.float() - full).abs().max().item() <= 8e-2 * full.abs().max().item()


def test_direct_gradient_placement_refuses_a_second_delivery():
    """with direct placement a parameter's slot in the flat gradient buffer is an ASSIGNMENT target.  A parameter
    consumed by two Functions in one backward (tied weights, a module applied twice) would keep only the last contribution and
    release its data-parallel bucket early: the optimizer's delivery bookkeeping must refuse instead (plain autograd, i.e. the
    eager path, accumulates such gradients as torch does)."""
    from segmentation_factory_amd import functional as Fh
    from segmentation_factory_amd.optim import FusedAGCAdamW
    lin = torch.nn.Linear(32, 32).cuda()
    opt = FusedAGCAdamW([{'params': list(lin.parameters()), 'weight_decay': 0.0}], lr=1e-3)
    opt.enable_direct_grads()
    x = torch.randn(64, 32, device='cuda', requires_grad=True)
    opt.begin_backward()
    Fh.linear(x, lin.weight, lin.bias).sum().backward()              # one consumer: fine, and nothing lands in .grad
    assert lin.weight.grad is None and opt.flat_grads.abs().sum().item() > 0
    opt.begin_backward()
    y = Fh.linear(Fh.linear(x, lin.weight, lin.bias), lin.weight, lin.bias)
    with pytest.raises(RuntimeError, match='two gradients'):
        y.sum().backward()


def test_hist_accum_matches_torch_promotion():
    """segf_hist_accum = Metrics.update's `self.hist += bincount(...)` (util/metrics.py:27): float32 += int64, counts cleared."""
    from segmentation_factory_amd import hip
    g = torch.Generator().manual_seed(3)
    counts = torch.randint(0, 1 << 26, (19, 19), generator=g, dtype=torch.int64)      # beyond 2^24: the rounding of quirk Q5 shows
    hist = torch.randint(0, 1 << 25, (19, 19), generator=g).float()
    want = hist.clone()
    want += counts
    h, c = hist.cuda(), counts.cuda()
    hip.hist_accum_(h, c)
    assert torch.equal(h.cpu(), want) and c.abs().sum().item() == 0


@pytest.mark.parametrize('shape', [(3072, 768, 51200), (768, 3072, 51200), (1536, 6144, 12800), (2048, 512, 65536 - 64)])
def test_linear_weight_gradient_on_the_eight_phase_tile(hipmod, shape):
    """r05: nn.Linear weight gradients that are matrix-pipe work (both feature counts multiples of 256, >= 100 GFLOP, >= 256 FLOP per
    operand byte: ConvNeXtV2-L stage 3 / 4 at 640^2 batch 32, convnextv2.py:56-68 backward; MiT-B2 stage 4 at batch 32) take the
    eight-phase kernel (gemm8.hip, reduction-major x reduction-major, split-K) + a column-sum pass for the bias gradient, alone and as
    members of segf_gemm_dw_db_grouped (bitwise the same), instead of the grouped 128-tile kernel (policy gemm8_dw = 0): against
    float64 on the rounded operands and against the 128-tile result."""
    M, N, K = shape
    g = torch.Generator(device='cuda').manual_seed(61)
    dy = (torch.randn(K, M, device='cuda', generator=g) * 0.1).to(torch.bfloat16)
    x = torch.randn(K, N, device='cuda', generator=g).to(torch.bfloat16)
    sk = hipmod.pick_splitk(M, N, K)
    assert 1 <= sk <= 16
    with hipmod.trace() as tr:
        dw, db = hipmod.gemm_dw_db(dy, x, M, N, K, split_k=sk)
    assert any('gemm8_kernel' in k for k in tr.kernels), tr.kernels
    small = (256, 64, 16384)                                  # a groupable member beside it
    dy2 = (torch.randn(small[2], small[0], device='cuda', generator=g) * 0.1).to(torch.bfloat16)
    x2 = torch.randn(small[2], small[1], device='cuda', generator=g).to(torch.bfloat16)
    sk2 = hipmod.pick_splitk(*small)
    dw2_ref, db2_ref = hipmod.gemm_dw_db(dy2, x2, *small, split_k=sk2)
    items = [(dy2, x2, *small, sk2, torch.empty(small[0], small[1], device='cuda'), torch.empty(small[0], device='cuda')),
             (dy, x, M, N, K, sk, torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda'))]
    hipmod.gemm_dw_db_grouped(items, shared_split=True)
    torch.cuda.synchronize()
    assert torch.equal(items[1][6], dw) and torch.equal(items[1][7], db)
    assert torch.equal(items[0][6], dw2_ref) and torch.equal(items[0][7], db2_ref)
    with hipmod.policy_override(gemm8_dw=0), hipmod.trace() as tr0:
        sk0 = hipmod.pick_splitk(M, N, K)
        dw0, db0 = hipmod.gemm_dw_db(dy, x, M, N, K, split_k=sk0)
    assert not any('gemm8_kernel' in k for k in tr0.kernels), tr0.kernels
    ref = (dy.double().t() @ x.double()).cpu()
    rb = dy.double().sum(0).cpu()
    assert (dw.double().cpu() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert (db.double().cpu() - rb).abs().max().item() <= 1e-4 * max(1.0, rb.abs().max().item())
    assert (dw - dw0).abs().max().item() <= 1e-3 * dw0.abs().max().item()
    assert (db - db0).abs().max().item() <= 1e-4 * max(1.0, db0.abs().max().item())


def test_grouped_weight_gradients_equal_per_layer_launches():
    """segf_gemm_dw_db_grouped (the deferred weight gradients of the captured train step; mit.py:43-59,98-99 backward) must give, layer by
    layer, BITWISE what segf_gemm_dw_db gives: MiT stage-3 / 4 shapes at the reference's default batch 4 (groupable: 128-tile split-K
    kernel), a stage-1 shape that takes the streaming kernel and a 256-tile shape (both run ungrouped inside the same call), 14 items
    (more than one group of 12)."""
    from segmentation_factory_amd import hip
    g = torch.Generator(device='cuda').manual_seed(5)
    shapes = [(160, 160, 4096), (320, 160, 4096), (160, 160, 4096), (640, 160, 4096), (160, 640, 4096),
              (256, 256, 1024), (512, 256, 1024), (256, 256, 1024), (1024, 256, 1024), (256, 1024, 1024),
              (64, 64, 16384), (256, 64, 16384),
              (32, 128, 65536),                      # streaming kernel (small output over many tokens)
              (768, 768, 65536)]                     # 256-tile kernel
    items, want = [], []
    for (M, N, K) in shapes:
        dy = (torch.randn(K, M, device='cuda', generator=g) * 0.1).to(torch.bfloat16)
        x = torch.randn(K, N, device='cuda', generator=g).to(torch.bfloat16)
        sk = hip.pick_splitk(M, N, K)
        dw_ref, db_ref = hip.gemm_dw_db(dy, x, M, N, K, split_k=sk)
        want.append((dw_ref.clone(), db_ref.clone()))
        items.append((dy, x, M, N, K, sk, torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda')))
    hip.gemm_dw_db_grouped(items)
    torch.cuda.synchronize()
    for (dy, x, M, N, K, sk, dw, db), (dw_ref, db_ref) in zip(items, want):
        assert torch.equal(dw, dw_ref), (M, N, K, (dw - dw_ref).abs().max().item())
        assert torch.equal(db, db_ref), (M, N, K)
    # and against fp32 torch on one member, so that "equal" is not "equally wrong"
    dy, x, M, N, K = items[3][0], items[3][1], *shapes[3]
    ref = dy.float().t() @ x.float()
    assert (items[3][6] - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    # the split-K reduce reads four adjacent outputs per thread where the shape allows (r04): the same sums in the same order as the
    # one-output form -- bitwise, grouped and per layer
    os.environ['SEGFAC_NO_REDUCE4'] = '1'
    try:
        items1 = [(dy, x, M, N, K, sk, torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda')) for (dy, x, M, N, K, sk, _, _) in items]
        hip.gemm_dw_db_grouped(items1)
        single = [hip.gemm_dw_db(it[0], it[1], it[2], it[3], it[4], split_k=it[5]) for it in items]
    finally:
        del os.environ['SEGFAC_NO_REDUCE4']
    torch.cuda.synchronize()
    for a, b, c in zip(items, items1, single):
        assert torch.equal(a[6], b[6]) and torch.equal(a[7], b[7]) and torch.equal(a[6], c[0]) and torch.equal(a[7], c[1]), a[2:5]
    # shared_split (what the captured step passes): the members of a group may run with FEWER slices than they were given (one common K
    # range per slice, so that the group as a whole fills the chip): another summation order -- fp32 round-off against the per-layer
    # results -- and deterministic
    outs = []
    for _ in range(2):
        items2 = [(dy, x, M, N, K, sk, torch.empty(M, N, device='cuda'), torch.empty(M, device='cuda')) for (dy, x, M, N, K, sk, _, _) in items]
        hip.gemm_dw_db_grouped(items2, shared_split=True)
        torch.cuda.synchronize()
        outs.append(items2)
    for a, b, c in zip(items, outs[0], outs[1]):
        assert torch.equal(b[6], c[6]) and torch.equal(b[7], c[7])
        assert (a[6] - b[6]).abs().max().item() <= 2e-5 * a[6].abs().max().item() + 1e-6, a[2:5]
        assert (a[7] - b[7]).abs().max().item() <= 2e-5 * a[7].abs().max().item() + 1e-6, a[2:5]


def test_prep_grouped_equals_single_kernels():
    """segf_prep_grouped (the small layout jobs of a step in one launch: patch / spatial-reduction conv weight re-layouts, mit.py:105,47;
    the decode head's folded-weight packing, heads/segformer.py:42-56; gradient hand-over) must give, job by job, BITWISE what
    segf_cast2d / segf_permute021 / segf_zero give -- 30 jobs (more than one group of 24), every dtype pair, strided columns,
    a permute with padded rows and one into the rows of a wider matrix."""
    from segmentation_factory_amd import hip
    g = torch.Generator(device='cuda').manual_seed(11)

    def rnd(*shape, dtype=torch.float32):
        return torch.randn(*shape, generator=g, device='cuda').to(dtype)
    jobs, checks = [], []
    for k in range(6):
        for sdt, ddt in ((torch.float32, torch.bfloat16), (torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32)):
            rows, cols = 17 + 5 * k, 33 + 8 * k
            src = rnd(rows, cols + 8, dtype=sdt)[:, 3:3 + cols]
            dst = torch.full((rows, cols + 16), 7.0, dtype=ddt, device='cuda')
            ref = dst.clone()
            hip.cast2d(src, ref[:, 8:8 + cols])
            jobs.append(('cast', src, dst[:, 8:8 + cols]))
            checks.append((dst, ref))
    # a column into a column, a 1-D vector, zero fills of a strided block and of a vector
    src = rnd(40, 12)
    dst = torch.full((40, 9), 3.0, device='cuda')
    ref = dst.clone()
    hip.cast2d(src[:, 5:6], ref[:, 2:3])
    jobs.append(('cast', src[:, 5:6], dst[:, 2:3]))
    checks.append((dst, ref))
    v, dv = rnd(257), torch.empty(257, dtype=torch.bfloat16, device='cuda')
    jobs.append(('cast', v, dv))
    checks.append((dv, v.to(torch.bfloat16)))
    z = torch.full((64, 40), 5.0, dtype=torch.bfloat16, device='cuda')
    refz = z.clone()
    refz[:, 33:] = 0
    jobs.append(('zero', z[:, 33:]))
    checks.append((z, refz))
    z1 = torch.full((1000,), 2.0, device='cuda')
    jobs.append(('zero', z1))
    checks.append((z1, torch.zeros(1000, device='cuda')))
    # permutes: OIHW -> O (kh kw) I in bf16 (a MiT spatial-reduction conv at sr = 8), with the last dim padded, fp32 -> fp32, and into rows of
    # a 160-wide matrix (the stem)
    w = rnd(64, 32, 64)
    out = torch.empty((64, 64, 32), dtype=torch.bfloat16, device='cuda')
    jobs.append(('perm', w, out, 64, 32, 64, 32))
    checks.append((out, hip.permute021(w, 64, 32, 64, torch.bfloat16)))
    w2 = rnd(5, 19, 45)
    out2 = torch.empty((5, 45, 24), dtype=torch.float32, device='cuda')
    jobs.append(('perm', w2, out2, 5, 19, 45, 24))
    checks.append((out2, hip.permute021(w2, 5, 19, 45, torch.float32, ld_out=24)))
    w3 = rnd(32, 3, 49)
    out3 = torch.full((32, 160), 9.0, dtype=torch.bfloat16, device='cuda')
    ref3 = out3.clone()
    ref3[:, :147] = hip.permute021(w3, 32, 3, 49, torch.bfloat16).view(32, 147)
    jobs.append(('perm', w3, out3, 32, 3, 49, 3, 160))
    checks.append((out3, ref3))
    assert len(jobs) > 24
    hip.prep_grouped(jobs)
    torch.cuda.synchronize()
    for k, (got, want) in enumerate(checks):
        assert torch.equal(got, want), k


def test_derived_weights_registry_matches_direct_layouts():
    """functional.derived_scope: the re-laid-out weight copies of a step come from ONE launch at the start of the step and must equal what each
    layer used to build for itself (ConvPatchFn: mit.py:105,47; the stem's 160-column form; DWConv7Fn: convnext.py:29) -- outputs and
    gradients of a patch conv, an spatial-reduction conv and a depthwise 7 x 7 with and without the registry, second pass (registry replayed),
    after a weight update in between."""
    from segmentation_factory_amd import functional as Fh
    g = torch.Generator().manual_seed(2)
    B = 2
    img = torch.randn(B, 3, 128, 128, generator=g).cuda()
    tok = torch.randn(B * 32 * 32, 32, generator=g).cuda().to(torch.bfloat16)
    ws = [torch.randn(32, 3, 7, 7, generator=g).cuda().requires_grad_(), torch.randn(32, 32, 4, 4, generator=g).mul(0.05).cuda().requires_grad_(),
          torch.randn(32, 1, 7, 7, generator=g).mul(0.1).cuda().requires_grad_()]
    bs = [torch.randn(32, generator=g).cuda().requires_grad_() for _ in range(3)]

    def run():
        y1 = Fh.conv_patch(img, ws[0], bs[0], (B, 128, 128, 3, 7, 4, 3), image=True, dtype=torch.bfloat16)
        y2 = Fh.conv_patch(tok, ws[1], bs[1], (B, 32, 32, 32, 4, 4, 0))
        y3 = Fh.dwconv7x7(tok, ws[2], bs[2], B, 32, 32)
        for p in ws + bs:
            p.grad = None
        (y1.float().sum() + y2.float().square().sum() + y3.float().square().sum()).backward()
        return [y1, y2, y3] + [p.grad.clone() for p in ws + bs]
    reg = Fh.DerivedWeights()
    for step in range(3):
        want = run()
        with Fh.derived_scope(reg):
            got = run()
        assert len(reg.entries) == 3
        for a, b in zip(got, want):
            assert torch.equal(a, b), step
        with torch.no_grad():
            for p in ws:
                p.mul_(1.25)


def test_dwconv3x3_deferred_finalize_equals_direct():
    """segf_dwconv3x3_gelu_bwd with dw == NULL leaves its partial sums for segf_colreduce_finalize_grouped (scatter_c = C), which must
    write BITWISE the dw [C][9] | db [C] of the direct call (mit.py:62-71 backward) -- two layers of different geometry in one grouped
    launch, next to a plain (LayerNorm-type) member."""
    from segmentation_factory_amd import hip
    g = torch.Generator(device='cuda').manual_seed(4)
    items, checks = [], []
    for (B, H, W, Cc) in ((2, 32, 32, 128), (4, 16, 16, 1024)):
        x = torch.randn(B * H * W, Cc, generator=g, device='cuda').to(torch.bfloat16)
        dy = torch.randn(B * H * W, Cc, generator=g, device='cuda').to(torch.bfloat16)
        w9 = torch.randn(Cc, 9, generator=g, device='cuda') * 0.2
        b = torch.randn(Cc, generator=g, device='cuda') * 0.1
        dx0, dw0, db0 = hip.dwconv3x3_gelu_bwd(x, w9, b, dy, B, H, W, Cc, True)
        flat = torch.full((10 * Cc,), 7.0, device='cuda')
        dx1, item = hip.dwconv3x3_gelu_bwd(x, w9, b, dy, B, H, W, Cc, True, dw_out=flat[:9 * Cc].view(Cc, 9), db_out=flat[9 * Cc:], defer=True)
        items.append(item)
        checks.append((dx0, dx1, torch.cat([dw0.reshape(-1), db0]), flat))
    part = torch.randn(5, 96, generator=g, device='cuda')
    out = torch.empty(96, device='cuda')
    items.insert(1, (part, 5, 96, out))
    hip.colreduce_finalize_grouped(items)
    torch.cuda.synchronize()
    for dx0, dx1, want, got in checks:
        assert torch.equal(dx0, dx1) and torch.equal(want, got)
    assert torch.allclose(out, part.sum(0), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('dtype', DTYPES)
def test_layernorm_backward_scaled_second_output(dtype, monkeypatch):
    """segf_layernorm_bwd_scaled: dxs must be BITWISE segf_scale_rows(dx) (the DropPath backward of the residual branch that consumes dx,
    mit.py:143-146 / drop_path.py:18-25), dx / dgamma / dbeta unchanged; and through the autograd Functions: a MiT-style chain
    x1 = x0 + s1 * Linear(.), LayerNorm(x1) -> x2 = x1 + s2 * Linear(.) gives the same gradients with the fused output on and off."""
    from segmentation_factory_amd import hip, functional as Fh
    g = torch.Generator().manual_seed(9)
    B, N, Cc = 3, 200, 64
    x = _dev(torch.randn(B * N, Cc, generator=g), dtype)
    dy = _dev(torch.randn(B * N, Cc, generator=g), dtype)
    dres = _dev(torch.randn(B * N, Cc, generator=g), dtype)
    gamma = (torch.rand(Cc, generator=g) + 0.5).cuda()
    beta = torch.randn(Cc, generator=g).cuda()
    sc = torch.tensor([0.0, 1.25, 1.25]).cuda()
    _, mean, rstd = hip.layernorm_fwd(x, gamma, beta, 1e-5)
    dx0, dg0, db0 = hip.layernorm_bwd(x, dy, gamma, mean, rstd, dres=dres)
    dx1, dg1, db1 = hip.layernorm_bwd(x, dy, gamma, mean, rstd, dres=dres, rscale=sc, rows_per_group=N)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert torch.equal(dx1.scaled, hip.scale_rows(dx0, sc, N))

    w1 = (torch.randn(Cc, Cc, generator=g) * 0.1).cuda().requires_grad_()
    w2 = (torch.randn(Cc, Cc, generator=g) * 0.1).cuda().requires_grad_()
    b1 = torch.randn(Cc, generator=g).cuda().requires_grad_()
    b2 = torch.randn(Cc, generator=g).cuda().requires_grad_()
    ga = gamma.clone().requires_grad_()
    be = beta.clone().requires_grad_()
    s1, s2 = sc, torch.tensor([1.25, 0.0, 1.25]).cuda()
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv('SEGFAC_NO_SCALED_LN_BWD', '1')
        x0 = x.clone().requires_grad_()
        for p in (w1, w2, b1, b2, ga, be):
            p.grad = None
        x1 = Fh.linear(x0, w1, b1, residual=x0, rscale=s1, rows_per_group=N)
        xr, h = Fh.layer_norm_res(x1, ga, be, 1e-5)
        x2 = Fh.linear(h, w2, b2, residual=xr, rscale=s2, rows_per_group=N)
        y = Fh.layer_norm(x2, ga, be, 1e-5)
        calls = []
        real_scale_rows = hip.scale_rows
        monkeypatch.setattr(hip, 'scale_rows', lambda *a, **k: (calls.append(1), real_scale_rows(*a, **k))[1])
        (y.float() * dy.float()).sum().backward()
        monkeypatch.setattr(hip, 'scale_rows', real_scale_rows)
        outs.append([x0.grad.clone()] + [p.grad.clone() for p in (w1, w2, b1, b2, ga, be)])
        # both scaled copies were picked up (single-consumer hand-over, checked by address AND version): no scale_rows launch is left
        assert off or len(Fh._SCALED_DY) == 0
        assert len(calls) == (2 if off else 0), (off, len(calls))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('geom', [(2, 32, 32, 64, 4), (1, 24, 64, 32, 8), (3, 16, 16, 160, 2)])
def test_layernorm_patch_major_second_output(dtype, geom):
    """segf_layernorm_fwd_patch / segf_layernorm_bwd_patch: the norm in front of MiT's spatial-reduction convolution (mit.py:143, 47) writes
    its output a second time as the im2col matrix of the k = s = sr convolution (BITWISE segf_im2col of the first output) and reads that
    output's gradient in the same order (BITWISE the backward with dy2 = segf_col2im of it)."""
    from segmentation_factory_amd import hip
    B, H, W, Cc, sr = geom
    g = torch.Generator().manual_seed(13)
    x = _dev(torch.randn(B * H * W, Cc, generator=g), dtype)
    gamma = (torch.rand(Cc, generator=g) + 0.5).cuda()
    beta = torch.randn(Cc, generator=g).cuda()
    lw, ls = W.bit_length() - 1, sr.bit_length() - 1
    y0, mean0, rstd0 = hip.layernorm_fwd(x, gamma, beta, 1e-5)
    y, mean, rstd, col = hip.layernorm_fwd(x, gamma, beta, 1e-5, patch=(lw, ls))
    Ho, Wo = H // sr, W // sr
    assert torch.equal(y, y0) and torch.equal(mean, mean0) and torch.equal(rstd, rstd0)
    assert col.shape == (B * Ho * Wo, sr * sr * Cc)
    assert torch.equal(col, hip.im2col(y0, dtype, False, B, H, W, Cc, sr, sr, sr, 0, Ho, Wo, sr * sr * Cc))
    dy = _dev(torch.randn(B * H * W, Cc, generator=g), dtype)
    dcol = _dev(torch.randn(B * Ho * Wo, sr * sr * Cc, generator=g), dtype)
    dres = _dev(torch.randn(B * H * W, Cc, generator=g), dtype)
    want = hip.layernorm_bwd(x, dy, gamma, mean, rstd, dy2=hip.col2im(dcol, B, H, W, Cc, sr, sr, sr, 0, Ho, Wo), dres=dres)
    got = hip.layernorm_bwd(x, dy, gamma, mean, rstd, dy2=dcol, dres=dres, dy2_patch=(lw, ls))
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_mit_block_with_patch_major_norm_matches_im2col_path(monkeypatch):
    """backbones.Block with spatial reduction: the path without im2col / col2im passes (layer_norm_res_patch + conv_from_col) against the
    im2col path (SEGFAC_NO_LN_PATCH=1): forward bitwise; gradients equal up to ONE bf16 rounding (the two consumers' gradients of the
    norm now meet in fp32 inside the LayerNorm backward instead of in the epilogue of q's data-gradient product)."""
    from segmentation_factory_amd.backbones import Block
    torch.manual_seed(5)
    B, H, W, dim = 2, 32, 32, 64
    blk = Block(dim, 2, sr_ratio=4, dpr=0.0).cuda()
    x0 = torch.randn(B * H * W, dim).cuda().to(torch.bfloat16)
    dy = torch.randn(B * H * W, dim).cuda().to(torch.bfloat16)
    outs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv('SEGFAC_NO_LN_PATCH', '1')
        x = x0.clone().requires_grad_()
        for p in blk.parameters():
            p.grad = None
        y = blk.tokens(x, B, H, W, (None, None))
        y.backward(dy)
        outs.append((y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    sc = outs[1][1].float().abs().max().item()
    assert (outs[0][1].float() - outs[1][1].float()).abs().max().item() <= 2.0 ** -7 * sc
    for k, gnew in outs[0][2].items():
        gold = outs[1][2][k]
        assert (gnew - gold).abs().max().item() <= 2e-2 * gold.abs().max().item() + 1e-6, k


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('geom', [(4, 16, 16, 1024), (2, 32, 32, 640), (3, 8, 8, 64), (2, 24, 20, 40), (1, 16, 16, 32)])
def test_dwconv3x3_backward_one_launch_form_of_small_maps(dtype, geom, monkeypatch):
    """dwconv3x3_bwd_small_kernel (MiT stages 3 / 4 at 512^2, mit.py:62-71 backward: the map of one image in LDS, du / dx / dw / db in one
    launch) against the three-pass walk form of the same library (SEGFAC_DW_NO_SMALL=1): dx BITWISE (same tap order, same GELU helper, du
    rounded to the storage type in between), dw / db to fp32 round-off (another summation order); the deferred-finalize form likewise."""
    from segmentation_factory_amd import hip
    B, H, W, Cc = geom
    if dtype == torch.float32 and H * W > 512:
        pytest.skip('fp32: maps up to 512 pixels take the one-launch form')
    g = torch.Generator().manual_seed(23)
    x = _dev(torch.randn(B * H * W, Cc, generator=g), dtype)
    dy = _dev(torch.randn(B * H * W, Cc, generator=g), dtype)
    w9 = (torch.randn(Cc, 9, generator=g) * 0.3).cuda()
    b = (torch.randn(Cc, generator=g) * 0.2).cuda()
    monkeypatch.setenv('SEGFAC_DW_SMALL_ALWAYS', '1')          # (the dispatch takes this form up to one round of workgroups only)
    assert hip.lib().segf_dwconv3x3_bwd_blocks(hip.dt_of(x), B, H, W, Cc) == B
    for gelu in (True, False):
        dx1, dw1, db1 = hip.dwconv3x3_gelu_bwd(x, w9, b, dy, B, H, W, Cc, gelu)
        flat = torch.empty(10 * Cc, device='cuda')
        dx2, item = hip.dwconv3x3_gelu_bwd(x, w9, b, dy, B, H, W, Cc, gelu, dw_out=flat[:9 * Cc].view(Cc, 9), db_out=flat[9 * Cc:], defer=True)
        hip.colreduce_finalize_grouped([item])
        monkeypatch.setenv('SEGFAC_DW_NO_SMALL', '1')
        dx0, dw0, db0 = hip.dwconv3x3_gelu_bwd(x, w9, b, dy, B, H, W, Cc, gelu)
        monkeypatch.delenv('SEGFAC_DW_NO_SMALL')
        torch.cuda.synchronize()
        assert torch.equal(dx1, dx0) and torch.equal(dx2, dx0)
        assert torch.equal(flat[:9 * Cc].view(Cc, 9), dw1) and torch.equal(flat[9 * Cc:], db1)
        sc = dw0.abs().max().item()
        assert (dw1 - dw0).abs().max().item() <= 2e-5 * sc + 1e-6 and (db1 - db0).abs().max().item() <= 2e-5 * db0.abs().max().item() + 1e-6


@pytest.mark.parametrize('geom', [(32, 16, 16, 3840, 768), (8, 40, 40, 768, 768), (4, 32, 32, 512, 256), (8, 20, 20, 3840, 768), (6, 20, 20, 1024, 512)])
def test_conv3x3_split_k_form_of_few_tile_outputs(geom, monkeypatch):
    """segf_conv3x3 modes 0 / 1 over K slices (segf_conv3x3_fwd_splitk: UPerHead's PPM bottleneck, ppm.py:19, and the small FPN levels have too
    few 256 x 256 output tiles to fill the chip; pixel counts that are not multiples of 256 end in a ragged row tile): against the unsplit form of the same library -- the fp32 sums differ in order only, so the
    bf16 outputs agree to one rounding -- and against fp64 torch.conv2d on a sample."""
    from segmentation_factory_amd import hip
    B, H, W, Cin, Cout = geom
    P = B * H * W
    assert hip.lib().segf_conv3x3_fwd_splitk(0, B, H, W, Cin, Cout) > 1
    g = torch.Generator(device='cuda').manual_seed(31)
    x = torch.randn(P, Cin, device='cuda', generator=g).to(torch.bfloat16)
    dy = torch.randn(P, Cout, device='cuda', generator=g).to(torch.bfloat16)
    wm = (torch.randn(Cout, 9 * Cin, device='cuda', generator=g) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    wt = (torch.randn(Cin, 9 * Cout, device='cuda', generator=g) / (9 * Cout) ** 0.5).to(torch.bfloat16)
    y1, d1 = hip.conv3x3(0, x, wm, B, H, W, Cin, Cout), hip.conv3x3(1, dy, wt, B, H, W, Cin, Cout)
    monkeypatch.setenv('SEGFAC_CONV_NO_FWD_SPLIT', '1')
    y0, d0 = hip.conv3x3(0, x, wm, B, H, W, Cin, Cout), hip.conv3x3(1, dy, wt, B, H, W, Cin, Cout)
    for a, b in ((y1, y0), (d1, d0)):
        sc = b.float().abs().max().item()
        assert (a.float() - b.float()).abs().max().item() <= 2.0 ** -7 * sc
        assert (a != b).float().mean().item() < 0.2            # most outputs round to the same bf16
    # a sample of output pixels of the forward against fp64
    wd = wm.double().cpu().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.conv2d(x.double().cpu().view(B, H, W, Cin)[:1].permute(0, 3, 1, 2), wd, padding=1).permute(0, 2, 3, 1).reshape(H * W, Cout)
    assert (y1[:H * W].double().cpu() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 1e-6


@pytest.mark.parametrize('geom', [(3, 3, 5, 6), (10, 12, 9, 12), (5, 6, 10, 12), (4, 7, 12, 21), (7, 5, 16, 9), (1, 1, 3, 4), (6, 9, 4, 5)])
def test_nearest_up_any_size_pair_matches_aten(geom):
    """segf_nearest_up against torch's CPU F.interpolate(mode='nearest', size=...) (heads/fpn.py:30-31,35), forward (+ fused `out +
    lateral`) and backward: integer factors, and the size pairs inputs that are not multiples of 32 produce in FPNHead (3 x 3 ->
    5 x 6, 10 x 12 -> 9 x 12: ATen's floorf(dst * in / out) source index, including a SHRINKING step)."""
    import torch.nn.functional as F
    from segmentation_factory_amd import hip
    h, w, H, W = geom
    B, Cc = 2, 16
    g = torch.Generator().manual_seed(h * 100 + H)
    x = torch.randn(B, Cc, h, w, generator=g).requires_grad_()
    base = torch.randn(B, Cc, H, W, generator=g)
    dy = torch.randn(B, Cc, H, W, generator=g)
    ref = F.interpolate(x, size=(H, W), mode='nearest') + base
    ref.backward(dy)
    tok = lambda t: t.detach().permute(0, 2, 3, 1).reshape(-1, Cc).contiguous().cuda()
    got = hip.nearest_up(tok(x), B, h, w, Cc, H, W, base=tok(base))
    assert torch.equal(got.cpu(), tok(ref).cpu())
    dx = hip.nearest_up(tok(dy), B, h, w, Cc, H, W, bwd=True)
    assert torch.allclose(dx.cpu(), tok(x.grad).cpu(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('shape', [(25000, 19, 64), (25000, 40, 70), (24700, 70, 33), (25000, 100, 48), (25000, 150, 768), (25000, 160, 96),
                                   (4096, 3000, 130), (300, 150, 147), (257, 32, 32)])
def test_gemm_f32_matrix_pipe_forms(hipmod, shape):
    """fp32 storage (exact-parity mode; evaluate runs in it like the reference, engine.py:86-88): the products run on v_mfma_f32_32x32x2_f32
    -- exact f32, a k-ordered fmaf chain -- in three tile forms: one column tile of 128 x (32 .. 160) for narrow outputs over many rows
    (19 / 150 classes), 128 x 128 for large outputs, 64 x 64 otherwise.  All three layouts (+ bias, residual with row scales, split-K)
    against float64, at fp32 round-off; and against the vector FMA kernel (gemm_f32_no_mfma), which sums in another order."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    dy = torch.randn(M, N, generator=g)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    rs = torch.rand(M // 100 + 1, generator=g)
    tol = 2e-5 * (K ** 0.5)
    y = hipmod.gemm(0, _dev(x), _dev(w), M, N, K, bias=_dev(bias), residual=_dev(res), rscale=_dev(rs), rows_per_group=100)
    want = res.double() + rs.double().repeat_interleave(100)[:M, None] * (x.double() @ w.double().t() + bias.double())
    assert (y.cpu().double() - want).abs().max().item() <= tol
    with hipmod.policy_override(gemm_f32_no_mfma=1):
        y0 = hipmod.gemm(0, _dev(x), _dev(w), M, N, K, bias=_dev(bias), residual=_dev(res), rscale=_dev(rs), rows_per_group=100)
    assert (y - y0).abs().max().item() <= tol
    dx = hipmod.gemm(1, _dev(dy), _dev(w), M, K, N)
    assert (dx.cpu().double() - dy.double() @ w.double()).abs().max().item() <= 2e-5 * (N ** 0.5) * 4
    for sk in (1, 3):
        dw = hipmod.gemm(2, _dev(dy), _dev(x), N, K, M, out_dtype=torch.float32, split_k=sk)
        ref = dy.double().t() @ x.double()
        # (a chain of M fp32 additions: ~ sqrt(M) eps |sum| of accumulated round-off, x4 margin)
        assert (dw.cpu().double() - ref).abs().max().item() <= 4 * (M ** 0.5) * 6e-8 * ref.abs().max().item()
