"""ctypes binding of libsegfac_hip.so (the C ABI declared in include/segfac.h).

Only raw device pointers, sizes and the current HIP stream cross this boundary; torch is used
for device memory (caching allocator) and stream handles, never for the arithmetic.  There is no
CPU fallback: if the shared library is missing, or a kernel returns non-zero, a RuntimeError is
raised (the reference's own error convention is Python exceptions).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEGFAC_HIP_LIB: another build of the same library (same-box A/B of kernel changes, tools/ab_bench.sh); default in-tree
LIB_PATH = os.environ.get('SEGFAC_HIP_LIB') or os.path.join(_HERE, 'libsegfac_hip.so')
F32, BF16 = 0, 1
ERR_SHAPE = -1          # SEGF_ERR_SHAPE
_lib = None

_i, _l, _f, _p = C.c_int, C.c_int64, C.c_float, C.c_void_p
_PROTOS = {
    'segf_cast': (_i, [_p, _i, _p, _i, _l, _p]),
    'segf_cast2d': (_i, [_p, _i, _l, _p, _i, _l, _l, _l, _p]),
    'segf_permute021': (_i, [_p, _i, _p, _i, _l, _l, _l, _l, _p]),
    'segf_prep_grouped': (_i, [_i, _p, _p]),
    'segf_scale_rows': (_i, [_i, _p, _l, _p, _l, _p, _l, _l, _l, _p]),
    'segf_add': (_i, [_i, _p, _l, _p, _l, _p, _l, _l, _l, _p]),
    'segf_colsum_ws': (_l, [_l, _l]),
    'segf_colsum': (_i, [_i, _p, _l, _l, _l, _p, _p, _p]),
    'segf_gemm': (_i, [_i, _i, _l, _l, _l, _p, _l, _p, _l, _p, _i, _l, _p, _p, _l, _p, _l, _i, _p, _p]),
    'segf_gemm_pick_splitk': (_i, [_l, _l, _l]),
    'segf_conv3x3_pick_splitk': (_i, [_i, _i, _l]),
    'segf_conv3x3_fwd_splitk': (_i, [_i, _i, _i, _i, _i, _i]),
    'segf_argmax_rows': (_i, [_i, _l, _i, _p, _l, _p, _p]),
    'segf_gemm_dw_db_ws': (_l, [_l, _l, _l, _i]),
    'segf_gemm_pro_supported': (_i, [_i, _i, _l, _l, _l, _l]),
    'segf_gemm_pro': (_i, [_i, _i, _l, _l, _l, _p, _l, _p, _l, _p, _i, _l, _p, _i, _p, _p, _p, _l, _i, _p]),
    'segf_bn_affine_table': (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _p, _p]),
    'segf_gemm_dw_db': (_i, [_i, _l, _l, _l, _p, _l, _p, _l, _p, _i, _l, _i, _p, _p, _p]),
    'segf_gemm_dw_db_grouped': (_i, [_i, _i, _p, _p]),
    'segf_layernorm_bwd_blocks': (_i, [_l, _i]),
    'segf_colreduce_finalize_grouped': (_i, [_i, _p, _p]),
    'segf_layernorm_fwd': (_i, [_i, _l, _i, _p, _p, _p, _f, _p, _p, _p, _p]),
    'segf_layernorm_fwd_patch': (_i, [_i, _l, _i, _p, _p, _p, _f, _p, _p, _p, _p, _i, _i, _p]),
    'segf_layernorm_bwd_patch': (_i, [_i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p, _i, _i, _p]),
    'segf_layernorm_bwd_ws': (_l, [_l, _i]),
    'segf_layernorm_bwd': (_i, [_i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'segf_bn_ws': (_l, [_l, _i]),
    'segf_bn_stats': (_i, [_i, _l, _i, _p, _p, _p, _p, _p, _f, _f, _p, _p]),
    'segf_bn_apply': (_i, [_i, _l, _i, _p, _p, _p, _p, _p, _i, _p, _l, _p, _p]),
    'segf_bn_bwd': (_i, [_i, _l, _i, _p, _p, _p, _p, _p, _p, _i, _p, _l, _i, _p, _p, _p, _p, _p]),
    'segf_bn_cls_bwd_supported': (_i, [_i, _l, _i, _i, _l]),
    'segf_bn_cls_bwd_ws': (_l, [_l, _i, _l]),
    'segf_bn_cls_bwd': (_i, [_i, _l, _i, _i, _p, _l, _p, _l, _p, _p, _p, _p, _p, _i, _p, _l, _i, _p, _p, _p, _p, _p]),
    'segf_bn_cls_bwd_dw_supported': (_i, [_i, _l, _i, _i, _l, _i]),
    'segf_bn_cls_bwd_dw_ws': (_l, [_l, _i, _l]),
    'segf_bn_cls_bwd_dw': (_i, [_i, _l, _i, _i, _p, _l, _p, _l, _p, _p, _p, _p, _p, _i, _p, _l, _i, _p, _p, _p, _p, _p, _l, _i, _p, _p]),
    'segf_bn_cls_bwd_full_ws': (_l, [_l, _i, _i, _l]),
    'segf_bn_cls_bwd_full': (_i, [_i, _l, _i, _i, _p, _l, _p, _l, _p, _p, _p, _p, _p, _i, _p, _l, _i, _p, _p, _p, _p, _p, _l, _i, _p, _p, _p]),
    'segf_grn_ws': (_l, [_i, _l, _i, _i]),
    'segf_grn_fwd': (_i, [_i, _i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    'segf_grn_bwd': (_i, [_i, _i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    'segf_attention_fwd': (_i, [_i, _i, _i, _i, _i, _i, _p, _l, _p, _l, _p, _l, _f, _p, _l, _p, _p]),
    'segf_attention_bwd_ws': (_l, [_i, _i, _i, _i, _i]),
    'segf_attention_bwd': (_i, [_i, _i, _i, _i, _i, _i, _p, _l, _p, _l, _p, _l, _f, _p, _l, _p, _l, _p,
                                _p, _l, _p, _l, _p, _l, _p, _p]),
    'segf_dwconv3x3_gelu_fwd': (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    'segf_dwconv3x3_bwd_ws': (_l, [_i, _i, _i, _i]),
    'segf_dwconv3x3_bwd_blocks': (_i, [_i, _i, _i, _i, _i]),
    'segf_dwconv3x3_gelu_bwd': (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p]),
    'segf_dwconv7x7_fwd': (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    'segf_dwconv7x7_bwd_ws': (_l, [_i, _i, _i, _i]),
    'segf_dwconv7x7_bwd': (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    'segf_conv3x3': (_i, [_i, _i, _i, _i, _i, _i, _p, _l, _p, _l, _p, _i, _l, _p, _i, _p, _p]),
    'segf_gelu': (_i, [_i, _i, _p, _p, _p, _l, _p]),
    'segf_rowdot': (_i, [_p, _l, _p, _l, _p, _p, _p, _l, _l, _p]),
    'segf_adaptive_avgpool': (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    'segf_im2col': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _l, _p]),
    'segf_col2im': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _l, _p, _p]),
    'segf_bilinear_fwd': (_i, [_i, _i, _i, _i, _i, _p, _l, _i, _i, _p, _l, _i, _p]),
    'segf_bilinear_bwd': (_i, [_i, _i, _i, _i, _i, _p, _l, _i, _i, _p, _l, _i, _p]),
    'segf_bilinear_bwd_248': (_i, [_i, _i, _i, _i, _i, _p, _l, _p, _p, _p, _p]),
    'segf_nearest_up': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    'segf_upsample_add': (_i, [_i, _i, _i, _i, _i, _p, _l, _i, _p, _i, _i, _l, _p, _i, _i, _l, _p, _i, _i, _l, _p, _l, _i, _p]),
    'segf_upsample_add_stats_ws': (_l, [_i, _i, _i, _i]),
    'segf_upsample_add_stats': (_i, [_i, _i, _i, _i, _i, _p, _l, _i, _p, _i, _i, _l, _p, _i, _i, _l, _p, _i, _i, _l, _p, _l, _i, _p, _p, _p]),
    'segf_fuse_map_248_supported': (_i, [_i, _i, _i, _i, _i, _i]),
    'segf_fuse_map_248_ws': (_l, [_i, _i, _i, _i]),
    'segf_fuse_map_248': (_i, [_i, _i, _i, _i, _i, _p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _p, _l, _p, _p, _p]),
    'segf_bn_stats_from_sums': (_i, [_p, _l, _i, _p, _p, _p, _p, _f, _f, _p]),
    'segf_bilinear_to_nchw_f32': (_i, [_i, _i, _i, _i, _i, _p, _l, _i, _i, _p, _p]),
    'segf_ce_dice_stats_floats': (_l, [_i, _i]),
    'segf_ce_dice_lse_floats': (_l, [_i, _i, _i, _i, _i, _i, _i, _p, _l]),
    'segf_ce_dice_fwd': (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _l, _p, _l, _p, _i, _p, _p, _p, _p]),
    'segf_ce_dice_bwd_ws': (_l, [_i, _i, _i, _i, _i, _i, _i]),
    'segf_ce_dice_bwd': (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _l, _p, _l, _p, _i, _p, _p, _p, _l, _p, _p, _p]),
    'segf_debug_wave_reduce16': (_i, [_p, _p, _p]),
    'segf_argmax_confmat': (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _l, _p, _l, _p, _p, _p, _p, _p]),
    'segf_confmat_pairs': (_i, [_p, _p, _l, _i, _l, _p, _p, _p, _p]),
    'segf_agc_adamw': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _f, _f, _f, _f, _i, _f, _f, _p]),
    'segf_clip_grad_ws': (_l, []),
    'segf_clip_grad': (_i, [_p, _l, _i, _f, _p, _p]),
    'segf_zero': (_i, [_p, _l, _p]),
    'segf_add_i64': (_i, [_p, _l, _p]),
    'segf_hist_accum': (_i, [_p, _p, _l, _i, _p]),
    'segf_debug_spin': (_i, [_l, _p]),
    'segf_gemm8_option': (_i, [_i, _i]),
    'segf_bernoulli_scale': (_i, [_p, _p, _l, _l, _p, _p]),
    'segf_layernorm_bwd_fused': (_i, [_i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    'segf_layernorm_bwd_scaled': (_i, [_i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p]),
    'segf_quant_rows_fp8': (_i, [_i, _l, _i, _p, _l, _p, _l, _p, _p]),
    'segf_gemm_fp8_supported': (_i, [_l, _l, _l]),
    'segf_quant_tensor_fp8': (_i, [_i, _i, _l, _i, _p, _l, _p, _l, _p, _p, _p]),
    'segf_conv3x3_fp8_supported': (_i, [_i, _i, _i, _i, _i, _i]),
    'segf_conv3x3_fp8': (_i, [_i, _i, _i, _i, _i, _i, _p, _l, _p, _p, _l, _p, _p, _l, _p]),
    'segf_conv3x3_fp8_wgrad_supported': (_i, [_i, _i, _i, _i, _i]),
    'segf_conv3x3_fp8_wgrad': (_i, [_i, _i, _i, _i, _i, _p, _l, _p, _p, _l, _p, _p, _l, _i, _p, _p]),
    'segf_gemm_fp8': (_i, [_l, _l, _l, _p, _l, _p, _p, _l, _p, _p, _p, _l, _p, _l, _p, _l, _p]),
    'segf_linear_fp8_supported': (_i, [_i, _l, _l, _l]),
    'segf_linear_fp8': (_i, [_i, _l, _l, _l, _p, _l, _p, _p, _l, _p, _p, _l, _p, _p, _l, _p, _l, _p]),
    'segf_linear_fp8_wgrad_splitk': (_i, [_l, _l, _l]),
    'segf_linear_fp8_wgrad': (_i, [_l, _l, _l, _p, _l, _p, _p, _l, _p, _p, _l, _i, _p, _p]),
    'segf_input_train': (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    'segf_input_val_ws': (_l, [_i, _i, _i, _i]),
    'segf_input_val': (_i, [_p, _l, _p, _l, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    'segf_infer_preprocess': (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p]),
    'segf_event_create': (_i, [C.POINTER(C.c_void_p)]),
    'segf_event_destroy': (_i, [_p]),
    'segf_event_record': (_i, [_p, _p, _i]),
    'segf_stream_wait_event': (_i, [_p, _p]),
    'segf_version': (C.c_char_p, []),
    'segf_policy_count': (_i, []),
    'segf_policy_describe': (_i, [_i, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_char_p)]),
    'segf_policy_get': (_i, [C.c_char_p]),
    'segf_policy_set': (_i, [C.c_char_p, _i]),
    'segf_policy_reload': (None, []),
    'segf_trace_begin': (None, [_i]),
    'segf_trace_end': (_i, [C.c_char_p, _i]),
}


def lib():
    """Load the shared library (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'or `make -C segmentation_factory_amd/csrc`. There is no CPU/eager fallback.')
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def exported_symbols():
    return sorted(_PROTOS)


# ---- dispatch policy (csrc/policy.h; documented in include/segfac.h) and launch trace -------------------------------------------
_NO_SUCH = -2 ** 31


def policy(name):
    """Current value of one dispatch switch (field name 'gemm_no_narrow' or environment name 'SEGFAC_GEMM_NO_NARROW').  The library
    reads the environment once, at first use; `policy_set` / `policy_reload` change it afterwards."""
    v = lib().segf_policy_get(name.encode())
    if v == _NO_SUCH:
        raise KeyError(f'no dispatch switch named {name!r} (csrc/policy.h)')
    return v


def policy_set(name, value):
    prev = lib().segf_policy_set(name.encode(), int(value))
    if prev == _NO_SUCH:
        raise KeyError(f'no dispatch switch named {name!r} (csrc/policy.h)')
    return prev


def policy_reload():
    """Re-read every switch from the environment (after os.environ / monkeypatch.setenv changed it)."""
    lib().segf_policy_reload()


def policy_table():
    """[(field, environment name, value, default, description)] of every switch."""
    out = []
    for i in range(lib().segf_policy_count()):
        f, e, d = C.c_char_p(), C.c_char_p(), C.c_char_p()
        v, df = _i(), _i()
        _chk(lib().segf_policy_describe(i, C.byref(f), C.byref(e), C.byref(v), C.byref(df), C.byref(d)), 'segf_policy_describe')
        out.append((f.value.decode(), e.value.decode(), v.value, df.value, d.value.decode()))
    return out


class policy_override:
    """with policy_override(gemm_no_narrow=1): ...   (tests, A/B scripts)"""

    def __init__(self, **kw):
        self.kw, self.prev = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.prev[k] = policy_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            policy_set(k, v)
        return False


class trace:
    """Names of the kernels the library launches on this thread inside the block (instantiated names, template arguments included):
    `with hip.trace() as t: hip.gemm(...)` -> t.kernels.  dry_run=True records the decisions WITHOUT launching (no GPU needed; the
    entry points are then called through lib() with placeholder pointers, see tests/test_host_cpu.py)."""

    def __init__(self, dry_run=False, cap=1 << 16):
        self.dry, self.cap, self.kernels, self.launches = dry_run, cap, [], 0

    def __enter__(self):
        lib().segf_trace_begin(int(self.dry))
        return self

    def __exit__(self, *exc):
        buf = C.create_string_buffer(self.cap)
        self.launches = lib().segf_trace_end(buf, self.cap)
        self.kernels = [l for l in buf.value.decode().split('\n') if l]
        return False


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f'unsupported activation dtype {t.dtype}')


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError(f'{name} failed with code {rc} '
                           f'({"argument error" if rc < 0 else "hipError_t"})')


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('segmentation_factory_amd kernels need device tensors (no CPU fallback)')


def _f32(n, device):
    return torch.empty(max(int(n), 1), dtype=torch.float32, device=device)


# ---- plumbing ------------------------------------------------------------------------------
def cast(src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    _need_cuda(src)
    src = src.contiguous()
    if src.dtype == dtype:
        return src
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    _chk(lib().segf_cast(_ptr(src), dt_of(src), _ptr(dst), BF16 if dtype == torch.bfloat16 else F32,
                         src.numel(), _stream()), 'segf_cast')
    return dst


def cast_into(src: torch.Tensor, dst: torch.Tensor):
    """dst.flat[i] = src.flat[i] with dtype conversion, both contiguous (one launch; dst is reused across graph replays)."""
    _need_cuda(src, dst)
    assert src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    _chk(lib().segf_cast(_ptr(src), dt_of(src), _ptr(dst), dt_of(dst), src.numel(), _stream()), 'segf_cast')
    return dst


def cast2d(src: torch.Tensor, dst: torch.Tensor):
    """dst[r, c] = src[r, c] with dtype conversion; both 2-D with unit inner stride (any row stride), or 1-D."""
    _need_cuda(src, dst)
    if src.ndim == 1:
        src, dst = src.view(1, -1) if src.stride(0) == 1 else src.unsqueeze(1), dst.view(1, -1) if dst.stride(0) == 1 else dst.unsqueeze(1)
    rows, cols = src.shape
    assert tuple(dst.shape) == (rows, cols) and (cols == 1 or (src.stride(1) == 1 and dst.stride(1) == 1))
    _chk(lib().segf_cast2d(_ptr(src), dt_of(src), src.stride(0), _ptr(dst), dt_of(dst), dst.stride(0), rows, cols, _stream()),
         'segf_cast2d')
    return dst


def permute021(x: torch.Tensor, A: int, Bd: int, Cd: int, out_dtype: torch.dtype, ld_out=None, out=None) -> torch.Tensor:
    """out[a][c][b] = in[a][b][c]; last dim zero-padded to ld_out.  out: optional contiguous destination of A * Cd * ld_out elements."""
    _need_cuda(x)
    ld_out = Bd if ld_out is None else ld_out
    if out is None:
        out = torch.empty((A, Cd, ld_out), dtype=out_dtype, device=x.device)
    else:
        assert out.is_contiguous() and out.numel() == A * Cd * ld_out and out.dtype == out_dtype
    _chk(lib().segf_permute021(_ptr(x), dt_of(x), _ptr(out), BF16 if out_dtype == torch.bfloat16 else F32,
                               A, Bd, Cd, ld_out, _stream()), 'segf_permute021')
    return out


class SegfPrepItem(C.Structure):
    """include/segfac.h: one job of segf_prep_grouped"""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('rows', C.c_int64), ('cols', C.c_int64), ('ld_src', C.c_int64),
                ('ld_dst', C.c_int64), ('pb', C.c_int64), ('pc', C.c_int64), ('op', C.c_int32), ('src_dt', C.c_int32),
                ('dst_dt', C.c_int32), ('reserved', C.c_int32)]


def _as2d(src, dst):
    if src is not None and src.ndim == 1:
        src = src.view(1, -1) if src.stride(0) == 1 else src.unsqueeze(1)
    if dst.ndim == 1:
        dst = dst.view(1, -1) if dst.stride(0) == 1 else dst.unsqueeze(1)
    return src, dst


def prep_grouped(jobs):
    """Several small layout jobs in one launch (segf_prep_grouped); jobs must be independent of each other:
        ('cast', src, dst)                        dst[r, c] = src[r, c] (2-D with unit inner stride or a column, or 1-D), as cast2d
        ('zero', dst)                             dst[r, c] = 0
        ('perm', src, dst, A, Bd, Cd, ld_out[, a_stride])   dst[a][c][b] = src[a][b][c], last dim zero-padded to ld_out, as permute021;
                                                  a_stride: element distance between the out[a] blocks (rows of a wider matrix)"""
    if not jobs:
        return
    arr = (SegfPrepItem * len(jobs))()
    for k, job in enumerate(jobs):
        it = arr[k]
        it.pb = it.pc = it.ld_src = it.cols = 0
        it.reserved = 0
        if job[0] == 'perm':
            _, src, dst, A, Bd, Cd, ld_out = job[:7]
            astride = job[7] if len(job) > 7 else 0          # optional: distance between consecutive out[a] blocks (default Cd * ld_out)
            _need_cuda(src, dst)
            assert src.is_contiguous() and src.numel() == A * Bd * Cd
            if astride:
                assert astride >= Cd * ld_out and dst.ndim == 2 and dst.shape[0] == A and dst.stride(0) == astride and dst.stride(1) == 1
            else:
                assert dst.is_contiguous() and dst.numel() == A * Cd * ld_out
            it.op, it.src, it.dst, it.rows, it.pb, it.pc, it.ld_dst = 1, src.data_ptr(), dst.data_ptr(), A, Bd, Cd, ld_out
            it.cols = astride
            it.src_dt, it.dst_dt = dt_of(src), dt_of(dst)
        elif job[0] == 'cast':
            src, dst = _as2d(job[1], job[2])
            _need_cuda(src, dst)
            rows, cols = src.shape
            assert tuple(dst.shape) == (rows, cols) and (cols == 1 or (src.stride(1) == 1 and dst.stride(1) == 1)), (src.shape, dst.shape)
            it.op, it.src, it.dst, it.rows, it.cols = 0, src.data_ptr(), dst.data_ptr(), rows, cols
            it.ld_src, it.ld_dst, it.src_dt, it.dst_dt = max(src.stride(0), 1), max(dst.stride(0), 1), dt_of(src), dt_of(dst)
        else:
            assert job[0] == 'zero'
            _, dst = _as2d(None, job[1])
            _need_cuda(dst)
            rows, cols = dst.shape
            assert cols == 1 or dst.stride(1) == 1
            it.op, it.src, it.dst, it.rows, it.cols, it.ld_dst = 2, None, dst.data_ptr(), rows, cols, max(dst.stride(0), 1)
            it.src_dt = it.dst_dt = dt_of(dst)
    _chk(lib().segf_prep_grouped(len(jobs), C.cast(arr, C.c_void_p), _stream()), 'segf_prep_grouped')


def scale_rows(x: torch.Tensor, scale: torch.Tensor, rows_per_group: int) -> torch.Tensor:
    _need_cuda(x, scale)
    rows, cols = x.shape
    y = torch.empty((rows, cols), dtype=x.dtype, device=x.device)
    _chk(lib().segf_scale_rows(dt_of(x), _ptr(x), x.stride(0), _ptr(y), cols, _ptr(scale), rows, cols,
                               rows_per_group, _stream()), 'segf_scale_rows')
    return y


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _need_cuda(a, b)
    rows, cols = a.shape
    y = torch.empty((rows, cols), dtype=a.dtype, device=a.device)
    _chk(lib().segf_add(dt_of(a), _ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(y), cols, rows, cols, _stream()),
         'segf_add')
    return y


def colsum(x: torch.Tensor, out=None) -> torch.Tensor:
    _need_cuda(x)
    rows, cols = x.shape
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.numel() == cols and out.is_contiguous()
    ws = _f32(lib().segf_colsum_ws(rows, cols), x.device)
    _chk(lib().segf_colsum(dt_of(x), _ptr(x), x.stride(0), rows, cols, _ptr(out), _ptr(ws), _stream()), 'segf_colsum')
    return out


# ---- GEMM ------------------------------------------------------------------------------------
def gemm(layout: int, A: torch.Tensor, B: torch.Tensor, M: int, N: int, K: int, out=None, out_dtype=None, bias=None,
         residual=None, rscale=None, rows_per_group=1, split_k=1) -> torch.Tensor:
    """C[M,N] = A(m,k) B(k,n) (+bias) (residual + rscale * .).  A, B: 2-D tensors with unit inner stride."""
    _need_cuda(A, B)
    assert A.stride(-1) == 1 and B.stride(-1) == 1
    dt = dt_of(A)
    assert dt_of(B) == dt
    out_dtype = out_dtype or A.dtype
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    assert out.stride(-1) == 1
    ws = None
    if split_k > 1:
        ws = _f32(split_k * M * N, A.device)
    _chk(_timed(('gemm', layout, M, N, K), lambda: lib().segf_gemm(
        dt, layout, M, N, K, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), dt_of(out), out.stride(0), _ptr(bias),
        _ptr(residual), residual.stride(0) if residual is not None else 0, _ptr(rscale), rows_per_group, split_k, _ptr(ws),
        _stream())), 'segf_gemm')
    return out


def gemm_dw_db(dy: torch.Tensor, x: torch.Tensor, M: int, N: int, K: int, split_k=1, out=None, db_out=None):
    """(dW [M,N] fp32, db [M] fp32) = (dy^T x, column sums of dy) in one pass over dy [K,M]; x is [K,N]."""
    _need_cuda(dy, x)
    assert dy.stride(-1) == 1 and x.stride(-1) == 1 and dt_of(dy) == dt_of(x)
    dw = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=dy.device)
    assert dw.stride(-1) == 1
    db = db_out if db_out is not None else torch.empty(M, dtype=torch.float32, device=dy.device)
    assert db.dtype == torch.float32 and db.numel() == M and db.is_contiguous()
    ws = _f32(lib().segf_gemm_dw_db_ws(M, N, K, split_k), dy.device)
    _chk(_timed(('gemm', 2, M, N, K), lambda: lib().segf_gemm_dw_db(
        dt_of(dy), M, N, K, _ptr(dy), dy.stride(0), _ptr(x), x.stride(0), _ptr(dw), dt_of(dw), dw.stride(0), split_k, _ptr(ws),
        _ptr(db), _stream())), 'segf_gemm_dw_db')
    return dw, db


class SegfDwItem(C.Structure):
    """include/segfac.h: one layer of segf_gemm_dw_db_grouped"""
    _fields_ = [('M', C.c_int64), ('N', C.c_int64), ('K', C.c_int64), ('dy', C.c_void_p), ('lddy', C.c_int64), ('x', C.c_void_p),
                ('ldx', C.c_int64), ('dw', C.c_void_p), ('lddw', C.c_int64), ('db', C.c_void_p), ('ws', C.c_void_p), ('split_k', C.c_int),
                ('shared_split', C.c_int)]


def gemm_dw_db_grouped(items, shared_split=False):
    """items: [(dy [K, M], x [K, N], M, N, K, split_k, dw fp32 [M, N] view, db fp32 [M])]: the weight + bias gradients of several Linear
    layers in ONE C-ABI call (grouped launches where the shapes allow; each result bitwise equal to gemm_dw_db's)."""
    if not items:
        return
    arr = (SegfDwItem * len(items))()
    keep = []
    dt = None
    for k, (dy, x, M, N, K, split_k, dw, db) in enumerate(items):
        _need_cuda(dy, x, dw, db)
        assert dy.stride(-1) == 1 and x.stride(-1) == 1 and dw.stride(-1) == 1 and dt_of(dy) == dt_of(x)
        assert dw.dtype == torch.float32 and db.dtype == torch.float32 and db.numel() == M and db.is_contiguous()
        dt = dt_of(dy) if dt is None else dt
        assert dt_of(dy) == dt
        ws = _f32(lib().segf_gemm_dw_db_ws(M, N, K, split_k), dy.device)
        keep.append(ws)
        it = arr[k]
        it.M, it.N, it.K = M, N, K
        it.dy, it.lddy, it.x, it.ldx = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0)
        it.dw, it.lddw, it.db, it.ws, it.split_k, it.shared_split = dw.data_ptr(), dw.stride(0), db.data_ptr(), ws.data_ptr(), split_k, int(shared_split)
    _chk(lib().segf_gemm_dw_db_grouped(dt, len(items), C.cast(arr, C.c_void_p), _stream()), 'segf_gemm_dw_db_grouped')


def gemm_pro_supported(dtype, layout, M, N, K, rows_per_group):
    return bool(lib().segf_gemm_pro_supported(BF16 if dtype == torch.bfloat16 else F32, layout, M, N, K, rows_per_group))


def bn_affine_table(mean, rstd, gamma, beta, chan_scale, groups, Cc):
    """(scale, shift) fp32 [groups, C]: BatchNorm (+ Dropout2d channel scale) as the affine segf_gemm_pro applies."""
    _need_cuda(mean, rstd, gamma, beta)
    scale = torch.empty((groups, Cc), dtype=torch.float32, device=mean.device)
    shift = torch.empty_like(scale)
    _chk(lib().segf_bn_affine_table(_ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(chan_scale), groups, Cc, _ptr(scale),
                                    _ptr(shift), _stream()), 'segf_bn_affine_table')
    return scale, shift


def gemm_pro(layout, A, B, M, N, K, scale, shift, rows_per_group, act, bias=None, split_k=1, out=None):
    """segf_gemm with the activation operand read as act(x * scale[g] + shift[g]) (layout 0: A, bf16 out; layout 2: B, fp32 out)."""
    _need_cuda(A, B, scale, shift)
    assert A.stride(-1) == 1 and B.stride(-1) == 1
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype if layout == 0 else torch.float32, device=A.device)
    ws = _f32(split_k * M * N, A.device) if split_k > 1 else None
    _chk(_timed(('gemm', layout, M, N, K), lambda: lib().segf_gemm_pro(
        dt_of(A), layout, M, N, K, _ptr(A), A.stride(0), _ptr(B), B.stride(0), _ptr(out), dt_of(out), out.stride(0), _ptr(bias),
        split_k, _ptr(ws), _ptr(scale), _ptr(shift), rows_per_group, act, _stream())), 'segf_gemm_pro')
    return out


def quant_rows_fp8(x):
    """(q uint8 [rows, K] of e4m3fn bytes, scale fp32 [rows]): row-wise dynamic quantisation, x = q * scale."""
    _need_cuda(x)
    assert x.ndim == 2 and x.stride(1) == 1
    rows, K = x.shape
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
    scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    _chk(lib().segf_quant_rows_fp8(dt_of(x), rows, K, _ptr(x), x.stride(0), _ptr(q), K, _ptr(scale), _stream()), 'segf_quant_rows_fp8')
    return q, scale


def quant_tensor_fp8(x, e5m2=False):
    """(q uint8 [rows, cols], scale fp32 [1]): one dynamic scale for the whole tensor, x = q * scale; e4m3fn or e5m2 bytes."""
    _need_cuda(x)
    assert x.ndim == 2 and x.stride(1) == 1
    rows, cols = x.shape
    q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(1, dtype=torch.int32, device=x.device)
    _chk(lib().segf_quant_tensor_fp8(dt_of(x), int(e5m2), rows, cols, _ptr(x), x.stride(0), _ptr(q), cols, _ptr(scale), _ptr(ws),
                                     _stream()), 'segf_quant_tensor_fp8')
    return q, scale


def conv3x3_fp8_supported(mode, B, H, W, Cin, Cout):
    return bool(lib().segf_conv3x3_fp8_supported(mode, B, H, W, Cin, Cout))


def conv3x3_fp8(mode, xq, sx, wq, sw, B, H, W, Cin, Cout):
    """mode 0: y [P, Cout] bf16 from xq [P, Cin] e4m3 and wq [Cout, 9 Cin] e4m3; mode 1: dx [P, Cin] from gq [P, Cout] e5m2 and
    wq [Cin, 9 Cout] e4m3.  sx: tensor scale [1]; sw: one scale per weight row."""
    _need_cuda(xq, wq)
    assert xq.dtype == torch.uint8 and wq.dtype == torch.uint8 and xq.stride(-1) == 1 and wq.stride(-1) == 1
    P = B * H * W
    out = torch.empty((P, Cout if mode == 0 else Cin), dtype=torch.bfloat16, device=xq.device)
    key = ('conv3x3_fp8', mode, P, Cin, Cout)
    _chk(_timed(key, lambda: lib().segf_conv3x3_fp8(mode, B, H, W, Cin, Cout, _ptr(xq), xq.stride(0), _ptr(sx), _ptr(wq), wq.stride(0),
                                                    _ptr(sw), _ptr(out), out.stride(0), _stream())), 'segf_conv3x3_fp8')
    return out


_G8_FP8_STAGGER = None


def autotune_gemm8_fp8(force=False):
    """Pick the schedule of the fp8 eight-phase GEMM (segf_gemm8_option) for THIS device, once per process.  The staggered schedule needs
    26 % fewer cycles, but under it some MI355X devices drop their clock from 2.4 to about 1.5 GHz and finish later than with the
    lockstep schedule (UPerHead bottleneck conv, batch 32: 8.3 ms on devices that hold their clock, 12.7 ms on those that do not,
    10.9 ms in lockstep on both) -- so both are timed here under sustained load (about 0.1 s) and the faster one is kept.
    SEGFAC_G8_STAGGER=0/1 skips the measurement.  Never called while a stream is capturing."""
    global _G8_FP8_STAGGER
    if _G8_FP8_STAGGER is not None and not force:
        return _G8_FP8_STAGGER
    forced = policy('g8_stagger')
    if forced >= 0:
        _G8_FP8_STAGGER = int(forced != 0)
        lib().segf_gemm8_option(0, _G8_FP8_STAGGER)
        return _G8_FP8_STAGGER
    if torch.cuda.is_current_stream_capturing():
        return int(lib().segf_gemm8_option(0, -1))
    B, H, W, Cin, Cout = 8, 128, 128, 3072, 768
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(0)
    xq = torch.randint(0, 120, (B * H * W, Cin), dtype=torch.uint8, device=dev, generator=g)       # e4m3 bytes: finite values of mixed magnitude
    wq = torch.randint(0, 120, (Cout, 9 * Cin), dtype=torch.uint8, device=dev, generator=g)
    sx = torch.ones(1, device=dev)
    sw = torch.ones(Cout, device=dev)
    times = {0: [], 1: []}
    for rnd in range(2):
        for st in (1, 0):
            lib().segf_gemm8_option(0, st)
            conv3x3_fp8(0, xq, sx, wq, sw, B, H, W, Cin, Cout)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                conv3x3_fp8(0, xq, sx, wq, sw, B, H, W, Cin, Cout)
            e1.record()
            torch.cuda.synchronize()
            times[st].append(e0.elapsed_time(e1) / 8)
    _G8_FP8_STAGGER = int(min(times[1]) <= min(times[0]))
    lib().segf_gemm8_option(0, _G8_FP8_STAGGER)
    if os.environ.get('SEGFAC_VERBOSE'):
        print(f'[segfac] fp8 eight-phase GEMM: staggered {min(times[1]):.3f} ms, lockstep {min(times[0]):.3f} ms -> stagger = {_G8_FP8_STAGGER}')
    return _G8_FP8_STAGGER


def conv3x3_fp8_wgrad_supported(B, H, W, Cin, Cout):
    return bool(lib().segf_conv3x3_fp8_wgrad_supported(B, H, W, Cin, Cout))


def conv3x3_fp8_wgrad(xq, sx, gq, sg, B, H, W, Cin, Cout):
    """dW fp32 [Cout, 9 Cin] of the 3x3 convolution from the quantised input xq [P, Cin] (e4m3, scale sx) and the quantised output
    gradient gq [P, Cout] (e5m2, scale sg)."""
    _need_cuda(xq, gq)
    assert xq.dtype == torch.uint8 and gq.dtype == torch.uint8 and xq.stride(-1) == 1 and gq.stride(-1) == 1
    P = B * H * W
    out = torch.empty((Cout, 9 * Cin), dtype=torch.float32, device=xq.device)
    sk = pick_splitk(Cout, 9 * Cin, P)
    ws = _f32(sk * Cout * 9 * Cin, xq.device) if sk > 1 else None
    key = ('conv3x3_fp8_wgrad', P, Cin, Cout)
    _chk(_timed(key, lambda: lib().segf_conv3x3_fp8_wgrad(B, H, W, Cin, Cout, _ptr(xq), xq.stride(0), _ptr(sx), _ptr(gq), gq.stride(0), _ptr(sg),
                                                          _ptr(out), out.stride(0), sk, _ptr(ws) if ws is not None else None, _stream())),
         'segf_conv3x3_fp8_wgrad')
    return out


def linear_fp8_supported(mode, M, N, K):
    """mode 0 / 1: y [M, N] from [M, K] x [N, K]; mode 2: the weight gradient [N, K] over M tokens."""
    return bool(lib().segf_linear_fp8_supported(mode, M, N, K))


def linear_fp8(mode, aq, sa, bq, sb, bias=None, residual=None, rscale=None, rows_per_group=1):
    """bf16 [M, N] = sa * sb[None, :] * (aq . bq^T) (+ bias) (residual + rscale * .): aq [M, K] e4m3 (mode 0) or e5m2 (mode 1) with ONE
    scale sa [1], bq [N, K] e4m3 with a scale per row."""
    _need_cuda(aq, bq)
    assert aq.dtype == torch.uint8 and bq.dtype == torch.uint8 and aq.stride(-1) == 1 and bq.stride(-1) == 1
    M, K = aq.shape
    N = bq.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=aq.device)
    key = ('linear_fp8', mode, M, N, K)
    _chk(_timed(key, lambda: lib().segf_linear_fp8(mode, M, N, K, _ptr(aq), aq.stride(0), _ptr(sa), _ptr(bq), bq.stride(0), _ptr(sb),
                                                   _ptr(out), N, _ptr(bias), _ptr(residual),
                                                   residual.stride(0) if residual is not None else 0, _ptr(rscale), rows_per_group,
                                                   _stream())), 'segf_linear_fp8')
    return out


def linear_fp8_wgrad(gq, sg, xq, sx, out=None):
    """dW fp32 [N, K] = sg * sx * gq^T xq from the e5m2 gradient gq [T, N] and the e4m3 input xq [T, K] (tensor scales)."""
    _need_cuda(gq, xq)
    assert gq.dtype == torch.uint8 and xq.dtype == torch.uint8 and gq.stride(-1) == 1 and xq.stride(-1) == 1
    T, N = gq.shape
    K = xq.shape[1]
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=gq.device)
    sk = lib().segf_linear_fp8_wgrad_splitk(N, K, T)
    ws = _f32(sk * N * K, gq.device) if sk > 1 else None
    key = ('linear_fp8_wgrad', T, N, K)
    _chk(_timed(key, lambda: lib().segf_linear_fp8_wgrad(N, K, T, _ptr(gq), gq.stride(0), _ptr(sg), _ptr(xq), xq.stride(0), _ptr(sx),
                                                         _ptr(out), out.stride(0), sk, _ptr(ws), _stream())), 'segf_linear_fp8_wgrad')
    return out


def gemm_fp8_supported(M, N, K):
    return bool(lib().segf_gemm_fp8_supported(M, N, K))


def gemm_fp8(xq, sx, wq, sw, bias=None, residual=None, rscale=None, rows_per_group=1):
    """bf16 [M, N] = (xq . wq^T) * sx[:, None] * sw[None, :] (+ bias) (residual + rscale * .) on the fp8 matrix pipe."""
    _need_cuda(xq, wq)
    M, K = xq.shape
    N = wq.shape[0]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=xq.device)
    _chk(_timed(('gemm_fp8', M, N, K), lambda: lib().segf_gemm_fp8(
        M, N, K, _ptr(xq), xq.stride(0), _ptr(sx), _ptr(wq), wq.stride(0), _ptr(sw), _ptr(bias), _ptr(residual),
        residual.stride(0) if residual is not None else 0, _ptr(rscale), rows_per_group, _ptr(out), N, _stream())), 'segf_gemm_fp8')
    return out


class GraphEvent:
    """A HIP event that may be recorded INSIDE a stream capture as an external event-record node (hipEventRecordExternal) and
    waited on by a stream outside the graph at replay (graph.py).  torch.cuda.Event(external=True) raises on ROCm."""

    def __init__(self):
        h = C.c_void_p()
        _chk(lib().segf_event_create(C.byref(h)), 'segf_event_create')
        self.handle = h

    def record_external(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream()
        _chk(lib().segf_event_record(self.handle, s.cuda_stream, 1), 'segf_event_record')

    def record(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream()
        _chk(lib().segf_event_record(self.handle, s.cuda_stream, 0), 'segf_event_record')

    def wait(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream()
        _chk(lib().segf_stream_wait_event(s.cuda_stream, self.handle), 'segf_stream_wait_event')

    def __del__(self):
        try:
            if self.handle:
                lib().segf_event_destroy(self.handle)
        except Exception:
            pass


class KernelTimer:
    """HIP-event timing of selected launches on the stream they are enqueued on (bench.py's roofline leg).
    Usage: ``with KernelTimer(lambda key: key == ('gemm', 0, M, N, K)) as t: ...; t.summary()``."""
    active = None

    def __init__(self, match):
        self.match, self.events = match, {}

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *a):
        KernelTimer.active = None

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.events.items()}   # (launches, avg ms)


def _timed(key, fn):
    t = KernelTimer.active
    if t is None or not t.match(key):
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    t.events.setdefault(key, []).append((e0, e1))
    return r


def pick_splitk(M, N, K):
    return lib().segf_gemm_pick_splitk(M, N, K)


def pick_splitk_conv3x3(Cin, Cout, P):
    return lib().segf_conv3x3_pick_splitk(Cin, Cout, P)


# ---- norms ---------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, patch=None):
    """patch = (log2 W, log2 sr): also returns the output in the patch-major row order of a k = s = sr convolution's im2col matrix,
    [rows / sr^2, sr^2 C] (segf_layernorm_fwd_patch) -> (y, mean, rstd, col)."""
    _need_cuda(x, gamma, beta)
    rows, Cc = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if patch is not None:
        lw, ls = patch
        col = torch.empty((rows >> (2 * ls), Cc << (2 * ls)), dtype=x.dtype, device=x.device)
        _chk(lib().segf_layernorm_fwd_patch(dt_of(x), rows, Cc, _ptr(x), _ptr(gamma), _ptr(beta), eps, _ptr(y), _ptr(mean), _ptr(rstd),
                                            _ptr(col), lw, ls, _stream()), 'segf_layernorm_fwd_patch')
        return y, mean, rstd, col
    _chk(lib().segf_layernorm_fwd(dt_of(x), rows, Cc, _ptr(x), _ptr(gamma), _ptr(beta), eps, _ptr(y), _ptr(mean),
                                  _ptr(rstd), _stream()), 'segf_layernorm_fwd')
    return y, mean, rstd


def zeros(shape, dtype, device):
    """torch.zeros through the library's own fill kernel (keeps framework kernels out of the captured step)."""
    t = torch.empty(shape, dtype=dtype, device=device)
    _need_cuda(t)
    _chk(lib().segf_zero(_ptr(t), t.numel() * t.element_size(), _stream()), 'segf_zero')
    return t


def add_i64_(t, v=1):
    """t += v for an int64 scalar on the device (BatchNorm's num_batches_tracked)."""
    assert t.dtype == torch.int64 and t.numel() == 1 and t.is_cuda
    _chk(lib().segf_add_i64(_ptr(t), int(v), _stream()), 'segf_add_i64')
    return t


def debug_spin(us):
    """Test hook: hold the current stream for `us` microseconds."""
    _chk(lib().segf_debug_spin(int(us), _stream()), 'segf_debug_spin')


def hist_accum_(hist, counts, clear=True):
    """hist (fp32) += counts (int64) element by element, counts cleared: Metrics.update's `self.hist += bincount` (util/metrics.py:27)."""
    _need_cuda(hist, counts)
    assert hist.dtype == torch.float32 and counts.dtype == torch.int64 and hist.numel() == counts.numel()
    assert hist.is_contiguous() and counts.is_contiguous()
    _chk(lib().segf_hist_accum(_ptr(hist), _ptr(counts), hist.numel(), int(clear), _stream()), 'segf_hist_accum')
    return hist


def bernoulli_scale(state, keep_prob, n, row_len):
    """fp32 [n]: 1 / kp with probability kp, else 0, kp = keep_prob[i // row_len]; state = uint64-as-int64 [2] {seed, counter}."""
    _need_cuda(state, keep_prob)
    out = torch.empty(n, dtype=torch.float32, device=state.device)
    _chk(lib().segf_bernoulli_scale(_ptr(state), _ptr(keep_prob), n, row_len, _ptr(out), _stream()), 'segf_bernoulli_scale')
    return out


class SegfFinalizeItem(C.Structure):
    """include/segfac.h: one member of segf_colreduce_finalize_grouped"""
    _fields_ = [('partial', C.c_void_p), ('out', C.c_void_p), ('len', C.c_int64), ('nblk', C.c_int), ('scatter_c', C.c_int)]


def colreduce_finalize_grouped(items):
    """items: [(partial fp32 [nblk, n], nblk, n, out fp32 [n][, C])]: the finalize steps of several two-stage reductions in one launch."""
    if not items:
        return
    arr = (SegfFinalizeItem * len(items))()
    for k, item in enumerate(items):
        partial, nblk, n, out = item[:4]
        _need_cuda(partial, out)
        # (out may be the first of several ADJACENT views that together hold n elements: LayerNorm's dgamma | dbeta, a depthwise
        # convolution's dw | db -- the latter with a fifth member C: the [10][C] sums land as dw[C][9], db[C])
        assert partial.dtype == torch.float32 and out.dtype == torch.float32 and partial.numel() >= nblk * n
        arr[k].partial, arr[k].out, arr[k].len, arr[k].nblk = partial.data_ptr(), out.data_ptr(), n, nblk
        arr[k].scatter_c = item[4] if len(item) > 4 else 0
    _chk(lib().segf_colreduce_finalize_grouped(len(items), C.cast(arr, C.c_void_p), _stream()), 'segf_colreduce_finalize_grouped')


def layernorm_bwd(x, dy, gamma, mean, rstd, dgb_out=None, dy2=None, dres=None, defer=False, rscale=None, rows_per_group=1, dy2_patch=None):
    """dgb_out: optional (dgamma, dbeta) fp32 [C] views that are ADJACENT in memory (dbeta == dgamma + C): written in place.
    dy2 / dres: optional fan-in operands, dx = LN_bwd(dy + dy2) + dres (segf_layernorm_bwd_fused).
    defer=True (needs dgb_out): the kernel leaves its per-block partial sums and (dx, finalize item) is returned; the caller passes the
    item to colreduce_finalize_grouped later.
    rscale (fp32 [rows / rows_per_group]): the kernel also writes dxs = dx * rscale[row / rows_per_group] (what scale_rows(dx, ...) gives);
    it is returned as the attribute `dx.scaled` of the first result."""
    rows, Cc = x.shape
    dx = torch.empty_like(x)
    dxs = None
    if rscale is not None and rows < (1 << 22):
        _need_cuda(rscale)
        assert rscale.dtype == torch.float32 and rscale.is_contiguous() and rscale.numel() * rows_per_group >= rows
        dxs = torch.empty_like(x)
    ws = _f32(lib().segf_layernorm_bwd_ws(rows, Cc), x.device)
    # (dy2_patch = (log2 W, log2 sr): dy2 holds the same elements in patch-major row order, [rows / sr^2, sr^2 C])
    assert dres is None or (dres.shape == x.shape and dres.dtype == x.dtype and dres.is_contiguous())
    assert dy2 is None or (dy2.numel() == x.numel() and dy2.dtype == x.dtype and dy2.is_contiguous() and (dy2_patch is not None or dy2.shape == x.shape))
    if dgb_out is not None:
        dg, db = dgb_out
        assert db.data_ptr() == dg.data_ptr() + 4 * Cc and dg.numel() == Cc and db.numel() == Cc
    else:
        assert not defer
        dgb = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
        dg, db = dgb[0], dgb[1]
    lw, ls = dy2_patch if (dy2_patch is not None and dy2 is not None) else (-1, 0)
    _chk(lib().segf_layernorm_bwd_patch(dt_of(x), rows, Cc, _ptr(x), _ptr(dy), _ptr(dy2), _ptr(dres), _ptr(gamma), _ptr(mean),
                                        _ptr(rstd), _ptr(dx), None if defer else dg.data_ptr(), None if defer else db.data_ptr(),
                                        _ptr(ws), _ptr(rscale) if dxs is not None else None, int(rows_per_group), _ptr(dxs), lw, ls,
                                        _stream()), 'segf_layernorm_bwd_patch')
    dx.scaled = dxs
    if defer:
        return dx, (ws, int(lib().segf_layernorm_bwd_blocks(rows, Cc)), 2 * Cc, dg)
    return dx, dg, db


def bn_stats(x, running_mean, running_var, momentum, eps):
    rows, Cc = x.shape
    mean = torch.empty(Cc, dtype=torch.float32, device=x.device)
    rstd = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws = _f32(lib().segf_bn_ws(rows, Cc), x.device)
    _chk(lib().segf_bn_stats(dt_of(x), rows, Cc, _ptr(x), _ptr(mean), _ptr(rstd), _ptr(running_mean), _ptr(running_var),
                             momentum, eps, _ptr(ws), _stream()), 'segf_bn_stats')
    return mean, rstd


def bn_apply(x, mean, rstd, gamma, beta, act, chan_scale, rows_per_sample):
    rows, Cc = x.shape
    y = torch.empty_like(x)
    _chk(lib().segf_bn_apply(dt_of(x), rows, Cc, _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), act,
                             _ptr(chan_scale), rows_per_sample, _ptr(y), _stream()), 'segf_bn_apply')
    return y


def bn_bwd(x, dy, mean, rstd, gamma, beta, act, chan_scale, rows_per_sample, eval_mode):
    rows, Cc = x.shape
    dx = torch.empty_like(x)
    dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws = _f32(lib().segf_bn_ws(rows, Cc), x.device)
    _chk(lib().segf_bn_bwd(dt_of(x), rows, Cc, _ptr(x), _ptr(dy), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), act,
                           _ptr(chan_scale), rows_per_sample, int(eval_mode), _ptr(dx), _ptr(dgamma), _ptr(dbeta),
                           _ptr(ws), _stream()), 'segf_bn_bwd')
    return dx, dgamma, dbeta


def bn_cls_bwd_supported(dtype, M, Cc, K, rps):
    return dtype == torch.bfloat16 and bool(lib().segf_bn_cls_bwd_supported(BF16, M, Cc, K, rps))


def bn_cls_bwd(dy, w, x, mean, rstd, gamma, beta, act, chan_scale, rows_per_sample, eval_mode):
    """(dx, dgamma, dbeta) of BatchNorm(+act, +Dropout2d scale) whose output gradient is dy @ w (never materialised).
    dy: [M, K] class gradients (K % 32 == 0, pad columns zero); w: [K, C] bf16 classifier weight; x: [M, C] BatchNorm input."""
    _need_cuda(dy, w, x)
    M, Cc = x.shape
    K = w.shape[0]
    assert dy.shape[0] == M and dy.stride(1) == 1 and dy.stride(0) >= K and w.shape[1] == Cc and w.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x)
    dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
    ws = _f32(lib().segf_bn_cls_bwd_ws(M, Cc, rows_per_sample), x.device)
    _chk(lib().segf_bn_cls_bwd(BF16, M, Cc, K, _ptr(dy), dy.stride(0), _ptr(w), w.stride(0), _ptr(x), _ptr(mean), _ptr(rstd),
                               _ptr(gamma), _ptr(beta), act, _ptr(chan_scale), rows_per_sample, int(eval_mode), _ptr(dx),
                               _ptr(dgamma), _ptr(dbeta), _ptr(ws), _stream()), 'segf_bn_cls_bwd')
    return dx, dgamma, dbeta


def bn_cls_bwd_dw_supported(dtype, M, Cc, K, rps, C1):
    return dtype == torch.bfloat16 and bool(lib().segf_bn_cls_bwd_dw_supported(BF16, M, Cc, K, rps, C1))


def bn_cls_bwd_dw(dy, w, x, mean, rstd, gamma, beta, act, chan_scale, rows_per_sample, eval_mode, x1):
    """bn_cls_bwd plus the product that the consumer of dx needs next: dG fp32 [C, C1 + 8] = [dx^T x1 | colsum(dx) | 0]
    (segf_bn_cls_bwd_dw; x1: [M, C1] bf16, the stage-1 tokens of the folded SegFormerHead).  -> (dx, dgamma, dbeta, dG)."""
    _need_cuda(dy, w, x, x1)
    M, Cc = x.shape
    K, C1 = w.shape[0], x1.shape[1]
    assert dy.shape[0] == M and dy.stride(1) == 1 and dy.stride(0) >= K and w.shape[1] == Cc and w.is_contiguous() and x.is_contiguous()
    assert x1.shape[0] == M and x1.stride(1) == 1 and x1.dtype == torch.bfloat16
    dx = torch.empty_like(x)
    dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dG = torch.empty((Cc, C1 + 8), dtype=torch.float32, device=x.device)
    ws = _f32(lib().segf_bn_cls_bwd_dw_ws(M, Cc, rows_per_sample), x.device)
    _chk(lib().segf_bn_cls_bwd_dw(BF16, M, Cc, K, _ptr(dy), dy.stride(0), _ptr(w), w.stride(0), _ptr(x), _ptr(mean), _ptr(rstd),
                                  _ptr(gamma), _ptr(beta), act, _ptr(chan_scale), rows_per_sample, int(eval_mode), _ptr(dx),
                                  _ptr(dgamma), _ptr(dbeta), _ptr(ws), _ptr(x1), x1.stride(0), C1, _ptr(dG), _stream()),
         'segf_bn_cls_bwd_dw')
    return dx, dgamma, dbeta, dG


def bn_cls_bwd_full(dy, w, x, mean, rstd, gamma, beta, act, chan_scale, rows_per_sample, eval_mode, x1=None, want_dwcls=True):
    """bn_cls_bwd with everything that can ride along: dG (when x1 is given, as bn_cls_bwd_dw) and dwcls fp32 [K, C] = the
    classifier's weight gradient dy^T (act(bn(x)) * drop) (act 0 / 1 only).  -> (dx, dgamma, dbeta, dG | None, dwcls | None)."""
    _need_cuda(dy, w, x)
    M, Cc = x.shape
    K = w.shape[0]
    assert dy.shape[0] == M and dy.stride(1) == 1 and dy.stride(0) >= K and w.shape[1] == Cc and w.is_contiguous() and x.is_contiguous()
    assert act in (0, 1) or not want_dwcls
    dx = torch.empty_like(x)
    dgamma = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(Cc, dtype=torch.float32, device=x.device)
    dG, C1 = None, 0
    if x1 is not None:
        assert x1.shape[0] == M and x1.stride(1) == 1 and x1.dtype == torch.bfloat16
        C1 = x1.shape[1]
        dG = torch.empty((Cc, C1 + 8), dtype=torch.float32, device=x.device)
    dwcls = torch.empty((K, Cc), dtype=torch.float32, device=x.device) if want_dwcls else None
    ws = _f32(lib().segf_bn_cls_bwd_full_ws(M, Cc, K, rows_per_sample), x.device)
    _chk(lib().segf_bn_cls_bwd_full(BF16, M, Cc, K, _ptr(dy), dy.stride(0), _ptr(w), w.stride(0), _ptr(x), _ptr(mean), _ptr(rstd),
                                    _ptr(gamma), _ptr(beta), act, _ptr(chan_scale), rows_per_sample, int(eval_mode), _ptr(dx),
                                    _ptr(dgamma), _ptr(dbeta), _ptr(ws), _ptr(x1) if x1 is not None else None,
                                    x1.stride(0) if x1 is not None else 0, C1, _ptr(dG) if dG is not None else None,
                                    _ptr(dwcls) if dwcls is not None else None, _stream()), 'segf_bn_cls_bwd_full')
    return dx, dgamma, dbeta, dG, dwcls


def grn_fwd(x, gamma, beta, B, rps, pre_gelu=False):
    Cc = x.shape[1]
    y = torch.empty_like(x)
    a = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    sq = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    ws = _f32(lib().segf_grn_ws(B, rps, Cc, 0), x.device)
    _chk(lib().segf_grn_fwd(dt_of(x), B, rps, Cc, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(a), _ptr(sq), _ptr(ws),
                            int(pre_gelu), _stream()), 'segf_grn_fwd')
    return y, sq


def grn_bwd(x, dy, gamma, sq, B, rps, pre_gelu=False):
    Cc = x.shape[1]
    dx = torch.empty_like(x)
    dgb = torch.empty((2, Cc), dtype=torch.float32, device=x.device)
    ws = _f32(lib().segf_grn_ws(B, rps, Cc, 1), x.device)
    _chk(lib().segf_grn_bwd(dt_of(x), B, rps, Cc, _ptr(x), _ptr(dy), _ptr(gamma), _ptr(sq), _ptr(dx), dgb[0].data_ptr(),
                            dgb[1].data_ptr(), _ptr(ws), int(pre_gelu), _stream()), 'segf_grn_bwd')
    return dx, dgb[0], dgb[1]


# ---- attention ------------------------------------------------------------------------------------
def attention_fwd(q, k, v, B, heads, N, Nkv, hd, scale):
    """q: [B*N, >=heads*hd] view; k, v: [B*Nkv, ...] views (unit inner stride)."""
    o = torch.empty((B * N, heads * hd), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, heads, N), dtype=torch.float32, device=q.device)
    _chk(lib().segf_attention_fwd(dt_of(q), B, heads, N, Nkv, hd, _ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v),
                                  v.stride(0), scale, _ptr(o), o.stride(0), _ptr(lse), _stream()), 'segf_attention_fwd')
    return o, lse


def attention_bwd(q, k, v, o, d_o, lse, B, heads, N, Nkv, hd, scale, dk, dv):
    """dk / dv are written into caller-provided views (the two halves of the kv-linear gradient)."""
    dq = torch.empty((B * N, heads * hd), dtype=q.dtype, device=q.device)
    ws = _f32(lib().segf_attention_bwd_ws(B, heads, N, Nkv, hd), q.device)
    _chk(lib().segf_attention_bwd(dt_of(q), B, heads, N, Nkv, hd, _ptr(q), q.stride(0), _ptr(k), k.stride(0), _ptr(v),
                                  v.stride(0), scale, _ptr(o), o.stride(0), _ptr(d_o), d_o.stride(0), _ptr(lse),
                                  _ptr(dq), dq.stride(0), _ptr(dk), dk.stride(0), _ptr(dv), dv.stride(0), _ptr(ws),
                                  _stream()), 'segf_attention_bwd')
    return dq


# ---- spatial ----------------------------------------------------------------------------------------
def dwconv3x3_gelu_fwd(x, w9, bias, B, H, W, Cc, apply_gelu=True):
    y = torch.empty_like(x)
    _chk(lib().segf_dwconv3x3_gelu_fwd(dt_of(x), B, H, W, Cc, _ptr(x), _ptr(w9), _ptr(bias), int(apply_gelu), _ptr(y),
                                       _stream()), 'segf_dwconv3x3_gelu_fwd')
    return y


def dwconv3x3_gelu_bwd(x, w9, bias, dy, B, H, W, Cc, apply_gelu=True, dw_out=None, db_out=None, defer=False):
    """defer=True (needs dw_out and db_out ADJACENT in memory, db right behind dw): the kernels leave their per-block partial sums and
    (dx, finalize item) is returned; the caller passes the item to colreduce_finalize_grouped later."""
    du = torch.empty_like(x)
    dx = torch.empty_like(x)
    if defer:
        assert dw_out.dtype == db_out.dtype == torch.float32 and dw_out.is_contiguous() and dw_out.numel() == 9 * Cc
        assert db_out.numel() == Cc and db_out.data_ptr() == dw_out.data_ptr() + 36 * Cc
        ws = _f32(lib().segf_dwconv3x3_bwd_ws(B, H, W, Cc), x.device)
        _chk(lib().segf_dwconv3x3_gelu_bwd(dt_of(x), B, H, W, Cc, _ptr(x), _ptr(w9), _ptr(bias), int(apply_gelu), _ptr(dy),
                                           _ptr(du), _ptr(dx), None, None, _ptr(ws), _stream()), 'segf_dwconv3x3_gelu_bwd')
        return dx, (ws, int(lib().segf_dwconv3x3_bwd_blocks(dt_of(x), B, H, W, Cc)), 10 * Cc, dw_out, Cc)
    dw = dw_out if dw_out is not None else torch.empty((Cc, 9), dtype=torch.float32, device=x.device)
    db = db_out if db_out is not None else torch.empty(Cc, dtype=torch.float32, device=x.device)
    assert dw.is_contiguous() and dw.numel() == Cc * 9 and db.is_contiguous() and db.numel() == Cc and dw.dtype == db.dtype == torch.float32
    ws = _f32(lib().segf_dwconv3x3_bwd_ws(B, H, W, Cc), x.device)
    _chk(lib().segf_dwconv3x3_gelu_bwd(dt_of(x), B, H, W, Cc, _ptr(x), _ptr(w9), _ptr(bias), int(apply_gelu), _ptr(dy),
                                       _ptr(du), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws), _stream()),
         'segf_dwconv3x3_gelu_bwd')
    return dx, dw, db


def dwconv7x7_fwd(x, wt49, bias, B, H, W, Cc):
    """x: [B*H*W, C]; wt49: fp32 [49, C] (transposed depthwise weight)."""
    y = torch.empty_like(x)
    _chk(lib().segf_dwconv7x7_fwd(dt_of(x), B, H, W, Cc, _ptr(x), _ptr(wt49), _ptr(bias), _ptr(y), _stream()),
         'segf_dwconv7x7_fwd')
    return y


def dwconv7x7_bwd(x, wt49, dy, B, H, W, Cc, need_dx=True, need_db=True):
    dx = torch.empty_like(x) if need_dx else None
    dw = torch.empty((Cc, 49), dtype=torch.float32, device=x.device)
    db = torch.empty(Cc, dtype=torch.float32, device=x.device) if need_db else None
    ws = _f32(lib().segf_dwconv7x7_bwd_ws(B, H, W, Cc), x.device)
    _chk(lib().segf_dwconv7x7_bwd(dt_of(x), B, H, W, Cc, _ptr(x), _ptr(wt49), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db), _ptr(ws),
                                  _stream()), 'segf_dwconv7x7_bwd')
    return dx, dw, db


def conv3x3(mode, x, w, B, H, W, Cin, Cout, out=None, out_dtype=None, bias=None, split_k=1):
    """Implicit-GEMM 3x3 convolution (bf16).  mode 0: y = conv(x, w[Cout, 9*Cin]); mode 1: dx from (dy, wt[Cin, 9*Cout]);
    mode 2: dw[Cout, 9*Cin] fp32 from (x, dy).  x / w / out are 2-D views with unit inner stride."""
    _need_cuda(x, w)
    assert x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.stride(-1) == 1 and w.stride(-1) == 1
    P = B * H * W
    shape = {0: (P, Cout), 1: (P, Cin), 2: (Cout, 9 * Cin)}[mode]
    if out is None:
        out = torch.empty(shape, dtype=out_dtype or (torch.float32 if mode == 2 else x.dtype), device=x.device)
    if mode != 2:
        # few output tiles over a long reduction (PPM bottleneck, the small FPN levels): split-K slices of the eight-phase tile (segfac.h)
        split_k = int(lib().segf_conv3x3_fwd_splitk(mode, B, H, W, Cin, Cout)) if (bias is None and out.dtype == torch.bfloat16) else 1
    ws = _f32(split_k * shape[0] * shape[1], x.device) if split_k > 1 else None
    key = ('conv3x3', mode, P, Cin, Cout)
    _chk(_timed(key, lambda: lib().segf_conv3x3(mode, B, H, W, Cin, Cout, _ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(out),
                                                dt_of(out), out.stride(0), _ptr(bias), split_k, _ptr(ws), _stream())),
         'segf_conv3x3')
    return out


def gelu_fwd(u):
    y = torch.empty_like(u)
    _chk(lib().segf_gelu(dt_of(u), 0, _ptr(u), None, _ptr(y), u.numel(), _stream()), 'segf_gelu')
    return y


def gelu_bwd(u, dy):
    du = torch.empty_like(u)
    _chk(lib().segf_gelu(dt_of(u), 1, _ptr(u), _ptr(dy), _ptr(du), u.numel(), _stream()), 'segf_gelu')
    return du


def rowdot(a, b, extra_a=None, extra_b=None):
    """out[r] = sum_c a[r, c] * b[r, c] (+ extra_a[r] * extra_b[r]); fp32 2-D inputs."""
    rows, cols = a.shape
    out = torch.empty(rows, dtype=torch.float32, device=a.device)
    _chk(lib().segf_rowdot(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(extra_a), _ptr(extra_b), _ptr(out), rows, cols,
                           _stream()), 'segf_rowdot')
    return out


def adaptive_avgpool(x, B, H, W, Cc, S, bwd=False):
    """bwd=False: x [B*H*W, C] -> [B*S*S, C];  bwd=True: x = dout [B*S*S, C] -> din [B*H*W, C]."""
    out = torch.empty(((B * H * W) if bwd else (B * S * S), Cc), dtype=x.dtype, device=x.device)
    _chk(lib().segf_adaptive_avgpool(dt_of(x), int(bwd), B, H, W, Cc, S, _ptr(x), _ptr(out), _stream()), 'segf_adaptive_avgpool')
    return out


def im2col(x, dtype, in_nchw_f32, B, H, W, Cin, kh, kw, stride, pad, Ho, Wo, ldcol):
    _need_cuda(x)
    col = torch.empty((B * Ho * Wo, ldcol), dtype=dtype, device=x.device)
    _chk(lib().segf_im2col(BF16 if dtype == torch.bfloat16 else F32, int(in_nchw_f32), B, H, W, Cin, kh, kw, stride, pad,
                           Ho, Wo, _ptr(x), _ptr(col), ldcol, _stream()), 'segf_im2col')
    return col


def col2im(dcol, B, H, W, Cin, kh, kw, stride, pad, Ho, Wo):
    dx = torch.empty((B * H * W, Cin), dtype=dcol.dtype, device=dcol.device)
    _chk(lib().segf_col2im(dt_of(dcol), B, H, W, Cin, kh, kw, stride, pad, Ho, Wo, _ptr(dcol), dcol.stride(0), _ptr(dx),
                           _stream()), 'segf_col2im')
    return dx


def bilinear_fwd(x, B, h, w, Cc, H, W, out, align_corners=False):
    """x: [B*h*w, >=C] view; out: [B*H*W, >=C] view (may be a column slice of a concat buffer)."""
    _chk(lib().segf_bilinear_fwd(dt_of(x), B, h, w, Cc, _ptr(x), x.stride(0), H, W, _ptr(out), out.stride(0),
                                 int(align_corners), _stream()), 'segf_bilinear_fwd')
    return out


def argmax_rows(x, Cc):
    """int64 [rows]: arg max over the first Cc columns of x [rows, >= Cc] (lowest index on ties)."""
    _need_cuda(x)
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    _chk(lib().segf_argmax_rows(dt_of(x), x.shape[0], Cc, _ptr(x), x.stride(0), _ptr(out), _stream()), 'segf_argmax_rows')
    return out


def bilinear_bwd(dout, B, h, w, Cc, H, W, align_corners=False, ld_in=None):
    ld_in = Cc if ld_in is None else ld_in
    din = torch.empty((B * h * w, ld_in), dtype=dout.dtype, device=dout.device) if ld_in == Cc else zeros((B * h * w, ld_in), dout.dtype, dout.device)
    _chk(lib().segf_bilinear_bwd(dt_of(dout), B, h, w, Cc, _ptr(din), ld_in, H, W, _ptr(dout), dout.stride(0),
                                 int(align_corners), _stream()), 'segf_bilinear_bwd')
    return din


def nearest_up(x, B, h, w, Cc, H, W, base=None, bwd=False):
    """bwd=False: x [B*h*w, C] -> [B*H*W, C], nearest source per destination (integer factors or ATen's floorf(dst * in / out) for
    any other pair of sizes) (+ base); bwd=True: x = dout [B*H*W, C] -> sums over each source's destinations [B*h*w, C]."""
    _need_cuda(x, base)
    assert x.is_contiguous() and (base is None or base.is_contiguous())
    out = torch.empty(((B * h * w) if bwd else (B * H * W), Cc), dtype=x.dtype, device=x.device)
    _chk(lib().segf_nearest_up(dt_of(x), int(bwd), B, h, w, Cc, H, W, _ptr(x), _ptr(base), _ptr(out), _stream()), 'segf_nearest_up')
    return out


class InputSample(C.Structure):
    """segf_input_sample of include/segfac.h (72 bytes)."""
    _fields_ = [('img', C.c_uint64), ('lbl', C.c_uint64), ('img_stride', C.c_int64), ('lbl_stride', C.c_int64),
                ('src_h', C.c_int32), ('src_w', C.c_int32), ('top', C.c_int32), ('left', C.c_int32),
                ('flip', C.c_int32), ('order', C.c_int32), ('factor', C.c_float * 3), ('reserved', C.c_int32)]


def input_train(samples, B, H, W, mean3, std3, label_lut, out_img=None, out_lbl=None):
    """segf_input_train: `samples` = uint8 device tensor holding B packed InputSample records (their image / label pointers must
    stay alive until the stream has run the kernels) -> (fp32 [B, 3, H, W], int64 [B, H, W])."""
    _need_cuda(samples, mean3, std3, label_lut)
    assert samples.dtype == torch.uint8 and samples.numel() == B * C.sizeof(InputSample)
    assert mean3.dtype == torch.float32 and std3.dtype == torch.float32 and mean3.numel() == 3 and std3.numel() == 3
    assert label_lut is None or (label_lut.dtype == torch.int64 and label_lut.numel() == 256)
    dev = samples.device
    if out_img is None:
        out_img = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    if out_lbl is None:
        out_lbl = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    assert out_img.is_contiguous() and out_lbl.is_contiguous() and out_img.numel() == B * 3 * H * W and out_lbl.numel() == B * H * W
    lsum = torch.empty(max(B, 2), dtype=torch.int64, device=dev)
    _chk(lib().segf_input_train(_ptr(samples), B, H, W, _ptr(lsum), _ptr(mean3), _ptr(std3), _ptr(label_lut), _ptr(out_img),
                                _ptr(out_lbl), _stream()), 'segf_input_train')
    return out_img, out_lbl


def input_val(img, lbl, out_h, out_w, mean3, std3, label_lut):
    """segf_input_val: img uint8 [h, w, 3], lbl uint8 [h, w] (device) -> (fp32 [3, out_h, out_w], int64 [out_h, out_w])."""
    _need_cuda(img, lbl, mean3, std3, label_lut)
    assert img.dtype == torch.uint8 and lbl.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3 and img.stride(2) == 1 \
        and img.stride(1) == 3 and lbl.stride(1) == 1 and lbl.shape == img.shape[:2]
    assert label_lut is None or (label_lut.dtype == torch.int64 and label_lut.numel() == 256)
    h, w = int(img.shape[0]), int(img.shape[1])
    ws = torch.empty(int(lib().segf_input_val_ws(h, w, out_h, out_w)) + 16, dtype=torch.uint8, device=img.device)
    out_img = torch.empty((3, out_h, out_w), dtype=torch.float32, device=img.device)
    out_lbl = torch.empty((out_h, out_w), dtype=torch.int64, device=img.device)
    _chk(lib().segf_input_val(_ptr(img), img.stride(0), _ptr(lbl), lbl.stride(0), h, w, out_h, out_w, _ptr(ws), _ptr(mean3),
                              _ptr(std3), _ptr(label_lut), _ptr(out_img), _ptr(out_lbl), _stream()), 'segf_input_val')
    return out_img, out_lbl


def infer_preprocess(img, out_h, out_w, mean3, std3):
    """segf_infer_preprocess: uint8 [3, h, w] (device) -> fp32 [1, 3, out_h, out_w]: torchvision-0.15 tensor resize + / 255 + normalise."""
    _need_cuda(img, mean3, std3)
    assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[0] == 3 and img.is_contiguous()
    out = torch.empty((1, 3, out_h, out_w), dtype=torch.float32, device=img.device)
    _chk(lib().segf_infer_preprocess(_ptr(img), int(img.shape[1]), int(img.shape[2]), out_h, out_w, _ptr(mean3), _ptr(std3), _ptr(out),
                                     _stream()), 'segf_infer_preprocess')
    return out


def bilinear_bwd_248(dout, B, H, W, Cc):
    """(d2, d4, d8): transposed x2 / x4 / x8 bilinear resizes of dout [B*H*W, >= C] in one pass (segf_bilinear_bwd_248)."""
    _need_cuda(dout)
    outs = [torch.empty((B * (H // r) * (W // r), Cc), dtype=dout.dtype, device=dout.device) for r in (2, 4, 8)]
    _chk(lib().segf_bilinear_bwd_248(dt_of(dout), B, H, W, Cc, _ptr(dout), dout.stride(0), _ptr(outs[0]), _ptr(outs[1]),
                                     _ptr(outs[2]), _stream()), 'segf_bilinear_bwd_248')
    return outs


def upsample_add(base, srcs, B, H, W, Cc, align_corners=False):
    """out = base + sum_k bilinear_up(src_k); srcs = [(tokens [B*h*w, C], h, w), ...] (at most 3)."""
    out = torch.empty((B * H * W, Cc), dtype=base.dtype, device=base.device)
    a = []
    for k in range(3):
        if k < len(srcs):
            t, h, w = srcs[k]
            a += [_ptr(t), h, w, t.stride(0)]
        else:
            a += [None, 0, 0, 0]
    _chk(lib().segf_upsample_add(dt_of(base), B, H, W, Cc, _ptr(base), base.stride(0), len(srcs), *a, _ptr(out), Cc,
                                 int(align_corners), _stream()), 'segf_upsample_add')
    return out


def upsample_add_stats(base, srcs, B, H, W, Cc):
    """upsample_add that also returns sums fp32 [2, C] = per-channel (sum, sum of squares) of the stored result (the BatchNorm
    statistics of the consumer); None for the sums when the geometry is not the fused 1/2-1/4-1/8 case."""
    out = torch.empty((B * H * W, Cc), dtype=base.dtype, device=base.device)
    a = []
    for k in range(3):
        if k < len(srcs):
            t, h, w = srcs[k]
            a += [_ptr(t), h, w, t.stride(0)]
        else:
            a += [None, 0, 0, 0]
    sums = torch.empty((2, Cc), dtype=torch.float32, device=base.device)
    ws = _f32(max(1, lib().segf_upsample_add_stats_ws(B, H, W, Cc)), base.device)
    rc = lib().segf_upsample_add_stats(dt_of(base), B, H, W, Cc, _ptr(base), base.stride(0), len(srcs), *a, _ptr(out), Cc, 0,
                                       _ptr(sums), _ptr(ws), _stream())
    if rc == ERR_SHAPE:
        return upsample_add(base, srcs, B, H, W, Cc), None
    _chk(rc, 'segf_upsample_add_stats')
    return out, sums


def fuse_map_248_supported(dtype, B, H, W, Cc, C1):
    return dtype == torch.bfloat16 and bool(lib().segf_fuse_map_248_supported(BF16, B, H, W, Cc, C1))


def fuse_map_248(x1, g1, t2, t3, t4, B, H, W, with_sums=True):
    """out [B*H*W, C] = x1 g1^T + bilinear(t2) + bilinear(t3) + bilinear(t4) (1/2, 1/4, 1/8 maps, align_corners=False) and, when
    asked, sums fp32 [2, C] = per-channel (sum, sum of squares) of the result: the folded SegFormerHead's stride-4 map in one launch."""
    _need_cuda(x1, g1, t2, t3, t4)
    Cc, C1 = g1.shape
    out = torch.empty((B * H * W, Cc), dtype=torch.bfloat16, device=x1.device)
    sums = torch.empty((2, Cc), dtype=torch.float32, device=x1.device) if with_sums else None
    ws = _f32(max(1, lib().segf_fuse_map_248_ws(B, H, W, Cc)), x1.device) if with_sums else None
    _chk(lib().segf_fuse_map_248(B, H, W, Cc, C1, _ptr(x1), x1.stride(0), _ptr(g1), g1.stride(0), _ptr(t2), t2.stride(0),
                                 _ptr(t3), t3.stride(0), _ptr(t4), t4.stride(0), _ptr(out), Cc,
                                 _ptr(sums) if with_sums else None, _ptr(ws) if with_sums else None, _stream()), 'segf_fuse_map_248')
    return out, sums


def bn_stats_from_sums(sums, rows, running_mean, running_var, momentum, eps):
    Cc = sums.shape[1]
    mean = torch.empty(Cc, dtype=torch.float32, device=sums.device)
    rstd = torch.empty(Cc, dtype=torch.float32, device=sums.device)
    _chk(lib().segf_bn_stats_from_sums(_ptr(sums), rows, Cc, _ptr(mean), _ptr(rstd), _ptr(running_mean), _ptr(running_var),
                                       momentum, eps, _stream()), 'segf_bn_stats_from_sums')
    return mean, rstd


def bilinear_to_nchw_f32(x, B, h, w, Cc, H, W):
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=x.device)
    _chk(lib().segf_bilinear_to_nchw_f32(dt_of(x), B, h, w, Cc, _ptr(x), x.stride(0), H, W, _ptr(out), _stream()),
         'segf_bilinear_to_nchw_f32')
    return out


# ---- loss / metrics ------------------------------------------------------------------------------------
def ce_dice_fwd(logits, B, Cc, h, w, H, W, target, ignore_index, class_weight, dice, want_lse=False):
    """(loss[3], stats) or, with want_lse, (loss[3], stats, lse): lse = the per-pixel log-sum buffer for ce_dice_bwd of the
    same logits (None when this configuration does not produce one)."""
    stats = _f32(lib().segf_ce_dice_stats_floats(B, Cc), logits.device)
    loss = torch.empty(3, dtype=torch.float32, device=logits.device)
    lse = None
    if want_lse:
        n = lib().segf_ce_dice_lse_floats(dt_of(logits), B, Cc, h, w, H, W, _ptr(logits), logits.stride(0))
        lse = torch.empty(n, dtype=torch.float32, device=logits.device) if n else None
    _chk(lib().segf_ce_dice_fwd(dt_of(logits), B, Cc, h, w, H, W, _ptr(logits), logits.stride(0), _ptr(target),
                                int(ignore_index), _ptr(class_weight), int(dice), _ptr(stats), _ptr(loss), _ptr(lse), _stream()),
         'segf_ce_dice_fwd')
    return (loss, stats, lse) if want_lse else (loss, stats)


def ce_dice_bwd(logits, B, Cc, h, w, H, W, target, ignore_index, class_weight, dice, stats, grad_out, lse=None):
    """d loss / d (low-res logits), same row stride as `logits` (pad columns zeroed).  lse: ce_dice_fwd(want_lse=True)'s."""
    ld = logits.stride(0)
    dlow = torch.empty((B * h * w, ld), dtype=logits.dtype, device=logits.device)
    dt = dt_of(logits)
    nws = lib().segf_ce_dice_bwd_ws(dt, B, Cc, h, w, H, W)
    ws = _f32(nws, logits.device) if nws else None
    _chk(_timed(('ce_dice_bwd', B, Cc, h, w, H, W), lambda: lib().segf_ce_dice_bwd(
        dt, B, Cc, h, w, H, W, _ptr(logits), ld, _ptr(target), int(ignore_index), _ptr(class_weight), int(dice), _ptr(stats),
        _ptr(grad_out), _ptr(dlow), ld, _ptr(ws), _ptr(lse), _stream())), 'segf_ce_dice_bwd')
    return dlow


def argmax_confmat(logits, B, Cc, h, w, H, W, target, ignore_label, mat, hist, flag, pred_out=None):
    _chk(lib().segf_argmax_confmat(dt_of(logits), B, Cc, h, w, H, W, _ptr(logits), logits.stride(0), _ptr(target),
                                   int(ignore_label), _ptr(mat), _ptr(hist), _ptr(flag), _ptr(pred_out), _stream()),
         'segf_argmax_confmat')


def confmat_pairs(gt, pred, Cc, ignore_label, mat, hist, flag):
    _need_cuda(gt, pred)
    _chk(lib().segf_confmat_pairs(_ptr(gt), _ptr(pred), gt.numel(), Cc, int(ignore_label), _ptr(mat), _ptr(hist),
                                  _ptr(flag), _stream()), 'segf_confmat_pairs')


def clip_grad(grad, mode, value, ws=None):
    """segf_clip_grad on a flat fp32 gradient buffer, in place: mode 'norm' (clip_grad_norm_, L2) or 'value' (clip_grad_value_)."""
    _need_cuda(grad)
    assert grad.dtype == torch.float32 and grad.is_contiguous()
    m = {'norm': 0, 'value': 1}[mode]
    if m == 0 and ws is None:
        ws = _f32(lib().segf_clip_grad_ws(), grad.device)
    _chk(lib().segf_clip_grad(_ptr(grad), grad.numel(), m, float(value), _ptr(ws), _stream()), 'segf_clip_grad')
    return grad


def agc_adamw(param, grad, exp_avg, exp_avg_sq, unit_off, unit_len, unit_flags, lr, beta1, beta2, eps, weight_decay,
              step, clip_factor, agc_eps=1e-3, unit_step=None):
    _chk(lib().segf_agc_adamw(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(unit_off), _ptr(unit_len),
                              _ptr(unit_flags), _ptr(unit_step) if unit_step is not None else None, unit_len.numel(), lr, beta1, beta2, eps, weight_decay, step, clip_factor,
                              agc_eps, _stream()), 'segf_agc_adamw')
