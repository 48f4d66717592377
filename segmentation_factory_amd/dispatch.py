"""Which kernel does a shape take?  Recording and replaying C-ABI calls with the library's launch trace (include/segfac.h:
segf_trace_begin / segf_trace_end; csrc/policy.h).

`record()` wraps the loaded library: every call of an entry point whose arguments are plain integers, floats and DEVICE pointers is
noted as (entry point, arguments with pointers reduced to their alignment, kernels it launched).  `replay(entry)` repeats such a call
as a DRY RUN -- placeholder pointers, nothing launched, no GPU needed -- and returns the kernels the library would launch today.
tools/make_dispatch_table.py records one train step of every BASELINE configuration on the MI355X into
tests/golden/dispatch_table.json; tests/test_host_cpu.py::test_dispatch_of_baseline_shapes replays the table on the CPU: a dispatch
edit that moves a BASELINE shape to another kernel shows up as a diff (the shapes come from the reference's width rule,
models/build_models.py:43-54).
"""
import ctypes as C

from . import hip

# entry points that are not replayable (host-side struct arrays, events) or not dispatch decisions (queries, policy, trace)
_SKIP_PREFIX = ('segf_policy', 'segf_trace', 'segf_event', 'segf_stream', 'segf_version', 'segf_debug', 'segf_gemm8_option')
_SKIP = {'segf_prep_grouped', 'segf_colreduce_finalize_grouped', 'segf_input_train', 'segf_input_val', 'segf_infer_preprocess',
         'segf_bernoulli_scale', 'segf_agc_adamw', 'segf_clip_grad'}
_PLACEHOLDER = 0x7f0000000000          # never dereferenced: dry runs skip every launch


def _replayable(name):
    if name.startswith(_SKIP_PREFIX) or name in _SKIP:
        return False
    res, args = hip._PROTOS[name]
    return res is hip._i and all(a in (hip._i, hip._l, hip._f, hip._p) for a in args)


def _enc(argtype, v):
    if argtype is hip._p:
        if v is None or v == 0:
            return None
        return 'p%d' % (int(v) & 255)                 # the pointer's alignment is all a dispatch rule may look at
    if argtype is hip._f:
        return float(v)
    return int(v)


def _dec(argtype, v):
    if argtype is hip._p:
        return None if v is None else _PLACEHOLDER + int(v[1:])
    return v


class _Recorder:
    def __init__(self, real, sink):
        self._real, self._sink = real, sink

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name == 'segf_gemm_dw_db_grouped':
            def grouped(dt, n, arr, stream):
                items = C.cast(arr, C.POINTER(hip.SegfDwItem))
                desc = [[items[k].M, items[k].N, items[k].K, items[k].lddy, items[k].ldx, items[k].lddw, items[k].split_k, items[k].shared_split]
                        for k in range(n)]
                with hip.trace() as t:
                    rc = fn(dt, n, arr, stream)
                self._sink.append({'fn': name, 'args': [int(dt), desc], 'kernels': t.kernels})
                return rc
            return grouped
        if name not in hip._PROTOS or not _replayable(name):
            return fn
        argtypes = hip._PROTOS[name][1]

        def call(*args):
            with hip.trace() as t:
                rc = fn(*args)
            if t.kernels:
                self._sink.append({'fn': name, 'args': [_enc(a, v) for a, v in zip(argtypes, args)], 'kernels': t.kernels})
            return rc
        return call


class record:
    """with dispatch.record() as calls: <run anything through segmentation_factory_amd> -> calls = [{'fn', 'args', 'kernels'}, ...]
    (every launching call, in order; duplicates included)."""

    def __enter__(self):
        self.calls = []
        self._real = hip.lib()
        hip._lib = _Recorder(self._real, self.calls)
        return self.calls

    def __exit__(self, *exc):
        hip._lib = self._real
        return False


def unique(calls):
    """Collapse identical (entry point, arguments) calls; 'count' = how often the step makes the call."""
    seen, out = {}, []
    for c in calls:
        key = (c['fn'], repr(c['args']))
        if key in seen:
            e = out[seen[key]]
            e['count'] += 1
            assert e['kernels'] == c['kernels'], ('same call, different kernels', c, e)
        else:
            seen[key] = len(out)
            out.append(dict(c, count=1))
    return out


def replay(entry):
    """Dry run of one recorded call: the kernel names the library's dispatch picks NOW for these arguments."""
    lib = hip.lib()
    name, args = entry['fn'], entry['args']
    with hip.trace(dry_run=True) as t:
        if name == 'segf_gemm_dw_db_grouped':
            dt, desc = args
            arr = (hip.SegfDwItem * len(desc))()
            for k, (M, N, K, lddy, ldx, lddw, split_k, shared) in enumerate(desc):
                it = arr[k]
                it.M, it.N, it.K, it.lddy, it.ldx, it.lddw, it.split_k, it.shared_split = M, N, K, lddy, ldx, lddw, split_k, shared
                it.dy, it.x, it.dw, it.db, it.ws = (_PLACEHOLDER + 0x1000 * (5 * k + j) for j in range(5))
            rc = lib.segf_gemm_dw_db_grouped(dt, len(desc), C.cast(arr, C.c_void_p), None)
        else:
            argtypes = hip._PROTOS[name][1]
            rc = getattr(lib, name)(*[_dec(a, v) for a, v in zip(argtypes, args)])
    if rc != 0:
        raise RuntimeError(f'{name}{tuple(args)} returned {rc} in a dry run')
    return t.kernels
