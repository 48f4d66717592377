// Elementwise / re-layout plumbing kernels (HBM-bound streaming; 16 B per lane where alignment allows).
#include "colreduce.h"

extern "C" const char* segf_version(void) { return "segfac-hip 0.1 gfx950"; }

// ---- cast ---------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n, bool vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t n8 = n / 8;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
            float v[8];
            load8<S>(src + i * 8, v);
            store8<D>(dst + i * 8, v);
        }
        for (int64_t i = n8 * 8 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
            stf<D>(dst + i, ldf<S>(src + i));
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
            stf<D>(dst + i, ldf<S>(src + i));
    }
}

extern "C" int segf_cast(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, void* stream) {
    if (n <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    const int blocks = (int)imin64(cdiv64(cdiv64(n, 8), 256), 2048);
    SEGF_DISPATCH_DT(src_dt, S, {
        SEGF_DISPATCH_DT(dst_dt, D, {
            hipLaunchKernelGGL((cast_kernel<S, D>), dim3(blocks), dim3(256), 0, st, (const S*)src, (D*)dst, n, vec);
        })
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- strided 2-D cast / copy: dst[r][c] = (D) src[r][c] with leading dimensions (weight packing, column extraction) ----
template <typename S, typename D>
__global__ void cast2d_kernel(const S* __restrict__ src, int64_t lds, D* __restrict__ dst, int64_t ldd, int64_t rows, int64_t cols) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols, c = i - r * cols;
        stf<D>(dst + r * ldd + c, ldf<S>(src + r * lds + c));
    }
}
extern "C" int segf_cast2d(const void* src, int src_dt, int64_t ld_src, void* dst, int dst_dt, int64_t ld_dst, int64_t rows,
                           int64_t cols, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    if (ld_src < 1 || ld_dst < 1) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)imin64(cdiv64(rows * cols, 256), 2048);
    SEGF_DISPATCH_DT(src_dt, S, {
        SEGF_DISPATCH_DT(dst_dt, D, {
            hipLaunchKernelGGL((cast2d_kernel<S, D>), dim3(blocks), dim3(256), 0, st, (const S*)src, ld_src, (D*)dst, ld_dst, rows, cols);
        })
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- permute021: out[a][c][b] = in[a][b][c], 32x32 LDS-tiled transpose -----------------------------
template <typename S, typename D>
__global__ void permute021_kernel(const S* __restrict__ in, D* __restrict__ out, int64_t Bd, int64_t Cd, int64_t ld_out) {
    __shared__ float tile[32][33];
    const int64_t a = blockIdx.z;
    const int64_t b0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    const S* src = in + a * Bd * Cd;
    D* dst = out + a * Cd * ld_out;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int64_t b = b0 + i, c = c0 + tx;
        tile[i][tx] = (b < Bd && c < Cd) ? ldf<S>(src + b * Cd + c) : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int64_t c = c0 + i, b = b0 + tx;
        if (c < Cd && b < ld_out) stf<D>(dst + c * ld_out + b, b < Bd ? tile[tx][i] : 0.f);
    }
}

extern "C" int segf_permute021(const void* in, int in_dt, void* out, int out_dt, int64_t A, int64_t Bd, int64_t Cd,
                               int64_t ld_out, void* stream) {
    if (A <= 0 || Bd <= 0 || Cd <= 0) return 0;
    if (ld_out < Bd || A > 65535) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    // the b-tiles must also cover the zero padding columns [Bd, ld_out)
    dim3 grid((unsigned)cdiv64(Cd, 32), (unsigned)cdiv64(ld_out, 32), (unsigned)A);
    SEGF_DISPATCH_DT(in_dt, S, {
        SEGF_DISPATCH_DT(out_dt, D, {
            hipLaunchKernelGGL((permute021_kernel<S, D>), grid, dim3(256), 0, st, (const S*)in, (D*)out, Bd, Cd, ld_out);
        })
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- grouped tensor preparation: several cast2d / permute021 / zero-fill jobs in ONE launch -----------------------------------
// At the reference's default batch of 4 (train_gpu.py:71) a step holds ~140 launches that move a few kilobytes each (weight re-layouts
// of the patch / spatial-reduction convolutions, mit.py:105,47; packing of the decode head's folded weights, heads/segformer.py:42-56;
// gradient hand-over into the optimizer's flat buffer): each costs the ~4 us launch floor.  The job table rides in the kernel
// arguments; a block finds its job by a scan of the (wave-uniform) prefix table.  Element values are those of the single kernels.
#define PREP_MAX 24
struct PrepGroup { int n; int first[PREP_MAX + 1]; SegfPrepItem it[PREP_MAX]; };
__device__ __forceinline__ float prep_ld(const void* p, int64_t i, int dt) {
    return dt == SEGF_F32 ? ((const float*)p)[i] : bf2f(((const bf16_t*)p)[i]);
}
__device__ __forceinline__ void prep_st(void* p, int64_t i, int dt, float v) {
    if (dt == SEGF_F32) ((float*)p)[i] = v; else ((bf16_t*)p)[i] = f2bf(v);
}
__global__ void __launch_bounds__(256) prep_group_kernel(const PrepGroup g) {
    __shared__ float tile[32][33];
    int k = 0;
    while (k + 1 < g.n && (int)blockIdx.x >= g.first[k + 1]) ++k;
    const SegfPrepItem& it = g.it[k];
    const int blk = blockIdx.x - g.first[k], nblk = g.first[k + 1] - g.first[k];
    if (it.op == 1) {                                  // out[a][c][b] = in[a][b][c], columns [pb, ld_dst) of the output zero
        const int64_t tc = (it.pc + 31) / 32, tb = (it.ld_dst + 31) / 32, ntile = it.rows * tc * tb;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int64_t t = blk; t < ntile; t += nblk) {
            const int64_t a = t / (tc * tb), r = t - a * tc * tb;
            const int64_t b0 = (r / tc) * 32, c0 = (r % tc) * 32;
            const int64_t so = a * it.pb * it.pc, dof = a * (it.cols > 0 ? it.cols : it.pc * it.ld_dst);
            for (int i = ty; i < 32; i += 8) {
                const int64_t b = b0 + i, c = c0 + tx;
                tile[i][tx] = (b < it.pb && c < it.pc) ? prep_ld(it.src, so + b * it.pc + c, it.src_dt) : 0.f;
            }
            __syncthreads();
            for (int i = ty; i < 32; i += 8) {
                const int64_t c = c0 + i, b = b0 + tx;
                if (c < it.pc && b < it.ld_dst) prep_st(it.dst, dof + c * it.ld_dst + b, it.dst_dt, b < it.pb ? tile[tx][i] : 0.f);
            }
            __syncthreads();
        }
        return;
    }
    const int64_t total = it.rows * it.cols;
    for (int64_t i = (int64_t)blk * 256 + threadIdx.x; i < total; i += (int64_t)nblk * 256) {
        const int64_t r = i / it.cols, c = i - r * it.cols;
        prep_st(it.dst, r * it.ld_dst + c, it.dst_dt, it.op == 2 ? 0.f : prep_ld(it.src, r * it.ld_src + c, it.src_dt));
    }
}
extern "C" int segf_prep_grouped(int n, const SegfPrepItem* items, void* stream) {
    if (n <= 0) return 0;
    if (!items) return SEGF_ERR_SHAPE;
    for (int i = 0; i < n; ++i) {
        const SegfPrepItem& it = items[i];
        if (it.op < 0 || it.op > 2 || !it.dst || (it.op != 2 && !it.src)) return SEGF_ERR_SHAPE;
        if ((it.src_dt != SEGF_F32 && it.src_dt != SEGF_BF16) || (it.dst_dt != SEGF_F32 && it.dst_dt != SEGF_BF16)) return SEGF_ERR_DTYPE;
        if (it.rows < 0 || it.cols < 0 || it.ld_dst < 1 || (it.op == 0 && it.ld_src < 1)) return SEGF_ERR_SHAPE;
        if (it.op == 1 && (it.pb < 1 || it.pc < 1 || it.ld_dst < it.pb)) return SEGF_ERR_SHAPE;
    }
    for (int base = 0; base < n; base += PREP_MAX) {
        PrepGroup g;
        g.n = n - base < PREP_MAX ? n - base : PREP_MAX;
        int total = 0;
        for (int i = 0; i < g.n; ++i) {
            g.it[i] = items[base + i];
            const SegfPrepItem& it = g.it[i];
            const int64_t work = it.op == 1 ? it.rows * cdiv64(it.pc, 32) * cdiv64(it.ld_dst, 32) : cdiv64(it.rows * it.cols, 1024);
            g.first[i] = total;
            total += (int)imin64(work > 0 ? work : 1, 2048);
        }
        for (int i = g.n; i <= PREP_MAX; ++i) g.first[i] = total;
        hipLaunchKernelGGL(prep_group_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, g);
        SEGF_CHECK_LAUNCH();
    }
    return 0;
}

// ---- 2-D elementwise with leading dims ---------------------------------------------------------------
template <typename T, int MODE>   // MODE 0: y = x * scale[row / rpg]; MODE 1: y = a + b
__global__ void ew2d_kernel(const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb, T* __restrict__ y,
                            int64_t ldy, const float* __restrict__ scale, int64_t rows, int64_t cols, int64_t rpg, bool vec) {
    const int64_t nchunk = (cols + 7) / 8;
    const int64_t total = rows * nchunk;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto body = [&](const bool full) {      // full: every chunk complete and 16-byte aligned -> guard-free vector accesses
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
            const int64_t r = i / nchunk;
            const int c0 = (int)(i - r * nchunk) * 8;
            const int nv = full ? 8 : (int)(cols - c0 < 8 ? cols - c0 : 8);
            const bool vv = full ? true : vec;
            float va[8], vb[8];
            load8_guard<T>(a + r * lda + c0, nv, vv, va);
            if (MODE == 0) {
                const float s = scale[r / rpg];
#pragma unroll
                for (int j = 0; j < 8; ++j) va[j] *= s;
            } else {
                load8_guard<T>(b + r * ldb + c0, nv, vv, vb);
#pragma unroll
                for (int j = 0; j < 8; ++j) va[j] += vb[j];
            }
            store8_guard<T>(y + r * ldy + c0, nv, vv, va);
        }
    };
    if (vec && cols % 8 == 0) body(true); else body(false);
}

extern "C" int segf_scale_rows(int dt, const void* x, int64_t ldx, void* y, int64_t ldy, const float* scale,
                               int64_t rows, int64_t cols, int64_t rows_per_group, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    if (rows_per_group <= 0) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = rows * ((cols + 7) / 8);
    const int blocks = (int)imin64(cdiv64(total, 256), 4096);
    SEGF_DISPATCH_DT(dt, T, {
        const bool vec = vec_ok_host<T>(x, ldx) && vec_ok_host<T>(y, ldy);
        hipLaunchKernelGGL((ew2d_kernel<T, 0>), dim3(blocks), dim3(256), 0, st, (const T*)x, ldx, (const T*)nullptr, (int64_t)0,
                           (T*)y, ldy, scale, rows, cols, rows_per_group, vec);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

extern "C" int segf_add(int dt, const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy,
                        int64_t rows, int64_t cols, void* stream) {
    if (rows <= 0 || cols <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = rows * ((cols + 7) / 8);
    const int blocks = (int)imin64(cdiv64(total, 256), 4096);
    SEGF_DISPATCH_DT(dt, T, {
        const bool vec = vec_ok_host<T>(a, lda) && vec_ok_host<T>(b, ldb) && vec_ok_host<T>(y, ldy);
        hipLaunchKernelGGL((ew2d_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, (const T*)a, lda, (const T*)b, ldb, (T*)y,
                           ldy, (const float*)nullptr, rows, cols, (int64_t)1, vec);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- column sum ----------------------------------------------------------------------------------------
template <typename T> struct ColsumF {
    const T* x; int64_t ld; bool vec;
    struct Col {};
    __device__ void init(int, int, Col&) const {}
    __device__ void operator()(const Col&, int64_t r, int c0, int nv, float (&v)[1][8]) const { load8_guard<T>(x + r * ld + c0, nv, vec, v[0]); }
};

extern "C" int64_t segf_colsum_ws(int64_t rows, int64_t cols) { return cr_ws_floats(rows, (int)cols, 1); }

extern "C" int segf_colsum(int dt, const void* x, int64_t ldx, int64_t rows, int64_t cols, float* out, float* ws, void* stream) {
    if (cols <= 0) return 0;
    if (rows <= 0 || !ws) return rows <= 0 ? SEGF_ERR_SHAPE : SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    SEGF_DISPATCH_DT(dt, T, {
        ColsumF<T> f{(const T*)x, ldx, vec_ok_host<T>(x, ldx)};
        return colreduce_launch<1>(f, rows, (int)cols, ws, out, st);
    })
    return 0;
}


// ---- GELU(erf) forward / backward, flat (ConvNeXt Block.act, convnext.py:32,43: nn.GELU between the pointwise linears) -----
template <typename T, int MODE>   // MODE 0: y = gelu(u);  MODE 1: y = dy * gelu'(u)
__global__ void gelu_kernel(const T* __restrict__ u, const T* __restrict__ dy, T* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        // 8 channels stage by stage on packed pairs (common.h gelu_erf8): ~25 operations per element make this VALU-bound
        Raw8<T> ru = load8_raw<T>(u + i * 8), rg;
        if (MODE == 1) rg = load8_raw<T>(dy + i * 8);
        SEGF_LOADS_ISSUED();
        f32x2_t v[4], g[4];
        unpack8v<T>(ru, v);
        if (MODE == 1) unpack8v<T>(rg, g);
        gelu_erf8<MODE == 1>(v, g);
        store8v<T>(y + i * 8, v);
    }
}
extern "C" int segf_gelu(int dt, int mode, const void* u, const void* dy, void* y, int64_t n, void* stream) {
    if (n <= 0) return 0;
    if (n % 8 || ((uintptr_t)u % 16) || ((uintptr_t)y % 16) || (mode == 1 && (!dy || ((uintptr_t)dy % 16))) || mode < 0 || mode > 1)
        return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)imin64(cdiv64(n / 8, 256), 8192);
    SEGF_DISPATCH_DT(dt, T, {
        if (mode == 0) hipLaunchKernelGGL((gelu_kernel<T, 0>), dim3(blocks), dim3(256), 0, st, (const T*)u, (const T*)nullptr, (T*)y, n / 8);
        else hipLaunchKernelGGL((gelu_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, (const T*)u, (const T*)dy, (T*)y, n / 8);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- out[r] = sum_c a[r][c] * b[r][c] (+ extra_a[r] * extra_b[r]) : gradient of a per-row scale folded into a weight matrix
// (ConvNeXt layer scale gamma, convnext.py:34,46: W' = diag(gamma) W, b' = gamma o b) ----------------------------------
__global__ void __launch_bounds__(256) rowdot_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                      const float* __restrict__ ea, const float* __restrict__ eb, float* __restrict__ out,
                                                      int64_t rows, int64_t cols) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= rows) return;
    float s = 0.f;
    for (int64_t c = lane; c < cols; c += 64) s = fmaf(a[r * lda + c], b[r * ldb + c], s);
    s = wave_sum_all(s);
    if (lane == 0) out[r] = s + (ea ? ea[r] * eb[r] : 0.f);
}
extern "C" int segf_rowdot(const float* a, int64_t lda, const float* b, int64_t ldb, const float* extra_a, const float* extra_b,
                           float* out, int64_t rows, int64_t cols, void* stream) {
    if (rows <= 0) return 0;
    if (cols < 0 || (extra_a && !extra_b)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, st, a, lda, b, ldb, extra_a, extra_b, out, rows, cols);
    SEGF_CHECK_LAUNCH();
    return 0;
}


// ---- stream-ordering helpers for the graphed train step -------------------------------------------------------------------
// An EXTERNAL event record inside a stream capture becomes an event-record node of the hipGraph: at every replay it fires when
// the nodes captured before it have finished, and a stream outside the graph can wait on it (the per-bucket "gradients are
// final" signal of the data-parallel exchange, segmentation_factory_amd/graph.py).  PyTorch's torch.cuda.Event(external=True)
// refuses to do this on ROCm, HIP itself (7.x) provides it.
extern "C" int segf_event_create(void** event) {
    hipEvent_t e;
    const hipError_t r = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (r != hipSuccess) return (int)r;
    *event = (void*)e;
    return 0;
}
extern "C" int segf_event_destroy(void* event) { return (int)hipEventDestroy((hipEvent_t)event); }
extern "C" int segf_event_record(void* event, void* stream, int external) {
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t ev = (hipEvent_t)event;
    if (!external) return (int)hipEventRecord(ev, st);
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    hipError_t r = hipStreamGetCaptureInfo_v2(st, &status, nullptr, &graph, &deps, &ndeps);
    if (r != hipSuccess) return (int)r;
    if (status != hipStreamCaptureStatusActive) return SEGF_ERR_SHAPE;       // an external record only exists inside a capture
    r = hipEventRecordWithFlags(ev, st, hipEventRecordExternal);
    if (r == hipSuccess) return 0;
    (void)hipGetLastError();
    // the runtime refused the flag form: add the event-record node by hand behind the capture's current frontier and make it
    // the new frontier (same graph, same semantics)
    hipGraphNode_t node = nullptr;
    r = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev);
    if (r != hipSuccess) return (int)r;
    r = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
    return (int)r;
}
extern "C" int segf_stream_wait_event(void* stream, void* event) {
    return (int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0);
}


// ---- small device-side plumbing that keeps the captured train step free of framework kernels --------------------------------
__global__ void zero_fill_kernel(uint32_t* __restrict__ p, int64_t nwords, uint8_t* __restrict__ tail, int ntail) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
extern "C" int segf_zero(void* p, int64_t nbytes, void* stream) {
    if (nbytes <= 0) return 0;
    if ((uintptr_t)p % 4) return SEGF_ERR_SHAPE;
    const int64_t nwords = nbytes / 4;
    const int blocks = (int)imin64(cdiv64(nwords > 0 ? nwords : 1, 256), 4096);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t*)p, nwords,
                       (uint8_t*)p + nwords * 4, (int)(nbytes - nwords * 4));
    SEGF_CHECK_LAUNCH();
    return 0;
}
// Metrics.update (util/metrics.py:24-27): `self.hist += bincount(...)` with hist fp32 and the batch counts int64 -- each count is
// converted to fp32 (round to nearest even, torch's type promotion of float32 += int64) and added; quirk Q5: exact below 2^24 per
// cell.  The counts are cleared for the next batch in the same pass, so a captured evaluation step needs no separate fill.
__global__ void hist_accum_kernel(float* __restrict__ hist, int64_t* __restrict__ counts, int64_t n, int clear) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hist[i] += (float)counts[i];
    if (clear) counts[i] = 0;
}
extern "C" int segf_hist_accum(float* hist, int64_t* counts, int64_t n, int clear, void* stream) {
    if (n <= 0) return 0;
    if (!hist || !counts) return SEGF_ERR_SHAPE;
    hipLaunchKernelGGL(hist_accum_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream, hist, counts, n, clear);
    SEGF_CHECK_LAUNCH();
    return 0;
}
// TEST HOOK: one wave that holds its stream for `us` microseconds (s_memrealtime ticks at 100 MHz), capped at 200 ms so that every
// launch ends (tests/test_model_gpu.py: the data-parallel ordering test delays a gradient behind it).
__global__ void debug_spin_kernel(int64_t ticks) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (int guard = 0; guard < (1 << 24); ++guard) {
        if ((int64_t)(__builtin_amdgcn_s_memrealtime() - t0) >= ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}
extern "C" int segf_debug_spin(int64_t us, void* stream) {
    if (us < 0 || us > 200000) return SEGF_ERR_SHAPE;
    hipLaunchKernelGGL(debug_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, us * 100);
    SEGF_CHECK_LAUNCH();
    return 0;
}
__global__ void add_i64_kernel(int64_t* p, int64_t v) { if (threadIdx.x == 0) *p += v; }
extern "C" int segf_add_i64(int64_t* p, int64_t v, void* stream) {
    hipLaunchKernelGGL(add_i64_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, v);
    SEGF_CHECK_LAUNCH();
    return 0;
}

// Keep / drop scales of the stochastic layers (DropPath, models/layers/drop_path.py:18-25: x / kp * floor(kp + U); Dropout2d,
// heads/segformer.py:40: whole channels zeroed with p = 0.1, survivors scaled by 1 / 0.9):
//   out[i] = U_i < kp[i / row_len] ? 1 / kp[i / row_len] : 0,   U_i = uniform [0, 1) from a counter-based generator.
// state[0] = seed, state[1] = launch counter (advanced by the kernel itself, so a replayed hipGraph draws fresh numbers every
// step without any host involvement).
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void bernoulli_scale_kernel(uint64_t* __restrict__ state, const float* __restrict__ kp, int64_t n, int64_t row_len,
                                       float* __restrict__ out) {
    const uint64_t seed = state[0], ctr = state[1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t r = splitmix64(splitmix64(seed ^ (ctr * 0xD1B54A32D192ED03ull)) + (uint64_t)i);
        const float u = (float)(r >> 40) * (1.f / 16777216.f);
        const float k = kp[i / row_len];
        out[i] = u < k ? 1.f / k : 0.f;
    }
    __syncthreads();
    // the LAST workgroup to finish advances the counter (every workgroup has read it above): single-block launches in practice
    if (gridDim.x == 1 && threadIdx.x == 0) state[1] = ctr + 1;
}
extern "C" int segf_bernoulli_scale(uint64_t* state, const float* keep_prob, int64_t n, int64_t row_len, float* out, void* stream) {
    if (n <= 0) return 0;
    if (row_len <= 0 || n > (1 << 20)) return SEGF_ERR_SHAPE;       // one workgroup walks the whole (small) table
    hipLaunchKernelGGL(bernoulli_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, state, keep_prob, n, row_len, out);
    SEGF_CHECK_LAUNCH();
    return 0;
}
