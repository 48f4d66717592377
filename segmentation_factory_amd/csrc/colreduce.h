// Generic deterministic column reduction: out[o][c] = sum_rows f(row, c)[o].
// Functor protocol: `typename F::Col` = per-thread column state (per-channel parameters hoisted out of the row loop),
// `f.init(c0, nvalid, col)` once, then `f(col, row, c0, nvalid, v)` per row.
// Used for bias gradients, BatchNorm statistics / backward sums, depthwise-conv weight gradients.
// Stage 1: grid (row blocks, channel slabs) -> partial[blk][NOUT][C]; stage 2: colreduce_finalize.
#pragma once
#include <stdlib.h>
#include "common.h"

#define CR_THREADS 256
#define CR_MAX_BLOCKS 2048

// guarded 8-wide load: vector path when the chunk is complete and the row is 16-byte aligned
template <typename T>
__device__ __forceinline__ void load8_guard(const T* p, int nvalid, bool vec_ok, float (&v)[8]) {
    if (nvalid >= 8 && vec_ok) {
        load8<T>(p, v);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = j < nvalid ? ldf<T>(p + j) : 0.f;
    }
}
template <typename T>
__device__ __forceinline__ void store8_guard(T* p, int nvalid, bool vec_ok, const float (&v)[8]) {
    if (nvalid >= 8 && vec_ok) {
        store8<T>(p, v);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nvalid) stf<T>(p + j, v[j]);
    }
}
template <typename T> static inline bool vec_ok_host(const void* p, int64_t ld) {
    return (((uintptr_t)p) % 16 == 0) && ((ld * (int64_t)sizeof(T)) % 16 == 0);
}

struct CRPlan {
    int ch;        // chunk lanes per block (threads along channels)
    int rl;        // row lanes per block
    int slabs;     // grid.y
    int nblk;      // grid.x
    int64_t rows_per_blk;
};
static inline CRPlan cr_plan(int64_t rows, int C) {
    CRPlan p;
    const int nchunk = (C + 7) / 8;
    p.ch = nchunk < CR_THREADS ? nchunk : CR_THREADS;
    // a chunk count that does not divide the block (96 chunks of C = 768 -> 192 of 256 threads busy): take channel slabs of a
    // power-of-two width instead when one divides the chunk count
    if (CR_THREADS % p.ch) {
        for (int c2 = 64; c2 >= 8; c2 >>= 1)
            if (nchunk % c2 == 0) { p.ch = c2; break; }
    }
    p.rl = CR_THREADS / p.ch;
    p.slabs = (nchunk + p.ch - 1) / p.ch;
    int64_t want = cdiv64(rows, (int64_t)p.rl * 32);
    int cap = CR_MAX_BLOCKS / p.slabs;
    if (cap < 1) cap = 1;
    p.nblk = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    p.rows_per_blk = cdiv64(rows, p.nblk);
    return p;
}
static inline int64_t cr_ws_floats(int64_t rows, int C, int nout) {
    CRPlan p = cr_plan(rows, C);
    return (int64_t)p.nblk * nout * C;
}

// Block tail shared by the column reductions: sum acc[o][8] over the rl row lanes of the block (LDS, fixed order) and
// write partial[blockIdx.x][o][c0..c0+nvalid).
template <int NOUT>
__device__ __forceinline__ void colreduce_block_tail(float (&acc)[NOUT][8], bool active, bool writer, int tx, int ch, int rl,
                                                     int c0, int nvalid, int C, float* __restrict__ partial, float* red) {
    for (int o = 0; o < NOUT; ++o) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = active ? acc[o][j] : 0.f;
        __syncthreads();
        if (writer) {
            float s[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = 0.f;
            for (int y = 0; y < rl; ++y)
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += red[(y * ch + tx) * 8 + j];
            float* dst = partial + ((int64_t)blockIdx.x * NOUT + o) * C + c0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nvalid) dst[j] = s[j];
        }
    }
}

template <int NOUT, typename F>
__global__ void __launch_bounds__(CR_THREADS) colreduce_kernel(F f, int64_t rows, int C, int ch, int rl,
                                                                int64_t rows_per_blk, float* __restrict__ partial) {
    __shared__ float red[CR_THREADS * 8];
    const int tx = threadIdx.x % ch, ty = threadIdx.x / ch;
    const int chunk = blockIdx.y * ch + tx;
    const int c0 = chunk * 8;
    const int nvalid = c0 < C ? (C - c0 < 8 ? C - c0 : 8) : 0;
    const bool active = ty < rl && nvalid > 0;
    float acc[NOUT][8];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
    // blockIdx.z = batch index of the batched form: rows [z*rows, (z+1)*rows), partials [z][blk][NOUT][C]
    const int64_t rbase = (int64_t)blockIdx.z * rows;
    const int64_t r0 = rbase + (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r1 = r0 + rows_per_blk < rbase + rows ? r0 + rows_per_blk : rbase + rows;
    partial += (int64_t)blockIdx.z * gridDim.x * NOUT * C;
    if (active) {
        typename F::Col col;
        f.init(c0, nvalid, col);
        // two rows per iteration: both rows' loads are issued before either is consumed (memory-level parallelism).
        // The row loop exists twice, under `full chunk and 16-byte aligned rows` and under its negation: inside the first copy
        // the functor's guarded loads fold to plain vector loads (a guard left inside the loop is a divergent branch per load,
        // and each such branch ends in s_waitcnt vmcnt(0): the loads of one iteration would run one after the other).
        auto row_loop = [&](const int nv, const F& ff) {
            int64_t r = r0 + ty;
            for (; r + rl < r1; r += 2 * rl) {
                float v[NOUT][8], w[NOUT][8];
                ff(col, r, c0, nv, v);
                ff(col, r + rl, c0, nv, w);
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[o][j] += v[o][j] + w[o][j];
            }
            if (r < r1) {
                float v[NOUT][8];
                ff(col, r, c0, nv, v);
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[o][j] += v[o][j];
            }
        };
        if (nvalid >= 8 && f.vec) {
            F ff = f;
            ff.vec = true;
            row_loop(8, ff);
        } else {
            row_loop(nvalid, f);
        }
    }
    colreduce_block_tail<NOUT>(acc, active, ty == 0 && nvalid > 0, tx, ch, rl, c0, nvalid, C, partial, red);
}

// out[i] = sum_b partial[b][i], i < n.  16 outputs x 16 block-slices per workgroup: slice s adds b = s, s+16, ...
// (independent loads in flight), then the 16 slice sums are added in fixed order -> bitwise reproducible.
#define CRF_OUT 16
#define CRF_SL 16
static __global__ void __launch_bounds__(CRF_OUT * CRF_SL)
colreduce_finalize_kernel(const float* __restrict__ partial, int nblk, int64_t n, float* __restrict__ out) {
    __shared__ float red[CRF_SL][CRF_OUT + 1];
    const int o = threadIdx.x % CRF_OUT, sl = threadIdx.x / CRF_OUT;
    const int64_t i = (int64_t)blockIdx.x * CRF_OUT + o;
    partial += (int64_t)blockIdx.y * nblk * n;       // batched form: blockIdx.y = batch
    out += (int64_t)blockIdx.y * n;
    float acc = 0.f;
    if (i < n) {
#pragma unroll 4
        for (int b = sl; b < nblk; b += CRF_SL) acc += partial[(int64_t)b * n + i];
    }
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CRF_SL; ++k) t += red[k][o];
        out[i] = t;
    }
}
// the finalize steps of SEVERAL two-stage reductions in one launch (the dgamma / dbeta sums of the LayerNorm backward passes of a captured
// train step: 30 launches of ~6 us each at the reference's default batch): member i owns blocks [start[i], start[i + 1]) and is summed by
// the same code, in the same order, as colreduce_finalize_kernel
#define CRF_GROUP_MAX 32
struct ColreduceGroup {
    int n;
    unsigned start[CRF_GROUP_MAX + 1];
    const float* partial[CRF_GROUP_MAX]; float* out[CRF_GROUP_MAX];
    int nblk[CRF_GROUP_MAX]; int64_t len[CRF_GROUP_MAX];
    int scatter_c[CRF_GROUP_MAX];      // > 0: sums [10][C] of a depthwise 3x3 weight gradient -> out = dw[C][9] followed by db[C]
};
static __global__ void __launch_bounds__(CRF_OUT * CRF_SL) colreduce_finalize_group_kernel(const ColreduceGroup g) {
    __shared__ float red[CRF_SL][CRF_OUT + 1];
    int m = 0;
#pragma unroll 1
    while (m + 1 < g.n && blockIdx.x >= g.start[m + 1]) ++m;
    const float* __restrict__ partial = g.partial[m];
    const int nblk = g.nblk[m];
    const int64_t n = g.len[m];
    const int o = threadIdx.x % CRF_OUT, sl = threadIdx.x / CRF_OUT;
    const int64_t i = (int64_t)(blockIdx.x - g.start[m]) * CRF_OUT + o;
    float acc = 0.f;
    if (i < n) {
#pragma unroll 4
        for (int b = sl; b < nblk; b += CRF_SL) acc += partial[(int64_t)b * n + i];
    }
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CRF_SL; ++k) t += red[k][o];
        const int sc = g.scatter_c[m];
        if (sc > 0) {                          // what dw_scatter_kernel (conv.hip) does behind the single finalize
            const int kk = (int)(i / sc), c = (int)(i - (int64_t)kk * sc);
            g.out[m][kk < 9 ? c * 9 + kk : 9 * sc + c] = t;
        } else {
            g.out[m][i] = t;
        }
    }
}
// the same sums in the same order for LARGE outputs (the classifier's riding weight gradient: 160 x 768 outputs x ~85 partial slabs = 42 MB):
// a thread takes FOUR adjacent outputs of its slab residue class, so a 16-lane group reads 256 contiguous bytes of a slab instead of 64
// (41 -> 13 us at cfg2; the 16 residue-class sums of an output still meet in LDS and are added in residue order: bitwise the kernel above)
static __global__ void __launch_bounds__(CRF_OUT * CRF_SL)
colreduce_finalize_wide_kernel(const float* __restrict__ partial, int nblk, int64_t n, float* __restrict__ out) {
    __shared__ float4 red[CRF_SL][CRF_OUT + 1];
    const int o = threadIdx.x % CRF_OUT, sl = threadIdx.x / CRF_OUT;
    const int64_t i = ((int64_t)blockIdx.x * CRF_OUT + o) * 4;
    partial += (int64_t)blockIdx.y * nblk * n;
    out += (int64_t)blockIdx.y * n;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
#pragma unroll 4
        for (int b = sl; b < nblk; b += CRF_SL) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)b * n + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < n) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < CRF_SL; ++k) { const float4 v = red[k][o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(out + i) = t;
    }
}
static inline void colreduce_finalize_launch(const float* partial, int nblk, int64_t n, float* out, hipStream_t st, int nbatch = 1) {
    if (n >= 16384 && n % 4 == 0 && !((uintptr_t)partial & 15) && !((uintptr_t)out & 15) && !POL(no_wide_finalize)) {
        hipLaunchKernelGGL(colreduce_finalize_wide_kernel, dim3((unsigned)cdiv64(n, 4 * CRF_OUT), nbatch), dim3(CRF_OUT * CRF_SL), 0, st,
                           partial, nblk, n, out);
        return;
    }
    hipLaunchKernelGGL(colreduce_finalize_kernel, dim3((unsigned)cdiv64(n, CRF_OUT), nbatch), dim3(CRF_OUT * CRF_SL), 0, st, partial,
                       nblk, n, out);
}

template <int NOUT, typename F>
static inline int colreduce_launch(F f, int64_t rows, int C, float* ws, float* out, hipStream_t st) {
    CRPlan p = cr_plan(rows, C);
    hipLaunchKernelGGL((colreduce_kernel<NOUT, F>), dim3(p.nblk, p.slabs), dim3(CR_THREADS), 0, st, f, rows, C, p.ch,
                       p.rl, p.rows_per_blk, ws);
    SEGF_CHECK_LAUNCH();
    const int64_t n = (int64_t)NOUT * C;
    colreduce_finalize_launch(ws, p.nblk, n, out, st);
    SEGF_CHECK_LAUNCH();
    return 0;
}

// batched form: nbatch independent reductions over `rows` consecutive rows each -> out[nbatch][NOUT][C];
// ws >= nbatch * cr_ws_floats(rows, C, NOUT)
template <int NOUT, typename F>
static inline int colreduce_launch_batched(F f, int64_t rows, int nbatch, int C, float* ws, float* out, hipStream_t st) {
    CRPlan p = cr_plan(rows, C);
    hipLaunchKernelGGL((colreduce_kernel<NOUT, F>), dim3(p.nblk, p.slabs, nbatch), dim3(CR_THREADS), 0, st, f, rows, C, p.ch,
                       p.rl, p.rows_per_blk, ws);
    SEGF_CHECK_LAUNCH();
    colreduce_finalize_launch(ws, p.nblk, (int64_t)NOUT * C, out, st, nbatch);
    SEGF_CHECK_LAUNCH();
    return 0;
}
