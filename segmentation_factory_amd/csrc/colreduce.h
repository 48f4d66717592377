// Generic deterministic column reduction: out[o][c] = sum_rows f(row, c)[o].
// Used for bias gradients, BatchNorm statistics / backward sums, depthwise-conv weight gradients.
// Stage 1: grid (row blocks, channel slabs) -> partial[blk][NOUT][C]; stage 2: colreduce_finalize.
#pragma once
#include "common.h"

#define CR_THREADS 256
#define CR_MAX_BLOCKS 512

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
    p.rl = CR_THREADS / p.ch;
    p.slabs = (nchunk + p.ch - 1) / p.ch;
    int64_t want = cdiv64(rows, (int64_t)p.rl * 16);
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
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
    const int64_t r1 = r0 + rows_per_blk < rows ? r0 + rows_per_blk : rows;
    if (active) {
        for (int64_t r = r0 + ty; r < r1; r += rl) {
            float v[NOUT][8];
            f(r, c0, nvalid, v);
#pragma unroll
            for (int o = 0; o < NOUT; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][j] += v[o][j];
        }
    }
    // reduce over the rl row lanes, one output at a time (8 KB of LDS)
    for (int o = 0; o < NOUT; ++o) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = active ? acc[o][j] : 0.f;
        __syncthreads();
        if (ty == 0 && nvalid > 0) {
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

// out[i] = sum_b partial[b][i], i < n  (fixed order -> bitwise reproducible)
static __global__ void colreduce_finalize_kernel(const float* __restrict__ partial, int nblk, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * n + i];
    out[i] = s;
}

template <int NOUT, typename F>
static inline int colreduce_launch(F f, int64_t rows, int C, float* ws, float* out, hipStream_t st) {
    CRPlan p = cr_plan(rows, C);
    hipLaunchKernelGGL((colreduce_kernel<NOUT, F>), dim3(p.nblk, p.slabs), dim3(CR_THREADS), 0, st, f, rows, C, p.ch,
                       p.rl, p.rows_per_blk, ws);
    SEGF_CHECK_LAUNCH();
    const int64_t n = (int64_t)NOUT * C;
    hipLaunchKernelGGL(colreduce_finalize_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, ws, p.nblk, n, out);
    SEGF_CHECK_LAUNCH();
    return 0;
}
