// LayerNorm (rows = tokens, wave-shuffle statistics) and BatchNorm2d-on-NHWC (column statistics).
// HBM-bound streaming kernels: each activation is read once per pass with 16-B lane accesses.
//   LayerNorm: reference nn.LayerNorm in models/backbones/mit.py:107,129,136-140 (eps 1e-5) and the
//              channels-first LayerNorm of convnext.py:8-23 / convnextv2.py:40-66 (eps 1e-6).
//   BatchNorm: ConvModule of heads/segformer.py:21-29, layers/conv_module.py:4-9, mobilenetv2.py:5-11.
#include "colreduce.h"

// ---- LayerNorm -----------------------------------------------------------------------------------------
// A row is spread over LPR = 2^lpr_log2 lanes (8 elements per lane per step, VPT steps); a wave handles
// 64/LPR rows at once.  Two-pass statistics in registers (mean, then centred variance), fp32.
// Patch-major token order of a spatial-reduction convolution's im2col matrix (mit.py:47, k = s = sr): token (b, y, x) of a [B, H, W] map
// is row ((b Ho + y / sr) Wo + x / sr) of the matrix and chunk (y % sr, x % sr) of that row -- a PERMUTATION of the token rows.  With W
// and sr powers of two (H a multiple of sr): t = r >> lw = b H + y, x = r & (W - 1).
__device__ __forceinline__ int64_t patch_row(int64_t r, int lw, int ls) {
    const int64_t t = r >> lw, x = r & (((int64_t)1 << lw) - 1), sm = ((int64_t)1 << ls) - 1;
    return ((((t >> ls) << (lw - ls)) + (x >> ls)) << (2 * ls)) + ((t & sm) << ls) + (x & sm);
}
template <typename T, int VPT>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, T* __restrict__ y,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      int64_t rows, int C, float eps, int lpr_log2,
                                                      T* __restrict__ y2 /*nullable: the same rows in patch-major order*/, int lw, int ls) {
    const int lpr = 1 << lpr_log2;
    const int lane = threadIdx.x & 63;
    const int sub = lane & (lpr - 1), rin = lane >> lpr_log2;
    const int rows_per_wave = 64 >> lpr_log2;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float g[VPT][8], b[VPT][8];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c0 = (sub + i * lpr) * 8;
        if (c0 < C) { load8f(gamma + c0, g[i]); load8f(beta + c0, b[i]); }
    }
    const float invC = 1.f / (float)C;
    for (int64_t rbase = wave_global * rows_per_wave; rbase < rows; rbase += nwaves * rows_per_wave) {
        const int64_t r = rbase + rin;
        const bool rv = r < rows;
        float v[VPT][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (sub + i * lpr) * 8;
            if (rv && c0 < C) {
                load8<T>(x + r * C + c0, v[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[i][j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
            }
        }
        s = wave_sum(s, lpr);
        const float mu = s * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (sub + i * lpr) * 8;
            if (c0 < C) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mu; q += d * d; }
            }
        }
        q = wave_sum(q, lpr);
        const float rs = rsqrtf(q * invC + eps);
        if (rv) {
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int c0 = (sub + i * lpr) * 8;
                if (c0 < C) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mu) * rs * g[i][j] + b[i][j];
                    store8<T>(y + r * C + c0, o);
                    if (y2) store8<T>(y2 + patch_row(r, lw, ls) * C + c0, o);
                }
            }
            if (sub == 0) { mean_out[r] = mu; rstd_out[r] = rs; }
        }
    }
}

#define LN_BWD_MAX_BLOCKS 1024      // 4 workgroups per CU (one per CU left the HBM pipe half empty: 2.7 TB/s at 2M x 32)
template <typename T, int VPT>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                      const T* __restrict__ dy2 /*nullable: second consumer's gradient, added to dy*/,
                                                      const T* __restrict__ dres /*nullable: residual-path gradient, added to dx*/,
                                                      const float* __restrict__ gamma, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, T* __restrict__ dx,
                                                      float* __restrict__ partial /*[grid][2][C]*/, int64_t rows, int C,
                                                      int lpr_log2, const float* __restrict__ rsc /*nullable: per-row-group scale of a second output*/,
                                                      float inv_rpg, T* __restrict__ dxs /*dxs = (T) dx * rsc[row / rpg]: the DropPath backward of dx's consumer*/,
                                                      int lw /*>= 0: dy2 is stored in patch-major row order (patch_row)*/, int ls) {
    extern __shared__ float lds[];   // [4 waves][2][C]
    const int lpr = 1 << lpr_log2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane & (lpr - 1), rin = lane >> lpr_log2;
    const int rows_per_wave = 64 >> lpr_log2;
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wave;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    float g[VPT][8], dg[VPT][8], db[VPT][8];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c0 = (sub + i * lpr) * 8;
        if (c0 < C) load8f(gamma + c0, g[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }
    }
    const float invC = 1.f / (float)C;
    for (int64_t rbase = wave_global * rows_per_wave; rbase < rows; rbase += nwaves * rows_per_wave) {
        const int64_t r = rbase + rin;
        const bool rv = r < rows;
        const int64_t rc = rv ? r : rows - 1;          // rows past the end: clamped loads, contributions masked below
        const float mu = mean[rc], rs = rv ? rstd[rc] : 0.f;
        float xh[VPT][8], gy[VPT][8];
        float s1 = 0.f, s2 = 0.f;
        Raw8<T> rx[VPT], rd[VPT], rd2[VPT], rr[VPT];
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (sub + i * lpr) * 8, cc = c0 < C ? c0 : 0;
            rx[i] = load8_raw<T>(x + rc * C + cc);
            rd[i] = load8_raw<T>(dy + rc * C + cc);
            if (dy2) rd2[i] = load8_raw<T>(dy2 + (lw >= 0 ? patch_row(rc, lw, ls) : rc) * C + cc);          // workgroup-uniform branches: the loads stay batched
            if (dres) rr[i] = load8_raw<T>(dres + rc * C + cc);
        }
        SEGF_LOADS_ISSUED();
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (sub + i * lpr) * 8;
            if (rv && c0 < C) {
                float xv[8], dv[8];
                unpack8(rx[i], xv);
                unpack8(rd[i], dv);
                if (dy2) {
                    float d2[8];
                    unpack8(rd2[i], d2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) dv[j] += d2[j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[i][j] = (xv[j] - mu) * rs;
                    dg[i][j] += dv[j] * xh[i][j];
                    db[i][j] += dv[j];
                    gy[i][j] = dv[j] * g[i][j];
                    s1 += gy[i][j];
                    s2 += gy[i][j] * xh[i][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { xh[i][j] = 0.f; gy[i][j] = 0.f; }
            }
        }
        s1 = wave_sum(s1, lpr) * invC;
        s2 = wave_sum(s2, lpr) * invC;
        if (rv) {
#pragma unroll
            for (int i = 0; i < VPT; ++i) {
                const int c0 = (sub + i * lpr) * 8;
                if (c0 < C) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = rs * (gy[i][j] - s1 - xh[i][j] * s2);
                    if (dres) {
                        float rv8[8];
                        unpack8(rr[i], rv8);
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] += rv8[j];
                    }
                    store8<T>(dx + r * C + c0, o);
                    if (dxs) {
                        // what segf_scale_rows computes from the STORED dx (DropPath backward, drop_path.py:18-25): the value rounded to T first.
                        // row / rpg through the reciprocal: exact below 2^22 rows (host-checked)
                        const float sc = rsc[(int)(((float)r + 0.5f) * inv_rpg)];
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = round_to<T>(o[j]) * sc;
                        store8<T>(dxs + r * C + c0, o);
                    }
                }
            }
        }
    }
    // reduce dgamma / dbeta: across the row lanes of the wave (xor offsets >= lpr), then across waves via LDS
#pragma unroll
    for (int i = 0; i < VPT; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = dg[i][j], c = db[i][j];
            for (int o = 32; o >= lpr; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
            dg[i][j] = a; db[i][j] = c;
        }
    if (rin == 0) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (sub + i * lpr) * 8;
            if (c0 < C) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    lds[(wave * 2 + 0) * C + c0 + j] = dg[i][j];
                    lds[(wave * 2 + 1) * C + c0 + j] = db[i][j];
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        float s = 0.f;
        for (int w = 0; w < 4; ++w) s += lds[(w * 2 + which) * C + c];
        partial[((int64_t)blockIdx.x * 2 + which) * C + c] = s;
    }
}

template <typename T>
static void ln_fwd_launch(int vpt, int blocks, hipStream_t st, const T* x, const float* gamma, const float* beta, T* y,
                          float* mean, float* rstd, int64_t rows, int C, float eps, int lpr_log2, T* y2, int lw, int ls) {
#define LN_F(V) hipLaunchKernelGGL((ln_fwd_kernel<T, V>), dim3(blocks), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, rows, C, eps, lpr_log2, y2, lw, ls)
    if (vpt == 1) LN_F(1); else if (vpt == 2) LN_F(2); else if (vpt == 3) LN_F(3); else if (vpt == 4) LN_F(4); else if (vpt == 5) LN_F(5); else LN_F(6);
#undef LN_F
}
template <typename T>
static void ln_bwd_launch(int vpt, int blocks, size_t shm, hipStream_t st, const T* x, const T* dy, const T* dy2, const T* dres, const float* gamma,
                          const float* mean, const float* rstd, T* dx, float* ws, int64_t rows, int C, int lpr_log2, const float* rsc,
                          float inv_rpg, T* dxs, int lw, int ls) {
#define LN_B(V) hipLaunchKernelGGL((ln_bwd_kernel<T, V>), dim3(blocks), dim3(256), shm, st, x, dy, dy2, dres, gamma, mean, rstd, dx, ws, rows, C, lpr_log2, rsc, inv_rpg, dxs, lw, ls)
    // C in (2048, 3072] (convnextv2_huge's 2816-wide last stage): six chunks per lane and 96 KB of dynamic LDS -- above the 64 KB
    // a kernel gets by default, so the limit is raised on the function first (160 KB per CU on gfx950)
#define LN_B_BIG(V) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ln_bwd_kernel<T, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); LN_B(V); } while (0)
    if (vpt == 1) LN_B(1); else if (vpt == 2) LN_B(2); else if (vpt == 3) LN_B(3); else if (vpt == 4) LN_B(4); else if (vpt == 5) LN_B_BIG(5); else LN_B_BIG(6);
#undef LN_B_BIG
#undef LN_B
}

static inline int ln_plan(int C, int& lpr_log2) {
    const int nchunk = C / 8;
    lpr_log2 = 0;
    while ((1 << lpr_log2) < nchunk && lpr_log2 < 6) ++lpr_log2;
    return (nchunk + (1 << lpr_log2) - 1) >> lpr_log2;   // VPT
}

extern "C" int segf_layernorm_fwd_patch(int dt, int64_t rows, int C, const void* x, const float* gamma, const float* beta, float eps, void* y,
                                        float* mean, float* rstd, void* y2, int log2_w, int log2_sr, void* stream);
static int patch_geom_ok(int64_t rows, int log2_w, int log2_sr) {
    return log2_sr >= 1 && log2_w >= log2_sr && log2_w <= 20 && ((rows >> log2_w) << log2_w) == rows && ((rows >> log2_w) & (((int64_t)1 << log2_sr) - 1)) == 0;
}
extern "C" int segf_layernorm_fwd(int dt, int64_t rows, int C, const void* x, const float* gamma, const float* beta,
                                  float eps, void* y, float* mean, float* rstd, void* stream) {
    return segf_layernorm_fwd_patch(dt, rows, C, x, gamma, beta, eps, y, mean, rstd, nullptr, 0, 0, stream);
}
// ... with a second copy y2 of the output in the PATCH-MAJOR row order of a k = s = sr convolution's im2col matrix (mit.py:47): y2 viewed as
// [rows / sr^2][sr^2 C] IS that matrix (no segf_im2col pass over y).  The map width W = 2^log2_w and sr = 2^log2_sr; rows % (W sr) == 0.
extern "C" int segf_layernorm_fwd_patch(int dt, int64_t rows, int C, const void* x, const float* gamma, const float* beta, float eps, void* y,
                                        float* mean, float* rstd, void* y2, int log2_w, int log2_sr, void* stream) {
    if (rows <= 0) return 0;
    if (y2 && (!patch_geom_ok(rows, log2_w, log2_sr) || ((uintptr_t)y2 % 16))) return SEGF_ERR_SHAPE;
    if (C <= 0 || C % 8 != 0 || C > 3072) return SEGF_ERR_SHAPE;
    if (((uintptr_t)x % 16) || ((uintptr_t)y % 16) || ((uintptr_t)gamma % 16) || ((uintptr_t)beta % 16)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    int lpr_log2;
    const int vpt = ln_plan(C, lpr_log2);
    const int rows_per_block = 4 * (64 >> lpr_log2);
    const int blocks = (int)imin64(cdiv64(rows, rows_per_block), 4096);
    SEGF_DISPATCH_DT(dt, T, { ln_fwd_launch<T>(vpt, blocks, st, (const T*)x, gamma, beta, (T*)y, mean, rstd, rows, C, eps, lpr_log2, (T*)y2, log2_w, log2_sr); })
    SEGF_CHECK_LAUNCH();
    return 0;
}

static inline int ln_bwd_blocks(int64_t rows, int C) {
    int lpr_log2;
    ln_plan(C, lpr_log2);
    const int rows_per_block = 4 * (64 >> lpr_log2);
    const int64_t b4 = cdiv64(rows, (int64_t)rows_per_block * 4);
    // few rows (the stage-3 / 4 maps at the reference's default batch of 4, train_gpu.py:71): one row group per wave instead of four --
    // the launch is a chain of dependent load rounds on a fraction of the chip, and four rounds cost 10 us where one costs 6
    if (b4 < 256) return (int)imin64(cdiv64(rows, (int64_t)rows_per_block), LN_BWD_MAX_BLOCKS);
    return (int)imin64(b4, LN_BWD_MAX_BLOCKS);
}
extern "C" int64_t segf_layernorm_bwd_ws(int64_t rows, int C) { return (int64_t)ln_bwd_blocks(rows, C) * 2 * C; }

extern "C" int segf_layernorm_bwd_fused(int dt, int64_t rows, int C, const void* x, const void* dy, const void* dy2, const void* dres,
                                        const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                                        float* dbeta, float* ws, void* stream);
extern "C" int segf_layernorm_bwd_scaled(int dt, int64_t rows, int C, const void* x, const void* dy, const void* dy2, const void* dres,
                                         const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                                         float* dbeta, float* ws, const float* rscale, int64_t rows_per_group, void* dxs, void* stream);
extern "C" int segf_layernorm_bwd_patch(int dt, int64_t rows, int C, const void* x, const void* dy, const void* dy2, const void* dres,
                                        const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                                        float* dbeta, float* ws, const float* rscale, int64_t rows_per_group, void* dxs, int log2_w,
                                        int log2_sr, void* stream);
extern "C" int segf_layernorm_bwd(int dt, int64_t rows, int C, const void* x, const void* dy, const float* gamma,
                                  const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                                  float* ws, void* stream) {
    return segf_layernorm_bwd_fused(dt, rows, C, x, dy, nullptr, nullptr, gamma, mean, rstd, dx, dgamma, dbeta, ws, stream);
}
extern "C" int segf_layernorm_bwd_fused(int dt, int64_t rows, int C, const void* x, const void* dy, const void* dy2, const void* dres,
                                        const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                                        float* dbeta, float* ws, void* stream) {
    return segf_layernorm_bwd_scaled(dt, rows, C, x, dy, dy2, dres, gamma, mean, rstd, dx, dgamma, dbeta, ws, nullptr, 1, nullptr, stream);
}
// ... with a second output dxs[r][c] = (T) dx[r][c] * rscale[r / rows_per_group]: when dx's consumer is the backward of a residual branch
// `x + DropPath(f(.))` (mit.py:143-146), its first step is exactly this row scaling of the incoming gradient (one launch per branch)
extern "C" int segf_layernorm_bwd_scaled(int dt, int64_t rows, int C, const void* x, const void* dy, const void* dy2, const void* dres,
                                         const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                                         float* dbeta, float* ws, const float* rscale, int64_t rows_per_group, void* dxs, void* stream) {
    return segf_layernorm_bwd_patch(dt, rows, C, x, dy, dy2, dres, gamma, mean, rstd, dx, dgamma, dbeta, ws, rscale, rows_per_group, dxs, -1, 0, stream);
}
// ... with dy2 stored in the patch-major row order of segf_layernorm_fwd_patch's y2 (log2_w >= 0): the gradient of that second output, as the
// data-gradient product of the convolution leaves it (no segf_col2im pass)
extern "C" int segf_layernorm_bwd_patch(int dt, int64_t rows, int C, const void* x, const void* dy, const void* dy2, const void* dres,
                                        const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                                        float* dbeta, float* ws, const float* rscale, int64_t rows_per_group, void* dxs, int log2_w,
                                        int log2_sr, void* stream) {
    if (rows <= 0) return 0;
    if (log2_w >= 0 && (!dy2 || !patch_geom_ok(rows, log2_w, log2_sr))) return SEGF_ERR_SHAPE;
    if ((rscale != nullptr) != (dxs != nullptr)) return SEGF_ERR_SHAPE;
    if (rscale && (rows_per_group <= 0 || rows >= (1ll << 22) || ((uintptr_t)dxs % 16))) return SEGF_ERR_SHAPE;
    if (((uintptr_t)dy2 % 16) || ((uintptr_t)dres % 16)) return SEGF_ERR_SHAPE;
    if (C <= 0 || C % 8 != 0 || C > 3072) return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    if (dgamma && dbeta != dgamma + C) return SEGF_ERR_SHAPE;   // dgamma and dbeta are one [2][C] fp32 buffer (NULL: deferred finalize)
    hipStream_t st = (hipStream_t)stream;
    int lpr_log2;
    const int vpt = ln_plan(C, lpr_log2);
    const int blocks = ln_bwd_blocks(rows, C);
    const size_t shm = (size_t)4 * 2 * C * sizeof(float);
    const float inv_rpg = rscale ? 1.0f / (float)rows_per_group : 0.f;
    SEGF_DISPATCH_DT(dt, T, { ln_bwd_launch<T>(vpt, blocks, shm, st, (const T*)x, (const T*)dy, (const T*)dy2, (const T*)dres, gamma, mean, rstd, (T*)dx, ws, rows, C, lpr_log2, rscale, inv_rpg, (T*)dxs, log2_w, log2_sr); })
    SEGF_CHECK_LAUNCH();
    if (!dgamma) return 0;                       // deferred: the caller finalizes ws later (segf_colreduce_finalize_grouped)
    const int64_t n = 2 * (int64_t)C;
    colreduce_finalize_launch(ws, blocks, n, dgamma, st);
    SEGF_CHECK_LAUNCH();
    return 0;
}
extern "C" int segf_layernorm_bwd_blocks(int64_t rows, int C) { return rows > 0 ? ln_bwd_blocks(rows, C) : 0; }
extern "C" int segf_colreduce_finalize_grouped(int n, const SegfFinalizeItem* items, void* stream) {
    if (n <= 0) return 0;
    if (!items) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    ColreduceGroup g;
    g.n = 0; g.start[0] = 0;
    for (int i = 0; i < n; ++i) {
        const SegfFinalizeItem& it = items[i];
        if (it.len <= 0 || it.nblk <= 0) continue;
        if (!it.partial || !it.out) return SEGF_ERR_SHAPE;
        const int k = g.n;
        if (it.scatter_c < 0 || (it.scatter_c > 0 && it.len != 10 * (int64_t)it.scatter_c)) return SEGF_ERR_SHAPE;
        g.partial[k] = it.partial; g.out[k] = it.out; g.nblk[k] = it.nblk; g.len[k] = it.len; g.scatter_c[k] = it.scatter_c;
        g.start[k + 1] = g.start[k] + (unsigned)cdiv64(it.len, CRF_OUT);
        if (++g.n == CRF_GROUP_MAX || i == n - 1) {
            hipLaunchKernelGGL(colreduce_finalize_group_kernel, dim3(g.start[g.n]), dim3(CRF_OUT * CRF_SL), 0, st, g);
            SEGF_CHECK_LAUNCH();
            g.n = 0;
        }
    }
    if (g.n > 0) {
        hipLaunchKernelGGL(colreduce_finalize_group_kernel, dim3(g.start[g.n]), dim3(CRF_OUT * CRF_SL), 0, st, g);
        SEGF_CHECK_LAUNCH();
    }
    return 0;
}

// ---- BatchNorm on NHWC rows ---------------------------------------------------------------------------------
template <typename T> struct BnStatF {
    const T* x; int C; bool vec;
    struct Col {};
    __device__ void init(int, int, Col&) const {}
    __device__ void operator()(const Col&, int64_t r, int c0, int nv, float (&v)[2][8]) const {
        load8_guard<T>(x + r * C + c0, nv, vec, v[0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[1][j] = v[0][j] * v[0][j];
    }
};

// sums[2][C] -> mean, rstd, running stats (fp64 combine: E[x^2]-mean^2 is formed in double)
__global__ void bn_finalize_kernel(const float* __restrict__ sums, int C, double n, float eps, float momentum,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ rmean,
                                   float* __restrict__ rvar) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mu = (double)sums[c] / n;
    double var = (double)sums[C + c] / n - mu * mu;
    if (var < 0) var = 0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unb = n > 1 ? var * n / (n - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

extern "C" int64_t segf_bn_ws(int64_t rows, int C) { return cr_ws_floats(rows, C, 2) + 2 * (int64_t)C; }

extern "C" int segf_bn_stats(int dt, int64_t rows, int C, const void* x, float* mean, float* rstd, float* running_mean,
                             float* running_var, float momentum, float eps, float* ws, void* stream) {
    if (rows <= 0 || C <= 0) return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* sums = ws + cr_ws_floats(rows, C, 2);
    SEGF_DISPATCH_DT(dt, T, {
        BnStatF<T> f{(const T*)x, C, vec_ok_host<T>(x, C)};
        const int rc = colreduce_launch<2>(f, rows, C, ws, sums, st);
        if (rc) return rc;
    })
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, (double)rows, eps, momentum, mean,
                       rstd, running_mean, running_var);
    SEGF_CHECK_LAUNCH();
    return 0;
}

// mean / rstd / running statistics from per-channel sums [2][C] = (sum x, sum x^2) produced elsewhere (segf_upsample_add_stats)
extern "C" int segf_bn_stats_from_sums(const float* sums, int64_t rows, int C, float* mean, float* rstd, float* running_mean,
                                       float* running_var, float momentum, float eps, void* stream) {
    if (rows <= 0 || C <= 0 || !sums) return SEGF_ERR_SHAPE;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, C, (double)rows, eps,
                       momentum, mean, rstd, running_mean, running_var);
    SEGF_CHECK_LAUNCH();
    return 0;
}

__device__ __forceinline__ float bn_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}
__device__ __forceinline__ float bn_act_mask(float pre, int act) {
    if (act == 1) return pre > 0.f ? 1.f : 0.f;
    if (act == 2) return (pre > 0.f && pre < 6.f) ? 1.f : 0.f;
    return 1.f;
}

// y = act(a * x + b) * chan_scale, a = gamma * rstd, b = beta - mean * a: column-fixed threads, parameters in registers
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int act, const float* __restrict__ cscale,
                                                        FastDivU32 rps, T* __restrict__ y, int64_t rows, int C, bool vec) {
    const int nchunk = (C + 7) / 8;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t T_ = (int64_t)gridDim.x * 256;
    const int ch = (int)(g % nchunk);
    const int64_t rstep = T_ / nchunk;
    const int c0 = ch * 8;
    const int nv = C - c0 < 8 ? C - c0 : 8;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + (j < nv ? j : 0);
        a[j] = gamma[c] * rstd[c];
        b[j] = beta[c] - mean[c] * a[j];
    }
    // the row loop is instantiated under `full 16-byte-aligned chunk` and under its negation, so that the guarded accesses of
    // the first copy fold to plain vector loads / stores (see colreduce_kernel)
    auto row_loop = [&](const int nvv, const bool vv) {
#pragma unroll 2
        for (int64_t r = g / nchunk; r < rows; r += rstep) {
            float v[8];
            load8_guard<T>(x + r * C + c0, nvv, vv, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = bn_act(fmaf(v[j], a[j], b[j]), act);
            if (cscale) {
                float cs[8];
                load8_guard<float>(cscale + (int64_t)fastdiv((uint32_t)r, rps) * C + c0, nvv, vv, cs);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= cs[j];
            }
            store8_guard<T>(y + r * C + c0, nvv, vv, v);
        }
    };
    if (nv >= 8 && vec) row_loop(8, true); else row_loop(nv, vec);
}

extern "C" int segf_bn_apply(int dt, int64_t rows, int C, const void* x, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, int act, const float* chan_scale,
                             int64_t rows_per_sample, void* y, void* stream) {
    if (rows <= 0 || C <= 0) return 0;
    if (chan_scale && rows_per_sample <= 0) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = colfixed_blocks(rows, (C + 7) / 8, 4, 8192);
    SEGF_DISPATCH_DT(dt, T, {
        const bool vec = vec_ok_host<T>(x, C) && vec_ok_host<T>(y, C) && (C % 8 == 0);
        hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)x, mean, rstd, gamma, beta, act,
                           chan_scale, fastdiv_make((uint32_t)(rows_per_sample > 0 ? rows_per_sample : 1)), (T*)y, rows, C, vec);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// BatchNorm (+ Dropout2d channel scale) as per-(sample, channel) affine tables for segf_gemm_pro:
//   scale[g][c] = gamma rstd cs[g][c],  shift[g][c] = (beta - mean gamma rstd) cs[g][c]
// cs >= 0, so ReLU(a x + b) cs == ReLU(cs a x + cs b): the consumer applies the activation after the scaled affine.
__global__ void bn_affine_table_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, const float* __restrict__ cs, int G, int C,
                                       float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * C) return;
    const int c = i % C;
    const float a = gamma[c] * rstd[c], s = cs ? cs[i] : 1.f;
    scale[i] = a * s;
    shift[i] = (beta[c] - mean[c] * a) * s;
}
extern "C" int segf_bn_affine_table(const float* mean, const float* rstd, const float* gamma, const float* beta,
                                    const float* chan_scale, int groups, int C, float* scale, float* shift, void* stream) {
    if (groups <= 0 || C <= 0) return SEGF_ERR_SHAPE;
    hipLaunchKernelGGL(bn_affine_table_kernel, dim3((groups * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, rstd, gamma,
                       beta, chan_scale, groups, C, scale, shift);
    SEGF_CHECK_LAUNCH();
    return 0;
}

// backward sums: [0] = sum dyr, [1] = sum dyr * xhat, where dyr = dy * chan_scale * act'(pre)
struct BnCol { float mean[8], rstd[8], a[8], b[8]; };
__device__ __forceinline__ void bn_col_init(const float* mean, const float* rstd, const float* gamma, const float* beta, int c0,
                                            int nv, BnCol& col) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + (j < nv ? j : 0);
        col.mean[j] = mean[c]; col.rstd[j] = rstd[c];
        col.a[j] = gamma[c]; col.b[j] = beta[c];
    }
}
template <typename T> struct BnBwdF {
    const T* x; const T* dy; const float* mean; const float* rstd; const float* gamma; const float* beta;
    const float* cscale; FastDivU32 rps; int C; int act; bool vec;
    typedef BnCol Col;
    __device__ void init(int c0, int nv, Col& col) const { bn_col_init(mean, rstd, gamma, beta, c0, nv, col); }
    __device__ void operator()(const Col& col, int64_t r, int c0, int nv, float (&v)[2][8]) const {
        float xv[8], dv[8], cs[8];
        load8_guard<T>(x + r * C + c0, nv, vec, xv);
        load8_guard<T>(dy + r * C + c0, nv, vec, dv);
        if (cscale) load8_guard<float>(cscale + (int64_t)fastdiv((uint32_t)r, rps) * C + c0, nv, vec, cs);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (xv[j] - col.mean[j]) * col.rstd[j];
            float d = dv[j] * bn_act_mask(fmaf(xh, col.a[j], col.b[j]), act);
            if (cscale) d *= cs[j];
            v[0][j] = j < nv ? d : 0.f;
            v[1][j] = j < nv ? d * xh : 0.f;
        }
    }
};

template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                            const float* __restrict__ cscale, FastDivU32 rps,
                                                            const float* __restrict__ sums /*[2][C]: dbeta, dgamma*/,
                                                            int eval_mode, T* __restrict__ dx, int64_t rows, int C, bool vec) {
    const int nchunk = (C + 7) / 8;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t T_ = (int64_t)gridDim.x * 256;
    const int ch = (int)(g % nchunk);
    const int64_t rstep = T_ / nchunk;
    const int c0 = ch * 8;
    const int nv = C - c0 < 8 ? C - c0 : 8;
    const float invn = 1.f / (float)rows;
    BnCol col;
    bn_col_init(mean, rstd, gamma, beta, c0, nv, col);
    float k1[8], k2[8], gr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + (j < nv ? j : 0);
        k1[j] = eval_mode ? 0.f : sums[c] * invn;
        k2[j] = eval_mode ? 0.f : sums[C + c] * invn;
        gr[j] = col.a[j] * col.rstd[j];
    }
    auto row_loop = [&](const int nvv, const bool vv) {
#pragma unroll 2
        for (int64_t r = g / nchunk; r < rows; r += rstep) {
            float xv[8], dv[8], cs[8];
            load8_guard<T>(x + r * C + c0, nvv, vv, xv);
            load8_guard<T>(dy + r * C + c0, nvv, vv, dv);
            if (cscale) load8_guard<float>(cscale + (int64_t)fastdiv((uint32_t)r, rps) * C + c0, nvv, vv, cs);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (xv[j] - col.mean[j]) * col.rstd[j];
                float d = dv[j] * bn_act_mask(fmaf(xh, col.a[j], col.b[j]), act);
                if (cscale) d *= cs[j];
                xv[j] = gr[j] * (d - k1[j] - xh * k2[j]);
            }
            store8_guard<T>(dx + r * C + c0, nvv, vv, xv);
        }
    };
    if (nv >= 8 && vec) row_loop(8, true); else row_loop(nv, vec);
}

// dgamma = sums[1], dbeta = sums[0] are produced in-place: caller passes dbeta = ws-resident [C], dgamma [C]
__global__ void bn_copy_grads_kernel(const float* __restrict__ sums, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = sums[c];
    dgamma[c] = sums[C + c];
}

extern "C" int segf_bn_bwd(int dt, int64_t rows, int C, const void* x, const void* dy, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int act, const float* chan_scale, int64_t rows_per_sample,
                           int eval_mode, void* dx, float* dgamma, float* dbeta, float* ws, void* stream) {
    if (rows <= 0 || C <= 0) return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    if (chan_scale && rows_per_sample <= 0) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    float* sums = ws + cr_ws_floats(rows, C, 2);
    if (rows > 0xffffffffll) return SEGF_ERR_SHAPE;
    const FastDivU32 rps = fastdiv_make((uint32_t)(rows_per_sample > 0 ? rows_per_sample : 1));
    const int blocks = colfixed_blocks(rows, (C + 7) / 8, 4, 8192);
    SEGF_DISPATCH_DT(dt, T, {
        const bool vec = vec_ok_host<T>(x, C) && vec_ok_host<T>(dy, C) && vec_ok_host<T>(dx, C) && (C % 8 == 0);
        BnBwdF<T> f{(const T*)x, (const T*)dy, mean, rstd, gamma, beta, chan_scale, rps, C, act, vec};
        const int rc = colreduce_launch<2>(f, rows, C, ws, sums, st);
        if (rc) return rc;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)x, (const T*)dy, mean, rstd,
                           gamma, beta, act, chan_scale, rps, sums, eval_mode, (T*)dx, rows, C, vec);
    })
    SEGF_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_copy_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, dgamma, dbeta);
    SEGF_CHECK_LAUNCH();
    return 0;
}


// ---- Global Response Normalization (ConvNeXtV2, models/backbones/convnextv2.py:68-80) on NHWC rows -----------------------
//   Gx[b][c] = ||x[b,:,:,c]||_2 ;  Nx = Gx / (mean_c Gx + 1e-6) ;  y = gamma * (x * Nx) + beta + x = a[b][c] * x + beta[c]
// forward : batched column reduction (sum x^2) -> per-image coefficient kernel -> streaming apply
// backward: batched column reduction (sum dy*x, sum dy) -> coefficient kernel -> dx = a * dy + K * x, with
//           K[b][c] = dL/dGx / Gx,  dL/dGx[c] = A[c]/(m+eps) - (1/C) sum_c' A[c'] Gx[c'] / (m+eps)^2,  A = gamma * sum_p dy*x
#define GRN_EPS 1e-6f
// ACT: the GRN input is gelu(x) of what is stored (ConvNeXtV2: pwconv1 -> GELU -> GRN, convnextv2.py:92-94): the activation is
// applied on the way in, in fp32, and gelu(x) is never written -- the pre-activation is what the GELU backward needs anyway.
template <typename T, bool ACT = false> struct GrnSqF {
    const T* x; int C; bool vec;
    struct Col {};
    __device__ void init(int, int, Col&) const {}
    __device__ void operator()(const Col&, int64_t r, int c0, int nv, float (&v)[1][8]) const {
        load8_guard<T>(x + r * C + c0, nv, vec, v[0]);
        if (ACT) gelu_erf8_floats(v[0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[0][j] *= v[0][j];
    }
};
template <typename T, bool ACT = false> struct GrnBwdF {
    const T* x; const T* dy; int C; bool vec;
    struct Col {};
    __device__ void init(int, int, Col&) const {}
    __device__ void operator()(const Col&, int64_t r, int c0, int nv, float (&v)[2][8]) const {
        float xv[8];
        load8_guard<T>(x + r * C + c0, nv, vec, xv);
        load8_guard<T>(dy + r * C + c0, nv, vec, v[1]);
        if (ACT) gelu_erf8_floats(xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[0][j] = v[1][j] * xv[j];
    }
};
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum_all(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// one block per image.  fwd (S1 == nullptr): a[b][c] = 1 + gamma Nx, stash Gx.  bwd: also K[b][c] and per-image dgamma parts.
__global__ void __launch_bounds__(256) grn_coef_kernel(const float* __restrict__ sumsq, const float* __restrict__ gamma, int C,
                                                        float* __restrict__ a, float* __restrict__ G_out,
                                                        const float* __restrict__ S1, float* __restrict__ K, float* __restrict__ dg_part) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* sq = sumsq + (int64_t)b * C;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += sqrtf(sq[c]);
    const float m = block_sum_256(s, red) / (float)C;
    const float inv = 1.f / (m + GRN_EPS);
    float t = 0.f;
    if (S1)
        for (int c = threadIdx.x; c < C; c += 256) t += gamma[c] * S1[(int64_t)b * 2 * C + c] * sqrtf(sq[c]);
    const float AG = S1 ? block_sum_256(t, red) : 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float G = sqrtf(sq[c]);
        const float nx = G * inv;
        a[(int64_t)b * C + c] = 1.f + gamma[c] * nx;
        if (G_out) G_out[(int64_t)b * C + c] = sq[c];       // saved for the backward: per-image sum of squares
        if (S1) {
            const float s1 = S1[(int64_t)b * 2 * C + c];
            const float dG = gamma[c] * s1 * inv - AG * inv * inv / (float)C;
            K[(int64_t)b * C + c] = G > 0.f ? dG / G : 0.f;
            dg_part[(int64_t)b * C + c] = nx * s1;
        }
    }
}
// y = a[b][c] * x + (k ? k[b][c] * x2 : beta[c])     fwd: (x, a, beta);  bwd: dx = a * dy + K * x  (x := dy, x2 := x)
// ACT (see GrnSqF): fwd y = a * gelu(x) + beta;  bwd (x := dy, x2 := the stored pre-activation u): du = (a * dy + K * gelu(u)) * gelu'(u)
template <typename T, bool ACT = false>
__global__ void __launch_bounds__(256) grn_apply_kernel(const T* __restrict__ x, const T* __restrict__ x2, const float* __restrict__ a,
                                                         const float* __restrict__ k, const float* __restrict__ beta, T* __restrict__ y,
                                                         int64_t rows, FastDivU32 rps, int C) {
    const int nchunk = C / 8;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t rstep = ((int64_t)gridDim.x * 256) / nchunk;
    const int c0 = (int)(g % nchunk) * 8;
    float bt[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bt[j] = beta ? beta[c0 + j] : 0.f;
    for (int64_t r = g / nchunk; r < rows; r += rstep) {
        const int64_t b = fastdiv((uint32_t)r, rps);
        float v[8], av[8];
        load8<T>(x + r * C + c0, v);
        load8f(a + b * C + c0, av);
        if (k) {
            float w[8], kv[8];
            load8<T>(x2 + r * C + c0, w);
            load8f(k + b * C + c0, kv);
            if (ACT) {
                f32x2_t u2[4] = {f32x2_t{w[0], w[1]}, f32x2_t{w[2], w[3]}, f32x2_t{w[4], w[5]}, f32x2_t{w[6], w[7]}}, g2[4];
                gelu_erf8_both(u2, g2);                       // g2 = gelu(u), u2 = gelu'(u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[2 * i] = fmaf(av[2 * i], v[2 * i], kv[2 * i] * g2[i].x) * u2[i].x;
                    v[2 * i + 1] = fmaf(av[2 * i + 1], v[2 * i + 1], kv[2 * i + 1] * g2[i].y) * u2[i].y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(av[j], v[j], kv[j] * w[j]);
            }
        } else {
            if (ACT) gelu_erf8_floats(v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = fmaf(av[j], v[j], bt[j]);
        }
        store8<T>(y + r * C + c0, v);
    }
}
// dgamma[c] = sum_b dg_part[b][c], dbeta[c] = sum_b S[b][1][c]
__global__ void grn_param_grads_kernel(const float* __restrict__ dg_part, const float* __restrict__ S, int B, int C,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float g = 0.f, bsum = 0.f;
    for (int b = 0; b < B; ++b) { g += dg_part[(int64_t)b * C + c]; bsum += S[((int64_t)b * 2 + 1) * C + c]; }
    dgamma[c] = g; dbeta[c] = bsum;
}

// workspace (floats): forward B*cr_ws(rows,C,1) + B*C; backward B*cr_ws(rows,C,2) + 4*B*C
extern "C" int64_t segf_grn_ws(int B, int64_t rows_per_sample, int C, int bwd) {
    return (int64_t)B * cr_ws_floats(rows_per_sample, C, bwd ? 2 : 1) + (bwd ? 4 : 1) * (int64_t)B * C;
}
// y = gamma * (x * Nx) + beta + x;  a_out [B][C] (scratch) and g_out [B][C] (sum_p x^2, saved for the backward)
extern "C" int segf_grn_fwd(int dt, int B, int64_t rows_per_sample, int C, const void* x, const float* gamma, const float* beta,
                            void* y, float* a_out, float* g_out, float* ws, int pre_gelu, void* stream) {
    if (B <= 0 || rows_per_sample <= 0) return 0;
    if (C <= 0 || C % 8 || ((uintptr_t)x % 16) || ((uintptr_t)y % 16)) return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* sumsq = ws + (int64_t)B * cr_ws_floats(rows_per_sample, C, 1);
    const int64_t rows = (int64_t)B * rows_per_sample;
    SEGF_DISPATCH_DT(dt, T, {
        int rc;
        if (pre_gelu) { GrnSqF<T, true> f{(const T*)x, C, true}; rc = colreduce_launch_batched<1>(f, rows_per_sample, B, C, ws, sumsq, st); }
        else { GrnSqF<T, false> f{(const T*)x, C, true}; rc = colreduce_launch_batched<1>(f, rows_per_sample, B, C, ws, sumsq, st); }
        if (rc) return rc;
        hipLaunchKernelGGL(grn_coef_kernel, dim3(B), dim3(256), 0, st, sumsq, gamma, C, a_out, g_out, (const float*)nullptr,
                           (float*)nullptr, (float*)nullptr);
        const dim3 grid(colfixed_blocks(rows, C / 8, 4, 8192));
        if (pre_gelu) hipLaunchKernelGGL((grn_apply_kernel<T, true>), grid, dim3(256), 0, st, (const T*)x,
                           (const T*)nullptr, a_out, (const float*)nullptr, beta, (T*)y, rows, fastdiv_make((uint32_t)rows_per_sample), C);
        else hipLaunchKernelGGL((grn_apply_kernel<T, false>), grid, dim3(256), 0, st, (const T*)x,
                           (const T*)nullptr, a_out, (const float*)nullptr, beta, (T*)y, rows, fastdiv_make((uint32_t)rows_per_sample), C);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}
extern "C" int segf_grn_bwd(int dt, int B, int64_t rows_per_sample, int C, const void* x, const void* dy, const float* gamma,
                            const float* g_saved, void* dx, float* dgamma, float* dbeta, float* ws, int pre_gelu, void* stream) {
    if (B <= 0 || rows_per_sample <= 0) return 0;
    if (C <= 0 || C % 8 || ((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || ((uintptr_t)dx % 16)) return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* S = ws + (int64_t)B * cr_ws_floats(rows_per_sample, C, 2);      // [B][2][C]: sum dy*x, sum dy
    float* a = S + 2 * (int64_t)B * C;
    float* K = a + (int64_t)B * C;
    const int64_t rows = (int64_t)B * rows_per_sample;
    SEGF_DISPATCH_DT(dt, T, {
        int rc;
        if (pre_gelu) { GrnBwdF<T, true> f{(const T*)x, (const T*)dy, C, true}; rc = colreduce_launch_batched<2>(f, rows_per_sample, B, C, ws, S, st); }
        else { GrnBwdF<T, false> f{(const T*)x, (const T*)dy, C, true}; rc = colreduce_launch_batched<2>(f, rows_per_sample, B, C, ws, S, st); }
        if (rc) return rc;
        // dg_part reuses the front of the (now consumed) partial workspace
        hipLaunchKernelGGL(grn_coef_kernel, dim3(B), dim3(256), 0, st, g_saved, gamma, C, a, (float*)nullptr, S, K, ws);
        const dim3 grid(colfixed_blocks(rows, C / 8, 4, 8192));
        if (pre_gelu) hipLaunchKernelGGL((grn_apply_kernel<T, true>), grid, dim3(256), 0, st, (const T*)dy,
                           (const T*)x, a, K, (const float*)nullptr, (T*)dx, rows, fastdiv_make((uint32_t)rows_per_sample), C);
        else hipLaunchKernelGGL((grn_apply_kernel<T, false>), grid, dim3(256), 0, st, (const T*)dy,
                           (const T*)x, a, K, (const float*)nullptr, (T*)dx, rows, fastdiv_make((uint32_t)rows_per_sample), C);
    })
    SEGF_CHECK_LAUNCH();
    hipLaunchKernelGGL(grn_param_grads_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, S, B, C, dgamma, dbeta);
    SEGF_CHECK_LAUNCH();
    return 0;
}
