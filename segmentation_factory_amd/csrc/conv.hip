// Spatial ops on NHWC activations: depthwise 3x3 (+bias +GELU) forward/backward, im2col / col2im for the
// strided PatchEmbed and spatial-reduction convolutions (which then run as MFMA GEMMs).
//   DWConv + GELU : reference models/backbones/mit.py:62-71 (DWConv), :98-99 (F.gelu(self.dwconv(self.fc1(x))))
//   PatchEmbed    : mit.py:105,127 (Conv2d k7 s4 p3 / k3 s2 p1);  sr conv: mit.py:21,48 (Conv2d k=s=sr)
// All HBM-bound: channel-contiguous 16-B lane accesses, fp32 math.
#include "colreduce.h"

#define DW_PIX 4   // output pixels per thread along W (sliding 3x(PIX+2) window in registers)

// y[b][y][x][c] = act( sum_{ky,kx} w[c][ky*3+kx] * x[b][y+ky-1][x+kx-1][c] + bias[c] );  FLIP: correlation with the
// flipped kernel (= transposed conv for the data gradient), no bias / activation.
template <typename T, bool FLIP>
__global__ void __launch_bounds__(256) dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int apply_gelu, T* __restrict__ y,
                                                         int B, int H, int W, int C) {
    const int nchunk = C / 8;
    const int wg = (W + DW_PIX - 1) / DW_PIX;
    const int64_t total = (int64_t)B * H * wg * nchunk;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % nchunk);
        int64_t t = idx / nchunk;
        const int xg = (int)(t % wg); t /= wg;
        const int yy = (int)(t % H);
        const int b = (int)(t / H);
        const int c0 = ch * 8, x0 = xg * DW_PIX;
        float wk[9][8], bs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int kk = 0; kk < 9; ++kk) wk[kk][j] = w[(c0 + j) * 9 + (FLIP ? 8 - kk : kk)];
            bs[j] = (!FLIP && bias) ? bias[c0 + j] : 0.f;
        }
        float acc[DW_PIX][8];
#pragma unroll
        for (int p = 0; p < DW_PIX; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][j] = bs[j];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = yy + ky - 1;
            if (iy < 0 || iy >= H) continue;
            const T* row = x + (((int64_t)b * H + iy) * W) * C + c0;
#pragma unroll
            for (int cx = 0; cx < DW_PIX + 2; ++cx) {
                const int ix = x0 + cx - 1;
                if (ix < 0 || ix >= W) continue;
                float v[8];
                load8<T>(row + (int64_t)ix * C, v);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int p = cx - kx;            // output pixel that sees this column through tap kx
                    if (p >= 0 && p < DW_PIX) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[p][j] = fmaf(wk[ky * 3 + kx][j], v[j], acc[p][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < DW_PIX; ++p) {
            if (x0 + p < W) {
                if (!FLIP && apply_gelu) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[p][j] = gelu_erf(acc[p][j]);
                }
                store8<T>(y + (((int64_t)b * H + yy) * W + x0 + p) * C + c0, acc[p]);
            }
        }
    }
}

extern "C" int segf_dwconv3x3_gelu_fwd(int dt, int B, int H, int W, int C, const void* x, const float* w, const float* bias,
                                       int apply_gelu, void* y, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (C <= 0 || C % 8 != 0 || ((uintptr_t)x % 16) || ((uintptr_t)y % 16)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * ((W + DW_PIX - 1) / DW_PIX) * (C / 8);
    const int blocks = (int)imin64(cdiv64(total, 256), 8192);
    SEGF_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL((dwconv3x3_kernel<T, false>), dim3(blocks), dim3(256), 0, st, (const T*)x, w, bias, apply_gelu, (T*)y, B, H, W, C);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// backward pass 1 (column reduction over pixels, 10 sums per channel): recompute u = conv(x)+b at the pixel,
// du = dy * gelu'(u) (stored), dw[k] += du * x[p+off_k], db += du.
template <typename T> struct DwBwdF {
    const T* x; const T* dy; T* du; const float* w; const float* bias; int H, W, C, apply_gelu;
    __device__ void operator()(int64_t r, int c0, int nv, float (&out)[10][8]) const {
        const int xx = (int)(r % W);
        const int64_t t = r / W;
        const int yy = (int)(t % H);
        const int64_t b = t / H;
        float u[8], xv[9][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = bias ? bias[c0 + j] : 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = yy + ky - 1, ix = xx + kx - 1;
                const int kk = ky * 3 + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    load8<T>(x + ((b * H + iy) * W + ix) * C + c0, xv[kk]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) u[j] = fmaf(w[(c0 + j) * 9 + kk], xv[kk][j], u[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[kk][j] = 0.f;
                }
            }
        float g[8];
        load8<T>(dy + r * C + c0, g);
        if (apply_gelu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] *= gelu_erf_grad(u[j]);
        }
        store8<T>(du + r * C + c0, g);
#pragma unroll
        for (int kk = 0; kk < 9; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) out[kk][j] = g[j] * xv[kk][j];
#pragma unroll
        for (int j = 0; j < 8; ++j) out[9][j] = g[j];
    }
};

// [10][C] sums -> dw[C][9], db[C]
__global__ void dw_scatter_kernel(const float* __restrict__ sums, int C, float* __restrict__ dw, float* __restrict__ db) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int kk = 0; kk < 9; ++kk) dw[c * 9 + kk] = sums[kk * C + c];
    db[c] = sums[9 * C + c];
}

extern "C" int64_t segf_dwconv3x3_bwd_ws(int B, int H, int W, int C) {
    return cr_ws_floats((int64_t)B * H * W, C, 10) + 10 * (int64_t)C;
}

extern "C" int segf_dwconv3x3_gelu_bwd(int dt, int B, int H, int W, int C, const void* x, const float* w, const float* bias,
                                       int apply_gelu, const void* dy, void* du, void* dx, float* dw, float* db, float* ws,
                                       void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (C <= 0 || C % 8 != 0 || ((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || ((uintptr_t)du % 16) || ((uintptr_t)dx % 16))
        return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * H * W;
    float* sums = ws + cr_ws_floats(rows, C, 10);
    const int64_t total = (int64_t)B * H * ((W + DW_PIX - 1) / DW_PIX) * (C / 8);
    const int blocks = (int)imin64(cdiv64(total, 256), 8192);
    SEGF_DISPATCH_DT(dt, T, {
        DwBwdF<T> f{(const T*)x, (const T*)dy, (T*)du, w, bias, H, W, C, apply_gelu};
        const int rc = colreduce_launch<10>(f, rows, C, ws, sums, st);
        if (rc) return rc;
        hipLaunchKernelGGL((dwconv3x3_kernel<T, true>), dim3(blocks), dim3(256), 0, st, (const T*)du, w, (const float*)nullptr, 0, (T*)dx, B, H, W, C);
    })
    SEGF_CHECK_LAUNCH();
    hipLaunchKernelGGL(dw_scatter_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, dw, db);
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- im2col / col2im ---------------------------------------------------------------------------------------
// NHWC input, 8 channels per thread: col[m][(ky*kw+kx)*Cin + ci]
template <typename T>
__global__ void im2col_nhwc_kernel(const T* __restrict__ x, T* __restrict__ col, int64_t ldcol, int B, int H, int W, int Cin,
                                   int kh, int kw, int stride, int pad, int Ho, int Wo) {
    const int nch = Cin / 8;
    const int64_t total = (int64_t)B * Ho * Wo * kh * kw * nch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % nch);
        int64_t t = idx / nch;
        const int kx = (int)(t % kw); t /= kw;
        const int ky = (int)(t % kh); t /= kh;
        const int64_t m = t;
        const int ox = (int)(m % Wo);
        const int64_t t2 = m / Wo;
        const int oy = (int)(t2 % Ho);
        const int64_t b = t2 / Ho;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        float v[8];
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) load8<T>(x + ((b * H + iy) * W + ix) * Cin + ch * 8, v);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        store8<T>(col + m * ldcol + (int64_t)(ky * kw + kx) * Cin + ch * 8, v);
    }
}
// fp32 NCHW image input (Cin small, e.g. 3): one output element per thread, pad columns zeroed
template <typename T>
__global__ void im2col_nchw_kernel(const float* __restrict__ x, T* __restrict__ col, int64_t ldcol, int B, int H, int W, int Cin,
                                   int kh, int kw, int stride, int pad, int Ho, int Wo) {
    const int64_t total = (int64_t)B * Ho * Wo * ldcol;
    const int K = kh * kw * Cin;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int kcol = (int)(idx % ldcol);
        const int64_t m = idx / ldcol;
        float v = 0.f;
        if (kcol < K) {
            const int ci = kcol % Cin;
            const int kk = kcol / Cin;
            const int kx = kk % kw, ky = kk / kw;
            const int ox = (int)(m % Wo);
            const int64_t t2 = m / Wo;
            const int oy = (int)(t2 % Ho);
            const int64_t b = t2 / Ho;
            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((b * Cin + ci) * H + iy) * W + ix];
        }
        stf<T>(col + idx, v);
    }
}
// pad columns [K, ldcol) of the NHWC im2col matrix
template <typename T>
__global__ void zero_cols_kernel(T* __restrict__ col, int64_t ldcol, int64_t rows, int k0) {
    const int npad = (int)(ldcol - k0);
    const int64_t total = rows * npad;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / npad;
        stf<T>(col + r * ldcol + k0 + (idx - r * npad), 0.f);
    }
}

extern "C" int segf_im2col(int dt, int in_nchw_f32, int B, int H, int W, int Cin, int kh, int kw, int stride, int pad, int Ho,
                           int Wo, const void* x, void* col, int64_t ldcol, void* stream) {
    if (B <= 0 || Ho <= 0 || Wo <= 0) return 0;
    const int K = kh * kw * Cin;
    if (ldcol < K || stride <= 0) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * Ho * Wo;
    if (in_nchw_f32) {
        const int blocks = (int)imin64(cdiv64(rows * ldcol, 256), 8192);
        SEGF_DISPATCH_DT(dt, T, {
            hipLaunchKernelGGL((im2col_nchw_kernel<T>), dim3(blocks), dim3(256), 0, st, (const float*)x, (T*)col, ldcol, B, H, W, Cin, kh, kw, stride, pad, Ho, Wo);
        })
    } else {
        const int64_t esz = dt == SEGF_BF16 ? 2 : 4;
        if (Cin % 8 != 0 || ((uintptr_t)x % 16) || ((uintptr_t)col % 16) || ((ldcol * esz) % 16)) return SEGF_ERR_SHAPE;
        const int blocks = (int)imin64(cdiv64(rows * kh * kw * (Cin / 8), 256), 8192);
        SEGF_DISPATCH_DT(dt, T, {
            hipLaunchKernelGGL((im2col_nhwc_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)x, (T*)col, ldcol, B, H, W, Cin, kh, kw, stride, pad, Ho, Wo);
            if (ldcol > K) {
                const int zb = (int)imin64(cdiv64(rows * (ldcol - K), 256), 2048);
                hipLaunchKernelGGL((zero_cols_kernel<T>), dim3(zb), dim3(256), 0, st, (T*)col, ldcol, rows, K);
            }
        })
    }
    SEGF_CHECK_LAUNCH();
    return 0;
}

// dx[b][iy][ix][ci] = sum over (ky,kx) with (iy+pad-ky) % stride == 0 ... of dcol[(b,oy,ox)][(ky,kx,ci)]  (gather form, no atomics)
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcol, int64_t ldcol, T* __restrict__ dx, int B, int H, int W, int Cin, int kh,
                              int kw, int stride, int pad, int Ho, int Wo) {
    const int nch = Cin / 8;
    const int64_t total = (int64_t)B * H * W * nch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % nch);
        int64_t t = idx / nch;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const int64_t b = t / H;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int ky = 0; ky < kh; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || ty % stride) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
            for (int kx = 0; kx < kw; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || tx % stride) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                float v[8];
                load8<T>(dcol + ((b * Ho + oy) * Wo + ox) * ldcol + (int64_t)(ky * kw + kx) * Cin + ch * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        }
        store8<T>(dx + ((b * H + iy) * W + ix) * Cin + ch * 8, acc);
    }
}

extern "C" int segf_col2im(int dt, int B, int H, int W, int Cin, int kh, int kw, int stride, int pad, int Ho, int Wo,
                           const void* dcol, int64_t ldcol, void* dx, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t esz = dt == SEGF_BF16 ? 2 : 4;
    if (Cin % 8 != 0 || stride <= 0 || ldcol < (int64_t)kh * kw * Cin || ((uintptr_t)dcol % 16) || ((uintptr_t)dx % 16) ||
        ((ldcol * esz) % 16))
        return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)imin64(cdiv64((int64_t)B * H * W * (Cin / 8), 256), 8192);
    SEGF_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL((col2im_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)dcol, ldcol, (T*)dx, B, H, W, Cin, kh, kw, stride, pad, Ho, Wo);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}
