// Fused final bilinear upsample + CrossEntropy + Dice, and fused upsample + argmax + confusion matrix.
//   reference: models/build_models.py:65 (F.interpolate to the input size), engine.py:10-15 (criterion),
//              util/losses.py:126-177 (build_target / dice_coeff / multiclass_dice_coeff / dice_loss),
//              engine.py:89-91 + util/utils.py:99-109 + util/metrics.py:24-27 (evaluate).
// The reference materialises fp32 full-resolution logits, a softmax copy and a one-hot copy (3 x 157 MB per
// 512^2 x 150-class image) and then loops over batch x class in Python.  Here one wave owns one full-resolution
// pixel at a time with the classes spread over its 64 lanes (<= 3 classes per lane): the 4 low-resolution taps are
// read as coalesced class rows, softmax statistics are two wave reductions, and the per-(image, class) Dice sums
// I = sum p_c [t=c], P = sum p_c, T = sum [t=c] accumulate in lane registers.  Closed form:
//   loss = CE + 1 - mean_{b,c} (2 I + eps) / (P + T + eps),   (P + T == 0  =>  denominator := 2 I)
// Deterministic: per-block partials + fixed-order finalize, no atomics.
#include "common.h"

#define LS_NBLK 64          // blocks per image
#define LS_THREADS 256
#define LS_EPS 1e-6f

struct LossGeom { int B, C, h, w, H, W; int64_t ldl; };

// z[s] = upsampled logit of class lane + 64*s at full-res pixel (Y, X); invalid class slots get -inf
template <typename T, int NS>
__device__ __forceinline__ void pixel_logits(const T* __restrict__ img, const LossGeom& g, int Y, int X, int lane, float (&z)[NS]) {
    if (g.h == g.H && g.w == g.W) {
        const T* p = img + ((int64_t)Y * g.w + X) * g.ldl;
#pragma unroll
        for (int s = 0; s < NS; ++s) { const int c = lane + 64 * s; z[s] = c < g.C ? ldf<T>(p + c) : -INFINITY; }
        return;
    }
    int y0, y1, x0, x1; float ly, lx;
    bilinear_src(Y, g.h, g.H, 0, y0, y1, ly);
    bilinear_src(X, g.w, g.W, 0, x0, x1, lx);
    const T* p00 = img + ((int64_t)y0 * g.w + x0) * g.ldl;
    const T* p01 = img + ((int64_t)y0 * g.w + x1) * g.ldl;
    const T* p10 = img + ((int64_t)y1 * g.w + x0) * g.ldl;
    const T* p11 = img + ((int64_t)y1 * g.w + x1) * g.ldl;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = lane + 64 * s;
        if (c < g.C) {
            const float a = ldf<T>(p00 + c), b = ldf<T>(p01 + c), cc = ldf<T>(p10 + c), d = ldf<T>(p11 + c);
            z[s] = (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * cc + lx * d);
        } else z[s] = -INFINITY;
    }
}

// softmax over the wave: p[s], returns log-sum-exp
template <int NS>
__device__ __forceinline__ float wave_softmax(const float (&z)[NS], float (&p)[NS]) {
    float mx = z[0];
#pragma unroll
    for (int s = 1; s < NS; ++s) mx = fmaxf(mx, z[s]);
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) { p[s] = expf(z[s] - mx); sum += p[s]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int s = 0; s < NS; ++s) p[s] *= inv;
    return mx + logf(sum);
}

// partial layout per (b, blk): [3][C] (I, P, T) then {ce_sum, w_sum, n_valid, bad}
template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) ce_dice_fwd_kernel(const T* __restrict__ logits, LossGeom g,
                                                                  const int64_t* __restrict__ target, int64_t ignore_index,
                                                                  const float* __restrict__ cw, float* __restrict__ partial) {
    __shared__ float red[4][3 * 64 * NS + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int64_t npix = (int64_t)g.H * g.W;
    const int64_t nw = (int64_t)LS_NBLK * 4;
    const int64_t per = (npix + nw - 1) / nw;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p0 = wid * per, p1 = p0 + per < npix ? p0 + per : npix;
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * npix;
    float aI[NS], aP[NS], aT[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { aI[s] = 0.f; aP[s] = 0.f; aT[s] = 0.f; }
    float ce = 0.f, wsum = 0.f, nvalid = 0.f, bad = 0.f;
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t t = tg[p];
        if (t == ignore_index) continue;                 // wave-uniform
        if (t < 0 || t >= g.C) { bad = 1.f; continue; }  // the reference raises here (one_hot / cross_entropy)
        const int Y = (int)(p / g.W), X = (int)(p - (int64_t)Y * g.W);
        float z[NS], pr[NS];
        pixel_logits<T, NS>(img, g, Y, X, lane, z);
        const float lse = wave_softmax<NS>(z, pr);
        const float wt = cw ? cw[t] : 1.f;
        nvalid += 1.f; wsum += wt;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = lane + 64 * s;
            aP[s] += pr[s];
            if (c == (int)t) { aI[s] += pr[s]; aT[s] += 1.f; ce += wt * (lse - z[s]); }
        }
    }
    ce = wave_sum(ce);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        red[wave][0 * 64 * NS + 64 * s + lane] = aI[s];
        red[wave][1 * 64 * NS + 64 * s + lane] = aP[s];
        red[wave][2 * 64 * NS + 64 * s + lane] = aT[s];
    }
    if (lane == 0) {
        red[wave][3 * 64 * NS + 0] = ce; red[wave][3 * 64 * NS + 1] = wsum;
        red[wave][3 * 64 * NS + 2] = nvalid; red[wave][3 * 64 * NS + 3] = bad;
    }
    __syncthreads();
    float* dst = partial + ((int64_t)b * LS_NBLK + blockIdx.x) * (3 * g.C + 4);
    for (int i = threadIdx.x; i < 3 * 64 * NS + 4; i += LS_THREADS) {
        const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        if (i >= 3 * 64 * NS) dst[3 * g.C + (i - 3 * 64 * NS)] = v;
        else {
            const int which = i / (64 * NS), c = i - which * 64 * NS;
            if (c < g.C) dst[which * g.C + c] = v;
        }
    }
}

// one block: partials -> stats[B][C][3] + tail[4], loss[3]
__global__ void __launch_bounds__(256) ce_dice_finalize_kernel(const float* __restrict__ partial, int B, int C, int dice,
                                                                float* __restrict__ stats, float* __restrict__ loss) {
    __shared__ float red[256];
    __shared__ float tail[4];
    const int stride = 3 * C + 4;
    float dsum = 0.f;
    for (int i = threadIdx.x; i < B * C; i += 256) {
        const int b = i / C, c = i - b * C;
        float I = 0.f, P = 0.f, T = 0.f;
        for (int k = 0; k < LS_NBLK; ++k) {
            const float* src = partial + ((int64_t)b * LS_NBLK + k) * stride;
            I += src[c]; P += src[C + c]; T += src[2 * C + c];
        }
        stats[(int64_t)i * 3 + 0] = I; stats[(int64_t)i * 3 + 1] = P; stats[(int64_t)i * 3 + 2] = T;
        float sets = P + T;
        if (sets == 0.f) sets = 2.f * I;
        dsum += (2.f * I + LS_EPS) / (sets + LS_EPS);
    }
    red[threadIdx.x] = dsum;
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int k = 0; k < B * LS_NBLK; ++k) s += partial[(int64_t)k * stride + 3 * C + threadIdx.x];
        tail[threadIdx.x] = s;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* st = stats + (int64_t)B * C * 3;
        st[0] = tail[0]; st[1] = tail[1]; st[2] = tail[2]; st[3] = tail[3];
        const float ce = tail[0] / tail[1];                        // 0/0 = NaN like F.cross_entropy on an all-ignored batch
        const float dl = dice ? 1.f - red[0] / (float)(B * C) : 0.f;
        loss[0] = ce + dl; loss[1] = ce; loss[2] = dl;
    }
}

template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) ce_dice_bwd_kernel(const T* __restrict__ logits, LossGeom g,
                                                                  const int64_t* __restrict__ target, int64_t ignore_index,
                                                                  const float* __restrict__ cw, int dice,
                                                                  const float* __restrict__ stats, const float* __restrict__ grad_out,
                                                                  T* __restrict__ dfull, int64_t ldg) {
    __shared__ float gI[64 * NS], gP[64 * NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const float go = grad_out ? grad_out[0] : 1.f;
    const float invW = 1.f / stats[(int64_t)g.B * g.C * 3 + 1];
    for (int c = threadIdx.x; c < 64 * NS; c += LS_THREADS) {
        float a = 0.f, bb = 0.f;
        if (c < g.C && dice) {
            const float* st = stats + ((int64_t)b * g.C + c) * 3;
            const float I = st[0], P = st[1], Tt = st[2];
            const float sets = P + Tt;
            if (sets != 0.f) {   // sets == 0: d = (2I+eps)/(2I+eps) == 1 -> zero gradient
                const float nbc = 1.f / (float)(g.B * g.C);
                a = -nbc * 2.f / (sets + LS_EPS);                                   // d loss / d I
                bb = nbc * (2.f * I + LS_EPS) / ((sets + LS_EPS) * (sets + LS_EPS));  // d loss / d P
            }
        }
        gI[c] = a; gP[c] = bb;
    }
    __syncthreads();
    const int64_t npix = (int64_t)g.H * g.W;
    const int64_t nw = (int64_t)LS_NBLK * 4;
    const int64_t per = (npix + nw - 1) / nw;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p0 = wid * per, p1 = p0 + per < npix ? p0 + per : npix;
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * npix;
    T* dimg = dfull + (int64_t)b * npix * ldg;
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t t = tg[p];
        T* drow = dimg + p * ldg;
        if (t == ignore_index || t < 0 || t >= g.C) {
#pragma unroll
            for (int s = 0; s < NS; ++s) { const int c = lane + 64 * s; if (c < g.C) stf<T>(drow + c, 0.f); }
            continue;
        }
        const int Y = (int)(p / g.W), X = (int)(p - (int64_t)Y * g.W);
        float z[NS], pr[NS], G[NS];
        pixel_logits<T, NS>(img, g, Y, X, lane, z);
        wave_softmax<NS>(z, pr);
        float dot = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = lane + 64 * s;
            G[s] = gP[64 * s + lane] + (c == (int)t ? gI[64 * s + lane] : 0.f);
            dot += G[s] * pr[s];
        }
        dot = wave_sum(dot);
        const float wce = (cw ? cw[t] : 1.f) * invW;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = lane + 64 * s;
            if (c < g.C) {
                const float dz = pr[s] * (G[s] - dot) + wce * (pr[s] - (c == (int)t ? 1.f : 0.f));
                stf<T>(drow + c, go * dz);
            }
        }
    }
}

extern "C" int64_t segf_ce_dice_stats_floats(int B, int C) {
    return (int64_t)B * C * 3 + 4 + (int64_t)B * LS_NBLK * (3 * C + 4);
}

template <typename T>
static int ce_dice_fwd_launch(int ns, dim3 grid, hipStream_t st, const T* logits, LossGeom g, const int64_t* target,
                              int64_t ignore_index, const float* cw, float* partial) {
    if (ns == 1) hipLaunchKernelGGL((ce_dice_fwd_kernel<T, 1>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index, cw, partial);
    else if (ns == 2) hipLaunchKernelGGL((ce_dice_fwd_kernel<T, 2>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index, cw, partial);
    else hipLaunchKernelGGL((ce_dice_fwd_kernel<T, 3>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index, cw, partial);
    return 0;
}
template <typename T>
static int ce_dice_bwd_launch(int ns, dim3 grid, hipStream_t st, const T* logits, LossGeom g, const int64_t* target,
                              int64_t ignore_index, const float* cw, int dice, const float* stats, const float* grad_out,
                              T* dfull, int64_t ldg) {
    if (ns == 1) hipLaunchKernelGGL((ce_dice_bwd_kernel<T, 1>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index, cw, dice, stats, grad_out, dfull, ldg);
    else if (ns == 2) hipLaunchKernelGGL((ce_dice_bwd_kernel<T, 2>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index, cw, dice, stats, grad_out, dfull, ldg);
    else hipLaunchKernelGGL((ce_dice_bwd_kernel<T, 3>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index, cw, dice, stats, grad_out, dfull, ldg);
    return 0;
}

extern "C" int segf_ce_dice_fwd(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl,
                                const int64_t* target, int64_t ignore_index, const float* class_weight, int dice, float* stats,
                                float* loss, void* stream) {
    if (B <= 0 || C <= 0 || C > 192 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || ldl < C || B > 65535) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    LossGeom g{B, C, h, w, H, W, ldl};
    float* partial = stats + (int64_t)B * C * 3 + 4;
    const int ns = (C + 63) / 64;
    SEGF_DISPATCH_DT(dt, T, { ce_dice_fwd_launch<T>(ns, dim3(LS_NBLK, B), st, (const T*)logits, g, target, ignore_index, class_weight, partial); })
    SEGF_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_dice_finalize_kernel, dim3(1), dim3(256), 0, st, partial, B, C, dice, stats, loss);
    SEGF_CHECK_LAUNCH();
    return 0;
}

extern "C" int segf_ce_dice_bwd(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl,
                                const int64_t* target, int64_t ignore_index, const float* class_weight, int dice,
                                const float* stats, const float* grad_out, void* dlogits_full, int64_t ldg, void* stream) {
    if (B <= 0 || C <= 0 || C > 192 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || ldl < C || ldg < C || B > 65535) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    LossGeom g{B, C, h, w, H, W, ldl};
    const int ns = (C + 63) / 64;
    SEGF_DISPATCH_DT(dt, T, {
        ce_dice_bwd_launch<T>(ns, dim3(LS_NBLK, B), st, (const T*)logits, g, target, ignore_index, class_weight, dice, stats, grad_out, (T*)dlogits_full, ldg);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- fused upsample + argmax + confusion matrix ----------------------------------------------------------------
template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) argmax_confmat_kernel(const T* __restrict__ logits, LossGeom g,
                                                                     const int64_t* __restrict__ target, int64_t ignore_label,
                                                                     unsigned long long* __restrict__ mat,
                                                                     unsigned long long* __restrict__ hist, int* __restrict__ flag,
                                                                     int64_t* __restrict__ pred_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int64_t npix = (int64_t)g.H * g.W;
    const int64_t nw = (int64_t)LS_NBLK * 4;
    const int64_t per = (npix + nw - 1) / nw;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p0 = wid * per, p1 = p0 + per < npix ? p0 + per : npix;
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * npix;
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t t = tg[p];
        const bool in_mat = t >= 0 && t < g.C;
        const bool in_hist = t != ignore_label;
        if (!pred_out && !in_mat && !in_hist) continue;
        const int Y = (int)(p / g.W), X = (int)(p - (int64_t)Y * g.W);
        float z[NS];
        pixel_logits<T, NS>(img, g, Y, X, lane, z);
        float mx = z[0];
#pragma unroll
        for (int s = 1; s < NS; ++s) mx = fmaxf(mx, z[s]);
        mx = wave_max(mx);
        int best = 0x7fffffff;                 // first (lowest) index attaining the maximum, like torch.argmax
#pragma unroll
        for (int s = NS - 1; s >= 0; --s) if (z[s] == mx) best = lane + 64 * s;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(best, o, 64); best = other < best ? other : best; }
        if (lane == 0) {
            if (best >= g.C) best = 0;         // all-NaN row
            if (pred_out) pred_out[(int64_t)b * npix + p] = best;
            if (in_mat) atomicAdd(mat + t * g.C + best, 1ull);
            if (in_hist) {
                if (in_mat) atomicAdd(hist + t * g.C + best, 1ull);
                else atomicOr(flag, 1);        // label >= n that is not ignore_label: the reference's bincount shape check fails
            }
        }
    }
}

template <typename T>
static void argmax_launch(int ns, dim3 grid, hipStream_t st, const T* logits, LossGeom g, const int64_t* target, int64_t ign,
                          unsigned long long* mat, unsigned long long* hist, int* flag, int64_t* pred_out) {
    if (ns == 1) hipLaunchKernelGGL((argmax_confmat_kernel<T, 1>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ign, mat, hist, flag, pred_out);
    else if (ns == 2) hipLaunchKernelGGL((argmax_confmat_kernel<T, 2>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ign, mat, hist, flag, pred_out);
    else hipLaunchKernelGGL((argmax_confmat_kernel<T, 3>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ign, mat, hist, flag, pred_out);
}

extern "C" int segf_argmax_confmat(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl,
                                   const int64_t* target, int64_t ignore_label, int64_t* mat, int64_t* hist, int32_t* flag,
                                   int64_t* pred_out, void* stream) {
    if (B <= 0 || C <= 0 || C > 192 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || ldl < C || B > 65535) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    LossGeom g{B, C, h, w, H, W, ldl};
    const int ns = (C + 63) / 64;
    SEGF_DISPATCH_DT(dt, T, {
        argmax_launch<T>(ns, dim3(LS_NBLK, B), st, (const T*)logits, g, target, ignore_label, (unsigned long long*)mat,
                         (unsigned long long*)hist, (int*)flag, pred_out);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- confusion matrix from explicit (ground truth, prediction) pairs: ConfusionMatrix.update(a, b)
// (util/utils.py:99-109) and the bincount of Metrics.update (util/metrics.py:24-27) ------------------------------
__global__ void confmat_pairs_kernel(const int64_t* __restrict__ gt, const int64_t* __restrict__ pred, int64_t n, int C,
                                     int64_t ignore_label, unsigned long long* __restrict__ mat,
                                     unsigned long long* __restrict__ hist, int* __restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = gt[i], p = pred[i];
        if (p < 0 || p >= C) { atomicOr(flag, 2); continue; }
        const bool in_mat = t >= 0 && t < C;
        if (mat && in_mat) atomicAdd(mat + t * C + p, 1ull);
        if (hist && t != ignore_label) {
            if (in_mat) atomicAdd(hist + t * C + p, 1ull);
            else atomicOr(flag, 1);
        }
    }
}

extern "C" int segf_confmat_pairs(const int64_t* gt, const int64_t* pred, int64_t n, int C, int64_t ignore_label,
                                  int64_t* mat, int64_t* hist, int32_t* flag, void* stream) {
    if (n <= 0) return 0;
    if (C <= 0) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)imin64(cdiv64(n, 256), 2048);
    hipLaunchKernelGGL(confmat_pairs_kernel, dim3(blocks), dim3(256), 0, st, gt, pred, n, C, ignore_label,
                       (unsigned long long*)mat, (unsigned long long*)hist, (int*)flag);
    SEGF_CHECK_LAUNCH();
    return 0;
}
