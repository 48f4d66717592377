// Fused optimizer step over a flat fp32 parameter buffer: adaptive gradient clipping (unit-wise, AGC) + AdamW.
// Stands in for what engine.py:52-53 triggers through timm 0.9.2's NativeScaler: dispatch_clip_grad(mode='agc',
// value=0.02) followed by optimizer.step() of the AdamW that create_optimizer builds (train_gpu.py:99-102,269-270).
// timm is not installed in the build image, so this arithmetic is restated from timm 0.9.2 / torch.optim.AdamW
// semantics and pinned only by hand-derived known-answer tests ("parity unpinned"):
//   unit = one output row of a >=2-D weight (dim 0), or the whole tensor for 1-D parameters
//   max_norm = max(||p_unit||, agc_eps) * clip_factor;  g <- g * max_norm / max(||g_unit||, 1e-6)  if ||g_unit|| >= max_norm
//   p <- p * (1 - lr * wd);  m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;
//   p <- p - lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// unit_flags: bit 0 = weight decay applies, bit 1 = the unit's parameter received no gradient this step (skipped entirely).
// One wave per unit: two streaming passes over the unit (norms, then update); 7 x 4 B per parameter of HBM traffic.
#include "common.h"

__global__ void __launch_bounds__(256) agc_adamw_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         const int64_t* __restrict__ unit_off, const int32_t* __restrict__ unit_len,
                                                         const uint8_t* __restrict__ unit_flags, int32_t* __restrict__ unit_step,
                                                         int nunits, float lr, float b1, float b2, float eps, float wd, float bc1_,
                                                         float bc2_sqrt_, float clip_factor, float agc_eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = wave_global; u < nunits; u += nwaves) {
        const int64_t off = unit_off[u];
        const int len = unit_len[u];
        const int uflags = unit_flags[u];
        if (uflags & 2) continue;          // no gradient this step: torch.optim.AdamW skips `p.grad is None` (no decay, no moments)
        float bc1 = bc1_, bc2_sqrt = bc2_sqrt_;
        if (unit_step) {                   // per-parameter step count, as torch keeps it (state['step']): a skipped step does not count
            const int t = unit_step[u] + 1;      // every lane reads before lane 0 writes (same wave, program order)
            bc1 = 1.f - powf(b1, (float)t);
            bc2_sqrt = sqrtf(1.f - powf(b2, (float)t));
            if (lane == 0) unit_step[u] = t;
        }
        float gscale = 1.f;
        if (clip_factor > 0.f) {
            float pn = 0.f, gn = 0.f;
            for (int i = lane; i < len; i += 64) {
                const float p = param[off + i], g = grad[off + i];
                pn = fmaf(p, p, pn); gn = fmaf(g, g, gn);
            }
            pn = sqrtf(wave_sum(pn)); gn = sqrtf(wave_sum(gn));
            const float max_norm = fmaxf(pn, agc_eps) * clip_factor;
            if (!(gn < max_norm)) gscale = max_norm / fmaxf(gn, 1e-6f);
        }
        const float decay = (uflags & 1) ? 1.f - lr * wd : 1.f;
        const float step = lr / bc1;
        for (int i = lane; i < len; i += 64) {
            const float g = grad[off + i] * gscale;
            float p = param[off + i] * decay;
            const float mm = b1 * m[off + i] + (1.f - b1) * g;
            const float vv = b2 * v[off + i] + (1.f - b2) * g * g;
            p -= step * mm / (sqrtf(vv) / bc2_sqrt + eps);
            param[off + i] = p; m[off + i] = mm; v[off + i] = vv;
        }
    }
}

extern "C" int segf_agc_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* unit_offset,
                              const int32_t* unit_len, const uint8_t* unit_flags, int32_t* unit_step, int nunits, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step, float clip_factor, float agc_eps,
                              void* stream) {
    if (nunits <= 0) return 0;
    if (step < 1 && !unit_step) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    const int blocks = imin((nunits + 3) / 4, 4096);
    hipLaunchKernelGGL(agc_adamw_kernel, dim3(blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, unit_offset, unit_len,
                       unit_flags, unit_step, nunits, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, clip_factor, agc_eps);
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- the other two --clip-mode values of the reference (train_gpu.py:99-102 -> timm.utils.dispatch_clip_grad) over the flat
// gradient buffer, device-only (no host read of the norm, graph-capturable):
//   'norm'  = torch.nn.utils.clip_grad_norm_(parameters, value, norm_type=2.0): g *= min(1, value / (||g||_2 + 1e-6))
//   'value' = torch.nn.utils.clip_grad_value_(parameters, value):               g = clamp(g, -value, value)
// The norm is a fixed-order reduction (per-workgroup partials over fixed slices, then every workgroup re-reduces the partials in
// the same order in double): bitwise reproducible, and independent of how the buffer is cut into parameters.
constexpr int CLIP_BLOCKS = 1024;

__global__ void __launch_bounds__(256) grad_sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    float acc = 0.f;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) { const float v = g[i]; acc = fmaf(v, v, acc); }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) grad_scale_by_norm_kernel(float* __restrict__ g, int64_t n, const float* __restrict__ partial,
                                                                  int nparts, float max_norm) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float total = (float)sqrt(red[0]);
    const float coef = fminf(max_norm / (total + 1e-6f), 1.f);
    if (coef >= 1.f) return;
    const int64_t per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) g[i] *= coef;
}

__global__ void __launch_bounds__(256) grad_clamp_kernel(float* __restrict__ g, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) g[i] = fminf(fmaxf(g[i], -v), v);
}

extern "C" int64_t segf_clip_grad_ws(void) { return CLIP_BLOCKS; }

extern "C" int segf_clip_grad(float* grad, int64_t n, int mode, float value, float* ws, void* stream) {
    if (n <= 0) return 0;
    if (!grad || (mode != 0 && mode != 1) || !(value >= 0.f)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)imin64(cdiv64(n, 4096), CLIP_BLOCKS);
    if (mode == 0) {
        if (!ws) return SEGF_ERR_WORKSPACE;
        hipLaunchKernelGGL(grad_sumsq_kernel, dim3(blocks), dim3(256), 0, st, grad, n, ws);
        hipLaunchKernelGGL(grad_scale_by_norm_kernel, dim3(blocks), dim3(256), 0, st, grad, n, ws, blocks, value);
    } else {
        hipLaunchKernelGGL(grad_clamp_kernel, dim3(blocks), dim3(256), 0, st, grad, n, value);
    }
    SEGF_CHECK_LAUNCH();
    return 0;
}
