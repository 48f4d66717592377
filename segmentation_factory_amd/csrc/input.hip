// Device-side input pipeline: the reference's per-sample CPU transforms as gfx950 kernels over decoded
// uint8 images that already live in HBM.
//   train: ExtRandomCrop -> ExtColorJitter -> ExtRandomHorizontalFlip -> ExtToTensor -> ExtNormalize
//          (datasets/build_datasets.py:14-22; datasets/extra_transform.py:319-392, 426-509, 196-214, 259-281, 288-313)
//   val:   ExtResize -> ExtToTensor -> ExtNormalize                      (build_datasets.py:24-29; extra_transform.py:395-419)
// plus the label table + widening of the dataset classes (datasets/ade.py:122-124, cityscapes.py:159, coco_stuff.py:95-100).
// The arithmetic is Pillow's (the transforms act on PIL images): ImagingBlend in C float with truncation / clipping, rgb2l =
// (R*19595 + G*38470 + B*7471 + 0x8000) >> 16, the contrast mean int(sum / count + 0.5) in double, ImagingResample's two
// 22-bit fixed-point passes, ImagingScaleAffine's accumulated nearest index -- all integer / uint8 results are bit-exact, and
// the float tail ((u8 / 255) / 255 - mean) / std (quirk Q11: the second / 255) is four IEEE float32 operations, evaluated once
// per (channel, byte value) into an LDS table.  This file is compiled with -ffp-contract=off: a fused multiply-add anywhere in
// it would change results.  The random draws are made on the host in the reference's order (transforms.py); kernels only see
// their values.  HBM-bound: 4 bytes read, 20 bytes written per output pixel (fp32 NCHW image + int64 label).
#include "common.h"

namespace {

constexpr int OP_BRIGHTNESS = 1, OP_CONTRAST = 2, OP_SATURATION = 3;

__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Blend.c: (UINT8)(deg + alpha * (v - deg)); outside [0, 1] the value is clipped first.  One formula covers alpha == 0 / 1 (exact)
// and the interpolating range (the result lies between deg and v, so the clip never acts there).
__device__ __forceinline__ int blend1(int deg, int v, float a) {
    const float t = __fadd_rn((float)deg, __fmul_rn(a, (float)(v - deg)));
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

// applies the jitter operations of `order` (2 bits per step, 0 = end) to one pixel; stops BEFORE the contrast step when
// `until_contrast` (the luma sum that defines the contrast mean is taken over the image at that point)
template <bool UNTIL_CONTRAST>
__device__ __forceinline__ void jitter(int& r, int& g, int& b, int order, const float (&f)[3], int cmean) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int op = (order >> (2 * k)) & 3;
        if (op == 0) break;
        if (op == OP_CONTRAST && UNTIL_CONTRAST) break;
        const int l = luma(r, g, b);
        const int dr = op == OP_BRIGHTNESS ? 0 : (op == OP_CONTRAST ? cmean : l);
        r = blend1(dr, r, f[k]); g = blend1(dr, g, f[k]); b = blend1(dr, b, f[k]);
    }
}

__device__ __forceinline__ bool has_contrast(int order) {
    return (order & 3) == OP_CONTRAST || ((order >> 2) & 3) == OP_CONTRAST || ((order >> 4) & 3) == OP_CONTRAST;
}

// PX consecutive OUTPUT pixels (x0 .. x0+PX-1 of output row y) of a sample: source pixel of output x is
// (top + y, left + (flip ? W-1-x : x)); outside the source image the crop reads 0 (Image.crop).  The fast path fetches the 12
// image bytes / 4 label bytes of an in-range run as aligned dwords + v_alignbyte (the run starts at an arbitrary byte).
template <int PX, bool WANT_LBL>
__device__ __forceinline__ void load_run(const segf_input_sample& s, int y, int x0, int W, int (&rgb)[PX][3], int (&lb)[PX]) {
    const int sy = s.top + y;
    const int cs = s.flip ? s.left + W - PX - x0 : s.left + x0;          // first SOURCE column of the run (ascending)
    int src[PX][3], sl[PX];
    if (PX == 4 && sy < s.src_h && cs + 4 <= s.src_w) {
        const uintptr_t p = (uintptr_t)s.img + (uintptr_t)sy * s.img_stride + (uintptr_t)cs * 3;
        const uintptr_t base = p & ~(uintptr_t)3, last = (p + 11) & ~(uintptr_t)3;
        const unsigned sh = (unsigned)(p & 3);
        const unsigned w0 = *(const unsigned*)base, w1 = *(const unsigned*)(base + 4), w2 = *(const unsigned*)(base + 8);
        const unsigned w3 = *(const unsigned*)(base + 12 <= last ? base + 12 : last);     // never past the dword of the last byte
        const unsigned d0 = __builtin_amdgcn_alignbyte(w1, w0, sh), d1 = __builtin_amdgcn_alignbyte(w2, w1, sh),
                       d2 = __builtin_amdgcn_alignbyte(w3, w2, sh);
        src[0][0] = d0 & 255; src[0][1] = (d0 >> 8) & 255; src[0][2] = (d0 >> 16) & 255;
        src[1][0] = d0 >> 24; src[1][1] = d1 & 255; src[1][2] = (d1 >> 8) & 255;
        src[2][0] = (d1 >> 16) & 255; src[2][1] = d1 >> 24; src[2][2] = d2 & 255;
        src[3][0] = (d2 >> 8) & 255; src[3][1] = (d2 >> 16) & 255; src[3][2] = d2 >> 24;
        if (WANT_LBL) {
            const uintptr_t q = (uintptr_t)s.lbl + (uintptr_t)sy * s.lbl_stride + (uintptr_t)cs;
            const uintptr_t qb = q & ~(uintptr_t)3, ql = (q + 3) & ~(uintptr_t)3;
            const unsigned l0 = *(const unsigned*)qb, l1 = *(const unsigned*)(qb + 4 <= ql ? qb + 4 : ql);
            const unsigned d = __builtin_amdgcn_alignbyte(l1, l0, (unsigned)(q & 3));
            sl[0] = d & 255; sl[1] = (d >> 8) & 255; sl[2] = (d >> 16) & 255; sl[3] = d >> 24;
        }
    } else {
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const int sx = cs + k;
            const bool in = sy < s.src_h && sx < s.src_w && sx >= 0;
            const uint8_t* p = s.img + (int64_t)(in ? sy : 0) * s.img_stride + (int64_t)(in ? sx : 0) * 3;
            const int r = p[0], g = p[1], b = p[2];
            src[k][0] = in ? r : 0; src[k][1] = in ? g : 0; src[k][2] = in ? b : 0;
            if (WANT_LBL) {
                const int l = s.lbl[(int64_t)(in ? sy : 0) * s.lbl_stride + (in ? sx : 0)];
                sl[k] = in ? l : 0;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        const int j = s.flip ? PX - 1 - k : k;
        rgb[k][0] = src[j][0]; rgb[k][1] = src[j][1]; rgb[k][2] = src[j][2];
        lb[k] = WANT_LBL ? sl[j] : 0;
    }
}

// lsum[b] = sum over the crop of luma(pixel after the jitter steps that precede the contrast step): ImageStat.Stat(convert('L')).sum
template <int PX>
__global__ void __launch_bounds__(256) input_lsum_kernel(const segf_input_sample* __restrict__ samples, int H, int W,
                                                         unsigned long long* __restrict__ lsum) {
    const segf_input_sample s = samples[blockIdx.y];
    if (!has_contrast(s.order)) return;
    const int runs_per_row = W / PX, total = H * runs_per_row;
    const float f[3] = {s.factor[0], s.factor[1], s.factor[2]};
    unsigned acc = 0;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int y = idx / runs_per_row, x0 = (idx - y * runs_per_row) * PX;
        int rgb[PX][3], lb[PX];
        load_run<PX, false>(s, y, x0, W, rgb, lb);
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            jitter<true>(rgb[k][0], rgb[k][1], rgb[k][2], s.order, f, 0);
            acc += (unsigned)luma(rgb[k][0], rgb[k][1], rgb[k][2]);
        }
    }
    unsigned long long a64 = acc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a64 += __shfl_xor(a64, o, 64);
    if ((threadIdx.x & 63) == 0 && a64) atomicAdd(&lsum[blockIdx.y], a64);     // integer sum: order-independent, exact
}

// the float tail of one (channel, byte value): ExtToTensor's v / 255, ExtNormalize's / 255 again, (t - mean) / std
__device__ __forceinline__ float norm_value(int v, float mean, float sd) {
    return __fdiv_rn(__fsub_rn(__fdiv_rn(__fdiv_rn((float)v, 255.f), 255.f), mean), sd);
}

template <int PX>
__global__ void __launch_bounds__(256) input_train_kernel(const segf_input_sample* __restrict__ samples, int H, int W,
                                                          const unsigned long long* __restrict__ lsum,
                                                          const float* __restrict__ mean, const float* __restrict__ stdv,
                                                          const int64_t* __restrict__ lut, float* __restrict__ out_img,
                                                          int64_t* __restrict__ out_lbl) {
    __shared__ float nrm[3][256];
    __shared__ int64_t llut[256];
    for (int i = threadIdx.x; i < 768; i += 256) nrm[i >> 8][i & 255] = norm_value(i & 255, mean[i >> 8], stdv[i >> 8]);
    llut[threadIdx.x] = lut ? lut[threadIdx.x] : (int64_t)threadIdx.x;
    const int b = blockIdx.y;
    const segf_input_sample s = samples[b];
    // ImageEnhance.Contrast: mean = int(sum / count + 0.5), Python floats = IEEE doubles
    const int cmean = has_contrast(s.order) ? (int)((double)lsum[b] / (double)((int64_t)H * W) + 0.5) : 0;
    const float f[3] = {s.factor[0], s.factor[1], s.factor[2]};
    __syncthreads();
    const int runs_per_row = W / PX, total = H * runs_per_row;
    const int64_t plane = (int64_t)H * W;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int y = idx / runs_per_row, x0 = (idx - y * runs_per_row) * PX;
        int rgb[PX][3], lb[PX];
        load_run<PX, true>(s, y, x0, W, rgb, lb);
        float o[3][PX];
        int64_t ol[PX];
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            jitter<false>(rgb[k][0], rgb[k][1], rgb[k][2], s.order, f, cmean);
            o[0][k] = nrm[0][rgb[k][0]]; o[1][k] = nrm[1][rgb[k][1]]; o[2][k] = nrm[2][rgb[k][2]];
            ol[k] = llut[lb[k]];
        }
        const int64_t pix = (int64_t)y * W + x0;
        float* oi = out_img + (int64_t)b * 3 * plane + pix;
        int64_t* olp = out_lbl + (int64_t)b * plane + pix;
        if (PX == 4) {
#pragma unroll
            for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(oi + c * plane) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
            *reinterpret_cast<longlong2*>(olp) = make_longlong2(ol[0], ol[1]);
            *reinterpret_cast<longlong2*>(olp + 2) = make_longlong2(ol[2], ol[3]);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) oi[c * plane] = o[c][0];
            olp[0] = ol[0];
        }
    }
}

// ---- validation: Pillow's resize ----------------------------------------------------------------------------------------------
constexpr int PRECISION_BITS = 32 - 8 - 2;

__host__ __device__ inline int resample_ksize(int in_size, int out_size) {
    double fs = (double)in_size / (double)out_size;
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(fs) * 2 + 1;
}

// Resample.c precompute_coeffs + normalize_coeffs_8bpc (bilinear filter, support 1) for output index xx
__device__ void coeffs_for(int xx, int in_size, int out_size, int ksize, int* __restrict__ bounds, int* __restrict__ kk) {
    const double scale = (double)in_size / (double)out_size;
    const double fs = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * fs, ss = 1.0 / fs;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double a = (x + xmin - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        ww += a < 1.0 ? 1.0 - a : 0.0;
    }
    for (int x = 0; x < ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w = a < 1.0 ? 1.0 - a : 0.0;
            if (ww != 0.0) w /= ww;
        }
        kk[(int64_t)xx * ksize + x] = w < 0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
}

// Geometry.c ImagingScaleAffine (nearest): xo = a * 0.5; per output pixel xin = (int)xo; xo += a  (accumulated in double)
__device__ void nearest_table(int n_in, int n_out, int* __restrict__ tab) {
    const double a = (double)n_in / (double)n_out;
    double xo = a * 0.5;
    for (int i = 0; i < n_out; ++i) {
        int v = (int)xo;
        tab[i] = v < 0 ? 0 : (v >= n_in ? n_in - 1 : v);
        xo += a;
    }
}

struct ValTables { int *hb, *hk, *vb, *vk, *nx, *ny; uint8_t* tmp; int ksw, ksh; };

__host__ __device__ inline int64_t val_tables(void* ws, int src_h, int src_w, int out_h, int out_w, ValTables* t) {
    const int ksw = resample_ksize(src_w, out_w), ksh = resample_ksize(src_h, out_h);
    int64_t n = 0;
    int* base = (int*)ws;
    int* hb = base + n; n += 2 * (int64_t)out_w;
    int* hk = base + n; n += (int64_t)out_w * ksw;
    int* vb = base + n; n += 2 * (int64_t)out_h;
    int* vk = base + n; n += (int64_t)out_h * ksh;
    int* nx = base + n; n += out_w;
    int* ny = base + n; n += out_h;
    n = (n + 3) & ~(int64_t)3;
    if (t) { t->hb = hb; t->hk = hk; t->vb = vb; t->vk = vk; t->nx = nx; t->ny = ny; t->tmp = (uint8_t*)(base + n); t->ksw = ksw; t->ksh = ksh; }
    return n * 4 + (int64_t)src_h * out_w * 3;
}

__global__ void __launch_bounds__(256) input_val_tables_kernel(int src_h, int src_w, int out_h, int out_w, ValTables t) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < out_w) coeffs_for(g, src_w, out_w, t.ksw, t.hb, t.hk);
    if (g < out_h) coeffs_for(g, src_h, out_h, t.ksh, t.vb, t.vk);
    if (g == 0) nearest_table(src_w, out_w, t.nx);
    if (g == 1) nearest_table(src_h, out_h, t.ny);
}

// horizontal pass: tmp[y][xx][c] = clip8((2^21 + sum_k src[y][xmin + k][c] * coef[xx][k]) >> 22)
__global__ void __launch_bounds__(256) input_val_hpass_kernel(const uint8_t* __restrict__ img, int64_t img_stride, int src_h, int out_w,
                                                              ValTables t) {
    const int64_t total = (int64_t)src_h * out_w;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int y = (int)(idx / out_w), xx = (int)(idx - (int64_t)y * out_w);
        const int xmin = t.hb[2 * xx], n = t.hb[2 * xx + 1];
        const int* k = t.hk + (int64_t)xx * t.ksw;
        const uint8_t* p = img + (int64_t)y * img_stride + (int64_t)xmin * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < n; ++x) { const int kv = k[x]; s0 += p[3 * x] * kv; s1 += p[3 * x + 1] * kv; s2 += p[3 * x + 2] * kv; }
        uint8_t* o = t.tmp + idx * 3;
        s0 >>= PRECISION_BITS; s1 >>= PRECISION_BITS; s2 >>= PRECISION_BITS;
        o[0] = (uint8_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
        o[1] = (uint8_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
        o[2] = (uint8_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
    }
}

// vertical pass + float tail + nearest-resized label through the table
__global__ void __launch_bounds__(256) input_val_vpass_kernel(const uint8_t* __restrict__ lbl, int64_t lbl_stride, int out_h, int out_w,
                                                              ValTables t, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                              const int64_t* __restrict__ lut, float* __restrict__ out_img,
                                                              int64_t* __restrict__ out_lbl) {
    __shared__ float nrm[3][256];
    for (int i = threadIdx.x; i < 768; i += 256) nrm[i >> 8][i & 255] = norm_value(i & 255, mean[i >> 8], stdv[i >> 8]);
    __syncthreads();
    const int64_t total = (int64_t)out_h * out_w;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int yy = (int)(idx / out_w), xx = (int)(idx - (int64_t)yy * out_w);
        const int ymin = t.vb[2 * yy], n = t.vb[2 * yy + 1];
        const int* k = t.vk + (int64_t)yy * t.ksh;
        const uint8_t* p = t.tmp + ((int64_t)ymin * out_w + xx) * 3;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int y = 0; y < n; ++y) {
            const int kv = k[y];
            const uint8_t* q = p + (int64_t)y * out_w * 3;
            s0 += q[0] * kv; s1 += q[1] * kv; s2 += q[2] * kv;
        }
        s0 >>= PRECISION_BITS; s1 >>= PRECISION_BITS; s2 >>= PRECISION_BITS;
        s0 = s0 < 0 ? 0 : (s0 > 255 ? 255 : s0); s1 = s1 < 0 ? 0 : (s1 > 255 ? 255 : s1); s2 = s2 < 0 ? 0 : (s2 > 255 ? 255 : s2);
        out_img[idx] = nrm[0][s0]; out_img[total + idx] = nrm[1][s1]; out_img[2 * total + idx] = nrm[2][s2];
        const int l = lbl[(int64_t)t.ny[yy] * lbl_stride + t.nx[xx]];
        out_lbl[idx] = lut ? lut[l] : (int64_t)l;
    }
}

}  // namespace

extern "C" int segf_input_train(const segf_input_sample* samples, int B, int H, int W, uint64_t* lsum_ws, const float* mean3,
                                const float* std3, const int64_t* label_lut, float* out_img, int64_t* out_lbl, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (!samples || !lsum_ws || !mean3 || !std3 || !out_img || !out_lbl) return SEGF_ERR_SHAPE;
    if ((int64_t)H * W > (int64_t)1 << 30) return SEGF_ERR_SHAPE;
    if (((uintptr_t)out_img | (uintptr_t)out_lbl | (uintptr_t)lsum_ws | (uintptr_t)samples) % 16 != 0) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const bool px4 = W % 4 == 0;
    const int64_t runs = (int64_t)H * (px4 ? W / 4 : W);
    // enough workgroups to fill 256 CUs several times over even at batch 1; grid.y = sample
    const int bx = (int)imin64(cdiv64(runs, 256), B >= 64 ? 64 : 1024);
    int rc = segf_zero(lsum_ws, (int64_t)B * 8, stream);
    if (rc) return rc;
    if (px4) {
        hipLaunchKernelGGL(input_lsum_kernel<4>, dim3(bx, B), dim3(256), 0, st, samples, H, W, (unsigned long long*)lsum_ws);
        hipLaunchKernelGGL(input_train_kernel<4>, dim3(bx, B), dim3(256), 0, st, samples, H, W, (const unsigned long long*)lsum_ws,
                           mean3, std3, label_lut, out_img, out_lbl);
    } else {
        hipLaunchKernelGGL(input_lsum_kernel<1>, dim3(bx, B), dim3(256), 0, st, samples, H, W, (unsigned long long*)lsum_ws);
        hipLaunchKernelGGL(input_train_kernel<1>, dim3(bx, B), dim3(256), 0, st, samples, H, W, (const unsigned long long*)lsum_ws,
                           mean3, std3, label_lut, out_img, out_lbl);
    }
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- single-image inference (estimate_model.py:85-97): T.Resize((nH, nW)) on the uint8 CHW tensor + x / 255 + Normalize -----------
// torchvision 0.15.2 (environment.yml:22) resizes a TENSOR through torch.nn.functional.interpolate(bilinear, align_corners=False,
// antialias=False) on the float32 cast, rounds (half to even) and casts back to uint8.  ATen's arithmetic (UpSampleKernel.cpp):
// scale = in / out in float; src = scale * (dst + 0.5) - 0.5 clamped at 0; i0 = (int) src, l1 = src - i0, l0 = 1 - l1,
// i1 = min(i0 + 1, in - 1); out = wy0 * (wx0 * v00 + wx1 * v01) + wy1 * (wx0 * v10 + wx1 * v11) -- IEEE float32, no contraction
// (this file is built with -ffp-contract=off).  Then (u8 / 255 - mean) / std.
__global__ void __launch_bounds__(256) infer_preprocess_kernel(const uint8_t* __restrict__ img, int src_h, int src_w, int out_h, int out_w,
                                                               float sy, float sx, const float* __restrict__ mean3,
                                                               const float* __restrict__ std3, float* __restrict__ out) {
    const int64_t n = (int64_t)out_h * out_w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int oy = (int)(i / out_w), ox = (int)(i - (int64_t)oy * out_w);
        float fy = __fsub_rn(__fmul_rn(sy, __fadd_rn((float)oy, 0.5f)), 0.5f), fx = __fsub_rn(__fmul_rn(sx, __fadd_rn((float)ox, 0.5f)), 0.5f);
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        int y0 = (int)fy, x0 = (int)fx;
        y0 = y0 > src_h - 1 ? src_h - 1 : y0; x0 = x0 > src_w - 1 ? src_w - 1 : x0;
        const int y1 = y0 + 1 < src_h ? y0 + 1 : src_h - 1, x1 = x0 + 1 < src_w ? x0 + 1 : src_w - 1;
        float ly1 = __fsub_rn(fy, (float)y0), lx1 = __fsub_rn(fx, (float)x0);
        ly1 = ly1 < 0.f ? 0.f : (ly1 > 1.f ? 1.f : ly1); lx1 = lx1 < 0.f ? 0.f : (lx1 > 1.f ? 1.f : lx1);
        const float ly0 = __fsub_rn(1.f, ly1), lx0 = __fsub_rn(1.f, lx1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint8_t* p = img + (int64_t)c * src_h * src_w;
            const float v00 = (float)p[(int64_t)y0 * src_w + x0], v01 = (float)p[(int64_t)y0 * src_w + x1];
            const float v10 = (float)p[(int64_t)y1 * src_w + x0], v11 = (float)p[(int64_t)y1 * src_w + x1];
            const float top = __fadd_rn(__fmul_rn(lx0, v00), __fmul_rn(lx1, v01)), bot = __fadd_rn(__fmul_rn(lx0, v10), __fmul_rn(lx1, v11));
            float v = __fadd_rn(__fmul_rn(ly0, top), __fmul_rn(ly1, bot));
            v = rintf(v);                                                  // torch.round: half to even
            v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);                   // (the uint8 cast)
            out[(int64_t)c * n + i] = __fdiv_rn(__fsub_rn(__fdiv_rn(v, 255.f), mean3[c]), std3[c]);
        }
    }
}
extern "C" int segf_infer_preprocess(const uint8_t* img, int src_h, int src_w, int out_h, int out_w, const float* mean3, const float* std3,
                                     float* out, void* stream) {
    if (src_h <= 0 || src_w <= 0 || out_h <= 0 || out_w <= 0 || !img || !mean3 || !std3 || !out) return SEGF_ERR_SHAPE;
    const float sy = (float)src_h / (float)out_h, sx = (float)src_w / (float)out_w;
    const int blocks = (int)imin64(cdiv64((int64_t)out_h * out_w, 256), 4096);
    hipLaunchKernelGGL(infer_preprocess_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, src_h, src_w, out_h, out_w, sy, sx,
                       mean3, std3, out);
    SEGF_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t segf_input_val_ws(int src_h, int src_w, int out_h, int out_w) {
    if (src_h <= 0 || src_w <= 0 || out_h <= 0 || out_w <= 0) return 0;
    return val_tables(nullptr, src_h, src_w, out_h, out_w, nullptr);
}

extern "C" int segf_input_val(const uint8_t* img, int64_t img_stride, const uint8_t* lbl, int64_t lbl_stride, int src_h, int src_w,
                              int out_h, int out_w, void* ws, const float* mean3, const float* std3, const int64_t* label_lut,
                              float* out_img, int64_t* out_lbl, void* stream) {
    if (src_h <= 0 || src_w <= 0 || out_h <= 0 || out_w <= 0) return SEGF_ERR_SHAPE;
    if (!img || !lbl || !ws || !mean3 || !std3 || !out_img || !out_lbl) return SEGF_ERR_SHAPE;
    if ((uintptr_t)ws % 16 != 0) return SEGF_ERR_WORKSPACE;
    if (img_stride < (int64_t)src_w * 3 || lbl_stride < src_w) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    ValTables t;
    val_tables(ws, src_h, src_w, out_h, out_w, &t);
    const int nt = out_w > out_h ? out_w : out_h;
    hipLaunchKernelGGL(input_val_tables_kernel, dim3((int)cdiv64(nt < 2 ? 2 : nt, 256)), dim3(256), 0, st, src_h, src_w, out_h, out_w, t);
    hipLaunchKernelGGL(input_val_hpass_kernel, dim3((int)imin64(cdiv64((int64_t)src_h * out_w, 256), 8192)), dim3(256), 0, st, img,
                       img_stride, src_h, out_w, t);
    hipLaunchKernelGGL(input_val_vpass_kernel, dim3((int)imin64(cdiv64((int64_t)out_h * out_w, 256), 8192)), dim3(256), 0, st, lbl,
                       lbl_stride, out_h, out_w, t, mean3, std3, label_lut, out_img, out_lbl);
    SEGF_CHECK_LAUNCH();
    return 0;
}
