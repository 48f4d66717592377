// Bilinear resize on NHWC (F.interpolate mode='bilinear'): reference models/heads/segformer.py:48,
// models/modules/ppm.py:24 (align_corners=True), models/heads/upernet.py:41,46, models/build_models.py:65.
// Forward = gather of 4 taps; backward = gather-form transpose (each input pixel sums the output pixels that
// reference it -- no atomics, deterministic).  Outputs may be channel slices of a wider concat buffer (ld*).
#include "colreduce.h"

// fuse_map.hip: the transposed 1/2-1/4-1/8 resizes on the matrix pipe (bf16, C % 128 == 0)
int fuse_map_bwd_supported(int dt, int B, int H, int W, int C);
int fuse_map_bwd_launch(int B, int H, int W, int C, const void* dy, int64_t ldo, void* d2, void* d4, void* d8, hipStream_t st);

template <typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ in, int64_t ldi, T* __restrict__ out, int64_t ldo, int B, int h, int w,
                                    int C, int H, int W, int ac, bool vec) {
    const int nch = (C + 7) / 8;
    const int64_t total = (int64_t)B * H * W * nch;
    // two copies of the loop (all chunks full and aligned / general): in the first the four tap loads are plain vector loads
    // in flight together; a per-load guard inside the loop would serialise them (branch + s_waitcnt per load)
    auto body = [&](const bool full) {
        for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
            const int ch = (int)(idx % nch);
            int64_t t = idx / nch;
            const int X = (int)(t % W); t /= W;
            const int Y = (int)(t % H);
            const int64_t b = t / H;
            const int c0 = ch * 8, nv = full ? 8 : (C - c0 < 8 ? C - c0 : 8);
            const bool vv = full ? true : vec;
            int y0, y1, x0, x1; float ly, lx;
            bilinear_src(Y, h, H, ac, y0, y1, ly);
            bilinear_src(X, w, W, ac, x0, x1, lx);
            float v00[8], v01[8], v10[8], v11[8], o[8];
            const T* base = in + b * h * w * ldi + c0;
            load8_guard<T>(base + ((int64_t)y0 * w + x0) * ldi, nv, vv, v00);
            load8_guard<T>(base + ((int64_t)y0 * w + x1) * ldi, nv, vv, v01);
            load8_guard<T>(base + ((int64_t)y1 * w + x0) * ldi, nv, vv, v10);
            load8_guard<T>(base + ((int64_t)y1 * w + x1) * ldi, nv, vv, v11);
            const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = w00 * v00[j] + w01 * v01[j] + w10 * v10[j] + w11 * v11[j];
            store8_guard<T>(out + ((b * H + Y) * W + X) * ldo + c0, nv, vv, o);
        }
    };
    if (vec && C % 8 == 0) body(true); else body(false);
}

// out = base + sum_k bilinear_up(src_k): the folded SegFormerHead (heads/segformer.py:44-56).  Because bilinear
// interpolation is linear and its weights sum to one, conv1x1(cat(up(Linear_i(x_i)))) == sum_i up(x_i (F_i W_i)^T) + sum_i F_i b_i;
// the per-scale products are formed at their native resolution and this kernel adds them on the stride-4 grid.
struct UpSrc { const void* p; int h, w; int64_t ld; };
template <typename T>
__global__ void __launch_bounds__(256) upsample_add_kernel(const T* __restrict__ base, int64_t ldb, UpSrc s0, UpSrc s1, UpSrc s2,
                                                            int nsrc, T* __restrict__ out, int64_t ldo, int B, int H, int W, int C,
                                                            int ac) {
    const int nch = C / 8;
    const int64_t total = (int64_t)B * H * W * nch;
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % nch);
        int64_t t = idx / nch;
        const int X = (int)(t % W); t /= W;
        const int Y = (int)(t % H);
        const int64_t b = t / H;
        const int c0 = ch * 8;
        float acc[8];
        load8<T>(base + ((b * H + Y) * W + X) * ldb + c0, acc);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k >= nsrc) break;
            const UpSrc s = k == 0 ? s0 : (k == 1 ? s1 : s2);
            int y0, y1, x0, x1; float ly, lx;
            bilinear_src(Y, s.h, H, ac, y0, y1, ly);
            bilinear_src(X, s.w, W, ac, x0, x1, lx);
            const T* sb = reinterpret_cast<const T*>(s.p) + b * s.h * s.w * s.ld + c0;
            float v00[8], v01[8], v10[8], v11[8];
            load8<T>(sb + ((int64_t)y0 * s.w + x0) * s.ld, v00);
            load8<T>(sb + ((int64_t)y0 * s.w + x1) * s.ld, v01);
            load8<T>(sb + ((int64_t)y1 * s.w + x0) * s.ld, v10);
            load8<T>(sb + ((int64_t)y1 * s.w + x1) * s.ld, v11);
            const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w00 * v00[j] + w01 * v01[j] + w10 * v10[j] + w11 * v11[j];
        }
        store8<T>(out + ((b * H + Y) * W + X) * ldo + c0, acc);
    }
}

// The SegFormer head's case: three sources at exactly 1/2, 1/4 and 1/8 of the output grid (align_corners=False).  The generic
// kernel issues 13 16-byte loads per 16-byte store and is bound by the L1 / address path, not by HBM.  Here a thread owns a
// strip of four output pixels (X = 4k .. 4k+3) of one row: their horizontal taps are the static sets {2k-1..2k+2},
// {k-1..k+1} and {(k-1)>>1, +1} with literal weights, the vertical blend is done once per tap column, and the strip needs
// 22 loads for 4 stores.  Clamped border taps reproduce bilinear_src exactly (both taps of a clamped pair coincide).
template <typename T, int NT>
__device__ __forceinline__ void load_vblend(const T* __restrict__ sb, const UpSrc& s, int y0, int y1, float ly, int xfirst,
                                            float (&tv)[NT][8]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int x = xfirst + j;
        x = x < 0 ? 0 : (x > s.w - 1 ? s.w - 1 : x);
        float a[8], b[8];
        load8<T>(sb + ((int64_t)y0 * s.w + x) * s.ld, a);
        load8<T>(sb + ((int64_t)y1 * s.w + x) * s.ld, b);
#pragma unroll
        for (int c = 0; c < 8; ++c) tv[j][c] = fmaf(ly, b[c] - a[c], a[c]);
    }
}
// STATS: the kernel also returns the per-channel sum and sum of squares of what it stores (the BatchNorm statistics of the
// following ConvModule, heads/segformer.py:21-29), as per-workgroup partials [blk][2][C] for colreduce_finalize: the separate
// statistics pass over the [B*H*W, C] tensor disappears.  Column-fixed threads (the grid makes the thread count a multiple of
// the chunk count), sums over the ROUNDED stored values, fixed-order reduction inside the workgroup.
template <typename T, bool STATS>
__global__ void __launch_bounds__(256) upsample_add_248_kernel(const T* __restrict__ base, int64_t ldb, UpSrc s0, UpSrc s1, UpSrc s2,
                                                                T* __restrict__ out, int64_t ldo, int B, int H, int W, int C,
                                                                float* __restrict__ partial) {
    __shared__ float red[STATS ? 256 * 16 : 1];
    // Channel groups: blockIdx.y owns `nch` = C / 8 / gridDim.y consecutive 16-byte chunks (128 bytes per pixel when there are 8
    // of them) and walks the whole (image, row, strip) plane for them before the next group is dispatched.  The rows of the three
    // sources that neighbouring output rows share are then 128 bytes per pixel wide and stay in one XCD's L2 between their uses;
    // with all C channels per unit the sources were re-fetched 2.5x (PMC FETCH_SIZE: 10.1 GB per launch against 7.5 GB algorithmic).
    const int nch = C / 8 / (int)gridDim.y, nst = W / 4;
    const int64_t units = (int64_t)B * H * nst;
    const int64_t g = (int64_t)xcd_block() * 256 + threadIdx.x;
    const int64_t ustep = ((int64_t)gridDim.x * 256) / nch;
    const int ch = (int)blockIdx.y * nch + (int)(g % nch);
    float sm1[8], sm2[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { sm1[c] = 0.f; sm2[c] = 0.f; }
    for (int64_t u = g / nch; u < units; u += ustep) {
        int64_t t = u;
        const int k = (int)(t % nst); t /= nst;
        const int Y = (int)(t % H);
        const int64_t b = t / H;
        const int c0 = ch * 8;
        const int64_t pix0 = (b * H + Y) * W + 4 * k;
        float acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) load8<T>(base + (pix0 + i) * ldb + c0, acc[i]);
        int y0, y1; float ly;
        {   // 1/2-resolution source
            bilinear_src(Y, s0.h, H, 0, y0, y1, ly);
            float tv[4][8];
            load_vblend<T, 4>(reinterpret_cast<const T*>(s0.p) + b * s0.h * s0.w * s0.ld + c0, s0, y0, y1, ly, 2 * k - 1, tv);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc[0][c] += 0.25f * tv[0][c] + 0.75f * tv[1][c];
                acc[1][c] += 0.75f * tv[1][c] + 0.25f * tv[2][c];
                acc[2][c] += 0.25f * tv[1][c] + 0.75f * tv[2][c];
                acc[3][c] += 0.75f * tv[2][c] + 0.25f * tv[3][c];
            }
        }
        {   // 1/4
            bilinear_src(Y, s1.h, H, 0, y0, y1, ly);
            float tv[3][8];
            load_vblend<T, 3>(reinterpret_cast<const T*>(s1.p) + b * s1.h * s1.w * s1.ld + c0, s1, y0, y1, ly, k - 1, tv);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc[0][c] += 0.375f * tv[0][c] + 0.625f * tv[1][c];
                acc[1][c] += 0.125f * tv[0][c] + 0.875f * tv[1][c];
                acc[2][c] += 0.875f * tv[1][c] + 0.125f * tv[2][c];
                acc[3][c] += 0.625f * tv[1][c] + 0.375f * tv[2][c];
            }
        }
        {   // 1/8: strip k lies inside one source cell
            bilinear_src(Y, s2.h, H, 0, y0, y1, ly);
            float tv[2][8];
            load_vblend<T, 2>(reinterpret_cast<const T*>(s2.p) + b * s2.h * s2.w * s2.ld + c0, s2, y0, y1, ly, (k - 1) >> 1, tv);
            const float l0 = (k & 1) ? 0.0625f : 0.5625f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float lx = l0 + 0.125f * i;
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[i][c] += fmaf(lx, tv[1][c] - tv[0][c], tv[0][c]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            store8<T>(out + (pix0 + i) * ldo + c0, acc[i]);
            if (STATS) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float r = sizeof(T) == 2 ? bf2f(f2bf(acc[i][c])) : acc[i][c];
                    sm1[c] += r; sm2[c] = fmaf(r, r, sm2[c]);
                }
            }
        }
    }
    if (STATS) {
#pragma unroll
        for (int c = 0; c < 8; ++c) { red[threadIdx.x * 16 + c] = sm1[c]; red[threadIdx.x * 16 + 8 + c] = sm2[c]; }
        __syncthreads();
        if ((int)threadIdx.x < nch && (int)threadIdx.x < 256) {   // first thread of each chunk in this workgroup adds the later ones in order
            float a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = red[threadIdx.x * 16 + c];
            for (int o = threadIdx.x + nch; o < 256; o += nch)
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] += red[o * 16 + c];
            float* dst = partial + (int64_t)blockIdx.x * 2 * C + ch * 8;
#pragma unroll
            for (int c = 0; c < 8; ++c) { dst[c] = a[c]; dst[C + c] = a[8 + c]; }
        }
    }
}

static bool upsample_add_is_248(int H, int W, int C, int nsrc, int h0, int w0, int h1, int w1, int h2, int w2, int align_corners) {
    return nsrc == 3 && !align_corners && W % 8 == 0 && H % 8 == 0 && h0 * 2 == H && w0 * 2 == W && h1 * 4 == H && w1 * 4 == W &&
           h2 * 8 == H && w2 * 8 == W && C % 8 == 0 && !POL(upadd_generic);
}
static int upsample_add_248_groups(int C) {        // channel groups of 8 chunks (128 bytes per pixel) when C allows it
    const int nch = C / 8;
    // measured on MI355X (cfg2, batch 128): 128-byte channel groups leave the launch at 1.77 ms (4.2 TB/s) with or without them and
    // slow the fused-statistics variant (1.99 -> 2.26 ms: 8 instead of 96 reducing threads per workgroup), i.e. the 2.5x source
    // re-fetches that FETCH_SIZE reports are served by the Infinity Cache and are not what bounds the kernel: opt-in only
    (void)nch;
    return 1;
}
static int upsample_add_248_blocks(int B, int H, int W, int C) {
    return colfixed_blocks((int64_t)B * H * (W / 4), C / 8 / upsample_add_248_groups(C), 2, 16384);
}
static int upsample_add_impl(int dt, int B, int H, int W, int C, const void* base, int64_t ldb, int nsrc,
                             const void* src0, int h0, int w0, int64_t ld0, const void* src1, int h1, int w1, int64_t ld1,
                             const void* src2, int h2, int w2, int64_t ld2, void* out, int64_t ldo, int align_corners,
                             float* partial, void* stream);
extern "C" int segf_upsample_add(int dt, int B, int H, int W, int C, const void* base, int64_t ldb, int nsrc,
                                 const void* src0, int h0, int w0, int64_t ld0, const void* src1, int h1, int w1, int64_t ld1,
                                 const void* src2, int h2, int w2, int64_t ld2, void* out, int64_t ldo, int align_corners,
                                 void* stream) {
    return upsample_add_impl(dt, B, H, W, C, base, ldb, nsrc, src0, h0, w0, ld0, src1, h1, w1, ld1, src2, h2, w2, ld2, out, ldo,
                             align_corners, nullptr, stream);
}
// Same, plus sums[2][C] = per-channel (sum, sum of squares) of `out` over all B*H*W rows (BatchNorm statistics of the consumer);
// ws >= segf_upsample_add_stats_ws(...) floats.  Returns SEGF_ERR_SHAPE when the geometry is not the 1/2-1/4-1/8 case
// (the caller then runs segf_bn_stats on the result).
extern "C" int64_t segf_upsample_add_stats_ws(int B, int H, int W, int C) {
    return (int64_t)upsample_add_248_blocks(B, H, W, C) * 2 * C;
}
extern "C" int segf_upsample_add_stats(int dt, int B, int H, int W, int C, const void* base, int64_t ldb, int nsrc,
                                       const void* src0, int h0, int w0, int64_t ld0, const void* src1, int h1, int w1, int64_t ld1,
                                       const void* src2, int h2, int w2, int64_t ld2, void* out, int64_t ldo, int align_corners,
                                       float* sums, float* ws, void* stream) {
    if (!sums || !ws) return SEGF_ERR_WORKSPACE;
    if (!upsample_add_is_248(H, W, C, nsrc, h0, w0, h1, w1, h2, w2, align_corners) || C / 8 > 256) return SEGF_ERR_SHAPE;
    const int rc = upsample_add_impl(dt, B, H, W, C, base, ldb, nsrc, src0, h0, w0, ld0, src1, h1, w1, ld1, src2, h2, w2, ld2, out,
                                     ldo, align_corners, ws, stream);
    if (rc) return rc;
    colreduce_finalize_launch(ws, upsample_add_248_blocks(B, H, W, C), 2 * (int64_t)C, sums, (hipStream_t)stream);
    SEGF_CHECK_LAUNCH();
    return 0;
}
static int upsample_add_impl(int dt, int B, int H, int W, int C, const void* base, int64_t ldb, int nsrc,
                             const void* src0, int h0, int w0, int64_t ld0, const void* src1, int h1, int w1, int64_t ld1,
                             const void* src2, int h2, int w2, int64_t ld2, void* out, int64_t ldo, int align_corners,
                             float* partial, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    if (nsrc < 0 || nsrc > 3 || C % 8 != 0 || ldb < C || ldo < C) return SEGF_ERR_SHAPE;
    const int64_t esz = dt == SEGF_BF16 ? 2 : 4;
    const void* ptrs[5] = {base, out, src0, src1, src2};
    const int64_t lds[5] = {ldb, ldo, ld0, ld1, ld2};
    for (int i = 0; i < 2 + nsrc; ++i)
        if (!ptrs[i] || ((uintptr_t)ptrs[i] % 16) || ((lds[i] * esz) % 16) || lds[i] < C) return SEGF_ERR_SHAPE;
    const int hs[3] = {h0, h1, h2}, ws[3] = {w0, w1, w2};
    for (int i = 0; i < nsrc; ++i)
        if (hs[i] <= 0 || ws[i] <= 0) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    UpSrc s0{src0, h0, w0, ld0}, s1{src1, h1, w1, ld1}, s2{src2, h2, w2, ld2};
    if (upsample_add_is_248(H, W, C, nsrc, h0, w0, h1, w1, h2, w2, align_corners)) {
        const dim3 grid4((unsigned)upsample_add_248_blocks(B, H, W, C), (unsigned)upsample_add_248_groups(C));
        SEGF_DISPATCH_DT(dt, T, {
            if (partial) hipLaunchKernelGGL((upsample_add_248_kernel<T, true>), grid4, dim3(256), 0, st, (const T*)base, ldb,
                                            s0, s1, s2, (T*)out, ldo, B, H, W, C, partial);
            else hipLaunchKernelGGL((upsample_add_248_kernel<T, false>), grid4, dim3(256), 0, st, (const T*)base, ldb, s0,
                                    s1, s2, (T*)out, ldo, B, H, W, C, partial);
        })
        SEGF_CHECK_LAUNCH();
        return 0;
    }
    if (partial) return SEGF_ERR_SHAPE;
    const int blocks = (int)imin64(cdiv64((int64_t)B * H * W * (C / 8), 256), 16384);
    SEGF_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL((upsample_add_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)base, ldb, s0, s1, s2, nsrc, (T*)out,
                           ldo, B, H, W, C, align_corners);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// candidate output range [lo, hi] whose taps can touch input index i (checked exactly inside the loop)
__device__ __forceinline__ void out_range(int i, int in, int out, int ac, int& lo, int& hi) {
    const float inv = ac ? (in > 1 ? (float)(out - 1) / (float)(in - 1) : 0.f) : (float)out / (float)in;
    float a, b;
    if (ac) { a = (i - 1) * inv; b = (i + 1) * inv; }
    else { a = (i - 1 + 0.5f) * inv - 0.5f; b = (i + 1 + 0.5f) * inv - 0.5f; }
    lo = (int)floorf(a) - 1; hi = (int)ceilf(b) + 1;
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
    if (in == 1 || (ac && in <= 1)) { lo = 0; hi = out - 1; }
}

template <typename T>
__global__ void bilinear_bwd_kernel(T* __restrict__ din, int64_t ldi, const T* __restrict__ dout, int64_t ldo, int B, int h, int w,
                                    int C, int H, int W, int ac, bool vec) {
    const int nch = (C + 7) / 8;
    const int64_t total = (int64_t)B * h * w * nch;
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % nch);
        int64_t t = idx / nch;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h);
        const int64_t b = t / h;
        const int c0 = ch * 8, nv = C - c0 < 8 ? C - c0 : 8;
        int Ylo, Yhi, Xlo, Xhi;
        out_range(y, h, H, ac, Ylo, Yhi);
        out_range(x, w, W, ac, Xlo, Xhi);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int Y = Ylo; Y <= Yhi; ++Y) {
            int y0, y1; float ly;
            bilinear_src(Y, h, H, ac, y0, y1, ly);
            const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int X = Xlo; X <= Xhi; ++X) {
                int x0, x1; float lx;
                bilinear_src(X, w, W, ac, x0, x1, lx);
                const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
                if (wx == 0.f) continue;
                float v[8];
                load8_guard<T>(dout + ((b * H + Y) * W + X) * ldo + c0, nv, vec, v);
                const float ww = wy * wx;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(ww, v[j], acc[j]);
            }
        }
        store8_guard<T>(din + ((b * h + y) * w + x) * ldi + c0, nv, vec, acc);
    }
}

// Integer-ratio fast path of the transposed resize (H == R*h, W == R*w, align_corners=False; R = 2, 4, 8 cover every
// resize of the SegFormer / UPerNet heads at 512^2): the <= 2R x 2R window of output pixels that touch an input pixel and
// its separable weights are known per thread up front (weights come from the same bilinear_src as the forward, so the
// clamped borders are exact); the inner loop is load + fma only.
template <typename T, int R>
__global__ void __launch_bounds__(256) bilinear_bwd_int_kernel(T* __restrict__ din, int64_t ldi, const T* __restrict__ dout,
                                                                int64_t ldo, int B, int h, int w, int C) {
    constexpr int WIN = 2 * R;
    const int H = R * h, W = R * w, off = R / 2;
    const int nch = C / 8;
    const int64_t total = (int64_t)B * h * w * nch;
    // thread order (image, channel group of 8 chunks = 128 B, y, x, chunk in group): the rows of output pixels that share
    // gradient pixels then touch 2R x W x 128 B between reuses, which stays in one XCD's L2 (with all C channels per pixel the
    // window of the R = 8 case is 4.7 MB and every tap row re-fetched its gradient rows: FETCH_SIZE 2.3x the tensor)
    const int G = nch % 8 == 0 ? 8 : (nch % 4 == 0 ? 4 : 1), ngrp = nch / G;
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int chl = (int)(idx % G);
        int64_t t = idx / G;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h); t /= h;
        const int grp = (int)(t % ngrp);
        const int64_t b = t / ngrp;
        const int c0 = (grp * G + chl) * 8;
        const int Y0 = y == 0 ? 0 : R * (y - 1) + off, X0 = x == 0 ? 0 : R * (x - 1) + off;
        float wx[WIN];
#pragma unroll
        for (int j = 0; j < WIN; ++j) {
            const int X = X0 + j;
            int x0, x1; float lx;
            bilinear_src(X < W ? X : W - 1, w, W, 0, x0, x1, lx);
            wx[j] = X < W ? (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f) : 0.f;
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        const T* src = dout + (b * H * W) * ldo + c0;
        // every load is unconditional (out-of-window rows / columns are clamped and carry weight 0): a row's 2R loads are
        // in flight together; a `skip if weight == 0` branch around each load would end in s_waitcnt vmcnt(0) per load
        for (int i = 0; i < WIN; ++i) {
            const int Y = Y0 + i, Yc = Y < H ? Y : H - 1;
            int y0, y1; float ly;
            bilinear_src(Yc, h, H, 0, y0, y1, ly);
            const float wy = Y < H ? (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f) : 0.f;
            const T* row = src + (int64_t)Yc * W * ldo;
            Raw8<T> raw[WIN];
#pragma unroll
            for (int j = 0; j < WIN; ++j) raw[j] = load8_raw<T>(row + (int64_t)(X0 + j < W ? X0 + j : W - 1) * ldo);
            SEGF_LOADS_ISSUED();
#pragma unroll
            for (int j = 0; j < WIN; ++j) {
                float v[8];
                unpack8(raw[j], v);
                const float ww = wy * wx[j];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = fmaf(ww, v[q], acc[q]);
            }
        }
        store8<T>(din + ((b * h + y) * w + x) * ldi + c0, acc);
    }
}

extern "C" int segf_bilinear_fwd(int dt, int B, int h, int w, int C, const void* in, int64_t ldi, int H, int W, void* out,
                                 int64_t ldo, int align_corners, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    if (h <= 0 || w <= 0 || ldi < C || ldo < C) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)imin64(cdiv64((int64_t)B * H * W * ((C + 7) / 8), 256), 8192);
    SEGF_DISPATCH_DT(dt, T, {
        const bool vec = vec_ok_host<T>(in, ldi) && vec_ok_host<T>(out, ldo);
        hipLaunchKernelGGL((bilinear_fwd_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)in, ldi, (T*)out, ldo, B, h, w, C, H, W, align_corners, vec);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

extern "C" int segf_bilinear_bwd(int dt, int B, int h, int w, int C, void* din, int64_t ldi, int H, int W, const void* dout,
                                 int64_t ldo, int align_corners, void* stream) {
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0) return 0;
    if (H <= 0 || W <= 0 || ldi < C || ldo < C) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)imin64(cdiv64((int64_t)B * h * w * ((C + 7) / 8), 256), 16384);
    const int R = (H % h == 0 && W % w == 0 && H / h == W / w) ? H / h : 0;
    SEGF_DISPATCH_DT(dt, T, {
        const bool vec = vec_ok_host<T>(din, ldi) && vec_ok_host<T>(dout, ldo);
        if (vec && !align_corners && C % 8 == 0 && (R == 2 || R == 4 || R == 8)) {
            if (R == 2) hipLaunchKernelGGL((bilinear_bwd_int_kernel<T, 2>), dim3(blocks), dim3(256), 0, st, (T*)din, ldi, (const T*)dout, ldo, B, h, w, C);
            else if (R == 4) hipLaunchKernelGGL((bilinear_bwd_int_kernel<T, 4>), dim3(blocks), dim3(256), 0, st, (T*)din, ldi, (const T*)dout, ldo, B, h, w, C);
            else hipLaunchKernelGGL((bilinear_bwd_int_kernel<T, 8>), dim3(blocks), dim3(256), 0, st, (T*)din, ldi, (const T*)dout, ldo, B, h, w, C);
            SEGF_CHECK_LAUNCH();
            return 0;
        }
        hipLaunchKernelGGL((bilinear_bwd_kernel<T>), dim3(blocks), dim3(256), 0, st, (T*)din, ldi, (const T*)dout, ldo, B, h, w, C, H, W, align_corners, vec);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// materialised full-resolution logits for API parity with SegmentationModel.forward (build_models.py:62-66):
// NHWC low-res -> fp32 NCHW full-res.  Thread per (b, c, Y, X) with X fastest (coalesced NCHW stores; taps hit L2).
template <typename T>
__global__ void bilinear_to_nchw_kernel(const T* __restrict__ in, int64_t ldi, float* __restrict__ out, int B, int h, int w, int C,
                                        int H, int W) {
    const int64_t total = (int64_t)B * C * H * W;
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int X = (int)(idx % W);
        int64_t t = idx / W;
        const int Y = (int)(t % H); t /= H;
        const int c = (int)(t % C);
        const int64_t b = t / C;
        int y0, y1, x0, x1; float ly, lx;
        bilinear_src(Y, h, H, 0, y0, y1, ly);
        bilinear_src(X, w, W, 0, x0, x1, lx);
        const T* base = in + b * h * w * ldi + c;
        const float v00 = ldf<T>(base + ((int64_t)y0 * w + x0) * ldi), v01 = ldf<T>(base + ((int64_t)y0 * w + x1) * ldi);
        const float v10 = ldf<T>(base + ((int64_t)y1 * w + x0) * ldi), v11 = ldf<T>(base + ((int64_t)y1 * w + x1) * ldi);
        out[idx] = bilinear_aten(v00, v01, v10, v11, ly, lx);
    }
}

extern "C" int segf_bilinear_to_nchw_f32(int dt, int B, int h, int w, int C, const void* in, int64_t ldi, int H, int W, float* out,
                                         void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    if (h <= 0 || w <= 0 || ldi < C) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)imin64(cdiv64((int64_t)B * C * H * W, 256), 16384);
    SEGF_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL((bilinear_to_nchw_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)in, ldi, out, B, h, w, C, H, W);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}


// ---- AdaptiveAvgPool2d on NHWC (PPM, models/modules/ppm.py:13; bins [floor(i*in/out), ceil((i+1)*in/out)) ) ---------------
template <typename T, bool BWD>   // BWD: din[pix] = sum over bins containing pix of dout[bin] / |bin|
__global__ void adaptive_pool_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int H, int W, int C, int S) {
    const int nch = C / 8;
    const int64_t total = BWD ? (int64_t)B * H * W * nch : (int64_t)B * S * S * nch;
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(idx % nch) * 8;
        int64_t t = idx / nch;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (!BWD) {
            const int ox = (int)(t % S); t /= S;
            const int oy = (int)(t % S);
            const int64_t b = t / S;
            const int y0 = (oy * H) / S, y1 = ((oy + 1) * H + S - 1) / S, x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    float v[8];
                    load8<T>(in + ((b * H + y) * W + x) * C + c0, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[j];
                }
            const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] *= inv;
            store8<T>(out + ((b * S + oy) * S + ox) * C + c0, acc);
        } else {
            const int x = (int)(t % W); t /= W;
            const int y = (int)(t % H);
            const int64_t b = t / H;
            for (int oy = 0; oy < S; ++oy) {
                const int y0 = (oy * H) / S, y1 = ((oy + 1) * H + S - 1) / S;
                if (y < y0 || y >= y1) continue;
                for (int ox = 0; ox < S; ++ox) {
                    const int x0 = (ox * W) / S, x1 = ((ox + 1) * W + S - 1) / S;
                    if (x < x0 || x >= x1) continue;
                    float v[8];
                    load8<T>(in + ((b * S + oy) * S + ox) * C + c0, v);     // in = dout [B][S][S][C]
                    const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(inv, v[j], acc[j]);
                }
            }
            store8<T>(out + ((b * H + y) * W + x) * C + c0, acc);
        }
    }
}
// bwd == 0: in [B][H][W][C] -> out [B][S][S][C];  bwd == 1: in = dout [B][S][S][C] -> out = din [B][H][W][C]
extern "C" int segf_adaptive_avgpool(int dt, int bwd, int B, int H, int W, int C, int S, const void* in, void* out, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || S <= 0) return 0;
    if (C <= 0 || C % 8 || ((uintptr_t)in % 16) || ((uintptr_t)out % 16)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (bwd ? (int64_t)B * H * W : (int64_t)B * S * S) * (C / 8);
    const int blocks = (int)imin64(cdiv64(total, 256), 4096);
    SEGF_DISPATCH_DT(dt, T, {
        if (bwd) hipLaunchKernelGGL((adaptive_pool_kernel<T, true>), dim3(blocks), dim3(256), 0, st, (const T*)in, (T*)out, B, H, W, C, S);
        else hipLaunchKernelGGL((adaptive_pool_kernel<T, false>), dim3(blocks), dim3(256), 0, st, (const T*)in, (T*)out, B, H, W, C, S);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- nearest-neighbour upsampling on NHWC (F.interpolate mode='nearest', heads/fpn.py:31,35): the top-down step of FPNHead,
// `out = nearest(out, size of lateral)`, `out = out + lateral`, `out = nearest(out, x2)`.
// Forward (bwd=0): out[b][Y][X] = in[b][src(Y)][src(X)] (+ base[b][Y][X] when base != NULL).  Integer ratios (H % h == 0): src(Y) =
// Y / (H / h).  Any other pair of sizes (GEN; inputs that are not multiples of 32 give FPNHead 3 x 3 -> 5 x 6, 10 x 12 -> 9 x 12
// steps, fpn.py:30-31) follows ATen's nearest_neighbor_compute_source_index in its float arithmetic: src(Y) = min((int)floorf(Y *
// scale), h - 1) with scale = (float)h / H.
// Backward (bwd=1): out[b][y][x] = sum of in[b][Y][X] over the (contiguous) destination rows / columns whose source is (y, x) --
// gather form, deterministic.
__device__ __forceinline__ int nearest_src(int d, float scale, int n_in) {
    const int s = (int)floorf((float)d * scale);
    return s < n_in - 1 ? s : n_in - 1;
}
// first destination index in [0, n_out] whose source is >= y (n_out when there is none): a short scan around y / scale
__device__ __forceinline__ int nearest_first_dst(int y, float scale, int n_in, int n_out) {
    int d = (int)floorf((float)y / scale) - 1;
    d = d < 0 ? 0 : (d > n_out ? n_out : d);
    while (d > 0 && nearest_src(d - 1, scale, n_in) >= y) --d;
    while (d < n_out && nearest_src(d, scale, n_in) < y) ++d;
    return d;
}

template <typename T, bool BWD, bool GEN>
__global__ void __launch_bounds__(256) nearest_up_kernel(const T* __restrict__ in, const T* __restrict__ base, T* __restrict__ out,
                                                          int B, int h, int w, int C, int H, int W, float sy, float sx) {
    const int nch = C / 8, ry = H / h, rx = W / w;
    const int oh = BWD ? h : H, ow = BWD ? w : W;
    const int64_t total = (int64_t)B * oh * ow * nch;
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(idx % nch) * 8;
        int64_t t = idx / nch;
        const int X = (int)(t % ow); t /= ow;
        const int Y = (int)(t % oh);
        const int64_t b = t / oh;
        float acc[8];
        if (!BWD) {
            const int ys = GEN ? nearest_src(Y, sy, h) : Y / ry, xs = GEN ? nearest_src(X, sx, w) : X / rx;
            load8<T>(in + ((b * h + ys) * w + xs) * C + c0, acc);
            if (base) {
                float v[8];
                load8<T>(base + ((b * H + Y) * W + X) * C + c0, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            const int y0 = GEN ? nearest_first_dst(Y, sy, h, H) : Y * ry, y1 = GEN ? nearest_first_dst(Y + 1, sy, h, H) : Y * ry + ry;
            const int x0 = GEN ? nearest_first_dst(X, sx, w, W) : X * rx, x1 = GEN ? nearest_first_dst(X + 1, sx, w, W) : X * rx + rx;
            for (int yy = y0; yy < y1; ++yy)
                for (int xx = x0; xx < x1; ++xx) {
                    float v[8];
                    load8<T>(in + ((b * H + yy) * W + xx) * C + c0, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += v[j];
                }
        }
        store8<T>(out + ((b * oh + Y) * ow + X) * C + c0, acc);
    }
}

extern "C" int segf_nearest_up(int dt, int bwd, int B, int h, int w, int C, int H, int W, const void* in, const void* base,
                               void* out, void* stream) {
    if (B <= 0 || h <= 0 || w <= 0 || C <= 0) return 0;
    if (C % 8 != 0 || H <= 0 || W <= 0) return SEGF_ERR_SHAPE;
    if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)base) % 16 != 0) return SEGF_ERR_SHAPE;
    if (bwd && base) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = (int64_t)B * (bwd ? h : H) * (bwd ? w : W) * (C / 8);
    const int blocks = (int)imin64(cdiv64(total, 256), 8192);
    const bool gen = (H % h != 0) || (W % w != 0);        // (a size that shrinks, 10 x 12 -> 9 x 12, is never a multiple)
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    SEGF_DISPATCH_DT(dt, T, {
        if (gen) {
            if (bwd) hipLaunchKernelGGL((nearest_up_kernel<T, true, true>), dim3(blocks), dim3(256), 0, st, (const T*)in, (const T*)nullptr, (T*)out, B, h, w, C, H, W, sy, sx);
            else hipLaunchKernelGGL((nearest_up_kernel<T, false, true>), dim3(blocks), dim3(256), 0, st, (const T*)in, (const T*)base, (T*)out, B, h, w, C, H, W, sy, sx);
        } else {
            if (bwd) hipLaunchKernelGGL((nearest_up_kernel<T, true, false>), dim3(blocks), dim3(256), 0, st, (const T*)in, (const T*)nullptr, (T*)out, B, h, w, C, H, W, sy, sx);
            else hipLaunchKernelGGL((nearest_up_kernel<T, false, false>), dim3(blocks), dim3(256), 0, st, (const T*)in, (const T*)base, (T*)out, B, h, w, C, H, W, sy, sx);
        }
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- the three transposed resizes of the folded SegFormerHead in ONE pass over the gradient ---------------------------------
// dy [B][H][W][C] -> d2 [B][H/2][W/2][C], d4 [B][H/4][W/4][C], d8 [B][H/8][W/8][C]: the transposes of the bilinear
// (align_corners=False) x2 / x4 / x8 upsamplings that segf_upsample_add sums in the forward (heads/segformer.py:44-56 folded).
// Three segf_bilinear_bwd launches each stream the full [B*H*W, C] gradient (3 x 3.2 GB at cfg2, batch 128); here a thread owns
// ONE x8 output pixel and reads its 16 x 16 window of dy once; that window contains the complete windows of the 2 x 2 x4-outputs
// and the 4 x 4 x2-outputs nested in it (x4 output 2y+a: window rows [8y+4a-2, 8y+4a+6); x2 output 4y+a: rows [8y+2a-1, 8y+2a+3)),
// so every (gradient pixel, output) pair is accumulated exactly once.  The static supports are those of interior pixels; the
// weights are evaluated at run time with the forward's source-index arithmetic (bilinear_src), so that the clamped image borders
// -- where the true support is a subset of the static one -- come out exact.  Rows are reduced separably: 16 loads in flight,
// x-weighted row sums per nested output, then one y-weighted update per output row.
__device__ __forceinline__ float tap_w(int P, int inR, int out, int po) {     // weight of gradient row / column P for output po
    if (P < 0 || P >= out) return 0.f;
    int p0, p1; float l;
    bilinear_src(P, inR, out, 0, p0, p1, l);
    return (p0 == po ? 1.f - l : 0.f) + (p1 == po ? l : 0.f);
}
// four channels per thread (8-byte accesses for bf16): with eight the nested accumulators (1 + 4 + 16 outputs) plus a row of loads
// need ~300 VGPRs and spill; with four the kernel fits 256 without scratch and the loads of a whole window row stay in flight
template <typename T> struct Raw4;
template <> struct Raw4<float> { float4 a; };
template <> struct Raw4<bf16_t> { uint2 u; };
__device__ __forceinline__ Raw4<float> load4_raw(const float* p) { Raw4<float> r; r.a = *reinterpret_cast<const float4*>(p); return r; }
__device__ __forceinline__ Raw4<bf16_t> load4_raw(const bf16_t* p) { Raw4<bf16_t> r; r.u = *reinterpret_cast<const uint2*>(p); return r; }
__device__ __forceinline__ void unpack4(const Raw4<float>& r, float (&v)[4]) { v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; }
__device__ __forceinline__ void unpack4(const Raw4<bf16_t>& r, float (&v)[4]) {
    v[0] = __uint_as_float(r.u.x << 16); v[1] = __uint_as_float(r.u.x & 0xffff0000u);
    v[2] = __uint_as_float(r.u.y << 16); v[3] = __uint_as_float(r.u.y & 0xffff0000u);
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
    uint2 u; u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ void unpack4v(const Raw4<float>& r, f32x2_t (&v)[2]) { v[0] = f32x2_t{r.a.x, r.a.y}; v[1] = f32x2_t{r.a.z, r.a.w}; }
__device__ __forceinline__ void unpack4v(const Raw4<bf16_t>& r, f32x2_t (&v)[2]) {
    v[0] = f32x2_t{__uint_as_float(r.u.x << 16), __uint_as_float(r.u.x & 0xffff0000u)};
    v[1] = f32x2_t{__uint_as_float(r.u.y << 16), __uint_as_float(r.u.y & 0xffff0000u)};
}
template <typename T> __device__ __forceinline__ void store4v(T* p, const f32x2_t (&v)[2]) {
    const float f[4] = {v[0].x, v[0].y, v[1].x, v[1].y};
    store4(p, f);
}
template <typename T>
__global__ void __launch_bounds__(256, 2) bilinear_bwd_248_kernel(const T* __restrict__ dy, int64_t ldo, T* __restrict__ d2, T* __restrict__ d4,
                                                                T* __restrict__ d8, int B, int H, int W, int C, int no_flip,
                                                                int ring_only) {
    constexpr int CW = 4;
    const int h8 = H / 8, w8 = W / 8, h4 = H / 4, w4 = W / 4, h2 = H / 2, w2 = W / 2;
    const int nch = C / CW;
    const int G = nch % 16 == 0 ? 16 : (nch % 8 == 0 ? 8 : 2), ngrp = nch / G;    // 128-byte channel groups (see bilinear_bwd_int_kernel)
    // ring_only: the x8 pixels (= 8 x 8 blocks of the gradient) on the image border only -- the interior is done on the matrix pipe
    // (fuse_map.hip: fuse_map_bwd_kernel), the border keeps this kernel's clamped-index weight tables.  Ring position r of an
    // image: r < w8: (0, r); r < 2 w8: (h8 - 1, r - w8); then pairs (1 + q / 2, q odd ? w8 - 1 : 0).
    const int nring = 2 * w8 + 2 * (h8 - 2);
    const int64_t total = ring_only ? (int64_t)B * nring * nch : (int64_t)B * h8 * w8 * nch;
    // Weight tables for the four border variants of an x8 pixel along one axis (first, interior, last, the only one): the weights
    // of all interior pixels coincide (translation invariance), so 4 small tables per axis cover every thread of the launch and
    // the 48 column weights need no registers.
    // Columns, per variant: [0,16) x8 weights of the 16 window columns, [16,32) x4 outputs a = 0, 1 (8 columns each, from window
    // column 2 + 4a), [32,48) x2 outputs a = 0..3 (4 columns each, from window column 3 + 2a).
    // Rows, per variant and window row i: {x8, x4 a = 0, 1, x2 a = 0..3, pad}: 0 where row i is outside the output's support.
    __shared__ float wxt[4][48];
    __shared__ float wyt[4][16][8];
    for (int e = threadIdx.x; e < 4 * 48 + 4 * 16 * 8; e += blockDim.x) {
        if (e < 192) {
            const int var = e / 48, k = e % 48;
            const int xr = var == 0 ? 0 : (var == 1 ? (w8 > 2 ? 1 : 0) : (var == 2 ? w8 - 1 : 0));     // a representative pixel
            const int X0r = 8 * xr - 4;
            float wv;
            if (k < 16) wv = tap_w(X0r + k, w8, W, xr);
            else if (k < 32) { const int a = (k - 16) >> 3, j = (k - 16) & 7; wv = tap_w(X0r + 2 + 4 * a + j, w4, W, 2 * xr + a); }
            else { const int a = (k - 32) >> 2, j = (k - 32) & 3; wv = tap_w(X0r + 3 + 2 * a + j, w2, W, 4 * xr + a); }
            wxt[var][k] = wv;
        } else {
            const int f = e - 192, var = f / 128, i = (f % 128) / 8, k = f % 8;
            const int yr = var == 0 ? 0 : (var == 1 ? (h8 > 2 ? 1 : 0) : (var == 2 ? h8 - 1 : 0));
            const int Y = 8 * yr - 4 + i;
            float wv = 0.f;
            if (k == 0) wv = tap_w(Y, h8, H, yr);
            else if (k < 3) { const int a = k - 1; if (i >= 2 + 4 * a && i < 10 + 4 * a) wv = tap_w(Y, h4, H, 2 * yr + a); }
            else if (k < 7) { const int a = k - 3; if (i >= 3 + 2 * a && i < 7 + 2 * a) wv = tap_w(Y, h2, H, 4 * yr + a); }
            wyt[var][i][k] = wv;
        }
    }
    __syncthreads();
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int chl = (int)(idx % G);
        int64_t t = idx / G;
        int x8, y8;
        if (ring_only) {
            const int r = (int)(t % nring); t /= nring;
            if (r < w8) { y8 = 0; x8 = r; }
            else if (r < 2 * w8) { y8 = h8 - 1; x8 = r - w8; }
            else { const int q = r - 2 * w8; y8 = 1 + (q >> 1); x8 = (q & 1) ? w8 - 1 : 0; }
        } else {
            x8 = (int)(t % w8); t /= w8;
            y8 = (int)(t % h8); t /= h8;
        }
        const int grp = (int)(t % ngrp);
        const int64_t b = t / ngrp;
        const int c0 = (grp * G + chl) * CW;
        const int Y0 = 8 * y8 - 4, X0 = 8 * x8 - 4;
        const float* wtab = wxt[w8 == 1 ? 3 : (x8 == 0 ? 0 : (x8 == w8 - 1 ? 2 : 1))];
        const float* wytab = &wyt[h8 == 1 ? 3 : (y8 == 0 ? 0 : (y8 == h8 - 1 ? 2 : 1))][0][0];
        // 32-bit element offsets of the 16 (clamped) window columns relative to a gradient row
        int xoff[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int X = X0 + j;
            xoff[j] = (X < 0 ? 0 : (X >= W ? W - 1 : X)) * (int)ldo;
        }
        // accumulators on channel PAIRS (v_pk_fma_f32: two channels per issue slot)
        f32x2_t a8[2], a4[2][2][2], a2[4][4][2];
        const f32x2_t zero2 = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            a8[q] = zero2;
#pragma unroll
            for (int k = 0; k < 4; ++k) a4[k >> 1][k & 1][q] = zero2;
#pragma unroll
            for (int k = 0; k < 16; ++k) a2[k >> 2][k & 3][q] = zero2;
        }
        const T* src = dy + (b * H * W) * ldo + c0;
        // the row loop is NOT unrolled (16 x 16 hoisted addresses would not fit the register file): rows outside an output's
        // support simply carry weight 0 in its y-update
        // Odd x8 rows walk their window bottom-up: an x8 row shares its upper 8 window rows with the row above and its lower 8
        // with the row below (other workgroups, dispatched next to this one on the same XCD); with opposite walking directions
        // both neighbours touch a shared band during the same half of their loops, which keeps it in the 4 MB L2 between the two
        // reads (all top-down the reuse distance was 8 row steps = 8 MB streamed per XCD: FETCH_SIZE 1.7x the gradient)
        const bool up = (y8 & 1) && !no_flip;
#pragma unroll 1
        for (int ii = 0; ii < 16; ++ii) {
            const int i = up ? 15 - ii : ii;
            const int Y = Y0 + i, Yc = Y < 0 ? 0 : (Y >= H ? H - 1 : Y);
            const T* row = src + (int64_t)Yc * W * ldo;
            Raw4<T> raw[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) raw[j] = load4_raw(row + xoff[j]);
            SEGF_LOADS_ISSUED();
            const float* wt = wtab;
            const float* wy = wytab + 8 * i;
            asm volatile("" : "+v"(wt));          // keep the table reads inside the row loop (hoisted they cost 48 VGPRs)
            // x-weighted row sums: one for the x8 output, two for the x4 outputs, four for the x2 outputs (static column supports)
            f32x2_t r8[2], r4[2][2], r2[4][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                r8[q] = zero2; r4[0][q] = zero2; r4[1][q] = zero2; r2[0][q] = zero2; r2[1][q] = zero2; r2[2][q] = zero2; r2[3][q] = zero2;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                f32x2_t v[2];
                unpack4v(raw[j], v);
                const float w8j = wt[j];
#pragma unroll
                for (int q = 0; q < 2; ++q) r8[q] = v[q] * w8j + r8[q];
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    if (j >= 2 + 4 * a && j < 10 + 4 * a) {
                        const float ww = wt[16 + 8 * a + j - 2 - 4 * a];
#pragma unroll
                        for (int q = 0; q < 2; ++q) r4[a][q] = v[q] * ww + r4[a][q];
                    }
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    if (j >= 3 + 2 * a && j < 7 + 2 * a) {
                        const float ww = wt[32 + 4 * a + j - 3 - 2 * a];
#pragma unroll
                        for (int q = 0; q < 2; ++q) r2[a][q] = v[q] * ww + r2[a][q];
                    }
            }
            // y-weighted updates (the table holds 0 for rows outside an output's support and outside the image)
            const float4 wya = *reinterpret_cast<const float4*>(wy), wyb = *reinterpret_cast<const float4*>(wy + 4);
            const float wy4[2] = {wya.y, wya.z}, wy2[4] = {wya.w, wyb.x, wyb.y, wyb.z};
#pragma unroll
            for (int q = 0; q < 2; ++q) a8[q] = r8[q] * wya.x + a8[q];
#pragma unroll
            for (int ay = 0; ay < 2; ++ay)
#pragma unroll
                for (int ax = 0; ax < 2; ++ax)
#pragma unroll
                    for (int q = 0; q < 2; ++q) a4[ay][ax][q] = r4[ax][q] * wy4[ay] + a4[ay][ax][q];
#pragma unroll
            for (int ay = 0; ay < 4; ++ay)
#pragma unroll
                for (int ax = 0; ax < 4; ++ax)
#pragma unroll
                    for (int q = 0; q < 2; ++q) a2[ay][ax][q] = r2[ax][q] * wy2[ay] + a2[ay][ax][q];
        }
#pragma unroll
        for (int ay = 0; ay < 2; ++ay)
#pragma unroll
            for (int ax = 0; ax < 2; ++ax) store4v<T>(d4 + ((b * h4 + 2 * y8 + ay) * w4 + 2 * x8 + ax) * C + c0, a4[ay][ax]);
#pragma unroll
        for (int ay = 0; ay < 4; ++ay)
#pragma unroll
            for (int ax = 0; ax < 4; ++ax) store4v<T>(d2 + ((b * h2 + 4 * y8 + ay) * w2 + 4 * x8 + ax) * C + c0, a2[ay][ax]);
        store4v<T>(d8 + ((b * h8 + y8) * w8 + x8) * C + c0, a8);
    }
}

extern "C" int segf_bilinear_bwd_248(int dt, int B, int H, int W, int C, const void* dout, int64_t ldo, void* d2, void* d4, void* d8,
                                     void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    if (H % 8 || W % 8 || C % 8 || ldo < C || (ldo % 8)) return SEGF_ERR_SHAPE;
    if (((uintptr_t)dout | (uintptr_t)d2 | (uintptr_t)d4 | (uintptr_t)d8) % 16) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    // bf16, C % 128 == 0: on the matrix pipe (fuse_map.hip), every block including the border ring
    if (fuse_map_bwd_supported(dt, B, H, W, C)) return fuse_map_bwd_launch(B, H, W, C, dout, ldo, d2, d4, d8, st);
    const int ring = 0;
    const int64_t total = ring ? (int64_t)B * (2 * (W / 8) + 2 * (H / 8 - 2)) * (C / 4) : (int64_t)B * (H / 8) * (W / 8) * (C / 4);
    const int blocks = (int)imin64(cdiv64(total, 256), 32768);
    SEGF_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL((bilinear_bwd_248_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)dout, ldo, (T*)d2, (T*)d4, (T*)d8, B, H, W, C,
                           0, ring);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}
