// Spatial ops on NHWC activations: depthwise 3x3 (+bias +GELU) forward/backward, im2col / col2im for the
// strided PatchEmbed and spatial-reduction convolutions (which then run as MFMA GEMMs).
//   DWConv + GELU : reference models/backbones/mit.py:62-71 (DWConv), :98-99 (F.gelu(self.dwconv(self.fc1(x))))
//   PatchEmbed    : mit.py:105,127 (Conv2d k7 s4 p3 / k3 s2 p1);  sr conv: mit.py:21,48 (Conv2d k=s=sr)
// All HBM-bound: channel-contiguous 16-B lane accesses, fp32 math.
#include <stdlib.h>
#include "colreduce.h"

#define DW_PIX 4   // output pixels per thread along W (sliding 3x(PIX+2) window in registers)

// Depthwise 3x3 on NHWC, column-fixed threads (common.h): a thread keeps one 8-channel chunk, its 9x8 weights and bias in
// registers, and walks over (image, row, 4-pixel strip) units.
//   MODE 0: y = act( sum_k w[c][k] x[p+off_k][c] + bias[c] )        (forward; act = GELU(erf) when apply_gelu)
//   MODE 1: y = correlation with the flipped kernel, no bias / act   (data gradient: dx = conv^T(du))
//   MODE 2: y = dy * gelu'( conv(x) + bias )                         (backward pass A: du; equals dy when !apply_gelu)
template <typename T, int MODE>
__global__ void __launch_bounds__(256) dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int apply_gelu,
                                                         const T* __restrict__ dy, T* __restrict__ y, int B, int H, int W, int C) {
    const int nchunk = C / 8;
    const int wg = (W + DW_PIX - 1) / DW_PIX;
    const int units = B * H * wg;
    const int64_t g = (int64_t)xcd_block() * 256 + threadIdx.x;
    const int ustep = (int)(((int64_t)gridDim.x * 256) / nchunk);
    const int c0 = (int)(g % nchunk) * 8;
    // channel PAIRS as float2 vectors: the 36 tap products per pixel and the GELU polynomial issue as packed fp32 instructions
    // (these kernels are bound by VALU issue -- the tap fmas plus ~25 operations of GELU per element -- not by HBM)
    f32x2_t wk[9][4], bs[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) {
            const int ks = MODE == 1 ? 8 - kk : kk;
            wk[kk][jj] = f32x2_t{w[(c0 + 2 * jj) * 9 + ks], w[(c0 + 2 * jj + 1) * 9 + ks]};
        }
        bs[jj] = (MODE != 1 && bias) ? f32x2_t{bias[c0 + 2 * jj], bias[c0 + 2 * jj + 1]} : f32x2_t{0.f, 0.f};
    }
    for (int u = (int)(g / nchunk); u < units; u += ustep) {
        const int xg = u % wg;
        const int t = u / wg;
        const int yy = t % H;
        const int b = t / H;
        const int x0 = xg * DW_PIX;
        f32x2_t acc[DW_PIX][4];
#pragma unroll
        for (int p = 0; p < DW_PIX; ++p)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[p][jj] = bs[jj];
        // every tap load is unconditional (clamped address, value masked afterwards) and all 18 (+ the 4 gradient pixels of
        // MODE 2) are issued before the first use: a conditional load costs a branch plus a full s_waitcnt per load
        Raw8<T> raw[3][DW_PIX + 2], graw[DW_PIX];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = yy + ky - 1;
            const T* row = x + (((int64_t)b * H + ((iy >= 0 && iy < H) ? iy : yy)) * W) * C + c0;
#pragma unroll
            for (int cx = 0; cx < DW_PIX + 2; ++cx) {
                const int ix = x0 + cx - 1;
                raw[ky][cx] = load8_raw<T>(row + (int64_t)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C);
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int p = 0; p < DW_PIX; ++p)
                graw[p] = load8_raw<T>(dy + (((int64_t)b * H + yy) * W + (x0 + p < W ? x0 + p : W - 1)) * C + c0);
        }
        SEGF_LOADS_ISSUED();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = yy + ky - 1;
            const bool vy = iy >= 0 && iy < H;
#pragma unroll
            for (int cx = 0; cx < DW_PIX + 2; ++cx) {
                const int ix = x0 + cx - 1;
                const bool ok = vy && ix >= 0 && ix < W;
                f32x2_t v[4];
                zero_unless(raw[ky][cx], ok);
                unpack8v<T>(raw[ky][cx], v);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int p = cx - kx;            // output pixel that sees this column through tap kx
                    if (p >= 0 && p < DW_PIX) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) acc[p][jj] = wk[ky * 3 + kx][jj] * v[jj] + acc[p][jj];
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < DW_PIX; ++p) {
            if (x0 + p < W) {
                const int64_t off = (((int64_t)b * H + yy) * W + x0 + p) * C + c0;
                if (MODE == 0 && apply_gelu) gelu_erf8<false>(acc[p], nullptr);
                if (MODE == 2) {
                    f32x2_t gy[4];
                    unpack8v<T>(graw[p], gy);
                    if (apply_gelu) gelu_erf8<true>(acc[p], gy);
                    else {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) acc[p][jj] = gy[jj];
                    }
                }
                store8v<T>(y + off, acc[p]);
            }
        }
    }
}

// ---- vertical-walk form -------------------------------------------------------------------------------------------
// The strip kernel above loads the 3 x 6 input window of every 4-pixel strip anew: 18 loads per strip, every input element 4.5
// times through the L1 (HBM sees it about once).  Here a thread owns a 4-channel chunk and a 4-pixel-wide column and WALKS DOWN
// the image: every incoming input row (6 loads, 1.5 x) is unpacked once and scattered into the accumulators of the three
// output rows it touches (above: ky = 2, same: ky = 1, below: ky = 0); the row above is then complete and is finished and
// stored.  Same multiply-adds, a third of the load instructions, no window in registers.  The three accumulator rows rotate
// by unrolling the walk three times (no register moves).  Rows per segment R: each segment re-reads two rows.
template <typename T> struct Raw4;
template <> struct Raw4<float> { float4 a; };
template <> struct Raw4<bf16_t> { uint2 u; };
template <typename T> __device__ __forceinline__ Raw4<T> load4_raw(const T* p);
template <> __device__ __forceinline__ Raw4<float> load4_raw<float>(const float* p) { Raw4<float> r; r.a = *reinterpret_cast<const float4*>(p); return r; }
template <> __device__ __forceinline__ Raw4<bf16_t> load4_raw<bf16_t>(const bf16_t* p) { Raw4<bf16_t> r; r.u = *reinterpret_cast<const uint2*>(p); return r; }
__device__ __forceinline__ void unpack4v(const Raw4<float>& r, bool ok, f32x2_t (&v)[2]) {
    v[0] = ok ? f32x2_t{r.a.x, r.a.y} : f32x2_t{0.f, 0.f}; v[1] = ok ? f32x2_t{r.a.z, r.a.w} : f32x2_t{0.f, 0.f};
}
__device__ __forceinline__ void unpack4v(const Raw4<bf16_t>& r, bool ok, f32x2_t (&v)[2]) {
    const uint32_t a = ok ? r.u.x : 0u, b = ok ? r.u.y : 0u;
    v[0] = f32x2_t{__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u)};
    v[1] = f32x2_t{__uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u)};
}
__device__ __forceinline__ void store4v(float* p, const f32x2_t (&v)[2]) { *reinterpret_cast<float4*>(p) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y); }
__device__ __forceinline__ void store4v(bf16_t* p, const f32x2_t (&v)[2]) {
    uint2 u; u.x = pack2bf(v[0].x, v[0].y); u.y = pack2bf(v[1].x, v[1].y);
    *reinterpret_cast<uint2*>(p) = u;
}

template <typename T, int MODE>
__global__ void __launch_bounds__(256) dwconv3x3_walk_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int apply_gelu,
                                                              const T* __restrict__ dy, T* __restrict__ y, int B, int H, int W, int C,
                                                              int R, int nseg) {
    const int nchunk = C / 4;
    const int wg = (W + DW_PIX - 1) / DW_PIX;
    const int units = B * nseg * wg;
    const int64_t g = (int64_t)xcd_block() * 256 + threadIdx.x;
    const int ustep = (int)(((int64_t)gridDim.x * 256) / nchunk);
    const int c0 = (int)(g % nchunk) * 4;
    f32x2_t wk[9][2], bs[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int kk = 0; kk < 9; ++kk) {
            const int ks = MODE == 1 ? 8 - kk : kk;
            wk[kk][jj] = f32x2_t{w[(c0 + 2 * jj) * 9 + ks], w[(c0 + 2 * jj + 1) * 9 + ks]};
        }
        bs[jj] = (MODE != 1 && bias) ? f32x2_t{bias[c0 + 2 * jj], bias[c0 + 2 * jj + 1]} : f32x2_t{0.f, 0.f};
    }
    const int steps = (R + 2 + 2) / 3;                    // walk length R + 2 input rows, rounded up to whole rotations
    for (int u = (int)(g / nchunk); u < units; u += ustep) {
        const int xg = u % wg;
        const int t = u / wg;
        const int seg = t % nseg;
        const int b = t / nseg;
        const int x0 = xg * DW_PIX;
        const int ya = seg * R, yb = ya + R < H ? ya + R : H;
        const T* xb = x + (int64_t)b * H * W * C + c0;
        const T* gb = MODE == 2 ? dy + (int64_t)b * H * W * C + c0 : nullptr;
        T* yo_b = y + (int64_t)b * H * W * C + c0;
        int coff[DW_PIX + 2];
        bool cok[DW_PIX + 2];
#pragma unroll
        for (int cx = 0; cx < DW_PIX + 2; ++cx) {
            const int ix = x0 + cx - 1;
            cok[cx] = ix >= 0 && ix < W;
            coff[cx] = (ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C;
        }
        f32x2_t acc[3][DW_PIX][2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int p = 0; p < DW_PIX; ++p) { acc[a][p][0] = bs[0]; acc[a][p][1] = bs[1]; }
        // three raw-row buffers rotating with the accumulator roles: the loads run TWO rows ahead of the arithmetic (with one row
        // ahead a wave had 3 KB in flight, 24 KB per CU at two waves per SIMD: latency-bound at 3.6 TB/s)
        Raw4<T> rawb[3][DW_PIX + 2], grawb[3][DW_PIX];
        auto load_row = [&](int yin, Raw4<T> (&dst)[DW_PIX + 2]) {
            const int yc = yin < 0 ? 0 : (yin >= H ? H - 1 : yin);
            const T* row = xb + (int64_t)yc * W * C;
#pragma unroll
            for (int cx = 0; cx < DW_PIX + 2; ++cx) dst[cx] = load4_raw<T>(row + coff[cx]);
        };
        auto load_grow = [&](int yo, Raw4<T> (&dst)[DW_PIX]) {
            const int yc = yo < 0 ? 0 : (yo >= H ? H - 1 : yo);
            const T* row = gb + (int64_t)yc * W * C;
#pragma unroll
            for (int p = 0; p < DW_PIX; ++p) dst[p] = load4_raw<T>(row + coff[p + 1]);
        };
        load_row(ya - 1, rawb[0]);
        load_row(ya, rawb[1]);
        if (MODE == 2) { load_grow(ya - 2, grawb[0]); load_grow(ya - 1, grawb[1]); }
        int yin = ya - 1;
        for (int st = 0; st < steps; ++st) {
#pragma unroll
            for (int j = 0; j < 3; ++j, ++yin) {
                // roles of the accumulator rows in this sub-step: P = output row yin - 1 (completes), Cc = row yin, N = row yin + 1
                f32x2_t (&P)[DW_PIX][2] = acc[j % 3];
                f32x2_t (&Cc)[DW_PIX][2] = acc[(j + 1) % 3];
                f32x2_t (&N)[DW_PIX][2] = acc[(j + 2) % 3];
                Raw4<T> (&curc)[DW_PIX + 2] = rawb[j % 3];                // input row yin (loaded two sub-steps ago)
                Raw4<T> (&gcurc)[DW_PIX] = grawb[j % 3];                  // MODE 2: dy row yin - 1
                load_row(yin + 2, rawb[(j + 2) % 3]);
                if (MODE == 2) load_grow(yin + 1, grawb[(j + 2) % 3]);
                const bool vy = yin >= 0 && yin < H && yin <= yb;
#pragma unroll
                for (int cx = 0; cx < DW_PIX + 2; ++cx) {
                    f32x2_t v[2];
                    unpack4v(curc[cx], vy && cok[cx], v);
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int p = cx - kx;
                        if (p >= 0 && p < DW_PIX) {
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                P[p][jj] = wk[6 + kx][jj] * v[jj] + P[p][jj];
                                Cc[p][jj] = wk[3 + kx][jj] * v[jj] + Cc[p][jj];
                                N[p][jj] = wk[kx][jj] * v[jj] + N[p][jj];
                            }
                        }
                    }
                }
                const int yo = yin - 1;
                if (yo >= ya && yo < yb) {
                    T* orow = yo_b + (int64_t)yo * W * C;
                    // two pixels x four channels at a time through the eight-wide GELU helpers
#pragma unroll
                    for (int h = 0; h < DW_PIX / 2; ++h) {
                        f32x2_t a8[4] = {P[2 * h][0], P[2 * h][1], P[2 * h + 1][0], P[2 * h + 1][1]};
                        if (MODE == 0 && apply_gelu) gelu_erf8<false>(a8, nullptr);
                        if (MODE == 2) {
                            f32x2_t gy[4], g0[2], g1[2];
                            unpack4v(gcurc[2 * h], true, g0); unpack4v(gcurc[2 * h + 1], true, g1);
                            gy[0] = g0[0]; gy[1] = g0[1]; gy[2] = g1[0]; gy[3] = g1[1];
                            if (apply_gelu) gelu_erf8<true>(a8, gy);
                            else { a8[0] = gy[0]; a8[1] = gy[1]; a8[2] = gy[2]; a8[3] = gy[3]; }
                        }
                        const f32x2_t o0[2] = {a8[0], a8[1]}, o1[2] = {a8[2], a8[3]};
                        if (cok[2 * h + 1]) store4v(orow + coff[2 * h + 1], o0);
                        if (cok[2 * h + 2]) store4v(orow + coff[2 * h + 2], o1);
                    }
                }
#pragma unroll
                for (int p = 0; p < DW_PIX; ++p) { P[p][0] = bs[0]; P[p][1] = bs[1]; }       // becomes row yin + 2 of the next sub-step
            }
        }
    }
}

static inline void dw_walk_plan(int H, int& R, int& nseg, int64_t columns = 0) {
    // segments of at most 64 rows (two re-read rows per segment), equal length; shorter ones when the launch would otherwise
    // have too few threads to fill the chip (columns = batch x 4-pixel strips x channel chunks; small batches)
    nseg = (H + 63) / 64;
    while (columns > 0 && columns * nseg < 131072 && (H + nseg - 1) / nseg > 8) nseg *= 2;
    if (POL(dw_walk_rows) > 0) nseg = (H + POL(dw_walk_rows) - 1) / POL(dw_walk_rows);
    R = (H + nseg - 1) / nseg;
}
static inline bool dw_use_walk() { return !POL(dw_no_walk); }
template <typename T, int MODE>
static inline void dw_walk_launch(hipStream_t st, const T* x, const float* w, const float* bias, int apply_gelu, const T* dy, T* y,
                                  int B, int H, int W, int C) {
    int R, nseg;
    dw_walk_plan(H, R, nseg, (int64_t)B * ((W + DW_PIX - 1) / DW_PIX) * (C / 4));
    const int64_t units = (int64_t)B * nseg * ((W + DW_PIX - 1) / DW_PIX);
    const int blocks = colfixed_blocks(units, C / 4, 1, 32768);
    hipLaunchKernelGGL((dwconv3x3_walk_kernel<T, MODE>), dim3(blocks), dim3(256), 0, st, x, w, bias, apply_gelu, dy, y, B, H, W, C, R, nseg);
}

static inline int dw_blocks(int B, int H, int W, int C) {
    // 8 strips per thread: the 72 per-channel weights a thread keeps in registers are loaded once per 8 strips
    return colfixed_blocks((int64_t)B * H * ((W + DW_PIX - 1) / DW_PIX), C / 8, 8, 16384);
}

extern "C" int segf_dwconv3x3_gelu_fwd(int dt, int B, int H, int W, int C, const void* x, const float* w, const float* bias,
                                       int apply_gelu, void* y, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (C <= 0 || C % 8 != 0 || ((uintptr_t)x % 16) || ((uintptr_t)y % 16)) return SEGF_ERR_SHAPE;
    if ((int64_t)B * H * W >= (1ll << 31)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = dw_blocks(B, H, W, C);
    SEGF_DISPATCH_DT(dt, T, {
        if (dw_use_walk()) dw_walk_launch<T, 0>(st, (const T*)x, w, bias, apply_gelu, (const T*)nullptr, (T*)y, B, H, W, C);
        else
        hipLaunchKernelGGL((dwconv3x3_kernel<T, 0>), dim3(blocks), dim3(256), 0, st, (const T*)x, w, bias, apply_gelu,
                           (const T*)nullptr, (T*)y, B, H, W, C);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// backward pass B (weight / bias gradient): dw[c][k] = sum_p du[p][c] * x[p+off_k][c], db[c] = sum_p du[p][c].
// Block layout of the column reductions (colreduce.h): tx -> channel chunk, ty -> unit lane; a thread walks 4-pixel strips
// with the x window in registers (18 x-loads + 4 du-loads per strip) and keeps its 10x8 sums in registers.
#define DWG_MAX_BLOCKS 1024
struct DwgPlan { int ch, rl, slabs, nblk, units_per_blk; };
static inline DwgPlan dwg_plan(int B, int H, int W, int C) {
    DwgPlan p;
    const int nchunk = C / 8;
    p.ch = nchunk < 256 ? nchunk : 256;
    p.rl = 256 / p.ch;
    p.slabs = (nchunk + p.ch - 1) / p.ch;
    const int64_t units = (int64_t)B * H * ((W + DW_PIX - 1) / DW_PIX);
    int64_t want = cdiv64(units, (int64_t)p.rl * 4);
    int cap = DWG_MAX_BLOCKS / p.slabs;
    if (cap < 1) cap = 1;
    p.nblk = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    p.units_per_blk = (int)cdiv64(units, p.nblk);
    return p;
}

template <typename T>
__global__ void __launch_bounds__(256) dwconv3x3_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ du,
                                                               float* __restrict__ partial, int B, int H, int W, int C, int ch,
                                                               int rl, int units_per_blk) {
    __shared__ float red[256 * 8];
    const int tx = threadIdx.x % ch, ty = threadIdx.x / ch;
    const int chunk = blockIdx.y * ch + tx;
    const int c0 = chunk * 8;
    const bool active = ty < rl && c0 < C;
    const int wg = (W + DW_PIX - 1) / DW_PIX;
    const int units = B * H * wg;
    float acc[10][8];
#pragma unroll
    for (int o = 0; o < 10; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
    // units are dealt to the workgroups round-robin in groups of rl strips (not as one contiguous range each): the workgroups
    // that run at the same time on an XCD then walk ADJACENT image rows, and the two re-reads of every input row (as the row
    // above / below of its neighbours) hit that XCD's L2.  With contiguous ranges every workgroup streamed its own 16 rows,
    // 128 of them per 4 MB L2, and FETCH_SIZE showed x fetched three times (2.13 GB for 1.07 GB at [128,128,128,128]).
    (void)units_per_blk;
    if (active) {
        for (int u = (int)xcd_block() * rl + ty; u < units; u += (int)gridDim.x * rl) {
            const int xg = u % wg;
            const int t = u / wg;
            const int yy = t % H;
            const int b = t / H;
            const int x0 = xg * DW_PIX;
            // all loads of the strip (its gradient pixels and the 3 x (DW_PIX + 2) input window) are issued before the
            // first use (see SEGF_LOADS_ISSUED in common.h)
            Raw8<T> graw[DW_PIX], vraw[3][DW_PIX + 2];
#pragma unroll
            for (int p = 0; p < DW_PIX; ++p) {
                const int xp = x0 + p < W ? x0 + p : W - 1;
                graw[p] = load8_raw<T>(du + (((int64_t)b * H + yy) * W + xp) * C + c0);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = yy + ky - 1;
                const T* row = x + (((int64_t)b * H + ((iy >= 0 && iy < H) ? iy : yy)) * W) * C + c0;
#pragma unroll
                for (int cx = 0; cx < DW_PIX + 2; ++cx) {
                    const int ix = x0 + cx - 1;
                    vraw[ky][cx] = load8_raw<T>(row + (int64_t)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C);
                }
            }
            SEGF_LOADS_ISSUED();
            float g[DW_PIX][8];
#pragma unroll
            for (int p = 0; p < DW_PIX; ++p) {
                unpack8(graw[p], g[p]);
                const bool ok = x0 + p < W;
#pragma unroll
                for (int j = 0; j < 8; ++j) { g[p][j] = ok ? g[p][j] : 0.f; acc[9][j] += g[p][j]; }
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = yy + ky - 1;
                const bool vy = iy >= 0 && iy < H;
                float v[DW_PIX + 2][8];
#pragma unroll
                for (int cx = 0; cx < DW_PIX + 2; ++cx) {
                    const int ix = x0 + cx - 1;
                    const bool ok = vy && ix >= 0 && ix < W;
                    unpack8(vraw[ky][cx], v[cx]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[cx][j] = ok ? v[cx][j] : 0.f;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int p = cx - kx;
                        if (p >= 0 && p < DW_PIX) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) acc[ky * 3 + kx][j] = fmaf(g[p][j], v[cx][j], acc[ky * 3 + kx][j]);
                        }
                    }
                }
            }
        }
    }
    colreduce_block_tail<10>(acc, active, ty == 0 && c0 < C, tx, ch, rl, c0, 8, C, partial, red);
}

// Weight / bias gradient in the vertical-walk form: thread = (4-channel chunk, 4-pixel-wide column, row segment).  Every input
// row is loaded once (6 chunks) and meets the three gradient rows it pairs with (du rows yin + 1, yin, yin - 1 for ky = 0, 1, 2),
// which rotate through registers unpacked; 10 loads per row of strips instead of 22, and x is fetched once (the strip form's
// FETCH_SIZE was 2 x the algorithmic bytes even after the round-robin dealing).
struct DwwPlan { int ch, rl, slabs, nblk, R, nseg; };
static inline DwwPlan dww_plan(int B, int H, int W, int C) {
    DwwPlan p;
    const int nchunk = C / 4;
    // channel chunks per workgroup: the split into slabs that leaves the fewest of the 256 threads idle (160 chunks: one
    // slab would use 160 threads, two slabs of 80 use 240)
    double best = -1.0;
    p.ch = 1; p.slabs = nchunk;
    for (int sl = 1; sl <= 8; ++sl) {
        const int ch = (nchunk + sl - 1) / sl;
        if (ch > 256) continue;
        const double util = (double)(ch * (256 / ch)) / 256.0 * (double)nchunk / (double)(ch * sl);
        if (util > best + 1e-9) { best = util; p.ch = ch; p.slabs = sl; }
    }
    p.rl = 256 / p.ch;
    dw_walk_plan(H, p.R, p.nseg, (int64_t)B * ((W + DW_PIX - 1) / DW_PIX) * (C / 4));
    const int64_t units = (int64_t)B * p.nseg * ((W + DW_PIX - 1) / DW_PIX);
    int64_t want = cdiv64(units, p.rl);
    int cap = 2 * DWG_MAX_BLOCKS / p.slabs;
    if (cap < 1) cap = 1;
    p.nblk = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    return p;
}

template <typename T>
__global__ void __launch_bounds__(256) dwconv3x3_wgrad_walk_kernel(const T* __restrict__ x, const T* __restrict__ du,
                                                                    float* __restrict__ partial, int B, int H, int W, int C, int ch,
                                                                    int rl, int R, int nseg) {
    __shared__ float red[256 * 4];
    const int tx = threadIdx.x % ch, ty = threadIdx.x / ch;
    const int chunk = blockIdx.y * ch + tx;
    const int c0 = chunk * 4;
    const bool active = ty < rl && c0 < C;
    const int wg = (W + DW_PIX - 1) / DW_PIX;
    const int units = B * nseg * wg;
    f32x2_t acc[10][2];
#pragma unroll
    for (int o = 0; o < 10; ++o) { acc[o][0] = f32x2_t{0.f, 0.f}; acc[o][1] = f32x2_t{0.f, 0.f}; }
    const int steps = (R + 2 + 2) / 3;
    if (active) {
        for (int u = (int)xcd_block() * rl + ty; u < units; u += (int)gridDim.x * rl) {
            const int xg = u % wg;
            const int t = u / wg;
            const int seg = t % nseg;
            const int b = t / nseg;
            const int x0 = xg * DW_PIX;
            const int ya = seg * R, yb = ya + R < H ? ya + R : H;
            const T* xb = x + (int64_t)b * H * W * C + c0;
            const T* gb = du + (int64_t)b * H * W * C + c0;
            int coff[DW_PIX + 2];
            bool cok[DW_PIX + 2];
#pragma unroll
            for (int cx = 0; cx < DW_PIX + 2; ++cx) {
                const int ix = x0 + cx - 1;
                cok[cx] = ix >= 0 && ix < W;
                coff[cx] = (ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C;
            }
            f32x2_t gw[3][DW_PIX][2];                    // gradient rows yin - 1, yin, yin + 1 (rotating roles)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int p = 0; p < DW_PIX; ++p) { gw[a][p][0] = f32x2_t{0.f, 0.f}; gw[a][p][1] = f32x2_t{0.f, 0.f}; }
            Raw4<T> rawb[3][DW_PIX + 2], grawb[3][DW_PIX];       // rotating: loads run two rows ahead (see dwconv3x3_walk_kernel)
            auto load_row = [&](int yin, Raw4<T> (&dst)[DW_PIX + 2]) {
                const int yc = yin < 0 ? 0 : (yin >= H ? H - 1 : yin);
                const T* row = xb + (int64_t)yc * W * C;
#pragma unroll
                for (int cx = 0; cx < DW_PIX + 2; ++cx) dst[cx] = load4_raw<T>(row + coff[cx]);
            };
            auto load_grow = [&](int yo, Raw4<T> (&dst)[DW_PIX]) {
                const int yc = yo < 0 ? 0 : (yo >= H ? H - 1 : yo);
                const T* row = gb + (int64_t)yc * W * C;
#pragma unroll
                for (int p = 0; p < DW_PIX; ++p) dst[p] = load4_raw<T>(row + coff[p + 1]);
            };
            load_row(ya - 1, rawb[0]); load_row(ya, rawb[1]);
            load_grow(ya, grawb[0]); load_grow(ya + 1, grawb[1]);       // gradient rows run one row ahead of the input rows
            int yin = ya - 1;
            for (int st = 0; st < steps; ++st) {
#pragma unroll
                for (int j = 0; j < 3; ++j, ++yin) {
                    f32x2_t (&GP)[DW_PIX][2] = gw[j % 3];            // row yin - 1  (pairs with ky = 2)
                    f32x2_t (&GC)[DW_PIX][2] = gw[(j + 1) % 3];      // row yin      (ky = 1)
                    f32x2_t (&GN)[DW_PIX][2] = gw[(j + 2) % 3];      // row yin + 1  (ky = 0): arrives now
                    Raw4<T> (&cur)[DW_PIX + 2] = rawb[j % 3];
                    Raw4<T> (&gcur)[DW_PIX] = grawb[j % 3];
                    load_row(yin + 2, rawb[(j + 2) % 3]);
                    load_grow(yin + 3, grawb[(j + 2) % 3]);
                    const bool gok = yin + 1 >= ya && yin + 1 < yb;
#pragma unroll
                    for (int p = 0; p < DW_PIX; ++p) {
                        unpack4v(gcur[p], gok && cok[p + 1], GN[p]);
                        acc[9][0] += GN[p][0]; acc[9][1] += GN[p][1];
                    }
                    const bool vy = yin >= 0 && yin < H;
#pragma unroll
                    for (int cx = 0; cx < DW_PIX + 2; ++cx) {
                        f32x2_t v[2];
                        unpack4v(cur[cx], vy && cok[cx], v);
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int p = cx - kx;
                            if (p >= 0 && p < DW_PIX) {
#pragma unroll
                                for (int jj = 0; jj < 2; ++jj) {
                                    acc[6 + kx][jj] = GP[p][jj] * v[jj] + acc[6 + kx][jj];
                                    acc[3 + kx][jj] = GC[p][jj] * v[jj] + acc[3 + kx][jj];
                                    acc[kx][jj] = GN[p][jj] * v[jj] + acc[kx][jj];
                                }
                            }
                        }
                    }
                }
            }
        }
    }
    // block tail: sum over the rl unit lanes in fixed order, partial[blockIdx.x][o][c0 .. c0 + 4)
    for (int o = 0; o < 10; ++o) {
        __syncthreads();
        red[threadIdx.x * 4 + 0] = active ? acc[o][0].x : 0.f; red[threadIdx.x * 4 + 1] = active ? acc[o][0].y : 0.f;
        red[threadIdx.x * 4 + 2] = active ? acc[o][1].x : 0.f; red[threadIdx.x * 4 + 3] = active ? acc[o][1].y : 0.f;
        __syncthreads();
        if (ty == 0 && c0 < C) {
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int yy = 0; yy < rl; ++yy)
#pragma unroll
                for (int j = 0; j < 4; ++j) s4[j] += red[(yy * ch + tx) * 4 + j];
            float* dst = partial + ((int64_t)blockIdx.x * 10 + o) * C + c0;
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = s4[j];
        }
    }
}

// [10][C] sums -> dw[C][9], db[C]
__global__ void dw_scatter_kernel(const float* __restrict__ sums, int C, float* __restrict__ dw, float* __restrict__ db) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int kk = 0; kk < 9; ++kk) dw[c * 9 + kk] = sums[kk * C + c];
    db[c] = sums[9 * C + c];
}

// ---- small maps: the whole backward of the depthwise 3x3 (+ GELU) in ONE launch ---------------------------------------------------
// MiT stages 3 / 4 at 512^2 (32 x 32 and 16 x 16 maps, mit.py:62-71): the three passes above (du = dy gelu'(conv(x) + b); dw / db partial
// sums; dx = conv^T(du)) are three launches over maps that fit the LDS of one workgroup many times over -- at the reference's default
// batch each is a dependent link of 9 - 16 us in the step's launch chain.  Here a workgroup owns the map of one image for 8 nch channels:
// x is staged once, z and du are formed per (pixel, 8-channel chunk) in the tap order of the walk kernels (same fused multiply-adds, the
// same eight-wide GELU helper: du and dx are BITWISE what the three-pass form gives), du stays in LDS (rounded to T, as the three-pass form
// stores it), dx is formed from it, and the weight / bias sums are taken over the staged tiles in a fixed order (another summation order
// than the three-pass form: fp32 round-off).  Partials: part[b][10][C], finalized like the other forms (nblk = B).
template <typename T>
__global__ void __launch_bounds__(256) dwconv3x3_bwd_small_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, int apply_gelu,
                                                                   const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ part,
                                                                   int B, int H, int W, int C, int nch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dws_smem[];
    const int HW = H * W, U = HW * nch, groups = C / (8 * nch);
    const int b = blockIdx.x / groups, cbase = (blockIdx.x - b * groups) * 8 * nch;
    T* xs = reinterpret_cast<T*>(dws_smem);                        // [nch][HW][8]
    T* dus = xs + (size_t)U * 8;                                   // [nch][HW][8]
    float* wsh = reinterpret_cast<float*>(dus + (size_t)U * 8);   // [nch][10][8]: nine taps + bias, the 8 channels of a chunk contiguous
    float* red = wsh + nch * 80;                                   // [slices][10 nch][8]
    const T* xb = x + (int64_t)b * HW * C + cbase;
    const T* gb = dy + (int64_t)b * HW * C + cbase;
    T* ob = dx + (int64_t)b * HW * C + cbase;
    for (int i = threadIdx.x; i < nch * 80; i += 256) {
        const int ch = i / 80, r = i - ch * 80, k = r >> 3, c = r & 7;
        wsh[i] = k < 9 ? w[(cbase + 8 * ch + c) * 9 + k] : (bias ? bias[cbase + 8 * ch + c] : 0.f);
    }
    for (int u = threadIdx.x; u < U; u += 256) {
        const int ch = u / HW, p = u - ch * HW;
        *reinterpret_cast<Raw8<T>*>(xs + (size_t)u * 8) = load8_raw<T>(xb + (int64_t)p * C + 8 * ch);
    }
    __syncthreads();
    // du = dy gelu'(conv(x) + b), rounded to T (what the three-pass form stores): chunk by chunk, the chunk's taps in registers
    for (int ch = 0; ch < nch; ++ch) {
        f32x2_t wk[10][4];
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const float4 lo = *reinterpret_cast<const float4*>(wsh + ch * 80 + k * 8), hi = *reinterpret_cast<const float4*>(wsh + ch * 80 + k * 8 + 4);
            wk[k][0] = f32x2_t{lo.x, lo.y}; wk[k][1] = f32x2_t{lo.z, lo.w}; wk[k][2] = f32x2_t{hi.x, hi.y}; wk[k][3] = f32x2_t{hi.z, hi.w};
        }
        const T* xc = xs + (size_t)ch * HW * 8;
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int py = p / W, px = p - py * W;
            const Raw8<T> graw = load8_raw<T>(gb + (int64_t)p * C + 8 * ch);
            f32x2_t z[4] = {wk[9][0], wk[9][1], wk[9][2], wk[9][3]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = py + ky - 1, xx = px + kx - 1;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        f32x2_t v[4];
                        unpack8v<T>(*reinterpret_cast<const Raw8<T>*>(xc + (size_t)(yy * W + xx) * 8), v);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) z[jj] = wk[ky * 3 + kx][jj] * v[jj] + z[jj];
                    }
                }
            f32x2_t gy[4];
            unpack8v<T>(graw, gy);
            if (apply_gelu) gelu_erf8<true>(z, gy);
            else { z[0] = gy[0]; z[1] = gy[1]; z[2] = gy[2]; z[3] = gy[3]; }
            store8v<T>(dus + ((size_t)ch * HW + p) * 8, z);
        }
    }
    __syncthreads();
    // dx = conv^T(du): output (py, px) gathers du at (py + ky - 1, px + kx - 1) with the flipped tap, in the walk kernel's order
    for (int ch = 0; ch < nch; ++ch) {
        f32x2_t wk[9][4];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float4 lo = *reinterpret_cast<const float4*>(wsh + ch * 80 + k * 8), hi = *reinterpret_cast<const float4*>(wsh + ch * 80 + k * 8 + 4);
            wk[k][0] = f32x2_t{lo.x, lo.y}; wk[k][1] = f32x2_t{lo.z, lo.w}; wk[k][2] = f32x2_t{hi.x, hi.y}; wk[k][3] = f32x2_t{hi.z, hi.w};
        }
        const T* dc = dus + (size_t)ch * HW * 8;
        for (int p = threadIdx.x; p < HW; p += 256) {
            const int py = p / W, px = p - py * W;
            f32x2_t a[4] = {f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = py + ky - 1, xx = px + kx - 1;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        f32x2_t v[4];
                        unpack8v<T>(*reinterpret_cast<const Raw8<T>*>(dc + (size_t)(yy * W + xx) * 8), v);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) a[jj] = wk[8 - (ky * 3 + kx)][jj] * v[jj] + a[jj];
                    }
                }
            store8v<T>(ob + (int64_t)p * C + 8 * ch, a);
        }
    }
    // dw[c][k] = sum_p du[p][c] x[p + off(k)][c], db[c] = sum_p du[p][c]: a thread takes (chunk, tap) for one contiguous slice of the pixels,
    // all 8 channels at once; the slices meet in LDS and are added in slice order
    const int T4 = 10 * nch, S = 256 / T4;
    {
        const int tc = threadIdx.x % T4, sl = threadIdx.x / T4;
        if (sl < S) {
            const int ch = tc / 10, k = tc - ch * 10;
            const int p0 = (int)((int64_t)HW * sl / S), p1 = (int)((int64_t)HW * (sl + 1) / S);
            const int dyk = k < 9 ? k / 3 - 1 : 0, dxk = k < 9 ? k % 3 - 1 : 0;
            const T* dc = dus + (size_t)ch * HW * 8;
            const T* xc = xs + (size_t)ch * HW * 8;
            f32x2_t acc[4] = {f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}, f32x2_t{0.f, 0.f}};
            int py = p0 / W, px = p0 - py * W;
            for (int p = p0; p < p1; ++p) {
                f32x2_t d[4];
                unpack8v<T>(*reinterpret_cast<const Raw8<T>*>(dc + (size_t)p * 8), d);
                if (k == 9) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[jj] = acc[jj] + d[jj];
                } else {
                    const int yy = py + dyk, xx = px + dxk;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        f32x2_t v[4];
                        unpack8v<T>(*reinterpret_cast<const Raw8<T>*>(xc + (size_t)(yy * W + xx) * 8), v);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) acc[jj] = d[jj] * v[jj] + acc[jj];
                    }
                }
                if (++px == W) { px = 0; ++py; }
            }
            float* r = red + ((size_t)sl * T4 + tc) * 8;
            *reinterpret_cast<float4*>(r) = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
            *reinterpret_cast<float4*>(r + 4) = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < T4 * 8; o += 256) {
        const int tc = o >> 3, c = o & 7, ch = tc / 10, k = tc - ch * 10;
        float t = red[o];
        for (int q = 1; q < S; ++q) t += red[(size_t)q * T4 * 8 + o];
        part[((int64_t)b * 10 + k) * C + cbase + 8 * ch + c] = t;
    }
}
// nch (8-channel chunks per workgroup) of the one-launch form, 0 when the map does not take it
static inline int dw_small_nch(int dt, int B, int H, int W, int C) {
    if (POL(dw_no_small) || C % 8 || B > 4096) return 0;
    const int HW = H * W, umax = dt == SEGF_BF16 ? 1024 : 512;
    if (HW > umax || HW < 16) return 0;
    for (int nch = 4; nch >= 1; nch >>= 1)
        if (HW * nch <= umax && (C / 8) % nch == 0) {
            // one round of workgroups at most: a workgroup walks its map serially (26 - 33 us whatever the batch), which beats the three
            // launches (39 - 42 us) only while the launch chain, not the arithmetic, is what the step waits for (measured: batch 4 +0.6 %,
            // batch 16 -0.7 %, batch 128 -1.8 % without this bound)
            if ((int64_t)B * (C / (8 * nch)) > 512 && !POL(dw_small_always)) return 0;
            return nch;
        }
    return 0;
}

extern "C" int64_t segf_dwconv3x3_bwd_ws(int B, int H, int W, int C) {
    DwgPlan p = dwg_plan(B, H, W, C > 0 ? C : 8);
    DwwPlan q = dww_plan(B, H > 0 ? H : 1, W > 0 ? W : 1, C > 0 ? C : 8);
    int nblk = p.nblk > q.nblk ? p.nblk : q.nblk;
    if (B > nblk) nblk = B;                       // (the one-launch form of small maps leaves one partial slab per image)
    return (int64_t)nblk * 10 * C + 10 * (int64_t)C;
}

extern "C" int segf_dwconv3x3_gelu_bwd(int dt, int B, int H, int W, int C, const void* x, const float* w, const float* bias,
                                       int apply_gelu, const void* dy, void* du, void* dx, float* dw, float* db, float* ws,
                                       void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (C <= 0 || C % 8 != 0 || ((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || ((uintptr_t)du % 16) || ((uintptr_t)dx % 16))
        return SEGF_ERR_SHAPE;
    if ((int64_t)B * H * W >= (1ll << 31)) return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (const int nch = dw_small_nch(dt, B, H, W, C)) {
        const size_t esz = dt == SEGF_BF16 ? 2 : 4;
        const size_t shm = (size_t)H * W * nch * 8 * esz * 2 + (size_t)nch * 80 * 4 + 256 * 8 * 4;      // tiles + taps + slice sums
        const int blocks = B * (C / (8 * nch));
        SEGF_DISPATCH_DT(dt, T, {
            hipLaunchKernelGGL((dwconv3x3_bwd_small_kernel<T>), dim3(blocks), dim3(256), shm, st, (const T*)x, w, bias, apply_gelu,
                               (const T*)dy, (T*)dx, ws, B, H, W, C, nch);
        })
        SEGF_CHECK_LAUNCH();
        if (!dw) return 0;          // deferred: [B][10 C] partial sums stay in ws (segf_dwconv3x3_bwd_blocks == B)
        float* sums_s = ws + (int64_t)B * 10 * C;
        colreduce_finalize_launch(ws, B, 10 * (int64_t)C, sums_s, st);
        SEGF_CHECK_LAUNCH();
        hipLaunchKernelGGL(dw_scatter_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums_s, C, dw, db);
        SEGF_CHECK_LAUNCH();
        return 0;
    }
    DwgPlan p = dwg_plan(B, H, W, C);
    const DwwPlan q = dww_plan(B, H, W, C);
    const bool walk = dw_use_walk();
    if (walk) p.nblk = q.nblk;
    float* sums = ws + (int64_t)p.nblk * 10 * C;
    const int blocks = dw_blocks(B, H, W, C);
    SEGF_DISPATCH_DT(dt, T, {
        // A: du = dy * gelu'(conv(x) + b);  B: dw / db partial sums;  C: dx = conv^T(du)
        if (dw_use_walk()) dw_walk_launch<T, 2>(st, (const T*)x, w, bias, apply_gelu, (const T*)dy, (T*)du, B, H, W, C);
        else
        hipLaunchKernelGGL((dwconv3x3_kernel<T, 2>), dim3(blocks), dim3(256), 0, st, (const T*)x, w, bias, apply_gelu,
                           (const T*)dy, (T*)du, B, H, W, C);
        if (walk)
        hipLaunchKernelGGL((dwconv3x3_wgrad_walk_kernel<T>), dim3(q.nblk, q.slabs), dim3(256), 0, st, (const T*)x, (const T*)du, ws,
                           B, H, W, C, q.ch, q.rl, q.R, q.nseg);
        else
        hipLaunchKernelGGL((dwconv3x3_wgrad_kernel<T>), dim3(p.nblk, p.slabs), dim3(256), 0, st, (const T*)x, (const T*)du, ws,
                           B, H, W, C, p.ch, p.rl, p.units_per_blk);
        if (dw_use_walk()) dw_walk_launch<T, 1>(st, (const T*)du, w, (const float*)nullptr, 0, (const T*)nullptr, (T*)dx, B, H, W, C);
        else
        hipLaunchKernelGGL((dwconv3x3_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, (const T*)du, w, (const float*)nullptr, 0,
                           (const T*)nullptr, (T*)dx, B, H, W, C);
    })
    SEGF_CHECK_LAUNCH();
    if (!dw) return 0;          // deferred: the partial sums [blocks][10 C] stay in ws for segf_colreduce_finalize_grouped (scatter_c = C)
    colreduce_finalize_launch(ws, p.nblk, 10 * (int64_t)C, sums, st);
    SEGF_CHECK_LAUNCH();
    hipLaunchKernelGGL(dw_scatter_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, dw, db);
    SEGF_CHECK_LAUNCH();
    return 0;
}
extern "C" int segf_dwconv3x3_bwd_blocks(int dt, int B, int H, int W, int C) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
    if (dw_small_nch(dt, B, H, W, C)) return B;
    return dw_use_walk() ? dww_plan(B, H, W, C).nblk : dwg_plan(B, H, W, C).nblk;
}

// ---- depthwise 7x7 (ConvNeXt Block.dwconv, models/backbones/convnext.py:29,39; convnextv2.py:88,101) ---------------------
// Column-fixed threads (8-channel chunk per thread), 8 output pixels per strip.  The weights arrive pre-transposed as
// wt[49][C] fp32 so that one kernel row (7 taps x 8 channels) is 14 16-byte loads held in registers while the strip's
// 14 input columns of that row stream through.  FLIP = correlation with the flipped kernel (data gradient).
#define DW7_PIX 8
template <typename T, bool FLIP>
__global__ void __launch_bounds__(256) dwconv7x7_kernel(const T* __restrict__ x, const float* __restrict__ wt,
                                                         const float* __restrict__ bias, T* __restrict__ y, int B, int H, int W, int C) {
    const int nchunk = C / 8;
    const int wg = (W + DW7_PIX - 1) / DW7_PIX;
    const int units = B * H * wg;
    const int64_t g = (int64_t)xcd_block() * 256 + threadIdx.x;
    const int ustep = (int)(((int64_t)gridDim.x * 256) / nchunk);
    const int c0 = (int)(g % nchunk) * 8;
    // channel pairs on float2 vectors -> packed fp32 fmas (49 taps x 8 pixels per strip: the kernel is VALU-bound)
    f32x2_t bs[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) bs[jj] = (!FLIP && bias) ? f32x2_t{bias[c0 + 2 * jj], bias[c0 + 2 * jj + 1]} : f32x2_t{0.f, 0.f};
    for (int u = (int)(g / nchunk); u < units; u += ustep) {
        const int xg = u % wg;
        const int t = u / wg;
        const int yy = t % H;
        const int b = t / H;
        const int x0 = xg * DW7_PIX;
        f32x2_t acc[DW7_PIX][4];
#pragma unroll
        for (int p = 0; p < DW7_PIX; ++p)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[p][jj] = bs[jj];
        for (int ky = 0; ky < 7; ++ky) {
            const int iy = yy + ky - 3;
            if (iy < 0 || iy >= H) continue;            // wave-divergent only at the image border rows
            f32x2_t wk[7][4];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float* wp = wt + (int64_t)(FLIP ? 48 - (ky * 7 + kx) : ky * 7 + kx) * C + c0;
                const float4 wa = *reinterpret_cast<const float4*>(wp), wb = *reinterpret_cast<const float4*>(wp + 4);
                wk[kx][0] = f32x2_t{wa.x, wa.y}; wk[kx][1] = f32x2_t{wa.z, wa.w};
                wk[kx][2] = f32x2_t{wb.x, wb.y}; wk[kx][3] = f32x2_t{wb.z, wb.w};
            }
            const T* row = x + (((int64_t)b * H + iy) * W) * C + c0;
            // unconditional, clamped loads in two batches of 7 columns, all issued before the first use
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                Raw8<T> raw[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const int ix = x0 + half * 7 + q - 3;
                    raw[q] = load8_raw<T>(row + (int64_t)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C);
                }
                SEGF_LOADS_ISSUED();
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const int cx = half * 7 + q;
                    const int ix = x0 + cx - 3;
                    f32x2_t v[4];
                    zero_unless(raw[q], ix >= 0 && ix < W);
                    unpack8v<T>(raw[q], v);
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) {
                        const int p = cx - kx;
                        if (p >= 0 && p < DW7_PIX) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) acc[p][jj] = wk[kx][jj] * v[jj] + acc[p][jj];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < DW7_PIX; ++p)
            if (x0 + p < W) store8v<T>(y + (((int64_t)b * H + yy) * W + x0 + p) * C + c0, acc[p]);
    }
}

// weight / bias gradient, one kernel row per grid.z: sums[ky][kx<7][C] = sum_p dy[p][c] x[p + (ky-3, kx-3)][c];
// slot 7 of row ky == 3 carries db[c] = sum_p dy[p][c].  Same block layout / deterministic tail as colreduce.h.
struct Dw7Plan { int ch, rl, slabs, nblk, units_per_blk; };
static inline Dw7Plan dw7_plan(int B, int H, int W, int C) {
    Dw7Plan p;
    const int nchunk = C / 8;
    p.ch = nchunk < 256 ? nchunk : 256;
    p.rl = 256 / p.ch;
    p.slabs = (nchunk + p.ch - 1) / p.ch;
    const int64_t units = (int64_t)B * H * ((W + DW7_PIX - 1) / DW7_PIX);
    int64_t want = cdiv64(units, (int64_t)p.rl * 2);
    int cap = 256 / p.slabs;
    if (cap < 1) cap = 1;
    p.nblk = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    p.units_per_blk = (int)cdiv64(units, p.nblk);
    return p;
}
template <typename T>
__global__ void __launch_bounds__(256) dwconv7x7_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               float* __restrict__ partial, int B, int H, int W, int C, int ch,
                                                               int rl, int units_per_blk) {
    __shared__ float red[256 * 8];
    const int tx = threadIdx.x % ch, ty = threadIdx.x / ch;
    const int c0 = (blockIdx.y * ch + tx) * 8;
    const bool active = ty < rl && c0 < C;
    const int ky = blockIdx.z;
    const int wg = (W + DW7_PIX - 1) / DW7_PIX;
    const int units = B * H * wg;
    f32x2_t ap[8][4];          // channel pairs: packed fp32 fmas
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) ap[o][jj] = f32x2_t{0.f, 0.f};
    // units are dealt to the workgroups round-robin in groups of rl strips (not as one contiguous range each): the workgroups
    // that run at the same time on an XCD then walk ADJACENT image rows, and the two re-reads of every input row (as the row
    // above / below of its neighbours) hit that XCD's L2.  With contiguous ranges every workgroup streamed its own 16 rows,
    // 128 of them per 4 MB L2, and FETCH_SIZE showed x fetched three times (2.13 GB for 1.07 GB at [128,128,128,128]).
    (void)units_per_blk;
    if (active) {
        for (int u = (int)xcd_block() * rl + ty; u < units; u += (int)gridDim.x * rl) {
            const int xg = u % wg;
            const int t = u / wg;
            const int yy = t % H;
            const int b = t / H;
            const int x0 = xg * DW7_PIX;
            const int iy = yy + ky - 3;
            Raw8<T> graw[DW7_PIX];
#pragma unroll
            for (int p = 0; p < DW7_PIX; ++p)
                graw[p] = load8_raw<T>(dy + (((int64_t)b * H + yy) * W + (x0 + p < W ? x0 + p : W - 1)) * C + c0);
            SEGF_LOADS_ISSUED();
            f32x2_t gq[DW7_PIX][4];
#pragma unroll
            for (int p = 0; p < DW7_PIX; ++p) {
                zero_unless(graw[p], x0 + p < W);
                unpack8v<T>(graw[p], gq[p]);
                if (ky == 3) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) ap[7][jj] += gq[p][jj];
                }
            }
            if (iy < 0 || iy >= H) continue;
            const T* row = x + (((int64_t)b * H + iy) * W) * C + c0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                Raw8<T> raw[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const int ix = x0 + half * 7 + q - 3;
                    raw[q] = load8_raw<T>(row + (int64_t)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C);
                }
                SEGF_LOADS_ISSUED();
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const int cx = half * 7 + q;
                    const int ix = x0 + cx - 3;
                    f32x2_t v[4];
                    zero_unless(raw[q], ix >= 0 && ix < W);
                    unpack8v<T>(raw[q], v);
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) {
                        const int p = cx - kx;
                        if (p >= 0 && p < DW7_PIX) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) ap[kx][jj] = gq[p][jj] * v[jj] + ap[kx][jj];
                        }
                    }
                }
            }
        }
    }
    float acc[8][8];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { acc[o][2 * jj] = ap[o][jj].x; acc[o][2 * jj + 1] = ap[o][jj].y; }
    // partial[(ky * nblk + blk)][8][C]
    float* pz = partial + (int64_t)ky * gridDim.x * 8 * C;
    colreduce_block_tail<8>(acc, active, ty == 0 && c0 < C, tx, ch, rl, c0, 8, C, pz, red);
}
// sums[7][8][C] -> dw[C][49], db[C]
__global__ void dw7_scatter_kernel(const float* __restrict__ sums, int C, float* __restrict__ dw, float* __restrict__ db) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx) dw[c * 49 + ky * 7 + kx] = sums[((int64_t)ky * 8 + kx) * C + c];
    if (db) db[c] = sums[((int64_t)3 * 8 + 7) * C + c];
}

static inline int dw7_blocks(int B, int H, int W, int C) {
    // strips per thread: 4 on large maps, fewer while that would leave under ~8 workgroups per CU -- a strip is a serial chain of
    // 7 x 3 load round trips and 2744 packed multiply-adds (~10 us): [8, 40, 40, 768] ran 150 workgroups x 4 strips = 45 us
    const int64_t units = (int64_t)B * H * ((W + DW7_PIX - 1) / DW7_PIX);
    int upt = (int)(units * (C / 8) / 256 / 2048);
    upt = upt < 1 ? 1 : (upt > 4 ? 4 : upt);
    return colfixed_blocks(units, C / 8, upt, 16384);
}
static inline int dw7_check(int B, int H, int W, int C, const void* a, const void* b) {
    if (C <= 0 || C % 8 != 0 || ((uintptr_t)a % 16) || ((uintptr_t)b % 16)) return SEGF_ERR_SHAPE;
    if ((int64_t)B * H * W >= (1ll << 31)) return SEGF_ERR_SHAPE;
    return 0;
}

extern "C" int segf_dwconv7x7_fwd(int dt, int B, int H, int W, int C, const void* x, const float* wt, const float* bias, void* y,
                                  void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (dw7_check(B, H, W, C, x, y) || ((uintptr_t)wt % 16)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    SEGF_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL((dwconv7x7_kernel<T, false>), dim3(dw7_blocks(B, H, W, C)), dim3(256), 0, st, (const T*)x, wt, bias, (T*)y, B, H, W, C);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}
extern "C" int64_t segf_dwconv7x7_bwd_ws(int B, int H, int W, int C) {
    Dw7Plan p = dw7_plan(B, H, W, C > 0 ? C : 8);
    return (int64_t)7 * p.nblk * 8 * C + 56 * (int64_t)C;
}
// dx = conv^T(dy) (nullable: skipped when dx == nullptr); dw[C][49], db[C] fp32
extern "C" int segf_dwconv7x7_bwd(int dt, int B, int H, int W, int C, const void* x, const float* wt, const void* dy, void* dx,
                                  float* dw, float* db, float* ws, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (dw7_check(B, H, W, C, x, dy) || ((uintptr_t)wt % 16) || ((uintptr_t)dx % 16)) return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Dw7Plan p = dw7_plan(B, H, W, C);
    float* sums = ws + (int64_t)7 * p.nblk * 8 * C;
    SEGF_DISPATCH_DT(dt, T, {
        if (dx)
            hipLaunchKernelGGL((dwconv7x7_kernel<T, true>), dim3(dw7_blocks(B, H, W, C)), dim3(256), 0, st, (const T*)dy, wt,
                               (const float*)nullptr, (T*)dx, B, H, W, C);
        hipLaunchKernelGGL((dwconv7x7_wgrad_kernel<T>), dim3(p.nblk, p.slabs, 7), dim3(256), 0, st, (const T*)x, (const T*)dy, ws, B,
                           H, W, C, p.ch, p.rl, p.units_per_blk);
    })
    SEGF_CHECK_LAUNCH();
    colreduce_finalize_launch(ws, p.nblk, 8 * (int64_t)C, sums, st, 7);          // the seven kernel rows as one batched launch
    SEGF_CHECK_LAUNCH();
    hipLaunchKernelGGL(dw7_scatter_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, dw, db);
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
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
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
// fp32 NCHW image input (Cin small, e.g. 3): a thread builds 8 consecutive columns of one im2col row (scalar gathers from
// the image, which stays L2-resident) and issues ONE 16-byte store; pad columns [K, ldcol) are zeroed.  ldcol % 8 == 0.
// Column-fixed threads: the (channel, ky, kx) of a thread's eight columns are decoded once; per row the eight gathers are
// unconditional (clamped address, masked value) and issued together.
template <typename T>
__global__ void __launch_bounds__(256) im2col_nchw_kernel(const float* __restrict__ x, T* __restrict__ col, int64_t ldcol, int B,
                                                           int H, int W, int Cin, int kh, int kw, int stride, int pad, int Ho,
                                                           int Wo) {
    const int nch = (int)(ldcol / 8);
    const int64_t rows = (int64_t)B * Ho * Wo;
    const int K = kh * kw * Cin;
    const int64_t g = (int64_t)xcd_block() * 256 + threadIdx.x;
    const int64_t rstep = ((int64_t)gridDim.x * 256) / nch;
    const int ch = (int)(g % nch);
    int kyj[8], kxj[8], cij[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kcol = ch * 8 + j;
        const int kk = kcol / Cin;
        cij[j] = kcol < K ? kcol % Cin : -1;
        kxj[j] = kk % kw; kyj[j] = kk / kw;
    }
    for (int64_t m = g / nch; m < rows; m += rstep) {
        const uint32_t t2 = (uint32_t)m / (uint32_t)Wo;       // rows < 2^31 (checked on the host)
        const int ox = (int)((uint32_t)m - t2 * (uint32_t)Wo);
        const int64_t b = t2 / (uint32_t)Ho;
        const int oy = (int)(t2 - (uint32_t)b * (uint32_t)Ho);
        const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
        float v[8];
        bool ok[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int iy = iy0 + kyj[j], ix = ix0 + kxj[j];
            ok[j] = cij[j] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            v[j] = x[((b * Cin + (cij[j] < 0 ? 0 : cij[j])) * H + iyc) * W + ixc];
        }
        SEGF_LOADS_ISSUED();
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ok[j] ? v[j] : 0.f;
        store8<T>(col + m * ldcol + ch * 8, v);
    }
}
// Row-staged form of the same: a workgroup owns one output row (b, oy).  Its kh input rows of all Cin planes are read once,
// coalesced, into LDS (as T); the row's Wo x ldcol/8 chunks are then assembled from LDS and stored with 16-byte writes.  The
// gather form above reads every image element ~k*k/stride^2 times through uncoalesced 4-byte loads and runs at ~2 TB/s of
// output; this one is bound by the 16-byte stores.  Needs kh * Cin * W * sizeof(T) of LDS (21 KB for the 7x7 stride-4 stem).
#define IM2COL_ROW_PAD 2      // elements (one dword for bf16)
template <typename T>
__global__ void __launch_bounds__(256) im2col_nchw_rows_kernel(const float* __restrict__ x, T* __restrict__ col, int64_t ldcol, int B,
                                                                int H, int W, int Cin, int kh, int kw, int stride, int pad, int Ho,
                                                                int Wo) {
    extern __shared__ unsigned char im2col_lds[];
    T* rows = reinterpret_cast<T*>(im2col_lds);                 // [ky][ci][W + IM2COL_ROW_PAD]
    // (rows W * sizeof(T) = a multiple of 256 bytes apart put the same column of every (ky, ci) row on the same LDS bank: the
    // 20 chunk lanes of one output pixel read 8 values each from ~7 x 3 rows -> 5-way conflicts; one extra dword per row
    // moves row r to bank r)
    const int WP = W + IM2COL_ROW_PAD;
    const int bo = xcd_block();
    const int oy = bo % Ho, b = bo / Ho;
    const int iy0 = oy * stride - pad;
    const int nrow = kh * Cin;
    for (int i = threadIdx.x; i < nrow * (W / 4); i += 256) {           // W % 4 == 0 (checked on the host)
        const int rr = i / (W / 4), x4 = (i - rr * (W / 4)) * 4;
        const int ky = rr / Cin, ci = rr - ky * Cin;
        const int iy = iy0 + ky;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H) v = *reinterpret_cast<const float4*>(x + (((int64_t)b * Cin + ci) * H + iy) * W + x4);
        T* dst = rows + (int64_t)rr * WP + x4;
        stf<T>(dst, v.x); stf<T>(dst + 1, v.y); stf<T>(dst + 2, v.z); stf<T>(dst + 3, v.w);
    }
    // column -> (staged row offset, kx): the two divisions by run-time constants per output VALUE made this kernel VALU-bound
    // (0.49 ms for the 7x7 stem at batch 128); they depend on the column only, so a 1 KB table per workgroup replaces them
    __shared__ int ktab[2][256];
    const int nch = (int)(ldcol / 8), K = kh * kw * Cin;
    for (int kcol = threadIdx.x; kcol < nch * 8 && kcol < 256; kcol += 256) {
        const int kk = kcol / Cin, ci = kcol - kk * Cin;
        const int ky = kk / kw, kx = kk - ky * kw;
        ktab[0][kcol] = kcol < K ? (ky * Cin + ci) * WP : -1;
        ktab[1][kcol] = kx;
    }
    __syncthreads();
    T* out = col + ((int64_t)b * Ho + oy) * Wo * ldcol;
    const bool tab_ok = nch * 8 <= 256;
    if (sizeof(T) == 2 && tab_ok && nch <= 256) {
        // column-fixed form: a thread keeps ONE 16-byte chunk of the row (its 8 (staged-row offset, kx) pairs live in registers)
        // and walks the output pixels ppp at a time; the bf16 values go from LDS to the store as raw halfwords.  (With the chunk
        // changing per iteration the loop spent ~150 instructions per store on a division, 16 table reads and two conversions.)
        const int ppp = 256 / nch;                                        // pixels per pass
        const int ch = (int)threadIdx.x % nch, p0 = (int)threadIdx.x / nch;
        int ro[8], kxs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ro[j] = ktab[0][ch * 8 + j]; kxs[j] = ktab[1][ch * 8 + j]; }
        const uint16_t* rows16 = reinterpret_cast<const uint16_t*>(rows);
        if (p0 < ppp) {
            for (int ox = p0; ox < Wo; ox += ppp) {
                const int ix0 = ox * stride - pad;
                uint32_t h[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ix = ix0 + kxs[j];
                    const bool ok = ro[j] >= 0 && ix >= 0 && ix < W;
                    const uint32_t raw = rows16[ok ? ro[j] + ix : 0];
                    h[j] = ok ? raw : 0u;
                }
                uint4 o;
                o.x = h[0] | (h[1] << 16); o.y = h[2] | (h[3] << 16); o.z = h[4] | (h[5] << 16); o.w = h[6] | (h[7] << 16);
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(out) + (int64_t)ox * ldcol + ch * 8) = o;
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < Wo * nch; i += 256) {
        const int ox = i / nch, ch = i - ox * nch;
        const int ix0 = ox * stride - pad;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kcol = ch * 8 + j;
            int ro, kx;
            if (tab_ok) { ro = ktab[0][kcol]; kx = ktab[1][kcol]; }
            else {
                const int kk = kcol / Cin, ci = kcol - kk * Cin;
                const int ky = kk / kw;
                kx = kk - ky * kw;
                ro = kcol < K ? (ky * Cin + ci) * WP : -1;
            }
            const int ix = ix0 + kx;
            const bool ok = ro >= 0 && ix >= 0 && ix < W;
            v[j] = ok ? ldf<T>(rows + ro + ix) : 0.f;
        }
        store8<T>(out + (int64_t)ox * ldcol + ch * 8, v);
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
    if (rows >= (1ll << 31)) return SEGF_ERR_SHAPE;
    if (in_nchw_f32) {
        const int64_t esz0 = dt == SEGF_BF16 ? 2 : 4;
        if (ldcol % 8 || ((uintptr_t)col % 16) || ((ldcol * esz0) % 16)) return SEGF_ERR_SHAPE;
        const int64_t lds_bytes = (int64_t)kh * Cin * (W + IM2COL_ROW_PAD) * esz0;
        if (W % 4 == 0 && ((uintptr_t)x % 16) == 0 && lds_bytes <= 64 * 1024 && (int64_t)B * Ho <= 0x7fffffff) {
            SEGF_DISPATCH_DT(dt, T, {
                hipLaunchKernelGGL((im2col_nchw_rows_kernel<T>), dim3((unsigned)(B * Ho)), dim3(256), (size_t)lds_bytes, st,
                                   (const float*)x, (T*)col, ldcol, B, H, W, Cin, kh, kw, stride, pad, Ho, Wo);
            })
            SEGF_CHECK_LAUNCH();
            return 0;
        }
        const int blocks = colfixed_blocks(rows, (int)(ldcol / 8), 4, 16384);
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

// Fast form for kernel <= NC * stride per axis (every conv of the path: k = s -> NC 1; k3 s2, k7 s4 -> NC 2): an input pixel is
// covered by at most NC x NC windows, ky = (iy + pad) % stride + j * stride.  The NC*NC candidate loads are unconditional
// (clamped address, masked value) and in flight together; the generic kernel below walks all kh x kw taps with a branch each.
template <typename T, int NC>
__global__ void __launch_bounds__(256) col2im_nc_kernel(const T* __restrict__ dcol, int64_t ldcol, T* __restrict__ dx, int B, int H,
                                                         int W, int Cin, int kh, int kw, int stride, int pad, int Ho, int Wo) {
    const int nch = Cin / 8;
    const int64_t total = (int64_t)B * H * W * nch;
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t i32 = (uint32_t)idx;                    // total < 2^32 (checked on the host)
        const uint32_t tq = i32 / (uint32_t)nch;
        const int ch = (int)(i32 - tq * (uint32_t)nch);
        const uint32_t t2 = tq / (uint32_t)W;
        const int ix = (int)(tq - t2 * (uint32_t)W);
        const int64_t b = t2 / (uint32_t)H;
        const int iy = (int)(t2 - (uint32_t)b * (uint32_t)H);
        const int ky0 = (iy + pad) % stride, kx0 = (ix + pad) % stride;
        Raw8<T> raw[NC][NC];
        bool ok[NC][NC];
#pragma unroll
        for (int jy = 0; jy < NC; ++jy) {
            const int ky = ky0 + jy * stride, oy = (iy + pad - ky) / stride;
            const bool vy = ky < kh && iy + pad - ky >= 0 && oy < Ho;
#pragma unroll
            for (int jx = 0; jx < NC; ++jx) {
                const int kx = kx0 + jx * stride, ox = (ix + pad - kx) / stride;
                ok[jy][jx] = vy && kx < kw && ix + pad - kx >= 0 && ox < Wo;
                const int oyc = ok[jy][jx] ? oy : 0, oxc = ok[jy][jx] ? ox : 0, tap = ok[jy][jx] ? ky * kw + kx : 0;
                raw[jy][jx] = load8_raw<T>(dcol + ((b * Ho + oyc) * Wo + oxc) * ldcol + (int64_t)tap * Cin + ch * 8);
            }
        }
        SEGF_LOADS_ISSUED();
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int jy = 0; jy < NC; ++jy)
#pragma unroll
            for (int jx = 0; jx < NC; ++jx) {
                float v[8];
                unpack8(raw[jy][jx], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += ok[jy][jx] ? v[j] : 0.f;
            }
        store8<T>(dx + ((b * H + iy) * W + ix) * Cin + ch * 8, acc);
    }
}

// dx[b][iy][ix][ci] = sum over (ky,kx) with (iy+pad-ky) % stride == 0 ... of dcol[(b,oy,ox)][(ky,kx,ci)]  (gather form, no atomics)
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcol, int64_t ldcol, T* __restrict__ dx, int B, int H, int W, int Cin, int kh,
                              int kw, int stride, int pad, int Ho, int Wo) {
    const int nch = Cin / 8;
    const int64_t total = (int64_t)B * H * W * nch;
    for (int64_t idx = (int64_t)xcd_block() * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
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
    const int nc = (kh > kw ? kh : kw) <= stride ? 1 : ((kh > kw ? kh : kw) <= 2 * stride ? 2 : 0);
    const bool small = (int64_t)B * H * W * (Cin / 8) < (1ll << 32);
    SEGF_DISPATCH_DT(dt, T, {
        if (nc == 1 && small)
            hipLaunchKernelGGL((col2im_nc_kernel<T, 1>), dim3(blocks), dim3(256), 0, st, (const T*)dcol, ldcol, (T*)dx, B, H, W, Cin, kh, kw, stride, pad, Ho, Wo);
        else if (nc == 2 && small)
            hipLaunchKernelGGL((col2im_nc_kernel<T, 2>), dim3(blocks), dim3(256), 0, st, (const T*)dcol, ldcol, (T*)dx, B, H, W, Cin, kh, kw, stride, pad, Ho, Wo);
        else
            hipLaunchKernelGGL((col2im_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)dcol, ldcol, (T*)dx, B, H, W, Cin, kh, kw, stride, pad, Ho, Wo);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}
