// Fused final bilinear upsample + CrossEntropy + Dice, and fused upsample + argmax + confusion matrix.
//   reference: models/build_models.py:65 (F.interpolate to the input size), engine.py:10-15 (criterion),
//              util/losses.py:126-177 (build_target / dice_coeff / multiclass_dice_coeff / dice_loss),
//              engine.py:89-91 + util/utils.py:99-109 + util/metrics.py:24-27 (evaluate).
// The reference materialises fp32 full-resolution logits, a softmax copy and a one-hot copy (3 x 157 MB per
// 512^2 x 150-class image) and then loops over batch x class in Python.  Closed form:
//   loss = CE + 1 - mean_{b,c} (2 I + eps) / (P + T + eps),   (P + T == 0  =>  denominator := 2 I)
//   I = sum_valid p_c [t=c], P = sum_valid p_c, T = sum_valid [t=c]
// Layout of the work: a wave owns one *cell* = the sc x sc full-resolution pixels that interpolate between the same
// four low-resolution taps (sc = H/h, a power of two; align_corners=False puts cell (j,k) at Y in [sc*j+sc/2, sc*j+3sc/2)).
// The classes are spread over the 64 lanes (<= 3 per lane); the four tap rows are read once per cell as coalesced
// class rows, every pixel of the cell is interpolated in registers, softmax statistics are DPP wave reductions, and the
// per-(image, class) Dice sums stay in lane registers.  Forward reads the low-res logits ~once plus the labels; the
// backward recomputes the softmax per pixel and scatters d(logit) back onto the four taps through an LDS tile
// (8x8 taps per workgroup, 9x9 cells incl. halo, four parity colours -> no atomics, bitwise reproducible), writing the
// gradient of the LOW-resolution logits directly: no full-resolution tensor exists in either direction.
// Non-power-of-two ratios fall back to the per-pixel kernels + segf_bilinear_bwd (still HIP).
// Deterministic: per-block partials + fixed-order finalize, no float atomics.
#include <stdlib.h>
#include "colreduce.h"

#include "loss_geom.h"

static inline int pow2_scale(int h, int w, int H, int W) {
    // sc >= 1 when H == sc*h, W == sc*w and sc is a power of two (exact source-index arithmetic); else 0
    if (h <= 0 || w <= 0 || H % h || W % w) return 0;
    const int sc = H / h;
    if (W / w != sc || (sc & (sc - 1)) || sc > 8) return 0;   // sc*sc labels live one per lane
    return sc;
}

// ---- per-lane class rows -------------------------------------------------------------------------------------
template <typename T, int NS>
__device__ __forceinline__ void load_tap(const T* __restrict__ p, int lane, int C, float (&t)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {     // unconditional loads from a clamped index: all taps of a cell stay in flight together
        const int c = lane + 64 * s;
        const float v = ldf<T>(p + (c < C ? c : C - 1));
        t[s] = c < C ? v : 0.f;
    }
}

// z[s] = upsampled logit of class lane + 64*s at full-res pixel (Y, X); invalid class slots get -inf  (generic path)
template <typename T, int NS, bool ATEN = false>
__device__ __forceinline__ void pixel_logits(const T* __restrict__ img, const LossGeom& g, int Y, int X, int lane, float (&z)[NS]) {
    if (g.h == g.H && g.w == g.W) {
        const T* p = img + ((int64_t)Y * g.w + X) * g.ldl;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = lane + 64 * s;
            const float v = ldf<T>(p + (c < g.C ? c : g.C - 1));
            z[s] = c < g.C ? v : -INFINITY;
        }
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
        const int c = lane + 64 * s, ci = c < g.C ? c : g.C - 1;
        const float a = ldf<T>(p00 + ci), b = ldf<T>(p01 + ci), cc = ldf<T>(p10 + ci), d = ldf<T>(p11 + ci);
        // ATEN (evaluation kernels): the operation order of the reference's CPU F.interpolate, so that arg max ties resolve alike
        const float v = ATEN ? bilinear_aten(a, b, cc, d, ly, lx) : (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * cc + lx * d);
        z[s] = c < g.C ? v : -INFINITY;
    }
}

// softmax over the wave: p[s]; returns log-sum-exp (wave-uniform)
template <int NS>
__device__ __forceinline__ float wave_softmax(const float (&z)[NS], float (&p)[NS]) {
    float mx = z[0];
#pragma unroll
    for (int s = 1; s < NS; ++s) mx = fmaxf(mx, z[s]);
    mx = wave_max_all(mx);
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) { p[s] = __expf(z[s] - mx); sum += p[s]; }
    sum = wave_sum_all(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int s = 0; s < NS; ++s) p[s] *= inv;
    return mx + __logf(sum);
}

// softmax with a wave-uniform upper bound `mb` of max(z) in place of the exact maximum (every pixel of a cell is a convex
// combination of the cell's four taps, so the taps' maximum bounds all of them: one wave reduction per CELL instead of
// one per pixel).  Mathematically identical (softmax is shift invariant); if the bound is so loose that the sum
// underflows, the exact-maximum path is taken instead (wave-uniform branch, practically never).
template <int NS>
__device__ __forceinline__ float wave_softmax_bounded(const float (&z)[NS], float mb, float (&p)[NS]) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) { p[s] = __expf(z[s] - mb); sum += p[s]; }
    sum = wave_sum_all(sum);
    if (!(sum > 1e-30f)) return wave_softmax<NS>(z, p);
    const float inv = 1.f / sum;
#pragma unroll
    for (int s = 0; s < NS; ++s) p[s] *= inv;
    return mb + __logf(sum);
}
template <int NS>
__device__ __forceinline__ float cell_max_bound(const float (&t00)[NS], const float (&t01)[NS], const float (&t10)[NS],
                                                const float (&t11)[NS], int lane, int C) {
    float mx = -INFINITY;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (lane + 64 * s < C) mx = fmaxf(mx, fmaxf(fmaxf(t00[s], t01[s]), fmaxf(t10[s], t11[s])));
    return wave_max_all(mx);
}

// A cell = the sc x sc full-res pixels between low-res taps (cj, ck) .. (cj+1, ck+1); cj in [-1, h-1] (taps clamp).
struct Cell { int cj, ck, y0, y1, x0, x1; };
__device__ __forceinline__ Cell make_cell(int cj, int ck, int h, int w, int halo) {
    // halo == 0 (sc == 1, no resize): the cell is the pixel itself, all four taps coincide
    Cell c; c.cj = cj; c.ck = ck;
    c.y0 = cj < 0 ? 0 : cj; c.y1 = cj + halo > h - 1 ? h - 1 : cj + halo;
    c.x0 = ck < 0 ? 0 : ck; c.x1 = ck + halo > w - 1 ? w - 1 : ck + halo;
    return c;
}
template <typename T, int NS>
__device__ __forceinline__ void load_cell_taps(const T* __restrict__ img, const LossGeom& g, const Cell& c, int lane,
                                               float (&t00)[NS], float (&t01)[NS], float (&t10)[NS], float (&t11)[NS]) {
    load_tap<T, NS>(img + ((int64_t)c.y0 * g.w + c.x0) * g.ldl, lane, g.C, t00);
    load_tap<T, NS>(img + ((int64_t)c.y0 * g.w + c.x1) * g.ldl, lane, g.C, t01);
    load_tap<T, NS>(img + ((int64_t)c.y1 * g.w + c.x0) * g.ldl, lane, g.C, t10);
    load_tap<T, NS>(img + ((int64_t)c.y1 * g.w + c.x1) * g.ldl, lane, g.C, t11);
}
// fractional weight of pixel coordinate d inside its cell (exact for power-of-two sc; matches bilinear_src)
__device__ __forceinline__ float cell_frac(int d, int in, int out) {
    int i0, i1; float l;
    bilinear_src(d, in, out, 0, i0, i1, l);
    return l;
}

// Labels of the cell's sc x sc pixels, one per lane (lane = a * sc + bb), fetched with a single vector load:
// code = class index, or -1 (skip: outside the image or equal to `skip_label`), or -2 (out-of-range label).
__device__ __forceinline__ int cell_label_codes(const int64_t* __restrict__ tg, const LossGeom& g, int sc, int off, int cj, int ck,
                                                int lane, int64_t skip_label, int64_t* raw = nullptr) {
    int code = -1;
    if (lane < sc * sc) {
        const int a = lane / sc, bb = lane - a * sc;
        const int Y = sc * cj + off + a, X = sc * ck + off + bb;
        if (Y >= 0 && Y < g.H && X >= 0 && X < g.W) {
            const int64_t t = tg[(int64_t)Y * g.W + X];
            if (raw) *raw = t;
            code = t == skip_label ? -1 : ((t < 0 || t >= g.C) ? -2 : (int)t);
        }
    }
    return code;
}

// ---- forward: partial[blk][b][3C+4] = {I[C], P[C], T[C], ce_sum, w_sum, n_valid, bad} -------------------------------
template <int NS>
__device__ __forceinline__ void fwd_block_tail(float (&aI)[NS], float (&aP)[NS], float (&aT)[NS], float ce, float wsum,
                                               float nvalid, float bad, int lane, int wave, int C, float* __restrict__ dst) {
    __shared__ float red[4][3 * 64 * NS + 4];
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
    for (int i = threadIdx.x; i < 3 * 64 * NS + 4; i += LS_THREADS) {
        const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        if (i >= 3 * 64 * NS) dst[3 * C + (i - 3 * 64 * NS)] = v;
        else {
            const int which = i / (64 * NS), c = i - which * 64 * NS;
            if (c < C) dst[which * C + c] = v;
        }
    }
}

// ---- 16 pixels of a cell at a time -----------------------------------------------------------------------------------
// Per-pixel wave reductions (sum of exponentials, <G, p>) dominate the per-pixel form: 11 dependent instructions each.
// Here a wave keeps the 16 pixels' class values in registers and reduces their 16 per-lane partial sums TOGETHER, as a
// transposing reduction: v_permlane32_swap / v_permlane16_swap halve the number of live values while crossing the
// 32- and 16-lane boundaries (8 + 4 swap-add pairs), four DPP butterflies finish the 16-lane rows for the remaining
// 4 values, and 16 v_readlane turn the results into wave-uniform scalars: 56 instructions per 16 pixels instead of 176.
// (inline asm: hipcc 7.2 miscompiles the two-result __builtin_amdgcn_permlane{16,32}_swap when both results feed
// arithmetic -- it adds result 0 to itself; the s_nops cover the VALU-write -> permlane-read wait states, which hipcc does
// not insert around asm statements)
__device__ __forceinline__ float swap_add32(float a, float b) {      // [a.lo + a.hi | b.lo + b.hi]
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swap_add16(float a, float b) {      // rows: [a.r0+a.r1 | b.r0+b.r1 | a.r2+a.r3 | b.r2+b.r3]
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32_e32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ void wave_reduce16(const float (&v)[16], float (&out)[16]) {
    float u[8], t[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = swap_add32(v[i], v[i + 8]);     // lanes 0-31 keep pixel i, lanes 32-63 pixel i+8
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = swap_add16(u[i], u[i + 4]);     // 16-lane row r keeps pixel i + 4*(r&1) + 8*(r>>1)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t[i] += dpp_mov<DPP_XOR1>(t[i]); t[i] += dpp_mov<DPP_XOR2>(t[i]);
        t[i] += dpp_mov<DPP_HALF_MIRROR>(t[i]); t[i] += dpp_mov<DPP_MIRROR>(t[i]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i + 4 * (r & 1) + 8 * (r >> 1)] = readlane_f(t[i], 16 * r);
}

// 8-value form (half the live registers of the 16-value form at the same cost per value): lanes 0-31 keep value i,
// lanes 32-63 value i+4; 16-lane row r then keeps value i + 2*(r&1) + 4*(r>>1), i < 2.
__device__ __forceinline__ void wave_reduce8(const float (&v)[8], float (&out)[8]) {
    float u[4], t[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = swap_add32(v[i], v[i + 4]);
#pragma unroll
    for (int i = 0; i < 2; ++i) t[i] = swap_add16(u[i], u[i + 2]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        t[i] += dpp_mov<DPP_XOR1>(t[i]); t[i] += dpp_mov<DPP_XOR2>(t[i]);
        t[i] += dpp_mov<DPP_HALF_MIRROR>(t[i]); t[i] += dpp_mov<DPP_MIRROR>(t[i]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i) out[i + 2 * (r & 1) + 4 * (r >> 1)] = readlane_f(t[i], 16 * r);
}

// Shared front end of the batched kernels: the cell's taps are brought into the "exp2 domain" once,
//   u = (t - mb) * log2(e)        (mb = wave-uniform upper bound of every pixel's maximum logit; invalid class lanes: -1e30)
// so that a pixel's exponent argument is a plain bilinear combination of the four u's (weights sum to one), evaluated
// separably: one fma per row for the left / right columns and ONE fma per (pixel, class) for the column weight, with
// the weights (k + 0.5) / SC folded into the instructions as literals.  exp2(-1e30) == 0 removes the class mask.
template <int NS>
__device__ __forceinline__ void cell_scaled_taps(float (&t00)[NS], float (&t01)[NS], float (&t10)[NS], float (&t11)[NS], float mb,
                                                 int lane, int C, float (&d0)[NS], float (&d1)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool ok = lane + 64 * s < C;
        t00[s] = ok ? (t00[s] - mb) * LS_LOG2E : -1e30f; t01[s] = ok ? (t01[s] - mb) * LS_LOG2E : -1e30f;
        t10[s] = ok ? (t10[s] - mb) * LS_LOG2E : -1e30f; t11[s] = ok ? (t11[s] - mb) * LS_LOG2E : -1e30f;
        d0[s] = t10[s] - t00[s]; d1[s] = t11[s] - t01[s];
    }
}

// forward over cells, SC in {2, 4, 8}: batches of 16 pixels = 16/SC rows of the cell (SC == 2: the 4 pixels, 12 slots idle).
// The pixel loops are branch-free: an ignored / out-of-image pixel runs with weight 0.  If the cell bound is ever too loose
// for a valid pixel (its exponent sum underflows: needs a >88 logit spread between neighbouring taps of one class) the
// wave raises *retry and the exact-maximum per-pixel kernel, launched right behind and idle otherwise, redoes the image set.
template <typename T, int NS, int SC>
__global__ void __launch_bounds__(LS_THREADS, 4) ce_dice_fwd_cells16_kernel(const T* __restrict__ logits, LossGeom g,
                                                                          const int64_t* __restrict__ target, int64_t ignore_index,
                                                                          const float* __restrict__ cw, float* __restrict__ partial,
                                                                          int* __restrict__ retry) {
    constexpr int PB = 8;                                 // pixels per batch (one wave_reduce8)
    constexpr int ROWS = SC >= 8 ? 1 : (SC == 4 ? 2 : SC);     // cell rows per batch
    constexpr int COLS = SC >= 8 ? 8 : SC;                // pixels of a row per batch (SC == 8: a row is one batch)
    constexpr int NPB = ROWS * COLS;                      // live pixels per batch (4 for SC == 2, else 8)
    constexpr int NB = SC * SC / NPB;                     // batches per cell
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    constexpr int off = SC / 2;
    const int ncx = g.w + 1, ncell = (g.h + 1) * ncx;
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * g.H * g.W;
    float aI[NS], aP[NS], aT[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { aI[s] = 0.f; aP[s] = 0.f; aT[s] = 0.f; }
    float cez = 0.f;                                   // per-lane: sum of wt * (exp2-domain logit of the label class) over owned pixels
    float cel2 = 0.f, wsum = 0.f, nvalid = 0.f, bad = 0.f;       // wave-uniform
    for (int cell = blockIdx.x * 4 + wave; cell < ncell; cell += gridDim.x * 4) {
        const Cell c = make_cell(cell / ncx - 1, cell % ncx - 1, g.h, g.w, 1);
        float t00[NS], t01[NS], t10[NS], t11[NS], d0[NS], d1[NS];
        load_cell_taps<T, NS>(img, g, c, lane, t00, t01, t10, t11);
        const int codes = cell_label_codes(tg, g, SC, off, c.cj, c.ck, lane, ignore_index);
        const float mb = cell_max_bound<NS>(t00, t01, t10, t11, lane, g.C);
        cell_scaled_taps<NS>(t00, t01, t10, t11, mb, lane, g.C, d0, d1);
        if (__builtin_amdgcn_readfirstlane(__any(codes == -2))) bad = 1.f;
#pragma unroll 1
        for (int bi = 0; bi < NB; ++bi) {        // not unrolled: one batch's 24 exponentials + sums live at a time
            // batch bi covers cell rows [row0, row0 + ROWS) x columns [col0, col0 + COLS)
            constexpr int BPR = SC / COLS;                // batches per cell row (1 unless SC > 8)
            const int row0 = (bi / BPR) * ROWS, col0 = (bi % BPR) * COLS;
            float e[PB][NS], ps[PB], tot[PB];
#pragma unroll
            for (int i = NPB; i < PB; ++i) ps[i] = 0.f;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const float ly = (float)(row0 + r + 0.5f) / (float)SC;
                float L[NS], D[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    L[s] = fmaf(ly, d0[s], t00[s]);
                    D[s] = fmaf(ly, d1[s], t01[s]) - L[s];
                }
#pragma unroll
                for (int q = 0; q < COLS; ++q) {
                    const int i = r * COLS + q;
                    const float lx = (float)(col0 + q + 0.5f) / (float)SC;
                    const int t = __builtin_amdgcn_readlane(codes, (row0 + r) * SC + col0 + q);
                    const int tt = t < 0 ? 0 : t;
                    const float wt = t < 0 ? 0.f : (cw ? cw[tt] : 1.f);          // wave-uniform; 0 for ignored pixels
                    float sum = 0.f, zt = 0.f;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const float zq = fmaf(lx, D[s], L[s]);
                        e[i][s] = __builtin_amdgcn_exp2f(zq);
                        sum += e[i][s];
                        zt = lane + 64 * s == tt ? zq : zt;                    // the lane that owns the label class
                    }
                    ps[i] = sum;
                    cez = fmaf(wt, zt, cez);
                }
            }
            wave_reduce8(ps, tot);
            unsigned slow = 0;
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                const int t = __builtin_amdgcn_readlane(codes, (row0 + i / COLS) * SC + col0 + i % COLS);
                const int tt = t < 0 ? 0 : t;
                const bool under = !(tot[i] > 1e-30f);
                if (t >= 0 && under) slow |= 1u << i;                 // scalar
                const float okf = (t >= 0 && !under) ? 1.f : 0.f;
                const float wt = okf * (cw ? cw[tt] : 1.f);
                const float inv = okf * __builtin_amdgcn_rcpf(fmaxf(tot[i], 1e-30f));
                cel2 = fmaf(wt, __builtin_amdgcn_logf(fmaxf(tot[i], 1e-30f)), cel2);
                nvalid += okf; wsum += wt;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const float pr = e[i][s] * inv;
                    aP[s] += pr;
                    const bool own = lane + 64 * s == tt;
                    aI[s] += own ? pr : 0.f;
                    aT[s] += own ? okf : 0.f;
                }
            }
            if (slow && lane == 0) atomicOr(retry, 1);
        }
    }
    // CE = sum wt (lse - z_t) = ln2 * sum wt (log2(tot) - zq_t): the bound mb cancels
    const float ce = LS_LN2 * (cel2 - wave_sum_all(cez));
    fwd_block_tail<NS>(aI, aP, aT, ce, wsum, nvalid, bad, lane, wave, g.C,
                       partial + ((int64_t)blockIdx.x * g.B + b) * (3 * g.C + 4));
}

template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) ce_dice_fwd_cells_kernel(const T* __restrict__ logits, LossGeom g, int sc,
                                                                        const int64_t* __restrict__ target, int64_t ignore_index,
                                                                        const float* __restrict__ cw, float* __restrict__ partial,
                                                                        const int* __restrict__ run_if) {
    // retry pass behind the batched kernel: idle unless it raised the flag.  Agent-scope atomic load (L2): a plain load is a
    // scalar-cache read, and inside a replayed hipGraph that cache can still hold the word from an earlier tenant of the buffer
    if (run_if && __hip_atomic_load(run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int halo = sc > 1 ? 1 : 0, off = sc >> 1;
    // in-cell interpolation weight of pixel offset a: (a + 0.5) / sc -- exact for the power-of-two ratios of this path and
    // equal to bilinear_src's fraction; at the clamped borders both taps coincide, so any weight gives the same value
    const float inv_sc = 1.f / (float)sc, fhalo = (float)halo;
    const int ncx = g.w + halo, ncell = (g.h + halo) * ncx;
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * g.H * g.W;
    float aI[NS], aP[NS], aT[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { aI[s] = 0.f; aP[s] = 0.f; aT[s] = 0.f; }
    float ce = 0.f, wsum = 0.f, nvalid = 0.f, bad = 0.f;
    for (int cell = blockIdx.x * 4 + wave; cell < ncell; cell += gridDim.x * 4) {
        const Cell c = make_cell(cell / ncx - halo, cell % ncx - halo, g.h, g.w, halo);
        float t00[NS], t01[NS], t10[NS], t11[NS];
        load_cell_taps<T, NS>(img, g, c, lane, t00, t01, t10, t11);
        const int codes = cell_label_codes(tg, g, sc, off, c.cj, c.ck, lane, ignore_index);
        const float mb = cell_max_bound<NS>(t00, t01, t10, t11, lane, g.C);
        for (int a = 0; a < sc; ++a) {
            const int Y = sc * c.cj + off + a;
            if (Y < 0 || Y >= g.H) continue;
            const float ly = (a + 0.5f) * inv_sc * fhalo;
            float L[NS], R[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                L[s] = (1.f - ly) * t00[s] + ly * t10[s];
                R[s] = (1.f - ly) * t01[s] + ly * t11[s];
            }
            for (int bb = 0; bb < sc; ++bb) {
                const int t = __builtin_amdgcn_readlane(codes, a * sc + bb);     // wave-uniform
                if (t == -1) continue;                           // ignored / outside
                if (t == -2) { bad = 1.f; continue; }            // the reference raises here (one_hot / cross_entropy)
                const float lx = (bb + 0.5f) * inv_sc * fhalo;
                float z[NS], pr[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) z[s] = (lane + 64 * s) < g.C ? (1.f - lx) * L[s] + lx * R[s] : -INFINITY;
                const float lse = wave_softmax_bounded<NS>(z, mb, pr);
                const float wt = cw ? cw[t] : 1.f;
                nvalid += 1.f; wsum += wt;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    aP[s] += pr[s];
                    if (lane + 64 * s == t) { aI[s] += pr[s]; aT[s] += 1.f; ce += wt * (lse - z[s]); }
                }
            }
        }
    }
    ce = wave_sum_all(ce);
    fwd_block_tail<NS>(aI, aP, aT, ce, wsum, nvalid, bad, lane, wave, g.C,
                       partial + ((int64_t)blockIdx.x * g.B + b) * (3 * g.C + 4));
}

// generic ratio: one wave per full-res pixel at a time
template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) ce_dice_fwd_kernel(const T* __restrict__ logits, LossGeom g,
                                                                  const int64_t* __restrict__ target, int64_t ignore_index,
                                                                  const float* __restrict__ cw, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int64_t npix = (int64_t)g.H * g.W;
    const int64_t nw = (int64_t)gridDim.x * 4;
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
        if (t == ignore_index) continue;
        if (t < 0 || t >= g.C) { bad = 1.f; continue; }
        const int Y = (int)(p / g.W), X = (int)(p - (int64_t)Y * g.W);
        float z[NS], pr[NS];
        pixel_logits<T, NS>(img, g, Y, X, lane, z);
        const float lse = wave_softmax<NS>(z, pr);
        const float wt = cw ? cw[t] : 1.f;
        nvalid += 1.f; wsum += wt;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            aP[s] += pr[s];
            if (lane + 64 * s == (int)t) { aI[s] += pr[s]; aT[s] += 1.f; ce += wt * (lse - z[s]); }
        }
    }
    ce = wave_sum_all(ce);
    fwd_block_tail<NS>(aI, aP, aT, ce, wsum, nvalid, bad, lane, wave, g.C,
                       partial + ((int64_t)blockIdx.x * g.B + b) * (3 * g.C + 4));
}

// stats[b][3C+4] (already summed over blocks) -> stats tail[4] at stats[B*(3C+4)], loss[3] = {total, ce, dice_loss}
__global__ void __launch_bounds__(256) ce_dice_loss_kernel(float* __restrict__ stats, int B, int C, int dice, float* __restrict__ loss) {
    __shared__ float red[256];
    __shared__ float tail[4];
    const int stride = 3 * C + 4;
    float dsum = 0.f;
    for (int i = threadIdx.x; i < B * C; i += 256) {
        const int b = i / C, c = i - b * C;
        const float* st = stats + (int64_t)b * stride;
        const float I = st[c], P = st[C + c], T = st[2 * C + c];
        float sets = P + T;
        if (sets == 0.f) sets = 2.f * I;
        dsum += (2.f * I + LS_EPS) / (sets + LS_EPS);
    }
    red[threadIdx.x] = dsum;
    if (threadIdx.x < 4) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += stats[(int64_t)b * stride + 3 * C + threadIdx.x];
        tail[threadIdx.x] = s;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* st = stats + (int64_t)B * stride;
        st[0] = tail[0]; st[1] = tail[1]; st[2] = tail[2]; st[3] = tail[3];
        const float ce = tail[0] / tail[1];                        // 0/0 = NaN like F.cross_entropy on an all-ignored batch
        const float dl = dice ? 1.f - red[0] / (float)(B * C) : 0.f;
        loss[0] = ce + dl; loss[1] = ce; loss[2] = dl;
    }
}

// ---- backward ----------------------------------------------------------------------------------------------------
// d loss / d I and d loss / d P of the Dice term for this lane's classes
template <int NS>
__device__ __forceinline__ void dice_coefs(const float* __restrict__ stats, int b, int B, int C, int dice, int lane,
                                           float (&gI)[NS], float (&gP)[NS]) {
    const float* st = stats + (int64_t)b * (3 * C + 4);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int c = lane + 64 * s;
        float a = 0.f, bb = 0.f;
        if (c < C && dice) {
            const float I = st[c], P = st[C + c], Tt = st[2 * C + c];
            const float sets = P + Tt;
            if (sets != 0.f) {   // sets == 0: d = (2I+eps)/(2I+eps) == 1 -> zero gradient
                const float nbc = 1.f / (float)(B * C);
                a = -nbc * 2.f / (sets + LS_EPS);
                bb = nbc * (2.f * I + LS_EPS) / ((sets + LS_EPS) * (sets + LS_EPS));
            }
        }
        gI[s] = a; gP[s] = bb;
    }
}
// dz[s] = go * d loss / d z_c for one valid pixel with label t
template <int NS>
__device__ __forceinline__ void pixel_grad(const float (&z)[NS], int t, int lane, const float (&gI)[NS], const float (&gP)[NS],
                                           float wce, float go, float (&dz)[NS], float mb = INFINITY) {
    float pr[NS], G[NS];
    if (mb == INFINITY) wave_softmax<NS>(z, pr); else wave_softmax_bounded<NS>(z, mb, pr);
    float dot = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        G[s] = gP[s] + (lane + 64 * s == t ? gI[s] : 0.f);
        dot += G[s] * pr[s];
    }
    dot = wave_sum_all(dot);
#pragma unroll
    for (int s = 0; s < NS; ++s)
        dz[s] = go * (pr[s] * (G[s] - dot) + wce * (pr[s] - (lane + 64 * s == t ? 1.f : 0.f)));
}

// Workgroup = LS_TILE x LS_TILE low-res taps of one image; it evaluates the (LS_TILE+1)^2 cells that touch them.
template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) ce_dice_bwd_cells_kernel(const T* __restrict__ logits, LossGeom g, int sc,
                                                                        const int64_t* __restrict__ target, int64_t ignore_index,
                                                                        const float* __restrict__ cw, int dice,
                                                                        const float* __restrict__ stats,
                                                                        const float* __restrict__ grad_out, T* __restrict__ dlow,
                                                                        int64_t ldd, const int* __restrict__ run_if) {
    if (run_if && __hip_atomic_load(run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // idle retry pass
    __shared__ float accum[LS_TILE * LS_TILE][64 * NS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int tiles_x = (g.w + LS_TILE - 1) / LS_TILE;
    const int ty0 = (blockIdx.x / tiles_x) * LS_TILE, tx0 = (blockIdx.x % tiles_x) * LS_TILE;
    const int halo = sc > 1 ? 1 : 0, off = sc >> 1;
    const float inv_sc = 1.f / (float)sc, fhalo = (float)halo;
    for (int i = threadIdx.x; i < LS_TILE * LS_TILE * 64 * NS; i += LS_THREADS) (&accum[0][0])[i] = 0.f;
    float gI[NS], gP[NS];
    dice_coefs<NS>(stats, b, g.B, g.C, dice, lane, gI, gP);
    const float go = grad_out ? grad_out[0] : 1.f;
    const float invW = 1.f / stats[(int64_t)g.B * (3 * g.C + 4) + 1];
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * g.H * g.W;
    __syncthreads();
    const int ncl = LS_TILE + halo;            // local cells per edge: lj in [0, ncl), cj = ty0 - halo + lj
    const int ncol = halo ? 4 : 1;             // parity colours (no two cells of one colour share a tap)
    for (int col = 0; col < ncol; ++col) {
        const int pj = col >> 1, pk = col & 1;
        const int nj = halo ? (ncl - pj + 1) / 2 : ncl, nk = halo ? (ncl - pk + 1) / 2 : ncl;
        for (int idx = wave; idx < nj * nk; idx += 4) {
            const int lj = halo ? 2 * (idx / nk) + pj : idx / nk, lk = halo ? 2 * (idx % nk) + pk : idx % nk;
            const int cj = ty0 - halo + lj, ck = tx0 - halo + lk;
            if (cj > g.h - 1 || ck > g.w - 1) continue;
            const Cell c = make_cell(cj, ck, g.h, g.w, halo);
            float t00[NS], t01[NS], t10[NS], t11[NS], A00[NS], A01[NS], A10[NS], A11[NS];
            load_cell_taps<T, NS>(img, g, c, lane, t00, t01, t10, t11);
            int codes = cell_label_codes(tg, g, sc, off, cj, ck, lane, ignore_index);
            if (codes == -2) codes = -1;
            const float mb = cell_max_bound<NS>(t00, t01, t10, t11, lane, g.C);
#pragma unroll
            for (int s = 0; s < NS; ++s) { A00[s] = 0.f; A01[s] = 0.f; A10[s] = 0.f; A11[s] = 0.f; }
            for (int a = 0; a < sc; ++a) {
                const int Y = sc * cj + off + a;
                if (Y < 0 || Y >= g.H) continue;
                const float ly = (a + 0.5f) * inv_sc * fhalo;
                float L[NS], R[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    L[s] = (1.f - ly) * t00[s] + ly * t10[s];
                    R[s] = (1.f - ly) * t01[s] + ly * t11[s];
                }
                for (int bb = 0; bb < sc; ++bb) {
                    const int t = __builtin_amdgcn_readlane(codes, a * sc + bb);     // wave-uniform
                    if (t < 0) continue;
                    const float lx = (bb + 0.5f) * inv_sc * fhalo;
                    float z[NS], dz[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) z[s] = (lane + 64 * s) < g.C ? (1.f - lx) * L[s] + lx * R[s] : -INFINITY;
                    pixel_grad<NS>(z, t, lane, gI, gP, (cw ? cw[t] : 1.f) * invW, go, dz, mb);
                    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        A00[s] = fmaf(w00, dz[s], A00[s]); A01[s] = fmaf(w01, dz[s], A01[s]);
                        A10[s] = fmaf(w10, dz[s], A10[s]); A11[s] = fmaf(w11, dz[s], A11[s]);
                    }
                }
            }
            // taps of this cell that belong to the tile (the others are recomputed by the neighbouring workgroups)
            const int ly0 = c.y0 - ty0, ly1 = c.y1 - ty0, lx0 = c.x0 - tx0, lx1 = c.x1 - tx0;
            const bool vy0 = ly0 >= 0 && ly0 < LS_TILE, vy1 = ly1 >= 0 && ly1 < LS_TILE;
            const bool vx0 = lx0 >= 0 && lx0 < LS_TILE, vx1 = lx1 >= 0 && lx1 < LS_TILE;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int cc = lane + 64 * s;
                if (vy0 && vx0) accum[ly0 * LS_TILE + lx0][cc] += A00[s];
                if (vy0 && vx1) accum[ly0 * LS_TILE + lx1][cc] += A01[s];
                if (vy1 && vx0) accum[ly1 * LS_TILE + lx0][cc] += A10[s];
                if (vy1 && vx1) accum[ly1 * LS_TILE + lx1][cc] += A11[s];
            }
        }
        __syncthreads();
    }
    for (int tap = wave; tap < LS_TILE * LS_TILE; tap += 4) {
        const int y = ty0 + tap / LS_TILE, x = tx0 + tap % LS_TILE;
        if (y >= g.h || x >= g.w) continue;
        T* drow = dlow + (((int64_t)b * g.h + y) * g.w + x) * ldd;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cc = lane + 64 * s;
            if (cc < ldd) stf<T>(drow + cc, cc < g.C ? accum[tap][cc] : 0.f);
        }
    }
}

// Batched backward, SC in {2, 4, 8}: same tile / colour structure as ce_dice_bwd_cells_kernel, but a cell's pixels are
// processed 8 at a time in the exp2 domain with two transposing reductions per batch (sum of exponentials, <G, e>) instead
// of three per-pixel reductions, branch-free pixel loops, and the tap scatter done separably (column weights per pixel,
// row weights once per cell row).  A too-loose cell bound raises *retry; the exact kernel behind redoes the image set.
template <typename T, int NS, int SC>
__global__ void __launch_bounds__(LS_THREADS, 4) ce_dice_bwd_cells8_kernel(const T* __restrict__ logits, LossGeom g,
                                                                            const int64_t* __restrict__ target, int64_t ignore_index,
                                                                            const float* __restrict__ cw, int dice,
                                                                            const float* __restrict__ stats,
                                                                            const float* __restrict__ grad_out, T* __restrict__ dlow,
                                                                            int64_t ldd, int* __restrict__ retry) {
    constexpr int PB = 8;
    constexpr int ROWS = SC >= 8 ? 1 : (SC == 4 ? 2 : SC);
    constexpr int COLS = SC >= 8 ? 8 : SC;
    constexpr int NPB = ROWS * COLS;
    constexpr int NB = SC * SC / NPB;
    constexpr int BPR = SC / COLS;
    constexpr int off = SC / 2;
    __shared__ float accum[LS_TILE * LS_TILE][64 * NS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int tiles_x = (g.w + LS_TILE - 1) / LS_TILE;
    const int ty0 = (blockIdx.x / tiles_x) * LS_TILE, tx0 = (blockIdx.x % tiles_x) * LS_TILE;
    for (int i = threadIdx.x; i < LS_TILE * LS_TILE * 64 * NS; i += LS_THREADS) (&accum[0][0])[i] = 0.f;
    float gI[NS], gP[NS];
    dice_coefs<NS>(stats, b, g.B, g.C, dice, lane, gI, gP);
    const float go = grad_out ? grad_out[0] : 1.f;
    const float invW = 1.f / stats[(int64_t)g.B * (3 * g.C + 4) + 1];
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * g.H * g.W;
    __syncthreads();
    constexpr int ncl = LS_TILE + 1;
    for (int col = 0; col < 4; ++col) {
        const int pj = col >> 1, pk = col & 1;
        const int nj = (ncl - pj + 1) / 2, nk = (ncl - pk + 1) / 2;
        for (int idx = wave; idx < nj * nk; idx += 4) {
            const int lj = 2 * (idx / nk) + pj, lk = 2 * (idx % nk) + pk;
            const int cj = ty0 - 1 + lj, ck = tx0 - 1 + lk;
            if (cj > g.h - 1 || ck > g.w - 1) continue;
            const Cell c = make_cell(cj, ck, g.h, g.w, 1);
            float t00[NS], t01[NS], t10[NS], t11[NS], d0[NS], d1[NS], A00[NS], A01[NS], A10[NS], A11[NS];
            load_cell_taps<T, NS>(img, g, c, lane, t00, t01, t10, t11);
            const int codes = cell_label_codes(tg, g, SC, off, cj, ck, lane, ignore_index);      // -1 / -2: no gradient
            const float mb = cell_max_bound<NS>(t00, t01, t10, t11, lane, g.C);
            cell_scaled_taps<NS>(t00, t01, t10, t11, mb, lane, g.C, d0, d1);
#pragma unroll
            for (int s = 0; s < NS; ++s) { A00[s] = 0.f; A01[s] = 0.f; A10[s] = 0.f; A11[s] = 0.f; }
            unsigned slow = 0;
#pragma unroll 1
            for (int bi = 0; bi < NB; ++bi) {
                const int row0 = (bi / BPR) * ROWS, col0 = (bi % BPR) * COLS;
                float e[PB][NS], ps[PB], dp[PB], tot[PB], dtot[PB];
#pragma unroll
                for (int i = NPB; i < PB; ++i) { ps[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const float ly = (float)(row0 + r + 0.5f) / (float)SC;
                    float L[NS], D[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        L[s] = fmaf(ly, d0[s], t00[s]);
                        D[s] = fmaf(ly, d1[s], t01[s]) - L[s];
                    }
#pragma unroll
                    for (int q = 0; q < COLS; ++q) {
                        const int i = r * COLS + q;
                        const float lx = (float)(col0 + q + 0.5f) / (float)SC;
                        const int t = __builtin_amdgcn_readlane(codes, (row0 + r) * SC + col0 + q);
                        const int tt = t < 0 ? 0 : t;
                        float sum = 0.f, dsum = 0.f;
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            e[i][s] = __builtin_amdgcn_exp2f(fmaf(lx, D[s], L[s]));
                            sum += e[i][s];
                            const float G = gP[s] + (lane + 64 * s == tt ? gI[s] : 0.f);
                            dsum = fmaf(G, e[i][s], dsum);
                        }
                        ps[i] = sum; dp[i] = dsum;
                    }
                }
                wave_reduce8(ps, tot);
                wave_reduce8(dp, dtot);
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    const float ly = (float)(row0 + r + 0.5f) / (float)SC;
                    float Rl[NS], Rr[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) { Rl[s] = 0.f; Rr[s] = 0.f; }
#pragma unroll
                    for (int q = 0; q < COLS; ++q) {
                        const int i = r * COLS + q;
                        const float lx = (float)(col0 + q + 0.5f) / (float)SC;
                        const int t = __builtin_amdgcn_readlane(codes, (row0 + r) * SC + col0 + q);
                        const int tt = t < 0 ? 0 : t;
                        const bool under = !(tot[i] > 1e-30f);
                        if (t >= 0 && under) slow |= 1u;
                        const float okf = (t >= 0 && !under) ? 1.f : 0.f;
                        const float inv = okf * __builtin_amdgcn_rcpf(fmaxf(tot[i], 1e-30f));
                        const float wce = okf * (cw ? cw[tt] : 1.f) * invW;
                        const float c1 = go * inv, k = wce - dtot[i] * inv, c2 = go * wce;
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            const bool own = lane + 64 * s == tt;
                            const float G = gP[s] + (own ? gI[s] : 0.f);
                            const float dz = fmaf(c1 * e[i][s], G + k, own ? -c2 : 0.f);      // go * (p (G - <G,p>) + wce (p - [c==t]))
                            Rl[s] = fmaf(1.f - lx, dz, Rl[s]);
                            Rr[s] = fmaf(lx, dz, Rr[s]);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        A00[s] = fmaf(1.f - ly, Rl[s], A00[s]); A10[s] = fmaf(ly, Rl[s], A10[s]);
                        A01[s] = fmaf(1.f - ly, Rr[s], A01[s]); A11[s] = fmaf(ly, Rr[s], A11[s]);
                    }
                }
            }
            if (slow && lane == 0) atomicOr(retry, 1);
            // the taps live in the exp2 domain: d zq / d logit = log2(e) is NOT wanted -- dz above is already d loss / d logit
            const int ly0 = c.y0 - ty0, ly1 = c.y1 - ty0, lx0 = c.x0 - tx0, lx1 = c.x1 - tx0;
            const bool vy0 = ly0 >= 0 && ly0 < LS_TILE, vy1 = ly1 >= 0 && ly1 < LS_TILE;
            const bool vx0 = lx0 >= 0 && lx0 < LS_TILE, vx1 = lx1 >= 0 && lx1 < LS_TILE;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int cc = lane + 64 * s;
                if (vy0 && vx0) accum[ly0 * LS_TILE + lx0][cc] += A00[s];
                if (vy0 && vx1) accum[ly0 * LS_TILE + lx1][cc] += A01[s];
                if (vy1 && vx0) accum[ly1 * LS_TILE + lx0][cc] += A10[s];
                if (vy1 && vx1) accum[ly1 * LS_TILE + lx1][cc] += A11[s];
            }
        }
        __syncthreads();
    }
    for (int tap = wave; tap < LS_TILE * LS_TILE; tap += 4) {
        const int y = ty0 + tap / LS_TILE, x = tx0 + tap % LS_TILE;
        if (y >= g.h || x >= g.w) continue;
        T* drow = dlow + (((int64_t)b * g.h + y) * g.w + x) * ldd;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cc = lane + 64 * s;
            if (cc < ldd) stf<T>(drow + cc, cc < g.C ? accum[tap][cc] : 0.f);
        }
    }
}

// ---- MFMA form of the cell kernels (SC == 4) ---------------------------------------------------------------------------
// The two contractions of the fused loss run on the matrix pipe in exact f32 (v_mfma_f32_16x16x4_f32 == an fmaf chain):
//   interpolation  Z[pixel][class] = sum_tap W[pixel][tap] * U[tap][class]      (K = the cell's 4 taps, M = its 16 pixels)
//   tap scatter    dU[tap][class]  = sum_pix W[pixel][tap] * dZ[pixel][class]   (K = pixels, 4 per instruction)
// and the VALU is left with what only it can do: one exp2 per (pixel, class) and a handful of multiplies.
// Lane = (class column c = lane & 15 of a 16-class tile, row group gq = lane >> 4); the accumulator layout puts pixel
// 4 gq + r = (cell row gq, cell column r) in register r, so the softmax sums are NT register adds plus one 16-lane DPP row
// reduction per pixel, and the dZ registers feed the scatter MFMA as its B operand without any lane movement (register r of
// all four row groups = pixels {r, 4+r, 8+r, 12+r} = the four k-slices of one instruction; A carries the matching weights).
// The label class of a pixel is handled on the side by a "label path" in which the wave's lanes are (tap = c & 3,
// pixel = 4 gq + (c >> 2)): one gathered tap value per lane and a quad reduction give the label logit; I / T (forward) and
// the [c == t] terms of the gradient (backward) go through shared-memory float adds, where only lanes of ONE instruction
// ever collide (a wave owns its histogram / its colour's taps), so the summation order is fixed.

// Front end shared by forward and backward, split in two so that the loads of the NEXT cell are in flight while the current
// one is evaluated: issue() = this lane's tap row (MFMA operand) + its label; finish() = label-class gather (issued before
// the next cell's loads so that waiting for it does not drain them), exp2-domain operands, Z tiles -> exponentials.
template <typename T, int NT>
struct MfmaCellLoads {
    Cell cc;
    float u[NT];
    int64_t traw;
    bool inside;
    __device__ __forceinline__ void issue(const T* __restrict__ img, const int64_t* __restrict__ tg, const LossGeom& g, int cj, int ck,
                                          int c, int gq) {
        cc = make_cell(cj, ck, g.h, g.w, 1);
        const T* pk = img + ((int64_t)((gq >> 1) ? cc.y1 : cc.y0) * g.w + ((gq & 1) ? cc.x1 : cc.x0)) * g.ldl;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int cls = 16 * t + c;
            u[t] = ldf<T>(pk + (cls < g.C ? cls : g.C - 1));
        }
        const int Y = 4 * cj + 2 + gq, X = 4 * ck + 2 + (c >> 2);
        inside = Y >= 0 && Y < g.H && X >= 0 && X < g.W;
        traw = tg[(int64_t)(Y < 0 ? 0 : (Y >= g.H ? g.H - 1 : Y)) * g.W + (X < 0 ? 0 : (X >= g.W ? g.W - 1 : X))];
    }
};
template <typename T, int NT>
struct MfmaCell {
    int code, tt;            // label path: this lane's pixel label code (>= 0 class, -1 skip, -2 out of range) and max(code, 0)
    float ulraw, ul, mb;     // label path: tap value of class tt at tap (c & 3): raw, and in the exp2 domain
    lossf4 e[NT];            // e[t][r] = exp2(zq) of class 16 t + c at pixel (gq, r)
    float s[4];              // per pixel r: sum over classes of e  (uniform over the 16 lanes of the row group)
    __device__ __forceinline__ void gather(const T* __restrict__ img, const LossGeom& g, const MfmaCellLoads<T, NT>& L, int c,
                                           int64_t skip_label) {
        const int kl = c & 3;
        const T* pl = img + ((int64_t)((kl >> 1) ? L.cc.y1 : L.cc.y0) * g.w + ((kl & 1) ? L.cc.x1 : L.cc.x0)) * g.ldl;
        code = (!L.inside || L.traw == skip_label) ? -1 : ((L.traw < 0 || L.traw >= g.C) ? -2 : (int)L.traw);
        tt = code < 0 ? 0 : code;
        ulraw = ldf<T>(pl + tt);
    }
    __device__ __forceinline__ void finish(const LossGeom& g, const MfmaCellLoads<T, NT>& L, int c, float wA) {
        // lanes past the last class hold a duplicate of class C-1 (clamped load), so they can take part in the maximum as they
        // are; plain v_max_f32 (fmaxf would first canonicalise each operand with a second v_max)
        float mx = L.u[0];
#pragma unroll
        for (int t = 1; t < NT; ++t) asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(L.u[t]));
        mb = wave_max_all(mx);
        // the elementwise work is written on 4-vectors so that it lowers to packed fp32 instructions (v_pk_add/mul/fma_f32:
        // two lanes' worth of flops per issue slot) -- these kernels are bound by VALU issue, not by memory
        lossf4 s4 = {0.f, 0.f, 0.f, 0.f};
        const float nmb = -mb * LS_LOG2E;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float ut = 16 * t + c < g.C ? fmaf(L.u[t], LS_LOG2E, nmb) : -1e30f;
            const lossf4 z = __builtin_amdgcn_mfma_f32_16x16x4f32(wA, ut, lossf4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            e[t] = lossf4{__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1]), __builtin_amdgcn_exp2f(z[2]),
                          __builtin_amdgcn_exp2f(z[3])};
            s4 += e[t];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] = row_sum16(s4[r]);
        ul = (ulraw - mb) * LS_LOG2E;
    }
};

template <typename T, int NT>
__global__ void __launch_bounds__(LS_THREADS) ce_dice_fwd_mfma4_kernel(const T* __restrict__ logits, LossGeom g,
                                                                       const int64_t* __restrict__ target, int64_t ignore_index,
                                                                       const float* __restrict__ cw, float* __restrict__ partial,
                                                                       int* __restrict__ retry) {
    __shared__ float hI[4][NT * 16], hT[4][NT * 16], redP[4][NT * 16], redS[4][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, gq = lane >> 4;
    const int b = blockIdx.y;
    for (int i = lane; i < NT * 16; i += 64) { hI[wave][i] = 0.f; hT[wave][i] = 0.f; }
    // A operand of the interpolation: lane = (pixel p = c -> cell row c >> 2, column c & 3; tap k = gq)
    const float wA = tap_weight(gq, ((c >> 2) + 0.5f) * 0.25f, ((c & 3) + 0.5f) * 0.25f);
    // label path: lane = (tap c & 3; pixel = cell row gq, column c >> 2)
    const float wL = tap_weight(c & 3, (gq + 0.5f) * 0.25f, ((c >> 2) + 0.5f) * 0.25f);
    const int ncx = g.w + 1, ncell = (g.h + 1) * ncx;
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * g.H * g.W;
    lossf4 aP[NT];                                  // per pixel row r separately (packed fma); summed over r at the end
#pragma unroll
    for (int t = 0; t < NT; ++t) aP[t] = lossf4{0.f, 0.f, 0.f, 0.f};
    float cel = 0.f, wsum = 0.f, nvalid = 0.f;      // per lane (quad leaders)
    bool bad = false, slow = false;
    const int stride = gridDim.x * 4;
    int cell = blockIdx.x * 4 + wave;
    MfmaCellLoads<T, NT> nx;
    if (cell < ncell) nx.issue(img, tg, g, cell / ncx - 1, cell % ncx - 1, c, gq);
    while (cell < ncell) {
        const MfmaCellLoads<T, NT> cur = nx;
        MfmaCell<T, NT> m;
        m.gather(img, g, cur, c, ignore_index);
        cell += stride;
        if (cell < ncell) nx.issue(img, tg, g, cell / ncx - 1, cell % ncx - 1, c, gq);
        m.finish(g, cur, c, wA);
        // validity of the pixels of my accumulator rows: pixel (gq, r) is the label-path pixel of lanes 16 gq + 4 r ..
        const unsigned long long okm = __ballot(m.code >= 0);
        const unsigned okrow = (unsigned)(okm >> (16 * gq)) & 0xffffu;
        lossf4 inv;
#pragma unroll
        for (int r = 0; r < 4; ++r) inv[r] = ((okrow >> (4 * r)) & 1u) ? __builtin_amdgcn_rcpf(fmaxf(m.s[r], 1e-30f)) : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) aP[t] += m.e[t] * inv;
        // label path
        float zt = wL * m.ul;
        zt += dpp_mov<DPP_XOR1>(zt); zt += dpp_mov<DPP_XOR2>(zt);
        const float tot = sel4(m.s, c >> 2);
        const bool valid = m.code >= 0, under = valid && !(tot > 1e-30f);
        bad |= m.code == -2; slow |= under;
        if (valid && !under && (c & 3) == 0) {
            const float wt = cw ? cw[m.tt] : 1.f;
            atomicAdd(&hI[wave][m.tt], __builtin_amdgcn_exp2f(zt) * __builtin_amdgcn_rcpf(tot));
            atomicAdd(&hT[wave][m.tt], 1.f);
            cel = fmaf(wt, __builtin_amdgcn_logf(tot) - zt, cel);
            wsum += wt; nvalid += 1.f;
        }
    }
    if (__any(slow) && lane == 0) atomicOr(retry, 1);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float v = (aP[t][0] + aP[t][1]) + (aP[t][2] + aP[t][3]);
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (gq == 0) redP[wave][16 * t + c] = v;
    }
    const float ce = LS_LN2 * wave_sum_all(cel), ws = wave_sum_all(wsum), nv = wave_sum_all(nvalid);
    if (lane == 0) { redS[wave][0] = ce; redS[wave][1] = ws; redS[wave][2] = nv; redS[wave][3] = __any(bad) ? 1.f : 0.f; }
    __syncthreads();
    float* dst = partial + ((int64_t)blockIdx.x * g.B + b) * (3 * g.C + 4);
    for (int i = threadIdx.x; i < NT * 16; i += LS_THREADS) {
        if (i < g.C) {
            dst[i] = (hI[0][i] + hI[1][i]) + (hI[2][i] + hI[3][i]);
            dst[g.C + i] = (redP[0][i] + redP[1][i]) + (redP[2][i] + redP[3][i]);
            dst[2 * g.C + i] = (hT[0][i] + hT[1][i]) + (hT[2][i] + hT[3][i]);
        }
    }
    if (threadIdx.x < 4) dst[3 * g.C + threadIdx.x] = (redS[0][threadIdx.x] + redS[1][threadIdx.x]) + (redS[2][threadIdx.x] + redS[3][threadIdx.x]);
}

// Backward: workgroup = LS_TILE x LS_TILE low-res taps (+ the halo cells), four parity colours as in the VALU kernels.  A wave
// walks its cells of all four colours as one sequence so that the next cell's loads overlap the current cell's arithmetic
// also across the colour barriers.  The scatter MFMA puts tap j in product row 4 j, i.e. in register 0 of row group j: every
// lane then owns one (tap, class) element per class tile and the tile update is one full-wave LDS read-add-write per class
// tile (taps outside the tile go to a scratch row).
template <typename T, int NT>
__global__ void __launch_bounds__(LS_THREADS) ce_dice_bwd_mfma4_kernel(const T* __restrict__ logits, LossGeom g,
                                                                       const int64_t* __restrict__ target, int64_t ignore_index,
                                                                       const float* __restrict__ cw, int dice,
                                                                       const float* __restrict__ stats,
                                                                       const float* __restrict__ grad_out, T* __restrict__ dlow,
                                                                       int64_t ldd, int* __restrict__ retry) {
    constexpr int NTAP = LS_TILE * LS_TILE;
    __shared__ float accum[NTAP + 1][NT * 16];       // row NTAP: scratch
    __shared__ float gIs[NT * 16];
    __shared__ __attribute__((aligned(16))) float c1s[4][16], cks[4][16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, gq = lane >> 4;
    const int b = blockIdx.y;
    const int tiles_x = (g.w + LS_TILE - 1) / LS_TILE;
    const int ty0 = (blockIdx.x / tiles_x) * LS_TILE, tx0 = (blockIdx.x % tiles_x) * LS_TILE;
    for (int i = threadIdx.x; i < (NTAP + 1) * NT * 16; i += LS_THREADS) (&accum[0][0])[i] = 0.f;
    float gP[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float gi;
        dice_coef_one(stats, b, g.B, g.C, dice, 16 * t + c, gi, gP[t]);
        if (wave == 0 && gq == 0) gIs[16 * t + c] = gi;
    }
    const float wA = tap_weight(gq, ((c >> 2) + 0.5f) * 0.25f, ((c & 3) + 0.5f) * 0.25f);
    const float wL = tap_weight(c & 3, (gq + 0.5f) * 0.25f, ((c >> 2) + 0.5f) * 0.25f);
    // A operand of the scatter for accumulator register r: lane = (product row c, k = gq); row 4 j carries tap j, the weight
    // is the one of tap j at pixel (gq, r)
    // At the clamped image border two taps of a cell coincide: the first of the pair then takes the whole weight and the
    // second is parked on the scratch row (two row groups must never update one LDS word in the same instruction).
    const int kB = c >> 2;
    const float wBy = (c & 3) == 0 ? ((kB >> 1) ? (gq + 0.5f) * 0.25f : 1.f - (gq + 0.5f) * 0.25f) : 0.f;
    const float wByc = (c & 3) == 0 ? ((kB >> 1) ? 0.f : 1.f) : 0.f;
    float wBx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) wBx[r] = (kB & 1) ? (r + 0.5f) * 0.25f : 1.f - (r + 0.5f) * 0.25f;
    const float wBxc = (kB & 1) ? 0.f : 1.f;
    const float go = grad_out ? grad_out[0] : 1.f;
    const float invW = 1.f / stats[(int64_t)g.B * (3 * g.C + 4) + 1];
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * g.H * g.W;
    __syncthreads();
    constexpr int ncl = LS_TILE + 1;
    bool slow = false;
    // the wave's cell sequence: colour col, i-th cell of that colour for this wave
    int col = 0, idx = wave;
    auto cell_of = [&](int col_, int idx_, int& cj, int& ck) {
        const int pj = col_ >> 1, pk = col_ & 1;
        const int nk = (ncl - pk + 1) / 2;
        cj = ty0 - 1 + 2 * (idx_ / nk) + pj; ck = tx0 - 1 + 2 * (idx_ % nk) + pk;
    };
    auto count_of = [&](int col_) { return ((ncl - (col_ >> 1) + 1) / 2) * ((ncl - (col_ & 1) + 1) / 2); };
    MfmaCellLoads<T, NT> nx;
    int cj, ck;
    cell_of(col, idx, cj, ck);
    nx.issue(img, tg, g, cj > g.h - 1 ? g.h - 1 : cj, ck > g.w - 1 ? g.w - 1 : ck, c, gq);
    bool nx_ok = cj <= g.h - 1 && ck <= g.w - 1;
    while (col < 4) {
        const MfmaCellLoads<T, NT> cur = nx;
        const bool cur_ok = nx_ok;
        MfmaCell<T, NT> m;
        m.gather(img, g, cur, c, ignore_index);
        int ncol = col, nidx = idx + 4;
        if (nidx >= count_of(col)) { ++ncol; nidx = wave; }
        if (ncol < 4) {
            cell_of(ncol, nidx, cj, ck);
            nx_ok = cj <= g.h - 1 && ck <= g.w - 1;
            nx.issue(img, tg, g, cj > g.h - 1 ? g.h - 1 : cj, ck > g.w - 1 ? g.w - 1 : ck, c, gq);
        }
        if (cur_ok) {
            const Cell& cc = cur.cc;
            m.finish(g, cur, c, wA);
            lossf4 dp4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) dp4 += gP[t] * m.e[t];
            float dp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[r] = row_sum16(dp4[r]);
            // label path: per-pixel coefficients, the [c == t] part of the gradient, and c1 / c1*k for the class lanes
            float zt = wL * m.ul;
            zt += dpp_mov<DPP_XOR1>(zt); zt += dpp_mov<DPP_XOR2>(zt);
            const float tot = sel4(m.s, c >> 2), dpm = sel4(dp, c >> 2);
            const bool valid = m.code >= 0, under = valid && !(tot > 1e-30f);
            slow |= under;
            const float okf = (valid && !under) ? 1.f : 0.f;
            const float inv = okf * __builtin_amdgcn_rcpf(fmaxf(tot, 1e-30f));
            const float wce = okf * (cw ? cw[m.tt] : 1.f) * invW;
            const float et = __builtin_amdgcn_exp2f(zt), git = gIs[m.tt];
            const float c1 = go * inv;
            const float k = wce - (dpm + git * et) * inv;                  // wce - <G, p>
            const float dlt = okf * (c1 * et * git - go * wce);            // extra d loss / d z of the label class
            if ((c & 3) == 0) { c1s[wave][4 * gq + (c >> 2)] = c1; cks[wave][4 * gq + (c >> 2)] = c1 * k; }
            __builtin_amdgcn_wave_barrier();
            const float4 c1r = *reinterpret_cast<const float4*>(&c1s[wave][4 * gq]);
            const float4 ckr = *reinterpret_cast<const float4*>(&cks[wave][4 * gq]);
            __builtin_amdgcn_wave_barrier();
            const lossf4 c1v = {c1r.x, c1r.y, c1r.z, c1r.w}, ckv = {ckr.x, ckr.y, ckr.z, ckr.w};
            // tile row of a tap of this cell (taps outside the tile are recomputed by the neighbouring workgroups -> scratch row)
            const bool ycol = cc.y0 == cc.y1, xcol = cc.x0 == cc.x1;
            auto tap_row = [&](int kk) {
                const int ly = ((kk >> 1) ? cc.y1 : cc.y0) - ty0, lx = ((kk & 1) ? cc.x1 : cc.x0) - tx0;
                return (ly >= 0 && ly < LS_TILE && lx >= 0 && lx < LS_TILE) ? ly * LS_TILE + lx : NTAP;
            };
            const bool dup = ((gq >> 1) && ycol) || ((gq & 1) && xcol);
            float* arow = &accum[dup ? NTAP : tap_row(gq)][c];
            const float wy = ycol ? wByc : wBy;
            float wB[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wB[r] = wy * (xcol ? wBxc : wBx[r]);
            float upd[NT];
            if constexpr (sizeof(T) == 2) {
                // bf16 activations: the gradient is rounded to bf16 on store anyway, so the scatter runs as ONE
                // v_mfma_f32_16x16x32_bf16 per class tile (k-slot 8 gq + j = pixel (gq, j), j < 4; slots 4..7 empty) instead of
                // four f32 instructions; the weights (odd multiples of 1/64) are exact in bf16
                union { uint32_t u[4]; lossbf8 v; } A, Bv;
                A.u[0] = pack2bf(wB[0], wB[1]); A.u[1] = pack2bf(wB[2], wB[3]); A.u[2] = 0u; A.u[3] = 0u;
                Bv.u[2] = 0u; Bv.u[3] = 0u;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const lossf4 dz = m.e[t] * (c1v * gP[t] + ckv);          // go * p * (gP - <G,p> + wce)
                    Bv.u[0] = pack2bf(dz[0], dz[1]); Bv.u[1] = pack2bf(dz[2], dz[3]);
                    const lossf4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.v, Bv.v, lossf4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    upd[t] = acc[0];
                }
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    lossf4 acc = {0.f, 0.f, 0.f, 0.f};
                    const lossf4 dz = m.e[t] * (c1v * gP[t] + ckv);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[r], dz[r], acc, 0, 0, 0);
                    upd[t] = acc[0];          // product row 4 gq = tap gq, class 16 t + c
                }
            }
            float old[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) old[t] = arow[16 * t];
#pragma unroll
            for (int t = 0; t < NT; ++t) arow[16 * t] = old[t] + upd[t];
            // label class: every lane adds its tap's share for its pixel (same-wave LDS operations execute in order)
            atomicAdd(&accum[tap_row(c & 3)][m.tt], wL * dlt);
        }
        if (ncol != col) __syncthreads();
        col = ncol; idx = nidx;
    }
    if (__any(slow) && lane == 0) atomicOr(retry, 1);
    for (int tap = wave; tap < NTAP; tap += 4) {
        const int y = ty0 + tap / LS_TILE, x = tx0 + tap % LS_TILE;
        if (y >= g.h || x >= g.w) continue;
        T* drow = dlow + (((int64_t)b * g.h + y) * g.w + x) * ldd;
        for (int cls = lane; cls < ldd; cls += 64) stf<T>(drow + cls, cls < g.C ? accum[tap][cls] : 0.f);
    }
}

// generic ratio: full-resolution gradient, one wave per pixel at a time; segf_bilinear_bwd follows
template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) ce_dice_bwd_kernel(const T* __restrict__ logits, LossGeom g,
                                                                  const int64_t* __restrict__ target, int64_t ignore_index,
                                                                  const float* __restrict__ cw, int dice,
                                                                  const float* __restrict__ stats, const float* __restrict__ grad_out,
                                                                  T* __restrict__ dfull, int64_t ldg) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    float gI[NS], gP[NS];
    dice_coefs<NS>(stats, b, g.B, g.C, dice, lane, gI, gP);
    const float go = grad_out ? grad_out[0] : 1.f;
    const float invW = 1.f / stats[(int64_t)g.B * (3 * g.C + 4) + 1];
    const int64_t npix = (int64_t)g.H * g.W;
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t per = (npix + nw - 1) / nw;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p0 = wid * per, p1 = p0 + per < npix ? p0 + per : npix;
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * npix;
    T* dimg = dfull + (int64_t)b * npix * ldg;
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t t = tg[p];
        T* drow = dimg + p * ldg;
        float z[NS], dz[NS];
        if (t == ignore_index || t < 0 || t >= g.C) {
#pragma unroll
            for (int s = 0; s < NS; ++s) dz[s] = 0.f;
        } else {
            const int Y = (int)(p / g.W), X = (int)(p - (int64_t)Y * g.W);
            pixel_logits<T, NS>(img, g, Y, X, lane, z);
            pixel_grad<NS>(z, (int)t, lane, gI, gP, (cw ? cw[t] : 1.f) * invW, go, dz);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = lane + 64 * s;
            if (c < ldg) stf<T>(drow + c, c < g.C ? dz[s] : 0.f);
        }
    }
}

// test hook: out[i] = sum over the 64 rows of in[row][i], through the transposing wave reduction used by the loss kernels
__global__ void wave_reduce16_test_kernel(const float* __restrict__ in, float* __restrict__ out) {
    float v[16], o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i];
    wave_reduce16(v, o);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i] = o[i];
    }
}
extern "C" int segf_debug_wave_reduce16(const float* in, float* out, void* stream) {
    hipLaunchKernelGGL(wave_reduce16_test_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    SEGF_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t segf_ce_dice_stats_floats(int B, int C) {
    return (int64_t)B * (3 * C + 4) + 4 + (int64_t)LS_NBLK * B * (3 * C + 4) + 4;      // + retry flags
}

#define LS_NS_DISPATCH(ns, CALL) do { if ((ns) == 1) { CALL(1); } else if ((ns) == 2) { CALL(2); } else { CALL(3); } } while (0)

__global__ void zero_words_kernel(uint32_t* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
__global__ void zero_ints_kernel(int* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0;
}

#define LS_NT_DISPATCH(C, CALL) do { const int nt_ = ((C) + 15) / 16;                                                             \
        if (nt_ <= 2) { CALL(2); } else if (nt_ <= 4) { CALL(4); } else if (nt_ <= 10) { CALL(10); } else { CALL(12); } } while (0)
static bool loss_use_mfma(int sc) { return sc == 4 && !POL(loss_no_mfma); }

template <typename T>
static void fwd_launch(int ns, int sc, dim3 grid, hipStream_t st, const T* logits, LossGeom g, const int64_t* target,
                       int64_t ignore_index, const float* cw, float* partial, int* retry, float* lse) {
    bool band = false;
    if constexpr (sizeof(T) == 2) {
        if (loss_use_mfma(sc)) band = loss_band_fwd_launch((const bf16_t*)logits, g, target, ignore_index, cw, partial, retry, lse, st);
    }
    if (band) {}
    else if (loss_use_mfma(sc)) {
#define CALLM(NT) hipLaunchKernelGGL((ce_dice_fwd_mfma4_kernel<T, NT>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index, cw, partial, retry)
        LS_NT_DISPATCH(g.C, CALLM);
#undef CALLM
    }
#define CALL(NS)                                                                                                                \
    do {                                                                                                                        \
        if (sc >= 2) {                                                                                                          \
            if (loss_use_mfma(sc)) {}                                                                                           \
            else if (sc == 4) hipLaunchKernelGGL((ce_dice_fwd_cells16_kernel<T, NS, 4>), grid, dim3(LS_THREADS), 0, st, logits, g, \
                                            target, ignore_index, cw, partial, retry);                                          \
            else if (sc == 2) hipLaunchKernelGGL((ce_dice_fwd_cells16_kernel<T, NS, 2>), grid, dim3(LS_THREADS), 0, st, logits,  \
                                                 g, target, ignore_index, cw, partial, retry);                                  \
            else hipLaunchKernelGGL((ce_dice_fwd_cells16_kernel<T, NS, 8>), grid, dim3(LS_THREADS), 0, st, logits, g, target,    \
                                    ignore_index, cw, partial, retry);                                                          \
            if (!POL(loss_no_retry))                                                                                \
            hipLaunchKernelGGL((ce_dice_fwd_cells_kernel<T, NS>), grid, dim3(LS_THREADS), 0, st, logits, g, sc, target,         \
                               ignore_index, cw, partial, (const int*)retry);                                                   \
        } else if (sc) hipLaunchKernelGGL((ce_dice_fwd_cells_kernel<T, NS>), grid, dim3(LS_THREADS), 0, st, logits, g, sc,       \
                                          target, ignore_index, cw, partial, (const int*)nullptr);                              \
        else hipLaunchKernelGGL((ce_dice_fwd_kernel<T, NS>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index,    \
                                cw, partial);                                                                                   \
    } while (0)
    LS_NS_DISPATCH(ns, CALL);
#undef CALL
}
template <typename T>
static void bwd_cells_launch(int ns, int sc, dim3 grid, hipStream_t st, const T* logits, LossGeom g, const int64_t* target,
                             int64_t ignore_index, const float* cw, int dice, const float* stats, const float* grad_out,
                             T* dlow, int64_t ldd, int* retry, const float* lse) {
    // with the forward's per-pixel log-sums in use, retry[1] (= "they are not usable", set by the forward) must survive
    if (sc >= 2) hipLaunchKernelGGL(zero_ints_kernel, dim3(1), dim3(64), 0, st, retry, lse ? 1 : 4);
    bool band = false;
    if constexpr (sizeof(T) == 2) {
        if (loss_use_mfma(sc))
            band = loss_band_bwd_launch((const bf16_t*)logits, g, target, ignore_index, cw, dice, stats, grad_out, (bf16_t*)dlow, ldd, retry, lse, st);
    }
    if (band) {}
    else if (loss_use_mfma(sc)) {
#define CALLM(NT) hipLaunchKernelGGL((ce_dice_bwd_mfma4_kernel<T, NT>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index, cw, dice, stats, grad_out, dlow, ldd, retry)
        LS_NT_DISPATCH(g.C, CALLM);
#undef CALLM
    }
#define CALL(NS)                                                                                                                \
    do {                                                                                                                        \
        if (sc >= 2) {                                                                                                          \
            if (loss_use_mfma(sc)) {}                                                                                           \
            else if (sc == 4) hipLaunchKernelGGL((ce_dice_bwd_cells8_kernel<T, NS, 4>), grid, dim3(LS_THREADS), 0, st, logits, g, \
                                            target, ignore_index, cw, dice, stats, grad_out, dlow, ldd, retry);                 \
            else if (sc == 2) hipLaunchKernelGGL((ce_dice_bwd_cells8_kernel<T, NS, 2>), grid, dim3(LS_THREADS), 0, st, logits,   \
                                                 g, target, ignore_index, cw, dice, stats, grad_out, dlow, ldd, retry);         \
            else hipLaunchKernelGGL((ce_dice_bwd_cells8_kernel<T, NS, 8>), grid, dim3(LS_THREADS), 0, st, logits, g, target,     \
                                    ignore_index, cw, dice, stats, grad_out, dlow, ldd, retry);                                 \
        }                                                                                                                       \
        hipLaunchKernelGGL((ce_dice_bwd_cells_kernel<T, NS>), grid, dim3(LS_THREADS), 0, st, logits, g, sc, target,             \
                           ignore_index, cw, dice, stats, grad_out, dlow, ldd, sc >= 2 ? (const int*)retry : (const int*)nullptr); \
    } while (0)
    LS_NS_DISPATCH(ns, CALL);
#undef CALL
}
template <typename T>
static void bwd_generic_launch(int ns, dim3 grid, hipStream_t st, const T* logits, LossGeom g, const int64_t* target,
                               int64_t ignore_index, const float* cw, int dice, const float* stats, const float* grad_out,
                               T* dfull, int64_t ldg) {
#define CALL(NS) hipLaunchKernelGGL((ce_dice_bwd_kernel<T, NS>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ignore_index, cw, dice, stats, grad_out, dfull, ldg)
    LS_NS_DISPATCH(ns, CALL);
#undef CALL
}

// floats of the per-pixel log-sum buffer the forward can leave for the backward (0: this configuration does not produce one)
extern "C" int64_t segf_ce_dice_lse_floats(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl) {
    if (dt != SEGF_BF16 || B <= 0 || C <= 0 || h <= 0 || w <= 0 || POL(loss_no_lse)) return 0;
    if (!loss_use_mfma(pow2_scale(h, w, H, W))) return 0;
    LossGeom g{B, C, h, w, H, W, ldl};
    if (!loss_band_fwd_covers((const bf16_t*)logits, g)) return 0;
    return (int64_t)B * (h + 1) * (w + 1) * 16;
}

extern "C" int segf_ce_dice_fwd(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl,
                                const int64_t* target, int64_t ignore_index, const float* class_weight, int dice, float* stats,
                                float* loss, float* pix_lse, void* stream) {
    if (B <= 0 || C <= 0 || C > 192 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || ldl < C || B > 65535) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    LossGeom g{B, C, h, w, H, W, ldl};
    if (pix_lse && !segf_ce_dice_lse_floats(dt, B, C, h, w, H, W, logits, ldl)) return SEGF_ERR_SHAPE;
    float* partial = stats + (int64_t)B * (3 * C + 4) + 4;
    const int ns = (C + 63) / 64;
    const int sc = pow2_scale(h, w, H, W);
    const dim3 grid(LS_NBLK, B);
    int* retry = reinterpret_cast<int*>(partial + (int64_t)LS_NBLK * B * (3 * C + 4));
    if (sc >= 2) {      // a kernel node rather than hipMemsetAsync: inside a captured graph the 16-byte memset node did not take effect
        hipLaunchKernelGGL(zero_ints_kernel, dim3(1), dim3(64), 0, st, retry, 4);
        SEGF_CHECK_LAUNCH();
    }
    SEGF_DISPATCH_DT(dt, T, { fwd_launch<T>(ns, sc, grid, st, (const T*)logits, g, target, ignore_index, class_weight, partial, retry, pix_lse); })
    SEGF_CHECK_LAUNCH();
    colreduce_finalize_launch(partial, LS_NBLK, (int64_t)B * (3 * C + 4), stats, st);
    SEGF_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_dice_loss_kernel, dim3(1), dim3(256), 0, st, stats, B, C, dice, loss);
    SEGF_CHECK_LAUNCH();
    return 0;
}

// workspace (floats) of segf_ce_dice_bwd: 0 on the fused path, the full-resolution gradient on the generic path
extern "C" int64_t segf_ce_dice_bwd_ws(int dt, int B, int C, int h, int w, int H, int W) {
    if (pow2_scale(h, w, H, W)) return 0;
    const int64_t ldg = (C + 7) / 8 * 8;
    const int64_t bytes = (int64_t)B * H * W * ldg * (dt == SEGF_BF16 ? 2 : 4);
    return (bytes + 3) / 4;
}

extern "C" int segf_bilinear_bwd(int dt, int B, int h, int w, int C, void* din, int64_t ldi, int H, int W, const void* dout,
                                 int64_t ldo, int align_corners, void* stream);

extern "C" int segf_ce_dice_bwd(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl,
                                const int64_t* target, int64_t ignore_index, const float* class_weight, int dice,
                                const float* stats, const float* grad_out, void* dlogits, int64_t ldd, float* ws,
                                const float* pix_lse, void* stream) {
    if (B <= 0 || C <= 0 || C > 192 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || ldl < C || ldd < C || B > 65535) return SEGF_ERR_SHAPE;
    if (pix_lse && !segf_ce_dice_lse_floats(dt, B, C, h, w, H, W, logits, ldl)) return SEGF_ERR_SHAPE;
    if (ldd > 64 * ((C + 63) / 64)) return SEGF_ERR_SHAPE;     // pad columns are zero-filled by the class lanes
    hipStream_t st = (hipStream_t)stream;
    LossGeom g{B, C, h, w, H, W, ldl};
    const int ns = (C + 63) / 64;
    const int sc = pow2_scale(h, w, H, W);
    if (sc) {
        const dim3 grid(((h + LS_TILE - 1) / LS_TILE) * ((w + LS_TILE - 1) / LS_TILE), B);
        // the retry word shares the forward's flag slot at the end of the stats buffer (the forward has long consumed it)
        int* retry = reinterpret_cast<int*>(const_cast<float*>(stats) + segf_ce_dice_stats_floats(B, C) - 4);
        SEGF_DISPATCH_DT(dt, T, { bwd_cells_launch<T>(ns, sc, grid, st, (const T*)logits, g, target, ignore_index, class_weight, dice, stats, grad_out, (T*)dlogits, ldd, retry, pix_lse); })
        SEGF_CHECK_LAUNCH();
        return 0;
    }
    if (!ws) return SEGF_ERR_WORKSPACE;
    const int64_t ldg = (C + 7) / 8 * 8;
    SEGF_DISPATCH_DT(dt, T, { bwd_generic_launch<T>(ns, dim3(LS_NBLK, B), st, (const T*)logits, g, target, ignore_index, class_weight, dice, stats, grad_out, (T*)ws, ldg); })
    SEGF_CHECK_LAUNCH();
    if (ldd > C) {      // a kernel node, not hipMemsetAsync: a captured memset node did not take effect in graph replay (see segf_ce_dice_fwd)
        const int64_t nwords = ((int64_t)B * h * w * ldd * (dt == SEGF_BF16 ? 2 : 4) + 3) / 4;
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)imin64(cdiv64(nwords, 256), 4096)), dim3(256), 0, st, (uint32_t*)dlogits, nwords);
        SEGF_CHECK_LAUNCH();
    }
    return segf_bilinear_bwd(dt, B, h, w, C, dlogits, ldd, H, W, ws, ldg, 0, stream);
}

// ---- fused upsample + argmax + confusion matrix ----------------------------------------------------------------
// prediction = lowest class index attaining the maximum (torch.argmax); indices < 192 are exact in fp32
template <int NS>
__device__ __forceinline__ int wave_argmax(const float (&z)[NS], int lane, int C) {
    float mx = z[0];
#pragma unroll
    for (int s = 1; s < NS; ++s) mx = fmaxf(mx, z[s]);
    mx = wave_max_all(mx);
    float best = 1e9f;
#pragma unroll
    for (int s = NS - 1; s >= 0; --s) if (z[s] == mx) best = (float)(lane + 64 * s);
    best = wave_min_all(best);
    const int bi = (int)best;
    return bi < C ? bi : 0;        // all-NaN row
}
__device__ __forceinline__ void confmat_count(int64_t t, int pred, int C, int64_t ignore_label, unsigned long long* mat,
                                              unsigned long long* hist, int* flag) {
    const bool in_mat = t >= 0 && t < C;
    if (in_mat) atomicAdd(mat + t * C + pred, 1ull);
    if (t != ignore_label) {
        if (in_mat) atomicAdd(hist + t * C + pred, 1ull);
        else atomicOr(flag, 1);    // label >= n that is not ignore_label: the reference's bincount shape check fails
    }
}

template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) argmax_confmat_cells_kernel(const T* __restrict__ logits, LossGeom g, int sc,
                                                                           const int64_t* __restrict__ target, int64_t ignore_label,
                                                                           unsigned long long* __restrict__ mat,
                                                                           unsigned long long* __restrict__ hist, int* __restrict__ flag,
                                                                           int64_t* __restrict__ pred_out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int halo = sc > 1 ? 1 : 0, off = sc >> 1;
    const float inv_sc = 1.f / (float)sc, fhalo = (float)halo;
    const int ncx = g.w + halo, ncell = (g.h + halo) * ncx;
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t npix = (int64_t)g.H * g.W;
    const int64_t* tg = target + (int64_t)b * npix;
    for (int cell = blockIdx.x * 4 + wave; cell < ncell; cell += gridDim.x * 4) {
        const Cell c = make_cell(cell / ncx - halo, cell % ncx - halo, g.h, g.w, halo);
        float t00[NS], t01[NS], t10[NS], t11[NS];
        // code -3 = inside the image but counted nowhere (label == ignore_label and outside [0, C)); the raw label
        // stays in its lane for the counting step
        int64_t raw = -1;
        load_cell_taps<T, NS>(img, g, c, lane, t00, t01, t10, t11);
        int codes = cell_label_codes(tg, g, sc, off, c.cj, c.ck, lane, INT64_MIN, &raw);
        if (codes == -2 && raw == ignore_label) codes = -3;
        int preds = 0;
        // Tap weights as ATen's compute_source_index_and_lambda produces them: inside the image lambda1 = (a + 0.5) / sc (exact
        // for the power-of-two ratios of this path); in the clamped top / left border cells (cj < 0 / ck < 0) the source index is
        // clamped to 0, lambda1 = 0 and the first tap carries the whole weight; in the bottom / right border cells both taps are
        // the last row / column and keep their fractional weights.  Value = bilinear_aten (ATen's operation order).
        for (int a = 0; a < sc; ++a) {
            const int Y = sc * c.cj + off + a;
            if (Y < 0 || Y >= g.H) continue;
            const float ly = c.cj < 0 ? 0.f : (a + 0.5f) * inv_sc * fhalo;
            for (int bb = 0; bb < sc; ++bb) {
                const int t = __builtin_amdgcn_readlane(codes, a * sc + bb);     // wave-uniform
                if (t == -1 || (t == -3 && !pred_out)) continue;
                const int X = sc * c.ck + off + bb;
                const float lx = c.ck < 0 ? 0.f : (bb + 0.5f) * inv_sc * fhalo;
                float z[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    z[s] = (lane + 64 * s) < g.C ? bilinear_aten(t00[s], t01[s], t10[s], t11[s], ly, lx) : -INFINITY;
                const int best = wave_argmax<NS>(z, lane, g.C);
                if (lane == a * sc + bb) preds = best;
            }
        }
        // every pixel's lane counts its own (label, prediction) pair: sc*sc atomics issued together
        if (codes != -1 && (codes != -3 || pred_out)) {
            const int a = lane / sc, bb = lane - a * sc;
            const int Y = sc * c.cj + off + a, X = sc * c.ck + off + bb;
            if (pred_out) pred_out[(int64_t)b * npix + (int64_t)Y * g.W + X] = preds;
            if (codes != -3) confmat_count(raw, preds, g.C, ignore_label, mat, hist, flag);
        }
    }
}

template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) argmax_confmat_kernel(const T* __restrict__ logits, LossGeom g,
                                                                     const int64_t* __restrict__ target, int64_t ignore_label,
                                                                     unsigned long long* __restrict__ mat,
                                                                     unsigned long long* __restrict__ hist, int* __restrict__ flag,
                                                                     int64_t* __restrict__ pred_out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int64_t npix = (int64_t)g.H * g.W;
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t per = (npix + nw - 1) / nw;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p0 = wid * per, p1 = p0 + per < npix ? p0 + per : npix;
    const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * npix;
    for (int64_t p = p0; p < p1; ++p) {
        const int64_t t = tg[p];
        if (!pred_out && !(t >= 0 && t < g.C) && t == ignore_label) continue;
        const int Y = (int)(p / g.W), X = (int)(p - (int64_t)Y * g.W);
        float z[NS];
        pixel_logits<T, NS, true>(img, g, Y, X, lane, z);
        const int best = wave_argmax<NS>(z, lane, g.C);
        if (lane == 0) {
            if (pred_out) pred_out[(int64_t)b * npix + p] = best;
            confmat_count(t, best, g.C, ignore_label, mat, hist, flag);
        }
    }
}

// LANE = PIXEL form for ratios 4 and 8 (every head of this library but FPNHead predicts at stride 4).  The cell kernel above puts the
// classes on the lanes and pays two 6-step wave reductions per full-resolution pixel (maximum, then lowest index attaining it), and BOTH
// kernels above count with two 8-byte global atomics per pixel into a [C][C] matrix -- 64 lanes, 64 unrelated addresses per instruction,
// the slowest shape an atomic can have (MI355X_MICROARCH.md, "Global float atomics": ~0.08 TB/s): 1.75 ms for a 32-image batch of which
// the arithmetic is a fifth.  Here
//   * a wave owns 64 pixels (sc = 4: four cells), stages the cells' four tap rows in LDS as fp32 [cell][class][4 taps] (one 16-byte
//     broadcast read per class and lane), and every lane walks the classes of ITS pixel with a running (best, index): the four weight
//     products of bilinear_aten once per pixel, then 1 mul + 3 fma + compare + 2 selects per class, no cross-lane traffic.  Same value per
//     (pixel, class) as the cell kernel (the same bilinear_aten on the same fp32 taps), lowest index on ties (strict >);
//   * the counts go into a per-workgroup LDS matrix (uint32 [C][C], C <= 152: 92 KB; one persistent 384-thread workgroup per CU) and
//     leave it once, at the end, as COALESCED 8-byte atomics of the non-zero entries.  `hist` differs from `mat` only in the row of
//     an ignore_label that lies inside [0, C) (counted in mat, not in hist), so one LDS matrix serves both.
#define AMX_CT 152
#define AMX_WAVES 6
template <typename T, int SC>
__global__ void __launch_bounds__(64 * AMX_WAVES) argmax_confmat_pix_kernel(const T* __restrict__ logits, LossGeom g,
                                                                             const int64_t* __restrict__ target, int64_t ignore_label,
                                                                             unsigned long long* __restrict__ mat,
                                                                             unsigned long long* __restrict__ hist, int* __restrict__ flag,
                                                                             int64_t* __restrict__ pred_out) {
    constexpr int PP = SC * SC, CPW = 64 / PP, NSM = (AMX_CT + 63) / 64;          // pixels per cell, cells per wave, class slots per lane
    __shared__ __attribute__((aligned(16))) float taps[AMX_WAVES][CPW][AMX_CT][4];          // [wave][cell][class][t00, t01, t10, t11]
    __shared__ unsigned cnt[AMX_CT * AMX_CT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nn = g.C * g.C;
    for (int i = threadIdx.x; i < nn; i += 64 * AMX_WAVES) cnt[i] = 0u;
    __syncthreads();
    constexpr int off = SC >> 1;
    constexpr float inv_sc = 1.f / (float)SC;
    const int ncx = g.w + 1, ncell = (g.h + 1) * ncx;
    const int cpi = (ncell + CPW - 1) / CPW;                        // chunks (of CPW cells) per image
    const int64_t nchunk = (int64_t)g.B * cpi;
    const int64_t npix = (int64_t)g.H * g.W;
    const int ci = lane / PP, pl = lane - ci * PP, a = pl / SC, bb = pl - a * SC;
    float (*tw)[AMX_CT][4] = taps[wave];
    bool bad = false;
    // the tap rows of the NEXT chunk travel in registers while the current chunk is evaluated (a wave's chunk is otherwise a serial
    // chain: global loads -> LDS -> 150 dependent LDS reads, with 1.5 waves per SIMD to cover it)
    float4 R[CPW][NSM];
    auto fetch = [&](int64_t chunk) {
        const int64_t cq = chunk < nchunk ? chunk : nchunk - 1;
        const int b = (int)(cq / cpi);
        const int cell0 = (int)(cq - (int64_t)b * cpi) * CPW;
        const T* img = logits + (int64_t)b * g.h * g.w * g.ldl;
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
            const int cell = cell0 + k < ncell ? cell0 + k : ncell - 1;
            const Cell c = make_cell(cell / ncx - 1, cell % ncx - 1, g.h, g.w, 1);
            const T* r00 = img + ((int64_t)c.y0 * g.w + c.x0) * g.ldl;
            const T* r01 = img + ((int64_t)c.y0 * g.w + c.x1) * g.ldl;
            const T* r10 = img + ((int64_t)c.y1 * g.w + c.x0) * g.ldl;
            const T* r11 = img + ((int64_t)c.y1 * g.w + c.x1) * g.ldl;
#pragma unroll
            for (int s = 0; s < NSM; ++s) {
                const int cc = lane + 64 * s < g.C ? lane + 64 * s : g.C - 1;          // clamped: every load unconditional
                R[k][s] = make_float4(ldf<T>(r00 + cc), ldf<T>(r01 + cc), ldf<T>(r10 + cc), ldf<T>(r11 + cc));
            }
        }
    };
    const int64_t stride = (int64_t)gridDim.x * AMX_WAVES;
    int64_t chunk = (int64_t)blockIdx.x * AMX_WAVES + wave;
    if (chunk < nchunk) fetch(chunk);
    for (; chunk < nchunk; chunk += stride) {
        const int b = (int)(chunk / cpi);
        const int cell0 = (int)(chunk - (int64_t)b * cpi) * CPW;
        const int64_t* tg = target + (int64_t)b * npix;
#pragma unroll
        for (int k = 0; k < CPW; ++k)
#pragma unroll
            for (int s = 0; s < NSM; ++s)
                if (lane + 64 * s < g.C) *reinterpret_cast<float4*>(tw[k][lane + 64 * s]) = R[k][s];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        fetch(chunk + stride);                                       // in flight under this chunk's class walk
        // ---- this lane's pixel
        const int cell = cell0 + ci;
        const int cj = cell / ncx - 1, ck = cell - (cell / ncx) * ncx - 1;
        const int Y = SC * cj + off + a, X = SC * ck + off + bb;
        const bool inside = cell < ncell && Y >= 0 && Y < g.H && X >= 0 && X < g.W;
        int64_t t = ignore_label;
        if (inside) t = tg[(int64_t)Y * g.W + X];
        const bool in_mat = t >= 0 && t < g.C;
        const bool need = inside && (pred_out != nullptr || in_mat);       // counted, or the prediction is wanted
        if (inside && !in_mat && t != ignore_label) bad = true;            // label >= n that is not ignore_label (flagged, counted nowhere)
        // ATen's weights: see the cell kernel (clamped top / left border cells carry the whole weight on the first tap)
        const float ly = cj < 0 ? 0.f : ((float)a + 0.5f) * inv_sc, lx = ck < 0 ? 0.f : ((float)bb + 0.5f) * inv_sc;
        const float wy0 = 1.f - ly, wx0 = 1.f - lx;
        const float w00 = __fmul_rn(wy0, wx0), w01 = __fmul_rn(wy0, lx), w10 = __fmul_rn(ly, wx0), w11 = __fmul_rn(ly, lx);
        float best = -INFINITY;
        int idx = 0;
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
            const float (*tc)[4] = tw[ci];
#pragma unroll 8
            for (int cc = 0; cc < g.C; ++cc) {
                const float4 tp = *reinterpret_cast<const float4*>(tc[cc]);
                const float v = __fmaf_rn(tp.w, w11, __fmaf_rn(tp.z, w10, __fmaf_rn(tp.x, w00, __fmul_rn(tp.y, w01))));
                const bool gt = v > best;
                best = gt ? v : best;
                idx = gt ? cc : idx;
            }
        }
        if (need) {
            if (pred_out) pred_out[(int64_t)b * npix + (int64_t)Y * g.W + X] = idx;
            if (in_mat) atomicAdd(&cnt[(int)t * g.C + idx], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the next iteration overwrites the tap rows
        __builtin_amdgcn_wave_barrier();
    }
    if (bad) atomicOr(flag, 1);
    __syncthreads();
    const int ign_row = (ignore_label >= 0 && ignore_label < g.C) ? (int)ignore_label : -1;
    for (int i = threadIdx.x; i < nn; i += 64 * AMX_WAVES) {
        const unsigned v = cnt[i];
        if (v) {
            atomicAdd(mat + i, (unsigned long long)v);
            if (i / g.C != ign_row) atomicAdd(hist + i, (unsigned long long)v);
        }
    }
}

template <typename T>
static void argmax_launch(int ns, int sc, dim3 grid, hipStream_t st, const T* logits, LossGeom g, const int64_t* target,
                          int64_t ign, unsigned long long* mat, unsigned long long* hist, int* flag, int64_t* pred_out) {
    if ((sc == 4 || sc == 8) && g.C <= AMX_CT) {
        // persistent workgroups (one per CU at most: 150 KB of LDS each), fewer when the batch is small so that the final flush of the
        // per-workgroup matrices (C^2 entries each) does not outweigh the counting
        const int cpw = 64 / (sc * sc);
        const int64_t nchunk = (int64_t)g.B * (((int64_t)(g.h + 1) * (g.w + 1) + cpw - 1) / cpw);
        int64_t nwg = nchunk / (AMX_WAVES * 8);
        nwg = nwg < 16 ? 16 : (nwg > 256 ? 256 : nwg);
        if (sc == 4) hipLaunchKernelGGL((argmax_confmat_pix_kernel<T, 4>), dim3((unsigned)nwg), dim3(64 * AMX_WAVES), 0, st, logits, g, target, ign, mat,
                                        hist, flag, pred_out);
        else hipLaunchKernelGGL((argmax_confmat_pix_kernel<T, 8>), dim3((unsigned)nwg), dim3(64 * AMX_WAVES), 0, st, logits, g, target, ign, mat,
                                hist, flag, pred_out);
        return;
    }
#define CALL(NS)                                                                                                                \
    do {                                                                                                                        \
        if (sc) hipLaunchKernelGGL((argmax_confmat_cells_kernel<T, NS>), grid, dim3(LS_THREADS), 0, st, logits, g, sc, target,  \
                                   ign, mat, hist, flag, pred_out);                                                             \
        else hipLaunchKernelGGL((argmax_confmat_kernel<T, NS>), grid, dim3(LS_THREADS), 0, st, logits, g, target, ign, mat,     \
                                hist, flag, pred_out);                                                                          \
    } while (0)
    LS_NS_DISPATCH(ns, CALL);
#undef CALL
}

extern "C" int segf_argmax_confmat(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl,
                                   const int64_t* target, int64_t ignore_label, int64_t* mat, int64_t* hist, int32_t* flag,
                                   int64_t* pred_out, void* stream) {
    if (B <= 0 || C <= 0 || C > 192 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || ldl < C || B > 65535) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    LossGeom g{B, C, h, w, H, W, ldl};
    const int ns = (C + 63) / 64;
    const int sc = pow2_scale(h, w, H, W);
    const dim3 grid(LS_NBLK, B);
    unsigned long long* m = (unsigned long long*)mat;
    unsigned long long* hs = (unsigned long long*)hist;
    SEGF_DISPATCH_DT(dt, T, { argmax_launch<T>(ns, sc, grid, st, (const T*)logits, g, target, ignore_label, m, hs, (int*)flag, pred_out); })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- argmax over the class row of NHWC logits (inference: estimate_model.py:104-106 softmax(dim=1).argmax(dim=1); softmax is
// monotonic, so the arg max of the logits is the prediction; lowest index on ties like torch.argmax) -----------------------
template <typename T, int NS>
__global__ void __launch_bounds__(LS_THREADS) argmax_rows_kernel(const T* __restrict__ x, int64_t ld, int64_t rows, int C,
                                                                  int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
        float z[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = lane + 64 * s;
            const float v = ldf<T>(x + r * ld + (c < C ? c : C - 1));
            z[s] = c < C ? v : -INFINITY;
        }
        const int best = wave_argmax<NS>(z, lane, C);
        if (lane == 0) out[r] = best;
    }
}
extern "C" int segf_argmax_rows(int dt, int64_t rows, int C, const void* x, int64_t ld, int64_t* out, void* stream) {
    if (rows <= 0) return 0;
    if (C <= 0 || C > 192 || ld < C) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int ns = (C + 63) / 64;
    const dim3 grid((unsigned)(rows / 4 + 1 < 16384 ? rows / 4 + 1 : 16384));
#define CALL(NS) SEGF_DISPATCH_DT(dt, T, { hipLaunchKernelGGL((argmax_rows_kernel<T, NS>), grid, dim3(LS_THREADS), 0, st, (const T*)x, ld, rows, C, out); })
    LS_NS_DISPATCH(ns, CALL);
#undef CALL
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
