// Band-sweep backward of the fused bilinear upsample (x4) + CrossEntropy + Dice, bf16 logits.
//   reference: models/build_models.py:65 (F.interpolate), engine.py:10-15 (criterion), util/losses.py:126-177, backward.
// Same mathematics as ce_dice_bwd_mfma4_kernel in loss.hip (a *cell* = the 4 x 4 full-resolution pixels between four low-res
// taps; interpolation and tap scatter on the matrix pipe, one exp2 per (pixel, class) on the VALU), different organisation:
//
//   * a WAVE owns a band of 7 low-res tap columns and walks down the image one cell row at a time.  Per cell row it
//     evaluates the 8 cells that touch its columns (12.5 % recomputation in x, none in y apart from one lead-in row per
//     segment) and keeps the gradient of its 2 x 7 live taps (the cell row's upper and lower tap row) in the accumulators of
//     the scatter MFMA: product row 8 * (tap row parity) + column.  When the walk leaves a tap row, that row is complete:
//     it is rounded and stored straight from the registers.  No shared tile, no workgroup barriers, no colouring.
//   * the per-pixel softmax coefficients are folded into the A operand of the scatter (A1 = w * c1 * k, A2 = w * c1 for the
//     class-dependent Dice term, two MFMAs on the same B = exp tile), so the VALU never forms dz per (pixel, class).  The
//     label-path lanes write these products as bf16 directly in A-operand order into a 1 KB shared buffer of their wave.
//   * two cells share one 16x16x32 scatter instruction (k = 2 cells x 16 pixels), where the tile kernel left half of K empty.
//   * a lane fetches its tap's logits as one 16-byte (+ one 4-byte) vector: column n of class tile t is class 8 n + t
//     (tiles 0..7; the rest likewise), which the matrix instructions do not care about.  The same mapping makes the stores
//     16-byte rows.  The scale log2(e) and the stabiliser ride in the interpolation MFMA (A = w * log2 e, C = -max * log2 e).
//   * the [class == label] terms go through float adds in the wave's own shared slab (one instruction = one cell, so the
//     order of colliding adds is fixed) and are folded in when a tap row retires.
// Bitwise reproducible: every sum has a fixed order.  Underflowing cells raise the retry flag (handled by the VALU kernel).
//
// LSE variant (the one the training step runs): the forward band kernel leaves, per (cell, pixel), nl = -log2(sum_c exp z) in
// the exp2 domain (64 bytes per cell, + 9 % traffic).  The backward then takes nl as the C operand of the interpolation MFMA:
// exp2 delivers the probabilities themselves, and the per-cell class maximum (9 max + a 6-step wave reduction), the per-pixel
// sum of exponentials (one add per (pixel, class) + a 16-lane reduction), the reciprocal and the underflow test all disappear
// from the VALU stream: 229 -> 150 vector instructions per 41 exp2.
#include <stdlib.h>
#include "loss_geom.h"

template <int NT>
struct BandParts {      // class tiles are grouped in parts of 8 / 4 / 2 / 1 tiles = that many consecutive classes per lane
    static constexpr int pick(int r) { return r >= 8 ? 8 : (r >= 4 ? 4 : (r >= 2 ? 2 : (r >= 1 ? 1 : 0))); }
    static constexpr int P0 = pick(NT), P1 = pick(NT - P0), P2 = pick(NT - P0 - P1);
    static_assert(P0 + P1 + P2 == NT, "class tile count not covered by three parts");
    static constexpr int words(int p) { return (p + 1) / 2; }
    static constexpr int W0 = 0, W1 = words(P0), W2 = words(P0) + words(P1), NRAW = words(P0) + words(P1) + words(P2);
    static constexpr int part(int t) { return t < P0 ? 0 : (t < P0 + P1 ? 1 : 2); }
    static constexpr int psize(int p) { return p == 0 ? P0 : (p == 1 ? P1 : P2); }
    static constexpr int tbase(int p) { return p == 0 ? 0 : (p == 1 ? P0 : P0 + P1); }
    static constexpr int wbase(int p) { return p == 0 ? W0 : (p == 1 ? W1 : W2); }
};

template <int NT>
__device__ __forceinline__ int band_class(int c, int t) {      // class held by column c of class tile t
    using BP = BandParts<NT>;
    const int p = BP::part(t);
    return 16 * BP::tbase(p) + BP::psize(p) * c + (t - BP::tbase(p));
}
template <int NT>
__device__ __forceinline__ int band_slot(int cls) {             // class -> 16 * tile + column
    using BP = BandParts<NT>;
    constexpr int L0 = BP::P0 == 8 ? 3 : (BP::P0 == 4 ? 2 : (BP::P0 == 2 ? 1 : 0));
    constexpr int L1 = BP::P1 == 8 ? 3 : (BP::P1 == 4 ? 2 : (BP::P1 == 2 ? 1 : 0));
    constexpr int L2 = BP::P2 == 4 ? 2 : (BP::P2 == 2 ? 1 : 0);
    int sh = L0, tb = 0;
    if (BP::P1 > 0 && cls >= 16 * BP::P0) { sh = L1; tb = BP::P0; }
    if (BP::P2 > 0 && cls >= 16 * (BP::P0 + BP::P1)) { sh = L2; tb = BP::P0 + BP::P1; }
    const int rel = cls - 16 * tb;
    return 16 * (tb + (rel & ((1 << sh) - 1))) + (rel >> sh);
}

// Sums of four per-lane values over the 16 lanes of a row group, the sum of v[r] delivered to the lanes with (lane & 15) >> 2
// == r: butterfly with halving payload (xor 8: keep two, xor 7: keep one, then xor 1, xor 2): 6 selects + 5 DPP adds
// instead of 16 DPP adds + a 4-way select.
#define DPP_ROR8 0x128
__device__ __forceinline__ float row_sum16_own(const float (&v)[4], bool lo8, bool even4) {
    float a0 = lo8 ? v[0] : v[2], a1 = lo8 ? v[1] : v[3];
    const float b0 = lo8 ? v[2] : v[0], b1 = lo8 ? v[3] : v[1];
    a0 += dpp_mov<DPP_ROR8>(b0); a1 += dpp_mov<DPP_ROR8>(b1);
    float a = even4 ? a0 : a1;
    const float bb = even4 ? a1 : a0;
    a += dpp_mov<DPP_HALF_MIRROR>(bb);
    a += dpp_mov<DPP_XOR1>(a); a += dpp_mov<DPP_XOR2>(a);
    return a;
}
__device__ __forceinline__ float wave_max_all_fast(float v) {     // operands are never NaN here: no canonicalisation
    float t;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(v));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(t));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(v));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(t));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
    return readlane_f(v, 63);
}

template <int P>
__device__ __forceinline__ void band_load_part(const bf16_t* __restrict__ p, uint32_t* raw) {
    if constexpr (P == 8) { const uint4 v = *reinterpret_cast<const uint4*>(p); raw[0] = v.x; raw[1] = v.y; raw[2] = v.z; raw[3] = v.w; }
    else if constexpr (P == 4) { const uint2 v = *reinterpret_cast<const uint2*>(p); raw[0] = v.x; raw[1] = v.y; }
    else if constexpr (P == 2) { raw[0] = *reinterpret_cast<const uint32_t*>(p); }
    else if constexpr (P == 1) { raw[0] = *p; }
}
template <int P>
__device__ __forceinline__ void band_store_part(bf16_t* __restrict__ p, const float* v) {
    if constexpr (P == 8) {
        uint4 u; u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]); u.z = pack2bf(v[4], v[5]); u.w = pack2bf(v[6], v[7]);
        *reinterpret_cast<uint4*>(p) = u;
    } else if constexpr (P == 4) {
        uint2 u; u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = u;
    } else if constexpr (P == 2) { *reinterpret_cast<uint32_t*>(p) = pack2bf(v[0], v[1]); }
    else if constexpr (P == 1) { *p = f2bf(v[0]); }
}

template <int NT>
struct BandLoads {
    uint32_t raw[BandParts<NT>::NRAW];
    int64_t traw;
    lossf4 nl4;         // LSE variant: -log2(sum exp) of the four pixels (row lane >> 4) of the cell, exp2 domain
    float nl1;          //              and of this lane's label-path pixel
};

#ifndef BAND_SPLIT_D2
#define BAND_SPLIT_D2 1     // 1: the class-dependent Dice part accumulates in its own MFMA accumulators (40 more registers)
#endif
#ifndef BAND_OCC
#define BAND_OCC 2
#endif
#ifndef BAND_AHEAD
#define BAND_AHEAD 1
#endif
#if BAND_AHEAD
#define BAND_AHEAD_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define BAND_AHEAD_FENCE() do {} while (0)
#endif
#define BAND_COLS 7         // tap columns owned by a wave; it evaluates BAND_COLS + 1 cells per cell row

template <int NT, bool FULL0, bool LSE>
__global__ void __launch_bounds__(LS_THREADS, BAND_OCC) ce_dice_bwd_band_kernel(const bf16_t* __restrict__ logits, LossGeom g,
                                                                      const int64_t* __restrict__ target, int64_t ignore_index,
                                                                      const float* __restrict__ cw, int dice,
                                                                      const float* __restrict__ stats,
                                                                      const float* __restrict__ grad_out, bf16_t* __restrict__ dlow,
                                                                      int64_t ldd, int* __restrict__ retry, int nbands, int nseg,
                                                                      int seg_rows, const float* __restrict__ lse) {
    using BP = BandParts<NT>;
    constexpr int NCOL = NT * 16;
    __shared__ __attribute__((aligned(16))) float corr[4][4 * NCOL * 4];        // [wave][row group][tile][column][r]
    __shared__ __attribute__((aligned(16))) uint16_t abuf[4][2][16 * 32];        // [wave][variant][product row][k group][8]
    __shared__ float2 gIw[4][NCOL];                                              // per class: Dice coefficient gI, CE weight
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, gq = lane >> 4;
    const int task = blockIdx.x * 4 + wave;
    if (task >= g.B * nseg * nbands) return;            // no workgroup barrier below
    if (LSE && retry[1]) {                               // the forward met an underflowing cell: its log-sums are not usable
        if (lane == 0) atomicOr(retry, 1);
        return;
    }
    const int band = task % nbands, seg = (task / nbands) % nseg, b = task / (nbands * nseg);
    float* mycorr = corr[wave];
    for (int i = lane; i < 4 * NCOL; i += 64) reinterpret_cast<float4*>(mycorr)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = lane; i < 2 * 16 * 32 / 8; i += 64) reinterpret_cast<uint4*>(&abuf[wave][0][0])[i] = make_uint4(0u, 0u, 0u, 0u);
    float gP[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        float gi;
        const int cls = band_class<NT>(c, t);
        dice_coef_one(stats, b, g.B, g.C, dice, cls, gi, gP[t]);
        if (gq == 0) gIw[wave][cls] = make_float2(gi, (cw && cls < g.C) ? cw[cls] : 1.f);
    }
    // interpolation A operand: lane = (pixel c -> cell row c >> 2, column c & 3; tap gq), scaled to the exp2 domain
    const float wAL = tap_weight(gq, ((c >> 2) + 0.5f) * 0.25f, ((c & 3) + 0.5f) * 0.25f) * LS_LOG2E;
    // label path: lane = (tap kl = c & 3; pixel = cell row gq, column pc = c >> 2)
    const int kl = c & 3, pc = c >> 2;
    const bool lo8 = c < 8, even4 = (pc & 1) == 0;
    const float wLy = (kl >> 1) ? (gq + 0.5f) * 0.25f : 1.f - (gq + 0.5f) * 0.25f;
    const float wLx = (kl & 1) ? (pc + 0.5f) * 0.25f : 1.f - (pc + 0.5f) * 0.25f;
    const float wL = wLy * wLx;
    const float go = grad_out ? grad_out[0] : 1.f;
    const float invW = 1.f / stats[(int64_t)g.B * (3 * g.C + 4) + 1];
    const bf16_t* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * g.H * g.W;
    const uint32_t ldl = (uint32_t)g.ldl;
    const int X0 = BAND_COLS * band;
    const int y_lo = seg * seg_rows, y_hi = y_lo + seg_rows < g.h ? y_lo + seg_rows : g.h;
    // class offsets of this lane's chunks (clamped to a readable chunk when the whole chunk is padding past the row)
    uint32_t coff[3];
    {
        const int o0 = BP::P0 * c, o1 = 16 * BP::P0 + BP::P1 * c, o2 = 16 * (BP::P0 + BP::P1) + BP::P2 * c;
        coff[0] = o0 + BP::P0 <= (int)ldl ? o0 : 0;
        coff[1] = o1 + BP::P1 <= (int)ldl ? o1 : 0;
        coff[2] = o2 + BP::P2 <= (int)ldl ? o2 : 0;
    }
    bool cmask[NT];                                      // class of (c, t) is a real class
#pragma unroll
    for (int t = 0; t < NT; ++t) cmask[t] = band_class<NT>(c, t) < g.C;

    // Address arithmetic: per cell row the lane-dependent parts (tap row, label row) are formed once (RowCtx); per cell only the
    // column parts remain, from wave-uniform scalars.  All offsets are 32-bit byte offsets from wave-uniform base pointers.
    struct RowCtx { uint32_t tap0, tap1, tap2, gat, lab, lrow; bool inY; };
    const char* lseb = reinterpret_cast<const char*>(lse) + (LSE ? (int64_t)b * (g.h + 1) * (g.w + 1) * 64 : 0);
    const char* imgb = reinterpret_cast<const char*>(img);
    const char* tgb = reinterpret_cast<const char*>(tg);
    const uint32_t ldlb = 2u * ldl;
    auto make_row = [&](int cj) {
        RowCtx R;
        const int y0 = cj < 0 ? 0 : cj, y1 = cj + 1 > g.h - 1 ? g.h - 1 : cj + 1;
        const uint32_t rt = (uint32_t)(((gq >> 1) ? y1 : y0) * g.w) * ldlb;
        R.tap0 = rt + 2u * coff[0]; R.tap1 = rt + 2u * coff[1]; R.tap2 = rt + 2u * coff[2];
        R.gat = (uint32_t)(((kl >> 1) ? y1 : y0) * g.w) * ldlb;
        const int Y = 4 * cj + 2 + gq;
        R.inY = Y >= 0 && Y < g.H;
        R.lab = (uint32_t)((Y < 0 ? 0 : (Y >= g.H ? g.H - 1 : Y)) * g.W) * 8u;
        R.lrow = (uint32_t)((cj + 1) * (g.w + 1)) * 64u;
        return R;
    };
    const int pc8 = 8 * pc;
    auto issue = [&](BandLoads<NT>& L, const RowCtx& R, int ck) {
        const int x0 = ck < 0 ? 0 : ck, x1 = ck + 1 > g.w - 1 ? g.w - 1 : ck + 1;
        // (the compiler turns a select of two scalar products into a quarter-rate vector multiply of the select: ask for the
        // full-rate 24-bit one -- tap indices and the row stride are far below 2^24 and the product below 2^31, host-checked)
        const uint32_t xo = (uint32_t)__umul24((uint32_t)((gq & 1) ? x1 : x0), ldlb);
        band_load_part<BP::P0>(reinterpret_cast<const bf16_t*>(imgb + (R.tap0 + xo)), L.raw + BP::W0);
        if constexpr (BP::P1 > 0) band_load_part<BP::P1>(reinterpret_cast<const bf16_t*>(imgb + (R.tap1 + xo)), L.raw + BP::W1);
        if constexpr (BP::P2 > 0) band_load_part<BP::P2>(reinterpret_cast<const bf16_t*>(imgb + (R.tap2 + xo)), L.raw + BP::W2);
        int X8 = 8 * (4 * ck + 2) + pc8;
        X8 = X8 < 0 ? 0 : (X8 > 8 * (g.W - 1) ? 8 * (g.W - 1) : X8);
        L.traw = *reinterpret_cast<const int64_t*>(tgb + (R.lab + (uint32_t)X8));
        if constexpr (LSE) {
            const uint32_t co = R.lrow + (uint32_t)(ck + 1) * 64u + 16u * (uint32_t)gq;
            L.nl4 = *reinterpret_cast<const lossf4*>(lseb + co);
            L.nl1 = *reinterpret_cast<const float*>(lseb + (co + 4u * (uint32_t)pc));
        }
    };
    const bool ign_in_range = ignore_index >= 0 && ignore_index < g.C;

    lossf4 D[NT], D2[NT];                                // gradient of the live taps (row 8 * (tap row & 1) + column): D + gP * D2
    uint32_t Bp[NT][4];                                  // exp tiles of the current cell pair, bf16, scatter B operand
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        D[t] = lossf4{0.f, 0.f, 0.f, 0.f}; D2[t] = lossf4{0.f, 0.f, 0.f, 0.f};
        Bp[t][0] = 0u; Bp[t][1] = 0u; Bp[t][2] = 0u; Bp[t][3] = 0u;
    }
    bool slow = false;
    // Loads run two cells ahead of the arithmetic.  Two register sets, nb[0] for the first cell of a pair and nb[1] for the
    // second: a cell takes what it needs out of its set at the top and then refills the same set for the cell two ahead.  A
    // cursor (pcj, pi) walks the cell sequence of the loops below with every cell row padded to an even number of cells (the
    // phantom cell repeats the last real one), so the parity of a cell within its pair is also the parity of its position in
    // the sequence; it stays on the last cell once it gets there, so every cell issues the same loads.
    // (Measured at batch 128: one cell ahead 2.07 ms, two ahead 2.04 -- memory latency is not what this kernel waits for.  With
    // the label-term LDS adds removed it runs 0.16 ms faster, without the scatter MFMAs 0.14, piecewise-constant labels cost the
    // same as random ones: tools/probe/loss_probe.py.)
    const int nrow = g.w - X0 + 1 < 8 ? g.w - X0 + 1 : 8;      // cells per cell row of this band (>= 2)
    const int nrow2 = (nrow + 1) & ~1;
    int pcj = y_lo - 1, pi = 0;
    auto advance = [&]() {
        if (pi + 1 < nrow2) ++pi;
        else if (pcj + 1 < y_hi) { pi = 0; ++pcj; }
    };
    BandLoads<NT> nb[2];
    RowCtx Rn = make_row(y_lo - 1);
    issue(nb[0], Rn, X0 - 1 + pi); advance();
    issue(nb[1], Rn, X0 - 1 + pi); advance();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (int cj = y_lo - 1; cj < y_hi; ++cj) {
        const int ty_top = cj & 1, ty_bot = ty_top ^ 1;
        const RowCtx Rc = Rn;
        Rn = make_row(cj + 1 < y_hi ? cj + 1 : cj);
        // scatter weights in y: at the clamped image border the real tap takes the whole weight
        const float wye = cj < 0 ? ((kl >> 1) ? 1.f : 0.f) : (cj >= g.h - 1 ? ((kl >> 1) ? 0.f : 1.f) : wLy);
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            if (X0 - 1 + 2 * q > g.w - 1) break;         // wave-uniform: no cell of this pair exists
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int i = 2 * q + s, ck = X0 - 1 + i;
                BandLoads<NT>& L = nb[s];
                auto refill = [&]() {     // the cell two ahead in the walk: in this cell row or in the next one
                    const bool nextrow = pcj > cj;
                    RowCtx Rx;
                    Rx.tap0 = nextrow ? Rn.tap0 : Rc.tap0; Rx.tap1 = nextrow ? Rn.tap1 : Rc.tap1; Rx.tap2 = nextrow ? Rn.tap2 : Rc.tap2;
                    Rx.lab = nextrow ? Rn.lab : Rc.lab; Rx.lrow = nextrow ? Rn.lrow : Rc.lrow; Rx.inY = false; Rx.gat = 0;
                    issue(L, Rx, X0 - 1 + (pi < nrow ? pi : nrow - 1));
                    advance();
                };
                if (ck > g.w - 1) {                       // second cell of the pair past the image: its B half must be finite
#pragma unroll
                    for (int t = 0; t < NT; ++t) { Bp[t][2 * s] = 0u; Bp[t][2 * s + 1] = 0u; }
                    refill();
                    continue;
                }
                // everything this cell needs from its register set, then the refill
                const int64_t traw = L.traw;
                const lossf4 nl4 = L.nl4;
                const float nl1 = L.nl1;
                float u[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int p = BP::part(t), k = t - BP::tbase(p);
                    const uint32_t wd = L.raw[BP::wbase(p) + (k >> 1)];
                    u[t] = __uint_as_float((k & 1) ? (wd & 0xffff0000u) : (wd << 16));
                    if (!(FULL0 && p == 0)) u[t] = cmask[t] ? u[t] : -1e30f;
                }
                // label-class gather first (waiting for the label must not drain the next cells' loads)
                const int x0 = ck < 0 ? 0 : ck, x1 = ck + 1 > g.w - 1 ? g.w - 1 : ck + 1;
                // pixels of the cell inside the image: all columns except at the two border cells (wave-uniform selection of a lane mask)
                const bool inside = Rc.inY && (ck < 0 ? pc >= 2 : (ck >= g.w - 1 ? pc < 2 : true));
                const bool valid0 = inside && (uint64_t)traw < (uint64_t)g.C && !(ign_in_range && traw == ignore_index);
                const int tt = valid0 ? (int)traw : 0;
                // (as the aligned 32-bit word holding it: a 16-bit load is zero-extended at once, which would put the wait here)
                const uint32_t uloff = Rc.gat + (uint32_t)__umul24((uint32_t)((kl & 1) ? x1 : x0), ldlb) + 2u * (uint32_t)tt;
                uint32_t ulbits = *reinterpret_cast<const uint32_t*>(imgb + (uloff & ~3u));
                const float2 gw = gIw[wave][tt];          // (read here: its LDS latency passes under the class tiles)
                refill();
                float mb = 0.f;
                lossf4 cin;
                if constexpr (LSE) cin = nl4;
                else {
                    float mx = u[0];
#pragma unroll
                    for (int t = 1; t < NT; ++t) asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(u[t]));
                    mb = wave_max_all_fast(mx);
                    const float nmb = -mb * LS_LOG2E;
                    cin = lossf4{nmb, nmb, nmb, nmb};
                }
                // scalar adds / fmas on purpose (and -fno-slp-vectorize for this file): beside MFMAs a packed f32 instruction
                // costs far more than the two plain ones it replaces (MI355X_MICROARCH.md, per-instruction constants)
                float s4[4] = {0.f, 0.f, 0.f, 0.f}, dp4[4] = {0.f, 0.f, 0.f, 0.f};
                // the interpolation MFMA of tile t + 1 is issued before the exp2s of tile t (BAND_AHEAD): left to itself the
                // scheduler issues MFMA t, waits out its 10 wait states, then runs the exp2s -- one stall per class tile
                lossf4 zc = __builtin_amdgcn_mfma_f32_16x16x4f32(wAL, u[0], cin, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    lossf4 zn = zc;
                    if (t + 1 < NT) zn = __builtin_amdgcn_mfma_f32_16x16x4f32(wAL, u[t + 1], cin, 0, 0, 0);
                    BAND_AHEAD_FENCE();
                    float e[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        e[r] = __builtin_amdgcn_exp2f(zc[r]);
                        if constexpr (!LSE) s4[r] += e[r];
                        dp4[r] = fmaf(gP[t], e[r], dp4[r]);
                    }
                    Bp[t][2 * s] = pack2bf(e[0], e[1]); Bp[t][2 * s + 1] = pack2bf(e[2], e[3]);
                    BAND_AHEAD_FENCE();
                    zc = zn;
                }
                float tot = 1.f;
                if constexpr (!LSE) tot = row_sum16_own(s4, lo8, even4);
                float dpm = row_sum16_own(dp4, lo8, even4);
                // label path.  The empty asm ties the gathered label logit to a value that exists only now: without it the
                // scheduler converts it right after the gather was issued and the wave waits out the load at the top of the cell
                if constexpr (LSE) asm volatile("" : "+v"(ulbits), "+v"(dpm));
                else asm volatile("" : "+v"(ulbits), "+v"(tot));
                const float ulraw = __uint_as_float((uloff & 2u) ? (ulbits & 0xffff0000u) : (ulbits << 16));
                float zt = LSE ? wL * (ulraw * LS_LOG2E) : wL * ((ulraw - mb) * LS_LOG2E);
                zt += dpp_mov<DPP_XOR1>(zt); zt += dpp_mov<DPP_XOR2>(zt);
                if constexpr (LSE) zt += nl1;
                const bool valid = valid0, under = LSE ? false : (valid && !(tot > 1e-30f));
                slow |= under;
                const float okf = (valid && !under) ? 1.f : 0.f;
                const float inv = LSE ? okf : okf * __builtin_amdgcn_rcpf(fmaxf(tot, 1e-30f));
                const float wce = okf * gw.y * invW;
                const float et = __builtin_amdgcn_exp2f(zt), git = gw.x;
                const float c1 = go * inv;
                const float kk = wce - (dpm + git * et) * inv;                 // wce - <G, p>
                const float dlt = okf * (c1 * et * git - go * wce);            // extra d loss / d z of the label class
                const float wxe = ck < 0 ? ((kl & 1) ? 1.f : 0.f) : (ck >= g.w - 1 ? ((kl & 1) ? 0.f : 1.f) : wLx);
                const float wLe = wye * wxe;
                const int row = ((kl >> 1) ? ty_bot : ty_top) * 8 + ((kl & 1) ? i : (i == 0 ? 7 : i - 1));
                const int ai = row * 32 + gq * 8 + 4 * s + pc;
                const float a2 = wLe * c1;
                abuf[wave][0][ai] = f2bf(a2 * kk);
                abuf[wave][1][ai] = f2bf(a2);
                const int slot = band_slot<NT>(tt);
                atomicAdd(&mycorr[(((row >> 2) * NT + (slot >> 4)) * 16 + (slot & 15)) * 4 + (row & 3)], wLe * dlt);
            }
            // scatter of the pair: D += A1 x E, D2 += A2 x E (combined as D + gP * D2 when the tap row retires)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            union { uint4 u; lossbf8 v; } A1, A2;
            uint4* ap1 = reinterpret_cast<uint4*>(&abuf[wave][0][c * 32 + gq * 8]);
            uint4* ap2 = reinterpret_cast<uint4*>(&abuf[wave][1][c * 32 + gq * 8]);
            A1.u = *ap1; A2.u = *ap2;
            *ap1 = make_uint4(0u, 0u, 0u, 0u); *ap2 = make_uint4(0u, 0u, 0u, 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                union { uint32_t u[4]; lossbf8 v; } Bv;
                Bv.u[0] = Bp[t][0]; Bv.u[1] = Bp[t][1]; Bv.u[2] = Bp[t][2]; Bv.u[3] = Bp[t][3];
                D[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1.v, Bv.v, D[t], 0, 0, 0);
#if BAND_SPLIT_D2
                D2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2.v, Bv.v, D2[t], 0, 0, 0);
#else
                const lossf4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2.v, Bv.v, lossf4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) D[t][r] = fmaf(gP[t], d2[r], D[t][r]);
#endif
            }
        }
        // tap row cj is complete: fold in the label terms, store, and free its accumulators
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if ((gq >> 1) == ty_top) {
            float4* cr = reinterpret_cast<float4*>(mycorr) + (gq * NT) * 16 + c;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 v = cr[t * 16];
                cr[t * 16] = make_float4(0.f, 0.f, 0.f, 0.f);
                D[t][0] = fmaf(gP[t], D2[t][0], D[t][0]) + v.x; D[t][1] = fmaf(gP[t], D2[t][1], D[t][1]) + v.y;
                D[t][2] = fmaf(gP[t], D2[t][2], D[t][2]) + v.z; D[t][3] = fmaf(gP[t], D2[t][3], D[t][3]) + v.w;
            }
            if (cj >= y_lo) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tx = 4 * (gq & 1) + r, X = X0 + tx;
                    if (tx < BAND_COLS && X < g.w) {
                        bf16_t* drow = dlow + (((int64_t)b * g.h + cj) * g.w + X) * ldd;
                        float v[8];
                        {
                            const int o = BP::P0 * c;
#pragma unroll
                            for (int k = 0; k < BP::P0; ++k) v[k] = D[k][r];
                            if (o + BP::P0 <= (int)ldd) band_store_part<BP::P0>(drow + o, v);
                        }
                        if constexpr (BP::P1 > 0) {
                            const int o = 16 * BP::P0 + BP::P1 * c;
#pragma unroll
                            for (int k = 0; k < BP::P1; ++k) v[k] = D[BP::P0 + k][r];
                            if (o + BP::P1 <= (int)ldd) band_store_part<BP::P1>(drow + o, v);
                        }
                        if constexpr (BP::P2 > 0) {
                            const int o = 16 * (BP::P0 + BP::P1) + BP::P2 * c;
#pragma unroll
                            for (int k = 0; k < BP::P2; ++k) v[k] = D[BP::P0 + BP::P1 + k][r];
                            if (o + BP::P2 <= (int)ldd) band_store_part<BP::P2>(drow + o, v);
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) { D[t] = lossf4{0.f, 0.f, 0.f, 0.f}; D2[t] = lossf4{0.f, 0.f, 0.f, 0.f}; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (__any(slow) && lane == 0) atomicOr(retry, 1);
}

// ---- forward ------------------------------------------------------------------------------------------------------
// Same cell front end (vector tap loads, interpolation MFMA with the scale and the stabiliser folded in, one exp2 per
// (pixel, class)); cells need no neighbours here, so a wave simply strides over the cells of its image.  The per-class sums
// P_c = sum_pixels p_pc are a contraction over pixels = one more scatter-type MFMA per pair of cells (A row 0 = ok_p / S_p, B =
// the exp tiles in bf16), instead of four multiply-adds per class tile and cell on the VALU; I_c, T_c and the CE sum stay fp32
// (shared-memory adds of the label path, fixed order).  Output: per-(workgroup, image) partials in ce_dice_fwd_mfma4_kernel's layout.
template <int NT, bool FULL0>
__global__ void __launch_bounds__(LS_THREADS) ce_dice_fwd_band_kernel(const bf16_t* __restrict__ logits, LossGeom g,
                                                                      const int64_t* __restrict__ target, int64_t ignore_index,
                                                                      const float* __restrict__ cw, float* __restrict__ partial,
                                                                      int* __restrict__ retry, float* __restrict__ lse) {
    using BP = BandParts<NT>;
    constexpr int NCOL = NT * 16;
    __shared__ float hI[4][NCOL], hT[4][NCOL], redP[4][NCOL], redS[4][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, gq = lane >> 4;
    // (r05 experiment, removed: one image per XCD -- hardware id -> (xcd = id % 8, slot = id / 8), image = 8 (slot / blocks per image) +
    // xcd.  Every tap row is read by the four cells around it, i.e. by waves of different workgroups, and an image's 128 workgroups sit on
    // all eight XCDs, so every L2 fetches the whole image: with the remap the counter traffic fell 1.74 -> 1.34 GB per launch at batch
    // 128 -- and the launch got 3.5 % SLOWER in the captured step (1.358 -> 1.405 ms): the kernel is bound by VALU issue, not by what it
    // fetches, and sixteen images queued behind each other on one XCD end less evenly than 128 images dealt over all of them.)
    const int b = blockIdx.y;
    const unsigned bx = blockIdx.x;
    for (int i = lane; i < NCOL; i += 64) { hI[wave][i] = 0.f; hT[wave][i] = 0.f; redP[wave][i] = 0.f; }
    const float wAL = tap_weight(gq, ((c >> 2) + 0.5f) * 0.25f, ((c & 3) + 0.5f) * 0.25f) * LS_LOG2E;
    const int kl = c & 3, pc = c >> 2;
    const bool lo8 = c < 8, even4 = (pc & 1) == 0;
    const float wL = tap_weight(kl, (gq + 0.5f) * 0.25f, (pc + 0.5f) * 0.25f);
    const bf16_t* img = logits + (int64_t)b * g.h * g.w * g.ldl;
    const int64_t* tg = target + (int64_t)b * g.H * g.W;
    const char* imgb = reinterpret_cast<const char*>(img);
    const char* tgb = reinterpret_cast<const char*>(tg);
    const uint32_t ldlb = 2u * (uint32_t)g.ldl;
    uint32_t coffb[3];
    {
        const int o0 = BP::P0 * c, o1 = 16 * BP::P0 + BP::P1 * c, o2 = 16 * (BP::P0 + BP::P1) + BP::P2 * c;
        coffb[0] = 2u * (o0 + BP::P0 <= (int)g.ldl ? o0 : 0);
        coffb[1] = 2u * (o1 + BP::P1 <= (int)g.ldl ? o1 : 0);
        coffb[2] = 2u * (o2 + BP::P2 <= (int)g.ldl ? o2 : 0);
    }
    bool cmask[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cmask[t] = band_class<NT>(c, t) < g.C;
    const int ncx = g.w + 1, ncell = (g.h + 1) * ncx;
    const int pc8 = 8 * pc;
    float* lsei = lse ? lse + (int64_t)b * ncell * 16 : nullptr;

    struct CellGeo { int cj, ck; };
    auto geo = [&](int cell) { CellGeo G; G.cj = cell / ncx - 1; G.ck = cell - (G.cj + 1) * ncx - 1; return G; };
    auto issue = [&](BandLoads<NT>& L, const CellGeo& G) {
        const int y0 = G.cj < 0 ? 0 : G.cj, y1 = G.cj + 1 > g.h - 1 ? g.h - 1 : G.cj + 1;
        const int x0 = G.ck < 0 ? 0 : G.ck, x1 = G.ck + 1 > g.w - 1 ? g.w - 1 : G.ck + 1;
        const uint32_t t00 = (uint32_t)(y0 * g.w + x0) * ldlb, t01 = (uint32_t)(y0 * g.w + x1) * ldlb;     // scalar
        const uint32_t t10 = (uint32_t)(y1 * g.w + x0) * ldlb, t11 = (uint32_t)(y1 * g.w + x1) * ldlb;
        const uint32_t tap = (gq >> 1) ? ((gq & 1) ? t11 : t10) : ((gq & 1) ? t01 : t00);
        band_load_part<BP::P0>(reinterpret_cast<const bf16_t*>(imgb + (tap + coffb[0])), L.raw + BP::W0);
        if constexpr (BP::P1 > 0) band_load_part<BP::P1>(reinterpret_cast<const bf16_t*>(imgb + (tap + coffb[1])), L.raw + BP::W1);
        if constexpr (BP::P2 > 0) band_load_part<BP::P2>(reinterpret_cast<const bf16_t*>(imgb + (tap + coffb[2])), L.raw + BP::W2);
        int Y = 4 * G.cj + 2 + gq;
        Y = Y < 0 ? 0 : (Y > g.H - 1 ? g.H - 1 : Y);
        int X8 = 8 * (4 * G.ck + 2) + pc8;
        X8 = X8 < 0 ? 0 : (X8 > 8 * (g.W - 1) ? 8 * (g.W - 1) : X8);
        L.traw = *reinterpret_cast<const int64_t*>(tgb + ((uint32_t)(Y * g.W) * 8u + (uint32_t)X8));
    };

    lossf4 D[NT];                                        // row 0: sum over pixels of p for the classes of this column
    uint32_t Bp[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        D[t] = lossf4{0.f, 0.f, 0.f, 0.f};
        Bp[t][0] = 0u; Bp[t][1] = 0u; Bp[t][2] = 0u; Bp[t][3] = 0u;
    }
    float cel = 0.f, wsum = 0.f, nvalid = 0.f;
    bool bad = false, slow = false;
    const int stride = gridDim.x * 4;
    int cell = (int)bx * 4 + wave;
    // the wave's cells are cell, cell + stride, cell + 2 stride, ...: their (row, column) advance by a fixed step with one carry --
    // no division per cell (it was two per cell: the cell's own and the prefetched one's, ~45 scalar instructions)
    const int dq = stride / ncx, dr = stride - dq * ncx;
    CellGeo Gn = geo(cell < ncell ? cell : 0);           // geometry of the cell whose loads are in flight
    auto advance = [&](const CellGeo& G) {
        CellGeo N;
        int c = G.ck + 1 + dr, r = G.cj + 1 + dq;
        if (c >= ncx) { c -= ncx; r += 1; }
        N.cj = r - 1; N.ck = c - 1;
        return N;
    };
    BandLoads<NT> nx;
    if (cell < ncell) issue(nx, Gn);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (; cell < ncell; cell += 2 * stride) {
        uint32_t Apk[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cidx = cell + s * stride;
            if (cidx >= ncell) {
#pragma unroll
                for (int t = 0; t < NT; ++t) { Bp[t][2 * s] = 0u; Bp[t][2 * s + 1] = 0u; }
                continue;
            }
            const CellGeo G = Gn;
            const BandLoads<NT> cur = nx;
            const int y0 = G.cj < 0 ? 0 : G.cj, y1 = G.cj + 1 > g.h - 1 ? g.h - 1 : G.cj + 1;
            const int x0 = G.ck < 0 ? 0 : G.ck, x1 = G.ck + 1 > g.w - 1 ? g.w - 1 : G.ck + 1;
            const int Y = 4 * G.cj + 2 + gq, X = 4 * G.ck + 2 + pc;
            const bool inside = Y >= 0 && Y < g.H && X >= 0 && X < g.W;
            const bool inrange = (uint64_t)cur.traw < (uint64_t)g.C;
            const bool skip = !inside || cur.traw == ignore_index;
            const bool valid0 = !skip && inrange;
            bad |= !skip && !inrange;
            const int tt = valid0 ? (int)cur.traw : 0;
            const uint32_t g00 = (uint32_t)(y0 * g.w + x0) * ldlb, g01 = (uint32_t)(y0 * g.w + x1) * ldlb;
            const uint32_t g10 = (uint32_t)(y1 * g.w + x0) * ldlb, g11 = (uint32_t)(y1 * g.w + x1) * ldlb;
            const uint32_t uloff = ((kl >> 1) ? ((kl & 1) ? g11 : g10) : ((kl & 1) ? g01 : g00)) + 2u * (uint32_t)tt;
            uint32_t ulbits = *reinterpret_cast<const uint32_t*>(imgb + (uloff & ~3u));
            {
                const int nidx = cidx + stride;
                if (nidx < ncell) Gn = advance(G);            // (past the end: the same cell again, the loads stay unconditional)
                issue(nx, Gn);
            }
            float u[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int p = BP::part(t), k = t - BP::tbase(p);
                const uint32_t wd = cur.raw[BP::wbase(p) + (k >> 1)];
                u[t] = __uint_as_float((k & 1) ? (wd & 0xffff0000u) : (wd << 16));
                if (!(FULL0 && p == 0)) u[t] = cmask[t] ? u[t] : -1e30f;
            }
            float mx = u[0];
#pragma unroll
            for (int t = 1; t < NT; ++t) asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(u[t]));
            const float mb = wave_max_all_fast(mx);
            const float nmb = -mb * LS_LOG2E;
            const lossf4 cin = {nmb, nmb, nmb, nmb};
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            lossf4 zc = __builtin_amdgcn_mfma_f32_16x16x4f32(wAL, u[0], cin, 0, 0, 0);      // one tile ahead, as in the backward
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                lossf4 zn = zc;
                if (t + 1 < NT) zn = __builtin_amdgcn_mfma_f32_16x16x4f32(wAL, u[t + 1], cin, 0, 0, 0);
                BAND_AHEAD_FENCE();
                float e[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { e[r] = __builtin_amdgcn_exp2f(zc[r]); s4[r] += e[r]; }
                Bp[t][2 * s] = pack2bf(e[0], e[1]); Bp[t][2 * s + 1] = pack2bf(e[2], e[3]);
                BAND_AHEAD_FENCE();
                zc = zn;
            }
            float tot = row_sum16_own(s4, lo8, even4);
            asm volatile("" : "+v"(ulbits), "+v"(tot));
            const float ulraw = __uint_as_float((uloff & 2u) ? (ulbits & 0xffff0000u) : (ulbits << 16));
            float zt = wL * ((ulraw - mb) * LS_LOG2E);
            zt += dpp_mov<DPP_XOR1>(zt); zt += dpp_mov<DPP_XOR2>(zt);
            const bool under = valid0 && !(tot > 1e-30f);
            slow |= under;
            const bool ok = valid0 && !under;
            const float inv = ok ? __builtin_amdgcn_rcpf(fmaxf(tot, 1e-30f)) : 0.f;
            const float lg = __builtin_amdgcn_logf(tot);
            if (lse) {
                // -log2(sum exp) of pixel (gq, pc) in the exp2 domain, for the backward (its interpolation MFMA adds it to the
                // scaled logits, so exp2 yields the probability).  The four tap lanes of a pixel store the same word.  A pixel
                // whose sum underflowed (label or not) makes the whole buffer unusable: retry[1]
                slow |= lse != nullptr && !(tot > 1e-30f);
                lsei[cidx * 16 + 4 * gq + pc] = nmb - lg;
            }
            if (ok && kl == 0) {
                const float wt = cw ? cw[tt] : 1.f;
                atomicAdd(&hI[wave][tt], __builtin_amdgcn_exp2f(zt) * inv);
                atomicAdd(&hT[wave][tt], 1.f);
                cel = fmaf(wt, lg - zt, cel);
                wsum += wt; nvalid += 1.f;
            }
            // A operand row 0 (lanes with c == 0): ok / S of pixels (gq, 0..3) = this lane and lanes c = 4, 8, 12 of the row group
            const float a1 = dpp_mov<0x104>(inv), a2 = dpp_mov<0x108>(inv), a3 = dpp_mov<0x10C>(inv);
            const uint32_t p01 = pack2bf(inv, a1), p23 = pack2bf(a2, a3);
            Apk[2 * s] = c == 0 ? p01 : 0u; Apk[2 * s + 1] = c == 0 ? p23 : 0u;
        }
        union { uint32_t u[4]; lossbf8 v; } Av;
        Av.u[0] = Apk[0]; Av.u[1] = Apk[1]; Av.u[2] = Apk[2]; Av.u[3] = Apk[3];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            union { uint32_t u[4]; lossbf8 v; } Bv;
            Bv.u[0] = Bp[t][0]; Bv.u[1] = Bp[t][1]; Bv.u[2] = Bp[t][2]; Bv.u[3] = Bp[t][3];
            D[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Av.v, Bv.v, D[t], 0, 0, 0);
        }
    }
    if (__any(slow) && lane == 0) { atomicOr(retry, 1); if (lse) atomicOr(retry + 1, 1); }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (gq == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) redP[wave][band_class<NT>(c, t)] = D[t][0];
    }
    const float ce = LS_LN2 * wave_sum_all(cel), ws = wave_sum_all(wsum), nv = wave_sum_all(nvalid);
    if (lane == 0) { redS[wave][0] = ce; redS[wave][1] = ws; redS[wave][2] = nv; redS[wave][3] = __any(bad) ? 1.f : 0.f; }
    __syncthreads();
    float* dst = partial + ((int64_t)bx * g.B + b) * (3 * g.C + 4);
    for (int i = threadIdx.x; i < NCOL; i += LS_THREADS) {
        if (i < g.C) {
            dst[i] = (hI[0][i] + hI[1][i]) + (hI[2][i] + hI[3][i]);
            dst[g.C + i] = (redP[0][i] + redP[1][i]) + (redP[2][i] + redP[3][i]);
            dst[2 * g.C + i] = (hT[0][i] + hT[1][i]) + (hT[2][i] + hT[3][i]);
        }
    }
    if (threadIdx.x < 4) dst[3 * g.C + threadIdx.x] = (redS[0][threadIdx.x] + redS[1][threadIdx.x]) + (redS[2][threadIdx.x] + redS[3][threadIdx.x]);
}

bool loss_band_fwd_covers(const bf16_t* logits, LossGeom g) {
    if (POL(loss_no_band) || POL(loss_no_band_fwd)) return false;
    if (g.H != 4 * g.h || g.W != 4 * g.w || g.C > 192) return false;
    if (g.ldl % 8 || g.ldl < g.C || ((uintptr_t)logits & 15)) return false;
    if ((int64_t)g.h * g.w * g.ldl >= (1ll << 30) || (int64_t)g.H * g.W >= (1ll << 28)) return false;
    return true;
}

bool loss_band_fwd_launch(const bf16_t* logits, LossGeom g, const int64_t* target, int64_t ignore_index, const float* cw,
                          float* partial, int* retry, float* lse, hipStream_t st) {
    if (!loss_band_fwd_covers(logits, g)) return false;
    const int nt = (g.C + 15) / 16;
    const int NTb = nt <= 2 ? 2 : (nt <= 4 ? 4 : (nt <= 10 ? 10 : 12));
    const dim3 grid(LS_NBLK, g.B);
#define BAND_CALL(NT, FULL) hipLaunchKernelGGL((ce_dice_fwd_band_kernel<NT, FULL>), grid, dim3(LS_THREADS), 0, st, logits, g, target, \
                                               ignore_index, cw, partial, retry, lse)
    if (NTb == 2) { if (g.C >= 32) BAND_CALL(2, true); else BAND_CALL(2, false); }
    else if (NTb == 4) { if (g.C >= 64) BAND_CALL(4, true); else BAND_CALL(4, false); }
    else if (NTb == 10) { if (g.C >= 128) BAND_CALL(10, true); else BAND_CALL(10, false); }
    else { if (g.C >= 128) BAND_CALL(12, true); else BAND_CALL(12, false); }
#undef BAND_CALL
    return true;
}

static int band_seg_rows(int B, int h, int nbands) {
    if (POL(loss_band_rows) > 0) return POL(loss_band_rows);
    // enough wave tasks for several rounds of the chip (256 CUs x 8 resident waves of this kernel) so that the tail evens out;
    // segments of at least 8 tap rows (each segment recomputes one cell row).  Measured at B = 128, 128 x 128 taps: 16 rows
    // 2.29 ms, 32 rows 2.33 ms, 64 rows 2.35 ms, 128 rows 2.93 ms
    int rows = h;
    while (rows > 8 && (int64_t)B * nbands * ((h + rows - 1) / rows) < 6 * 3072) rows = (rows + 1) / 2;
    return rows;
}

bool loss_band_bwd_launch(const bf16_t* logits, LossGeom g, const int64_t* target, int64_t ignore_index, const float* cw, int dice,
                          const float* stats, const float* grad_out, bf16_t* dlow, int64_t ldd, int* retry, const float* lse,
                          hipStream_t st) {
    if (POL(loss_no_band)) return false;
    if (lse && !loss_band_fwd_covers(logits, g)) return false;        // (the caller has checked this: the buffer was never written)
    if (g.H != 4 * g.h || g.W != 4 * g.w || g.C > 192) return false;
    const int nt = (g.C + 15) / 16;
    const int NTb = nt <= 2 ? 2 : (nt <= 4 ? 4 : (nt <= 10 ? 10 : 12));
    if (g.ldl % 8 || ldd % 8 || ldd > 16 * NTb || g.ldl < g.C || ldd < g.C) return false;
    if (((uintptr_t)logits | (uintptr_t)dlow) & 15) return false;
    if ((int64_t)g.h * g.w * g.ldl >= (1ll << 31) || (int64_t)g.H * g.W >= (1ll << 31)) return false;
    if (g.ldl >= (1 << 22) || g.w >= (1 << 22)) return false;             // 24-bit multiplies in the tap addressing
    const int nbands = (g.w + BAND_COLS - 1) / BAND_COLS;
    const int seg_rows = band_seg_rows(g.B, g.h, nbands);
    const int nseg = (g.h + seg_rows - 1) / seg_rows;
    const int64_t tasks = (int64_t)g.B * nseg * nbands;
    const dim3 grid((unsigned)((tasks + 3) / 4));
#define BAND_CALL(NT, FULL) do { if (lse) hipLaunchKernelGGL((ce_dice_bwd_band_kernel<NT, FULL, true>), grid, dim3(LS_THREADS), 0, st, logits, g, target, \
                                               ignore_index, cw, dice, stats, grad_out, dlow, ldd, retry, nbands, nseg, seg_rows, lse);        \
        else hipLaunchKernelGGL((ce_dice_bwd_band_kernel<NT, FULL, false>), grid, dim3(LS_THREADS), 0, st, logits, g, target, \
                                               ignore_index, cw, dice, stats, grad_out, dlow, ldd, retry, nbands, nseg, seg_rows, lse); } while (0)
    if (NTb == 2) { if (g.C >= 32) BAND_CALL(2, true); else BAND_CALL(2, false); }
    else if (NTb == 4) { if (g.C >= 64) BAND_CALL(4, true); else BAND_CALL(4, false); }
    else if (NTb == 10) { if (g.C >= 128) BAND_CALL(10, true); else BAND_CALL(10, false); }
    else { if (g.C >= 128) BAND_CALL(12, true); else BAND_CALL(12, false); }
#undef BAND_CALL
    return true;
}
