// GEMM for nn.Linear / 1x1 conv / im2col'd conv: forward (x W^T), data gradient (dy W) and weight
// gradient (dy^T x) -- reference models/backbones/mit.py:45,52,58,98-99, models/heads/segformer.py:13,24,39.
//
// bf16 path: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 as 4x4 v_mfma_f32_16x16x32_bf16 tiles,
//   register-staged global->LDS double buffering (one barrier per K step), XOR-swizzled LDS images:
//     K-contiguous operand  : [128 rows][64 k] 128-B rows, 16-B chunk c stored at c ^ (row & 7) -> ds_read_b128 conflict-free
//     reduction-major operand: [64 k][128 cols] 256-B rows, chunk c stored at c ^ 2*((k&3) + 4*((k>>3)&1)),
//                              fragments through ds_read_b64_tr_b16 (hardware transpose), conflict-free
//   MFMA is issued "swapped" (A := weight-side rows n, B := token-side cols m) so each lane owns 4 consecutive n
//   of one output row m -> 8-byte (bf16) / 16-byte (f32) row-contiguous stores.
// f32 path: exact-fp32 FMA kernel (64x64x16 tile, 4x4 per thread) used by the parity mode.
#include <stdlib.h>
#include <type_traits>
#include "colreduce.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

struct GemmArgs {
    const void* A; const void* B; void* C;
    const float* bias; const void* residual; const float* rscale;
    int64_t M, N, K, lda, ldb, ldc, ldr, rpg;
    int64_t kchunk;          // K range per grid.z slice
    float* ws;               // split-K partials [z][M][N] (nullptr when split_k == 1)
    int a_vec, b_vec, c_vec, r_vec, use_tr;
    int fast;                // uniform guard-free tile loads for full K steps (debug switch SEGFAC_GEMM_NO_FASTLOAD)
    int xcd_slabs;           // layout 2, split-K: the tiles of one K slab on one XCD (gemm_bf16_tile)
    int c_vec16;             // C rows allow 16-byte stores (bf16 output, LDS-staged epilogue)
    // implicit 3x3 convolution (stride 1, pad 1) over an NHWC operand [B][cH][cW][ld >= cC]: the gathered operand's K (layout 0,
    // operand A) or N (layout 2, operand B) axis is (tap = ky*3+kx, channel); csign = +1 reads pixel + offset(tap) (forward,
    // weight gradient), -1 reads pixel - offset(tap) (data gradient = correlation of dy with the transposed weights)
    int cH, cW, cC, csign;
    // layout 2 only: colsum[m] = sum_k A(k, m) (the bias gradient next to the weight gradient): the first unused column of the
    // last column tile is staged as all-ones, so its accumulators are the column sums; colsum_ws = split-K partials [z][M]
    float* colsum; float* colsum_ws;
    // operand prologue (256-tile kernel, layouts 0 / 2): the ACTIVATION operand (A in layout 0, B in layout 2) is read as
    // act(x * pro_scale[g][f] + pro_shift[g][f]), g = token / pro_rpg, f = feature -- BatchNorm(+ReLU)(+Dropout2d scale) of
    // the producer applied on the way into LDS, so the normalised tensor is never materialised
    const float* pro_scale; const float* pro_shift; int64_t pro_rpg; int64_t pro_ld; int pro_act;
    // fp8 operands (256-tile kernel, layout 0, template FP8): A / B hold OCP fp8 bytes and every K-axis quantity of this struct (K,
    // lda, ldb, kchunk, cC) counts 2-BYTE UNITS -- the loaders and LDS images move bytes and never look inside -- C[m][n] =
    // acc * f8_sa[0] * f8_sb[n]: one dequantisation scale for the activation tensor, one per weight row
    const float* f8_sa; const float* f8_sb;
};

#define GB_BM 128
#define GB_BN 128
#define GB_BK 64
#define GB_TILE_BYTES 16384
#define GB_STG_LD 132      // fp32 epilogue staging row stride (128 + 4 pad)

__device__ __forceinline__ uint4 ld8_bf16_guard(const bf16_t* p, int nvalid) {
    bf16_t t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = j < nvalid ? p[j] : (bf16_t)0;
    uint4 u;
    u.x = t[0] | ((uint32_t)t[1] << 16); u.y = t[2] | ((uint32_t)t[3] << 16);
    u.z = t[4] | ((uint32_t)t[5] << 16); u.w = t[6] | ((uint32_t)t[7] << 16);
    return u;
}

// K-contiguous operand: element (row, k) at base[row*ld + k]; tile = rows [row0,row0+128) x k [k0,k0+64)
__device__ __forceinline__ void gload_kc(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rmax,
                                         int64_t k0, int64_t kend, int vec, uint4 (&reg)[4]) {
    const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
    const int64_t k = k0 + c * 8;
    if ((vec & 2) && k0 + GB_BK <= kend) {
        // whole K step in range (workgroup-uniform): four plain vector loads in flight together.  Rows past the operand are
        // clamped, not zeroed: they only feed output rows / columns that are never stored.  (A per-load `if (in range)`
        // costs a divergent branch and an s_waitcnt vmcnt(0) per load, i.e. 8 serialised HBM latencies per K step.)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + r + 32 * i;
            reg[i] = *reinterpret_cast<const uint4*>(base + (row < rmax ? row : rmax - 1) * ld + k);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + r + 32 * i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < rmax && k < kend) {
            const bf16_t* p = base + row * ld + k;
            if (vec && k + 8 <= kend) v = *reinterpret_cast<const uint4*>(p);
            else v = ld8_bf16_guard(p, (int)(kend - k < 8 ? kend - k : 8));
        }
        reg[i] = v;
    }
}
__device__ __forceinline__ void swrite_kc(unsigned char* tile, const uint4 (&reg)[4]) {
    const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r + 32 * i;
        *reinterpret_cast<uint4*>(tile + row * 128 + ((c ^ (row & 7)) << 4)) = reg[i];
    }
}
// reduction-major operand: element (k, col) at base[k*ld + col]; tile = k [k0,k0+64) x cols [col0,col0+128)
__device__ __forceinline__ int rm_swz(int krow) { return 2 * ((krow & 3) + 4 * ((krow >> 3) & 1)); }
__device__ __forceinline__ void gload_rm(const bf16_t* __restrict__ base, int64_t ld, int64_t col0, int64_t cmax,
                                         int64_t k0, int64_t kend, int vec, uint4 (&reg)[4]) {
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int64_t col = col0 + c * 8;
    if ((vec & 2) && k0 + GB_BK <= kend && (cmax & 7) == 0) {      // see gload_kc: uniform fast path, clamped columns
        const bf16_t* p = base + (k0 + r) * ld + (col < cmax ? col : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) reg[i] = *reinterpret_cast<const uint4*>(p + (int64_t)(16 * i) * ld);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t k = k0 + r + 16 * i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < kend && col < cmax) {
            const bf16_t* p = base + k * ld + col;
            if (vec && col + 8 <= cmax) v = *reinterpret_cast<const uint4*>(p);
            else v = ld8_bf16_guard(p, (int)(cmax - col < 8 ? cmax - col : 8));
        }
        reg[i] = v;
    }
}
__device__ __forceinline__ void swrite_rm(unsigned char* tile, const uint4 (&reg)[4]) {
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int krow = r + 16 * i;
        *reinterpret_cast<uint4*>(tile + krow * 256 + ((c ^ rm_swz(krow)) << 4)) = reg[i];
    }
}

// ---- branch-free loaders for the deep-prefetch loop (gemm_bf16_kernel<.., DEEP>): every thread issues exactly four 16-byte loads per
// operand and K step, whatever the step -- the compiler can then COUNT the loads of the younger step when it waits for the older
// one.  Rows / columns past the operand are clamped (they only feed outputs that are never stored); a chunk or row past the end
// of K re-reads a valid one of the same row / the last valid row (same cache lines as the valid lanes: no extra HBM traffic) and
// is zeroed when the registers go to LDS (`ok` bit i = load i is real).  Returns the ok bits.
__device__ __forceinline__ unsigned gload_kc_nb(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rmax,
                                                int64_t k0, int64_t kend, uint4 (&reg)[4]) {
    const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
    const int64_t k = k0 + c * 8;
    const bool ok = k < kend;
    const int64_t kc = ok ? k : kend - 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + r + 32 * i;
        reg[i] = *reinterpret_cast<const uint4*>(base + (row < rmax ? row : rmax - 1) * ld + kc);
    }
    return ok ? 0xfu : 0u;
}
__device__ __forceinline__ unsigned gload_rm_nb(const bf16_t* __restrict__ base, int64_t ld, int64_t col0, int64_t cmax,
                                                int64_t k0, int64_t kend, uint4 (&reg)[4]) {
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int64_t col = col0 + c * 8;
    const bf16_t* p = base + (col < cmax ? col : 0);
    unsigned ok = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t k = k0 + r + 16 * i;
        ok |= (k < kend ? 1u : 0u) << i;
        reg[i] = *reinterpret_cast<const uint4*>(p + (k < kend ? k : kend - 1) * ld);
    }
    return ok;
}
__device__ __forceinline__ void zero_unless_ok(uint4 (&reg)[4], unsigned ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t m = (ok >> i) & 1u ? 0xffffffffu : 0u;
        reg[i].x &= m; reg[i].y &= m; reg[i].z &= m; reg[i].w &= m;
    }
}

// implicit-im2col variants: the tile rows are output pixels, k = tap * C + channel.  All index arithmetic that does not change
// from one K step to the next lives in a per-thread ConvState set up once per workgroup (the tap / channel split and the pixel
// coordinates are integer divisions -- 64-bit ones cost ~80 instructions each, and the loaders used to redo up to nine of them
// per K step); the loaders then only advance the state by one K step.
struct ConvState {
    int ry[4], rx[4];      // K-contiguous gather (layout 0, operand A): output pixel of each of the thread's four tile rows
    int tap, ci;           //   tap / channel of the thread's 8-column chunk at the CURRENT K step
    int dyc, dxc, cci;     // reduction-major gather (layout 2, operand B): tap offset and channel of the thread's fixed column chunk
    int py[4], px[4];      //   pixel coordinates of the thread's four K rows at the current K step
};
// rows: row index of the thread's i-th tile row = row0 + rbase + rstride * i
__device__ __forceinline__ void conv_state_init_kc(ConvState& st, const GemmArgs& a, int64_t row0, int rbase, int rstride, int64_t k) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + rbase + rstride * i;
        st.rx[i] = (int)(row % a.cW);
        st.ry[i] = (int)((row / a.cW) % a.cH);
    }
    st.tap = (int)(k / a.cC);
    st.ci = (int)(k - (int64_t)st.tap * a.cC);
}
__device__ __forceinline__ void conv_state_init_rm(ConvState& st, const GemmArgs& a, int64_t col, int64_t cmax, int64_t k0, int rbase,
                                                   int rstride) {
    const int64_t cc = col < cmax ? col : 0;
    const int tap = (int)(cc / a.cC);
    st.cci = (int)(cc - (int64_t)tap * a.cC);
    st.dyc = a.csign * (tap / 3 - 1); st.dxc = a.csign * (tap % 3 - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t k = k0 + rbase + rstride * i;
        st.px[i] = (int)(k % a.cW);
        st.py[i] = (int)((k / a.cW) % a.cH);
    }
}
// K-contiguous gather; advances (tap, ci) by one K step afterwards
__device__ __forceinline__ unsigned gload_kc_conv_s(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rmax, int64_t k0,
                                                    int64_t kend, const GemmArgs& a, ConvState& st, int rbase, int rstride,
                                                    uint4 (&reg)[4]) {
    // Every gather is unconditional: padding taps / rows past the end read a clamped in-range address and are zeroed at the
    // LDS write through the returned validity bits (a branch per load would serialise the four loads, see gload_kc)
    const int64_t k = k0 + (threadIdx.x & 7) * 8;
    const bool kok = k < kend;
    const int tap = kok ? st.tap : 0, ci = kok ? st.ci : 0;
    const int dy = a.csign * (tap / 3 - 1), dx = a.csign * (tap % 3 - 1);
    unsigned okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + rbase + rstride * i;
        const int yy = st.ry[i] + dy, xx = st.rx[i] + dx;
        const bool ok = row < rmax && kok && yy >= 0 && yy < a.cH && xx >= 0 && xx < a.cW;
        okm |= ok ? 1u << i : 0u;
        const int64_t src = ok ? row + (int64_t)dy * a.cW + dx : (row < rmax ? row : rmax - 1);
        reg[i] = *reinterpret_cast<const uint4*>(base + src * ld + ci);
    }
    st.ci += GB_BK;
    while (st.ci >= a.cC) { st.ci -= a.cC; ++st.tap; }
    return okm;
}
// reduction-major gather (tile rows = pixels k, columns = (tap, channel)); advances the pixel coordinates by one K step
__device__ __forceinline__ unsigned gload_rm_conv_s(const bf16_t* __restrict__ base, int64_t ld, int64_t col, int64_t cmax, int64_t k0,
                                                    int64_t kend, const GemmArgs& a, ConvState& st, int rbase, int rstride,
                                                    uint4 (&reg)[4]) {
    const bool cok = col < cmax;
    unsigned okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t k = k0 + rbase + rstride * i;
        const int64_t kc = k < kend ? k : kend - 1;
        const int yy = st.py[i] + st.dyc, xx = st.px[i] + st.dxc;
        const bool ok = k < kend && cok && yy >= 0 && yy < a.cH && xx >= 0 && xx < a.cW;
        okm |= ok ? 1u << i : 0u;
        reg[i] = *reinterpret_cast<const uint4*>(base + (ok ? kc + (int64_t)st.dyc * a.cW + st.dxc : kc) * ld + st.cci);
        st.px[i] += GB_BK;
        while (st.px[i] >= a.cW) { st.px[i] -= a.cW; if (++st.py[i] == a.cH) st.py[i] = 0; }
    }
    return okm;
}
__device__ __forceinline__ void zero_invalid(uint4 (&reg)[4], unsigned okm) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (!((okm >> i) & 1u)) reg[i] = make_uint4(0, 0, 0, 0);
}

// fragment for mfma_f32_16x16x32_bf16: lane l holds X[idx = l&15][k = 8*(l>>4) + j], j = 0..7 of k-step s
__device__ __forceinline__ bf16x8 frag_kc(const unsigned char* tile, int rb, int s, int lane) {
    const int row = rb + (lane & 15);
    const int chunk = 4 * s + (lane >> 4);
    const uint4 u = *reinterpret_cast<const uint4*>(tile + row * 128 + ((chunk ^ (row & 7)) << 4));
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ bf16x8 frag_rm_tr(const unsigned char* tile, int cb, int s, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int u = (cb >> 2) + p;          // 8-byte unit inside the 256-B row
    const int chunk = u >> 1, half = u & 1;
    s16x4 lo, hi;
    {
        const int krow = 32 * s + 8 * g + q;
        const unsigned char* a = tile + krow * 256 + ((chunk ^ rm_swz(krow)) << 4) + half * 8;
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    }
    {
        const int krow = 32 * s + 8 * g + 4 + q;
        const unsigned char* a = tile + krow * 256 + ((chunk ^ rm_swz(krow)) << 4) + half * 8;
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    }
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 frag_rm_scalar(const unsigned char* tile, int cb, int s, int lane) {
    const int g = lane >> 4, col = cb + (lane & 15);
    s16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int krow = 32 * s + 8 * g + j;
        v[j] = *reinterpret_cast<const short*>(tile + krow * 256 + (((col >> 3) ^ rm_swz(krow)) << 4) + (col & 7) * 2);
    }
    return __builtin_bit_cast(bf16x8, v);
}

template <typename OutT> __device__ __forceinline__ void store4(OutT* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}

template <typename InT, typename OutT>
__device__ __forceinline__ void gemm_epilogue4(const GemmArgs& a, int64_t m, int64_t n0, float (&v)[4], unsigned zslice) {
    // v[r] is the accumulator of C[m][n0 + r]
    if (a.colsum && m < a.M && n0 <= a.N && a.N < n0 + 4)      // the all-ones column (layout 2, see gemm_bf16_kernel)
        (a.colsum_ws ? a.colsum_ws + (int64_t)zslice * a.M : a.colsum)[m] = v[a.N - n0];
    if (m >= a.M || n0 >= a.N) return;
    const int nv = (int)(a.N - n0 < 4 ? a.N - n0 : 4);
    if (a.ws) {   // split-K partial: raw accumulators, fp32, ld = N
        float* dst = a.ws + ((int64_t)zslice * a.M + m) * a.N + n0;
        for (int r = 0; r < nv; ++r) dst[r] = v[r];
        return;
    }
    if (a.bias)
        for (int r = 0; r < nv; ++r) v[r] += a.bias[n0 + r];
    if (a.residual) {
        const float s = a.rscale ? a.rscale[m / a.rpg] : 1.f;
        const InT* rp = reinterpret_cast<const InT*>(a.residual) + m * a.ldr + n0;
        for (int r = 0; r < nv; ++r) v[r] = ldf<InT>(rp + r) + s * v[r];
    }
    OutT* dst = reinterpret_cast<OutT*>(a.C) + m * a.ldc + n0;
    if (nv == 4 && a.c_vec) store4<OutT>(dst, v);
    else
        for (int r = 0; r < nv; ++r) stf<OutT>(dst + r, v[r]);
}

// NBUF = 2: double-buffered K loop (64 KB LDS, 2 workgroups per CU).  NBUF = 1: launches whose K fits one 64-deep step
// (the stage-1 MiT linears and the folded head products, K = 32 / 64) need no second buffer; 34 KB of LDS lets 4 workgroups
// share a CU, which is what hides the load -> MFMA -> store latency chain of these purely HBM-bound launches.
// DEEP (layouts 0 / 1, no gather): TWO K steps of operand loads in flight (two register sets, branch-free loaders) -- with K = 160 ...
// 1024 a tile is a chain of 3 ... 16 load -> LDS -> MFMA round trips whose arithmetic is a few hundred cycles each; the MiT
// stage-3 / 4 products ([131072 x 160] x [160 x 160]: 51 us against 14 us of HBM time) are bound by that chain, not by bytes.
// The tile program is a device function of (arguments, logical grid, linear workgroup id) so that a GROUPED launch can run the tiles
// of several products from one grid (gemm_bf16_dw_group_kernel: the weight gradients of up to GDW_MAX Linear layers).
template <int LAYOUT, typename OutT, bool TR, bool CONV = false, int NBUF = 2, bool DEEP = false>
__device__ __forceinline__ void gemm_bf16_tile(const GemmArgs& a, const unsigned gx, const unsigned gy, const unsigned gz, const unsigned orig) {
    static_assert(!DEEP || (!CONV && NBUF == 2), "deep prefetch: no gather, double-buffered LDS");
    constexpr int SMEM_BYTES = NBUF == 2 ? 4 * GB_TILE_BYTES : (64 * GB_STG_LD * 4 > 2 * GB_TILE_BYTES ? 64 * GB_STG_LD * 4 : 2 * GB_TILE_BYTES);
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
    unsigned char (*smem)[2][GB_TILE_BYTES] = reinterpret_cast<unsigned char (*)[2][GB_TILE_BYTES]>(smem_raw);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so the hardware id is
    // remapped (bijectively) such that each XCD walks a CONTIGUOUS range of logical tiles, n-tile fastest: the column tiles
    // of one row tile -- and, for split-K, all tiles of one K slab -- run back to back on one XCD and share their operand
    // through that L2 instead of each fetching it from HBM.  Speed only; any placement computes the same result.
    const unsigned nwg = gx * gy * gz;
    const unsigned q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    // (split-K weight-gradient launches keep the hardware order: measured slower with the remap)
    const unsigned wgid = LAYOUT == 2 ? orig : (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    unsigned bx = wgid % gx, by = (wgid / gx) % gy, bz = wgid / (gx * gy);
    if (LAYOUT == 2 && !CONV && a.xcd_slabs) {
        // r05, split-K weight gradients: ids that are equal modulo 8 run on one XCD, so the K SLAB is made the index that moves with
        // (id mod 8) and the tiles of a slab follow each other at a stride of 8 ids: all gx * gy tiles of a slab -- which read the same
        // rows of dy and x -- run on ONE XCD at about the same time and share them through its L2, and every XCD gets the same number of
        // slabs.  In hardware order the 30 tiles of a [1280 x 320] slab were dealt over all eight L2s: 2.8 x the operand bytes crossed
        // the fabric (cfg4: 11.1 GB per grouped launch).  Slabs past the last multiple of 8 keep the hardware order.
        const unsigned T = gx * gy, full = (gz >> 3) << 3;
        if (orig < full * T) {
            const unsigned t = (orig >> 3) % T;
            bz = (orig & 7) + 8 * (orig / (8 * T));
            bx = t % gx; by = t / gx;
        } else {
            // fewer than eight slabs (left): inside a slab, pin the tile index of the WIDER operand to the XCD -- the [K x 3072] operand of
            // a [3072 x 768] gradient is then fetched by one L2 per row tile (its six column tiles run there) instead of by six, the
            // narrow operand still by all eight: 2.4 x the operand bytes instead of 6.4 x.  Needs a multiple of 8 tiles along that index.
            const unsigned r = orig - full * T;
            const bool pin_y = gy >= gx;
            const unsigned P = pin_y ? gy : gx, Q = pin_y ? gx : gy;
            if ((P & 7u) == 0) {
                const unsigned j = r >> 3, per = (P >> 3) * Q;
                const unsigned sl = j / per, w = j - sl * per, pi = w / Q, qq = w - pi * Q, pp = (r & 7u) + 8 * pi;
                bz = full + sl;
                if (pin_y) { by = pp; bx = qq; } else { bx = pp; by = qq; }
            }
        }
    }
    const int64_t m0 = (int64_t)by * GB_BM, n0 = (int64_t)bx * GB_BN;
    const int64_t kbeg = (int64_t)bz * a.kchunk;
    const int64_t kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(a.B);

    f32x4 acc[4][4];   // [tn][tm]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // bias gradient riding on the weight gradient (layout 2, a.colsum): the first unused column of the last column tile is
    // staged as all-ones, so its accumulator column is sum_k A(k, m); csn = that column's index inside this tile (or -1)
    const int csn = (LAYOUT == 2 && !CONV && a.colsum != nullptr && a.N >= n0 && a.N < n0 + GB_BN) ? (int)(a.N - n0) : -1;
    // ... and when N fills its last column tile (no spare column anywhere: N % 128 == 0) the column sums come from four extra
    // MFMAs per K step against an all-ones B fragment, in the waves that hold the first 64 columns of the first column tile
    // (wave-uniform): still one pass over dy, no separate column-reduction launch
    const bool cs_mfma = LAYOUT == 2 && !CONV && a.colsum != nullptr && (a.N % GB_BN) == 0 && bx == 0 && wn == 0;
    f32x4 accs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) accs[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 fones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
    uint4 ra[4], rb[4];
    ConvState cst;
    if (CONV && LAYOUT == 0) conv_state_init_kc(cst, a, m0, threadIdx.x >> 3, 32, kbeg + (threadIdx.x & 7) * 8);
    if (CONV && LAYOUT == 2) conv_state_init_rm(cst, a, n0 + (threadIdx.x & 15) * 8, a.N, kbeg, threadIdx.x >> 4, 16);
    unsigned oka = 0xfu, okb = 0xfu;          // validity bits of the gathered (implicit-conv) operand's four loads
    auto gload = [&](int64_t k0) {
        if (LAYOUT == 2) gload_rm(A, a.lda, m0, a.M, k0, kend, a.a_vec, ra);
        else if (CONV) oka = gload_kc_conv_s(A, a.lda, m0, a.M, k0, kend, a, cst, threadIdx.x >> 3, 32, ra);
        else gload_kc(A, a.lda, m0, a.M, k0, kend, a.a_vec, ra);
        if (LAYOUT == 0) gload_kc(B, a.ldb, n0, a.N, k0, kend, a.b_vec, rb);
        else if (CONV && LAYOUT == 2) okb = gload_rm_conv_s(B, a.ldb, n0 + (threadIdx.x & 15) * 8, a.N, k0, kend, a, cst, threadIdx.x >> 4, 16, rb);
        else gload_rm(B, a.ldb, n0, a.N, k0, kend, a.b_vec, rb);
        if (LAYOUT == 2 && csn >= 0 && (int)(threadIdx.x & 15) == (csn >> 3)) {      // this thread stages the ones column
            const uint32_t sh = 16u * (csn & 1), one = 0x3f80u << sh, keep = ~(0xffffu << sh);
            const int wsel = (csn & 7) >> 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (k0 + (int)(threadIdx.x >> 4) + 16 * i < kend) {
                    if (wsel == 0) rb[i].x = (rb[i].x & keep) | one;
                    else if (wsel == 1) rb[i].y = (rb[i].y & keep) | one;
                    else if (wsel == 2) rb[i].z = (rb[i].z & keep) | one;
                    else rb[i].w = (rb[i].w & keep) | one;
                }
            }
        }
    };
    auto swrite = [&](int buf) {
        if (CONV) {
            SEGF_LOADS_ISSUED();                 // keep the zeroing selects (and with them the wait for the gathers) down here
            if (LAYOUT != 2) zero_invalid(ra, oka); else zero_invalid(rb, okb);
        }
        if (LAYOUT == 2) swrite_rm(smem[buf][0], ra); else swrite_kc(smem[buf][0], ra);
        if (LAYOUT == 0) swrite_kc(smem[buf][1], rb); else swrite_rm(smem[buf][1], rb);
    };

    const int nk = (int)((kend - kbeg + GB_BK - 1) / GB_BK);
    auto compute = [&](int buf) {
        const unsigned char* ta = smem[buf][0];
        const unsigned char* tb = smem[buf][1];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int mb = wm * 64 + t * 16, nb = wn * 64 + t * 16;
                if (LAYOUT == 2) fa[t] = TR ? frag_rm_tr(ta, mb, s, lane) : frag_rm_scalar(ta, mb, s, lane);
                else fa[t] = frag_kc(ta, mb, s, lane);
                if (LAYOUT == 0) fb[t] = frag_kc(tb, nb, s, lane);
                else fb[t] = TR ? frag_rm_tr(tb, nb, s, lane) : frag_rm_scalar(tb, nb, s, lane);
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[tm], acc[tn][tm], 0, 0, 0);
            if (LAYOUT == 2 && !CONV && cs_mfma) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) accs[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fones, fa[tm], accs[tm], 0, 0, 0);
            }
        }
    };
    if constexpr (DEEP) {
        uint4 ra1[4], rb1[4];                                  // second register set (the first is ra / rb)
        unsigned ma0 = 0, mb0 = 0, ma1 = 0, mb1 = 0;           // ok bits of the sets
        auto gl = [&](int kt, uint4 (&xa)[4], uint4 (&xb)[4], unsigned& ma, unsigned& mb) {
            const int64_t k0 = kbeg + (int64_t)kt * GB_BK;
            ma = LAYOUT == 2 ? gload_rm_nb(A, a.lda, m0, a.M, k0, kend, xa) : gload_kc_nb(A, a.lda, m0, a.M, k0, kend, xa);
            mb = LAYOUT == 0 ? gload_kc_nb(B, a.ldb, n0, a.N, k0, kend, xb) : gload_rm_nb(B, a.ldb, n0, a.N, k0, kend, xb);
        };
        auto sw = [&](int buf, uint4 (&xa)[4], uint4 (&xb)[4], unsigned ma, unsigned mb) {
            zero_unless_ok(xa, ma); zero_unless_ok(xb, mb);
            if (LAYOUT == 2 && csn >= 0 && (int)(threadIdx.x & 15) == (csn >> 3)) {      // the all-ones column of the bias gradient
                const uint32_t sh = 16u * (csn & 1), one = 0x3f80u << sh, keep = ~(0xffffu << sh);
                const int wsel = (csn & 7) >> 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if ((ma >> i) & 1u) {                  // row k of this load lies inside the K range (A and B share the k rows)
                        if (wsel == 0) xb[i].x = (xb[i].x & keep) | one;
                        else if (wsel == 1) xb[i].y = (xb[i].y & keep) | one;
                        else if (wsel == 2) xb[i].z = (xb[i].z & keep) | one;
                        else xb[i].w = (xb[i].w & keep) | one;
                    }
                }
            }
            if (LAYOUT == 2) swrite_rm(smem[buf][0], xa); else swrite_kc(smem[buf][0], xa);
            if (LAYOUT == 0) swrite_kc(smem[buf][1], xb); else swrite_rm(smem[buf][1], xb);
        };
        // invariant at the top of a step kt (even): buffer 0 holds step kt, set 1 holds step kt + 1 (in flight)
        if (nk > 0) gl(0, ra, rb, ma0, mb0);
        if (nk > 1) gl(1, ra1, rb1, ma1, mb1);
        if (nk > 0) sw(0, ra, rb, ma0, mb0);
        __syncthreads();
        int kt = 0;
        // (the scheduling fences keep the order loads | products | wait + LDS write: left alone, hipcc sinks the loads below the
        // products and hoists the LDS writes above them, and every wait becomes vmcnt(0))
#define DEEP_FENCE() __builtin_amdgcn_sched_barrier(0)
        for (; kt + 3 < nk; kt += 2) {                         // both steps of the pair have a step two ahead to fetch
            gl(kt + 2, ra, rb, ma0, mb0);
            DEEP_FENCE();
            compute(0);
            DEEP_FENCE();
            sw(1, ra1, rb1, ma1, mb1);
            __syncthreads();
            gl(kt + 3, ra1, rb1, ma1, mb1);
            DEEP_FENCE();
            compute(1);
            DEEP_FENCE();
            sw(0, ra, rb, ma0, mb0);
            __syncthreads();
        }
        if (kt + 2 < nk) {                                     // three steps left
            gl(kt + 2, ra, rb, ma0, mb0);
            DEEP_FENCE();
            compute(0);
            DEEP_FENCE();
            sw(1, ra1, rb1, ma1, mb1);
            __syncthreads();
            compute(1);
            sw(0, ra, rb, ma0, mb0);
            __syncthreads();
            compute(0);
        } else if (kt + 1 < nk) {                              // two
            compute(0);
            sw(1, ra1, rb1, ma1, mb1);
            __syncthreads();
            compute(1);
        } else if (kt < nk) {
            compute(0);
        }
#undef DEEP_FENCE
        __syncthreads();
    } else
    {
    if (nk > 0) {
        gload(kbeg);
        swrite(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kbeg + (int64_t)(kt + 1) * GB_BK);
        const unsigned char* ta = smem[buf][0];
        const unsigned char* tb = smem[buf][1];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int mb = wm * 64 + t * 16, nb = wn * 64 + t * 16;
                if (LAYOUT == 2) fa[t] = TR ? frag_rm_tr(ta, mb, s, lane) : frag_rm_scalar(ta, mb, s, lane);
                else fa[t] = frag_kc(ta, mb, s, lane);
                if (LAYOUT == 0) fb[t] = frag_kc(tb, nb, s, lane);
                else fb[t] = TR ? frag_rm_tr(tb, nb, s, lane) : frag_rm_scalar(tb, nb, s, lane);
            }
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[tm], acc[tn][tm], 0, 0, 0);
            if (LAYOUT == 2 && !CONV && cs_mfma) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) accs[tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fones, fa[tm], accs[tm], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) swrite(buf ^ 1);
        __syncthreads();
    }
    }
    if (LAYOUT == 2 && !CONV && cs_mfma && (lane >> 4) == 0) {          // every product row holds the sums: row 0 = register 0 of lanes 0..15
        float* cdst = a.colsum_ws ? a.colsum_ws + (int64_t)bz * a.M : a.colsum;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int64_t m = m0 + wm * 64 + tm * 16 + (lane & 15);
            if (m < a.M) cdst[m] = accs[tm][0];
        }
    }
    // D[i][j]: i (rows, 4*(lane>>4)+r) <-> n, j (cols, lane&15) <-> m
    if (sizeof(OutT) == 2 && !a.ws && a.c_vec16) {
        // bf16 output: stage the fp32 accumulators through LDS (two half-tiles of 64 rows) so that every thread applies the
        // epilogue to 8 consecutive columns of one row and issues a 16-byte store: full 256-byte row segments per 16 lanes
        // instead of 8-byte pieces scattered over 16 rows (the memory-bound GEMMs of the step are store-bound otherwise)
        float* stg = reinterpret_cast<float*>(&smem[0][0][0]);     // [64][GB_STG_LD] floats = 33.8 KB of the 64 KB
        const int tchunk = threadIdx.x & 15, trow = threadIdx.x >> 4;
        const int64_t ncol = n0 + tchunk * 8;
        float bs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bs[j] = (a.bias && ncol + j < a.N) ? a.bias[ncol + j] : 0.f;
        const bool full = ncol + 8 <= a.N;
        for (int half = 0; half < 2; ++half) {
            if (half) __syncthreads();
            if (wm == half) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn)
                        *reinterpret_cast<f32x4*>(stg + (tm * 16 + (lane & 15)) * GB_STG_LD + wn * 64 + tn * 16 + 4 * (lane >> 4)) = acc[tn][tm];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = trow + 16 * i;
                const int64_t m = m0 + half * 64 + r;
                if (m >= a.M || ncol >= a.N) continue;
                float v[8];
                const float4 lo = *reinterpret_cast<const float4*>(stg + r * GB_STG_LD + tchunk * 8);
                const float4 hi = *reinterpret_cast<const float4*>(stg + r * GB_STG_LD + tchunk * 8 + 4);
                v[0] = lo.x + bs[0]; v[1] = lo.y + bs[1]; v[2] = lo.z + bs[2]; v[3] = lo.w + bs[3];
                v[4] = hi.x + bs[4]; v[5] = hi.y + bs[5]; v[6] = hi.z + bs[6]; v[7] = hi.w + bs[7];
                if (a.residual) {
                    const float sc = a.rscale ? a.rscale[m / a.rpg] : 1.f;
                    const bf16_t* rp = reinterpret_cast<const bf16_t*>(a.residual) + m * a.ldr + ncol;
                    float rv[8];
                    if (full && a.r_vec) load8<bf16_t>(rp, rv);
                    else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) rv[j] = ncol + j < a.N ? bf2f(rp[j]) : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = rv[j] + sc * v[j];
                }
                bf16_t* dst = reinterpret_cast<bf16_t*>(a.C) + m * a.ldc + ncol;
                if (full) store8<bf16_t>(dst, v);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (ncol + j < a.N) dst[j] = f2bf(v[j]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
        const int64_t m = m0 + wm * 64 + tm * 16 + (lane & 15);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            const int64_t n = n0 + wn * 64 + tn * 16 + 4 * (lane >> 4);
            float v[4] = {acc[tn][tm][0], acc[tn][tm][1], acc[tn][tm][2], acc[tn][tm][3]};
            gemm_epilogue4<bf16_t, OutT>(a, m, n, v, bz);
        }
    }
}
template <int LAYOUT, typename OutT, bool TR, bool CONV = false, int NBUF = 2, bool DEEP = false>
__global__ void __launch_bounds__(256, 2) gemm_bf16_kernel(GemmArgs a) {
    gemm_bf16_tile<LAYOUT, OutT, TR, CONV, NBUF, DEEP>(a, gridDim.x, gridDim.y, gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
}

// GROUPED weight gradients: the split-K dW = dy^T x products (+ riding bias gradients) of up to GDW_MAX nn.Linear layers in ONE launch.
// At the reference's default batch (4 per GPU, train_gpu.py:71) a MiT-B0 step issues 48 such products of ~16 us each on a handful of
// workgroups, every one followed by its ~5 us split-K reduce -- a quarter of the step (skipping them: 4.71 -> 3.55 ms); the layers'
// gradients do not depend on each other, so their tiles can share one grid.  Member i owns the linear workgroup ids
// [start[i], start[i + 1]); inside its range the member's own logical grid (gx, gy, gz) and arguments are used, so every tile computes
// exactly what the per-layer launch computes (bitwise).  The arguments travel by value (kernel-argument segment): capturable, no table.
#define GDW_MAX 12
struct GemmDwGroup {
    int n;
    unsigned start[GDW_MAX + 1];
    unsigned gx[GDW_MAX], gy[GDW_MAX], gz[GDW_MAX];
    GemmArgs m[GDW_MAX];
};
template <bool DEEP>
__global__ void __launch_bounds__(256, 2) gemm_bf16_dw_group_kernel(const GemmDwGroup g) {
    int i = 0;
#pragma unroll 1
    while (i + 1 < g.n && blockIdx.x >= g.start[i + 1]) ++i;
    gemm_bf16_tile<2, float, true, false, 2, DEEP>(g.m[i], g.gx[i], g.gy[i], g.gz[i], blockIdx.x - g.start[i]);
}

// ---- 256 x 256 x 64 tile variant (8 waves: 2 along m x 4 along n, wave tile 128 x 64 = 8 x 4 MFMA tiles) ----------------
// Twice the operand reuse of the 128^2 tile: each staged byte feeds twice the MFMA work, and a skinny dimension (e.g. the
// 152-wide classifier) fits ONE tile, so the large operand is streamed once instead of once per 128 columns.  Same LDS
// images and fragment algebra as the 128^2 kernel (ds_read_b64_tr_b16 for reduction-major operands), 128 KB LDS
// (double-buffered), one workgroup per CU.
#define GG_B 256
#define GG_THREADS 512
#define GG_TILE_BYTES 32768
#define GG_STG_LD 260
template <int T> __device__ __forceinline__ void gload_kc_t(const bf16_t* __restrict__ base, int64_t ld, int64_t row0, int64_t rmax,
                                                            int64_t k0, int64_t kend, int vec, uint4 (&reg)[4]) {
    const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
    const int64_t k = k0 + c * 8;
    if ((vec & 2) && k0 + GB_BK <= kend) {      // see gload_kc
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + r + (T / 8) * i;
            reg[i] = *reinterpret_cast<const uint4*>(base + (row < rmax ? row : rmax - 1) * ld + k);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + r + (T / 8) * i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < rmax && k < kend) {
            const bf16_t* p = base + row * ld + k;
            if (vec && k + 8 <= kend) v = *reinterpret_cast<const uint4*>(p);
            else v = ld8_bf16_guard(p, (int)(kend - k < 8 ? kend - k : 8));
        }
        reg[i] = v;
    }
}
template <int T> __device__ __forceinline__ void swrite_kc_t(unsigned char* tile, const uint4 (&reg)[4]) {
    const int c = threadIdx.x & 7, r = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r + (T / 8) * i;
        *reinterpret_cast<uint4*>(tile + row * 128 + ((c ^ (row & 7)) << 4)) = reg[i];
    }
}
// reduction-major operand, tile = 64 k rows x R columns (R*2 bytes per row)
template <int R, int T> __device__ __forceinline__ void gload_rm_t(const bf16_t* __restrict__ base, int64_t ld, int64_t col0, int64_t cmax,
                                                                   int64_t k0, int64_t kend, int vec, uint4 (&reg)[4]) {
    constexpr int CPR = R / 8;
    const int c = threadIdx.x % CPR, r = threadIdx.x / CPR;
    const int64_t col = col0 + c * 8;
    if ((vec & 2) && k0 + GB_BK <= kend && (cmax & 7) == 0) {      // see gload_kc
        const bf16_t* p = base + (k0 + r) * ld + (col < cmax ? col : 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) reg[i] = *reinterpret_cast<const uint4*>(p + (int64_t)((T / CPR) * i) * ld);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t k = k0 + r + (T / CPR) * i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (k < kend && col < cmax) {
            const bf16_t* p = base + k * ld + col;
            if (vec && col + 8 <= cmax) v = *reinterpret_cast<const uint4*>(p);
            else v = ld8_bf16_guard(p, (int)(cmax - col < 8 ? cmax - col : 8));
        }
        reg[i] = v;
    }
}
template <int R, int T> __device__ __forceinline__ void swrite_rm_t(unsigned char* tile, const uint4 (&reg)[4]) {
    constexpr int CPR = R / 8;
    const int c = threadIdx.x % CPR, r = threadIdx.x / CPR;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int krow = r + (T / CPR) * i;
        *reinterpret_cast<uint4*>(tile + krow * (R * 2) + ((c ^ rm_swz(krow)) << 4)) = reg[i];
    }
}
template <int R> __device__ __forceinline__ bf16x8 frag_rm_tr_t(const unsigned char* tile, int cb, int s, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int u = (cb >> 2) + p;
    const int chunk = u >> 1, half = u & 1;
    s16x4 lo, hi;
    {
        const int krow = 32 * s + 8 * g + q;
        const unsigned char* a = tile + krow * (R * 2) + ((chunk ^ rm_swz(krow)) << 4) + half * 8;
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    }
    {
        const int krow = 32 * s + 8 * g + 4 + q;
        const unsigned char* a = tile + krow * (R * 2) + ((chunk ^ rm_swz(krow)) << 4) + half * 8;
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    }
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// The operand prologue on 8 bf16 values: y = act(x * sc + sh), rounded to bf16.  r05: the affine on the packed fp32 FMA (two values per
// instruction), the activation AFTER the rounding on the packed 16-bit integer pipe -- a non-negative bf16 pattern orders like a signed
// 16-bit integer and every negative one is < 0, so ReLU is max(pattern, 0) and the upper clamp of ReLU6 is min(pattern, bits(6.0)); both
// commute with the (monotonic) rounding, and "no activation" is max with the most negative / min with the most positive integer:
// 24 vector instructions per 8 values whatever `act` is (was ~54 with run-time selects: the classifier forward spent 39 % of its
// cycles in this function against 19 % in its MFMAs).
__device__ __forceinline__ void pro_apply(uint4& r, const float (&sc)[8], const float (&sh)[8], int act) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    const short lo_s = act >= 1 ? (short)0 : (short)-32768, hi_s = act == 2 ? (short)0x40C0 : (short)0x7FFF;
    const s16x2_t lo = {lo_s, lo_s}, hi = {hi_s, hi_s};
    uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2_t x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u)};
        const f32x2_t s2 = {sc[2 * j], sc[2 * j + 1]}, h2 = {sh[2 * j], sh[2 * j + 1]};
        const f32x2_t y = __builtin_elementwise_fma(x, s2, h2);
        s16x2_t q = __builtin_bit_cast(s16x2_t, pack2bf(y[0], y[1]));
        q = __builtin_elementwise_min(__builtin_elementwise_max(q, lo), hi);
        w[j] = __builtin_bit_cast(uint32_t, q);
    }
    r.x = w[0]; r.y = w[1]; r.z = w[2]; r.w = w[3];
}

// SHAPE 0: the 8 waves as 2 (M) x 4 (N), wave tile 128 x 64.  SHAPE 1 (N <= 160, layout 0): 4 x 2 waves, wave tile 64 x 80 -- only
// the first 160 columns of the 256-wide B tile are multiplied (the classifier's 150 -> 160 classes fill 10 of its 16 column
// tiles: 37.5 % of the MFMA work of SHAPE 0 was padding).  SHAPE 2 (M <= 160, layout 2): 2 x 4 waves, wave tile 80 x 64.
// The loaders and the LDS image are the same for all three.
// DEEP (layout 0, full 64-deep K steps, vector-aligned operands): TWO K steps of operand tiles are in flight in registers while one
// is multiplied (register sets alternate; every load unconditional so that the compiler can wait for one set and leave the other
// in flight).  With one workgroup per CU and one 52 KB step in flight the classifier product was latency-bound at 2.9 TB/s.
// FP8 (layout 0 only): 1 = both operands e4m3, 2 = the token-side operand (A: a gradient) e5m2, the weight side e4m3.  A 16-byte
// fragment then holds 16 values and feeds two v_mfma_f32_16x16x32_fp8 instructions (one per 8-byte half; both operands split the
// same way, so the k pairing is consistent): the same bytes per K step and the same instruction count as bf16, twice the K.
template <int LAYOUT, typename OutT, bool CONV, bool PRO = false, int SHAPE = 0, bool DEEP = false, int FP8 = 0>
__global__ void __launch_bounds__(GG_THREADS) gemm_bf16_big_kernel(GemmArgs a) {
    static_assert(FP8 == 0 || (LAYOUT == 0 && !PRO && !DEEP && SHAPE == 0 && sizeof(OutT) == 2), "fp8: forward-type products, bf16 out");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][GG_TILE_BYTES];
    // DEEP + PRO: the operand affine of this workgroup's sample, staged once (K <= 1024 features): read from global memory
    // inside the K loop, the table loads are younger than the tile loads in flight and waiting for them drains the queue
    __shared__ __attribute__((aligned(16))) float pro_lds[(PRO && DEEP) ? 2 * 1024 : 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int TM = SHAPE == 1 ? 4 : (SHAPE == 2 ? 5 : 8), TN = SHAPE == 1 ? 5 : 4;      // 16 x 16 tiles per wave
    const int wm = SHAPE == 1 ? (wave & 3) : (wave & 1), wn = SHAPE == 1 ? (wave >> 2) : (wave >> 1);
    const int wrow = wm * (16 * TM), wcol = wn * (16 * TN);       // wave tile origin inside the workgroup tile
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned nwg = gx * gy * gridDim.z;
    const unsigned orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const unsigned wgid = LAYOUT == 2 ? orig : (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const unsigned bx = wgid % gx, by = (wgid / gx) % gy, bz = wgid / (gx * gy);
    const int64_t m0 = (int64_t)by * GG_B, n0 = (int64_t)bx * GG_B;
    const int64_t kbeg = (int64_t)bz * a.kchunk;
    const int64_t kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(a.A);
    const bf16_t* B = reinterpret_cast<const bf16_t*>(a.B);

    f32x4 acc[TN][TM];   // [tn][tm]
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 ra[4], rb[4];
    ConvState cst;
    if (CONV && LAYOUT == 0) conv_state_init_kc(cst, a, m0, threadIdx.x >> 3, GG_THREADS / 8, kbeg + (threadIdx.x & 7) * 8);
    if (CONV && LAYOUT == 2)
        conv_state_init_rm(cst, a, n0 + (threadIdx.x % (GG_B / 8)) * 8, a.N, kbeg, threadIdx.x / (GG_B / 8), GG_THREADS / (GG_B / 8));
    unsigned oka = 0xfu, okb = 0xfu;
    int64_t pro_k0 = 0;
    auto gload = [&](int64_t k0) {
        if (LAYOUT == 2) gload_rm_t<GG_B, GG_THREADS>(A, a.lda, m0, a.M, k0, kend, a.a_vec, ra);
        else if (CONV) oka = gload_kc_conv_s(A, a.lda, m0, a.M, k0, kend, a, cst, threadIdx.x >> 3, GG_THREADS / 8, ra);
        else gload_kc_t<GG_THREADS>(A, a.lda, m0, a.M, k0, kend, a.a_vec, ra);
        if (LAYOUT == 0) gload_kc_t<GG_THREADS>(B, a.ldb, n0, a.N, k0, kend, a.b_vec, rb);
        else if (CONV && LAYOUT == 2)
            okb = gload_rm_conv_s(B, a.ldb, n0 + (threadIdx.x % (GG_B / 8)) * 8, a.N, k0, kend, a, cst, threadIdx.x / (GG_B / 8),
                                  GG_THREADS / (GG_B / 8), rb);
        else gload_rm_t<GG_B, GG_THREADS>(B, a.ldb, n0, a.N, k0, kend, a.b_vec, rb);
        if (PRO) pro_k0 = k0;
    };
    auto swrite = [&](int buf) {
        if (CONV) {
            SEGF_LOADS_ISSUED();
            if (LAYOUT != 2) zero_invalid(ra, oka); else zero_invalid(rb, okb);
        }
        if (PRO) {
            // affine of this thread's 8 features for the staged tile (group = sample of the tile's tokens); fetched here, after
            // the MFMA phase, from L1 / L2 (the tables are a few hundred KB): held across the MFMAs the 16 values spill
            SEGF_LOADS_ISSUED();
            float psc[8], psh[8];
            const int64_t grp = (LAYOUT == 0 ? m0 : pro_k0) / a.pro_rpg;
            int64_t f0 = LAYOUT == 0 ? pro_k0 + (threadIdx.x & 7) * 8 : n0 + (threadIdx.x % (GG_B / 8)) * 8;
            f0 = f0 + 8 <= a.pro_ld ? f0 : a.pro_ld - 8;
            if (PRO && DEEP) {
                const float4 s0 = *reinterpret_cast<const float4*>(pro_lds + f0), s1 = *reinterpret_cast<const float4*>(pro_lds + f0 + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(pro_lds + 1024 + f0), h1 = *reinterpret_cast<const float4*>(pro_lds + 1024 + f0 + 4);
                psc[0] = s0.x; psc[1] = s0.y; psc[2] = s0.z; psc[3] = s0.w; psc[4] = s1.x; psc[5] = s1.y; psc[6] = s1.z; psc[7] = s1.w;
                psh[0] = h0.x; psh[1] = h0.y; psh[2] = h0.z; psh[3] = h0.w; psh[4] = h1.x; psh[5] = h1.y; psh[6] = h1.z; psh[7] = h1.w;
            } else {
            load8f(a.pro_scale + grp * a.pro_ld + f0, psc);
            load8f(a.pro_shift + grp * a.pro_ld + f0, psh);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) pro_apply(LAYOUT == 0 ? ra[i] : rb[i], psc, psh, a.pro_act);
        }
        if (LAYOUT == 2) swrite_rm_t<GG_B, GG_THREADS>(smem[buf][0], ra); else swrite_kc_t<GG_THREADS>(smem[buf][0], ra);
        if (LAYOUT == 0) swrite_kc_t<GG_THREADS>(smem[buf][1], rb); else swrite_rm_t<GG_B, GG_THREADS>(smem[buf][1], rb);
    };

    const int nk = (int)((kend - kbeg + GB_BK - 1) / GB_BK);
    auto compute = [&](int buf) {
        const unsigned char* ta = smem[buf][0];
        const unsigned char* tb = smem[buf][1];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fb[TN];
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int nb = wcol + t * 16;
                fb[t] = LAYOUT == 0 ? frag_kc(tb, nb, s, lane) : frag_rm_tr_t<GG_B>(tb, nb, s, lane);
            }
            constexpr int MG = TM == 8 ? 4 : TM;       // row tiles per group (SHAPE 0: the 8 row tiles in two groups of 4: 8 live fragments instead of 12)
#pragma unroll
            for (int hm = 0; hm < TM / MG; ++hm) {
                bf16x8 fa[MG];
#pragma unroll
                for (int t = 0; t < MG; ++t) {
                    const int mb = wrow + (hm * MG + t) * 16;
                    fa[t] = LAYOUT == 2 ? frag_rm_tr_t<GG_B>(ta, mb, s, lane) : frag_kc(ta, mb, s, lane);
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int t = 0; t < MG; ++t)
                    {
                        if constexpr (FP8 == 0) {
                            acc[tn][hm * MG + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[t], acc[tn][hm * MG + t], 0, 0, 0);
                        } else {
                            typedef long f8x2 __attribute__((ext_vector_type(2)));
                            const f8x2 wb = __builtin_bit_cast(f8x2, fb[tn]), xa = __builtin_bit_cast(f8x2, fa[t]);
                            f32x4 c = acc[tn][hm * MG + t];
                            if constexpr (FP8 == 1) {
                                c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wb[0], xa[0], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wb[1], xa[1], c, 0, 0, 0);
                            } else {
                                c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(wb[0], xa[0], c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(wb[1], xa[1], c, 0, 0, 0);
                            }
                            acc[tn][hm * MG + t] = c;
                        }
                    }
            }
        }
    };
    if (DEEP && LAYOUT == 0 && !CONV) {
        if (PRO) {
            const int64_t grp = m0 / a.pro_rpg;
            for (int f = threadIdx.x; f < (int)a.pro_ld; f += GG_THREADS) {
                pro_lds[f] = a.pro_scale[grp * a.pro_ld + f];
                pro_lds[1024 + f] = a.pro_shift[grp * a.pro_ld + f];
            }
            __syncthreads();
        }
        static_assert(!DEEP || SHAPE == 1, "DEEP: the narrow-output shape (N <= 160)");
        // r05: THREE K steps of the streamed operand A in flight (was two), two of the weight operand B, and only the 160 weight rows the
        // narrow shape multiplies are loaded and staged (was all 256, clamped).  The classifier forward was latency-bound: 64 KB of A in
        // flight per CU against ~5 us of loaded-HBM latency is 3.3 TB/s over 256 CUs; 96 KB is the same latency at 1.5 x the rate.  Loads
        // return in order, so a wait for B(t) also waits for every A issued before it: B(t + 2) is issued between A(t + 2) and A(t + 3)
        // -- the wait in front of staging step t + 1 then leaves A(t + 2), B(t + 2), A(t + 3) in flight.  Step s lives in A set s % 3, B set
        // s % 2, LDS buffer s & 1: the loop is unrolled six times so that every set is a fixed register array.
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 sa0[4], sa1[4], sa2[4], sb0[3], sb1[3];
        const int c8 = (threadIdx.x & 7) * 8, r0 = threadIdx.x >> 3;
        const bool b2 = r0 < 32;                     // weight rows 128 + r0 < 160: waves 0-3 (wave-uniform)
        // addresses = a workgroup-uniform base (scalar registers; the K step joins there) + a per-thread 32-bit byte offset that never
        // changes: one vector register per row instead of a 64-bit pointer and a 64-bit add per load
        const unsigned char* baseA = reinterpret_cast<const unsigned char*>(A + m0 * a.lda + kbeg);
        const unsigned char* baseB = reinterpret_cast<const unsigned char*>(B + n0 * a.ldb + kbeg);
        uint32_t oa[4], ob[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t rr = r0 + (GG_THREADS / 8) * i, last = a.M - 1 - m0;
            oa[i] = (uint32_t)(((rr < last ? rr : last) * a.lda + c8) * 2);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int64_t rr = r0 + (GG_THREADS / 8) * i, last = a.N - 1 - n0;
            ob[i] = (uint32_t)(((rr < last ? rr : last) * a.ldb + c8) * 2);
        }
        // unconditional tile loads (K steps past the end re-read the last one and are never multiplied)
#define DEEP_LOAD_A(SA, T_)                                                                                            \
        do {                                                                                                           \
            const unsigned char* pk_ = baseA + (size_t)((T_) < nk ? (T_) : nk - 1) * (GB_BK * 2);                      \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) SA[i_] = *reinterpret_cast<const u32x4*>(pk_ + oa[i_]);   \
        } while (0)
#define DEEP_LOAD_B(SB, T_)                                                                                            \
        do {                                                                                                           \
            const unsigned char* pk_ = baseB + (size_t)((T_) < nk ? (T_) : nk - 1) * (GB_BK * 2);                      \
            SB[0] = *reinterpret_cast<const u32x4*>(pk_ + ob[0]);                                                      \
            SB[1] = *reinterpret_cast<const u32x4*>(pk_ + ob[1]);                                                      \
            SB[2] = *reinterpret_cast<const u32x4*>(pk_ + ob[2]);        /* (rows >= 160: a clamped re-read, not staged) */ \
        } while (0)
#define DEEP_WRITE(SA, SB, T_, BUF)                                                                                    \
        do {                                                                                                           \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) ra[i_] = make_uint4(SA[i_][0], SA[i_][1], SA[i_][2], SA[i_][3]); \
            pro_k0 = kbeg + (int64_t)((T_) < nk ? (T_) : nk - 1) * GB_BK;                                              \
            deep_write_a(BUF);                                                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                                         \
                const int row_ = r0 + (GG_THREADS / 8) * i_;                                                           \
                if (i_ < 2 || b2)                                                                                      \
                    *reinterpret_cast<uint4*>(smem[BUF][1] + row_ * 128 + (((threadIdx.x & 7) ^ (row_ & 7)) << 4)) =   \
                        make_uint4(SB[i_][0], SB[i_][1], SB[i_][2], SB[i_][3]);                                        \
            }                                                                                                          \
        } while (0)
        auto deep_write_a = [&](int buf) {
            if (PRO) {
                float psc[8], psh[8];
                int64_t f0 = pro_k0 + c8;
                f0 = f0 + 8 <= a.pro_ld ? f0 : a.pro_ld - 8;
                const float4 s0 = *reinterpret_cast<const float4*>(pro_lds + f0), s1 = *reinterpret_cast<const float4*>(pro_lds + f0 + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(pro_lds + 1024 + f0), h1 = *reinterpret_cast<const float4*>(pro_lds + 1024 + f0 + 4);
                psc[0] = s0.x; psc[1] = s0.y; psc[2] = s0.z; psc[3] = s0.w; psc[4] = s1.x; psc[5] = s1.y; psc[6] = s1.z; psc[7] = s1.w;
                psh[0] = h0.x; psh[1] = h0.y; psh[2] = h0.z; psh[3] = h0.w; psh[4] = h1.x; psh[5] = h1.y; psh[6] = h1.z; psh[7] = h1.w;
#pragma unroll
                for (int i = 0; i < 4; ++i) pro_apply(ra[i], psc, psh, a.pro_act);
            }
            swrite_kc_t<GG_THREADS>(smem[buf][0], ra);
        };
        // Fragment reads of this path: ONE LDS address per (operand, K sub-step, buffer) and lane -- 8 registers -- with the row tile as the
        // instruction's immediate offset (rows 16 apart = 2048 bytes; a row's swizzle term depends on lane & 7 only, and K sub-step 1 is
        // sub-step 0 with bit 6 of the chunk offset flipped).  The generic compute() recomputes row * 128 + swizzle per fragment, and the
        // compiler kept all 36 addresses in registers.
        typedef __attribute__((address_space(3))) const u32x4* lds_frag_p;
        lds_frag_p fpA[2][2], fpB[2][2];                     // [K sub-step][buffer]
        {
            const uint32_t sbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)&smem[0][0][0];
            const int lr = lane & 15;
            const uint32_t swz0 = (uint32_t)(((lane >> 4) ^ (lr & 7)) << 4);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const uint32_t sw = swz0 ^ (uint32_t)(s2 * 64);
                    fpA[s2][b] = (lds_frag_p)(uintptr_t)(sbase + (uint32_t)b * 2 * GG_TILE_BYTES + (uint32_t)(wrow + lr) * 128 + sw);
                    fpB[s2][b] = (lds_frag_p)(uintptr_t)(sbase + (uint32_t)b * 2 * GG_TILE_BYTES + GG_TILE_BYTES + (uint32_t)(wcol + lr) * 128 + sw);
                }
        }
        auto compute_d = [&](auto bufc) {
            constexpr int BUF = decltype(bufc)::value;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 fb[TN], fa[TM];
#pragma unroll
                for (int t = 0; t < TN; ++t) fb[t] = __builtin_bit_cast(bf16x8, fpB[s2][BUF][t * 128]);
#pragma unroll
                for (int t = 0; t < TM; ++t) fa[t] = __builtin_bit_cast(bf16x8, fpA[s2][BUF][t * 128]);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int t = 0; t < TM; ++t) acc[tn][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[tn], fa[t], acc[tn][t], 0, 0, 0);
            }
        };
        // step s computes from LDS buffer s & 1; behind it: stage step s + 1, then issue B(s + 3) and A(s + 4) into the sets just freed
#define DEEP_STEP(KT, J, SAW, SBW)                                                                                     \
        compute_d(std::integral_constant<int, ((J) & 1)>{});                                                          \
        DEEP_WRITE(SAW, SBW, (KT) + (J) + 1, ((J) + 1) & 1);                                                           \
        DEEP_LOAD_B(SBW, (KT) + (J) + 3);                                                                              \
        DEEP_LOAD_A(SAW, (KT) + (J) + 4);                                                                              \
        __syncthreads();                                                                                               \
        if ((KT) + (J) + 1 >= nk) break;
#define DEEP_SIX(KT)                                                                                                   \
        DEEP_STEP(KT, 0, sa1, sb1) DEEP_STEP(KT, 1, sa2, sb0) DEEP_STEP(KT, 2, sa0, sb1)                               \
        DEEP_STEP(KT, 3, sa1, sb0) DEEP_STEP(KT, 4, sa2, sb1) DEEP_STEP(KT, 5, sa0, sb0)
        DEEP_LOAD_B(sb0, 0);
        DEEP_LOAD_A(sa0, 0);
        DEEP_LOAD_B(sb1, 1);
        DEEP_LOAD_A(sa1, 1);
        DEEP_LOAD_A(sa2, 2);
        DEEP_WRITE(sa0, sb0, 0, 0);
        DEEP_LOAD_B(sb0, 2);
        DEEP_LOAD_A(sa0, 3);
        __syncthreads();
        // The first 12 steps (K <= 768: the whole product) are STRAIGHT-LINE code: at a loop header the compiler's wait-count pass
        // merges the positions of the loads in flight pessimistically and drains the queue (vmcnt(0)) once per trip -- with the two-step
        // loop of r04 that was a full drain every other K step, which is what kept the classifier forward at 3.3 TB/s.
        do { DEEP_SIX(0) DEEP_SIX(6) } while (0);
        for (int kt = 12; kt < nk; kt += 6) { DEEP_SIX(kt) }
#undef DEEP_SIX
#undef DEEP_STEP
#undef DEEP_LOAD_A
#undef DEEP_LOAD_B
#undef DEEP_WRITE
    } else {
        if (nk > 0) {
            gload(kbeg);
            swrite(0);
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) gload(kbeg + (int64_t)(kt + 1) * GB_BK);
            compute(buf);
            if (kt + 1 < nk) swrite(buf ^ 1);
            __syncthreads();
        }
    }
    if (sizeof(OutT) == 2 && !a.ws && a.c_vec16) {
        float* stg = reinterpret_cast<float*>(&smem[0][0][0]);     // [64][GG_STG_LD] floats = 66.5 KB
        const int tchunk = threadIdx.x & 31, trow = threadIdx.x >> 5;
        const int64_t ncol = n0 + tchunk * 8;
        float bs[8], f8s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bs[j] = (a.bias && ncol + j < a.N) ? a.bias[ncol + j] : 0.f;
        if (FP8) {
            const float sa = a.f8_sa[0];
#pragma unroll
            for (int j = 0; j < 8; ++j) f8s[j] = ncol + j < a.N ? sa * a.f8_sb[ncol + j] : 0.f;
        }
        const bool full = ncol + 8 <= a.N;
        for (int pass = 0; pass < 4; ++pass) {              // 64 rows per pass
            if (pass) __syncthreads();
            if (SHAPE == 1 ? wm == pass : wm == (pass >> 1)) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        *reinterpret_cast<f32x4*>(stg + (t * 16 + (lane & 15)) * GG_STG_LD + wcol + tn * 16 + 4 * (lane >> 4)) =
                            acc[tn][SHAPE == 1 ? t : (pass & 1) * 4 + t];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = trow + 16 * i;
                const int64_t m = m0 + pass * 64 + r;
                if (m >= a.M || ncol >= a.N) continue;
                float v[8];
                const float4 lo = *reinterpret_cast<const float4*>(stg + r * GG_STG_LD + tchunk * 8);
                const float4 hi = *reinterpret_cast<const float4*>(stg + r * GG_STG_LD + tchunk * 8 + 4);
                if (FP8) {
                    v[0] = fmaf(lo.x, f8s[0], bs[0]); v[1] = fmaf(lo.y, f8s[1], bs[1]); v[2] = fmaf(lo.z, f8s[2], bs[2]); v[3] = fmaf(lo.w, f8s[3], bs[3]);
                    v[4] = fmaf(hi.x, f8s[4], bs[4]); v[5] = fmaf(hi.y, f8s[5], bs[5]); v[6] = fmaf(hi.z, f8s[6], bs[6]); v[7] = fmaf(hi.w, f8s[7], bs[7]);
                } else {
                v[0] = lo.x + bs[0]; v[1] = lo.y + bs[1]; v[2] = lo.z + bs[2]; v[3] = lo.w + bs[3];
                v[4] = hi.x + bs[4]; v[5] = hi.y + bs[5]; v[6] = hi.z + bs[6]; v[7] = hi.w + bs[7];
                }
                if (a.residual) {
                    const float sc = a.rscale ? a.rscale[m / a.rpg] : 1.f;
                    const bf16_t* rp = reinterpret_cast<const bf16_t*>(a.residual) + m * a.ldr + ncol;
                    float rv[8];
                    if (full && a.r_vec) load8<bf16_t>(rp, rv);
                    else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) rv[j] = ncol + j < a.N ? bf2f(rp[j]) : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = rv[j] + sc * v[j];
                }
                bf16_t* dst = reinterpret_cast<bf16_t*>(a.C) + m * a.ldc + ncol;
                if (full) store8<bf16_t>(dst, v);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (ncol + j < a.N) dst[j] = f2bf(v[j]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int64_t m = m0 + wrow + tm * 16 + (lane & 15);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int64_t n = n0 + wcol + tn * 16 + 4 * (lane >> 4);
            float v[4] = {acc[tn][tm][0], acc[tn][tm][1], acc[tn][tm][2], acc[tn][tm][3]};
            gemm_epilogue4<bf16_t, OutT>(a, m, n, v, bz);
        }
    }
}

// one decision for both the launcher and the split-K / workspace sizing: the 256^2 tile when both dimensions exceed one
// 128 tile, K spans more than one step, and the launch still has enough workgroups for 256 CUs
static inline bool gemm_use_big(int layout, int64_t M, int64_t N, int64_t K) {
    if (POL(gemm_no_big)) return false;
    if (M <= 128 || N <= 128 || K <= GB_BK) return false;
    const int64_t tiles = cdiv64(M, GG_B) * cdiv64(N, GG_B);
    if (layout == 2) return K >= 65536;                       // token-count K: split-K supplies the parallelism (at K = 16384 the
                                                              // 128x128 tile with twice the slices measured 1.2-1.9x faster, at
                                                              // K = 32768 still 1.2-1.4x: [256x256] 53 vs 67 us, [1024x256] 78 vs 98)
    return tiles >= 192;
}

// ---- exact fp32 FMA kernel ---------------------------------------------------------------------------
#define GF_BM 64
#define GF_BN 64
#define GF_BK 16
template <int LAYOUT>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs a) {
    __shared__ float As[GF_BK][GF_BM + 4];
    __shared__ float Bs[GF_BK][GF_BN + 4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;     // tx -> n (4 each), ty -> m (4 each)
    const int64_t m0 = (int64_t)blockIdx.y * GF_BM, n0 = (int64_t)blockIdx.x * GF_BN;
    const int64_t kbeg = (int64_t)blockIdx.z * a.kchunk;
    const int64_t kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const float* A = reinterpret_cast<const float*>(a.A);
    const float* B = reinterpret_cast<const float*>(a.B);
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int64_t k0 = kbeg; k0 < kend; k0 += GF_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + 256 * i;
            int kk, mm;
            if (LAYOUT == 2) { mm = idx & 63; kk = idx >> 6; } else { kk = idx & 15; mm = idx >> 4; }
            const int64_t m = m0 + mm, k = k0 + kk;
            float v = 0.f;
            if (m < a.M && k < kend) v = LAYOUT == 2 ? A[k * a.lda + m] : A[m * a.lda + k];
            As[kk][mm] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + 256 * i;
            int kk, nn;
            if (LAYOUT == 0) { kk = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kk = idx >> 6; }
            const int64_t n = n0 + nn, k = k0 + kk;
            float v = 0.f;
            if (n < a.N && k < kend) v = LAYOUT == 0 ? B[n * a.ldb + k] : B[k * a.ldb + n];
            Bs[kk][nn] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GF_BK; ++kk) {
            const float4 av = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
            const float4 bv = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
            const float am[4] = {av.x, av.y, av.z, av.w};
            const float bn[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(am[i], bn[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        gemm_epilogue4<float, float>(a, m0 + ty * 4 + i, n0 + tx * 4, v, blockIdx.z);
    }
}

// ---- exact fp32 on the MATRIX pipe (r05) -------------------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32: f32 operands, f32 accumulate, bit for bit a k-ordered fmaf chain (one rounding per product, no wider internal
// sum: cdna_hip_programming.md "FP32-input MFMA") at the fp32 VECTOR peak (64 FLOP / clk / SIMD) -- but with one operand register per
// lane per instruction and the vector unit left free, where the FMA kernel above spends four LDS-fed fmaf per loaded value and runs at
// a third of that peak on large shapes and far below it on the small ones (one 4-byte global load per thread and step).  `evaluate`
// runs in fp32 like the reference (engine.py:86-88), so this kernel IS the eval forward's GEMM: 64 x 64 x 16 tiles, four waves of one
// 32 x 32 tile each, operands staged through registers (next step's 16-byte loads in flight under this step's eight MFMAs) into LDS
// images read by ds_read_b128 (K-contiguous operand: 20-float rows, conflict-free) or ds_read_b32 (reduction-major operand).
// The k index of step s in lane half h is 8 h + s for BOTH operands (any pairing of the 16 k values is a valid order of the sum).
// Product orientation: D[n][m] = sum_k W[n][k] X[m][k] -- the N-side operand is the MFMA's A (rows), so a lane ends up with four
// consecutive n of one m per accumulator quad: the epilogue of the other kernels (bias / residual / DropPath scale / split-K slab).
#define GFM_K 16
#define GFM_LDK 20
typedef float gfm_f32x16 __attribute__((ext_vector_type(16)));
template <bool KC>
__device__ __forceinline__ float4 gfm_load(const float* __restrict__ base, int64_t ld, int64_t r0, int64_t rows, int64_t k0, int64_t kend, bool vec, int t) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {                                        // [rows][K]: thread t -> row t >> 2, k chunk 4 (t & 3)
        const int64_t r = r0 + (t >> 2), k = k0 + 4 * (t & 3);
        if (r < rows) {
            const float* p = base + r * ld + k;
            if (vec && k + 3 < kend) v = *reinterpret_cast<const float4*>(p);
            else { if (k < kend) v.x = p[0]; if (k + 1 < kend) v.y = p[1]; if (k + 2 < kend) v.z = p[2]; if (k + 3 < kend) v.w = p[3]; }
        }
    } else {                                         // [K][rows]: thread t -> k row t >> 4, column chunk 4 (t & 15)
        const int64_t k = k0 + (t >> 4), r = r0 + 4 * (t & 15);
        if (k < kend) {
            const float* p = base + k * ld + r;
            if (vec && r + 3 < rows) v = *reinterpret_cast<const float4*>(p);
            else { if (r < rows) v.x = p[0]; if (r + 1 < rows) v.y = p[1]; if (r + 2 < rows) v.z = p[2]; if (r + 3 < rows) v.w = p[3]; }
        }
    }
    return v;
}
// TM x TN tiles of 32 x 32 per wave: workgroup tile (64 TM) x (64 TN).  (2, 2) = 128 x 128 for the large products (32 FLOP per operand
// byte from L2; the 64 x 64 tile's 16 FLOP / B bound the first version at ~64 TFLOP/s), (1, 1) for outputs with few tiles.
// WM = waves along M (2 or 4; 4 / WM along N): (TM, TN, WM) = (1, 5, 4) is a 128 x 160 tile -- a narrow output (the 150 -> 160 classes of
// the decode head, heads/segformer.py:57-58) in ONE column tile, so the [tokens x 768] operand is streamed once and only 6 % of the
// MFMA columns are padding (two 128-wide tiles: 37 %).
template <int LAYOUT, int TM, int TN, int WM = 2>
__global__ void __launch_bounds__(256) gemm_f32_mfma_kernel(GemmArgs a) {
    constexpr bool A_KC = LAYOUT != 2, B_KC = LAYOUT == 0;       // A: [M][K] (layouts 0, 1) or [K][M]; B: [N][K] (layout 0) or [K][N]
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int PM = (BM + 63) / 64, PN = (BN + 63) / 64;      // 64-row loader passes (the LDS images hold whole passes)
    constexpr int LDM = 64 * PM + 4, LDN = 64 * PN + 4;          // reduction-major row strides (floats)
    __shared__ __attribute__((aligned(16))) float Xs[A_KC ? 64 * PM * GFM_LDK : GFM_K * LDM];
    __shared__ __attribute__((aligned(16))) float Ws[B_KC ? 64 * PN * GFM_LDK : GFM_K * LDN];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave % WM, wn = wave / WM, i = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
    const int64_t kbeg = (int64_t)blockIdx.z * a.kchunk;
    const int64_t kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const float* A = reinterpret_cast<const float*>(a.A);
    const float* B = reinterpret_cast<const float*>(a.B);
    const bool avec = (a.a_vec & 1) != 0, bvec = (a.b_vec & 1) != 0;
    gfm_f32x16 acc[TN][TM];
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int v = 0; v < TM; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;
    float4 xr[PM], wr[PN];
#pragma unroll
    for (int j = 0; j < PM; ++j) xr[j] = gfm_load<A_KC>(A, a.lda, m0 + 64 * j, a.M, kbeg, kend, avec, t);
#pragma unroll
    for (int j = 0; j < PN; ++j) wr[j] = gfm_load<B_KC>(B, a.ldb, n0 + 64 * j, a.N, kbeg, kend, bvec, t);
    for (int64_t k0 = kbeg; k0 < kend; k0 += GFM_K) {
#pragma unroll
        for (int j = 0; j < PM; ++j)
            *reinterpret_cast<float4*>(Xs + (A_KC ? (64 * j + (t >> 2)) * GFM_LDK + 4 * (t & 3) : (t >> 4) * LDM + 64 * j + 4 * (t & 15))) = xr[j];
#pragma unroll
        for (int j = 0; j < PN; ++j)
            *reinterpret_cast<float4*>(Ws + (B_KC ? (64 * j + (t >> 2)) * GFM_LDK + 4 * (t & 3) : (t >> 4) * LDN + 64 * j + 4 * (t & 15))) = wr[j];
        __syncthreads();
        if (k0 + GFM_K < kend) {                    // the next step's loads fly under this step's MFMAs
#pragma unroll
            for (int j = 0; j < PM; ++j) xr[j] = gfm_load<A_KC>(A, a.lda, m0 + 64 * j, a.M, k0 + GFM_K, kend, avec, t);
#pragma unroll
            for (int j = 0; j < PN; ++j) wr[j] = gfm_load<B_KC>(B, a.ldb, n0 + 64 * j, a.N, k0 + GFM_K, kend, bvec, t);
        }
        float xf[TM][8], wf[TN][8];
#pragma unroll
        for (int v = 0; v < TM; ++v) {
            const int row = 32 * (TM * wm + v) + i;
            if (A_KC) {
                const float4 u0 = *reinterpret_cast<const float4*>(Xs + row * GFM_LDK + 8 * h);
                const float4 u1 = *reinterpret_cast<const float4*>(Xs + row * GFM_LDK + 8 * h + 4);
                xf[v][0] = u0.x; xf[v][1] = u0.y; xf[v][2] = u0.z; xf[v][3] = u0.w; xf[v][4] = u1.x; xf[v][5] = u1.y; xf[v][6] = u1.z; xf[v][7] = u1.w;
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) xf[v][s] = Xs[(8 * h + s) * LDM + row];
            }
        }
#pragma unroll
        for (int u = 0; u < TN; ++u) {
            const int row = 32 * (TN * wn + u) + i;
            if (B_KC) {
                const float4 u0 = *reinterpret_cast<const float4*>(Ws + row * GFM_LDK + 8 * h);
                const float4 u1 = *reinterpret_cast<const float4*>(Ws + row * GFM_LDK + 8 * h + 4);
                wf[u][0] = u0.x; wf[u][1] = u0.y; wf[u][2] = u0.z; wf[u][3] = u0.w; wf[u][4] = u1.x; wf[u][5] = u1.y; wf[u][6] = u1.z; wf[u][7] = u1.w;
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) wf[u][s] = Ws[(8 * h + s) * LDN + row];
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int u = 0; u < TN; ++u)
#pragma unroll
                for (int v = 0; v < TM; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[u][s], xf[v][s], acc[u][v], 0, 0, 0);
        __syncthreads();
    }
    // D[n][m]: column (lane & 31) = m, rows (r & 3) + 8 (r >> 2) + 4 h = n
#pragma unroll
    for (int v = 0; v < TM; ++v) {
        const int64_t m = m0 + 32 * (TM * wm + v) + i;
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float c4[4] = {acc[u][v][4 * q], acc[u][v][4 * q + 1], acc[u][v][4 * q + 2], acc[u][v][4 * q + 3]};
                gemm_epilogue4<float, float>(a, m, n0 + 32 * (TN * wn + u) + 8 * q + 4 * h, c4, blockIdx.z);
            }
    }
}

// ---- split-K reduction: C = sum_z ws[z] (fixed order): 16 outputs x 16 slice-lanes per workgroup, slice lane s adds
// z = s, s+16, ... with independent loads in flight, then the 16 lane sums are added in fixed order --------------------
// The workgroups past `main_blocks` reduce the bias-gradient slices that ride on the same product (cs_ws [split][cs_n] ->
// cs_out [cs_n], the arithmetic of colreduce_finalize_kernel): one launch per weight gradient instead of two.
template <typename OutT>
__device__ __forceinline__ void splitk_reduce_body(const unsigned bid, const float* __restrict__ ws, int split, int64_t M, int64_t N,
                                                   OutT* __restrict__ C, int64_t ldc, unsigned main_blocks,
                                                   const float* __restrict__ cs_ws, float* __restrict__ cs_out, int64_t cs_n) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const bool cs = bid >= main_blocks;
    const int64_t total = cs ? cs_n : M * N;
    const float* __restrict__ src = cs ? cs_ws : ws;
    const int64_t i = (int64_t)(cs ? bid - main_blocks : bid) * 16 + o;
    float acc = 0.f;
    if (i < total) {
#pragma unroll 4
        for (int z = sl; z < split; z += 16) acc += src[(int64_t)z * total + i];
    }
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < total) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][o];
        if (cs) {
            cs_out[i] = t;
        } else {
            const int64_t m = i / N, n = i - m * N;
            stf<OutT>(C + m * ldc + n, t);
        }
    }
}
template <typename OutT>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ ws, int split, int64_t M, int64_t N,
                                                             OutT* __restrict__ C, int64_t ldc, unsigned main_blocks,
                                                             const float* __restrict__ cs_ws, float* __restrict__ cs_out, int64_t cs_n) {
    splitk_reduce_body<OutT>(blockIdx.x, ws, split, M, N, C, ldc, main_blocks, cs_ws, cs_out, cs_n);
}

// The 16 x 16 form with FOUR adjacent outputs per thread: the same sums in the same order (residue classes z mod 16, added in residue order
// -- bitwise the form above), but a 16-lane group reads 256 contiguous bytes of a slab instead of 64.  N % 4 == 0, ldc % 4 == 0, ws and C
// 16-byte aligned, fp32 output; the bias-gradient slices keep the scalar form.
__device__ __forceinline__ void splitk_reduce_body4(const unsigned bid, const float* __restrict__ ws, int split, int64_t M, int64_t N,
                                                    float* __restrict__ C, int64_t ldc, unsigned main_blocks,
                                                    const float* __restrict__ cs_ws, float* __restrict__ cs_out, int64_t cs_n) {
    __shared__ float4 red[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    if (bid >= main_blocks) {
        float* r1 = reinterpret_cast<float*>(&red[0][0]);                  // [16][17] floats of the same array
        const int64_t i = (int64_t)(bid - main_blocks) * 16 + o;
        float acc = 0.f;
        if (i < cs_n) {
#pragma unroll 4
            for (int z = sl; z < split; z += 16) acc += cs_ws[(int64_t)z * cs_n + i];
        }
        r1[sl * 17 + o] = acc;
        __syncthreads();
        if (sl == 0 && i < cs_n) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += r1[k * 17 + o];
            cs_out[i] = t;
        }
        return;
    }
    const int64_t total = M * N;
    const int64_t i = ((int64_t)bid * 16 + o) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total) {
#pragma unroll 4
        for (int z = sl; z < split; z += 16) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)z * total + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[sl][o] = acc;
    __syncthreads();
    if (sl == 0 && i < total) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float4 v = red[k][o]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const int64_t m = i / N, n = i - m * N;
        *reinterpret_cast<float4*>(C + m * ldc + n) = t;
    }
}
__global__ void __launch_bounds__(256) splitk_reduce4_kernel(const float* __restrict__ ws, int split, int64_t M, int64_t N,
                                                              float* __restrict__ C, int64_t ldc, unsigned main_blocks,
                                                              const float* __restrict__ cs_ws, float* __restrict__ cs_out, int64_t cs_n) {
    splitk_reduce_body4(blockIdx.x, ws, split, M, N, C, ldc, main_blocks, cs_ws, cs_out, cs_n);
}

// Wide form for large outputs (the weight gradients of the wider models: 3072 x 768 fp32 x 8 slices is 75 MB of partials): one
// thread per four consecutive outputs, the slices added in order z = 0, 1, ... with four 16-byte loads in flight -- every slab is
// read as whole contiguous rows.  (The 16 x 16 form above reads 64-byte pieces and ran at ~1 TB/s on these; it stays for small
// outputs with many slices, where this one would leave most of the chip idle.)  N % 4 == 0, ldc % 4 == 0, ws 16-byte aligned.
template <typename OutT>
__device__ __forceinline__ void splitk_reduce_wide_body(const unsigned bid, const float* __restrict__ ws, int split, int64_t M, int64_t N,
                                                        OutT* __restrict__ C, int64_t ldc, unsigned main_blocks,
                                                        const float* __restrict__ cs_ws, float* __restrict__ cs_out, int64_t cs_n) {
    if (bid >= main_blocks) {          // bias-gradient slices riding on the product: [split][cs_n] -> [cs_n], same order
        const int64_t i = (int64_t)(bid - main_blocks) * 256 + threadIdx.x;
        if (i < cs_n) {
            float t = 0.f;
            for (int z = 0; z < split; ++z) t += cs_ws[(int64_t)z * cs_n + i];
            cs_out[i] = t;
        }
        return;
    }
    const int64_t total = M * N;
    const int64_t i = ((int64_t)bid * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const float* __restrict__ src = ws + i;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 4 <= split; z += 4) {
        const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)z * total);
        const float4 b = *reinterpret_cast<const float4*>(src + (int64_t)(z + 1) * total);
        const float4 c = *reinterpret_cast<const float4*>(src + (int64_t)(z + 2) * total);
        const float4 d = *reinterpret_cast<const float4*>(src + (int64_t)(z + 3) * total);
        acc.x = ((acc.x + a.x) + b.x) + c.x + d.x; acc.y = ((acc.y + a.y) + b.y) + c.y + d.y;
        acc.z = ((acc.z + a.z) + b.z) + c.z + d.z; acc.w = ((acc.w + a.w) + b.w) + c.w + d.w;
    }
    for (; z < split; ++z) {
        const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)z * total);
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    const int64_t m = i / N, n = i - m * N;
    OutT* o = C + m * ldc + n;
    if constexpr (sizeof(OutT) == 4) *reinterpret_cast<float4*>(o) = acc;
    else { uint2 u; u.x = pack2bf(acc.x, acc.y); u.y = pack2bf(acc.z, acc.w); *reinterpret_cast<uint2*>(o) = u; }
}
template <typename OutT>
__global__ void __launch_bounds__(256) splitk_reduce_wide_kernel(const float* __restrict__ ws, int split, int64_t M, int64_t N,
                                                                  OutT* __restrict__ C, int64_t ldc, unsigned main_blocks,
                                                                  const float* __restrict__ cs_ws, float* __restrict__ cs_out, int64_t cs_n) {
    splitk_reduce_wide_body<OutT>(blockIdx.x, ws, split, M, N, C, ldc, main_blocks, cs_ws, cs_out, cs_n);
}
// the reduces of a grouped weight-gradient launch, in one launch: member i runs its own form (wide or 16 x 16) over its own block range
struct ReduceGroup {
    int n;
    unsigned start[GDW_MAX + 1], main_blocks[GDW_MAX];
    int wide[GDW_MAX], split[GDW_MAX];
    const float* ws[GDW_MAX]; float* C[GDW_MAX]; const float* cs_ws[GDW_MAX]; float* cs_out[GDW_MAX];
    int64_t M[GDW_MAX], N[GDW_MAX], ldc[GDW_MAX], cs_n[GDW_MAX];
};
__global__ void __launch_bounds__(256) splitk_reduce_group_kernel(const ReduceGroup r) {
    int i = 0;
#pragma unroll 1
    while (i + 1 < r.n && blockIdx.x >= r.start[i + 1]) ++i;
    const unsigned bid = blockIdx.x - r.start[i];
    if (r.wide[i] == 1) splitk_reduce_wide_body<float>(bid, r.ws[i], r.split[i], r.M[i], r.N[i], r.C[i], r.ldc[i], r.main_blocks[i], r.cs_ws[i], r.cs_out[i], r.cs_n[i]);
    else if (r.wide[i] == 2) splitk_reduce_body4(bid, r.ws[i], r.split[i], r.M[i], r.N[i], r.C[i], r.ldc[i], r.main_blocks[i], r.cs_ws[i], r.cs_out[i], r.cs_n[i]);
    else splitk_reduce_body<float>(bid, r.ws[i], r.split[i], r.M[i], r.N[i], r.C[i], r.ldc[i], r.main_blocks[i], r.cs_ws[i], r.cs_out[i], r.cs_n[i]);
}
static inline bool splitk_reduce_is_wide(const float* ws, int64_t M, int64_t N, const void* C, int64_t ldc) {
    return M * N >= 65536 && N % 4 == 0 && ldc % 4 == 0 && !((uintptr_t)ws & 15) && !((uintptr_t)C & 15) && !POL(no_wide_reduce);
}
// form of an fp32 reduce: 1 = wide (slices in order), 2 = 16 x 16 with four outputs per thread, 0 = 16 x 16; and its block counts
static inline int splitk_reduce_form(const float* ws, int64_t M, int64_t N, const void* C, int64_t ldc) {
    if (splitk_reduce_is_wide(ws, M, N, C, ldc)) return 1;
    if (M * N >= 4096 && N % 4 == 0 && ldc % 4 == 0 && !((uintptr_t)ws & 15) && !((uintptr_t)C & 15) && !POL(no_reduce4)) return 2;
    return 0;
}
static inline unsigned splitk_reduce_main_blocks(int form, int64_t total) { return (unsigned)cdiv64(total, form == 1 ? 1024 : (form == 2 ? 64 : 16)); }
static inline unsigned splitk_reduce_cs_blocks(int form, int64_t cs_n) { return (unsigned)cdiv64(cs_n, form == 1 ? 256 : 16); }

// segf_gemm_dw_db_grouped collects the reduce passes of the products it cannot group (streaming / 256-tile kernels) here and issues
// them as grouped launches too: while the sink is set, gemm_impl appends its fp32 reduce instead of launching it
static thread_local ReduceGroup* g_reduce_sink = nullptr;
static bool reduce_sink_take(const float* ws, int split, int64_t M, int64_t N, float* C, int64_t ldc, const float* cs_ws, float* cs_out,
                             int64_t cs_n) {
    ReduceGroup* r = g_reduce_sink;
    if (!r || r->n >= GDW_MAX) return false;
    const int k = r->n;
    const int form = splitk_reduce_form(ws, M, N, C, ldc);
    const unsigned blocks = splitk_reduce_main_blocks(form, M * N), csb = cs_ws ? splitk_reduce_cs_blocks(form, cs_n) : 0u;
    r->wide[k] = form; r->split[k] = split; r->ws[k] = ws; r->C[k] = C; r->cs_ws[k] = cs_ws; r->cs_out[k] = cs_out;
    r->M[k] = M; r->N[k] = N; r->ldc[k] = ldc; r->cs_n[k] = cs_n; r->main_blocks[k] = blocks;
    r->start[k + 1] = r->start[k] + blocks + csb;
    ++r->n;
    return true;
}

// one entry for every split-K product: picks the form by output size
template <typename OutT>
static void splitk_reduce_launch(hipStream_t st, const float* ws, int split, int64_t M, int64_t N, OutT* C, int64_t ldc,
                                 const float* cs_ws, float* cs_out, int64_t cs_n) {
    const int64_t total = M * N;
    const bool wide = splitk_reduce_is_wide(ws, M, N, C, ldc);
    if constexpr (sizeof(OutT) == 4) {
        if (splitk_reduce_form(ws, M, N, C, ldc) == 2) {
            const unsigned blocks = splitk_reduce_main_blocks(2, total), csb = cs_ws ? splitk_reduce_cs_blocks(2, cs_n) : 0u;
            hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(blocks + csb), dim3(256), 0, st, ws, split, M, N, (float*)C, ldc, blocks, cs_ws, cs_out, cs_n);
            return;
        }
    }
    if (wide) {
        const unsigned blocks = (unsigned)cdiv64(total, 1024);
        const unsigned csb = cs_ws ? (unsigned)cdiv64(cs_n, 256) : 0u;
        hipLaunchKernelGGL((splitk_reduce_wide_kernel<OutT>), dim3(blocks + csb), dim3(256), 0, st, ws, split, M, N, C, ldc, blocks,
                           cs_ws, cs_out, cs_n);
    } else {
        const unsigned blocks = (unsigned)cdiv64(total, 16);
        const unsigned csb = cs_ws ? (unsigned)cdiv64(cs_n, 16) : 0u;
        hipLaunchKernelGGL((splitk_reduce_kernel<OutT>), dim3(blocks + csb), dim3(256), 0, st, ws, split, M, N, C, ldc, blocks,
                           cs_ws, cs_out, cs_n);
    }
}

// ---- streaming kernel for huge-M, small-K, small-N products (stage-1/2 linears, the head's stage-1 projection) ------------
// y = x W^T with M ~ 10^6 tokens and K, N <= 128 is pure HBM streaming: one 64..256-byte row in, one out.  The tiled
// kernels above stage both operands through LDS in 128-wide tiles and waste most of a tile on N = 32.  Here the weight
// fragments live in registers for the whole launch (NT * KS <= 16 fragments), a wave walks 16-token groups, and the product
// is computed transposed (C^T = W X^T): the activation rows ARE the MFMA B operand as they lie in memory
// (lane (m = l & 15, g = l >> 4) loads the 16 bytes x[m][32 s + 8 g ..]), no LDS at all; the accumulator rows (features)
// are permuted through the weight-row order so that every lane ends up with runs of 8 consecutive output features of one
// token and stores them with 16-byte writes that tile whole 64-byte segments.  N > 16*NT is covered by blockIdx.y chunks (x is re-read from L2, it is the small side).
__device__ const float skinny_one = 1.f;
template <int LAYOUT, int KS, int NT>
__global__ void __launch_bounds__(256) gemm_skinny_kernel(GemmArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mi = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.y * (16 * NT);
    const bf16_t* __restrict__ A = static_cast<const bf16_t*>(a.A);
    const bf16_t* __restrict__ B = static_cast<const bf16_t*>(a.B);
    bf16_t* __restrict__ C = static_cast<bf16_t*>(a.C);
    const bf16_t* __restrict__ R = static_cast<const bf16_t*>(a.residual);
    // weight fragments: MFMA row i of tile nt carries feature n0 + 32 (nt >> 1) + 8 (i >> 2) + 4 (nt & 1) + (i & 3): the tile
    // pair (2q, 2q+1) gives lane group g the 8 consecutive features 32 q + 8 g .. + 7, so one 16-byte store instruction covers
    // whole 64-byte row segments
    bf16x8 Wf[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + 32 * (nt >> 1) + 8 * (mi >> 2) + 4 * (nt & 1) + (mi & 3);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (LAYOUT == 0) {
                Wf[nt][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(B + (int64_t)n * a.ldb + 32 * s + 8 * g));
            } else {
                s16x8 w;
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = (short)B[(int64_t)(32 * s + 8 * g + j) * a.ldb + n];
                Wf[nt][s] = __builtin_bit_cast(bf16x8, w);
            }
        }
    }
    float bv[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[nt][r] = a.bias ? a.bias[n0 + 32 * (nt >> 1) + 8 * g + 4 * (nt & 1) + r] : 0.f;
    const int64_t ngroups = (a.M + 15) / 16, gstride = (int64_t)gridDim.x * 4;
    int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    // output staging: the accumulator layout gives a lane 16 bytes of one token, so a store instruction wrote 64-byte pieces of
    // 16 rows (half cache lines: the [2M x 32] -> 768 projection ran at 3.1 TB/s where a plain fill reaches 6.8).  The wave's
    // 16 x (16 NT) tile goes through its own LDS slab and leaves as whole 32 NT-byte row runs, 16 bytes per lane.
    // (Dealing the column chunks of the same rows to neighbouring workgroups on top of this measured no better.)
    constexpr int SROW = 32 * NT + 16;                                  // staged row bytes (+16: bank spread)
    __shared__ __attribute__((aligned(16))) unsigned char ostage[4][16 * SROW];
    unsigned char* ost = ostage[wave];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 xa[KS], xb[KS];
    auto rowptr = [&](int64_t gp) {
        int64_t m = gp * 16 + mi;
        m = m < a.M ? m : a.M - 1;
        return A + m * a.lda + 8 * g;
    };
    {
        const bf16_t* p = rowptr(grp < ngroups ? grp : 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) xa[s] = *reinterpret_cast<const u32x4*>(p + 32 * s);
    }
    // per-sample DropPath scale of the residual form, one group ahead like the rows (a null table reads a constant 1)
    const float* rsp = a.rscale ? a.rscale : &skinny_one;
    float rsn;
    {
        const int64_t m0_ = (grp < ngroups ? grp : 0) * 16 + mi;
        rsn = rsp[a.rscale ? (m0_ < a.M ? m0_ : a.M - 1) / a.rpg : 0];
    }
    // One group of 16 tokens.  FULL = every row exists and HASR = a residual operand are compile-time: all loads and stores of
    // the main loop are unconditional, so the compiler waits for the prefetched rows with a counted vmcnt and the group's stores
    // stay in flight (a guarded store or load forces vmcnt(0): the wave drained its stores before every next group).
    auto one_group = [&](int64_t gp, auto full_tag, auto r_tag) {
        constexpr bool FULL = decltype(full_tag)::value, HASR = decltype(r_tag)::value;
        {
            const int64_t nxt = gp + gstride;
            const bf16_t* p = rowptr(nxt < ngroups ? nxt : gp);           // unconditional (the last one re-reads)
#pragma unroll
            for (int s = 0; s < KS; ++s) xb[s] = *reinterpret_cast<const u32x4*>(p + 32 * s);
            SEGF_LOADS_ISSUED();                                          // (or the scheduler sinks the prefetch down to the stores)
        }
        const int64_t m = gp * 16 + mi;
        const bool mok = FULL || m < a.M;
        const float rs = rsn;                                             // this group's DropPath scale (loaded one group ahead)
        if (HASR) {
            const int64_t nxt = gp + gstride, mn = (nxt < ngroups ? nxt : gp) * 16 + mi;
            rsn = rsp[a.rscale ? (mn < a.M ? mn : a.M - 1) / a.rpg : 0];
        }
        u32x4 rres[(NT + 1) / 2];
        if (HASR) {
            const bf16_t* rp = R + (mok ? m : a.M - 1) * a.ldr + n0 + 8 * g;
#pragma unroll
            for (int q = 0; q < (NT + 1) / 2; ++q) rres[q] = *reinterpret_cast<const u32x4*>(rp + 32 * q);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[nt][s], __builtin_bit_cast(bf16x8, xa[s]), acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = acc[2 * q][r] + bv[2 * q][r]; v[4 + r] = acc[2 * q + 1][r] + bv[2 * q + 1][r]; }
            if (HASR) {
                const u32x4 w = rres[q];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[2 * j] = __uint_as_float(w[j] << 16) + rs * v[2 * j];
                    v[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u) + rs * v[2 * j + 1];
                }
            }
            uint4 o;
            o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); o.z = pack2bf(v[4], v[5]); o.w = pack2bf(v[6], v[7]);
            *reinterpret_cast<uint4*>(ost + mi * SROW + (32 * q + 8 * g) * 2) = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            constexpr int CPR = 2 * NT;                                  // 16-byte chunks per staged row
#pragma unroll
            for (int i = 0; i < (16 * CPR) / 64; ++i) {
                const int idx = lane + 64 * i, row = idx / CPR, cc = idx - row * CPR;
                const uint4 o = *reinterpret_cast<const uint4*>(ost + row * SROW + 16 * cc);
                const int64_t mr = gp * 16 + row;
                if (FULL || mr < a.M) *reinterpret_cast<uint4*>(C + mr * a.ldc + n0 + 8 * cc) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xa[s] = xb[s];
            asm volatile("" : "+v"(xa[s]));       // keep the copy (and its counted wait) here, behind this group's stores
        }
    };
    const int64_t nfull = a.M / 16;                                       // groups whose 16 rows all exist
    // everything loaded so far is consumed here once: otherwise its first use sits inside the loop and the loop header gets a
    // vmcnt(0) that also waits for the previous group's stores on every trip
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) asm volatile("" : "+v"(Wf[nt][s2]));
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bv[nt][r]));
    }
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) asm volatile("" : "+v"(xa[s2]));
    asm volatile("" : "+v"(rsn));
    if (R) {
        for (; grp < nfull; grp += gstride) one_group(grp, std::true_type{}, std::true_type{});
        if (grp < ngroups) one_group(grp, std::false_type{}, std::true_type{});
    } else {
        for (; grp < nfull; grp += gstride) one_group(grp, std::true_type{}, std::false_type{});
        if (grp < ngroups) one_group(grp, std::false_type{}, std::false_type{});     // at most one ragged group, in one wave
    }
}

// Wide-output form of the streaming product (layout 0, N = NCH chunks of 128 features, e.g. the folded head's [2M x 32] -> 768
// projection): with the column chunks as separate workgroups every output row was written in six 256-byte pieces at six
// different times (847 us = 3.9 TB/s for a product that is 96 % stores).  Here the whole weight matrix sits in LDS in fragment
// order (N x K bf16 <= 48 KB), a wave loops over the chunks for its 16 tokens, stages the [16 x N] result in its own LDS slab and
// writes WHOLE rows (N * 2 bytes contiguous per token, 16 bytes per lane).  No residual operand (bias only).
template <int KS, int NCH>
__global__ void __launch_bounds__(256) gemm_skinny_rows_kernel(GemmArgs a) {
    constexpr int NT = 8, NF = NCH * NT * KS;                         // fragments of 1 KB
    constexpr int SROW = NCH * 256 + 16;                              // staged output row bytes
    __shared__ __attribute__((aligned(16))) unsigned char wlds[NF * 1024];
    __shared__ __attribute__((aligned(16))) unsigned char ostage[4][16 * SROW];
    __shared__ float blds[NCH * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mi = lane & 15, g = lane >> 4;
    const bf16_t* __restrict__ A = static_cast<const bf16_t*>(a.A);
    const bf16_t* __restrict__ B = static_cast<const bf16_t*>(a.B);
    bf16_t* __restrict__ C = static_cast<bf16_t*>(a.C);
    // fragment (chunk c, tile nt, step s), lane (mi, g): 16 bytes of weight row n = 128 c + 32 (nt >> 1) + 8 (mi >> 2) + 4 (nt & 1) + (mi & 3)
    for (int f = wave; f < NF; f += 4) {
        const int s = f % KS, nt = (f / KS) % NT, c = f / (KS * NT);
        const int n = 128 * c + 32 * (nt >> 1) + 8 * (mi >> 2) + 4 * (nt & 1) + (mi & 3);
        *reinterpret_cast<uint4*>(wlds + f * 1024 + lane * 16) = *reinterpret_cast<const uint4*>(B + (int64_t)n * a.ldb + 32 * s + 8 * g);
    }
    for (int i = threadIdx.x; i < NCH * 128; i += 256) blds[i] = a.bias ? a.bias[i] : 0.f;
    __syncthreads();
    unsigned char* ost = ostage[wave];
    const int64_t ngroups = (a.M + 15) / 16, gstride = (int64_t)gridDim.x * 4;
    int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 xa[KS], xb[KS];
    auto rowptr = [&](int64_t gp) {
        int64_t m = gp * 16 + mi;
        m = m < a.M ? m : a.M - 1;
        return A + m * a.lda + 8 * g;
    };
    {
        const bf16_t* p = rowptr(grp < ngroups ? grp : 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) xa[s] = *reinterpret_cast<const u32x4*>(p + 32 * s);
    }
    // One group of 16 tokens.  FULL = every row exists: all loads and stores unconditional, so that the compiler can count them
    // and wait for the prefetched rows with vmcnt(#stores) -- with a guarded store it waits vmcnt(0), i.e. for the 24 KB of stores
    // of the group to DRAIN, before every next group (the wave then alternates between computing and draining: 4.5 TB/s).
    auto one_group = [&](int64_t gp, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        {
            const int64_t nxt = gp + gstride;
            const bf16_t* p = rowptr(nxt < ngroups ? nxt : gp);           // unconditional (the last one re-reads)
#pragma unroll
            for (int s = 0; s < KS; ++s) xb[s] = *reinterpret_cast<const u32x4*>(p + 32 * s);
            SEGF_LOADS_ISSUED();                                          // (or the scheduler sinks the prefetch down to the stores)
        }
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 wf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wlds + ((c * NT + nt) * KS + s) * 1024 + lane * 16));
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, xa[s]), acc[nt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int q = 0; q < NT / 2; ++q) {
                const float4 b0 = *reinterpret_cast<const float4*>(blds + 128 * c + 32 * q + 8 * g);
                const float4 b1 = *reinterpret_cast<const float4*>(blds + 128 * c + 32 * q + 8 * g + 4);
                uint4 o;
                o.x = pack2bf(acc[2 * q][0] + b0.x, acc[2 * q][1] + b0.y); o.y = pack2bf(acc[2 * q][2] + b0.z, acc[2 * q][3] + b0.w);
                o.z = pack2bf(acc[2 * q + 1][0] + b1.x, acc[2 * q + 1][1] + b1.y); o.w = pack2bf(acc[2 * q + 1][2] + b1.z, acc[2 * q + 1][3] + b1.w);
                *reinterpret_cast<uint4*>(ost + mi * SROW + (128 * c + 32 * q + 8 * g) * 2) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            constexpr int CPR = NCH * 16;                                 // 16-byte chunks per output row
#pragma unroll
            for (int i = 0; i < (16 * CPR) / 64; ++i) {
                const int idx = lane + 64 * i, row = idx / CPR, cc = idx - row * CPR;
                const uint4 o = *reinterpret_cast<const uint4*>(ost + row * SROW + 16 * cc);
                const int64_t mr = gp * 16 + row;
                if (FULL || mr < a.M) *reinterpret_cast<uint4*>(C + mr * a.ldc + 8 * cc) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xa[s] = xb[s];
            asm volatile("" : "+v"(xa[s]));
        }
    };
    const int64_t nfull = a.M / 16;                                       // groups whose 16 rows all exist
    for (; grp < nfull; grp += gstride) one_group(grp, std::true_type{});
    if (grp < ngroups) one_group(grp, std::false_type{});                 // at most one ragged group, in one wave
}

// shapes the streaming kernel takes: bf16 in/out, layouts 0/1, K in {32, 64, 128}, N a multiple of the chunk width,
// 16-byte aligned rows on every operand, enough tokens to amortise the register-resident weights
// ---- streaming products with a WIDE reduction and a narrow output: y[M][N] = x[M][K] W, N = 32 / 64, K = 256 .. 1024, M ~ 10^6 --
// (the folded SegFormerHead's stage-1 / stage-2 data gradients: K = 768, 3.2 GB of dy per launch at cfg2; the stage-2 MLP's
// fc2 forward and fc1 data gradient: K = 256.  The tiled kernel fills a quarter / half of its 128 output columns and ran the
// head's term at 3.8 TB/s.)  The four waves of a workgroup split K: wave w keeps the weight fragments of its K / 4 columns in
// registers and, per 16-token group, loads its 16-byte pieces of the rows straight into the MFMA B operand (as
// gemm_skinny_kernel does: transposed product, no LDS on the way in; next group's loads in flight).  The four partial
// [16 x N] tiles meet in LDS (two slabs, one barrier per group) and are added in fixed order wave 0..3 by all 256 threads, which
// apply the epilogue (bias, residual + per-sample DropPath scale) and write the group's bf16 rows in whole runs.
// LAYOUT 0: W stored [N][K] (forward); LAYOUT 1: W stored [K][N] (data gradient).
template <int LAYOUT, int KS, int NT>
__global__ void __launch_bounds__(256, 2) gemm_skinny_k_kernel(GemmArgs a) {
    static_assert(NT == 2 || NT == 4, "32 or 64 output features");
    constexpr int NF = 16 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mi = lane & 15, g = lane >> 4;
    const bf16_t* __restrict__ A = static_cast<const bf16_t*>(a.A);
    const bf16_t* __restrict__ B = static_cast<const bf16_t*>(a.B);
    bf16_t* __restrict__ C = static_cast<bf16_t*>(a.C);
    const bf16_t* __restrict__ R = static_cast<const bf16_t*>(a.residual);
    const int k0 = wave * (32 * KS);
    // weight fragments: MFMA row i of tile nt carries feature 32 (nt >> 1) + 8 (i >> 2) + 4 (nt & 1) + (i & 3), so that lane group g
    // ends up with the 8 consecutive features 32 q + 8 g .. + 7 of token mi from the tile pair (2 q, 2 q + 1)
    bf16x8 Wf[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = 32 * (nt >> 1) + 8 * (mi >> 2) + 4 * (nt & 1) + (mi & 3);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (LAYOUT == 0) {
                Wf[nt][s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(B + (int64_t)n * a.ldb + k0 + 32 * s + 8 * g));
            } else {
                s16x8 w;
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = (short)B[(int64_t)(k0 + 32 * s + 8 * g + j) * a.ldb + n];
                Wf[nt][s] = __builtin_bit_cast(bf16x8, w);
            }
        }
    }
    __shared__ __attribute__((aligned(16))) float part[2][4][16][NF + 4];   // [slab][wave][token][feature (+4: bank spread)]
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 xa[KS], xb[KS];
    const int64_t ngroups = (a.M + 15) / 16, gstride = gridDim.x;
    auto rowptr = [&](int64_t gp) {
        int64_t m = gp * 16 + mi;
        m = m < a.M ? m : a.M - 1;
        return A + m * a.lda + k0 + 8 * g;
    };
    int64_t grp = blockIdx.x;
    {
        const bf16_t* p = rowptr(grp < ngroups ? grp : 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) xa[s] = *reinterpret_cast<const u32x4*>(p + 32 * s);
    }
    // this thread's outputs of a group: token ot, features of .. of + NF / 16 - 1 (2 or 4 consecutive ones)
    constexpr int FPT = NF / 16;
    const int ot = threadIdx.x >> 4, of = FPT * (threadIdx.x & 15);
    float bv[FPT];
#pragma unroll
    for (int j = 0; j < FPT; ++j) bv[j] = a.bias ? a.bias[of + j] : 0.f;
    const float* rsp = a.rscale ? a.rscale : &skinny_one;
    int slab = 0;
    for (; grp < ngroups; grp += gstride) {
        {
            const int64_t nxt = grp + gstride;
            const bf16_t* p = rowptr(nxt < ngroups ? nxt : grp);          // unconditional (the last one re-reads)
#pragma unroll
            for (int s = 0; s < KS; ++s) xb[s] = *reinterpret_cast<const u32x4*>(p + 32 * s);
            SEGF_LOADS_ISSUED();
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[nt][s], __builtin_bit_cast(bf16x8, xa[s]), acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) {
            float* pw = &part[slab][wave][mi][32 * q + 8 * g];
            *reinterpret_cast<float4*>(pw) = make_float4(acc[2 * q][0], acc[2 * q][1], acc[2 * q][2], acc[2 * q][3]);
            *reinterpret_cast<float4*>(pw + 4) = make_float4(acc[2 * q + 1][0], acc[2 * q + 1][1], acc[2 * q + 1][2], acc[2 * q + 1][3]);
        }
        const int64_t mr = grp * 16 + ot;
        const int64_t mc = mr < a.M ? mr : a.M - 1;
        // residual row piece and DropPath scale of this thread's outputs: issued before the barrier, consumed after it
        uint32_t rraw[FPT / 2];
        float rs = 1.f;
        if (R) {
            const bf16_t* rp = R + mc * a.ldr + of;
#pragma unroll
            for (int j = 0; j < FPT / 2; ++j) rraw[j] = *reinterpret_cast<const uint32_t*>(rp + 2 * j);
            rs = rsp[a.rscale ? mc / a.rpg : 0];
        }
        __syncthreads();                 // (the other slab is free again: its readers have all arrived here)
        {
            float v[FPT];
#pragma unroll
            for (int j = 0; j < FPT; ++j) {
                const float p0 = part[slab][0][ot][of + j], p1 = part[slab][1][ot][of + j], p2 = part[slab][2][ot][of + j],
                            p3 = part[slab][3][ot][of + j];
                v[j] = (((p0 + p1) + p2) + p3) + bv[j];
            }
            if (R) {
#pragma unroll
                for (int j = 0; j < FPT / 2; ++j) {
                    v[2 * j] = __uint_as_float(rraw[j] << 16) + rs * v[2 * j];
                    v[2 * j + 1] = __uint_as_float(rraw[j] & 0xffff0000u) + rs * v[2 * j + 1];
                }
            }
            if (mr < a.M) {
                if constexpr (FPT == 2) *reinterpret_cast<uint32_t*>(C + mr * a.ldc + of) = pack2bf(v[0], v[1]);
                else *reinterpret_cast<uint2*>(C + mr * a.ldc + of) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            }
        }
        slab ^= 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) xa[s] = xb[s];
    }
}
static bool gemm_skinny_k_ok(int layout, int64_t M, int64_t N, int64_t K) {
    if (POL(gemm_no_skinny_k)) return false;
    if (N == 64 && K > 768) return false;       // 64 outputs: the weight fragments of K / 4 > 192 columns pass the register file
    return (layout == 0 || layout == 1) && (N == 32 || N == 64) && M >= 65536 && K % 128 == 0 && K >= 256 && K <= 1024;
}
template <int LAYOUT, int NT>
static void gemm_skinny_k_launch(int ks, unsigned gx, hipStream_t st, const GemmArgs& a) {
    switch (ks) {
    case 2: hipLaunchKernelGGL((gemm_skinny_k_kernel<LAYOUT, 2, NT>), dim3(gx), dim3(256), 0, st, a); break;
    case 3: hipLaunchKernelGGL((gemm_skinny_k_kernel<LAYOUT, 3, NT>), dim3(gx), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((gemm_skinny_k_kernel<LAYOUT, 4, NT>), dim3(gx), dim3(256), 0, st, a); break;
    case 5: hipLaunchKernelGGL((gemm_skinny_k_kernel<LAYOUT, 5, NT>), dim3(gx), dim3(256), 0, st, a); break;
    case 6: hipLaunchKernelGGL((gemm_skinny_k_kernel<LAYOUT, 6, NT>), dim3(gx), dim3(256), 0, st, a); break;
    case 7: hipLaunchKernelGGL((gemm_skinny_k_kernel<LAYOUT, 7, NT>), dim3(gx), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((gemm_skinny_k_kernel<LAYOUT, 8, NT>), dim3(gx), dim3(256), 0, st, a); break;
    }
}

static int gemm_skinny_nt(int layout, int64_t M, int64_t N, int64_t K) {
    if (layout > 1 || M < 16384 || (K != 32 && K != 64 && K != 128 && !(K == 160 && layout == 0))) return 0;
    const int ks = (int)(K / 32);
    if (ks == 5) return N == 32 ? 2 : 0;    // K = 160: ONLY the stem (7 x 7 x 3 = 147 im2col columns padded to whole 32-steps, 32 outputs); MiT stage 3's 160-wide products would re-read x once per 32 output columns
    int nt = 16 / ks;                       // NT * KS <= 16 fragments
    if (nt > 8) nt = 8;
    while (nt >= 2 && N % (16 * nt)) nt >>= 1;
    return nt >= 2 ? nt : 0;
}
template <int LAYOUT>
static bool gemm_skinny_launch(int ks, int nt, dim3 grid, hipStream_t st, const GemmArgs& a) {
#define SK(KS_, NT_) if (ks == KS_ && nt == NT_) { hipLaunchKernelGGL((gemm_skinny_kernel<LAYOUT, KS_, NT_>), grid, dim3(256), 0, st, a); return true; }
    SK(1, 2) SK(1, 4) SK(1, 8) SK(2, 2) SK(2, 4) SK(2, 8) SK(4, 2) SK(4, 4) SK(5, 2)
#undef SK
    return false;
}

// ---- streaming weight gradient for small outputs -----------------------------------------------------------------------------
// dW[M,N] = A^T B over K ~ 10^5..10^6 tokens with M, N <= a few hundred (the MiT stage-1/2 linears, the patch embedding) is
// pure streaming of the two token-major operands; the 128^2-tile kernel spends most of each tile on padding (a [32 x 147] output
// ran at 20 % of its HBM time).  Here a WAVE owns a K slice: it walks 32-token groups, stages the two row blocks in its own LDS
// slab (16-byte chunks as they lie in memory; the next group's loads are in flight meanwhile), reads both operands back through
// ds_read_b64_tr_b16 (tokens become the contiguous k of the fragments) and keeps the whole [16 MT x 16 NT] output block in
// MFMA accumulators; no workgroup barrier anywhere.  Every wave writes one split-K slice [z][M][N] (summed by
// splitk_reduce_kernel in fixed order); the bias gradient rides along as an all-ones column at index N.
// grid = (slices / 4, column blocks, row blocks).
__device__ __forceinline__ bf16x8 frag_tok_tr(const unsigned char* tile, int rowbytes, int cb, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned char* a0 = tile + (8 * g + q) * rowbytes + (cb + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * rowbytes));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
template <int MT, int NT, bool CSUM>
__global__ void __launch_bounds__(256) gemm_dw_skinny_kernel(GemmArgs a, int slices) {
    constexpr int RA = 32 * MT + 16, RB = 32 * NT + 16;                  // row bytes of the staged blocks (+16: bank spread)
    constexpr int CA = 2 * MT, CB = 2 * NT;                               // 16-byte chunks per staged row
    __shared__ __attribute__((aligned(16))) unsigned char slab[4][32 * (RA + RB)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int z = blockIdx.x * 4 + wave;
    if (z >= slices) return;
    const int64_t m0 = (int64_t)blockIdx.z * (16 * MT), n0 = (int64_t)blockIdx.y * (16 * NT);
    unsigned char* ta = slab[wave];
    unsigned char* tb = ta + 32 * RA;
    const int64_t k0 = (int64_t)z * a.kchunk;
    const int64_t kend = k0 + a.kchunk < a.K ? k0 + a.kchunk : a.K;      // host: (kend - k0) % 32 == 0
    // Per-lane chunk offsets, fixed for the whole walk (bytes from the first token of a group).  No masks anywhere: columns
    // past M / N (row padding, or a clamped chunk when the block overhangs the row) only ever reach accumulator rows / columns
    // that are never stored, and the host guarantees whole 32-token groups.
    uint32_t oa[MT], ob[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ci = lane + 64 * i, row = ci / CA, c = ci - row * CA;
        const int64_t col = m0 + 8 * c;
        oa[i] = (uint32_t)((row * a.lda + (col + 8 <= a.lda ? col : 0)) * 2);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int ci = lane + 64 * i, row = ci / CB, c = ci - row * CB;
        const int64_t col = n0 + 8 * c;
        ob[i] = (uint32_t)((row * a.ldb + (col + 8 <= a.ldb ? col : 0)) * 2);
    }
    f32x4 acc[MT][NT], accs[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // B fragment of all ones: MFMA(A fragment, ones) = the column sums of dy = the bias gradient, in every output column
    const uint4 ones4 = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const bf16x8 fones = __builtin_bit_cast(bf16x8, ones4);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));       // (native vectors: arrays of HIP's uint4 class stayed in scratch)
    u32x4 ra[MT], rb[NT];
    const char* pa0 = reinterpret_cast<const char*>(a.A);
    const char* pb0 = reinterpret_cast<const char*>(a.B);
    {
        const char* pa_ = pa0 + k0 * a.lda * 2;
        const char* pb_ = pb0 + k0 * a.ldb * 2;
#pragma unroll
        for (int i = 0; i < MT; ++i) ra[i] = *reinterpret_cast<const u32x4*>(pa_ + oa[i]);
#pragma unroll
        for (int i = 0; i < NT; ++i) rb[i] = *reinterpret_cast<const u32x4*>(pb_ + ob[i]);
    }
    for (int64_t kg = k0; kg < kend; kg += 32) {
        // stage the group (the previous group's fragment reads are complete: same-wave LDS operations execute in order)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int ci = lane + 64 * i, row = ci / CA, c = ci - row * CA;
            *reinterpret_cast<u32x4*>(ta + row * RA + 16 * c) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int ci = lane + 64 * i, row = ci / CB, c = ci - row * CB;
            *reinterpret_cast<u32x4*>(tb + row * RB + 16 * c) = rb[i];
        }
        {
            const int64_t kn = kg + 32 < kend ? kg + 32 : kg;        // unconditional: the loads stay countable (the last one is a re-read)
            const char* pa_ = pa0 + kn * a.lda * 2;
            const char* pb_ = pb0 + kn * a.ldb * 2;
#pragma unroll
            for (int i = 0; i < MT; ++i) ra[i] = *reinterpret_cast<const u32x4*>(pa_ + oa[i]);
#pragma unroll
            for (int i = 0; i < NT; ++i) rb[i] = *reinterpret_cast<const u32x4*>(pb_ + ob[i]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bf16x8 fa[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            fa[i] = frag_tok_tr(ta, RA, 16 * i, lane);
            if (CSUM) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fones, accs[i], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bf16x8 fb = frag_tok_tr(tb, RB, 16 * j, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // accumulator (i, j): rows m0 + 16 i + 4 (lane >> 4) + r, column n0 + 16 j + (lane & 15)
    float* wsz = a.ws + (int64_t)z * a.M * a.N;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int64_t n = n0 + 16 * j + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + 16 * i + 4 * (lane >> 4) + r;
                if (m < a.M && n < a.N) wsz[m * a.N + n] = acc[i][j][r];
            }
        }
        if (CSUM && blockIdx.y == 0 && (lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + 16 * i + 4 * (lane >> 4) + r;
                if (m < a.M) a.colsum_ws[(int64_t)z * a.M + m] = accs[i][r];
            }
        }
    }
}

// Shapes the streaming weight-gradient kernel takes and its blocking; slices = waves along K
struct DwSkinny { int mt, nt, rowblocks, colblocks, slices; };
static bool gemm_dw_skinny_plan(int64_t M, int64_t N, int64_t K, bool colsum, DwSkinny& p) {
    if (POL(gemm_no_dw_skinny)) return false;
    if (K < 65536 || M > 256 || N > 288) return false;
    (void)colsum;                                 // the bias gradient costs no output column (all-ones B fragment)
    if (K % 32) return false;                     // whole 32-token groups only (no masking in the kernel)
    const int tm = (int)cdiv64(M, 16), tn = (int)cdiv64(N, 16);
    // row blocks of 2 / 4 / 8 tiles, column blocks of 2 / 4 / 8 / 10 tiles, at most 32 accumulator tiles per wave
    static const int mts[] = {2, 4, 8}, nts[] = {2, 4, 8, 10};
    int64_t best = -1;
    for (int mt : mts)
        for (int nt : nts) {
            if (mt * nt > 32 || (mt == 8 && nt > 2)) continue;
            const int rbk = (tm + mt - 1) / mt, cbk = (tn + nt - 1) / nt;
            // bytes streamed per token: A row blocks are re-read per column block and vice versa (padding included)
            const int64_t cost = (int64_t)cbk * rbk * (16 * mt + 16 * nt);
            if (best < 0 || cost < best) { best = cost; p.mt = mt; p.nt = nt; p.rowblocks = rbk; p.colblocks = cbk; }
        }
    if (best < 0 || p.rowblocks * p.colblocks > 1) return false;       // blocked outputs re-read an operand: measured no better than
                                                                       // the tiled kernel ([64 x 256]: 108 vs 94 us, [256 x 64]: 89 vs 93)
    int64_t sl = K / 512;                                                // >= 16 groups of 32 tokens per wave
    int64_t capw = 2048;
    const int64_t cap = capw / ((int64_t)p.rowblocks * p.colblocks);
    if (sl > cap) sl = cap;
    if (sl < 4) return false;
    // the slice count gemm_impl arrives at from this request (K chunks are whole 64-token steps)
    const int64_t kchunk = cdiv64(cdiv64(K, sl), 64) * 64;
    p.slices = (int)cdiv64(K, kchunk);
    return true;
}
template <int MT>
static bool gemm_dw_skinny_launch_nt(int nt, dim3 grid, hipStream_t st, const GemmArgs& a, int slices) {
#define DS(NT_) if (nt == NT_) { if (a.colsum_ws) hipLaunchKernelGGL((gemm_dw_skinny_kernel<MT, NT_, true>), grid, dim3(256), 0, st, a, slices); \
                            else hipLaunchKernelGGL((gemm_dw_skinny_kernel<MT, NT_, false>), grid, dim3(256), 0, st, a, slices); return true; }
    DS(2)
    if constexpr (MT <= 4) { DS(4) DS(8) }
    if constexpr (MT == 2) { DS(10) }
#undef DS
    return false;
}

// nn.Linear weight gradients that are MATRIX-PIPE work, not streaming: dW [M x N] = dy^T x over K tokens with both feature counts
// multiples of 256, >= 100 GFLOP and >= 256 FLOP per operand byte (ConvNeXtV2-L stage 3 / 4 at 640^2, batch 32: [3072 x 768] over
// 51200 tokens = 241 GFLOP, 27 blocks x 2; MiT-B2 stage 4 at batch 32).  gemm_use_big's token-count rule (K >= 65536, made for the
// SMALL outputs of the MiT linears, where split-K supplies the parallelism) left them on the 128-tile grouped kernel at ~630 TFLOP/s;
// the eight-phase tile runs reduction-major x reduction-major products at ~1.2 PFLOP/s (gemm8.hip).  The bias gradient then takes its
// own column-sum pass (one more read of dy: 0.3 GB against 241 GFLOP).  SEGFAC_GEMM8_DW=0 switches the rule off.
static inline bool dw_on_gemm8(int64_t M, int64_t N, int64_t K) {
    if (!POL(gemm8_dw) || POL(no_gemm8) || POL(no_gemm8t) || POL(gemm_no_big) || !POL(gemm8_linear)) return false;
    if (M % 256 || N % 256 || K % 64 || K < 2048) return false;
    const double gflop = 2e-9 * (double)M * (double)N * (double)K;
    return gflop >= (double)POL(gemm8_dw_min_gflop) && M * N >= 256 * (M + N);
}
extern "C" int segf_gemm_pick_splitk(int64_t M, int64_t N, int64_t K) {
    // layout 2 (weight gradient): K = token count.  Aim for >= 512 workgroups, >= 4 K-steps per slice.
    {   // small outputs: the streaming kernel's slice count (one slice per wave); the bias-gradient column is assumed
        DwSkinny p;
        if (gemm_dw_skinny_plan(M, N, K, true, p)) return p.slices;
    }
    if (!gemm_use_big(2, M, N, K) && dw_on_gemm8(M, N, K)) {
        // one 256 x 256 tile per CU and round: the smallest slice count (slices of >= 16 K tiles) whose last round is >= 90 % full, else
        // the fullest ([6144 x 1536] = 144 tiles: 5 slices = 720 workgroups = 2.8 rounds; [3072 x 768] = 36 tiles: 7 slices = 252)
        const int64_t tiles = (M / 256) * (N / 256);
        int best = 1;
        double bf = 0.0;
        for (int c = 1; c <= 16 && K / c >= 16 * 64; ++c) {
            const int64_t wg = tiles * c;
            const double fill = (double)wg / (double)(cdiv64(wg, 256) * 256);
            if (fill >= 0.9) { best = c; break; }
            if (fill > bf + 1e-9) { bf = fill; best = c; }
        }
        return best;
    }
    const bool big = gemm_use_big(2, M, N, K);
    const int64_t tiles = big ? cdiv64(M, GG_B) * cdiv64(N, GG_B) : cdiv64(M, GB_BM) * cdiv64(N, GB_BN);
    // one wave of workgroups: 256 CUs x (1 big-tile | 2 small-tile) resident workgroups.  Rounded DOWN: 3 tiles x 86 slices =
    // 258 workgroups would run as two rounds (256 + 2) and take twice as long as 3 x 85
    const int64_t res = big ? 256 : 512;                          // resident workgroups
    int64_t s = res / tiles;
    if (tiles > res) {
        // more tiles than one round of workgroups (UPerHead's 3072 -> 768 3x3 conv: 3 x 108 = 324 tiles of 256^2 run as 256 + 68,
        // 63 % of the machine on average: 565 vs 787 TFLOP/s measured against the forward of the same shape): the smallest slice
        // count whose last round is >= 93 % full (324 x 3 = 972 workgroups = 3.8 rounds; the extra reduce pass is ~0.1 ms of 39)
        s = 1;
        for (int64_t c = 1; c <= 8; ++c) {
            const int64_t wg = tiles * c;
            if ((double)wg / (double)(cdiv64(wg, res) * res) >= 0.93) { s = c; break; }
        }
    }
    const int64_t maxs = K / (4 * GB_BK);
    if (s > maxs) s = maxs;
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    return (int)s;
}

struct GemmPro { const float* scale; const float* shift; int64_t rpg; int64_t ld; int act; };
// gemm8.hip: the eight-phase 256 x 256 tile (LDS-DMA staging, counted waits).  kind 0 / 1 / 2 = layout 0 / 1 / 2; conv = implicit 3x3
int gemm8_supported(int kind, int conv, int64_t M, int64_t N, int64_t K, int64_t kchunk, int cC);
int gemm8_linear_ok(int64_t M, int64_t N, int64_t K, int64_t kchunk);
int gemm8_launch(int kind, int conv, int fp8, int64_t M, int64_t N, int64_t K, int64_t kchunk, int split_k, const void* A, int64_t lda,
                 const void* B, int64_t ldb, void* C, int64_t ldc, int cH, int cW, int cC, int csign, const float* f8_sa,
                 const float* f8_sb, const float* bias, const void* residual, int64_t ldr, const float* rscale, int64_t rpg, float* ws,
                 hipStream_t st);
static int gemm_impl(int dt, int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                     int64_t ldb, void* C, int c_dt, int64_t ldc, const float* bias, const void* residual, int64_t ldr,
                     const float* rscale, int64_t rows_per_group, int split_k, float* ws, float* colsum, void* stream,
                     const GemmPro* pro = nullptr);

extern "C" int segf_gemm(int dt, int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                         int64_t ldb, void* C, int c_dt, int64_t ldc, const float* bias, const void* residual, int64_t ldr,
                         const float* rscale, int64_t rows_per_group, int split_k, float* ws, void* stream) {
    return gemm_impl(dt, layout, M, N, K, A, lda, B, ldb, C, c_dt, ldc, bias, residual, ldr, rscale, rows_per_group, split_k, ws,
                     nullptr, stream);
}

extern "C" int segf_colsum(int dt, const void* x, int64_t ldx, int64_t rows, int64_t cols, float* out, float* ws, void* stream);
extern "C" int64_t segf_colsum_ws(int64_t rows, int64_t cols);

// weight gradient + bias gradient in one pass over dy:  C[M,N] = A^T B (layout 2),  dbias[m] = sum_k A(k, m)
extern "C" int64_t segf_gemm_dw_db_ws(int64_t M, int64_t N, int64_t K, int split_k) {
    if (split_k < 1) split_k = 1;
    const int64_t g = split_k > 1 ? (int64_t)split_k * M * N : 0;
    const int64_t c1 = (int64_t)split_k * M, c2 = segf_colsum_ws(K, M);
    return g + (c1 > c2 ? c1 : c2);
}
extern "C" int segf_gemm_dw_db(int dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                               void* C, int c_dt, int64_t ldc, int split_k, float* ws, float* dbias, void* stream) {
    if (!dbias || !ws) return SEGF_ERR_WORKSPACE;
    if (split_k < 1) split_k = 1;
    DwSkinny sk;
    const bool skinny = dt == SEGF_BF16 && c_dt == SEGF_F32 && gemm_dw_skinny_plan(M, N, K, true, sk) && sk.slices == split_k &&
                        (uintptr_t)A % 16 == 0 && (lda * 2) % 16 == 0 && (uintptr_t)B % 16 == 0 && (ldb * 2) % 16 == 0 &&
                        !POL(gemm_no_tr);
    // fused in the streaming kernel (all-ones fragment), in the 128-tile kernel (all-ones column when N leaves one free, extra
    // MFMAs when it does not); the 256-tile kernel has no registers to spare for it
    const bool fused = dt == SEGF_BF16 && c_dt == SEGF_F32 && (skinny || !(gemm_use_big(2, M, N, K) || dw_on_gemm8(M, N, K))) &&
                       !POL(gemm_no_fused_db);
    if (!fused) {       // big-tile / fp32 kernels: separate column reduction (still one C-ABI call)
        const int rc = gemm_impl(dt, 2, M, N, K, A, lda, B, ldb, C, c_dt, ldc, nullptr, nullptr, 0, nullptr, 1, split_k, ws, nullptr, stream);
        if (rc) return rc;
        return segf_colsum(dt, A, lda, K, M, dbias, ws + (split_k > 1 ? (int64_t)split_k * M * N : 0), stream);
    }
    return gemm_impl(dt, 2, M, N, K, A, lda, B, ldb, C, c_dt, ldc, nullptr, nullptr, 0, nullptr, 1, split_k, ws, dbias, stream);
}

// The weight + bias gradients of SEVERAL nn.Linear layers in one call: items that take the 128-tile split-K kernel with the fused bias
// column (what segf_gemm_dw_db launches for them) are gathered into grouped launches of up to GDW_MAX members -- one product launch and
// one reduce launch per group instead of two launches per layer; every other item is executed by segf_gemm_dw_db itself.  Each item's
// result is bitwise what segf_gemm_dw_db computes for it.
// largest output (elements) of a member that alone would take the 256-tile kernel and still joins a group (measured flat between 256 K
// and 4 M elements)
static inline int64_t dw_group_big_max() { return 1024 * 1024; }
extern "C" int segf_gemm_dw_db_grouped(int dt, int n, const SegfDwItem* items, void* stream) {
    if (n <= 0) return 0;
    if (!items) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    GemmDwGroup g; ReduceGroup r;
    int pend[GDW_MAX], npend = 0;                 // indices of the groupable items of the open group
    const bool no_group = POL(no_grouped_dw) != 0;
    const bool no_shared = POL(dw_no_shared_split) != 0;
    auto flush = [&]() -> int {
        if (npend == 0) return 0;
        // Slice counts.  Each item arrives with the count segf_gemm_pick_splitk gives a product that runs ALONE (enough slices to fill the
        // chip by itself); in a group the members fill it together, and a member whose `shared_split` flag is set lets the library lower
        // its count: all flagged members then take one common K range per slice, the smallest that brings the group down to ~3 rounds of
        // resident workgroups (or keeps it where it is).  At batch 16 the stage-3 / 4 layers ran 64 slices of FOUR K steps each -- a
        // 64 KB partial tile written and summed per 8 MFLOP; with the shared range they run 8 - 16 slices.
        int64_t kslice[GDW_MAX];
        bool any_flag = false;
        for (int j = 0; j < npend; ++j) {
            const SegfDwItem& it = items[pend[j]];
            const int split_k = it.split_k < 1 ? 1 : it.split_k;
            kslice[j] = cdiv64(cdiv64(it.K, split_k), GB_BK) * GB_BK;
            any_flag |= it.shared_split != 0;
        }
        if (any_flag && !no_shared && npend > 1) {
            auto total_wg = [&](int64_t kap) {
                int64_t t = 0;
                for (int j = 0; j < npend; ++j) {
                    const SegfDwItem& it = items[pend[j]];
                    const int64_t ks = it.shared_split ? (kap > kslice[j] ? kap : kslice[j]) : kslice[j];
                    t += cdiv64(it.M, GB_BM) * cdiv64(it.N, GB_BN) * cdiv64(it.K, ks);
                }
                return t;
            };
            const int64_t target = 1536;          // 3 rounds of 2 resident workgroups on 256 CUs
            int64_t kap = 4 * GB_BK;
            // (at most 32 K steps per slice: with long slices the members' different tile shapes leave an uneven last round -- batch 128
            // measured -0.4 % without this cap, and there the per-layer counts already give slices of 16 steps)
            while (total_wg(kap) > target && kap < 32 * GB_BK) kap += GB_BK;
            for (int j = 0; j < npend; ++j) {
                const SegfDwItem& it = items[pend[j]];
                if (it.shared_split && kap > kslice[j] && cdiv64(it.K, kap) >= 2) kslice[j] = kap;      // (never below two slices: the reduce pass carries the output)
            }
        }
        g.n = 0; r.n = 0; g.start[0] = 0; r.start[0] = 0;
        for (int j = 0; j < npend; ++j) {
            const SegfDwItem& it = items[pend[j]];
            const int64_t M = it.M, N = it.N, K = it.K, kchunk = kslice[j];
            const int slices = (int)cdiv64(K, kchunk);
            // the arguments gemm_impl builds for this product (layout 2, fp32 output, split-K partials in ws, bias column riding)
            GemmArgs a;
            a.A = it.dy; a.B = it.x; a.C = it.dw; a.bias = nullptr; a.residual = nullptr; a.rscale = nullptr;
            a.M = M; a.N = N; a.K = K; a.lda = it.lddy; a.ldb = it.ldx; a.ldc = it.lddw; a.ldr = 0; a.rpg = 1;
            a.kchunk = kchunk; a.ws = it.ws;
            a.a_vec = 1; a.b_vec = 1; a.fast = 1; a.xcd_slabs = POL(dw_no_xcd_slabs) ? 0 : 1;
            a.c_vec = ((uintptr_t)it.dw % 16 == 0) && ((it.lddw * 4) % 16 == 0);
            a.r_vec = 0; a.c_vec16 = a.c_vec;
            a.use_tr = 1;
            a.cH = a.cW = a.cC = 0; a.csign = 1;
            a.colsum = it.db; a.colsum_ws = it.ws + (int64_t)slices * M * N;
            a.pro_scale = nullptr; a.pro_shift = nullptr; a.pro_rpg = 1; a.pro_ld = 0; a.pro_act = 0;
            a.f8_sa = nullptr; a.f8_sb = nullptr;
            const unsigned gx = (unsigned)cdiv64(N, GB_BN), gy = (unsigned)cdiv64(M, GB_BM), gz = (unsigned)slices;
            const int k = g.n;
            g.m[k] = a; g.gx[k] = gx; g.gy[k] = gy; g.gz[k] = gz;
            g.start[k + 1] = g.start[k] + gx * gy * gz;
            ++g.n;
            const int form = splitk_reduce_form(it.ws, M, N, it.dw, it.lddw);
            const unsigned blocks = splitk_reduce_main_blocks(form, M * N), csb = splitk_reduce_cs_blocks(form, M);
            r.wide[k] = form; r.split[k] = slices; r.ws[k] = it.ws; r.C[k] = it.dw; r.cs_ws[k] = a.colsum_ws; r.cs_out[k] = it.db;
            r.M[k] = M; r.N[k] = N; r.ldc[k] = it.lddw; r.cs_n[k] = M; r.main_blocks[k] = blocks;
            r.start[k + 1] = r.start[k] + blocks + csb;
            ++r.n;
        }
        npend = 0;
        if (g.n == 1) {          // a lone member: the ordinary launch pair (same arithmetic)
            const GemmArgs& a = g.m[0];
            hipLaunchKernelGGL((gemm_bf16_kernel<2, float, true, false, 2, true>), dim3(g.gx[0], g.gy[0], g.gz[0]), dim3(256), 0, st, a);
        } else {
            hipLaunchKernelGGL((gemm_bf16_dw_group_kernel<true>), dim3(g.start[g.n]), dim3(256), 0, st, g);
        }
        SEGF_CHECK_LAUNCH();
        if (r.n > 0) {
            hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(r.start[r.n]), dim3(256), 0, st, r);
            SEGF_CHECK_LAUNCH();
        }
        return 0;
    };
    ReduceGroup r2;
    r2.n = 0; r2.start[0] = 0;
    auto flush2 = [&]() -> int {
        if (r2.n == 0) return 0;
        hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3(r2.start[r2.n]), dim3(256), 0, st, r2);
        SEGF_CHECK_LAUNCH();
        r2.n = 0;
        return 0;
    };
    for (int i = 0; i < n; ++i) {
        const SegfDwItem& it = items[i];
        int split_k = it.split_k < 1 ? 1 : it.split_k;
        const int64_t M = it.M, N = it.N, K = it.K;
        DwSkinny sk;
        const bool aligned = (uintptr_t)it.dy % 16 == 0 && (it.lddy * 2) % 16 == 0 && (uintptr_t)it.x % 16 == 0 && (it.ldx * 2) % 16 == 0;
        const bool skinny = gemm_dw_skinny_plan(M, N, K, true, sk) && sk.slices == split_k && aligned;
        const int64_t kchunk = cdiv64(cdiv64(K, split_k), GB_BK) * GB_BK;
        const int slices = (int)cdiv64(K > 0 ? K : 1, kchunk > 0 ? kchunk : GB_BK);
        const bool groupable = !no_group && dt == SEGF_BF16 && M > 0 && N > 0 && K > 0 && it.dw && it.db && it.ws && !skinny && aligned &&
                               (!gemm_use_big(2, M, N, K) || (it.shared_split && M * N <= dw_group_big_max())) && !dw_on_gemm8(M, N, K) &&
                               !POL(gemm_no_fused_db) && !POL(gemm_no_fastload) &&
                               !POL(gemm_no_tr) && !POL(gemm_no_deep128) && M % 8 == 0 && N % 8 == 0 && slices > 1 && cdiv64(M, GB_BM) <= 65535;
        // (a member that alone would take the 256-tile kernel + a separate column-sum pass -- the stage-3 / 4 layers at batch 128 -- joins the
        // group too when it lets the library choose its split: one pass over dy for both gradients; batch 128 +0.5 %)
        if (!groupable) {                        // (the items are independent of each other: no need to close the open group)
            // its product launches now; its reduce pass joins the others' in r2 (issued when full and at the end of the call)
            g_reduce_sink = no_group ? nullptr : &r2;
            const int rc = segf_gemm_dw_db(dt, M, N, K, it.dy, it.lddy, it.x, it.ldx, it.dw, SEGF_F32, it.lddw, split_k, it.ws, it.db, stream);
            g_reduce_sink = nullptr;
            if (rc) return rc;
            if (r2.n == GDW_MAX) { const int rc2 = flush2(); if (rc2) return rc2; }
            continue;
        }
        pend[npend++] = i;
        if (npend == GDW_MAX) { const int rc = flush(); if (rc) return rc; }
    }
    { const int rc = flush(); if (rc) return rc; }
    return flush2();
}

// Product whose activation operand is normalised on the way in (BatchNorm + ReLU + Dropout2d scale of ConvModule, heads/
// segformer.py:21-29,40, folded into per-(sample, channel) scale / shift tables): layout 0 = y = act(x s + t) W^T,
// layout 2 = dW = dy^T act(x s + t).  Only the 256-tile kernel implements it: ask segf_gemm_pro_supported first.
extern "C" int segf_gemm_pro_supported(int dt, int layout, int64_t M, int64_t N, int64_t K, int64_t rows_per_group) {
    if (dt != SEGF_BF16 || (layout != 0 && layout != 2) || rows_per_group <= 0) return 0;
    // (the 256-tile kernel is used whenever the prologue is requested, also for outputs narrower than its usual threshold: the
    // 19-class heads of the Cityscapes configurations run its narrow wave shapes with some waves idle, which a product that
    // streams a [tokens x 768] operand does not notice)
    if ((layout == 0 ? M : K) < 4096 || K % GB_BK || (layout == 2 && K < 32768)) return 0;
    if (POL(gemm_no_tr) || POL(gemm_no_pro)) return 0;
    if (layout == 0) return (rows_per_group % GG_B == 0 && K % 8 == 0) ? 1 : 0;
    return (rows_per_group % GB_BK == 0 && N % 8 == 0) ? 1 : 0;
}
extern "C" int segf_gemm_pro(int dt, int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                             int64_t ldb, void* C, int c_dt, int64_t ldc, const float* bias, int split_k, float* ws,
                             const float* pro_scale, const float* pro_shift, int64_t rows_per_group, int act, void* stream) {
    if (!pro_scale || !pro_shift || !segf_gemm_pro_supported(dt, layout, M, N, K, rows_per_group)) return SEGF_ERR_SHAPE;
    if ((uintptr_t)pro_scale % 16 || (uintptr_t)pro_shift % 16) return SEGF_ERR_SHAPE;
    if ((layout == 0) != (c_dt == SEGF_BF16)) return SEGF_ERR_DTYPE;       // forward writes bf16, weight gradient fp32
    const GemmPro pro{pro_scale, pro_shift, rows_per_group, layout == 0 ? K : N, act};
    return gemm_impl(dt, layout, M, N, K, A, lda, B, ldb, C, c_dt, ldc, bias, nullptr, 0, nullptr, 1, split_k, ws, nullptr, stream, &pro);
}

static int gemm_impl(int dt, int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B,
                     int64_t ldb, void* C, int c_dt, int64_t ldc, const float* bias, const void* residual, int64_t ldr,
                     const float* rscale, int64_t rows_per_group, int split_k, float* ws, float* colsum, void* stream,
                     const GemmPro* pro) {
    if (M <= 0 || N <= 0) return 0;
    if (K < 0 || layout < 0 || layout > 2) return SEGF_ERR_SHAPE;
    if (dt != SEGF_F32 && dt != SEGF_BF16) return SEGF_ERR_DTYPE;
    if (c_dt != SEGF_F32 && c_dt != SEGF_BF16) return SEGF_ERR_DTYPE;
    if (dt == SEGF_F32 && c_dt != SEGF_F32) return SEGF_ERR_DTYPE;
    if (split_k < 1) split_k = 1;
    if (split_k > 1 && (bias || residual)) return SEGF_ERR_SHAPE;
    if (split_k > 1 && !ws) return SEGF_ERR_WORKSPACE;
    if (rscale && rows_per_group <= 0) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.bias = bias; a.residual = residual; a.rscale = rscale;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.rpg = rows_per_group > 0 ? rows_per_group : 1;
    const int64_t kstep = dt == SEGF_BF16 ? GB_BK : GF_BK;
    int64_t kchunk = cdiv64(cdiv64(K, split_k), kstep) * kstep;
    if (kchunk <= 0) kchunk = kstep;
    split_k = (int)cdiv64(K > 0 ? K : 1, kchunk);
    a.kchunk = kchunk;
    a.ws = split_k > 1 ? ws : nullptr;
    const size_t esz = dt == SEGF_BF16 ? 2 : 4, csz = c_dt == SEGF_BF16 ? 2 : 4;
    a.a_vec = ((uintptr_t)A % 16 == 0) && ((lda * esz) % 16 == 0);
    a.b_vec = ((uintptr_t)B % 16 == 0) && ((ldb * esz) % 16 == 0);
    a.fast = POL(gemm_no_fastload) ? 0 : 1;
    a.xcd_slabs = (layout == 2 && split_k > 1 && !POL(dw_no_xcd_slabs)) ? 1 : 0;
    // bit 0: vector loads allowed, bit 1: guard-free full-K-step loads.  Layouts 0 / 1 gain 25-45 % from the latter; the split-K
    // weight-gradient launches (layout 2) measured 8-15 % SLOWER with every load in flight, so they keep the guarded loads
    if (a.fast && layout != 2) { a.a_vec *= 3; a.b_vec *= 3; }
    a.c_vec = ((uintptr_t)C % (4 * csz) == 0) && ((ldc * csz) % (4 * csz) == 0);
    a.r_vec = residual ? (((uintptr_t)residual % 16 == 0) && ((ldr * esz) % 16 == 0)) : 0;
    a.c_vec16 = ((uintptr_t)C % 16 == 0) && ((ldc * csz) % 16 == 0);
    a.cH = a.cW = a.cC = 0; a.csign = 1;
    a.colsum = colsum;
    a.colsum_ws = (colsum && split_k > 1) ? ws + (int64_t)split_k * M * N : nullptr;
    a.pro_scale = pro ? pro->scale : nullptr; a.pro_shift = pro ? pro->shift : nullptr;
    a.pro_rpg = pro ? pro->rpg : 1; a.pro_ld = pro ? pro->ld : 0; a.pro_act = pro ? pro->act : 0;
    a.f8_sa = nullptr; a.f8_sb = nullptr;
    a.use_tr = POL(gemm_no_tr) ? 0 : 1;      // debugging switch: transposed fragments by scalar LDS loads instead of ds_read_b64_tr_b16
    if (dt == SEGF_BF16) {
        if (!pro && c_dt == SEGF_BF16 && split_k == 1 && a.a_vec && a.c_vec16 && (!residual || a.r_vec) &&
            (layout == 1 || a.b_vec) && !POL(gemm_no_skinny)) {
            if (layout == 0 && !residual && K == 32 && N == 768 && M >= 65536 && a.b_vec && !POL(gemm_no_skinny_rows)) {
                const int64_t groups = cdiv64(M, 16);
                int64_t gx = cdiv64(groups, 4 * 8);
                if (gx > 1024) gx = 1024;
                hipLaunchKernelGGL((gemm_skinny_rows_kernel<1, 6>), dim3((unsigned)gx), dim3(256), 0, st, a);
                SEGF_CHECK_LAUNCH();
                return 0;
            }
            if (gemm_skinny_k_ok(layout, M, N, K) && ldc % 4 == 0 && (!residual || ldr % 2 == 0)) {
                const int64_t groups = cdiv64(M, 16);
                const unsigned gx = (unsigned)imin64(groups, 256 * 12);     // several rounds of 16-token groups per workgroup
                const int ks = (int)(K / 128);
                if (layout == 0) { if (N == 32) gemm_skinny_k_launch<0, 2>(ks, gx, st, a); else gemm_skinny_k_launch<0, 4>(ks, gx, st, a); }
                else { if (N == 32) gemm_skinny_k_launch<1, 2>(ks, gx, st, a); else gemm_skinny_k_launch<1, 4>(ks, gx, st, a); }
                SEGF_CHECK_LAUNCH();
                return 0;
            }
            const int nt = gemm_skinny_nt(layout, M, N, K);
            if (nt) {
                const int64_t groups = cdiv64(M, 16);
                int64_t gx = cdiv64(groups, 4 * 4);           // >= 4 token groups per wave
                if (gx > 2048) gx = 2048;
                const dim3 grid((unsigned)gx, (unsigned)(N / (16 * nt)));
                const bool ok = layout == 0 ? gemm_skinny_launch<0>((int)(K / 32), nt, grid, st, a)
                                            : gemm_skinny_launch<1>((int)(K / 32), nt, grid, st, a);
                if (ok) { SEGF_CHECK_LAUNCH(); return 0; }
            }
        }
        if (layout == 2 && !pro && a.ws && c_dt == SEGF_F32 && a.use_tr && (a.a_vec & 1) && (a.b_vec & 1) && !bias && !residual) {
            DwSkinny p;
            // the caller sized ws for split_k slices (segf_gemm_pick_splitk gives this kernel's count): take it only then
            if (gemm_dw_skinny_plan(M, N, K, true, p) && p.slices == split_k && kchunk % 32 == 0 && lda * 2 * 32 < (1ll << 31) && ldb * 2 * 32 < (1ll << 31)) {
                if (!colsum) { a.colsum = nullptr; a.colsum_ws = nullptr; }
                const dim3 gridk((unsigned)((p.slices + 3) / 4), (unsigned)p.colblocks, (unsigned)p.rowblocks);
                const bool ok = p.mt == 2 ? gemm_dw_skinny_launch_nt<2>(p.nt, gridk, st, a, p.slices)
                              : p.mt == 4 ? gemm_dw_skinny_launch_nt<4>(p.nt, gridk, st, a, p.slices)
                                          : gemm_dw_skinny_launch_nt<8>(p.nt, gridk, st, a, p.slices);
                if (ok) { SEGF_CHECK_LAUNCH(); goto reduce; }
            }
        }
        // the eight-phase tile (gemm8.hip) for nn.Linear products with plain epilogues.  r03 measured it as a loss on every BASELINE model
        // (12-load prologue, drain, direct 8-byte stores against 4 .. 48 K tiles); after the read-section diet of r05 it wins wherever it
        // has >= 192 tiles to run, from K = 256 on, ragged last tiles included (tools/probe/linear8_probe.py, same process, us:
        // [12800 x 3072] K = 768 102 -> 84; [12800 x 768] K = 3072 (150 tiles) 98 -> 69; [51200 x 384] K = 1536 (75 % of the launched tiles
        // are output) 108 -> 77; [131072 x 320] K = 1280 (62.5 %) 219 -> 152; [131072 x 1280] K = 320 220 -> 197; [3200 x 6144] K = 1536
        // 106 -> 80; [32768 x 512] K = 2048 74 -> 57 = 1.21 PFLOP/s).  Not taken: fewer than 128 tiles (78 .. 96 tiles: -2 .. +6 %), and
        // 128 .. 191 tiles unless K >= 2048.  SEGFAC_GEMM8_LINEAR=0 switches it off (cfg5 101.8 -> 98.9 images/s, same box).
        if (!pro && a.use_tr && !colsum && (a.a_vec & 1) && (a.b_vec & 1) && POL(gemm8_linear) && M > 128 && N > 128) {
            const bool f32o8 = c_dt == SEGF_F32 || a.ws;
            // whole-tile count the launch pays for, and the share of it that is output (ragged last tiles: [51200 x 384] = 75 %)
            const int64_t tiles8 = cdiv64(M, 256) * cdiv64(N, 256);
            const bool dense = 100 * M * N >= (int64_t)POL(gemm8_linear_min_fill) * tiles8 * 65536;
            // ... and only where the gain pays for what a launch of this kernel costs the kernels AFTER it (in-situ traces of the cfg2 / cfg4
            // step, tools/probe/trace_ab.sh: the launches that followed an eight-phase launch ran 3 - 10 % longer -- the chip gives clock back
            // after the dense MFMA burst -- ~10 us per launch summed over the step): the gain is ~20 - 30 % of the product's time, i.e. worth it
            // from ~36 GFLOP on (K >= 512) and, for the 4 - 7 K tiles of a shorter reduction, from ~100 GFLOP on.
            const double gflop = 2e-9 * (double)M * (double)N * (double)K;
            const bool worth = gflop >= (K >= 512 ? (double)POL(gemm8_linear_min_gflop) : 100.0);
            const bool fits = K >= POL(gemm8_linear_min_k) && dense && worth && (tiles8 >= 192 || (tiles8 >= POL(gemm8_linear_min_tiles) && K >= 2048));
            if (layout != 2 && fits && !f32o8 && split_k == 1 && gemm8_linear_ok(M, N, K, a.kchunk) && (!residual || a.r_vec)) {
                const int rc8 = gemm8_launch(layout, 0, 0, M, N, K, a.kchunk, 1, A, lda, B, ldb, C, ldc, 0, 0, 0, 1, nullptr, nullptr, bias, residual,
                                             ldr, rscale, a.rpg, nullptr, st);
                if (rc8 != SEGF_ERR_SHAPE) return rc8;
            }
            if (layout == 2 && (gemm_use_big(layout, M, N, K) || dw_on_gemm8(M, N, K)) && c_dt == SEGF_F32 && !bias && !residual &&
                gemm8_supported(2, 0, M, N, K, a.kchunk, 0)) {
                const int rc8 = gemm8_launch(2, 0, 0, M, N, K, a.kchunk, split_k, A, lda, B, ldb, C, ldc, 0, 0, 0, 1, nullptr, nullptr, nullptr,
                                             nullptr, 0, nullptr, 1, a.ws, st);
                if (rc8 != SEGF_ERR_SHAPE) { if (rc8) return rc8; goto reduce; }
            }
        }
        // short reductions (K <= 704: the MiT stage-3 / 4 linears at 160 / 640 / 256) stay on the 128-tile kernel: with a handful of K
        // steps the 256-tile kernel's operand reuse buys nothing (the product is bound by its output) and its one workgroup per CU
        // exposes every load -> LDS -> MFMA round trip; two workgroups per CU with two K steps in flight: cfg2 +0.4 %, batch 16
        // +0.7 %, cfg4 +0.2 % (same box).  The narrow shapes (N <= 160) keep their one-tile kernel; the implicit-GEMM convolutions
        // (other entry points) are not concerned.
        const int smallk = 704;
        const bool short_k = layout != 2 && K <= smallk && N > 160 && !pro;
        if (((gemm_use_big(layout, M, N, K) && !short_k) || pro) && a.use_tr) {
            dim3 gridb((unsigned)cdiv64(N, GG_B), (unsigned)cdiv64(M, GG_B), (unsigned)split_k);
            if (gridb.y > 65535u) return SEGF_ERR_SHAPE;
            const bool f32o = c_dt == SEGF_F32 || a.ws;
#define LAUNCH_G(L)                                                                                              \
    do {                                                                                                         \
        if (f32o) hipLaunchKernelGGL((gemm_bf16_big_kernel<L, float, false>), gridb, dim3(GG_THREADS), 0, st, a);  \
        else hipLaunchKernelGGL((gemm_bf16_big_kernel<L, bf16_t, false>), gridb, dim3(GG_THREADS), 0, st, a);      \
    } while (0)
            // partly filled workgroup tiles: the narrow wave shapes (see the kernel's header)
            const bool narrow_n = layout == 0 && !f32o && N <= 160 && !POL(gemm_no_narrow);
            const bool narrow_n1 = layout == 1 && !f32o && !a.pro_scale && N <= 160 && !POL(gemm_no_narrow);
            const bool narrow_m = layout == 2 && f32o && M <= 160 && !POL(gemm_no_narrow);
            const bool deep = layout == 0 && split_k == 1 && (a.a_vec & 2) && (a.b_vec & 2) && K % GB_BK == 0 && K >= 2 * GB_BK &&
                              (!a.pro_scale || a.pro_ld <= 1024) && !POL(gemm_no_deep);
            if (a.pro_scale) {
                if (layout == 0 && !f32o) {
                    if (narrow_n && deep) hipLaunchKernelGGL((gemm_bf16_big_kernel<0, bf16_t, false, true, 1, true>), gridb, dim3(GG_THREADS), 0, st, a);
                    else if (narrow_n) hipLaunchKernelGGL((gemm_bf16_big_kernel<0, bf16_t, false, true, 1>), gridb, dim3(GG_THREADS), 0, st, a);
                    else hipLaunchKernelGGL((gemm_bf16_big_kernel<0, bf16_t, false, true>), gridb, dim3(GG_THREADS), 0, st, a);
                } else if (layout == 2 && f32o) {
                    if (narrow_m) hipLaunchKernelGGL((gemm_bf16_big_kernel<2, float, false, true, 2>), gridb, dim3(GG_THREADS), 0, st, a);
                    else hipLaunchKernelGGL((gemm_bf16_big_kernel<2, float, false, true>), gridb, dim3(GG_THREADS), 0, st, a);
                } else return SEGF_ERR_SHAPE;
            } else if (narrow_n && deep) hipLaunchKernelGGL((gemm_bf16_big_kernel<0, bf16_t, false, false, 1, true>), gridb, dim3(GG_THREADS), 0, st, a);
            else if (narrow_n) hipLaunchKernelGGL((gemm_bf16_big_kernel<0, bf16_t, false, false, 1>), gridb, dim3(GG_THREADS), 0, st, a);
            else if (narrow_m) hipLaunchKernelGGL((gemm_bf16_big_kernel<2, float, false, false, 2>), gridb, dim3(GG_THREADS), 0, st, a);
            else if (narrow_n1) hipLaunchKernelGGL((gemm_bf16_big_kernel<1, bf16_t, false, false, 1>), gridb, dim3(GG_THREADS), 0, st, a);
            else
            if (layout == 0) LAUNCH_G(0); else if (layout == 1) LAUNCH_G(1); else LAUNCH_G(2);
#undef LAUNCH_G
            SEGF_CHECK_LAUNCH();
            goto reduce;
        }
        if (pro) return SEGF_ERR_SHAPE;            // only the 256-tile kernel applies operand prologues
        dim3 grid((unsigned)cdiv64(N, GB_BN), (unsigned)cdiv64(M, GB_BM), (unsigned)split_k);
        if (grid.y > 65535u) return SEGF_ERR_SHAPE;
        const bool one_step = kchunk <= GB_BK && layout != 2;     // single K step: the 34 KB single-buffer variant
        // two K steps in flight (branch-free loaders) whenever the operands are vectorisable and there are several steps.  Measured
        // (same box, on / off): batch 4 836 / 814 img/s, 16: 2249 / 2211, 32: 3054 / 3014, 128: neutral; cfg5 86.0 / 84.2; the
        // split-K weight gradients (layout 2) add +1 % at batch 4 and are neutral elsewhere
        const bool vec_ab = a.a_vec && a.b_vec;
        const bool deep128 = !one_step && a.use_tr && a.fast && vec_ab && !POL(gemm_no_deep128) &&
                             (layout == 2 ? (M % 8 == 0 && N % 8 == 0)
                                          : (K % 8 == 0 && (layout == 0 || N % 8 == 0) && split_k == 1));
#define LAUNCH_B(L, OT)                                                                                      \
    do {                                                                                                     \
        if (one_step && L != 2) hipLaunchKernelGGL((gemm_bf16_kernel<(L == 2 ? 0 : L), OT, true, false, 1>), grid, dim3(256), 0, st, a); \
        else if (deep128) hipLaunchKernelGGL((gemm_bf16_kernel<L, OT, true, false, 2, true>), grid, dim3(256), 0, st, a); \
        else if (L == 0 || a.use_tr) hipLaunchKernelGGL((gemm_bf16_kernel<L, OT, true>), grid, dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((gemm_bf16_kernel<L, OT, false>), grid, dim3(256), 0, st, a);                 \
    } while (0)
        if (c_dt == SEGF_F32 || a.ws) {
            if (layout == 0) LAUNCH_B(0, float); else if (layout == 1) LAUNCH_B(1, float); else LAUNCH_B(2, float);
        } else {
            if (layout == 0) LAUNCH_B(0, bf16_t); else if (layout == 1) LAUNCH_B(1, bf16_t); else LAUNCH_B(2, bf16_t);
        }
#undef LAUNCH_B
    } else {
        dim3 grid((unsigned)cdiv64(N, GF_BN), (unsigned)cdiv64(M, GF_BM), (unsigned)split_k);
        if (grid.y > 65535u) return SEGF_ERR_SHAPE;
        if (!POL(gemm_f32_no_mfma)) {               // exact fp32 on the matrix pipe
            // 128 x 128 tiles when they still give every CU a workgroup (or the output is so large that operand traffic decides); a
            // narrow output (N <= 160) over many rows in ONE column tile of 128 x (32 TN); 64 x 64 otherwise
            const int64_t big_tiles = cdiv64(M, 128) * cdiv64(N, 128) * split_k;
            const bool narrow = N <= 160 && cdiv64(M, 128) * split_k >= 192;
            const int tn = (int)cdiv64(N, 32);
            const bool big = !narrow && M >= 128 && N >= 96 && big_tiles >= 192;
            const dim3 gm((unsigned)(narrow ? 1 : cdiv64(N, big ? 128 : 64)), (unsigned)cdiv64(M, (big || narrow) ? 128 : 64), (unsigned)split_k);
            if (gm.y > 65535u) return SEGF_ERR_SHAPE;
#define LAUNCH_FM(L)                                                                                          \
    do {                                                                                                      \
        if (narrow && tn == 1) hipLaunchKernelGGL((gemm_f32_mfma_kernel<L, 1, 1, 4>), gm, dim3(256), 0, st, a);      \
        else if (narrow && tn == 2) hipLaunchKernelGGL((gemm_f32_mfma_kernel<L, 1, 2, 4>), gm, dim3(256), 0, st, a); \
        else if (narrow && tn == 3) hipLaunchKernelGGL((gemm_f32_mfma_kernel<L, 1, 3, 4>), gm, dim3(256), 0, st, a); \
        else if (narrow && tn == 4) hipLaunchKernelGGL((gemm_f32_mfma_kernel<L, 1, 4, 4>), gm, dim3(256), 0, st, a); \
        else if (narrow) hipLaunchKernelGGL((gemm_f32_mfma_kernel<L, 1, 5, 4>), gm, dim3(256), 0, st, a);            \
        else if (big) hipLaunchKernelGGL((gemm_f32_mfma_kernel<L, 2, 2>), gm, dim3(256), 0, st, a);           \
        else hipLaunchKernelGGL((gemm_f32_mfma_kernel<L, 1, 1>), gm, dim3(256), 0, st, a);                    \
    } while (0)
            if (layout == 0) LAUNCH_FM(0); else if (layout == 1) LAUNCH_FM(1); else LAUNCH_FM(2);
#undef LAUNCH_FM
        } else
        if (layout == 0) hipLaunchKernelGGL((gemm_f32_kernel<0>), grid, dim3(256), 0, st, a);
        else if (layout == 1) hipLaunchKernelGGL((gemm_f32_kernel<1>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_f32_kernel<2>), grid, dim3(256), 0, st, a);
    }
    SEGF_CHECK_LAUNCH();
reduce:
    if (a.ws) {
        // (the bias gradient's slices ride in the same launch)
        if (c_dt == SEGF_F32 && reduce_sink_take(ws, split_k, M, N, (float*)C, ldc, a.colsum_ws, a.colsum, M)) return 0;
        if (c_dt == SEGF_F32) splitk_reduce_launch<float>(st, ws, split_k, M, N, (float*)C, ldc, a.colsum_ws, a.colsum, M);
        else splitk_reduce_launch<bf16_t>(st, ws, split_k, M, N, (bf16_t*)C, ldc, a.colsum_ws, a.colsum, M);
        SEGF_CHECK_LAUNCH();
    }
    return 0;
}


// ---- 3x3 convolution (stride 1, pad 1) on NHWC as an implicit GEMM: no im2col matrix is materialised -------------------------
// reference: ConvModule(.., 3, 1, 1) of models/heads/upernet.py:26,28, models/modules/ppm.py:19, models/heads/fpn.py:19.
//   mode 0  y[pix][co]  = sum_{tap,ci} x[pix+off(tap)][ci] * w[co][tap*Cin+ci]        x: [P][ldx], w: [Cout][9*Cin], y: [P][ldy]
//   mode 1  dx[pix][ci] = sum_{tap,co} dy[pix-off(tap)][co] * wt[ci][tap*Cout+co]     x := dy [P][ldx], w := wt [Cin][9*Cout]
//   mode 2  dw[co][tap*Cin+ci] = sum_pix dy[pix][co] * x[pix+off(tap)][ci]            x: [P][ldx], w := dy [P][ldw], y := dw fp32
// P = B*H*W.  bf16 only (the fp32 parity mode goes through segf_im2col + segf_gemm).  Channel counts must be multiples of 8.
// The 3x3 weight gradient has a LARGE output (Cout x 9 Cin: 81 tiles of 256^2 at 768 -> 768) even where its reduction is short (the
// 32^2 .. 80^2 maps of UPerHead's FPN, upernet.py:26-28): the eight-phase tile with a few slices then beats the 128-tile kernel that
// gemm_use_big's token-count rule (made for the small outputs of nn.Linear) sends it to -- [8 x 80 x 80, 768 -> 768]: 381 -> 577 TFLOP/s.
static inline bool conv3x3_wgrad_big(int64_t M, int64_t N, int64_t K) {
    if (gemm_use_big(2, M, N, K)) return true;
    return !POL(gemm_no_big) && M % 256 == 0 && N % 256 == 0 && (M / 256) * (N / 256) >= 48 && K >= 4096;
}
// Forward / data gradient of a 3x3 convolution whose output has too few 256 x 256 tiles to fill the chip but a long reduction (UPerHead's PPM
// bottleneck 3840 -> 768 on a 16 x 16 map, ppm.py:19: 96 tiles at batch 32; the 40 x 40 / 20 x 20 levels of cfg5): split the (channel block,
// tap) walk over 2 - 8 slices of the eight-phase tile, fp32 partials, one reduce pass to bf16.  0 / 1 = no split (the caller passes ws
// of split * P * Cout floats otherwise; no bias in this form).
extern "C" int segf_conv3x3_fwd_splitk(int mode, int B, int H, int W, int Cin, int Cout) {
    if (mode < 0 || mode > 1 || POL(conv_no_fwd_split) || POL(no_gemm8)) return 1;
    const int64_t M = (int64_t)B * H * W, N = mode == 0 ? Cout : Cin, Kc = mode == 0 ? Cin : Cout, K = 9 * Kc;
    if (N % 256 || Kc % 64 || M <= 0) return 1;                  // (any pixel count: the last row tile may be ragged)
    const int64_t tiles = cdiv64(M, 256) * (N / 256);
    if (tiles >= 160 || K < 4096) return 1;
    int best = 1;
    double bu = (double)tiles / 256.0;
    for (int c = 2; c <= 8; ++c) {
        if (K / c < 24 * 64) break;                     // slices of at least 24 K steps
        const int64_t wg = tiles * c;
        const double u = (double)wg / (double)(cdiv64(wg, 256) * 256);
        if (u > bu + 0.05) { bu = u; best = c; }
        if (u >= 0.9) break;
    }
    return best;
}
// split-K count for segf_conv3x3 mode 2 (the caller sizes ws with it)
extern "C" int segf_conv3x3_pick_splitk(int Cin, int Cout, int64_t P) {
    const int64_t M = Cout, N = 9 * (int64_t)Cin;
    if (gemm_use_big(2, M, N, P) || !conv3x3_wgrad_big(M, N, P)) return segf_gemm_pick_splitk(M, N, P);
    // fewest slices (<= 8) whose last round of 256 workgroups is >= 90 % full, slices of at least 16 K steps
    const int64_t tiles = (M / 256) * (N / 256);
    int best = 1;
    double bu = 0.0;
    for (int c = 1; c <= 8; ++c) {
        if (P / c < 16 * 64) break;
        const int64_t wg = tiles * c;
        const double u = (double)wg / (double)(cdiv64(wg, 256) * 256);
        if (u > bu + 0.02) { bu = u; best = c; }
        if (u >= 0.9) break;
    }
    return best;
}
extern "C" int segf_conv3x3(int mode, int B, int H, int W, int Cin, int Cout, const void* x, int64_t ldx, const void* w, int64_t ldw,
                            void* y, int y_dt, int64_t ldy, const float* bias, int split_k, float* ws, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    if (mode < 0 || mode > 2 || Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8) return SEGF_ERR_SHAPE;
    if (y_dt != SEGF_F32 && y_dt != SEGF_BF16) return SEGF_ERR_DTYPE;
    if (((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((ldx * 2) % 16) || ((ldw * 2) % 16)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = (int64_t)B * H * W;
    GemmArgs a;
    a.bias = bias; a.residual = nullptr; a.rscale = nullptr; a.ldr = 0; a.rpg = 1;
    a.a_vec = 1; a.b_vec = 1; a.r_vec = 0; a.use_tr = 1;
    a.cH = H; a.cW = W; a.csign = mode == 1 ? -1 : 1;
    a.colsum = nullptr; a.colsum_ws = nullptr; a.pro_scale = nullptr; a.pro_shift = nullptr; a.pro_rpg = 1; a.pro_ld = 0; a.pro_act = 0;
    a.f8_sa = nullptr; a.f8_sb = nullptr;
    int layout;
    if (mode == 0) { layout = 0; a.M = P; a.N = Cout; a.K = 9 * (int64_t)Cin; a.A = x; a.lda = ldx; a.B = w; a.ldb = ldw; a.cC = Cin; }
    else if (mode == 1) { layout = 0; a.M = P; a.N = Cin; a.K = 9 * (int64_t)Cout; a.A = x; a.lda = ldx; a.B = w; a.ldb = ldw; a.cC = Cout; }
    else { layout = 2; a.M = Cout; a.N = 9 * (int64_t)Cin; a.K = P; a.A = w; a.lda = ldw; a.B = x; a.ldb = ldx; a.cC = Cin; }
    if (mode == 2 && bias) return SEGF_ERR_SHAPE;
    a.C = y; a.ldc = ldy;
    if (split_k < 1) split_k = 1;
    // forward / data gradient: split-K only in the form segf_conv3x3_fwd_splitk proposes (few output tiles, long reduction)
    if (mode != 2 && !(split_k > 1 && ws && y_dt == SEGF_BF16 && !bias && split_k == segf_conv3x3_fwd_splitk(mode, B, H, W, Cin, Cout))) split_k = 1;
    if (split_k > 1 && !ws) return SEGF_ERR_WORKSPACE;
    int64_t kchunk = cdiv64(cdiv64(a.K, split_k), GB_BK) * GB_BK;
    split_k = (int)cdiv64(a.K, kchunk);
    a.kchunk = kchunk;
    a.ws = split_k > 1 ? ws : nullptr;
    const size_t csz = y_dt == SEGF_BF16 ? 2 : 4;
    a.c_vec = ((uintptr_t)y % (4 * csz) == 0) && ((ldy * csz) % (4 * csz) == 0);
    a.c_vec16 = ((uintptr_t)y % 16 == 0) && ((ldy * csz) % 16 == 0);
    const bool f32out = y_dt == SEGF_F32 || a.ws;
    if (layout == 0 && mode != 2 && a.ws) {      // the few-tiles form: eight-phase tiles over K slices, fp32 partials, one reduce to bf16
        const int rc8 = gemm8_launch(0, 1, 0, a.M, a.N, a.K, a.kchunk, split_k, a.A, a.lda, a.B, a.ldb, y, ldy, H, W, a.cC, a.csign, nullptr, nullptr,
                                     nullptr, nullptr, 0, nullptr, 1, a.ws, st);
        if (rc8) return rc8;
        goto reduce3;
    }
    if (layout == 0 && !f32out && gemm8_supported(0, 1, a.M, a.N, a.K, a.kchunk, a.cC))
        return gemm8_launch(0, 1, 0, a.M, a.N, a.K, a.kchunk, 1, a.A, a.lda, a.B, a.ldb, y, ldy, H, W, a.cC, a.csign, nullptr, nullptr, bias,
                            nullptr, 0, nullptr, 1, nullptr, st);
    if (layout == 2 && y_dt == SEGF_F32 && conv3x3_wgrad_big(a.M, a.N, a.K) && gemm8_supported(2, 1, a.M, a.N, a.K, a.kchunk, a.cC)) {
        const int rc8 = gemm8_launch(2, 1, 0, a.M, a.N, a.K, a.kchunk, split_k, a.A, a.lda, a.B, a.ldb, y, ldy, H, W, a.cC, 1, nullptr, nullptr,
                                     nullptr, nullptr, 0, nullptr, 1, a.ws, st);
        if (rc8) return rc8;
        goto reduce3;
    }
    if (gemm_use_big(layout, a.M, a.N, a.K)) {
        dim3 gridb((unsigned)cdiv64(a.N, GG_B), (unsigned)cdiv64(a.M, GG_B), (unsigned)split_k);
        if (gridb.y > 65535u) return SEGF_ERR_SHAPE;
        if (layout == 0) {
            if (f32out) hipLaunchKernelGGL((gemm_bf16_big_kernel<0, float, true>), gridb, dim3(GG_THREADS), 0, st, a);
            else hipLaunchKernelGGL((gemm_bf16_big_kernel<0, bf16_t, true>), gridb, dim3(GG_THREADS), 0, st, a);
        } else {
            if (f32out) hipLaunchKernelGGL((gemm_bf16_big_kernel<2, float, true>), gridb, dim3(GG_THREADS), 0, st, a);
            else hipLaunchKernelGGL((gemm_bf16_big_kernel<2, bf16_t, true>), gridb, dim3(GG_THREADS), 0, st, a);
        }
        SEGF_CHECK_LAUNCH();
        goto reduce3;
    }
    {
    dim3 grid((unsigned)cdiv64(a.N, GB_BN), (unsigned)cdiv64(a.M, GB_BM), (unsigned)split_k);
    if (grid.y > 65535u) return SEGF_ERR_SHAPE;
    if (layout == 0) {
        if (f32out) hipLaunchKernelGGL((gemm_bf16_kernel<0, float, true, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_bf16_kernel<0, bf16_t, true, true>), grid, dim3(256), 0, st, a);
    } else {
        if (f32out) hipLaunchKernelGGL((gemm_bf16_kernel<2, float, true, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_bf16_kernel<2, bf16_t, true, true>), grid, dim3(256), 0, st, a);
    }
    SEGF_CHECK_LAUNCH();
    }
reduce3:
    if (a.ws) {
        if (y_dt == SEGF_F32) splitk_reduce_launch<float>(st, ws, split_k, a.M, a.N, (float*)y, ldy, nullptr, nullptr, 0);
        else splitk_reduce_launch<bf16_t>(st, ws, split_k, a.M, a.N, (bf16_t*)y, ldy, nullptr, nullptr, 0);
        SEGF_CHECK_LAUNCH();
    }
    return 0;
}


// ---- 3x3 convolution with fp8 operands (BASELINE cfg5 "fp8 MFMA weights": the UPerHead / PPM 3x3 convs are ~700 of its 2,050
// GFLOP per image; heads/upernet.py:26-31, modules/ppm.py:19).  Not a reference feature -- an option of this build.
//   mode 0  y[pix][co]  = sx * sw[co] * sum_{tap,ci} xq[pix+off(tap)][ci] wq[co][tap*Cin+ci]       xq e4m3 (one scale sx for the tensor),
//                                                                                                    wq e4m3 (one scale per row)
//   mode 1  dx[pix][ci] = sg * sw[ci] * sum_{tap,co} gq[pix-off(tap)][co] wtq[ci][tap*Cout+co]      gq e5m2 (gradient), wtq e4m3
// Same kernel, loaders and LDS images as the bf16 path (gemm_bf16_big_kernel<0, bf16, CONV, .., FP8>): the operands are addressed in
// 2-byte units, a K step of 64 units = 128 fp8 values.  Channel counts must be multiples of 16, rows 16-byte aligned.  Output bf16.
extern "C" int segf_conv3x3_fp8_supported(int mode, int B, int H, int W, int Cin, int Cout) {
    if (POL(no_fp8_conv) || mode < 0 || mode > 1 || B <= 0 || H <= 0 || W <= 0 || Cin % 16 || Cout % 16) return 0;
    const int64_t P = (int64_t)B * H * W, N = mode == 0 ? Cout : Cin, K = 9 * (int64_t)(mode == 0 ? Cin : Cout) / 2;
    return gemm_use_big(0, P, N, K) ? 1 : 0;
}
extern "C" int segf_conv3x3_fp8(int mode, int B, int H, int W, int Cin, int Cout, const void* xq, int64_t ldx, const float* sx,
                                const void* wq, int64_t ldw, const float* sw, void* y, int64_t ldy, void* stream) {
    if (!segf_conv3x3_fp8_supported(mode, B, H, W, Cin, Cout)) return SEGF_ERR_SHAPE;
    if (!xq || !wq || !sx || !sw || !y) return SEGF_ERR_SHAPE;
    if (((uintptr_t)xq % 16) || ((uintptr_t)wq % 16) || (ldx % 16) || (ldw % 16) || ((uintptr_t)y % 16) || ((ldy * 2) % 16)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = (int64_t)B * H * W;
    const int Kc = mode == 0 ? Cin : Cout;                      // channels of the gathered operand
    GemmArgs a;
    a.bias = nullptr; a.residual = nullptr; a.rscale = nullptr; a.ldr = 0; a.rpg = 1;
    a.a_vec = 1; a.b_vec = 1; a.r_vec = 0; a.use_tr = 1; a.fast = 0; a.xcd_slabs = 0;
    a.cH = H; a.cW = W; a.csign = mode == 1 ? -1 : 1;
    a.colsum = nullptr; a.colsum_ws = nullptr; a.pro_scale = nullptr; a.pro_shift = nullptr; a.pro_rpg = 1; a.pro_ld = 0; a.pro_act = 0;
    a.f8_sa = sx; a.f8_sb = sw;
    a.M = P; a.N = mode == 0 ? Cout : Cin; a.K = 9 * (int64_t)Kc / 2;          // 2-byte units
    a.A = xq; a.lda = ldx / 2; a.B = wq; a.ldb = ldw / 2; a.cC = Kc / 2;
    a.C = y; a.ldc = ldy;
    a.kchunk = cdiv64(a.K, GB_BK) * GB_BK; a.ws = nullptr;
    a.c_vec = 1; a.c_vec16 = 1;
    if (gemm8_supported(0, 1, a.M, a.N, a.K, a.kchunk, a.cC))
        return gemm8_launch(0, 1, mode == 0 ? 1 : 2, a.M, a.N, a.K, a.kchunk, 1, a.A, a.lda, a.B, a.ldb, y, ldy, H, W, a.cC, a.csign, sx, sw,
                            nullptr, nullptr, 0, nullptr, 1, nullptr, st);
    dim3 gridb((unsigned)cdiv64(a.N, GG_B), (unsigned)cdiv64(a.M, GG_B), 1);
    if (gridb.y > 65535u) return SEGF_ERR_SHAPE;
    if (mode == 0) hipLaunchKernelGGL((gemm_bf16_big_kernel<0, bf16_t, true, false, 0, false, 1>), gridb, dim3(GG_THREADS), 0, st, a);
    else hipLaunchKernelGGL((gemm_bf16_big_kernel<0, bf16_t, true, false, 0, false, 2>), gridb, dim3(GG_THREADS), 0, st, a);
    SEGF_CHECK_LAUNCH();
    return 0;
}

// mode 2 on fp8 operands: dW[co][tap*Cin+ci] = sg * sx * sum_pix gq[pix][co] xq[pix+off(tap)][ci]   (gq e5m2, xq e4m3: the tensors the
// forward and the data gradient already quantised; one byte per element, row strides in bytes).  fp32 out [Cout][9*Cin]; split over K
// (pixels) into fp32 slabs: ws >= split_k * Cout * 9 * Cin floats when split_k > 1 (segf_gemm_pick_splitk(Cout, 9 * Cin, B*H*W)).
extern "C" int segf_conv3x3_fp8_wgrad_supported(int B, int H, int W, int Cin, int Cout) {
    if (POL(no_fp8_conv) || POL(no_fp8_wgrad) || B <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t P = (int64_t)B * H * W, M = Cout, N = 9 * (int64_t)Cin;
    if (!gemm_use_big(2, M, N, P)) return 0;
    return gemm8_supported(3, 1, M, N, P, 512, Cin);
}
extern "C" int segf_conv3x3_fp8_wgrad(int B, int H, int W, int Cin, int Cout, const void* xq, int64_t ldx, const float* sx, const void* gq,
                                      int64_t ldg, const float* sg, float* dw, int64_t lddw, int split_k, float* ws, void* stream) {
    if (!segf_conv3x3_fp8_wgrad_supported(B, H, W, Cin, Cout) || !xq || !gq || !sx || !sg || !dw) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = (int64_t)B * H * W, M = Cout, N = 9 * (int64_t)Cin;
    if (split_k < 1) split_k = 1;
    if (split_k > 1 && !ws) return SEGF_ERR_WORKSPACE;
    int64_t kchunk = cdiv64(cdiv64(P, split_k), 128) * 128;
    split_k = (int)cdiv64(P, kchunk);
    if (!gemm8_supported(3, 1, M, N, P, kchunk, Cin)) return SEGF_ERR_SHAPE;
    const int rc = gemm8_launch(3, 1, 2, M, N, P, kchunk, split_k, gq, ldg, xq, ldx, dw, lddw, H, W, Cin, 1, sg, sx, nullptr, nullptr, 0, nullptr,
                                1, ws, st);
    if (rc) return rc;
    if (split_k > 1) {
        splitk_reduce_launch<float>(st, ws, split_k, M, N, dw, lddw, nullptr, nullptr, 0);
        SEGF_CHECK_LAUNCH();
    }
    return 0;
}

// ---- nn.Linear products on fp8 operands, one dynamic scale per activation / gradient TENSOR (segf_quant_tensor_fp8) and one per weight
// row (segf_quant_rows_fp8): the ConvNeXt block MLPs of BASELINE cfg5 (convnextv2.py:83-113: pwconv1 / pwconv2), all three products on
// the 256 x 256 eight-wave tile kernel (gemm8.hip) with the block-scaled K = 128 fp8 matrix instruction.
//   mode 0  y[m][n]  = sa * sb[n] * sum_k Aq[m][k] Bq[n][k] (+ bias[n]) (+ residual: y = residual + rscale[m / rpg] * (..))   A e4m3
//   mode 1  the same with A in e5m2 (the data gradient dx = dy W: Aq = the quantised gradient, Bq = W^T quantised per row)
//   mode 2  dW[n][k] = sg * sx * sum_t gq[t][n] xq[t][k]       gq e5m2 [T][N], xq e4m3 [T][K] (the tensors the forward / data gradient
//           already hold), fp32 out, split over the tokens: ws >= split_k * N * K floats (segf_gemm_pick_splitk(N, K, T))
// Row strides in bytes (= elements).  Not a reference feature -- an option of this build (SegmentationModel.set_fp8).
extern "C" int segf_linear_fp8_supported(int mode, int64_t M, int64_t N, int64_t K) {
    if (POL(no_fp8_linear) || M <= 0 || N <= 0 || K <= 0) return 0;
    if (mode == 2) {                     // M = tokens (the reduction), N x K = the weight
        if (N % 256 || K % 256 || (N / 256) * (K / 256) * (M / 1024) < 128) return 0;       // enough 256 x 256 x >= 1024-token pieces
        return gemm8_supported(3, 0, N, K, M, 512, 0);
    }
    if (mode != 0 && mode != 1) return 0;
    if (K % 128 || M % 256 || N % 256 || (M / 256) * (N / 256) < 128) return 0;      // K in 2-byte units: a multiple of 64
    return gemm8_supported(2, 0, M, N, K / 2, K / 2, 0);                              // (kind 2 = the shape rules without the tile-count bar)
}
extern "C" int segf_linear_fp8(int mode, int64_t M, int64_t N, int64_t K, const void* Aq, int64_t lda, const float* sa, const void* Bq,
                               int64_t ldb, const float* sb, void* C, int64_t ldc, const float* bias, const void* residual, int64_t ldr,
                               const float* rscale, int64_t rows_per_group, void* stream) {
    if (!segf_linear_fp8_supported(mode, M, N, K) || mode == 2) return SEGF_ERR_SHAPE;
    if (!Aq || !Bq || !sa || !sb || !C || (lda % 16) || (ldb % 16) || lda < K || ldb < K || ldc < N) return SEGF_ERR_SHAPE;
    return gemm8_launch(0, 0, mode == 0 ? 1 : 2, M, N, K / 2, K / 2, 1, Aq, lda / 2, Bq, ldb / 2, C, ldc, 0, 0, 0, 1, sa, sb, bias, residual,
                        ldr, rscale, rows_per_group, nullptr, (hipStream_t)stream);
}
// slices over the tokens for segf_linear_fp8_wgrad: one 256 x 256 tile per compute unit and slice, slices of >= 1024 tokens
extern "C" int segf_linear_fp8_wgrad_splitk(int64_t N, int64_t K, int64_t T) {
    const int64_t tiles = cdiv64(N, 256) * cdiv64(K, 256);
    int64_t s = (256 + tiles / 2) / tiles;
    if (s > T / 1024) s = T / 1024;
    if (s > 16) s = 16;
    return (int)(s < 1 ? 1 : s);
}
extern "C" int segf_linear_fp8_wgrad(int64_t N, int64_t K, int64_t T, const void* gq, int64_t ldg, const float* sg, const void* xq,
                                     int64_t ldx, const float* sx, float* dw, int64_t lddw, int split_k, float* ws, void* stream) {
    if (!segf_linear_fp8_supported(2, T, N, K)) return SEGF_ERR_SHAPE;
    if (!gq || !xq || !sg || !sx || !dw || ldg < N || ldx < K || lddw < K) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (split_k < 1) split_k = 1;
    if (split_k > 1 && !ws) return SEGF_ERR_WORKSPACE;
    int64_t kchunk = cdiv64(cdiv64(T, split_k), 128) * 128;
    split_k = (int)cdiv64(T, kchunk);
    if (!gemm8_supported(3, 0, N, K, T, kchunk, 0)) return SEGF_ERR_SHAPE;
    const int rc = gemm8_launch(3, 0, 2, N, K, T, kchunk, split_k, gq, ldg, xq, ldx, dw, lddw, 0, 0, 0, 1, sg, sx, nullptr, nullptr, 0, nullptr,
                                1, ws, st);
    if (rc) return rc;
    if (split_k > 1) {
        splitk_reduce_launch<float>(st, ws, split_k, N, K, dw, lddw, nullptr, nullptr, 0);
        SEGF_CHECK_LAUNCH();
    }
    return 0;
}
