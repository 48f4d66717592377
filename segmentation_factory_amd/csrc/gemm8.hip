// 256 x 256 x 64 bf16 / fp8 GEMM tile in EIGHT PHASES per two K tiles: the large products of the UPerNet configurations --
// the implicit-GEMM 3x3 convolutions of UPerHead / PPM in all three directions (heads/upernet.py:26-31, modules/ppm.py:19: 75 % of
// BASELINE cfg3's step, 42 % of cfg5's) and the big nn.Linear products of the ConvNeXt / MiT blocks (convnextv2.py:90-95, mit.py:98-99).
//
// The two-phase kernel of gemm.hip (load -> barrier -> 64 MFMAs per wave -> LDS write -> barrier, one workgroup per CU) reaches
// 0.79 PFLOP/s on these shapes: every K step pays its LDS write pass (ds_write_b128: 79 B/clk per CU), its barrier and the drain of
// its loads in series with the matrix instructions.  Structure here (cdna_hip_programming.md, "The 256^2 8-phase template"):
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass.  The LDS image is
//     lane-linear per instruction, so the XOR swizzle that makes the fragment reads conflict-free is applied to the GLOBAL address of
//     each lane (logical chunk = physical chunk ^ swizzle(row));
//   * a K tile is four HALF-tiles of 16 KB (A rows 0-127 / 128-255, B rows 0-127 / 128-255 of the workgroup tile); a wave's 128 x 64
//     output is two 64-row pieces x two 32-column pieces, one in each half, so its four C QUADRANTS (m0,n0) (m0,n1) (m1,n1) (m1,n0) read
//     A0 B0 | A0 B1 | A1 B1 | A1 B0: each phase = one quadrant x K = 64 = 16 MFMAs, loads 8 (A) and / or 4 (B) fragments, and stages
//     ONE half-tile, in the order in which the halves fall free: B0[kt+1], A1[kt+1], A0[kt+2], B1[kt+2] -- three half-tiles
//     (1.5 K tiles) in flight across the barriers, retired by COUNTED waits (vmcnt(8) in phase 2, vmcnt(6) in phase 4), raw
//     s_barrier (a __syncthreads() would drain the DMA queue), one phase between a wait and the first read of what it retired.
//     Past the last K tile the schedule keeps issuing DUMMY loads (so the counts stay uniform) into the half where the tile WOULD go;
//   * operand layouts (per operand): KC = K-contiguous rows ([rows][64 k] half-tiles of 128-byte rows, fragments by ds_read_b128) or
//     RM = reduction-major ([64 k][128 columns], 256-byte rows, fragments by ds_read_b64_tr_b16 -- as INLINE ASM: in front of the
//     builtin hipcc puts s_waitcnt vmcnt(0) while an LDS-DMA is in flight, which drains the pipeline; the consumer side is ordered by
//     hand, s_waitcnt lgkmcnt(0) + sched_barrier(0) before the MFMAs);
//   * implicit convolution: forward / data gradient gather the KC operand A (a K tile of 64 units lies inside one tap, channels % 64
//     == 0: one wave-uniform pixel offset per K tile plus a per-lane border test); the weight gradient gathers the RM operand B (a
//     half-tile of 128 columns lies inside one tap, channels % 128 == 0; the pixel coordinates of a lane's rows advance by 64 per K
//     tile without a division).  Taps outside the image read a ZERO PAGE (LDS-DMA cannot write zeros itself);
//   * fp8 operands (KC x KC only): 2-byte units along K as in gemm.hip's FP8 mode (the loaders and LDS images move bytes), but the two
//     16-byte fragments of a row feed ONE v_mfma_scale_f32_16x16x128_f8f6f4 (block scales 2^0): the non-scaled fp8 MFMA runs at the
//     bf16 rate, the block-scaled one at twice that.
// Measured on the MI355X (3072 -> 768 3x3 @ 128^2, batch 32 = 22.3 TFLOP per direction): forward 28.3 -> 20.4 ms (787 -> 1094
// TFLOP/s), data gradient 28.6 -> 21.2, weight gradient 25.8 -> 23.3 (957 TFLOP/s); fp8 forward 18.0 -> 14.5 ms (1534 TFLOP/s).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 g8_bf16x8;
typedef __attribute__((ext_vector_type(4))) float g8_f32x4;
typedef __attribute__((ext_vector_type(4))) short g8_s16x4;
template <int N> using g8_ic = std::integral_constant<int, N>;

struct Gemm8Args {
    const unsigned char* A; const unsigned char* B; void* C;
    int64_t M, N, K;                 // K: contraction length (KC operands: 2-byte units; RM operands: rows)
    int64_t lda, ldb, ldc;           // row strides in 2-byte units (ldc: output elements)
    int64_t kchunk;                  // K range per grid.z slice (split-K: fp32 partial slabs in ws)
    int cH, cW, cC, csign;           // CONV: NHWC geometry of the gathered operand (cC in 2-byte units for the KC gather)
    const float* f8_sa; const float* f8_sb;
    const float* bias;               // [N] (bf16 output only)
    const bf16_t* residual; int64_t ldr; const float* rscale; int64_t rpg;      // C = residual + rscale[m / rpg] * (...)
    float* ws;                       // [z][M][N] fp32 when gridDim.z > 1
    int tile_order;                  // CONV gather on B (weight gradient): 1 = the taps / row tiles of a channel block are launch neighbours
    int stagger;                     // 1 = waves 4-7 run one barrier behind waves 0-3 (see the main loop)
    const unsigned char* zero;       // the zero page (CONV border taps); a kernel argument so that it sits in scalar registers -- as a global it
                                     // was re-materialised (s_getpc_b64 + two adds + two moves) in front of every gathered load
};
// the zero page of the CONV border taps: LDS-DMA cannot write zeros itself, so lanes whose tap lies outside the image load from here
__device__ __attribute__((aligned(256))) unsigned char g8_zero_page[256];

#define G8_HALF 16384
#define G8_BUF 65536
#define G8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G8_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G8_GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), \
                                                            (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)

__device__ __forceinline__ int g8_rm_swz(int krow) { return 2 * ((krow & 3) + 4 * ((krow >> 3) & 1)); }
// KC half-tile [128 rows][128 B]: lane (i = lane & 15, g = lane >> 4) of the 16-row tile at row0, K sub-step s
__device__ __forceinline__ g8_bf16x8 g8_frag_kc(const unsigned char* half, int row0, int s, int lane) {
    const int row = row0 + (lane & 15);
    return *reinterpret_cast<const g8_bf16x8*>(half + row * 128 + (((4 * s + (lane >> 4)) ^ (row & 7)) << 4));
}
// RM half-tile [64 k][256 B]: lane (i, g) receives T[k = 32 s + 8 g + j][column cb + i], j = 0..7 (two transposed 4-row reads).
// The per-lane byte offset inside a half-tile is split into a part that depends on the 16-column block (g8_tr_base: computed ONCE per
// kernel, 4 + 2 registers) and compile-time immediates: k + 4 and k + 32 keep the swizzle ((k & 3) and bit 3 of k are unchanged), so the
// second read of a fragment is +1024 bytes and K sub-step 1 is +8192 bytes; the half-tile's own offset joins the immediate where it fits.
__device__ __forceinline__ uint32_t g8_tr_base(int cb, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int u = (cb >> 2) + p;
    const int chunk = u >> 1, half = u & 1;
    const int k0 = 8 * g + q;
    return (uint32_t)(k0 * 256 + ((chunk ^ g8_rm_swz(k0)) << 4) + half * 8);
}
template <int OFF>
__device__ __forceinline__ g8_bf16x8 g8_frag_tr(uint32_t addr) {
    g8_s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "i"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "i"(OFF + 1024));
    return __builtin_bit_cast(g8_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// fp8 RM half-tile [128 k][128 B] (one byte per element): lane (i, g) of the 16-column block at cb receives T[k = 32 g + 16 s + j][cb + i],
// j = 0..15, as two ds_read_b64_tr_b8 (probed on the MI355X, tools/probe/tr8_probe.hip: lane 2 q + p of a 16-lane group supplies the
// address of row q, bytes 8 p .. 8 p + 7 of an 8-row x 16-byte block; lane i receives column i of the 8 rows, row q in byte q).
// Swizzle: 16-byte chunk c of row r sits at chunk c ^ f(r), f(r) = ((r >> 1) & 3) | (((r >> 5) & 1) << 2): the 8 rows of a read and
// the two 16-lane groups of a 32-lane half land on 16 different 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int g8_rm8_swz(int r) { return ((r >> 1) & 3) | (((r >> 5) & 1) << 2); }
__device__ __forceinline__ g8_bf16x8 g8_frag_tr8(const unsigned char* tile, int cb, int s, int lane) {
    typedef int g8_v2i __attribute__((ext_vector_type(2)));
    typedef int g8_v4i __attribute__((ext_vector_type(4)));
    const int g = lane >> 4, q = (lane & 15) >> 1, p = lane & 1, c = cb >> 4;
    const int r0 = 32 * g + 16 * s + q, r1 = r0 + 8;
    const uint32_t a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(tile + r0 * 128 + ((c ^ g8_rm8_swz(r0)) << 4) + 8 * p);
    const uint32_t a1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(tile + r1 * 128 + ((c ^ g8_rm8_swz(r1)) << 4) + 8 * p);
    g8_v2i lo, hi;
    asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(hi) : "v"(a1));
    return __builtin_bit_cast(g8_bf16x8, g8_v4i{lo[0], lo[1], hi[0], hi[1]});
}

// The same read with the per-lane address split as for bf16 (r05): the swizzle term f(r) of rows r = 32 g + 16 s + q (+ 8) does not depend on
// s or on the + 8 ((r >> 1) & 3 = (q >> 1) & 3, r >> 5 = g), so ONE base per 16-column block and lane, computed once per kernel, serves all
// four reads of a fragment pair through immediates (+ 1024 for rows + 8, + 2048 for K sub-step 1, + the half-tile's own offset) -- the
// form above recomputed two addresses per fragment, 48 vector adds per K tile in the read sections.
__device__ __forceinline__ uint32_t g8_tr8_base(int cb, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 1, p = lane & 1, c = cb >> 4;
    const int r0 = 32 * g + q;
    return (uint32_t)(r0 * 128 + ((c ^ g8_rm8_swz(r0)) << 4) + 8 * p);
}
template <int OFF>
__device__ __forceinline__ g8_bf16x8 g8_frag_tr8i(uint32_t addr) {
    typedef int g8_v2i __attribute__((ext_vector_type(2)));
    typedef int g8_v4i __attribute__((ext_vector_type(4)));
    g8_v2i lo, hi;
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "i"(OFF));
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "i"(OFF + 1024));
    return __builtin_bit_cast(g8_bf16x8, g8_v4i{lo[0], lo[1], hi[0], hi[1]});
}

// ALAY / BLAY: 0 = KC, 1 = RM (bf16), 2 = RM fp8 (K tile = 128 rows).  CONV gathers A when ALAY == 0 (forward / data gradient), B when
// both are RM (weight gradient).
template <int ALAY, int BLAY, bool CONV, int FP8, typename OutT>
__global__ void __launch_bounds__(512) gemm8_kernel(Gemm8Args a) {
    static_assert(FP8 == 0 || (ALAY == 0 && BLAY == 0) || (ALAY == 2 && BLAY == 2), "fp8: both operands K-contiguous or both fp8 reduction-major");
    static_assert((ALAY == 2) == (BLAY == 2) && (ALAY != 2 || FP8 != 0), "fp8 reduction-major: both operands");
    static_assert(!CONV || ALAY == 0 || (ALAY >= 1 && BLAY == ALAY), "gather: A (KC) or B (RM x RM)");
    constexpr bool CONV_A = CONV && ALAY == 0, CONV_B = CONV && ALAY >= 1;
    constexpr int KT = ALAY == 2 ? 128 : 64;                  // rows / units of K per tile
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * G8_BUF];       // [buf][A0, A1, B0, B1][16 KB]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned nwg = gx * gy;
    const unsigned orig = blockIdx.x + gx * blockIdx.y;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);      // bijective XCD remap
    unsigned bx = wgid % gx, by = wgid / gx;
    // (r05 experiment, removed: inside an XCD's share of the launch the ROW tiles as the fastest index -- one weight stream per XCD at a
    // time.  L2 hit 77 -> 91 %, fabric fetch 38.6 -> 14.7 GB per launch of the UPerHead bottleneck conv, and NO change in time, held clock
    // or MFMA-busy; fp8 2 % slower: profiles/r05_gemm8_pmc_summary.txt.  The forward is not bound by what it fetches.)
    if (CONV_B && a.tile_order && a.cC % 256 == 0 && gx == 9u * (unsigned)(a.cC / 256)) {
        // weight gradient: the 9 taps x gy row tiles of ONE 256-channel block are neighbours in the launch order (hence on one XCD, at
        // the same time): they read the same pixels of x shifted by a row / a column and the same dy tiles, so x is fetched once per
        // channel block instead of once per (tap, row tile) -- with the taps outermost FETCH_SIZE was 75 GB per launch for 4 GB of operands
        const unsigned ncb = (unsigned)(a.cC / 256), grp = 9u * gy;
        const unsigned chb = wgid / grp, rem = wgid - chb * grp;
        by = rem / 9u;
        bx = (rem - by * 9u) * ncb + chb;
    }
    const int64_t m0 = (int64_t)by * 256, n0 = (int64_t)bx * 256;
    const int64_t kbeg = (int64_t)blockIdx.z * a.kchunk;
    const int64_t kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const int nk = (int)((kend - kbeg) / KT);

    // ---- staging geometry.  KC: this wave stages rows 16 wave + 8 i + (lane >> 3) of a half-tile, physical chunk lane & 7 holds the
    // logical chunk (lane & 7) ^ (row & 7).  RM: rows 8 wave + 4 i + (lane >> 4), physical chunk lane & 15 = logical ^ rm_swz(row).
    const int kc_row = 16 * wave + (lane >> 3), kc_chunk = (lane & 7) ^ (lane >> 3);
    const int rm_row = ALAY == 2 ? kc_row : 8 * wave + (lane >> 4);          // fp8 RM: 8 rows x 128 B per instruction, like KC
    constexpr int RM_STEP = ALAY == 2 ? 8 : 4;                               // rows between a lane's two instructions
    int rm_chunk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
        rm_chunk[i] = ALAY == 2 ? ((lane & 7) ^ g8_rm8_swz(rm_row + 8 * i)) : ((lane & 15) ^ g8_rm_swz(rm_row + 4 * i));
    // KC operands: one base pointer per (half, i) at k = kbeg
    const unsigned char* gA[2][2];
    const unsigned char* gB[2][2];
    unsigned amask[2][2];                                     // CONV_A: bit t = tap t of this staged row lies inside the image
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (ALAY == 0) {
                const int64_t m = m0 + 128 * h + kc_row + 8 * i;
                if (CONV_A) {
                    const int x = (int)(m % a.cW), y = (int)((m / a.cW) % a.cH);
                    unsigned mk = 0;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int yy = y + a.csign * (t / 3 - 1), xx = x + a.csign * (t % 3 - 1);
                        mk |= (unsigned)(yy >= 0 && yy < a.cH && xx >= 0 && xx < a.cW) << t;
                    }
                    // (a ragged last row tile -- pixel counts that are not multiples of 256, e.g. 8 x 20 x 20: rows past the end read the
                    // zero page for every tap and are not stored)
                    amask[h][i] = m < a.M ? mk : 0u;
                    gA[h][i] = a.A + ((m < a.M ? m : 0) * a.lda + 8 * kc_chunk) * 2;           // the K position (tap, channel) joins per stage
                } else {
                    // (plain products with a ragged last row / column tile, r05: rows past the end re-read the last row -- finite values that
                    // only reach accumulators the epilogue does not store)
                    gA[h][i] = a.A + ((m < a.M ? m : a.M - 1) * a.lda + kbeg + 8 * kc_chunk) * 2;
                }
            }
            // (CONV_A: the weight row's K position is the absolute (tap, channel block) piece of each stage -- kbeg joins there)
            if (BLAY == 0) {
                const int64_t n = n0 + 128 * h + kc_row + 8 * i;
                gB[h][i] = a.B + ((CONV || n < a.N ? n : a.N - 1) * a.ldb + (CONV_A ? 0 : kbeg) + 8 * kc_chunk) * 2;
            }
        }
    // CONV_A walks K with the CHANNEL BLOCK outermost and the nine taps innermost: K tile T = (64-channel block T / 9, tap T % 9), so
    // that the nine taps of one channel block -- the same pixels shifted by a row / a column -- are read back to back in time.  Same-
    // device A/B against the order in which the weights store K (tap outermost, SEGFAC_G8_KORDER=0): +1.5 % bf16, +13 % fp8; the
    // counters do not show the L2 reuse this was meant to buy (FETCH_SIZE unchanged at ~20 M KB per launch), so the gain is elsewhere
    // (the scalar position arithmetic is a multiply-shift by the constant 9 here, by a run-time reciprocal there).
    // (The weight operand's K tile is then the 128-byte piece at (tap, channel block) of its row: a strided walk over the same bytes.)
    const int cv_row32 = CONV_A ? a.csign * (int)a.lda * 2 : 0;               // bytes per pixel step, signed
    // RM operands (bf16): per-lane fragment addresses inside a half-tile, one per 16-column block of this wave (g8_frag_tr)
    uint32_t trA[4], trB[2];
    if (ALAY == 1 || BLAY == 1) {
        const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
#pragma unroll
        for (int t = 0; t < 4; ++t) trA[t] = sb + g8_tr_base(64 * wm + 16 * t, lane);
#pragma unroll
        for (int u = 0; u < 2; ++u) trB[u] = sb + g8_tr_base(32 * wn + 16 * u, lane);
    }
    if (ALAY == 2) {
        const uint32_t sb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)smem;
#pragma unroll
        for (int t = 0; t < 4; ++t) trA[t] = sb + g8_tr8_base(64 * wm + 16 * t, lane);
#pragma unroll
        for (int u = 0; u < 2; ++u) trB[u] = sb + g8_tr8_base(32 * wn + 16 * u, lane);
    }
    // CONV_B: tap of each B half (workgroup constant) and the pixel coordinates of this lane's two rows at the NEXT K tile of each
    // half (the tiles of a half are staged in increasing order), advanced by 64 pixels per tile without a division
    // RM staging: a lane's byte offset inside the (wave-uniform) K tile of each operand -- the tile's own position is scalar arithmetic
    constexpr int EBR = ALAY == 2 ? 1 : 2;                                   // bytes per element of the RM operands
    uint32_t rmoffA[2] = {0u, 0u}, rmoffB[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (ALAY >= 1) rmoffA[i] = (uint32_t)((int64_t)(rm_row + RM_STEP * i) * a.lda * EBR + 16 * rm_chunk[i]);
        if (BLAY >= 1) rmoffB[i] = (uint32_t)((int64_t)(rm_row + RM_STEP * i) * a.ldb * EBR + 16 * rm_chunk[i]);
    }
    // layout 1 with a ragged last column tile: the 16-byte chunk of columns this lane stages must not run past the row (the very last row
    // of B would be read past its end): such lanes re-read the row's last chunk, per half (N is a multiple of 8: host check)
    constexpr bool RAG1 = ALAY == 0 && BLAY == 1 && !CONV;
    int64_t rmoffB1[2] = {(int64_t)rmoffB[0], (int64_t)rmoffB[1]};          // (half 1; signed: a clamped chunk may lie before the half's base)
    if (RAG1 && n0 + 256 > a.N) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t last = (a.N - 8) * 2;                                   // byte offset of the row's last chunk
            const int64_t row = (int64_t)(rm_row + RM_STEP * i) * a.ldb * 2;
            const int64_t c0 = n0 * 2 + 16 * rm_chunk[i], c1 = c0 + 256;          // this lane's chunk in half 0 / half 1, from the row start
            rmoffB[i] = (uint32_t)(row + (c0 <= last ? c0 : last) - n0 * 2);
            rmoffB1[i] = row + (c1 <= last ? c1 : last) - n0 * 2 - 256;
        }
    }
    // (r05: ONE coordinate state for both halves.  B1 is always staged one K tile ahead of B0 -- prologue B1[0], B0[0], B1[1]; loop B0[kt + 1] in
    // phase 1, B1[kt + 2] in phase 4 -- so the state holds "B0's next tile": a B0 stage uses it and advances it, the B1 stage that follows
    // uses it as it stands.  Per K tile that is one coordinate update per staged row instead of two.)
    int tdy[2], tdx[2], tci[2], py[2], px[2];
    int adv_q = 0, adv_r = 0;
    if (CONV_B) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t n = n0 + 128 * h;
            const int tap = (int)(n / a.cC);
            tdy[h] = tap / 3 - 1; tdx[h] = tap % 3 - 1; tci[h] = (int)(n - (int64_t)tap * a.cC);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t t = kbeg + rm_row + RM_STEP * i;
            const int x = (int)(t % a.cW), y = (int)((t / a.cW) % a.cH);
            py[i] = y; px[i] = x;
        }
        adv_q = (KT / a.cW) % a.cH; adv_r = KT % a.cW;                      // y advances modulo the image height: one conditional wrap
    }
    auto lds_half = [&](int buf, int half) -> unsigned char* { return smem + buf * G8_BUF + half * G8_HALF; };
    // half: 0 = A0, 1 = A1, 2 = B0, 3 = B1.  The address uses the K tile clamped to the last one, the destination the tile's own buffer
    // CONV_A: where K tile T sits in the (channel block, tap) walk -- the gathered operand's byte offset from a row's own pixel, the weight
    // row's byte offset, the tap's bit in the rows' validity masks.  Computed ONCE per K tile (r05; it was recomputed by each of the four
    // half-stages: a multiply-high, two multiplies and the offset arithmetic, 4 x ~20 scalar instructions per K tile in the read sections
    // that the partner wave's MFMA section has to cover) and handed to the stages of that tile.
    struct G8Pos { int offA, offB; unsigned tap; };
    auto pos_of = [&](int kt) -> G8Pos {
        G8Pos p = {0, 0, 0u};
        if (CONV_A) {
            const int ktc = kt < nk ? kt : nk - 1;
            const unsigned kabs = (unsigned)ktc + (unsigned)(kbeg / KT);          // (split-K: this slice starts at tile kbeg / KT of the walk)
            const unsigned chb = __umulhi(kabs, 0x38E38E39u) >> 1, tap = kabs - 9u * chb;       // / 9 by multiply-shift
            const int t3 = (int)((tap * 11u) >> 5);                                // tap / 3 for tap < 9
            // (32-bit: the offset is relative to the row's own pixel, |(cW + 1) lda 2| < 2^31 is checked on the host)
            p.offA = ((t3 - 1) * a.cW + ((int)tap - 3 * t3 - 1)) * cv_row32 + (int)chb * 128;
            p.offB = ((int)tap * a.cC + (int)chb * 64) * 2;
            p.tap = tap;
        }
        return p;
    };
    auto stage = [&](int kt, int half, const G8Pos& pos) {
        const int ktc = kt < nk ? kt : nk - 1;
        const bool isA = half < 2;
        const int h = half & 1;
        if ((isA ? ALAY : BLAY) == 0) {
            unsigned char* dst = lds_half(kt & 1, half) + (16 * wave) * 128;
            if (CONV_A) {
                if (isA) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned char* p = ((amask[h][i] >> pos.tap) & 1u) ? gA[h][i] + pos.offA : a.zero;
                        G8_GLDS(p, dst + i * 1024);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) G8_GLDS(gB[h][i] + pos.offB, dst + i * 1024);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) G8_GLDS((isA ? gA[h][i] : gB[h][i]) + (int64_t)ktc * 128, dst + i * 1024);
            }
        } else {
            // RM: bf16 = 4 rows x 256 B per instruction, fp8 = 8 rows x 128 B; element size EB bytes.  The K tile's position (and, for the
            // gathered operand, its tap) is wave-uniform: a scalar base pointer per stage + the lane's constant offset (one 64-bit add)
            constexpr int EB = ALAY == 2 ? 1 : 2;
            unsigned char* dst = lds_half(kt & 1, half) + (ALAY == 2 ? 16 * wave * 128 : 8 * wave * 256);
            const int64_t t0 = kbeg + (int64_t)ktc * KT;
            if (isA || !CONV_B) {
                const unsigned char* base = (isA ? a.A + (t0 * a.lda + m0 + 128 * h) * EB : a.B + (t0 * a.ldb + n0 + 128 * h) * EB);
#pragma unroll
                for (int i = 0; i < 2; ++i) G8_GLDS(base + (isA ? (int64_t)rmoffA[i] : ((RAG1 && h) ? rmoffB1[i] : (int64_t)rmoffB[i])), dst + i * 1024);
            } else {
                // past the last K tile only a dummy load (zero page): folded into the row bound (a scalar select) instead of a branch
                const unsigned lim_h = kt < nk ? (unsigned)a.cH : 0u;
                const unsigned char* base = a.B + ((t0 + (int64_t)tdy[h] * a.cW + tdx[h]) * a.ldb + tci[h]) * EB;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int yy = py[i] + tdy[h], xx = px[i] + tdx[h];
                    const bool ok = (unsigned)yy < lim_h && (unsigned)xx < (unsigned)a.cW;
                    const void* src = ok ? (const void*)(base + rmoffB[i]) : (const void*)a.zero;
                    G8_GLDS(src, dst + i * 1024);
                    if (h == 0) {                                   // (B0: the state moves on to the next K tile)
                        int nx = px[i] + adv_r, ny = py[i] + adv_q;
                        if (nx >= a.cW) { nx -= a.cW; ++ny; }
                        if (ny >= a.cH) ny -= a.cH;
                        px[i] = nx; py[i] = ny;
                    }
                }
            }
        }
    };
    g8_f32x4 acc[4][8];                // [column tile: 2 nq + u][row tile: 4 mq + t]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = g8_f32x4{0.f, 0.f, 0.f, 0.f};
    g8_bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    auto load_a = [&](int buf, auto mqc) {
        constexpr int MQ = decltype(mqc)::value;
        if constexpr (ALAY == 1) {
            const uint32_t bo = (uint32_t)buf << 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t va = trA[t] + bo;
                fa[t][0] = g8_frag_tr<MQ * G8_HALF>(va);
                fa[t][1] = g8_frag_tr<MQ * G8_HALF + 8192>(va);
            }
        } else if constexpr (ALAY == 2) {
            const uint32_t bo = (uint32_t)buf << 16;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t va = trA[t] + bo;
                fa[t][0] = g8_frag_tr8i<MQ * G8_HALF>(va);
                fa[t][1] = g8_frag_tr8i<MQ * G8_HALF + 2048>(va);
            }
        } else {
            const unsigned char* h = lds_half(buf, MQ);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) fa[t][s] = g8_frag_kc(h, 64 * wm + 16 * t, s, lane);
        }
    };
    auto load_b = [&](int buf, auto nqc, g8_bf16x8 (&fb)[2][2]) {
        constexpr int NQ = decltype(nqc)::value;
        if constexpr (BLAY == 1) {
            const uint32_t bo = (uint32_t)buf << 16;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t va = trB[u] + bo;
                fb[u][0] = g8_frag_tr<(2 + NQ) * G8_HALF>(va);
                fb[u][1] = g8_frag_tr<(2 + NQ) * G8_HALF + 8192>(va);
            }
        } else if constexpr (BLAY == 2) {
            const uint32_t bo = (uint32_t)buf << 16;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t va = trB[u] + bo;
                fb[u][0] = g8_frag_tr8i<(2 + NQ) * G8_HALF>(va);
                fb[u][1] = g8_frag_tr8i<(2 + NQ) * G8_HALF + 2048>(va);
            }
        } else {
            const unsigned char* h = lds_half(buf, 2 + NQ);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) fb[u][s] = g8_frag_kc(h, 32 * wn + 16 * u, s, lane);
        }
    };
    auto mma = [&](int mq, int nq, const g8_bf16x8 (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8 == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        acc[2 * nq + u][4 * mq + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u][s], fa[t][s], acc[2 * nq + u][4 * mq + t], 0, 0, 0);
        } else {
            // fp8: the two 16-byte fragments of a row (K sub-steps 0 and 1 = 32 of its 128 values per lane group) are ONE operand of the
            // block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 with all block scales 2^0 (twice the bf16 rate at 4x the K; the non-scaled
            // v_mfma_f32_16x16x32_fp8 runs at the bf16 rate).  A and B split their rows the same way, so the k pairing is consistent.
            // Format codes: 0 = e4m3, 1 = e5m2; the "swapped" product puts the weight side first.
            typedef int g8_i32x8 __attribute__((ext_vector_type(8)));
            typedef int g8_i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const g8_i32x4 w0 = __builtin_bit_cast(g8_i32x4, fb[u][0]), w1 = __builtin_bit_cast(g8_i32x4, fb[u][1]);
                const g8_i32x8 wv = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const g8_i32x4 x0 = __builtin_bit_cast(g8_i32x4, fa[t][0]), x1 = __builtin_bit_cast(g8_i32x4, fa[t][1]);
                    const g8_i32x8 xv = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    acc[2 * nq + u][4 * mq + t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        wv, xv, acc[2 * nq + u][4 * mq + t], 0 /* weights: e4m3 */, FP8 == 2 ? 1 : 0 /* tokens: e5m2 gradient or e4m3 */,
                        0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: K tiles 0 and 1 in the steady-state issue order A0, B1, B0, A1 | A0, B1 (B0[1], A1[1] follow in phases 1, 2)
    G8Pos p1;                                                           // position of K tile kt + 1 (carried from iteration to iteration)
    {
        const G8Pos p0 = pos_of(0);
        p1 = pos_of(1);
        stage(0, 0, p0); stage(0, 3, p0); stage(0, 2, p0); stage(0, 1, p0); stage(1, 0, p1); stage(1, 3, p1);
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                 // all of K tile 0 has landed (this wave's share)
    G8_BAR();
    // STAGGER: waves 4-7 (the SIMD partners of waves 0-3) run one barrier behind, so that on every SIMD one wave is in its MFMA
    // section while the other reads fragments and issues its DMA -- in lockstep all eight waves read together and then queue for the
    // matrix pipe together, and the pipe idles through every read section (measured: 43 % MFMA-busy).  Every wait placement below
    // keeps "wait before a phase's first barrier, read in the next phase": with the groups one barrier apart that is still at least
    // one barrier between ANY wave's wait and ANY wave's read, and a half-tile is restaged two phases after its last read.
    const bool late = wave >= 4 && a.stagger;
    if (late) G8_BAR();
    constexpr g8_ic<0> I0{}; constexpr g8_ic<1> I1{};
    for (int kt = 0; kt < nk; ++kt) {
        const int b = kt & 1;
        // phase 1: quadrant (m0, n0)
        load_a(b, I0); load_b(b, I0, fb0);
        stage(kt + 1, 2, p1);
        G8_BAR(); G8_LGKM0();
        mma(0, 0, fb0);
        G8_BAR();
        // phase 2: quadrant (m0, n1)
        load_b(b, I1, fb1);
        stage(kt + 1, 1, p1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // A1[kt] (read in phase 3)
        G8_BAR(); G8_LGKM0();
        mma(0, 1, fb1);
        G8_BAR();
        // phase 3: quadrant (m1, n1)
        load_a(b, I1);
        const G8Pos p2 = pos_of(kt + 2);
        stage(kt + 2, 0, p2);
        G8_BAR(); G8_LGKM0();
        mma(1, 1, fb1);
        G8_BAR();
        // phase 4: quadrant (m1, n0)
        stage(kt + 2, 3, p2);
        p1 = p2;
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");             // A0[kt+1], B1[kt+1], B0[kt+1] have landed: read from the next phase on
        G8_BAR();
        mma(1, 0, fb0);
        G8_BAR();
    }
    if (!late && a.stagger) G8_BAR();                                // pairs with the late group's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- epilogue: acc[2 nq + u][4 mq + t][r] = C[m0 + 128 mq + 64 wm + 16 t + fi][n0 + 128 nq + 32 wn + 16 u + 4 fg + r]
    const int fi = lane & 15, fg = lane >> 4;
    if constexpr (sizeof(OutT) == 4) {
        float* out = a.ws ? a.ws + (int64_t)blockIdx.z * a.M * a.N : reinterpret_cast<float*>(a.C);
        const int64_t ldo = a.ws ? a.N : a.ldc;
        const float sc8 = FP8 ? a.f8_sa[0] * a.f8_sb[0] : 1.f;           // fp8 weight gradient: both operands carry one tensor scale
#pragma unroll
        for (int nq = 0; nq < 2; ++nq)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t n = n0 + 128 * nq + 32 * wn + 16 * u + 4 * fg;
#pragma unroll
                for (int mq = 0; mq < 2; ++mq)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int64_t m = m0 + 128 * mq + 64 * wm + 16 * t + fi;
                        if (CONV_A && m >= a.M) continue;
                        *reinterpret_cast<g8_f32x4*>(out + m * ldo + n) = FP8 ? acc[2 * nq + u][4 * mq + t] * sc8 : acc[2 * nq + u][4 * mq + t];
                    }
            }
    } else {
        const float sa = FP8 ? a.f8_sa[0] : 1.f;
        bf16_t* C = reinterpret_cast<bf16_t*>(a.C);
#pragma unroll
        for (int nq = 0; nq < 2; ++nq)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t n = n0 + 128 * nq + 32 * wn + 16 * u + 4 * fg;
                if (!CONV && n >= a.N) continue;                                  // (ragged last column tile of a plain product; N % 4 == 0)
                float sc[4] = {1.f, 1.f, 1.f, 1.f}, bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (FP8) sc[r] = sa * a.f8_sb[n + r];
                    if (a.bias) bs[r] = a.bias[n + r];
                }
#pragma unroll
                for (int mq = 0; mq < 2; ++mq)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int64_t m = m0 + 128 * mq + 64 * wm + 16 * t + fi;
                        if ((CONV_A || (!CONV && sizeof(OutT) == 2)) && m >= a.M) continue;
                        const g8_f32x4 c = acc[2 * nq + u][4 * mq + t];
                        float v[4] = {fmaf(c[0], sc[0], bs[0]), fmaf(c[1], sc[1], bs[1]), fmaf(c[2], sc[2], bs[2]), fmaf(c[3], sc[3], bs[3])};
                        if (a.residual) {
                            const float rs = a.rscale ? a.rscale[m / a.rpg] : 1.f;
                            const uint2 rr = *reinterpret_cast<const uint2*>(a.residual + m * a.ldr + n);
                            v[0] = fmaf(rs, v[0], __uint_as_float(rr.x << 16)); v[1] = fmaf(rs, v[1], __uint_as_float(rr.x & 0xffff0000u));
                            v[2] = fmaf(rs, v[2], __uint_as_float(rr.y << 16)); v[3] = fmaf(rs, v[3], __uint_as_float(rr.y & 0xffff0000u));
                        }
                        *reinterpret_cast<uint2*>(C + m * a.ldc + n) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
                    }
            }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------------
// kind 0: C[m][n] = sum_k A[m][k] B[n][k]   (layout 0; conv = forward / data gradient gather on A)           bf16 out
// kind 1: C[m][n] = sum_k A[m][k] B[k][n]   (layout 1: data gradient of nn.Linear)                           bf16 out
// kind 2: C[m][n] = sum_k A[k][m] B[k][n]   (layout 2; conv = weight-gradient gather on B), split-K          fp32 out
// kind 3: the same on fp8 operands (A e5m2, B e4m3, one byte per element, K tiles of 128 rows), scaled by f8_sa[0] * f8_sb[0]
int gemm8_supported(int kind, int conv, int64_t M, int64_t N, int64_t K, int64_t kchunk, int cC) {
    if (POL(no_gemm8)) return 0;
    if (M % 256 || N % 256 || K % 64 || kchunk % 64 || kchunk < 256 || M / 256 > 65535) return 0;
    if (kind == 3) return (K % 128 || kchunk % 128 || kchunk < 512 || (conv && cC % 128) || POL(no_gemm8t)) ? 0 : 1;     // fp8 weight gradient
    if (conv && (cC % (kind == 2 ? 128 : 64))) return 0;
    if (kind == 2) return POL(no_gemm8t) ? 0 : 1;
    return (M / 256) * (N / 256) >= 192;
}
// nn.Linear products (kind 0 / 1, no gather, bf16): any M, N a multiple of 8 -- ragged last row / column tiles (r05).  WHETHER the shape
// should take this kernel is the caller's rule (gemm.hip: gemm_impl).
int gemm8_linear_ok(int64_t M, int64_t N, int64_t K, int64_t kchunk) {
    if (POL(no_gemm8)) return 0;
    return M >= 1 && N >= 8 && N % 8 == 0 && K % 64 == 0 && kchunk % 64 == 0 && kchunk >= 256 && (M + 255) / 256 <= 65535;
}
int gemm8_launch(int kind, int conv, int fp8, int64_t M, int64_t N, int64_t K, int64_t kchunk, int split_k, const void* A, int64_t lda,
                 const void* B, int64_t ldb, void* C, int64_t ldc, int cH, int cW, int cC, int csign, const float* f8_sa,
                 const float* f8_sb, const float* bias, const void* residual, int64_t ldr, const float* rscale, int64_t rpg, float* ws,
                 hipStream_t st) {
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 || (lda * (kind == 3 ? 1 : 2)) % 16 || (ldb * (kind == 3 ? 1 : 2)) % 16 || ldc % 4) return SEGF_ERR_SHAPE;
    if (residual && (((uintptr_t)residual % 8) || ldr % 4)) return SEGF_ERR_SHAPE;
    if (kind < 2 && split_k != 1 && !(conv && kind == 0 && !fp8 && ws)) return SEGF_ERR_SHAPE;      // (the gathered forward has a split-K form)
    if (M % 256 && !(kind == 0 || (kind == 1 && !conv))) return SEGF_ERR_SHAPE;      // ragged last row tile: the gathered forward and the plain products
    if (N % 256 && (conv || kind >= 2 || fp8 || N % 8)) return SEGF_ERR_SHAPE;        // ragged last column tile: plain bf16 products only
    if (conv && kind < 2 && kchunk != K && (fp8 || split_k < 2 || !ws || bias || residual)) return SEGF_ERR_SHAPE;      // gathered forward / data gradient: all of K, or fp32 split-K partials (bf16 operands, plain epilogue)
    if (conv && kind < 2 && ((int64_t)(cW + 1) * lda * 2 >= (1ll << 31) || (int64_t)9 * cC * 2 >= (1ll << 31))) return SEGF_ERR_SHAPE;      // 32-bit tap offsets in the gather
    Gemm8Args a{(const unsigned char*)A, (const unsigned char*)B, C, M, N, K, lda, ldb, ldc, kchunk, cH, cW, cC, csign, f8_sa, f8_sb, bias,
                (const bf16_t*)residual, ldr, rscale, rpg > 0 ? rpg : 1, split_k > 1 ? ws : nullptr, 1, 1, nullptr};
    {   // device address of the zero page, looked up once per process (a plain pointer: a repeated first lookup is harmless)
        static const unsigned char* zero_page = nullptr;
        if (!zero_page && !segf_trace().dry) {
            void* zp = nullptr;
            if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g8_zero_page)) != hipSuccess || !zp) return SEGF_ERR_SHAPE;
            zero_page = (const unsigned char*)zp;
        }
        a.zero = zero_page;
    }
    // fp8 operands: on some MI355X devices the staggered schedule (26 % fewer cycles) makes the chip drop its clock from 2.4 to 1.5 GHz
    // and ends up SLOWER than the lockstep one (12.7 vs 10.9 ms on the UPerHead bottleneck; 8.3 ms on devices that hold their clock).
    // g8_stagger_fp8 is set per process by the host layer after timing both on the device at hand (hip.py: autotune_gemm8_fp8).
    if (fp8) a.stagger = POL(g8_stagger_fp8);        // (segf_gemm8_option, policy.hip)
    if (POL(g8_stagger) >= 0) a.stagger = POL(g8_stagger);
    const dim3 grid((unsigned)((N + 255) / 256), (unsigned)((M + 255) / 256), (unsigned)split_k);
#define G8_GO(...) hipLaunchKernelGGL((gemm8_kernel<__VA_ARGS__>), grid, dim3(512), 0, st, a)
    if (kind == 0) {
        if (conv && fp8 == 0 && split_k > 1) G8_GO(0, 0, true, 0, float);          // split-K partials [z][M][N] in ws (summed by the caller)
        else if (conv) { if (fp8 == 0) G8_GO(0, 0, true, 0, bf16_t); else if (fp8 == 1) G8_GO(0, 0, true, 1, bf16_t); else G8_GO(0, 0, true, 2, bf16_t); }
        else { if (fp8 == 0) G8_GO(0, 0, false, 0, bf16_t); else if (fp8 == 1) G8_GO(0, 0, false, 1, bf16_t); else G8_GO(0, 0, false, 2, bf16_t); }
    } else if (kind == 1) {
        if (conv || fp8) return SEGF_ERR_SHAPE;
        G8_GO(0, 1, false, 0, bf16_t);
    } else if (kind == 2) {
        if (fp8) return SEGF_ERR_SHAPE;
        if (conv) G8_GO(1, 1, true, 0, float); else G8_GO(1, 1, false, 0, float);
    } else {                                   // kind 3: weight gradient on fp8 operands (A = dy e5m2 [P][M], B = x e4m3 [P][Cin]; strides in bytes)
        if (!f8_sa || !f8_sb) return SEGF_ERR_SHAPE;
        if (conv) G8_GO(2, 2, true, 2, float); else G8_GO(2, 2, false, 2, float);
    }
#undef G8_GO
    SEGF_CHECK_LAUNCH();
    return 0;
}
