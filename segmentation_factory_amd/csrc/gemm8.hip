// 256 x 256 x 64 bf16 / fp8 GEMM tile in EIGHT PHASES per two K tiles, for forward-type products with both operands K-contiguous
// (C[m][n] = sum_k A[m][k] B[n][k]) -- the implicit-GEMM 3x3 convolutions of UPerHead / PPM (forward and data gradient;
// heads/upernet.py:26-31, modules/ppm.py:19: 75 % of BASELINE cfg3's step, 42 % of cfg5's).
//
// The two-phase kernel of gemm.hip (load -> barrier -> 64 MFMAs per wave -> LDS write -> barrier, one workgroup per CU) reaches
// 0.79 PFLOP/s on these shapes: every K step pays its LDS write pass (ds_write_b128: 79 B/clk per CU), its barrier and the drain of
// its loads in series with the matrix instructions.  Structure here (cdna_hip_programming.md, "The 256^2 8-phase template"):
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass.  The LDS image is
//     lane-linear per instruction (8 rows x 128 B), so the XOR swizzle that makes the fragment reads conflict-free is applied to the
//     GLOBAL address of each lane (logical chunk = physical chunk ^ (row & 7));
//   * a K tile is four HALF-tiles of 16 KB (A rows 0-127 / 128-255, B rows 0-127 / 128-255 of the workgroup tile); a wave's 128 x 64
//     output is two 64-row pieces x two 32-column pieces, one in each half, so its four C QUADRANTS (m0,n0) (m0,n1) (m1,n1) (m1,n0) read
//     A0 B0 | A0 B1 | A1 B1 | A1 B0: each phase = one quadrant x K = 64 = 16 MFMAs, loads 8 (A) and / or 4 (B) fragments, and stages
//     ONE half-tile, in the order in which the halves fall free: B0[kt+1], A1[kt+1], A0[kt+2], B1[kt+2] -- three half-tiles
//     (1.5 K tiles) in flight across the barriers, retired by COUNTED waits (vmcnt(8) in phase 2, vmcnt(6) in phase 4), raw
//     s_barrier (a __syncthreads() would drain the DMA queue), one phase between a wait and the first read of what it retired;
//   * implicit convolution: a K tile of 64 units lies inside one tap (channels % 64 == 0), so the gather is one wave-uniform pixel
//     offset per K tile plus a per-lane border test; taps outside the image read a ZERO PAGE (LDS-DMA cannot write zeros itself);
//   * fp8 operands exactly as in gemm.hip's FP8 mode: 2-byte units along K, two v_mfma_f32_16x16x32_fp8 per 16-byte fragment.
#include <stdlib.h>
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 g8_bf16x8;
typedef __attribute__((ext_vector_type(4))) float g8_f32x4;

struct Gemm8Args {
    const unsigned char* A; const unsigned char* B; bf16_t* C;
    int64_t M, N, K;                 // K, lda, ldb, cC in 2-byte units
    int64_t lda, ldb, ldc;
    int cH, cW, cC, csign;           // CONV: NHWC geometry of the gathered operand A
    const float* f8_sa; const float* f8_sb;
    const float* bias;
};
// the zero page of the CONV border taps: LDS-DMA cannot write zeros itself, so lanes whose tap lies outside the image load from here
__device__ __attribute__((aligned(256))) unsigned char g8_zero_page[256];

#define G8_HALF 16384
#define G8_BUF 65536

template <bool CONV, int FP8>
__global__ void __launch_bounds__(512) gemm8_kernel(Gemm8Args a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * G8_BUF];       // [buf][A0, A1, B0, B1][128 rows][128 B]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned nwg = gx * gy;
    const unsigned orig = blockIdx.x + gx * blockIdx.y;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
    const unsigned wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);      // bijective XCD remap
    const unsigned bx = wgid % gx, by = wgid / gx;
    const int64_t m0 = (int64_t)by * 256, n0 = (int64_t)bx * 256;
    const int nk = (int)(a.K / 64);

    // ---- staging: this wave stages rows 16 wave + 8 i + (lane >> 3) (i = 0, 1) of every half-tile; physical chunk lane & 7 holds the
    // logical chunk (lane & 7) ^ (row & 7), row & 7 = lane >> 3
    const int srow = 16 * wave + (lane >> 3), lchunk = (lane & 7) ^ (lane >> 3);
    // B operand: plain rows
    const unsigned char* gB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) gB[h][i] = a.B + ((n0 + 128 * h + srow + 8 * i) * a.ldb + 8 * lchunk) * 2;
    // A operand: plain rows, or the pixel of each row for the gather
    const unsigned char* gA[2][2];
    int ay[2][2], ax[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t m = m0 + 128 * h + srow + 8 * i;
            if (CONV) {
                const int x = (int)(m % a.cW);
                const int64_t t = m / a.cW;
                ay[h][i] = (int)(t % a.cH); ax[h][i] = x;
            }
            gA[h][i] = a.A + (m * a.lda + 8 * lchunk) * 2;
        }
    auto lds_half = [&](int buf, int half) -> unsigned char* { return smem + buf * G8_BUF + half * G8_HALF; };
    // half: 0 = A0, 1 = A1, 2 = B0, 3 = B1.  K tile index clamped to the last one (dummy loads keep the counted waits uniform)
    auto stage = [&](int kt, int half) {
        const int ktc = kt < nk ? kt : nk - 1;
        unsigned char* dst = lds_half(kt & 1, half) + (16 * wave) * 128;     // (kt, not ktc: a dummy load lands where its tile WOULD go, never on live data)
        if (half >= 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB[half - 2][i] + (int64_t)ktc * 128),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        } else if (!CONV) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA[half][i] + (int64_t)ktc * 128),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        } else {
            const int k0 = ktc * 64, tap = k0 / a.cC, ch0 = k0 - tap * a.cC;           // wave-uniform
            const int dy = a.csign * (tap / 3 - 1), dx = a.csign * (tap % 3 - 1);
            const int64_t off = (((int64_t)dy * a.cW + dx) * a.lda + ch0) * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yy = ay[half][i] + dy, xx = ax[half][i] + dx;
                const bool ok = yy >= 0 && yy < a.cH && xx >= 0 && xx < a.cW;
                const unsigned char* p = ok ? gA[half][i] + off : g8_zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
            }
        }
    };
    // ---- fragments: lane (i = lane & 15, g = lane >> 4) of row tile `rt` (16 rows) of a half-tile, K sub-step s
    const int fi = lane & 15, fg = lane >> 4;
    auto frag = [&](const unsigned char* half, int row0, int s) -> g8_bf16x8 {
        const int row = row0 + fi;
        return *reinterpret_cast<const g8_bf16x8*>(half + row * 128 + (((4 * s + fg) ^ (row & 7)) << 4));
    };
    g8_f32x4 acc[4][8];                // [column tile: 2 nq + u][row tile: 4 mq + t]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = g8_f32x4{0.f, 0.f, 0.f, 0.f};
    g8_bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    auto load_a = [&](int buf, int mq) {
        const unsigned char* h = lds_half(buf, mq);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) fa[t][s] = frag(h, 64 * wm + 16 * t, s);
    };
    auto load_b = [&](int buf, int nq, g8_bf16x8 (&fb)[2][2]) {
        const unsigned char* h = lds_half(buf, 2 + nq);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) fb[u][s] = frag(h, 32 * wn + 16 * u, s);
    };
    auto mma = [&](int mq, int nq, const g8_bf16x8 (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    g8_f32x4 c = acc[2 * nq + u][4 * mq + t];
                    if constexpr (FP8 == 0) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u][s], fa[t][s], c, 0, 0, 0);
                    } else {
                        typedef long g8_l2 __attribute__((ext_vector_type(2)));
                        const g8_l2 wb = __builtin_bit_cast(g8_l2, fb[u][s]), xa = __builtin_bit_cast(g8_l2, fa[t][s]);
                        if constexpr (FP8 == 1) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wb[0], xa[0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wb[1], xa[1], c, 0, 0, 0);
                        } else {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(wb[0], xa[0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_bf8(wb[1], xa[1], c, 0, 0, 0);
                        }
                    }
                    acc[2 * nq + u][4 * mq + t] = c;
                }
        __builtin_amdgcn_s_setprio(0);
    };
#define G8_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define G8_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

    // ---- prologue: K tiles 0 and 1 in the steady-state issue order A0, B1, B0, A1 | A0, B1 (B0[1], A1[1] follow in phases 1, 2)
    stage(0, 0); stage(0, 3); stage(0, 2); stage(0, 1); stage(1, 0); stage(1, 3);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                 // all of K tile 0 has landed (this wave's share)
    G8_BAR();
    for (int kt = 0; kt < nk; ++kt) {
        const int b = kt & 1;
        // phase 1: quadrant (m0, n0)
        load_a(b, 0); load_b(b, 0, fb0);
        stage(kt + 1, 2);
        G8_BAR(); G8_LGKM0();
        mma(0, 0, fb0);
        G8_BAR();
        // phase 2: quadrant (m0, n1)
        load_b(b, 1, fb1);
        stage(kt + 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // A1[kt] (read in phase 3) -- already true in steady state by two phases
        G8_BAR(); G8_LGKM0();
        mma(0, 1, fb1);
        G8_BAR();
        // phase 3: quadrant (m1, n1)
        load_a(b, 1);
        stage(kt + 2, 0);
        G8_BAR(); G8_LGKM0();
        mma(1, 1, fb1);
        G8_BAR();
        // phase 4: quadrant (m1, n0)
        stage(kt + 2, 3);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");             // A0[kt+1], B1[kt+1], B0[kt+1] have landed: read from the next phase on
        G8_BAR();
        mma(1, 0, fb0);
        G8_BAR();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- epilogue: acc[2 nq + u][4 mq + t][r] = C[m0 + 128 mq + 64 wm + 16 t + fi][n0 + 128 nq + 32 wn + 16 u + 4 fg + r]
    const float sa = FP8 ? a.f8_sa[0] : 1.f;
#pragma unroll
    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t n = n0 + 128 * nq + 32 * wn + 16 * u + 4 * fg;
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
                    const g8_f32x4 c = acc[2 * nq + u][4 * mq + t];
                    const float v0 = fmaf(c[0], sc[0], bs[0]), v1 = fmaf(c[1], sc[1], bs[1]), v2 = fmaf(c[2], sc[2], bs[2]), v3 = fmaf(c[3], sc[3], bs[3]);
                    *reinterpret_cast<uint2*>(a.C + m * a.ldc + n) = make_uint2(pack2bf(v0, v1), pack2bf(v2, v3));
                }
        }
}

// shapes the eight-phase kernel takes: whole 256 x 256 tiles, whole 64-unit K tiles (inside one tap for the convolution), at least four
int gemm8_supported(int conv, int64_t M, int64_t N, int64_t K, int cC) {
    if (getenv("SEGFAC_NO_GEMM8")) return 0;
    if (M % 256 || N % 256 || K % 64 || K < 256 || M / 256 > 65535) return 0;
    if (conv && (cC % 64)) return 0;
    return (M / 256) * (N / 256) >= 192;
}
int gemm8_launch(int conv, int fp8, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb, void* C,
                 int64_t ldc, int cH, int cW, int cC, int csign, const float* f8_sa, const float* f8_sb, const float* bias, hipStream_t st) {
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) % 16 || (lda * 2) % 16 || (ldb * 2) % 16 || ldc % 4) return SEGF_ERR_SHAPE;
    Gemm8Args a{(const unsigned char*)A, (const unsigned char*)B, (bf16_t*)C, M, N, K, lda, ldb, ldc, cH, cW, cC, csign, f8_sa, f8_sb, bias};
    const dim3 grid((unsigned)(N / 256), (unsigned)(M / 256));
#define G8_GO(CONV_, FP8_) hipLaunchKernelGGL((gemm8_kernel<CONV_, FP8_>), grid, dim3(512), 0, st, a)
    if (conv) { if (fp8 == 0) G8_GO(true, 0); else if (fp8 == 1) G8_GO(true, 1); else G8_GO(true, 2); }
    else { if (fp8 == 0) G8_GO(false, 0); else if (fp8 == 1) G8_GO(false, 1); else G8_GO(false, 2); }
#undef G8_GO
    SEGF_CHECK_LAUNCH();
    return 0;
}

// =====================================================================================================================================
// The same eight-phase schedule for the WEIGHT GRADIENT of the 3x3 convolution (both operands reduction-major):
//   dW[co][tap * Cin + ci] = sum_pix dy[pix][co] * x[pix + off(tap)][ci]        (K = pixels; split over grid.z, fp32 partials)
// Half-tiles are [64 pixels][128 columns] (256-byte rows): A0 / A1 = dy columns co, B0 / B1 = x columns (tap, ci) -- Cin % 128 == 0,
// so a half-tile lies inside ONE tap and the gather is a workgroup-constant pixel offset plus a per-row border test.  One LDS-DMA
// instruction moves 4 rows x 256 B; the reduction-major XOR swizzle of gemm.hip (rm_swz) goes on the source address; fragments
// come out of the image through ds_read_b64_tr_b16 (tokens become the contiguous k of the MFMA operands).
__device__ __forceinline__ int g8_rm_swz(int krow) { return 2 * ((krow & 3) + 4 * ((krow >> 3) & 1)); }
// Inline asm, not the builtin: hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every __builtin_amdgcn_ds_read_tr16_b64 while an
// LDS-DMA is in flight (it cannot tell that the read does not alias the pending LDS writes; plain ds_read_b128 loads are
// disambiguated), which drains the three half-tiles the schedule keeps in flight.  The consumer side is ordered by hand:
// s_waitcnt lgkmcnt(0) + sched_barrier(0) before the MFMAs (G8_LGKM0).
__device__ __forceinline__ g8_bf16x8 g8_frag_tr(const unsigned char* tile, int cb, int s, int lane) {
    typedef __attribute__((ext_vector_type(4))) short g8_s16x4;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int u = (cb >> 2) + p;
    const int chunk = u >> 1, half = u & 1;
    const int k0 = 32 * s + 8 * g + q, k1 = k0 + 4;
    const uint32_t a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(tile + k0 * 256 + ((chunk ^ g8_rm_swz(k0)) << 4) + half * 8);
    const uint32_t a1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(tile + k1 * 256 + ((chunk ^ g8_rm_swz(k1)) << 4) + half * 8);
    g8_s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
    return __builtin_bit_cast(g8_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
struct Gemm8TArgs {
    const bf16_t* A; const bf16_t* B; float* C;      // A = dy [P][lda], B = x [P][ldb], C = dW fp32 [M][ldc] or split-K slabs
    int64_t M, N, K, lda, ldb, ldc, kchunk;
    int cH, cW, cC;
    float* ws;                                        // [z][M][N] when gridDim.z > 1
};
__global__ void __launch_bounds__(512) gemm8t_kernel(Gemm8TArgs a) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * G8_BUF];       // [buf][A0, A1, B0, B1][64 k rows][256 B]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int64_t m0 = (int64_t)blockIdx.y * 256, n0 = (int64_t)blockIdx.x * 256;
    const int64_t kbeg = (int64_t)blockIdx.z * a.kchunk;
    const int64_t kend = kbeg + a.kchunk < a.K ? kbeg + a.kchunk : a.K;
    const int nk = (int)((kend - kbeg) / 64);
    // staging: rows 8 wave + 4 i + (lane >> 4) of every half-tile, physical chunk lane & 15 = logical chunk ^ rm_swz(row)
    const int srow = 8 * wave + (lane >> 4);
    int lch[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) lch[i] = (lane & 15) ^ g8_rm_swz(srow + 4 * i);
    // x gather: tap of each B half (workgroup constant), pixel coordinates of this lane's two rows at the current K tile
    int tdy[2], tdx[2], tci[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t n = n0 + 128 * h;
        const int tap = (int)(n / a.cC);
        tdy[h] = tap / 3 - 1; tdx[h] = tap % 3 - 1; tci[h] = (int)(n - (int64_t)tap * a.cC);
    }
    auto lds_half = [&](int buf, int half) -> unsigned char* { return smem + buf * G8_BUF + half * G8_HALF; };
    // Pixel coordinates of this lane's two rows for the NEXT K tile of each B half (the tiles of a half are staged in increasing order:
    // 0, 1, 2, ...), advanced by 64 pixels per tile without a division
    int py[2][2], px[2][2];
    {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t t = kbeg + srow + 4 * i;
            const int x = (int)(t % a.cW), y = (int)((t / a.cW) % a.cH);
#pragma unroll
            for (int h = 0; h < 2; ++h) { py[h][i] = y; px[h][i] = x; }
        }
    }
    const int adv_q = 64 / a.cW, adv_r = 64 % a.cW;
    auto stage = [&](int kt, int half) {
        const int ktc = kt < nk ? kt : nk - 1;
        unsigned char* dst = lds_half(kt & 1, half) + (8 * wave) * 256;       // (kt, not ktc: a dummy load must not land on the last tile's live data)
        const int64_t t0 = kbeg + (int64_t)ktc * 64 + srow;
        if (half < 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.A + (t0 + 4 * i) * a.lda + m0 + 128 * half + 8 * lch[i]),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        } else {
            const int h = half - 2;
            const bool real = kt < nk;                                  // wave-uniform: past the end only a dummy load (zero page) is issued
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yy = py[h][i] + tdy[h], xx = px[h][i] + tdx[h];
                const bool ok = real && yy >= 0 && yy < a.cH && xx >= 0 && xx < a.cW;
                const bf16_t* p = a.B + (t0 + 4 * i + (int64_t)tdy[h] * a.cW + tdx[h]) * a.ldb + tci[h] + 8 * lch[i];
                const void* src = ok ? (const void*)p : (const void*)g8_zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                // advance to the next K tile of this half
                int nx = px[h][i] + adv_r, ny = py[h][i] + adv_q;
                if (nx >= a.cW) { nx -= a.cW; ++ny; }
                while (ny >= a.cH) ny -= a.cH;
                px[h][i] = nx; py[h][i] = ny;
            }
        }
    };
    g8_f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = g8_f32x4{0.f, 0.f, 0.f, 0.f};
    g8_bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    auto load_a = [&](int buf, int mq) {
        const unsigned char* h = lds_half(buf, mq);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 2; ++s) fa[t][s] = g8_frag_tr(h, 64 * wm + 16 * t, s, lane);
    };
    auto load_b = [&](int buf, int nq, g8_bf16x8 (&fb)[2][2]) {
        const unsigned char* h = lds_half(buf, 2 + nq);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 2; ++s) fb[u][s] = g8_frag_tr(h, 32 * wn + 16 * u, s, lane);
    };
    auto mma = [&](int mq, int nq, const g8_bf16x8 (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[2 * nq + u][4 * mq + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[u][s], fa[t][s], acc[2 * nq + u][4 * mq + t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    stage(0, 0); stage(0, 3); stage(0, 2); stage(0, 1); stage(1, 0); stage(1, 3);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    G8_BAR();
    for (int kt = 0; kt < nk; ++kt) {
        const int b = kt & 1;
        load_a(b, 0); load_b(b, 0, fb0);
        stage(kt + 1, 2);
        G8_BAR(); G8_LGKM0();
        mma(0, 0, fb0);
        G8_BAR();
        load_b(b, 1, fb1);
        stage(kt + 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        G8_BAR(); G8_LGKM0();
        mma(0, 1, fb1);
        G8_BAR();
        load_a(b, 1);
        stage(kt + 2, 0);
        G8_BAR(); G8_LGKM0();
        mma(1, 1, fb1);
        G8_BAR();
        stage(kt + 2, 3);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        G8_BAR();
        mma(1, 0, fb0);
        G8_BAR();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int fi = lane & 15, fg = lane >> 4;
    float* out = a.ws ? a.ws + (int64_t)blockIdx.z * a.M * a.N : a.C;
    const int64_t ldo = a.ws ? a.N : a.ldc;
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
                    *reinterpret_cast<g8_f32x4*>(out + m * ldo + n) = acc[2 * nq + u][4 * mq + t];
                }
        }
}

int gemm8t_supported(int64_t M, int64_t N, int64_t K, int64_t kchunk, int cC) {
    if (getenv("SEGFAC_NO_GEMM8") || getenv("SEGFAC_NO_GEMM8T")) return 0;
    if (M % 256 || N % 256 || K % 64 || kchunk % 64 || kchunk < 256 || cC % 128) return 0;
    return 1;
}
int gemm8t_launch(int64_t M, int64_t N, int64_t K, int64_t kchunk, int split_k, const void* dy, int64_t lda, const void* x, int64_t ldb,
                  float* C, int64_t ldc, float* ws, int cH, int cW, int cC, hipStream_t st) {
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)C) % 16 || (lda * 2) % 16 || (ldb * 2) % 16 || ldc % 4) return SEGF_ERR_SHAPE;
    Gemm8TArgs a{(const bf16_t*)dy, (const bf16_t*)x, C, M, N, K, lda, ldb, ldc, kchunk, cH, cW, cC, split_k > 1 ? ws : nullptr};
    hipLaunchKernelGGL(gemm8t_kernel, dim3((unsigned)(N / 256), (unsigned)(M / 256), (unsigned)split_k), dim3(512), 0, st, a);
    SEGF_CHECK_LAUNCH();
    return 0;
}
