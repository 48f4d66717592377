// MFMA kernels for the MiT spatial-reduction attention core, bf16 storage / fp32 accumulate
// (reference models/backbones/mit.py:52-57: attn = softmax(q k^T * scale); x = attn v).
//
// Everything is computed "transposed" so that the softmax axis lies along accumulator REGISTERS and the query index along
// LANES (v_mfma_f32_16x16x32_bf16, C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg):
//   S^T [key][query] = K Q^T     A = K rows (16 B per lane from the LDS tile), B = Q rows (registers, loaded once)
//   O^T [d][query]  += V^T P^T   B = P^T taken straight from the S^T accumulators (no lane movement, no LDS round trip):
//                                k-slot j of lane group g is key 4g+j (j<4, first 16-key tile) / 16+4g+(j-4) (second tile);
//                                A = V^T read from the row-major [key][d] LDS tile with ds_read_b64_tr_b16 in that same order.
// Online softmax over 32-key steps; per-query statistics live one per lane (+ two ds_bpermute hops across the 4 lane groups).
// K/V are staged through LDS in chunks of AM_KC keys, so Nkv = 256 (512^2 inputs) is one stage and Nkv = 2048
// (1024x2048, MiT-B2) is eight.  The backward is two kernels sharing the same fragment algebra:
//   dq  : S^T form again, dS^T feeds dQ^T += K^T dS^T directly from the accumulators
//   dkv : S form (rows = queries), each wave owns 64 keys whose K/V fragments stay in registers for the whole kernel;
//         Q / dO tiles go through LDS; dV^T += dO^T P and dK^T += Q^T dS take P / dS from the accumulators.
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "attention_mfma.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define AM_KC_OF(HD) ((HD) == 64 ? 128 : 256)      // keys per LDS stage: 32 KB of K + V either way (3+ workgroups per CU)
#define AM_THREADS 256
// waves per SIMD the query-side kernels are compiled for: head dim 64 with swizzled tiles keeps six fragment addresses live and spilled
// 150 - 250 bytes under the 128-register cap of four waves per SIMD; three waves (168 registers) hold everything
#ifndef AM_QOCC64
#define AM_QOCC64 3
#endif
#define AM_QOCC(HD) ((HD) == 64 ? AM_QOCC64 : 4)

__device__ __forceinline__ bf16x8 ld_frag_global(const bf16_t* __restrict__ p, bool valid) {
    uint4 u = make_uint4(0, 0, 0, 0);
    if (valid) u = *reinterpret_cast<const uint4*>(p);
    return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ bf16x8 ld_frag_lds(const bf16_t* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
}
// transposed fragment from a row-major [row][HD] bf16 LDS tile: element j<4 <- row r_lo + j, j>=4 <- row r_hi + (j-4),
// column c_base + (lane & 15)
template <int HD>
__device__ __forceinline__ bf16x8 ld_frag_tr(const bf16_t* tile, int r_lo, int r_hi, int c_base, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3;
    const bf16_t* a = tile + (r_lo + q) * HD + c_base + 4 * p;
    const bf16_t* b = tile + (r_hi + q) * HD + c_base + 4 * p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)b);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// LDS tiles at head dim 64 (128-byte rows) are XOR-SWIZZLED: 16-byte chunk ch of row r sits at chunk ch ^ (r & 7).  Unswizzled, the
// 16 rows of a ds_read_b128 fragment read (same column, rows 128 bytes apart) fall on 2 of the 16 sixteen-byte slots of the 256-byte
// bank row (8-way conflict) and the four same-parity rows of a transposed read on one (4-way): 33 % / 41 % / 49 % of the CU cycles of
// the forward / query-side / key-side kernels at 2048 keys were LDS bank-conflict cycles (profiles/r04_mfma_kernels_pmc_before.txt).
// With the swizzle both kinds of read are conflict-free (row-major reads: slot = (r & 1, ch ^ (r & 7)), distinct over a 16-lane group;
// transposed reads: the four rows of one parity land on four different chunk pairs).  Head dim 32 keeps its dense 64-byte rows.
template <int HD, bool SWZ = true>
__device__ __forceinline__ int am_chunk(int ch, int row_low3) { return (HD == 64 && SWZ) ? (ch ^ row_low3) : ch; }
// transposed fragment from a swizzled row-major [row][HD] tile; rows r_lo + q and r_hi + q with r_lo = r_hi = 4 g (mod 8),
// 16-column block d: element j < 4 <- row r_lo + j, j >= 4 <- row r_hi + (j - 4), column 16 d + (lane & 15)
template <int HD, bool SWZ = true>
__device__ __forceinline__ bf16x8 ld_frag_trs(const bf16_t* tile, int r_lo, int r_hi, int d, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4;
    const int col = (am_chunk<HD, SWZ>(2 * d + (p >> 1), (4 * g + q) & 7) << 3) + 4 * (p & 1);
    const bf16_t* a = tile + (r_lo + q) * HD + col;
    const bf16_t* b = tile + (r_hi + q) * HD + col;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)b);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// fragment times a power of two (exact in bf16): head dim 64 has scale = 2^-3, which then rides on an operand instead of on
// every score (S (q s) = s S (q) bit for bit; the gradients are rescaled once at the end)
__device__ __forceinline__ bf16x8 scale_frag_pow2(bf16x8 f, float s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)((float)f[j] * s);
    return r;
}
// two accumulator tiles (rows 4g+r of the first / second 16-row tile) -> one bf16 operand fragment
__device__ __forceinline__ bf16x8 pack_acc(const float (&a)[4], const float (&b)[4]) {
    uint4 u;
    u.x = pack2bf(a[0], a[1]); u.y = pack2bf(a[2], a[3]); u.z = pack2bf(b[0], b[1]); u.w = pack2bf(b[2], b[3]);
    return __builtin_bit_cast(bf16x8, u);
}
// reductions over the four 16-lane groups of a wave (lanes i, i + 16, i + 32, i + 48), result in all four: v_permlane16_swap /
// v_permlane32_swap exchange whole 16- / 32-lane rows between two registers in the VALU -- with both operands holding v, one register
// comes back as (row 0, row 0, row 2, row 2) and the other as (row 1, row 1, row 3, row 3) (resp. lower half twice / upper half twice).
// __shfl_xor(v, 16 / 32) compiled to ds_bpermute_b32: two LDS round trips (~100 cycles each) in the middle of the online-softmax
// dependency chain of every 32-key step.  Same values in the same association, bit for bit (max and + are commutative).
// (inline asm: hipcc 7.2 miscompiles the two-result builtins when both results feed arithmetic, loss.hip; the s_nops are the
// VALU-write -> permlane-read wait states)
__device__ __forceinline__ void am_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32_e32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void am_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float xgroup_max(float v) {
    float a = v, b = v;
    am_swap16(a, b); v = fmaxf(a, b);
    a = v; b = v;
    am_swap32(a, b); return fmaxf(a, b);
}
__device__ __forceinline__ float xgroup_sum(float v) {
    float a = v, b = v;
    am_swap16(a, b); v = a + b;
    a = v; b = v;
    am_swap32(a, b); return a + b;
}

// stage rows [r0, r0 + nrows) x HD of a [rows][ld] global matrix into a dense [nrows][HD] LDS tile (zero beyond rmax)
template <int HD, bool SWZ = true>
__device__ __forceinline__ void stage_rows(bf16_t* tile, const bf16_t* __restrict__ src, int64_t ld, int r0, int nrows, int rmax) {
    constexpr int CPR = HD / 8;                      // 16-byte chunks per row
    for (int i = threadIdx.x; i < nrows * CPR; i += AM_THREADS) {
        const int r = i / CPR, c = i - r * CPR;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (r0 + r < rmax) u = *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + r) * ld + c * 8);
        *reinterpret_cast<uint4*>(tile + r * HD + am_chunk<HD, SWZ>(c, r & 7) * 8) = u;
    }
}

// ---- forward ---------------------------------------------------------------------------------------------------------
// P2S: `scale` is a power of two (head dim 64: 2^-3) and rides on the Q fragments (exact), so the scores leave the MFMA already scaled
template <int HD, int QW, bool P2S, int OCC = 4, bool SWZ = true>
__global__ void __launch_bounds__(AM_THREADS, OCC) attn_mfma_fwd_kernel(const bf16_t* __restrict__ q, int64_t ldq,
                                                                    const bf16_t* __restrict__ k, int64_t ldk,
                                                                    const bf16_t* __restrict__ v, int64_t ldv,
                                                                    bf16_t* __restrict__ o, int64_t ldo, float* __restrict__ lse,
                                                                    int heads, int N, int Nkv, float scale) {
    constexpr int AM_KC = AM_KC_OF(HD);
    __shared__ __attribute__((aligned(16))) bf16_t Ks[AM_KC * HD];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[AM_KC * HD];
    constexpr int KS = HD / 32, DT = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * (16 * QW);
    const bf16_t* Qb = q + (int64_t)b * N * ldq + h * HD;
    const bf16_t* Kb = k + (int64_t)b * Nkv * ldk + h * HD;
    const bf16_t* Vb = v + (int64_t)b * Nkv * ldv + h * HD;
    bf16x8 Qf[QW][KS];
#pragma unroll
    for (int t = 0; t < QW; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int row = q0 + 16 * t + c;
            Qf[t][s] = ld_frag_global(Qb + (int64_t)row * ldq + 32 * s + 8 * g, row < N);
            if (P2S) Qf[t][s] = scale_frag_pow2(Qf[t][s], scale);
        }
    f32x4 O[DT][QW];
    float m[QW], l[QW];
#pragma unroll
    for (int t = 0; t < QW; ++t) {
        m[t] = -INFINITY; l[t] = 0.f;
#pragma unroll
        for (int d = 0; d < DT; ++d) O[d][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int kc0 = 0; kc0 < Nkv; kc0 += AM_KC) {
        const int nk = Nkv - kc0 < AM_KC ? Nkv - kc0 : AM_KC;
        const int nk32 = (nk + 31) & ~31;
        __syncthreads();
        stage_rows<HD, SWZ>(Ks, Kb, ldk, kc0, nk32, Nkv);
        stage_rows<HD, SWZ>(Vs, Vb, ldv, kc0, nk32, Nkv);
        __syncthreads();
        if (q0 >= N) continue;                       // wave-uniform; the wave still takes part in the barriers
        for (int kb = 0; kb < nk; kb += 32) {
            bf16x8 Kf[2][KS], Vf[DT];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < KS; ++s) Kf[kt][s] = ld_frag_lds(Ks + (kb + 16 * kt + c) * HD + (am_chunk<HD, SWZ>(4 * s + g, c & 7) << 3));
#pragma unroll
            for (int d = 0; d < DT; ++d) Vf[d] = ld_frag_trs<HD, SWZ>(Vs, kb + 4 * g, kb + 16 + 4 * g, d, lane);
            // MASK: only the last 32-key step of a key count that is not a multiple of 32 has keys to hide (wave-uniform choice)
            auto step = [&](auto maskc) {
                constexpr bool MASK = decltype(maskc)::value;
#pragma unroll
                for (int t = 0; t < QW; ++t) {
                    float sv[2][4];
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        f32x4 S = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < KS; ++s) S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kf[kt][s], Qf[t][s], S, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sc = P2S ? S[r] : S[r] * scale;
                            if (MASK) {
                                const int key = kc0 + kb + 16 * kt + 4 * g + r;
                                sv[kt][r] = key < Nkv ? sc : -INFINITY;
                            } else {
                                sv[kt][r] = sc;
                            }
                        }
                    }
                    float mx = fmaxf(fmaxf(fmaxf(sv[0][0], sv[0][1]), fmaxf(sv[0][2], sv[0][3])),
                                     fmaxf(fmaxf(sv[1][0], sv[1][1]), fmaxf(sv[1][2], sv[1][3])));
                    mx = xgroup_max(mx);
                    const float mnew = fmaxf(m[t], mx);
                    const float alpha = __expf(m[t] - mnew);
                    float ps = 0.f;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { sv[kt][r] = __expf(sv[kt][r] - mnew); ps += sv[kt][r]; }
                    l[t] = l[t] * alpha + ps;
                    m[t] = mnew;
                    const bf16x8 Pf = pack_acc(sv[0], sv[1]);
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        O[d][t] *= alpha;
                        O[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Vf[d], Pf, O[d][t], 0, 0, 0);
                    }
                }
            };
            if (kc0 + kb + 32 <= Nkv) step(std::false_type{}); else step(std::true_type{});
        }
    }
    if (q0 >= N) return;
    bf16_t* Ob = o + (int64_t)b * N * ldo + h * HD;
#pragma unroll
    for (int t = 0; t < QW; ++t) {
        const float lt = xgroup_sum(l[t]);
        const float inv = 1.f / lt;
        const int row = q0 + 16 * t + c;
        if (row < N) {
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const uint2 u = make_uint2(pack2bf(O[d][t][0] * inv, O[d][t][1] * inv), pack2bf(O[d][t][2] * inv, O[d][t][3] * inv));
                *reinterpret_cast<uint2*>(Ob + (int64_t)row * ldo + 16 * d + 4 * g) = u;
            }
            if (g == 0) lse[((int64_t)b * heads + h) * N + row] = m[t] + __logf(lt);
        }
    }
}

// a power of two well inside the normal range: multiplying bf16 / fp32 values by it is exact
// ---- head dim 64, long key sequences (MiT-B2 at 1024 x 2048: 2048 keys): K / V stages by LDS-DMA, double-buffered ------------------
// The kernels above stage 128 keys, wait for them, compute, and stage again: two barriers and one exposed global-load latency per
// stage, covered only by the other workgroups of the CU.  Here a stage is 64 keys (K 8 KB + V 8 KB); the NEXT stage's sixteen 1-KB
// pieces are issued as global_load_lds_dwordx4 (four per wave) into the other buffer right after the one barrier per stage, and land
// under the current stage's 32 matrix instructions per wave.  An LDS-DMA piece is lane-linear in LDS (wave-uniform base + 16 lane
// bytes = 8 rows of 128 bytes), so the XOR swizzle of the tiles is applied to the per-lane SOURCE address: lane l of a piece fetches
// logical chunk (l & 7) ^ (l >> 3) of row (l >> 3).  Rows beyond the last key are clamped (their scores are masked).
// transposed fragment as ld_frag_trs<64, true>, but INLINE ASM: in front of the ds_read_tr builtin hipcc 7.2 waits vmcnt(0) while an
// LDS-DMA is in flight (it cannot tell the DMA's LDS destination from the read's source), which would drain the prefetch at the first
// read of every stage.  The compiler does not track these reads: the caller waits (AMP_LGKM0) before the first use.
__device__ __forceinline__ bf16x8 amp_frag_trs(const bf16_t* tile, int r_lo, int r_hi, int d, int lane) {
    const int i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4;
    const int col = (((2 * d + (p >> 1)) ^ ((4 * g + q) & 7)) << 3) + 4 * (p & 1);
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(tile + (r_lo + q) * 64 + col);
    const uint32_t b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)(tile + (r_hi + q) * 64 + col);
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(b));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
#define AMP_LGKM0() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define AMP_KC 64
#define AMP_GLDS(SRC, DST) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC), \
                                                             (__attribute__((address_space(3))) void*)(DST), 16, 0, 0)
__device__ __forceinline__ void amp_stage(bf16_t* Kt, bf16_t* Vt, const bf16_t* __restrict__ Kb, int64_t ldk, const bf16_t* __restrict__ Vb,
                                          int64_t ldv, int kc0, int Nkv, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = 4 * i + wave;                                  // 8 rows x 128 bytes each
        int row = kc0 + 8 * piece + (lane >> 3);
        row = row < Nkv ? row : Nkv - 1;
        const int col = ((lane & 7) ^ (lane >> 3)) * 8;
        AMP_GLDS(Kb + (int64_t)row * ldk + col, Kt + piece * 512);
        AMP_GLDS(Vb + (int64_t)row * ldv + col, Vt + piece * 512);
    }
}

// raw single instructions: the library forms add a canonicalising v_max in front of every fmaxf on an MFMA result and range fix-ups
// around exp2f -- the forward below is bound by VALU issue (85 % of the SIMD cycles, ~4.2 cycles per vector instruction), so every
// instruction per (query, key) pair counts
__device__ __forceinline__ float am_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float am_max2(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define AMP_THR 6.0f        // lazy rescale: the running maximum is only raised when a score exceeds it by more than 2^6 (log2 units)

// r05: every LDS address of the two kernels below is ONE per-lane base + an instruction immediate.  The stage loop is written out for
// the two buffers (the buffer index is a template constant), so a 32-key step issues no address arithmetic at all: the forward spent
// 16 of its 82 vector instructions per step rebuilding the eight swizzled transposed-read addresses from the buffer parity, the
// query-side backward 16 of 113 (ISA count, tools/probe/isa_loop.py).  The compiler does not track these reads (inline asm): the caller
// waits (AMP_LGKM0) before the first use.
template <int OFF>
__device__ __forceinline__ bf16x8 amp_lds128(uint32_t a) {
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(a), "i"(OFF));
    return r;
}
// transposed fragment: element j < 4 <- row r + j of the tile at OFF, j >= 4 <- row r + 16 + (j - 4) (16 rows of 128 bytes further on)
template <int OFF>
__device__ __forceinline__ bf16x8 amp_trs(uint32_t a) {
    s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "i"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "i"(OFF + 2048));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
template <int V> struct amp_ic { static constexpr int value = V; };
// per-lane LDS byte addresses inside a swizzled [64 keys][64] tile pair (K at +0, V at +0x2000; second buffer at +0x4000):
// kbase[s]: row-major fragment of row c, 16-byte chunk 4 s + g;  vbase[d]: transposed read of rows 4 g + (i >> 2), column block d
struct AmpBases { uint32_t k[2], v[4]; };
__device__ __forceinline__ AmpBases amp_bases(const bf16_t* tile0, int lane) {
    const int g = lane >> 4, c = lane & 15, qq = c >> 2, p = c & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)tile0;
    AmpBases r;
#pragma unroll
    for (int s = 0; s < 2; ++s) r.k[s] = lds0 + c * 128 + (((4 * s + g) ^ (c & 7)) << 4);
#pragma unroll
    for (int d = 0; d < 4; ++d) r.v[d] = lds0 + (4 * g + qq) * 128 + ((2 * d + (p >> 1)) ^ ((4 * g + qq) & 7)) * 16 + 8 * (p & 1);
    return r;
}
// the per-lane part of a stage's source addresses (8 rows x 128 bytes per piece, two pieces per wave and operand), as 32-bit element
// offsets from the stage's first row: the stage's own part (kc0 * ld) is wave-uniform and stays on the scalar unit
struct AmpSrc { int k[2], v[2]; };
__device__ __forceinline__ AmpSrc amp_src(int64_t ldk, int64_t ldv, int wave, int lane) {
    AmpSrc r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (4 * i + wave) + (lane >> 3), col = ((lane & 7) ^ (lane >> 3)) * 8;
        r.k[i] = row * (int)ldk + col; r.v[i] = row * (int)ldv + col;
    }
    return r;
}
template <int BUF>
__device__ __forceinline__ void amp_stage_fast(bf16_t (*KV)[2][AMP_KC * 64], const bf16_t* __restrict__ Kb, int64_t ldk, const bf16_t* __restrict__ Vb,
                                               int64_t ldv, int kc0, int Nkv, const AmpSrc& so, int wave, int lane) {
    if (kc0 + AMP_KC <= Nkv) {                      // (wave-uniform) a full stage: uniform row base + per-lane 32-bit offset
        const bf16_t* ku = Kb + (int64_t)kc0 * ldk;
        const bf16_t* vu = Vb + (int64_t)kc0 * ldv;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            AMP_GLDS(ku + so.k[i], KV[BUF][0] + (4 * i + wave) * 512);
            AMP_GLDS(vu + so.v[i], KV[BUF][1] + (4 * i + wave) * 512);
        }
    } else amp_stage(KV[BUF][0], KV[BUF][1], Kb, ldk, Vb, ldv, kc0, Nkv, wave, lane);       // last, partial stage: clamped rows
}

// Forward at head dim 64, long key sequences.  The softmax runs in the exp2 domain: p = exp2(s c - m), c = scale log2(e), m = the
// running maximum of s c.  PRE = false: c sits in the exponent's multiply-add (one v_fma + one v_exp per (query, key) pair).
// PRE = true (r05, policy attn64_prescale, OFF by default): c rides on the Q fragments (bf16(q c), rounded once when the fragments
// are loaded) and -m is the INITIAL VALUE of the score accumulators (four registers per query tile holding -m, rewritten only when m
// moves), so the matrix pipe delivers s c - m and the pair costs one v_exp: 66 -> 50 vector instructions per 32-key step, forward
// 0.603 -> 0.557 ms on 8 x 131072 x 2048 keys.  The price is one more bf16 rounding per q element: against an fp32 reference the
// error of O grows x 1.24 on unit-normal inputs and x 2.3 on four times larger queries (tools/probe/attn64_accuracy.py) -- still
// below what rounding the SCORE to bf16 costs (the reference under autocast, models/backbones/mit.py:52-54), but parity comes
// first, so the default keeps the multiply-add (PRE = false: the arithmetic of r04's kernel, bit for bit in O).
// The running maximum is raised LAZILY: only when some score of the step exceeds it by more than AMP_THR does the wave rescale O and
// l (a wave-uniform branch); until then p <= 2^AMP_THR, which the fp32 accumulators and the bf16 P operand (a floating-point format:
// its relative precision does not depend on the magnitude) take without loss.  After the first few steps of a 2048-key row the
// branch is rarely taken, which removes the 16 accumulator multiplies per (query tile, step).  This changes the rounding order
// against the head-dim-32 kernels; head dim 64 only occurs in MiT-B2 and up, whose parity is pinned against the fp32 oracle with the
// bf16 tolerance (tests: test_attention, test_full_size_fp32_and_bf16_vs_oracle[cfg4]); the SegFormer-B0 fixtures never reach this kernel.
template <int QW, bool PRE>
__global__ void __launch_bounds__(AM_THREADS, 3) attn_fwd64p_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k,
                                                                    int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                                                    bf16_t* __restrict__ o, int64_t ldo, float* __restrict__ lse, int heads,
                                                                    int N, int Nkv, float scale) {
    constexpr int HD = 64, KS = 2, DT = 4;
    __shared__ __attribute__((aligned(1024))) bf16_t KV[2][2][AMP_KC * HD];          // [buffer][K, V]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * (16 * QW);
    const bf16_t* Qb = q + (int64_t)b * N * ldq + h * HD;
    const bf16_t* Kb = k + (int64_t)b * Nkv * ldk + h * HD;
    const bf16_t* Vb = v + (int64_t)b * Nkv * ldv + h * HD;
    const AmpSrc so = amp_src(ldk, ldv, wave, lane);
    amp_stage_fast<0>(KV, Kb, ldk, Vb, ldv, 0, Nkv, so, wave, lane);
    const AmpBases ab = amp_bases(&KV[0][0][0], lane);
    const float cs = scale * 1.44269504088896340736f;              // scores -> log2 units
    bf16x8 Qf[QW][KS];
#pragma unroll
    for (int t = 0; t < QW; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int row = q0 + 16 * t + c;
            Qf[t][s] = ld_frag_global(Qb + (int64_t)row * ldq + 32 * s + 8 * g, row < N);
            if constexpr (PRE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) Qf[t][s][j] = (__bf16)((float)Qf[t][s][j] * cs);
            }
        }
    // the softmax denominators are a fifth "value column" of ones: L[t] = ones^T P^T accumulates sum_k p in every lane (the matrix pipe
    // sums over all 32 keys of a step, i.e. over the four lane groups too), on the SAME bf16-rounded probabilities that enter P V --
    // one MFMA per (query tile, step) instead of eight vector adds and, at the end, no cross-lane sum
    f32x4 O[DT][QW], L[QW], Cn[QW];     // Cn (PRE): -m in all four registers = the C operand of the score products
    float m[QW];
    const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u));
#pragma unroll
    for (int t = 0; t < QW; ++t) {
        m[t] = PRE ? 0.f : -INFINITY; L[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; Cn[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) asm volatile("" : "+v"(Cn[t]));          // (opaque: kept in four registers, not re-splatted per product)
#pragma unroll
        for (int d = 0; d < DT; ++d) O[d][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int nst = (Nkv + AMP_KC - 1) / AMP_KC;
    // one 32-key step of stage `st` in buffer BUF, keys KB .. KB + 31 of the stage
    auto step = [&](auto bufc, auto kbc, int st) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, KB = decltype(kbc)::value;
        constexpr int OK = BUF * 0x4000 + KB * 128, OV = OK + 0x2000;
        const int kc0 = st * AMP_KC;
        bf16x8 Kf[2][KS], Vf[DT];
        Kf[0][0] = amp_lds128<OK>(ab.k[0]); Kf[0][1] = amp_lds128<OK>(ab.k[1]);
        Kf[1][0] = amp_lds128<OK + 2048>(ab.k[0]); Kf[1][1] = amp_lds128<OK + 2048>(ab.k[1]);
        Vf[0] = amp_trs<OV>(ab.v[0]); Vf[1] = amp_trs<OV>(ab.v[1]); Vf[2] = amp_trs<OV>(ab.v[2]); Vf[3] = amp_trs<OV>(ab.v[3]);
        AMP_LGKM0();
        float sv[QW][2][4], mx[QW];
#pragma unroll
        for (int t = 0; t < QW; ++t)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x4 S = PRE ? Cn[t] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kf[kt][s], Qf[t][s], S, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) sv[t][kt][r] = S[r];
            }
        if (kc0 + KB + 32 > Nkv) {                          // last step of a key count that is not a multiple of 32 (wave-uniform)
#pragma unroll
            for (int t = 0; t < QW; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kc0 + KB + 16 * kt + 4 * g + r >= Nkv) sv[t][kt][r] = -INFINITY;
        }
        if constexpr (QW == 2) {
            // The step's maxima, ONE asm block.  An inline-asm instruction that reads a matrix-instruction result is invisible to the
            // compiler's hazard recogniser (it pads "XDL write -> VALU read" only in front of instructions it classifies as VALU), and
            // the hardware does not interlock that read: r04's separate v_max3 statements sat two or three issue slots behind the
            // last score product and, now and then, read the register's PREVIOUS content -- a stale score of similar size, so the
            // running maximum moved differently and O / lse came out valid but not bit-reproducible (found in r05 as 5 of 16384 lse
            // rows differing between two runs).  s_nop 7 = the 8 wait states a four-pass result needs, whatever order the products
            // were issued in.
            float t0, t1;
            asm volatile("s_nop 7\n\t"
                         "v_max3_f32 %0, %4, %5, %6\n\t"
                         "v_max3_f32 %2, %7, %8, %9\n\t"
                         "v_max3_f32 %1, %12, %13, %14\n\t"
                         "v_max3_f32 %3, %15, %16, %17\n\t"
                         "v_max3_f32 %0, %0, %10, %11\n\t"
                         "v_max3_f32 %1, %1, %18, %19\n\t"
                         "v_max_f32 %0, %0, %2\n\t"
                         "v_max_f32 %1, %1, %3"
                         : "=&v"(mx[0]), "=&v"(mx[1]), "=&v"(t0), "=&v"(t1)
                         : "v"(sv[0][0][0]), "v"(sv[0][0][1]), "v"(sv[0][0][2]), "v"(sv[0][0][3]), "v"(sv[0][1][0]), "v"(sv[0][1][1]),
                           "v"(sv[0][1][2]), "v"(sv[0][1][3]), "v"(sv[1][0][0]), "v"(sv[1][0][1]), "v"(sv[1][0][2]), "v"(sv[1][0][3]),
                           "v"(sv[1][1][0]), "v"(sv[1][1][1]), "v"(sv[1][1][2]), "v"(sv[1][1][3]));
            float a0 = mx[0], b0 = mx[0], a1 = mx[1], b1 = mx[1];
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32_e32 %0, %1\n\tv_permlane16_swap_b32_e32 %2, %3\n\ts_nop 1" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
            a0 = b0 = am_max2(a0, b0); a1 = b1 = am_max2(a1, b1);
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1\n\tv_permlane32_swap_b32_e32 %2, %3\n\ts_nop 1" : "+v"(a0), "+v"(b0), "+v"(a1), "+v"(b1));
            mx[0] = am_max2(a0, b0); mx[1] = am_max2(a1, b1);
        } else {
#pragma unroll
            for (int t = 0; t < QW; ++t) {
                float r = fmaxf(fmaxf(fmaxf(sv[t][0][0], sv[t][0][1]), fmaxf(sv[t][0][2], sv[t][0][3])),
                                fmaxf(fmaxf(sv[t][1][0], sv[t][1][1]), fmaxf(sv[t][1][2], sv[t][1][3])));
                mx[t] = xgroup_max(r);
            }
        }
        if constexpr (PRE) {
            // the scores arrive as s c - m: mx is the step's maximum RELATIVE to the running one.  The very first step has m = 0 (not
            // a maximum of anything): it always sets m to its own maximum, whatever the sign.
            const bool first = KB == 0 && st == 0;
            bool grow = false;
#pragma unroll
            for (int t = 0; t < QW; ++t) grow = grow || (mx[t] > AMP_THR);
            if (first || __builtin_amdgcn_ballot_w64(grow) != 0) {
#pragma unroll
                for (int t = 0; t < QW; ++t) {
                    const float delta = first ? mx[t] : fmaxf(mx[t], 0.f);
                    const float alpha = __builtin_amdgcn_exp2f(-fmaxf(delta, 0.f));      // (first step: O = l = 0, any finite factor will do)
                    m[t] += delta;
                    Cn[t] -= delta;
                    asm volatile("" : "+v"(Cn[t]));
                    L[t] *= alpha;
#pragma unroll
                    for (int d = 0; d < DT; ++d) O[d][t] *= alpha;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sv[t][kt][r] -= delta;
                }
            }
        } else {
            bool grow = false;
#pragma unroll
            for (int t = 0; t < QW; ++t) { mx[t] *= cs; grow = grow || (mx[t] > m[t] + AMP_THR); }
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {       // rare after the first steps: raise the maxima, rescale O and l
#pragma unroll
                for (int t = 0; t < QW; ++t) {
                    const float mnew = fmaxf(m[t], mx[t]);
                    const float alpha = __builtin_amdgcn_exp2f(m[t] - mnew);          // exp2(-inf) = 0 on the first step
                    m[t] = mnew;
                    L[t] *= alpha;
#pragma unroll
                    for (int d = 0; d < DT; ++d) O[d][t] *= alpha;
                }
            }
        }
        bf16x8 Pf[QW];
#pragma unroll
        for (int t = 0; t < QW; ++t) {
            const float nm = -m[t];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    sv[t][kt][r] = __builtin_amdgcn_exp2f(PRE ? sv[t][kt][r] : fmaf(sv[t][kt][r], cs, nm));
            Pf[t] = pack_acc(sv[t][0], sv[t][1]);
        }
#pragma unroll
        for (int t = 0; t < QW; ++t) L[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, Pf[t], L[t], 0, 0, 0);
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int t = 0; t < QW; ++t) O[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Vf[d], Pf[t], O[d][t], 0, 0, 0);
    };
    auto stage = [&](auto bufc, int st) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of stage st have landed ...
        __syncthreads();                                       // ... everyone's have, and nobody reads the other buffer any more
        if (st + 1 < nst) amp_stage_fast<BUF ^ 1>(KV, Kb, ldk, Vb, ldv, (st + 1) * AMP_KC, Nkv, so, wave, lane);
        if (q0 >= N) return;                                   // wave-uniform; the wave still stages its pieces and joins the barriers
        step(bufc, amp_ic<0>{}, st);
        step(bufc, amp_ic<32>{}, st);
    };
    for (int st = 0; st < nst; st += 2) {
        stage(amp_ic<0>{}, st);
        if (st + 1 < nst) stage(amp_ic<1>{}, st + 1);
    }
    if (q0 >= N) return;
    bf16_t* Ob = o + (int64_t)b * N * ldo + h * HD;
#pragma unroll
    for (int t = 0; t < QW; ++t) {
        const float lt = L[t][0];
        const float inv = 1.f / lt;
        const int row = q0 + 16 * t + c;
        if (row < N) {
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const uint2 u = make_uint2(pack2bf(O[d][t][0] * inv, O[d][t][1] * inv), pack2bf(O[d][t][2] * inv, O[d][t][3] * inv));
                *reinterpret_cast<uint2*>(Ob + (int64_t)row * ldo + 16 * d + 4 * g) = u;
            }
            // log-sum-exp in natural units, as the backward takes it: m is in log2 units
            if (g == 0) lse[((int64_t)b * heads + h) * N + row] = m[t] * 0.69314718055994530942f + __logf(lt);
        }
    }
}

// Query-side backward at head dim 64, long key sequences: P = exp2(s c - lse log2(e)) (one v_fma + one v_exp per pair, the scale inside c),
// dS = P (dP - D) unscaled (dQ is scaled once at the end), K^T fragments read only after the score products have consumed the
// row-major K / V fragments (their registers are free by then: no spills under the three-waves-per-SIMD budget).
// r05: LDS addresses as per-lane bases + immediates (amp_bases), stage sources as a scalar row base + 32-bit lane offsets; PRE = true:
// c rides on the Q fragments exactly as in the forward (the SAME rounded bf16(q c), so the recomputed scores are the forward's) and
// -lse log2(e) is the initial value of the score accumulators, as -D already was for dP: the pair costs one v_exp and one multiply.
template <int QW, bool PRE>
__global__ void __launch_bounds__(AM_THREADS, 3) attn_bwd_dq64p_kernel(const bf16_t* __restrict__ q, int64_t ldq, const bf16_t* __restrict__ k,
                                                                       int64_t ldk, const bf16_t* __restrict__ v, int64_t ldv,
                                                                       const bf16_t* __restrict__ o, int64_t ldo,
                                                                       const bf16_t* __restrict__ dO, int64_t lddo,
                                                                       const float* __restrict__ lse, bf16_t* __restrict__ dq, int64_t lddq,
                                                                       float* __restrict__ Dbuf, int heads, int N, int Nkv, float scale) {
    constexpr int HD = 64, KS = 2, DT = 4;
    __shared__ __attribute__((aligned(1024))) bf16_t KV[2][2][AMP_KC * HD];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * (16 * QW);
    const bf16_t* Qb = q + (int64_t)b * N * ldq + h * HD;
    const bf16_t* Ob = o + (int64_t)b * N * ldo + h * HD;
    const bf16_t* dOb = dO + (int64_t)b * N * lddo + h * HD;
    const bf16_t* Kb = k + (int64_t)b * Nkv * ldk + h * HD;
    const bf16_t* Vb = v + (int64_t)b * Nkv * ldv + h * HD;
    const AmpSrc so = amp_src(ldk, ldv, wave, lane);
    amp_stage_fast<0>(KV, Kb, ldk, Vb, ldv, 0, Nkv, so, wave, lane);
    const AmpBases ab = amp_bases(&KV[0][0][0], lane);
    const float cs = scale * 1.44269504088896340736f;
    bf16x8 Qf[QW][KS], dOf[QW][KS];
    float Dq[QW], nl2[QW];                  // nl2 = -lse in log2 units (+inf -> -inf for rows beyond N: their probabilities are 0)
    f32x4 Cd[QW], Cl[QW];                   // -D (and, PRE, -lse log2 e) in all four registers: the C operands of the dP / score products
#pragma unroll
    for (int t = 0; t < QW; ++t) {
        const int row = q0 + 16 * t + c;
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Qf[t][s] = ld_frag_global(Qb + (int64_t)row * ldq + 32 * s + 8 * g, row < N);
            dOf[t][s] = ld_frag_global(dOb + (int64_t)row * lddo + 32 * s + 8 * g, row < N);
            const bf16x8 of = ld_frag_global(Ob + (int64_t)row * ldo + 32 * s + 8 * g, row < N);
#pragma unroll
            for (int j = 0; j < 8; ++j) part += (float)dOf[t][s][j] * (float)of[j];
            if constexpr (PRE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) Qf[t][s][j] = (__bf16)((float)Qf[t][s][j] * cs);
            }
        }
        Dq[t] = xgroup_sum(part);
        nl2[t] = row < N ? -lse[((int64_t)b * heads + h) * N + row] * 1.44269504088896340736f : -INFINITY;
        if (g == 0 && row < N) Dbuf[((int64_t)b * heads + h) * N + row] = Dq[t];
        Cd[t] = (f32x4){-Dq[t], -Dq[t], -Dq[t], -Dq[t]};
        Cl[t] = PRE ? (f32x4){nl2[t], nl2[t], nl2[t], nl2[t]} : (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) { asm volatile("" : "+v"(Cd[t])); asm volatile("" : "+v"(Cl[t])); }
    }
    f32x4 dQ[DT][QW];
#pragma unroll
    for (int t = 0; t < QW; ++t)
#pragma unroll
        for (int d = 0; d < DT; ++d) dQ[d][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nst = (Nkv + AMP_KC - 1) / AMP_KC;
    auto step = [&](auto bufc, auto kbc, int st) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, KB = decltype(kbc)::value;
        constexpr int OK = BUF * 0x4000 + KB * 128, OV = OK + 0x2000;
        const int kc0 = st * AMP_KC;
        float ds[QW][2][4];
        {
            bf16x8 Kf[2][KS], Vf[2][KS];
            Kf[0][0] = amp_lds128<OK>(ab.k[0]); Kf[0][1] = amp_lds128<OK>(ab.k[1]);
            Vf[0][0] = amp_lds128<OV>(ab.k[0]); Vf[0][1] = amp_lds128<OV>(ab.k[1]);
            Kf[1][0] = amp_lds128<OK + 2048>(ab.k[0]); Kf[1][1] = amp_lds128<OK + 2048>(ab.k[1]);
            Vf[1][0] = amp_lds128<OV + 2048>(ab.k[0]); Vf[1][1] = amp_lds128<OV + 2048>(ab.k[1]);
            AMP_LGKM0();
            f32x4 S[QW][2], dP[QW][2];
#pragma unroll
            for (int t = 0; t < QW; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    // (-D is the initial value of the dP accumulators: dP - D comes out of the matrix pipe, no subtraction per score)
                    S[t][kt] = Cl[t]; dP[t][kt] = Cd[t];
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        S[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kf[kt][s], Qf[t][s], S[t][kt], 0, 0, 0);
                        dP[t][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Vf[kt][s], dOf[t][s], dP[t][kt], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int t = 0; t < QW; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __builtin_amdgcn_exp2f(PRE ? S[t][kt][r] : fmaf(S[t][kt][r], cs, nl2[t]));
                        ds[t][kt][r] = p * dP[t][kt][r];
                    }
        }
        if (kc0 + KB + 32 > Nkv) {                          // keys beyond the last one carry no gradient (wave-uniform tail)
#pragma unroll
            for (int t = 0; t < QW; ++t)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kc0 + KB + 16 * kt + 4 * g + r >= Nkv) ds[t][kt][r] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);                  // the K^T reads stay behind the score products (register budget)
        bf16x8 KT[DT];
        KT[0] = amp_trs<OK>(ab.v[0]); KT[1] = amp_trs<OK>(ab.v[1]); KT[2] = amp_trs<OK>(ab.v[2]); KT[3] = amp_trs<OK>(ab.v[3]);
        bf16x8 dSf[QW];
#pragma unroll
        for (int t = 0; t < QW; ++t) dSf[t] = pack_acc(ds[t][0], ds[t][1]);
        AMP_LGKM0();
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int t = 0; t < QW; ++t) dQ[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KT[d], dSf[t], dQ[d][t], 0, 0, 0);
    };
    auto stage = [&](auto bufc, int st) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < nst) amp_stage_fast<BUF ^ 1>(KV, Kb, ldk, Vb, ldv, (st + 1) * AMP_KC, Nkv, so, wave, lane);
        if (q0 >= N) return;
        step(bufc, amp_ic<0>{}, st);
        step(bufc, amp_ic<32>{}, st);
    };
    for (int st = 0; st < nst; st += 2) {
        stage(amp_ic<0>{}, st);
        if (st + 1 < nst) stage(amp_ic<1>{}, st + 1);
    }
    if (q0 >= N) return;
    bf16_t* dQb = dq + (int64_t)b * N * lddq + h * HD;
#pragma unroll
    for (int t = 0; t < QW; ++t) {
        const int row = q0 + 16 * t + c;
        if (row < N) {
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                dQ[d][t] *= scale;
                const uint2 u = make_uint2(pack2bf(dQ[d][t][0], dQ[d][t][1]), pack2bf(dQ[d][t][2], dQ[d][t][3]));
                *reinterpret_cast<uint2*>(dQb + (int64_t)row * lddq + 16 * d + 4 * g) = u;
            }
        }
    }
}

// (r04's A/B variants of the head-dim-64 kernels -- three waves per SIMD, dense LDS tiles -- lost and are no longer instantiated)
static bool am_is_pow2(float s) {
    int e;
    return s > 0.f && frexpf(s, &e) == 0.5f && e > -60 && e < 60;
}

int attn_mfma_fwd(int hd, int B, int heads, int N, int Nkv, const void* q, int64_t ldq, const void* k, int64_t ldk,
                  const void* v, int64_t ldv, float scale, void* o, int64_t ldo, float* lse, hipStream_t st) {
    constexpr int QW = 2;
    dim3 grid((unsigned)cdiv64(N, 4 * 16 * QW), heads, B);      // (one query tile per wave measured 2 % slower at head dim 64)
#define AM_FWD(HDv, P2, OCCv, SWv) hipLaunchKernelGGL((attn_mfma_fwd_kernel<HDv, QW, P2, OCCv, SWv>), grid, dim3(AM_THREADS), 0, st, (const bf16_t*)q, ldq, \
        (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, heads, N, Nkv, scale)
    if (hd == 32) AM_FWD(32, false, 4, false);
    else if (Nkv >= 2 * AMP_KC) {
        if (!POL(attn64_prescale))
            hipLaunchKernelGGL((attn_fwd64p_kernel<QW, false>), grid, dim3(AM_THREADS), 0, st, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                               (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, heads, N, Nkv, scale);
        else
            hipLaunchKernelGGL((attn_fwd64p_kernel<QW, true>), grid, dim3(AM_THREADS), 0, st, (const bf16_t*)q, ldq, (const bf16_t*)k, ldk,
                               (const bf16_t*)v, ldv, (bf16_t*)o, ldo, lse, heads, N, Nkv, scale);
    }
    else if (!am_is_pow2(scale)) AM_FWD(64, false, 4, true);
    else AM_FWD(64, true, 4, true);
#undef AM_FWD
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- backward, query side: D = rowsum(dO * O), dQ = scale * (P o (dP - D)) K ---------------------------------------------
template <int HD, int QW, bool P2S, int OCC = 4, bool SWZ = true>
__global__ void __launch_bounds__(AM_THREADS, OCC) attn_mfma_bwd_dq_kernel(const bf16_t* __restrict__ q, int64_t ldq,
                                                                       const bf16_t* __restrict__ k, int64_t ldk,
                                                                       const bf16_t* __restrict__ v, int64_t ldv,
                                                                       const bf16_t* __restrict__ o, int64_t ldo,
                                                                       const bf16_t* __restrict__ dO, int64_t lddo,
                                                                       const float* __restrict__ lse, bf16_t* __restrict__ dq,
                                                                       int64_t lddq, float* __restrict__ Dbuf, int heads, int N,
                                                                       int Nkv, float scale) {
    constexpr int AM_KC = AM_KC_OF(HD);
    __shared__ __attribute__((aligned(16))) bf16_t Ks[AM_KC * HD];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[AM_KC * HD];
    constexpr int KS = HD / 32, DT = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * (16 * QW);
    const bf16_t* Qb = q + (int64_t)b * N * ldq + h * HD;
    const bf16_t* Ob = o + (int64_t)b * N * ldo + h * HD;
    const bf16_t* dOb = dO + (int64_t)b * N * lddo + h * HD;
    const bf16_t* Kb = k + (int64_t)b * Nkv * ldk + h * HD;
    const bf16_t* Vb = v + (int64_t)b * Nkv * ldv + h * HD;
    bf16x8 Qf[QW][KS], dOf[QW][KS];
    float Dq[QW], lq[QW];
#pragma unroll
    for (int t = 0; t < QW; ++t) {
        const int row = q0 + 16 * t + c;
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Qf[t][s] = ld_frag_global(Qb + (int64_t)row * ldq + 32 * s + 8 * g, row < N);
            dOf[t][s] = ld_frag_global(dOb + (int64_t)row * lddo + 32 * s + 8 * g, row < N);
            const bf16x8 of = ld_frag_global(Ob + (int64_t)row * ldo + 32 * s + 8 * g, row < N);
#pragma unroll
            for (int j = 0; j < 8; ++j) part += (float)dOf[t][s][j] * (float)of[j];
        }
        Dq[t] = xgroup_sum(part);
        lq[t] = row < N ? lse[((int64_t)b * heads + h) * N + row] : INFINITY;
        if (g == 0 && row < N) Dbuf[((int64_t)b * heads + h) * N + row] = Dq[t];
        if (P2S) {              // the scores come out of the MFMA scaled; dS stays unscaled and dQ is scaled once at the end (all exact)
#pragma unroll
            for (int s = 0; s < KS; ++s) Qf[t][s] = scale_frag_pow2(Qf[t][s], scale);
        }
    }
    f32x4 dQ[DT][QW];
#pragma unroll
    for (int t = 0; t < QW; ++t)
#pragma unroll
        for (int d = 0; d < DT; ++d) dQ[d][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kc0 = 0; kc0 < Nkv; kc0 += AM_KC) {
        const int nk = Nkv - kc0 < AM_KC ? Nkv - kc0 : AM_KC;
        const int nk32 = (nk + 31) & ~31;
        __syncthreads();
        stage_rows<HD, SWZ>(Ks, Kb, ldk, kc0, nk32, Nkv);
        stage_rows<HD, SWZ>(Vs, Vb, ldv, kc0, nk32, Nkv);
        __syncthreads();
        if (q0 >= N) continue;
        for (int kb = 0; kb < nk; kb += 32) {
            bf16x8 Kf[2][KS], Vf[2][KS], KT[DT];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int off = (kb + 16 * kt + c) * HD + (am_chunk<HD, SWZ>(4 * s + g, c & 7) << 3);
                    Kf[kt][s] = ld_frag_lds(Ks + off);
                    Vf[kt][s] = ld_frag_lds(Vs + off);
                }
#pragma unroll
            for (int d = 0; d < DT; ++d) KT[d] = ld_frag_trs<HD, SWZ>(Ks, kb + 4 * g, kb + 16 + 4 * g, d, lane);
            auto step = [&](auto maskc) {
                constexpr bool MASK = decltype(maskc)::value;
#pragma unroll
                for (int t = 0; t < QW; ++t) {
                    float ds[2][4];
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        f32x4 S = (f32x4){0.f, 0.f, 0.f, 0.f}, dP = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kf[kt][s], Qf[t][s], S, 0, 0, 0);
                            dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Vf[kt][s], dOf[t][s], dP, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float p = __expf(P2S ? S[r] - lq[t] : S[r] * scale - lq[t]);
                            if (MASK) {
                                const int key = kc0 + kb + 16 * kt + 4 * g + r;
                                p = key < Nkv ? p : 0.f;
                            }
                            ds[kt][r] = P2S ? p * (dP[r] - Dq[t]) : p * (dP[r] - Dq[t]) * scale;
                        }
                    }
                    const bf16x8 dSf = pack_acc(ds[0], ds[1]);
#pragma unroll
                    for (int d = 0; d < DT; ++d) dQ[d][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KT[d], dSf, dQ[d][t], 0, 0, 0);
                }
            };
            if (kc0 + kb + 32 <= Nkv) step(std::false_type{}); else step(std::true_type{});
        }
    }
    if (q0 >= N) return;
    bf16_t* dQb = dq + (int64_t)b * N * lddq + h * HD;
#pragma unroll
    for (int t = 0; t < QW; ++t) {
        const int row = q0 + 16 * t + c;
        if (row < N) {
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                if (P2S) dQ[d][t] *= scale;
                const uint2 u = make_uint2(pack2bf(dQ[d][t][0], dQ[d][t][1]), pack2bf(dQ[d][t][2], dQ[d][t][3]));
                *reinterpret_cast<uint2*>(dQb + (int64_t)row * lddq + 16 * d + 4 * g) = u;
            }
        }
    }
}

// ---- backward, key side: dV = P^T dO, dK = scale * (P o (dP - D))^T Q, summed over one query chunk ------------------------
// grid (key blocks of 64 KW, query chunks, B*heads); wave w owns 16 KW keys: its K / V fragments (B operands)
// stay in registers; Q / dO tiles of 32 queries pass through LDS.  Output: fp32 slab[z][b*Nkv + key][2C] partial sums.
// KW = 16-key tiles per wave: 4 (64 keys) at head dim 32; 2 at head dim 64, where 64 keys per wave need ~300 registers (one
// wave per SIMD, nothing to overlap the Q / dO staging with) and 32 keys fit two waves per SIMD
// EX2 (head dim 64): P = exp2(s c - lse log2(e)) with the scale inside c = scale log2(e) -- one v_fma + one v_exp per pair, the
// log-sum-exp stored negated in log2 units when the tile is staged; dS unscaled, dK scaled once at the end (as P2S, for any scale)
// QR = query rows per staged tile and barrier: 32, or (r05, head dim 64) 64 / 128 walked as 32-query parts -- the same products in the same
// order with a half / a quarter of the barriers and staging round trips: on 8 x 131072 queries x 2048 keys the backward 1.906 -> 1.753 ->
// 1.728 ms, cfg4 156.7 -> 161.5 -> 162.5 images/s (same box); 66 KB of LDS per workgroup at 128 rows, two workgroups per CU as before
template <int HD, bool P2S, bool SWZ = true, bool EX2 = false, int KW = (HD == 32 ? 4 : 2), int QR = 32>
__global__ void __launch_bounds__(AM_THREADS, 2) attn_mfma_bwd_dkv_kernel(const bf16_t* __restrict__ q, int64_t ldq,
                                                                        const bf16_t* __restrict__ k, int64_t ldk,
                                                                        const bf16_t* __restrict__ v, int64_t ldv,
                                                                        const bf16_t* __restrict__ dO, int64_t lddo,
                                                                        const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                                        float* __restrict__ slab, int heads, int N, int Nkv, int B,
                                                                        int qchunk, float scale) {
    // Q / dO tiles of 32 queries are double-buffered: the global loads of tile i + 1 are issued before the products of tile i and
    // written to the other buffer after them -- one barrier per tile, the load latency under the 32 MFMAs of a tile
    // (single-buffered, two barriers per tile with the loads in between: 2.16 ms per launch at head dim 64 on MiT-B2 1024 x 2048, batch 16; this form: +14 % images per second on that model)
    static_assert(QR == 32 || ((QR == 64 || QR == 128) && HD == 64), "64- / 128-query tiles: head dim 64 only");
    constexpr int NH = QR / 32;                           // 32-query halves per tile
    __shared__ __attribute__((aligned(16))) bf16_t Qs2[2][QR * HD];
    __shared__ __attribute__((aligned(16))) bf16_t dOs2[2][QR * HD];
    __shared__ float Ls2[2][QR], Ds2[2][QR];
    constexpr int KS = HD / 32, DT = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.z, b = bh / heads, h = bh - b * heads;
    const int z = blockIdx.y;
    const int key0 = blockIdx.x * (64 * KW) + wave * (16 * KW);
    const bf16_t* Qb = q + (int64_t)b * N * ldq + h * HD;
    const bf16_t* dOb = dO + (int64_t)b * N * lddo + h * HD;
    const bf16_t* Kb = k + (int64_t)b * Nkv * ldk + h * HD;
    const bf16_t* Vb = v + (int64_t)b * Nkv * ldv + h * HD;
    const float* lb = lse + ((int64_t)b * heads + h) * N;
    const float* Db = Dbuf + ((int64_t)b * heads + h) * N;
    bf16x8 Kf[KW][KS], Vf[KW][KS];
#pragma unroll
    for (int kt = 0; kt < KW; ++kt)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int key = key0 + 16 * kt + c;
            Kf[kt][s] = ld_frag_global(Kb + (int64_t)key * ldk + 32 * s + 8 * g, key < Nkv);
            Vf[kt][s] = ld_frag_global(Vb + (int64_t)key * ldv + 32 * s + 8 * g, key < Nkv);
            if (P2S && !EX2) Kf[kt][s] = scale_frag_pow2(Kf[kt][s], scale);       // K only feeds the scores here; dK is scaled once at the end
        }
    f32x4 dK[DT][KW], dV[DT][KW];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int kt = 0; kt < KW; ++kt) { dK[d][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV[d][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int qbeg = z * qchunk, qend = qbeg + qchunk < N ? qbeg + qchunk : N;
    const float cs = scale * 1.44269504088896340736f;
    // staging map: 32 rows x HD / 8 sixteen-byte chunks per matrix = HD * 4 chunks; 256 threads: one chunk of Q and one of dO each at
    // head dim 64, at head dim 32 the lower half of the workgroup takes Q and the upper half dO
    constexpr int CPR = HD / 8, NCH = 32 * CPR;
    const int sid = HD == 64 ? threadIdx.x : (threadIdx.x & (NCH - 1));
    const int srow = sid / CPR, scol = (sid - srow * CPR) * 8;
    const bool doQ = HD == 64 || threadIdx.x < NCH, doO = HD == 64 || threadIdx.x >= NCH;
    uint4 rq[NH], ro[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) { rq[j] = make_uint4(0, 0, 0, 0); ro[j] = make_uint4(0, 0, 0, 0); }
    float rl = INFINITY, rd = 0.f;
    // Rows beyond the chunk are read from its LAST row (finite values) and carry lse = +inf: their probabilities are exactly 0, so they
    // add nothing -- no zero fill, no divergent loads.  The row pointers advance by a wave-uniform step per tile (the 64-bit
    // row * stride products were ~25 vector instructions per tile in a loop whose VALU stream is as long as its MFMA stream).
    const int row0 = qbeg + srow < qend ? qbeg + srow : qend - 1;
    const char* qp = reinterpret_cast<const char*>(Qb + (int64_t)row0 * ldq + scol);
    const char* op = reinterpret_cast<const char*>(dOb + (int64_t)row0 * lddo + scol);
    const int64_t qstep = 64 * ldq, ostep = 64 * lddo;            // bytes per 32 rows
    auto fetch = [&](int qt0) {
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            const int row = qt0 + 32 * j + srow;
            if (row >= qend) {                               // (only in the last tile of a chunk whose length is not a multiple of the tile)
                qp = reinterpret_cast<const char*>(Qb + (int64_t)(qend - 1) * ldq + scol);
                op = reinterpret_cast<const char*>(dOb + (int64_t)(qend - 1) * lddo + scol);
            }
            if (doQ) rq[j] = *reinterpret_cast<const uint4*>(qp);
            if (doO) ro[j] = *reinterpret_cast<const uint4*>(op);
            qp += qstep; op += ostep;
        }
        if (threadIdx.x < QR) {
            const int r2 = qt0 + threadIdx.x;
            rl = r2 < qend ? lb[r2] : INFINITY;          // exp(s - inf) = 0: rows beyond the chunk contribute nothing
            if (EX2) rl *= -1.44269504088896340736f;
            rd = r2 < qend ? Db[r2] : 0.f;
            if (EX2) rd = -rd;                          // EX2: -D is the INITIAL VALUE of the dP accumulators (no subtraction per score)
        }
    };
    const int sswz = srow * HD + (am_chunk<HD, SWZ>(scol >> 3, srow & 7) << 3);      // swizzled LDS position of this thread's chunk
    auto put = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NH; ++j) {                       // (row 32 j + srow: the swizzle depends on the row's low three bits only)
            if (doQ) *reinterpret_cast<uint4*>(Qs2[buf] + 32 * j * HD + sswz) = rq[j];
            if (doO) *reinterpret_cast<uint4*>(dOs2[buf] + 32 * j * HD + sswz) = ro[j];
        }
        if (threadIdx.x < QR) { Ls2[buf][threadIdx.x] = rl; Ds2[buf][threadIdx.x] = rd; }
    };
    fetch(qbeg);
    put(0);
    __syncthreads();
    // the tile loop is written out for the two buffers: with the buffer a compile-time constant every LDS address of a tile is a per-lane
    // base plus an immediate (it was ~35 integer instructions per tile, a quarter of the loop's VALU stream)
    auto tile = [&](auto BUF, int qt0) {
        constexpr int buf = decltype(BUF)::value;
        const bool more = qt0 + QR < qend;
        if (more) fetch(qt0 + QR);
        if (key0 < Nkv) {
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
        if (hh > 0 && qt0 + 32 * hh >= qend) break;          // (workgroup-uniform: the chunk ended inside the tile's first half)
        const bf16_t* Qs = Qs2[buf] + 32 * hh * HD;
        const bf16_t* dOs = dOs2[buf] + 32 * hh * HD;
        const float* Ls = Ls2[buf] + 32 * hh;
        const float* Ds = Ds2[buf] + 32 * hh;
        float P[2][KW][4], dS[2][KW][4];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            bf16x8 Qa[KS], dOa[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int off = (16 * qt + c) * HD + (am_chunk<HD, SWZ>(4 * s + g, c & 7) << 3);
                Qa[s] = ld_frag_lds(Qs + off);
                dOa[s] = ld_frag_lds(dOs + off);
            }
            float lr[4], dr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { lr[r] = Ls[16 * qt + 4 * g + r]; dr[r] = Ds[16 * qt + 4 * g + r]; }
#pragma unroll
            for (int kt = 0; kt < KW; ++kt) {
                f32x4 S = (f32x4){0.f, 0.f, 0.f, 0.f}, dP = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (EX2) dP = (f32x4){dr[0], dr[1], dr[2], dr[3]};          // = -D (negated when staged): dP - D comes out of the matrix pipe
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qa[s], Kf[kt][s], S, 0, 0, 0);
                    dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dOa[s], Vf[kt][s], dP, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = EX2 ? __builtin_amdgcn_exp2f(fmaf(S[r], cs, lr[r])) : __expf(P2S ? S[r] - lr[r] : S[r] * scale - lr[r]);
                    P[qt][kt][r] = p;
                    dS[qt][kt][r] = EX2 ? p * dP[r] : (P2S ? p * (dP[r] - dr[r]) : p * (dP[r] - dr[r]) * scale);
                }
            }
        }
        bf16x8 dOT[DT], QT[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            dOT[d] = ld_frag_trs<HD, SWZ>(dOs, 4 * g, 16 + 4 * g, d, lane);
            QT[d] = ld_frag_trs<HD, SWZ>(Qs, 4 * g, 16 + 4 * g, d, lane);
        }
#pragma unroll
        for (int kt = 0; kt < KW; ++kt) {
            const bf16x8 Pf = pack_acc(P[0][kt], P[1][kt]);
            const bf16x8 dSf = pack_acc(dS[0][kt], dS[1][kt]);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                dV[d][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dOT[d], Pf, dV[d][kt], 0, 0, 0);
                dK[d][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(QT[d], dSf, dK[d][kt], 0, 0, 0);
            }
        }
        }
        }
        if (more) put(buf ^ 1);          // the other buffer was last read before the barrier that ended the previous tile
        __syncthreads();
    };
    for (int qt0 = qbeg; qt0 < qend; qt0 += 2 * QR) {
        tile(std::integral_constant<int, 0>{}, qt0);
        if (qt0 + QR < qend) tile(std::integral_constant<int, 1>{}, qt0 + QR);       // (workgroup-uniform: the barrier inside is reached by all)
    }
    if (key0 >= Nkv) return;
    const int C = heads * HD;
    float* sb = slab + (int64_t)z * B * Nkv * 2 * C;
#pragma unroll
    for (int kt = 0; kt < KW; ++kt) {
        const int key = key0 + 16 * kt + c;
        if (key < Nkv) {
            float* row = sb + ((int64_t)b * Nkv + key) * 2 * C + h * HD;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                if (P2S || EX2) dK[d][kt] *= scale;
                *reinterpret_cast<float4*>(row + 16 * d + 4 * g) = make_float4(dK[d][kt][0], dK[d][kt][1], dK[d][kt][2], dK[d][kt][3]);
                *reinterpret_cast<float4*>(row + C + 16 * d + 4 * g) = make_float4(dV[d][kt][0], dV[d][kt][1], dV[d][kt][2], dV[d][kt][3]);
            }
        }
    }
}

// ---- backward in ONE kernel when a workgroup can own all keys of a head (Nkv <= 4 waves x 64 keys at head dim 32: the 512^2
// inputs of BASELINE cfg1-3, 256 keys in every stage) ---------------------------------------------------------------------------
// The key-side kernel above already holds S and dS of a 32-query tile for all keys; the query side needs dQ = dS K, a contraction
// over KEYS, i.e. dS with the keys along the MFMA k index -- the transpose of how the accumulators hold it.  Each wave writes its
// dS (bf16, four consecutive queries per 8-byte store) into a private [key][query] slab and reads it back through
// ds_read_b64_tr_b16 as the B operand of dQ^T [d][q] += K^T [d][key] dS^T [key][q] (K^T fragments: the wave's own 64 keys, read
// once the same way); the four waves' partial dQ^T tiles meet in LDS (double-buffered: summed -- in fixed wave order -- and stored
// at the START of the next tile, behind the barrier that tile needs anyway).  D = rowsum(dO o O) is formed by the threads that
// stage the dO tile.  Five matrix products per (query tile, key tile) instead of the seven of the two-kernel form, Q / dO / lse
// read once instead of twice, no D round trip through memory; dK / dV partial slabs as before.
template <int HD, int KW>
__global__ void __launch_bounds__(AM_THREADS, 2) attn_mfma_bwd_fused_kernel(const bf16_t* __restrict__ q, int64_t ldq,
                                                                          const bf16_t* __restrict__ k, int64_t ldk,
                                                                          const bf16_t* __restrict__ v, int64_t ldv,
                                                                          const bf16_t* __restrict__ o, int64_t ldo,
                                                                          const bf16_t* __restrict__ dO, int64_t lddo,
                                                                          const float* __restrict__ lse, bf16_t* __restrict__ dq,
                                                                          int64_t lddq, float* __restrict__ slab, int heads, int N,
                                                                          int Nkv, int B, int qchunk, float scale) {
    static_assert(HD == 32, "fused backward: head dim 32");
    constexpr int KS = HD / 32, DT = HD / 16, WK = 16 * KW, DS_LD = 40;      // DS_LD: row stride (elements) of the dS^T slab
    __shared__ __attribute__((aligned(16))) bf16_t Qs2[2][32 * HD];
    __shared__ __attribute__((aligned(16))) bf16_t dOs2[2][32 * HD];
    __shared__ float Ls2[2][32], Ds2[2][32];
    __shared__ __attribute__((aligned(16))) bf16_t dSt[4][WK * DS_LD];       // per wave: dS^T [key][query]
    __shared__ __attribute__((aligned(16))) bf16_t Kst[4][WK * HD];          // per wave: its K rows (read once, transposed)
    __shared__ __attribute__((aligned(16))) float dQp[2][4][2 * DT][64 * 4];  // [buffer][wave][tile (dt, qt)][lane][4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int bh = blockIdx.z, b = bh / heads, h = bh - b * heads;
    const int z = blockIdx.y;
    const int key0 = wave * WK;
    const int nwav = (Nkv + WK - 1) / WK;                                     // waves that own keys
    const bf16_t* Qb = q + (int64_t)b * N * ldq + h * HD;
    const bf16_t* Ob = o + (int64_t)b * N * ldo + h * HD;
    const bf16_t* dOb = dO + (int64_t)b * N * lddo + h * HD;
    const bf16_t* Kb = k + (int64_t)b * Nkv * ldk + h * HD;
    const bf16_t* Vb = v + (int64_t)b * Nkv * ldv + h * HD;
    const float* lb = lse + ((int64_t)b * heads + h) * N;
    bf16_t* dQb = dq + (int64_t)b * N * lddq + h * HD;
    bf16x8 Kf[KW][KS], Vf[KW][KS];
#pragma unroll
    for (int kt = 0; kt < KW; ++kt)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int key = key0 + 16 * kt + c;
            Kf[kt][s] = ld_frag_global(Kb + (int64_t)key * ldk + 32 * s + 8 * g, key < Nkv);
            Vf[kt][s] = ld_frag_global(Vb + (int64_t)key * ldv + 32 * s + 8 * g, key < Nkv);
        }
    // K^T fragments of this wave's keys: A operand [d][key], k-slot j of lane group g = key 32 kh + 4 g + j (j < 4) / 32 kh + 16 + 4 g + j - 4
    bf16x8 KTf[DT][WK / 32];
    {
        constexpr int CPR = HD / 8;
        for (int i = lane; i < WK * CPR; i += 64) {
            const int r = i / CPR, cc = i - r * CPR;
            uint4 u = make_uint4(0, 0, 0, 0);
            if (key0 + r < Nkv) u = *reinterpret_cast<const uint4*>(Kb + (int64_t)(key0 + r) * ldk + cc * 8);
            *reinterpret_cast<uint4*>(Kst[wave] + r * HD + cc * 8) = u;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int kh = 0; kh < WK / 32; ++kh) KTf[dt][kh] = ld_frag_tr<HD>(Kst[wave], 32 * kh + 4 * g, 32 * kh + 16 + 4 * g, 16 * dt, lane);
    }
    f32x4 dK[DT][KW], dV[DT][KW];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int kt = 0; kt < KW; ++kt) { dK[d][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dV[d][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int qbeg = z * qchunk, qend = qbeg + qchunk < N ? qbeg + qchunk : N;
    // staging: threads 0..127 take the Q tile (32 rows x 4 chunks), 128..255 the dO tile and, with the same chunk of O, D = rowsum(dO o O)
    constexpr int CPR = HD / 8, NCH = 32 * CPR;
    const int sid = threadIdx.x & (NCH - 1);
    const int srow = sid / CPR, scol = (sid - srow * CPR) * 8;
    const bool doQ = threadIdx.x < NCH, doO = !doQ;
    uint4 rq = make_uint4(0, 0, 0, 0), ro = make_uint4(0, 0, 0, 0);
    float rl = INFINITY, rd = 0.f;
    auto fetch = [&](int qt0) {
        const int row = qt0 + srow;
        rq = make_uint4(0, 0, 0, 0); ro = make_uint4(0, 0, 0, 0);
        uint4 oo = make_uint4(0, 0, 0, 0);
        if (doQ && row < qend) rq = *reinterpret_cast<const uint4*>(Qb + (int64_t)row * ldq + scol);
        if (doO && row < qend) {
            ro = *reinterpret_cast<const uint4*>(dOb + (int64_t)row * lddo + scol);
            oo = *reinterpret_cast<const uint4*>(Ob + (int64_t)row * ldo + scol);
        }
        const bf16x8 a = __builtin_bit_cast(bf16x8, ro), bq = __builtin_bit_cast(bf16x8, oo);
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) part += (float)a[j] * (float)bq[j];
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);          // the four chunks of a row sit in adjacent lanes
        rd = part;
        if (threadIdx.x < 32) {
            const int r2 = qt0 + threadIdx.x;
            rl = r2 < qend ? lb[r2] : INFINITY;          // exp(s - inf) = 0: rows beyond the chunk contribute nothing
        }
    };
    auto put = [&](int buf) {
        if (doQ) *reinterpret_cast<uint4*>(Qs2[buf] + srow * HD + scol) = rq;
        else {
            *reinterpret_cast<uint4*>(dOs2[buf] + srow * HD + scol) = ro;
            if ((sid & (CPR - 1)) == 0) Ds2[buf][srow] = rd;
        }
        if (threadIdx.x < 32) Ls2[buf][threadIdx.x] = rl;
    };
    // sum of the waves' partial dQ^T tiles of the PREVIOUS query tile (buffer pb), in wave order; wave w finishes tile w = (dt, qt)
    auto finish_dq = [&](int pb, int qprev) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < nwav; ++w) acc += *reinterpret_cast<const f32x4*>(&dQp[pb][w][wave][lane * 4]);
        const int dt = wave >> 1, qt = wave & 1;
        const int row = qprev + 16 * qt + c;
        if (row < qend) {
            const uint2 u = make_uint2(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]));
            *reinterpret_cast<uint2*>(dQb + (int64_t)row * lddq + 16 * dt + 4 * g) = u;
        }
    };
    fetch(qbeg);
    put(0);
    __syncthreads();
    int buf = 0;
    for (int qt0 = qbeg; qt0 < qend; qt0 += 32, buf ^= 1) {
        const bool more = qt0 + 32 < qend;
        if (more) fetch(qt0 + 32);
        if (qt0 > qbeg) finish_dq(buf ^ 1, qt0 - 32);          // the previous tile's partials were complete at the last barrier
        const bf16_t* Qs = Qs2[buf];
        const bf16_t* dOs = dOs2[buf];
        const float* Ls = Ls2[buf];
        const float* Ds = Ds2[buf];
        if (key0 < Nkv) {
            float P[2][KW][4], dS[2][KW][4];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                bf16x8 Qa[KS], dOa[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    Qa[s] = ld_frag_lds(Qs + (16 * qt + c) * HD + 32 * s + 8 * g);
                    dOa[s] = ld_frag_lds(dOs + (16 * qt + c) * HD + 32 * s + 8 * g);
                }
                float lr[4], dr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { lr[r] = Ls[16 * qt + 4 * g + r]; dr[r] = Ds[16 * qt + 4 * g + r]; }
#pragma unroll
                for (int kt = 0; kt < KW; ++kt) {
                    f32x4 S = (f32x4){0.f, 0.f, 0.f, 0.f}, dP = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qa[s], Kf[kt][s], S, 0, 0, 0);
                        dP = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dOa[s], Vf[kt][s], dP, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = __expf(S[r] * scale - lr[r]);
                        P[qt][kt][r] = p;
                        dS[qt][kt][r] = p * (dP[r] - dr[r]) * scale;
                    }
                    // dS^T slab: key 16 kt + c, queries 16 qt + 4 g .. + 3
                    *reinterpret_cast<uint2*>(dSt[wave] + (16 * kt + c) * DS_LD + 16 * qt + 4 * g) =
                        make_uint2(pack2bf(dS[qt][kt][0], dS[qt][kt][1]), pack2bf(dS[qt][kt][2], dS[qt][kt][3]));
                }
            }
            bf16x8 dOT[DT], QT[DT];
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                dOT[d] = ld_frag_tr<HD>(dOs, 4 * g, 16 + 4 * g, 16 * d, lane);
                QT[d] = ld_frag_tr<HD>(Qs, 4 * g, 16 + 4 * g, 16 * d, lane);
            }
#pragma unroll
            for (int kt = 0; kt < KW; ++kt) {
                const bf16x8 Pf = pack_acc(P[0][kt], P[1][kt]);
                const bf16x8 dSf = pack_acc(dS[0][kt], dS[1][kt]);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    dV[d][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dOT[d], Pf, dV[d][kt], 0, 0, 0);
                    dK[d][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(QT[d], dSf, dK[d][kt], 0, 0, 0);
                }
            }
            // dQ^T [d][q] partial over this wave's keys
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                bf16x8 dSTf[WK / 32];
#pragma unroll
                for (int kh = 0; kh < WK / 32; ++kh)
                    dSTf[kh] = ld_frag_tr<DS_LD>(dSt[wave], 32 * kh + 4 * g, 32 * kh + 16 + 4 * g, 16 * qt, lane);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kh = 0; kh < WK / 32; ++kh) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(KTf[dt][kh], dSTf[kh], acc, 0, 0, 0);
                    *reinterpret_cast<f32x4*>(&dQp[buf][wave][dt * 2 + qt][lane * 4]) = acc;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (more) put(buf ^ 1);          // the other buffer was last read before the barrier that ended the previous tile
        __syncthreads();
    }
    if (qend > qbeg) finish_dq(buf ^ 1, qbeg + ((qend - qbeg - 1) / 32) * 32);
    if (key0 >= Nkv) return;
    const int C = heads * HD;
    float* sb = slab + (int64_t)z * B * Nkv * 2 * C;
#pragma unroll
    for (int kt = 0; kt < KW; ++kt) {
        const int key = key0 + 16 * kt + c;
        if (key < Nkv) {
            float* row = sb + ((int64_t)b * Nkv + key) * 2 * C + h * HD;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                *reinterpret_cast<float4*>(row + 16 * d + 4 * g) = make_float4(dK[d][kt][0], dK[d][kt][1], dK[d][kt][2], dK[d][kt][3]);
                *reinterpret_cast<float4*>(row + C + 16 * d + 4 * g) = make_float4(dV[d][kt][0], dV[d][kt][1], dV[d][kt][2], dV[d][kt][3]);
            }
        }
    }
}

int attn_mfma_bwd(int hd, int B, int heads, int N, int Nkv, const void* q, int64_t ldq, const void* k, int64_t ldk,
                  const void* v, int64_t ldv, float scale, const void* o, int64_t ldo, const void* d_o, int64_t lddo,
                  const float* lse, void* dq, int64_t lddq, float* Dbuf, float* slab, int nchunk, int qchunk, hipStream_t st) {
    constexpr int QW = 2;
    dim3 g1((unsigned)cdiv64(N, 4 * 16 * QW), heads, B);
    dim3 g2((unsigned)cdiv64(Nkv, hd == 32 ? 256 : 128), nchunk, B * heads);      // keys per workgroup = 4 waves x 16 KW
    const bf16_t* Q = (const bf16_t*)q; const bf16_t* K = (const bf16_t*)k; const bf16_t* V = (const bf16_t*)v;
    const bf16_t* O = (const bf16_t*)o; const bf16_t* DO = (const bf16_t*)d_o;
    if (hd == 32 && Nkv <= 256 && !POL(attn_no_fused_bwd)) {      // one workgroup owns all keys: dQ, dK, dV in one kernel
        dim3 g3(1, nchunk, B * heads);
        hipLaunchKernelGGL((attn_mfma_bwd_fused_kernel<32, 4>), g3, dim3(AM_THREADS), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, DO, lddo, lse,
                           (bf16_t*)dq, lddq, slab, heads, N, Nkv, B, qchunk, scale);
    } else if (hd == 32) {
        hipLaunchKernelGGL((attn_mfma_bwd_dq_kernel<32, QW, false, 4, false>), g1, dim3(AM_THREADS), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, DO, lddo,
                           lse, (bf16_t*)dq, lddq, Dbuf, heads, N, Nkv, scale);
        hipLaunchKernelGGL((attn_mfma_bwd_dkv_kernel<32, false, false>), g2, dim3(AM_THREADS), 0, st, Q, ldq, K, ldk, V, ldv, DO, lddo, lse, Dbuf,
                           slab, heads, N, Nkv, B, qchunk, scale);
    } else {
#define AM_DQ(P2, OCCv, SWv) hipLaunchKernelGGL((attn_mfma_bwd_dq_kernel<64, QW, P2, OCCv, SWv>), g1, dim3(AM_THREADS), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, \
        DO, lddo, lse, (bf16_t*)dq, lddq, Dbuf, heads, N, Nkv, scale)
#define AM_DKV(P2, SWv, EXv) hipLaunchKernelGGL((attn_mfma_bwd_dkv_kernel<64, P2, SWv, EXv>), g2, dim3(AM_THREADS), 0, st, Q, ldq, K, ldk, V, ldv, DO, lddo, lse, Dbuf, \
        slab, heads, N, Nkv, B, qchunk, scale)
        if (Nkv >= 2 * AMP_KC) {
            if (!POL(attn64_prescale))
                hipLaunchKernelGGL((attn_bwd_dq64p_kernel<QW, false>), g1, dim3(AM_THREADS), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, DO, lddo, lse,
                                   (bf16_t*)dq, lddq, Dbuf, heads, N, Nkv, scale);
            else
                hipLaunchKernelGGL((attn_bwd_dq64p_kernel<QW, true>), g1, dim3(AM_THREADS), 0, st, Q, ldq, K, ldk, V, ldv, O, ldo, DO, lddo, lse,
                                   (bf16_t*)dq, lddq, Dbuf, heads, N, Nkv, scale);
            // (a barrier-free form -- every wave staging its own copy of the Q / dO tile by LDS-DMA into a private double-buffered slab --
            // measured 2 % SLOWER: nine DMA pieces per tile and wave cost more issue time than the per-tile barrier they remove)
            // (Q / dO tiles of 128 queries per barrier, walked as four 32-query quarters: policy attn64_dkv_rows)
            if (POL(attn64_dkv_rows) >= 128)
                hipLaunchKernelGGL((attn_mfma_bwd_dkv_kernel<64, false, true, true, 2, 128>), g2, dim3(AM_THREADS), 0, st, Q, ldq, K, ldk, V, ldv, DO, lddo,
                                   lse, Dbuf, slab, heads, N, Nkv, B, qchunk, scale);
            else if (POL(attn64_dkv_rows) >= 64)
                hipLaunchKernelGGL((attn_mfma_bwd_dkv_kernel<64, false, true, true, 2, 64>), g2, dim3(AM_THREADS), 0, st, Q, ldq, K, ldk, V, ldv, DO, lddo,
                                   lse, Dbuf, slab, heads, N, Nkv, B, qchunk, scale);
            else
                AM_DKV(false, true, true);
        }
        else if (!am_is_pow2(scale)) { AM_DQ(false, 4, true); AM_DKV(false, true, false); }
        else { AM_DQ(true, 4, true); AM_DKV(true, true, false); }
#undef AM_DQ
#undef AM_DKV
    }
    SEGF_CHECK_LAUNCH();
    return 0;
}
