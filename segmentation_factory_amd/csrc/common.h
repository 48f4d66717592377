// Shared device helpers for the gfx950 kernels (wave64, fp32 math on f32/bf16 storage).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/segfac.h"

typedef uint16_t bf16_t;   // raw bfloat16 bits

#include "policy.h"

// Every kernel launch of the library goes through this form of hipLaunchKernelGGL: while a launch trace is open on the calling thread
// (segf_trace_begin, include/segfac.h) the kernel's name is recorded as the launch site spells it (plus the template arguments of the
// launching host function where the spelling uses them), and in a dry run the launch itself is skipped --
// which is how tests/test_host_cpu.py::test_dispatch_of_baseline_shapes reads the dispatch decisions without a GPU.
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                               \
    do {                                                                                                                \
        SegfTrace& tr__ = segf_trace();                                                                                 \
        if (tr__.on) segf_trace_note(#kernelName, __PRETTY_FUNCTION__);                                                 \
        if (!tr__.dry) { kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__); }          \
    } while (0)

#define SEGF_CHECK_LAUNCH()                                   \
    do {                                                      \
        if (!segf_trace().dry) {                              \
            hipError_t e__ = hipGetLastError();               \
            if (e__ != hipSuccess) return (int)e__;           \
        }                                                     \
    } while (0)

#define SEGF_DISPATCH_DT(dt, T, ...)                          \
    if ((dt) == SEGF_F32) { typedef float T; __VA_ARGS__ }    \
    else if ((dt) == SEGF_BF16) { typedef bf16_t T; __VA_ARGS__ } \
    else return SEGF_ERR_DTYPE;

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }

// "Column-fixed" streaming decomposition: a row is nchunk 16-byte chunks; the launch has T = blocks * 256 threads with
// T % nchunk == 0, so global thread g keeps channel chunk (g % nchunk) for its whole grid-stride loop and its work items
// are u = g / nchunk, g / nchunk + T / nchunk, ...  Per-channel parameters are loaded once per thread, accesses stay
// flat-coalesced, and the loop carries no division.
static inline int igcd(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }
static inline int colfixed_blocks(int64_t units, int nchunk, int units_per_thread, int max_blocks) {
    const int unit = nchunk / igcd(nchunk, 256);                      // blocks must be a multiple of this
    int64_t want = cdiv64(cdiv64(units, units_per_thread > 0 ? units_per_thread : 1) * nchunk, 256);
    if (want > max_blocks) want = max_blocks;
    int64_t blocks = cdiv64(want < 1 ? 1 : want, unit) * unit;
    return (int)blocks;
}

// XCD-aware logical workgroup id for 1-D grids.  The hardware deals workgroups round-robin over the 8 XCDs, each with its own
// L2: neighbouring workgroups of a spatial kernel (the rows above / below a depthwise-conv strip, the four taps that share a
// gradient pixel in a transposed resize) land on different L2s and each re-fetches the shared lines (measured with
// FETCH_SIZE: 2-3x the algorithmic bytes).  The bijection below gives every XCD a CONTIGUOUS range of logical ids.
__device__ __forceinline__ unsigned xcd_block() {
    const unsigned n = gridDim.x, o = blockIdx.x, q = n >> 3, r8 = n & 7, x = o & 7;
    return (x < r8 ? x * (q + 1) : r8 * (q + 1) + (x - r8) * q) + (o >> 3);
}

// exact unsigned 32-bit division by a launch-constant divisor (Granlund-Montgomery): 5 integer ops instead of the ~80 of a
// 64-bit software division in per-row index arithmetic (sample index = row / rows_per_sample)
struct FastDivU32 { uint32_t m, sh1, sh2; };
static inline FastDivU32 fastdiv_make(uint32_t d) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    FastDivU32 f;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l - f.sh1;
    return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, FastDivU32 f) {
    const uint32_t t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    // plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 (RNE, NaN stays NaN)
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
// the value a store of type T keeps (fp32: itself; bf16: rounded to nearest even)
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf2f(f2bf(v)); }

// 8 consecutive elements <-> 8 floats.  `aligned` = pointer is 16-byte (bf16) / 16-byte (f32) aligned.
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
// Two-phase form for loops that want several rows in flight: issue every load8_raw of the iteration, SEGF_LOADS_ISSUED(),
// then unpack8 + arithmetic.  hipcc's scheduler otherwise keeps `load, s_waitcnt vmcnt(0), convert` together per load8
// (it minimises live registers), which turns an iteration's independent loads into a dependent chain of HBM latencies;
// the scheduling barrier pins the loads above the first use, and the waits then count down (vmcnt(N-1), vmcnt(N-2), ...).
template <typename T> struct Raw8;
template <> struct Raw8<float> { float4 a, b; };
template <> struct Raw8<bf16_t> { uint4 u; };
template <typename T> __device__ __forceinline__ Raw8<T> load8_raw(const T* p);
template <> __device__ __forceinline__ Raw8<float> load8_raw<float>(const float* p) {
    Raw8<float> r;
    r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4);
    return r;
}
template <> __device__ __forceinline__ Raw8<bf16_t> load8_raw<bf16_t>(const bf16_t* p) {
    Raw8<bf16_t> r;
    r.u = *reinterpret_cast<const uint4*>(p);
    return r;
}
__device__ __forceinline__ void unpack8(const Raw8<float>& r, float (&v)[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
__device__ __forceinline__ void unpack8(const Raw8<bf16_t>& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.u.x << 16); v[1] = __uint_as_float(r.u.x & 0xffff0000u);
    v[2] = __uint_as_float(r.u.y << 16); v[3] = __uint_as_float(r.u.y & 0xffff0000u);
    v[4] = __uint_as_float(r.u.z << 16); v[5] = __uint_as_float(r.u.z & 0xffff0000u);
    v[6] = __uint_as_float(r.u.w << 16); v[7] = __uint_as_float(r.u.w & 0xffff0000u);
}
// zero a raw chunk (padding taps): on the packed words, i.e. 4 selects for 8 bf16 values instead of 8 after unpacking
__device__ __forceinline__ void zero_unless(Raw8<bf16_t>& r, bool keep) {
    r.u.x = keep ? r.u.x : 0u; r.u.y = keep ? r.u.y : 0u; r.u.z = keep ? r.u.z : 0u; r.u.w = keep ? r.u.w : 0u;
}
__device__ __forceinline__ void zero_unless(Raw8<float>& r, bool keep) {
    if (!keep) { r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = make_float4(0.f, 0.f, 0.f, 0.f); }
}
#define SEGF_LOADS_ISSUED() __builtin_amdgcn_sched_barrier(0)

template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {      // one v_cvt_pk_bf16_f32
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    uint4 u;
    u.x = pack2bf(v[0], v[1]); u.y = pack2bf(v[2], v[3]); u.z = pack2bf(v[4], v[5]); u.w = pack2bf(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) { load8<float>(p, v); }

// wave64 reductions.  The first four butterfly steps run as DPP-modified VALU ops (quad_perm xor 1 / xor 2,
// row_half_mirror, row_mirror: after each step both partners hold the same partial, so the mirrors act as xor 4 / xor 8);
// no LDS traffic.  Steps 16 / 32 use ds_bpermute (__shfl_xor) in the width-templated form, or four v_readlane + scalar adds
// in the *_all forms (full wave, result wave-uniform).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
#define DPP_XOR1 0xB1
#define DPP_XOR2 0x4E
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
__device__ __forceinline__ float readlane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v, int width = 64) {
    if (width > 1) v += dpp_mov<DPP_XOR1>(v);
    if (width > 2) v += dpp_mov<DPP_XOR2>(v);
    if (width > 4) v += dpp_mov<DPP_HALF_MIRROR>(v);
    if (width > 8) v += dpp_mov<DPP_MIRROR>(v);
    if (width > 16) v += __shfl_xor(v, 16, 64);
    if (width > 32) v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v, int width = 64) {
    if (width > 1) v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
    if (width > 2) v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    if (width > 4) v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
    if (width > 8) v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
    if (width > 16) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (width > 32) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
// full-wave reductions with a wave-uniform result (every lane must be active)
__device__ __forceinline__ float wave_sum_all(float v) {
    v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v); v += dpp_mov<DPP_HALF_MIRROR>(v); v += dpp_mov<DPP_MIRROR>(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ float wave_max_all(float v) {
    v = fmaxf(v, dpp_mov<DPP_XOR1>(v)); v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
    v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}
__device__ __forceinline__ float wave_min_all(float v) {
    v = fminf(v, dpp_mov<DPP_XOR1>(v)); v = fminf(v, dpp_mov<DPP_XOR2>(v));
    v = fminf(v, dpp_mov<DPP_HALF_MIRROR>(v)); v = fminf(v, dpp_mov<DPP_MIRROR>(v));
    return fminf(fminf(readlane_f(v, 0), readlane_f(v, 16)), fminf(readlane_f(v, 32), readlane_f(v, 48)));
}

// PyTorch bilinear source index (aten UpSample.h area_pixel_compute_source_index):
// align_corners=False: src = (dst+0.5)*in/out-0.5 clamped at 0; True: src = dst*(in-1)/(out-1).
__device__ __forceinline__ void bilinear_src(int d, int in, int out, int align_corners, int& i0, int& i1, float& l) {
    float s;
    if (align_corners) {
        const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
        s = scale * d;
    } else {
        const float scale = (float)in / (float)out;
        s = scale * (d + 0.5f) - 0.5f;
        if (s < 0.f) s = 0.f;
    }
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l = s - (float)i0;
}

// One bilinear sample in the operation order of ATen's CPU upsample_bilinear2d on a contiguous NCHW tensor (the reference's
// evaluation path, build_models.py:65 on the CPU): four weight products wy*wx, then  v01*w01  and three fused multiply-adds
// (v00, v10, v11).  Verified bit-identical against F.interpolate for power-of-two ratios and align_corners=True on small maps
// (tests/test_kernels_gpu.py::test_argmax_tie_policy); ATen switches to differently associated vector loops for larger maps,
// so beyond that this is "the same value up to the last fp32 rounding".  l = weight of the second tap (bilinear_src).
__device__ __forceinline__ float bilinear_aten(float v00, float v01, float v10, float v11, float ly, float lx) {
    const float wy0 = 1.f - ly, wx0 = 1.f - lx;
    const float w00 = __fmul_rn(wy0, wx0), w01 = __fmul_rn(wy0, lx), w10 = __fmul_rn(ly, wx0), w11 = __fmul_rn(ly, lx);
    return __fmaf_rn(v11, w11, __fmaf_rn(v10, w10, __fmaf_rn(v00, w00, __fmul_rn(v01, w01))));
}

// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26) in ~13 branch-free instructions; libm's erff costs ~40 with two
// divergent branches per element, which made the fused depthwise-conv + GELU kernels VALU-bound.  GELU only needs absolute
// accuracy (gelu(x) = x/2 * (1 + erf(x/sqrt2))): the resulting error is <= 0.8e-7 * |x|.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = 1.f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}
// Two-channel forms: the same arithmetic on float2 vectors lowers to packed fp32 instructions (v_pk_mul / v_pk_fma_f32, two
// channels per issue slot); only the transcendentals and the sign transfer stay per element.  Same polynomial, same accuracy.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t erf_fast2(f32x2_t z, f32x2_t e /* exp(-z*z) */) {
    const f32x2_t az = __builtin_elementwise_abs(z);
    const f32x2_t d = az * 0.3275911f + 1.f;
    const f32x2_t t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    f32x2_t p = t * 1.061405429f + -1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t + -0.284496736f;
    p = p * t + 0.254829592f;
    const f32x2_t y = 1.f - p * t * e;
    return f32x2_t{copysignf(y.x, z.x), copysignf(y.y, z.y)};
}
__device__ __forceinline__ f32x2_t exp_neg_half_sq2(f32x2_t x) {          // exp(-x*x/2), the Gaussian both GELU forms need
    const f32x2_t a = x * x * -0.72134752044448170368f;                    // -0.5 * log2(e)
    return f32x2_t{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
}
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
    const f32x2_t h = x * 0.5f;
    return h * erf_fast2(x * 0.70710678118654752440f, exp_neg_half_sq2(x)) + h;
}
__device__ __forceinline__ f32x2_t gelu_erf_grad2(f32x2_t x) {
    const f32x2_t e = exp_neg_half_sq2(x);
    const f32x2_t cdf = erf_fast2(x * 0.70710678118654752440f, e) * 0.5f + 0.5f;
    return x * e * 0.39894228040143267794f + cdf;
}
// Eight channels at a time, stage by stage (all reciprocals, then all polynomials, ...): the transcendental unit has a
// multi-cycle result latency, and element-at-a-time code makes the compiler pad every rcp / exp with s_nop
template <bool GRAD>
__device__ __forceinline__ void gelu_erf8(f32x2_t (&a)[4], const f32x2_t* __restrict__ gy) {
    f32x2_t e[4], t[4], p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_t q = a[i] * a[i] * -0.72134752044448170368f;
        const f32x2_t d = __builtin_elementwise_abs(a[i]) * (0.3275911f * 0.70710678118654752440f) + 1.f;
        e[i] = f32x2_t{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
        t[i] = f32x2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = t[i] * 1.061405429f + -1.453152027f;
        p[i] = p[i] * t[i] + 1.421413741f;
        p[i] = p[i] * t[i] + -0.284496736f;
        p[i] = p[i] * t[i] + 0.254829592f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_t y = 1.f - p[i] * t[i] * e[i];                               // erf(|x| / sqrt2)
        const f32x2_t er = {copysignf(y.x, a[i].x), copysignf(y.y, a[i].y)};
        if (GRAD) a[i] = gy[i] * (a[i] * e[i] * 0.39894228040143267794f + (er * 0.5f + 0.5f));
        else { const f32x2_t h = a[i] * 0.5f; a[i] = h * er + h; }
    }
}
// gelu(a) and gelu'(a) together (the fused GELU + GRN backward needs both): g <- gelu(a), a <- gelu'(a)
__device__ __forceinline__ void gelu_erf8_both(f32x2_t (&a)[4], f32x2_t (&g)[4]) {
    f32x2_t e[4], t[4], p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_t q = a[i] * a[i] * -0.72134752044448170368f;
        const f32x2_t d = __builtin_elementwise_abs(a[i]) * (0.3275911f * 0.70710678118654752440f) + 1.f;
        e[i] = f32x2_t{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
        t[i] = f32x2_t{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = t[i] * 1.061405429f + -1.453152027f;
        p[i] = p[i] * t[i] + 1.421413741f;
        p[i] = p[i] * t[i] + -0.284496736f;
        p[i] = p[i] * t[i] + 0.254829592f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2_t y = 1.f - p[i] * t[i] * e[i];
        const f32x2_t er = {copysignf(y.x, a[i].x), copysignf(y.y, a[i].y)};
        const f32x2_t h = a[i] * 0.5f;
        g[i] = h * er + h;
        a[i] = a[i] * e[i] * 0.39894228040143267794f + (er * 0.5f + 0.5f);
    }
}
// the same on eight floats in place (column-reduction functors)
__device__ __forceinline__ void gelu_erf8_floats(float (&v)[8]) {
    f32x2_t a[4] = {f32x2_t{v[0], v[1]}, f32x2_t{v[2], v[3]}, f32x2_t{v[4], v[5]}, f32x2_t{v[6], v[7]}};
    gelu_erf8<false>(a, nullptr);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = a[i].x; v[2 * i + 1] = a[i].y; }
}
// 8 consecutive elements as 4 channel pairs
template <typename T> __device__ __forceinline__ void unpack8v(const Raw8<T>& r, f32x2_t (&v)[4]);
template <> __device__ __forceinline__ void unpack8v<float>(const Raw8<float>& r, f32x2_t (&v)[4]) {
    v[0] = f32x2_t{r.a.x, r.a.y}; v[1] = f32x2_t{r.a.z, r.a.w}; v[2] = f32x2_t{r.b.x, r.b.y}; v[3] = f32x2_t{r.b.z, r.b.w};
}
template <> __device__ __forceinline__ void unpack8v<bf16_t>(const Raw8<bf16_t>& r, f32x2_t (&v)[4]) {
    v[0] = f32x2_t{__uint_as_float(r.u.x << 16), __uint_as_float(r.u.x & 0xffff0000u)};
    v[1] = f32x2_t{__uint_as_float(r.u.y << 16), __uint_as_float(r.u.y & 0xffff0000u)};
    v[2] = f32x2_t{__uint_as_float(r.u.z << 16), __uint_as_float(r.u.z & 0xffff0000u)};
    v[3] = f32x2_t{__uint_as_float(r.u.w << 16), __uint_as_float(r.u.w & 0xffff0000u)};
}
template <typename T> __device__ __forceinline__ void store8v(T* p, const f32x2_t (&v)[4]) {
    const float f[8] = {v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y};
    store8<T>(p, f);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.f + erf_fast(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
