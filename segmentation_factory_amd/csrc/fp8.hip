// FP8 (OCP e4m3fn) forward GEMM for the pointwise linears of ConvNeXt / ConvNeXtV2 (BASELINE cfg5: "ConvNeXtV2-L + UPerNet,
// fp8 MFMA weights"; convnextv2.py:90-95 pwconv1 / pwconv2).  The reference has no fp8 path: this is an OPTION of the MI355X
// build (SegmentationModel.set_fp8), parity is stated as a tolerance against the fp32 oracle (tests/test_model_gpu.py).
//   y[m][n] = ( sum_k xq[m][k] wq[n][k] ) * sx[m] * sw[n] + bias[n]
// xq / wq: e4m3fn bytes, sx: one scale per token row (dynamic, amax / 448), sw: one scale per output channel.
// The product runs on v_mfma_scale_f32_16x16x128_f8f6f4 (block-scaled MX instruction with all block scales = 2^0: twice the bf16
// MFMA rate, 4x the K per instruction), fp32 accumulate.  A and B fragments use the SAME (lane group, register, byte) -> k map
// (32 consecutive k per lane group), so the sum over k is complete whatever order the hardware visits it in.
// Tile 128 x 128 x 128, 4 waves (2 x 2, each 64 x 64 = 4 x 4 MFMA tiles), operands staged through LDS with the next tile's
// global loads held in registers meanwhile; the bf16 result leaves through an LDS image of the tile in 16-byte stores.
#include "colreduce.h"

typedef int fp8_v8i __attribute__((ext_vector_type(8)));
typedef float fp8_v4f __attribute__((ext_vector_type(4)));

#define F8_BM 128
#define F8_BN 128
#define F8_BK 128
#define F8_LD (F8_BK + 16)          // LDS row stride in bytes: 16-byte aligned rows on different banks
#define F8_MAX 448.f                // largest finite e4m3fn

// ---- row-wise quantisation: q[r][k] = e4m3(x[r][k] / s[r]), s[r] = amax_k |x[r][k]| / 448 ------------------------------------
// one wave per row; the row is read twice (second time from L1 / L2).  K % 8 == 0.
template <typename T>
__global__ void __launch_bounds__(256) quant_rows_fp8_kernel(const T* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                             float* __restrict__ scale, int64_t rows, int K) {
    const int lane = threadIdx.x & 63;
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int nch = K / 8;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
        const T* xr = x + r * ldx;
        float mx = 0.f;
        for (int c = lane; c < nch; c += 64) {
            float v[8];
            load8<T>(xr + 8 * c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(v[j]));
        }
        mx = wave_max_all(mx);
        const float s = mx > 0.f ? mx * (1.f / F8_MAX) : 1.f;
        const float inv = 1.f / s;
        if (lane == 0) scale[r] = s;
        uint8_t* qr = q + r * ldq;
        for (int c = lane; c < nch; c += 64) {
            float v[8];
            load8<T>(xr + 8 * c, v);
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
            *reinterpret_cast<int2*>(qr + 8 * c) = make_int2(lo, hi);
        }
    }
}

extern "C" int segf_quant_rows_fp8(int dt, int64_t rows, int K, const void* x, int64_t ldx, void* q, int64_t ldq, float* scale,
                                   void* stream) {
    if (rows <= 0) return 0;
    if (K <= 0 || K % 8 || ldx < K || ldq < K || (ldq % 8) || ((uintptr_t)x % 16) || ((uintptr_t)q % 8)) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (int)imin64(cdiv64(rows, 4), 8192);
    SEGF_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL((quant_rows_fp8_kernel<T>), dim3(blocks), dim3(256), 0, st, (const T*)x, ldx, (uint8_t*)q, ldq, scale, rows, K);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- tensor-wise quantisation: one scale for a whole [rows][cols] activation / gradient tensor (the implicit-GEMM 3x3 convolution
// gathers its K axis from nine neighbouring pixels, so per-token scales do not factor out of the sum).
//   amax pass: amax_bits = max over the tensor of |x| (as the unsigned bits of a non-negative float: atomicMax is exact and
//              order-independent); the caller zeroes amax_bits first.
//   quantise : scale = amax / FMAX (1 if the tensor is all zero), q = cvt(x / scale), FMAX = 448 (e4m3fn) or 57344 (e5m2).
template <typename T>
__global__ void __launch_bounds__(256) amax_tensor_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int nch, unsigned* __restrict__ amax_bits) {
    // four 16-byte loads in flight per thread, the maximum reduced in the workgroup, ONE atomic per workgroup: atomics on one address
    // pass the L2 one at a time (16,384 of them -- one per wave of 4,096 workgroups -- took 130 us on a 78 MB tensor; this form 17 us)
    __shared__ float red[4];
    float mx = 0.f;
    const int64_t total = rows * nch;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const bool flat = ldx == (int64_t)8 * nch;
    auto chunk = [&](int64_t i) -> const T* {
        if (flat) return x + 8 * i;
        const int64_t r = i / nch;
        return x + r * ldx + 8 * (i - r * nch);
    };
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < total; i += 4 * stride) {
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8<T>(chunk(i + u * stride), v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(v[u][j]));
    }
    for (; i < total; i += stride) {
        float v[8];
        load8<T>(chunk(i), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(v[j]));
    }
    mx = wave_max_all(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (mx > 0.f) atomicMax(amax_bits, __float_as_uint(mx));
    }
}
template <typename T, bool E5M2>
__global__ void __launch_bounds__(256) quant_tensor_fp8_kernel(const T* __restrict__ x, int64_t ldx, int64_t rows, int nch,
                                                               const unsigned* __restrict__ amax_bits, uint8_t* __restrict__ q, int64_t ldq,
                                                               float* __restrict__ scale_out) {
    const float amax = __uint_as_float(amax_bits[0]);
    const float s = amax > 0.f ? amax * (1.f / (E5M2 ? 57344.f : F8_MAX)) : 1.f;
    const float inv = 1.f / s;
    if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = s;
    const int64_t total = rows * nch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / nch;
        const int c = (int)(i - r * nch);
        float v[8];
        load8<T>(x + r * ldx + 8 * c, v);
        int lo = 0, hi = 0;
        if (E5M2) {
            lo = __builtin_amdgcn_cvt_pk_bf8_f32(v[0] * inv, v[1] * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_bf8_f32(v[2] * inv, v[3] * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_bf8_f32(v[4] * inv, v[5] * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_bf8_f32(v[6] * inv, v[7] * inv, hi, true);
        } else {
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
        }
        *reinterpret_cast<int2*>(q + r * ldq + 8 * c) = make_int2(lo, hi);
    }
}
// fmt 0 = e4m3fn, 1 = e5m2.  amax_ws: one uint32 of scratch (zeroed here with a kernel node: graph-safe).  scale: one float out.
__global__ void fp8_zero_word_kernel(unsigned* p) { p[0] = 0u; }
extern "C" int segf_quant_tensor_fp8(int dt, int fmt, int64_t rows, int cols, const void* x, int64_t ldx, void* q, int64_t ldq,
                                     float* scale, void* amax_ws, void* stream) {
    if (rows <= 0) return 0;
    if (cols <= 0 || cols % 8 || ldx < cols || ldq < cols || (ldq % 8) || ((uintptr_t)x % 16) || ((uintptr_t)q % 8) || !scale || !amax_ws ||
        fmt < 0 || fmt > 1)
        return SEGF_ERR_SHAPE;
    if ((ldx * (dt == SEGF_BF16 ? 2 : 4)) % 16) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int nch = cols / 8;
    const int blocks = (int)imin64(cdiv64(rows * nch, 256 * 4), 4096);
    const int ablocks = (int)imin64(cdiv64(rows * nch, 256 * 8), 1024);
    hipLaunchKernelGGL(fp8_zero_word_kernel, dim3(1), dim3(1), 0, st, (unsigned*)amax_ws);
    SEGF_DISPATCH_DT(dt, T, {
        hipLaunchKernelGGL((amax_tensor_kernel<T>), dim3(ablocks), dim3(256), 0, st, (const T*)x, ldx, rows, nch, (unsigned*)amax_ws);
        if (fmt == 0) hipLaunchKernelGGL((quant_tensor_fp8_kernel<T, false>), dim3(blocks), dim3(256), 0, st, (const T*)x, ldx, rows, nch,
                                         (const unsigned*)amax_ws, (uint8_t*)q, ldq, scale);
        else hipLaunchKernelGGL((quant_tensor_fp8_kernel<T, true>), dim3(blocks), dim3(256), 0, st, (const T*)x, ldx, rows, nch,
                                (const unsigned*)amax_ws, (uint8_t*)q, ldq, scale);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

// ---- GEMM ---------------------------------------------------------------------------------------------------------------
struct Fp8Args {
    const uint8_t* A; int64_t lda;       // [M][K] e4m3 (activations)
    const uint8_t* B; int64_t ldb;       // [N][K] e4m3 (weights, rows = output channels)
    const float* sa; const float* sb;    // [M], [N]
    const float* bias;                   // [N] or nullptr
    const bf16_t* residual; int64_t ldr; // [M][N] or nullptr: C = residual + rscale[m / rpg] * (...)
    const float* rscale; int64_t rpg;
    bf16_t* C; int64_t ldc;              // [M][N] bf16
    int64_t M, N, K;
};

__global__ void __launch_bounds__(256) gemm_fp8_kernel(Fp8Args a) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * F8_BM * F8_LD];       // A tile | B tile; reused by the epilogue
    uint8_t* la = lds;
    uint8_t* lb = lds + F8_BM * F8_LD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    // XCD-aware tile order: consecutive logical ids share the A row panel
    const int64_t tn = (a.N + F8_BN - 1) / F8_BN;
    const unsigned L = xcd_block();
    const int64_t m0 = (int64_t)(L / (unsigned)tn) * F8_BM, n0 = (int64_t)(L % (unsigned)tn) * F8_BN;
    // staging: thread t moves 4 x 16 bytes of each operand tile per K step: rows t/8 + 32 i, byte column 16 (t % 8)
    const int srow = threadIdx.x >> 3, scol = (threadIdx.x & 7) * 16;
    typedef int i32x4 __attribute__((ext_vector_type(4)));      // native vectors: arrays of HIP's int4 class were kept in scratch
    i32x4 ra[4], rb[4];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t m = m0 + srow + 32 * i, n = n0 + srow + 32 * i;
            m = m < a.M ? m : a.M - 1;                          // clamped rows: their products land in rows / columns that are never stored
            n = n < a.N ? n : a.N - 1;
            ra[i] = *reinterpret_cast<const i32x4*>(a.A + m * a.lda + k0 + scol);
            rb[i] = *reinterpret_cast<const i32x4*>(a.B + n * a.ldb + k0 + scol);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<i32x4*>(la + (srow + 32 * i) * F8_LD + scol) = ra[i];
            *reinterpret_cast<i32x4*>(lb + (srow + 32 * i) * F8_LD + scol) = rb[i];
        }
    };
    fp8_v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fp8_v4f{0.f, 0.f, 0.f, 0.f};
    gload(0);
    for (int64_t k0 = 0; k0 < a.K; k0 += F8_BK) {
        __syncthreads();                                        // the previous step's fragment reads are done
        lstore();
        __syncthreads();
        gload(k0 + F8_BK < a.K ? k0 + F8_BK : k0);              // next tile in flight during the MFMAs (unconditional: countable loads)
        fp8_v8i fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const uint8_t* pa = la + (wm * 64 + t * 16 + li) * F8_LD + 32 * lg;
            const uint8_t* pb = lb + (wn * 64 + t * 16 + li) * F8_LD + 32 * lg;
            const int4 a0 = *reinterpret_cast<const int4*>(pa), a1 = *reinterpret_cast<const int4*>(pa + 16);
            const int4 b0 = *reinterpret_cast<const int4*>(pb), b1 = *reinterpret_cast<const int4*>(pb + 16);
            fa[t] = fp8_v8i{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            fb[t] = fp8_v8i{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[i], fb[j], acc[i][j], 0 /* A: e4m3 */, 0 /* B: e4m3 */,
                                                                             0, 0x7f7f7f7f /* block scales 2^0 */, 0, 0x7f7f7f7f);
    }
    // epilogue: acc * sx[m] * sw[n] + bias[n] -> bf16 tile image in LDS ([128][128 + 8] bf16 = 34816 B) -> 16-byte global stores
    __syncthreads();
    bf16_t* lc = reinterpret_cast<bf16_t*>(lds);
    constexpr int LCD = F8_BN + 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cn = wn * 64 + j * 16 + li;
        const int64_t n = n0 + cn;
        const float sw = n < a.N ? a.sb[n] : 0.f, bz = (a.bias && n < a.N) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cm = wm * 64 + i * 16 + 4 * lg + r;
                const int64_t m = m0 + cm;
                const float sx = m < a.M ? a.sa[m] : 0.f;
                lc[cm * LCD + cn] = f2bf(fmaf(acc[i][j][r], sx * sw, bz));
            }
        }
    }
    __syncthreads();
    const bool vec = (a.ldc % 8 == 0) && ((uintptr_t)a.C % 16 == 0);
    for (int q = threadIdx.x; q < F8_BM * (F8_BN / 8); q += 256) {
        const int cm = q / (F8_BN / 8), c8 = (q % (F8_BN / 8)) * 8;
        const int64_t m = m0 + cm, n = n0 + c8;
        if (m >= a.M || n >= a.N) continue;
        const bf16_t* src = lc + cm * LCD + c8;
        bf16_t* dst = a.C + m * a.ldc + n;
        if (a.residual) {                                       // x + drop_path(linear(..)) (convnextv2.py:108-112): residual + rscale * v
            const float rs = a.rscale ? a.rscale[m / a.rpg] : 1.f;
            const bf16_t* rp = a.residual + m * a.ldr + n;
            for (int j = 0; j < 8 && n + j < a.N; ++j) dst[j] = f2bf(fmaf(rs, bf2f(src[j]), bf2f(rp[j])));
        } else if (vec && n + 8 <= a.N) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
        else for (int j = 0; j < 8 && n + j < a.N; ++j) dst[j] = src[j];
    }
}

extern "C" int segf_gemm_fp8_supported(int64_t M, int64_t N, int64_t K) {
    return (M > 0 && N > 0 && K >= F8_BK && K % F8_BK == 0 && !POL(no_fp8)) ? 1 : 0;
}
extern "C" int segf_gemm_fp8(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const float* scale_a, const void* B,
                             int64_t ldb, const float* scale_b, const float* bias, const void* residual, int64_t ldr,
                             const float* rscale, int64_t rows_per_group, void* C, int64_t ldc, void* stream) {
    if (!segf_gemm_fp8_supported(M, N, K) || lda < K || ldb < K || ldc < N) return SEGF_ERR_SHAPE;
    if (((uintptr_t)A % 16) || ((uintptr_t)B % 16) || (lda % 16) || (ldb % 16) || !scale_a || !scale_b) return SEGF_ERR_SHAPE;
    const int64_t tiles = cdiv64(M, F8_BM) * cdiv64(N, F8_BN);
    if (tiles > 0x7fffffff) return SEGF_ERR_SHAPE;
    if (rscale && rows_per_group <= 0) return SEGF_ERR_SHAPE;
    Fp8Args a{(const uint8_t*)A, lda, (const uint8_t*)B, ldb, scale_a, scale_b, bias, (const bf16_t*)residual, ldr, rscale,
              rows_per_group > 0 ? rows_per_group : 1, (bf16_t*)C, ldc, M, N, K};
    hipLaunchKernelGGL(gemm_fp8_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    SEGF_CHECK_LAUNCH();
    return 0;
}
