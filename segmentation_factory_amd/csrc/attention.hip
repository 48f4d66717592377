// MiT spatial-reduction attention core: O = softmax(Q K^T * scale) V per (batch, head)
// (reference models/backbones/mit.py:52-57).  K/V are the <= 256-token (512^2) or 2048-token
// (1024x2048) reduced sequence, head_dim 32 (B0) or 64 (B1-B5).
//
// Round-1 implementation: exact-fp32 online-softmax on the vector ALU.  One query per HD/32 lanes,
// K/V tiles broadcast from LDS (every lane of a wave reads the same K row -> LDS broadcast, no bank
// conflicts), scores never leave registers, log-sum-exp saved for the backward.  Attention is 2.1 of
// the 89 GFLOP/img of SegFormer-B0 as the reference builds it, so the fp32 VALU rate
// is not the step's bottleneck; an MFMA variant is the planned follow-up.
// Backward = two passes sharing the recomputed probabilities P = exp(S - lse):
//   pass 1 (query-parallel): D = rowsum(dO*O), dQ = scale * sum_j P(dP - D) K_j
//   pass 2 (key-parallel, query chunks -> deterministic partial slabs): dK_j, dV_j
#include <stdlib.h>
#include "attention_mfma.h"

#define AT_KT 64        // K/V rows per LDS tile
#define AT_QT 32        // query rows per LDS tile in the key-parallel pass
#define AT_THREADS 256

template <typename T>
__device__ __forceinline__ void load32(const T* p, bool vec, float (&v)[32]) {
    if (vec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t[8];
            load8<T>(p + 8 * i, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[8 * i + j] = t[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = ldf<T>(p + j);
    }
}
template <typename T>
__device__ __forceinline__ void store32(T* p, bool vec, const float (&v)[32]) {
    if (vec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = v[8 * i + j];
            store8<T>(p + 8 * i, t);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) stf<T>(p + j, v[j]);
    }
}

// cooperative tile load: rows [r0, r0+nrows) x HD columns of a [rows][ld] matrix -> LDS fp32 [nrows][HD]; rows >= rmax -> 0
template <typename T, int HD>
__device__ __forceinline__ void stage_rows(const T* __restrict__ base, int64_t ld, int64_t r0, int64_t rmax, int nrows,
                                           float* __restrict__ dst) {
    for (int i = threadIdx.x; i < nrows * HD; i += AT_THREADS) {
        const int r = i / HD, d = i - r * HD;
        dst[i] = (r0 + r < rmax) ? ldf<T>(base + (r0 + r) * ld + d) : 0.f;
    }
}

template <int TPR> __device__ __forceinline__ float part_sum(float v) {
    if (TPR == 2) v += __shfl_xor(v, 1, 64);
    return v;
}

template <typename T, int HD>
__global__ void __launch_bounds__(AT_THREADS) attn_fwd_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k,
                                                               int64_t ldk, const T* __restrict__ v, int64_t ldv,
                                                               T* __restrict__ o, int64_t ldo, float* __restrict__ lse,
                                                               int heads, int N, int Nkv, float scale, int vec) {
    constexpr int TPR = HD / 32;
    constexpr int QPB = AT_THREADS / TPR;
    __shared__ float Ks[AT_KT * HD];
    __shared__ float Vs[AT_KT * HD];
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * QPB + threadIdx.x / TPR;
    const int part = threadIdx.x % TPR;
    const bool qv = qi < N;
    float qr[32], acc[32];
    if (qv) load32<T>(q + ((int64_t)b * N + qi) * ldq + h * HD + part * 32, vec, qr);
#pragma unroll
    for (int d = 0; d < 32; ++d) { acc[d] = 0.f; if (!qv) qr[d] = 0.f; }
    float m = -INFINITY, l = 0.f;
    const T* kb = k + (int64_t)b * Nkv * ldk + h * HD;
    const T* vb = v + (int64_t)b * Nkv * ldv + h * HD;
    for (int j0 = 0; j0 < Nkv; j0 += AT_KT) {
        __syncthreads();
        stage_rows<T, HD>(kb, ldk, j0, Nkv, AT_KT, Ks);
        stage_rows<T, HD>(vb, ldv, j0, Nkv, AT_KT, Vs);
        __syncthreads();
        const int jn = Nkv - j0 < AT_KT ? Nkv - j0 : AT_KT;
        for (int jb = 0; jb < jn; jb += 16) {
            float sc[16];
            float mb = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const float* kr = Ks + (jb + jj) * HD + part * 32;
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) s = fmaf(qr[d], kr[d], s);
                s = part_sum<TPR>(s) * scale;
                if (jb + jj >= jn) s = -INFINITY;
                sc[jj] = s;
                mb = fmaxf(mb, s);
            }
            const float mn = fmaxf(m, mb);
            const float alpha = __expf(m - mn);     // m = -inf on the first block -> 0
            l *= alpha;
#pragma unroll
            for (int d = 0; d < 32; ++d) acc[d] *= alpha;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const float p = __expf(sc[jj] - mn);   // masked rows: exp(-inf) = 0
                l += p;
                const float* vr = Vs + (jb + jj) * HD + part * 32;
#pragma unroll
                for (int d = 0; d < 32; ++d) acc[d] = fmaf(p, vr[d], acc[d]);
            }
            m = mn;
        }
    }
    if (qv) {
        const float inv = 1.f / l;
#pragma unroll
        for (int d = 0; d < 32; ++d) acc[d] *= inv;
        store32<T>(o + ((int64_t)b * N + qi) * ldo + h * HD + part * 32, vec, acc);
        if (part == 0) lse[((int64_t)b * heads + h) * N + qi] = m + __logf(l);
    }
}

// ---- r05: the fp32 forward on the f32 matrix instruction (what `evaluate` runs: engine.py:86-88 evaluates in fp32) -------------------
// v_mfma_f32_32x32x2_f32 is exact fp32 (every product rounded once, fp32 accumulation) at twice the vector FMA rate, and -- what matters
// at the reference's --val_batch_size 1 -- it replaces the vector kernel's 16 384 serial FMAs per lane (one query per lane, 256 keys x
// 32 features, twice) by 32 MFMAs per 32 keys: the stage-1 call of SegFormer-B0 at batch 1 is 64 workgroups of that serial chain.
// One wave = 32 queries, no LDS.  Scores are formed TRANSPOSED, S^T[key][query] = K Q^T, so that a lane ends up with 16 of the 32 keys
// of ITS OWN query (column = lane & 31; the other 16 keys sit in lane ^ 32): the online-softmax state is per lane, one cross-half
// exchange for the maximum and one for the sum.  The probabilities then ARE the B operand of the second product as they stand:
// O^T[d][query] += V^T[d][key] P^T[key][query] with the keys taken in the order the score tile holds them (rows (r & 3) + 8 (r >> 2)
// + 4 half: a sum over keys does not care), V rows read straight from global memory (128 contiguous bytes per half-wave).
typedef float at_f32x16 __attribute__((ext_vector_type(16)));
template <int HD>
__global__ void __launch_bounds__(64) attn_f32_mfma_fwd_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                                int64_t ldk, const float* __restrict__ v, int64_t ldv,
                                                                float* __restrict__ o, int64_t ldo, float* __restrict__ lse,
                                                                int heads, int N, int Nkv, float scale) {
    constexpr int KD = HD / 2, DT = HD / 32;
    const int lane = threadIdx.x, j = lane & 31, h2 = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 32;
    const int qi = q0 + j < N ? q0 + j : N - 1;
    const float2* qrow = reinterpret_cast<const float2*>(q + ((int64_t)b * N + qi) * ldq + head * HD);
    float qf[KD];
#pragma unroll
    for (int t = 0; t < KD; ++t) { const float2 u = qrow[t]; qf[t] = h2 ? u.y : u.x; }
    at_f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float* kb = k + (int64_t)b * Nkv * ldk + head * HD;
    const float* vb = v + (int64_t)b * Nkv * ldv + head * HD;
    float kf[KD];
    {
        const int ki = j < Nkv ? j : Nkv - 1;
        const float2* krow = reinterpret_cast<const float2*>(kb + (int64_t)ki * ldk);
#pragma unroll
        for (int t = 0; t < KD; ++t) { const float2 u = krow[t]; kf[t] = h2 ? u.y : u.x; }
    }
    for (int j0 = 0; j0 < Nkv; j0 += 32) {
        at_f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int t = 0; t < KD; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[t], s, 0, 0, 0);
        // the V rows of this tile (keys in the order of the score registers) and the next tile's K rows fly under the softmax
        float vf[DT][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
            const float* vr = vb + (int64_t)(key < Nkv ? key : Nkv - 1) * ldv + j;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vf[dt][r] = vr[32 * dt];
        }
        if (j0 + 32 < Nkv) {
            const int ki = j0 + 32 + j < Nkv ? j0 + 32 + j : Nkv - 1;
            const float2* krow = reinterpret_cast<const float2*>(kb + (int64_t)ki * ldk);
#pragma unroll
            for (int t = 0; t < KD; ++t) { const float2 u = krow[t]; kf[t] = h2 ? u.y : u.x; }
        }
        float mb = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = j0 + (r & 3) + 8 * (r >> 2) + 4 * h2;
            const float sv = key < Nkv ? s[r] * scale : -INFINITY;
            s[r] = sv;
            mb = fmaxf(mb, sv);
        }
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float mn = fmaxf(m, mb);
        const float alpha = __expf(m - mn);          // m = -inf on the first tile -> 0
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float p = __expf(s[r] - mn); s[r] = p; ps += p; }
        ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[dt][r], s[r], oacc[dt], 0, 0, 0);
    }
    if (q0 + j < N) {
        const float inv = 1.f / l;
        float* orow = o + ((int64_t)b * N + q0 + j) * ldo + head * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)       // rows d = 32 dt + 8 g + 4 h2 + (0 .. 3) of this query's column
                *reinterpret_cast<float4*>(orow + 32 * dt + 8 * g + 4 * h2) =
                    make_float4(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
        if (h2 == 0) lse[((int64_t)b * heads + head) * N + q0 + j] = m + __logf(l);
    }
}

// pass 1: D and dQ, one query per TPR lanes
template <typename T, int HD>
__global__ void __launch_bounds__(AT_THREADS) attn_bwd_dq_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k,
                                                                  int64_t ldk, const T* __restrict__ v, int64_t ldv,
                                                                  const T* __restrict__ o, int64_t ldo, const T* __restrict__ d_o,
                                                                  int64_t lddo, const float* __restrict__ lse,
                                                                  T* __restrict__ dq, int64_t lddq, float* __restrict__ Dbuf,
                                                                  int heads, int N, int Nkv, float scale, int vec) {
    constexpr int TPR = HD / 32;
    constexpr int QPB = AT_THREADS / TPR;
    __shared__ float Ks[AT_KT * HD];
    __shared__ float Vs[AT_KT * HD];
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * QPB + threadIdx.x / TPR;
    const int part = threadIdx.x % TPR;
    const bool qv = qi < N;
    float qr[32], dor[32], acc[32];
    float D = 0.f, L = 0.f;
    if (qv) {
        const int64_t row = (int64_t)b * N + qi;
        load32<T>(q + row * ldq + h * HD + part * 32, vec, qr);
        load32<T>(d_o + row * lddo + h * HD + part * 32, vec, dor);
        load32<T>(o + row * ldo + h * HD + part * 32, vec, acc);
#pragma unroll
        for (int d = 0; d < 32; ++d) D = fmaf(dor[d], acc[d], D);
        L = lse[((int64_t)b * heads + h) * N + qi];
    } else {
#pragma unroll
        for (int d = 0; d < 32; ++d) { qr[d] = 0.f; dor[d] = 0.f; }
    }
    D = part_sum<TPR>(D);
    if (qv && part == 0) Dbuf[((int64_t)b * heads + h) * N + qi] = D;
#pragma unroll
    for (int d = 0; d < 32; ++d) acc[d] = 0.f;
    const T* kb = k + (int64_t)b * Nkv * ldk + h * HD;
    const T* vb = v + (int64_t)b * Nkv * ldv + h * HD;
    for (int j0 = 0; j0 < Nkv; j0 += AT_KT) {
        __syncthreads();
        stage_rows<T, HD>(kb, ldk, j0, Nkv, AT_KT, Ks);
        stage_rows<T, HD>(vb, ldv, j0, Nkv, AT_KT, Vs);
        __syncthreads();
        const int jn = Nkv - j0 < AT_KT ? Nkv - j0 : AT_KT;
        for (int j = 0; j < jn; ++j) {
            const float* kr = Ks + j * HD + part * 32;
            const float* vr = Vs + j * HD + part * 32;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s = fmaf(qr[d], kr[d], s); dp = fmaf(dor[d], vr[d], dp); }
            s = part_sum<TPR>(s) * scale;
            dp = part_sum<TPR>(dp);
            const float p = qv ? __expf(s - L) : 0.f;
            const float ds = p * (dp - D) * scale;
#pragma unroll
            for (int d = 0; d < 32; ++d) acc[d] = fmaf(ds, kr[d], acc[d]);
        }
    }
    if (qv) store32<T>(dq + ((int64_t)b * N + qi) * lddq + h * HD + part * 32, vec, acc);
}

// pass 2: dK, dV.  One key row per TPR lanes; grid.y = query chunk; partial slabs [chunk][B*Nkv][2][heads*HD]
template <typename T, int HD>
__global__ void __launch_bounds__(AT_THREADS) attn_bwd_dkv_kernel(const T* __restrict__ q, int64_t ldq, const T* __restrict__ k,
                                                                   int64_t ldk, const T* __restrict__ v, int64_t ldv,
                                                                   const T* __restrict__ d_o, int64_t lddo,
                                                                   const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                                   float* __restrict__ slab, int heads, int N, int Nkv, int B,
                                                                   int qchunk, float scale, int vec) {
    constexpr int TPR = HD / 32;
    constexpr int KPB = AT_THREADS / TPR;
    __shared__ float Qs[AT_QT * HD];
    __shared__ float Gs[AT_QT * HD];
    __shared__ float Ls[AT_QT];
    __shared__ float Ds[AT_QT];
    const int bh = blockIdx.z, b = bh / heads, h = bh - b * heads;
    const int kj = blockIdx.x * KPB + threadIdx.x / TPR;
    const int part = threadIdx.x % TPR;
    const bool kvld = kj < Nkv;
    float kr[32], vr[32], dk[32], dv[32];
    if (kvld) {
        load32<T>(k + ((int64_t)b * Nkv + kj) * ldk + h * HD + part * 32, vec, kr);
        load32<T>(v + ((int64_t)b * Nkv + kj) * ldv + h * HD + part * 32, vec, vr);
    }
#pragma unroll
    for (int d = 0; d < 32; ++d) { dk[d] = 0.f; dv[d] = 0.f; if (!kvld) { kr[d] = 0.f; vr[d] = 0.f; } }
    const int q0 = blockIdx.y * qchunk;
    const int q1 = q0 + qchunk < N ? q0 + qchunk : N;
    const T* qb = q + (int64_t)b * N * ldq + h * HD;
    const T* gb = d_o + (int64_t)b * N * lddo + h * HD;
    const float* lb = lse + ((int64_t)b * heads + h) * N;
    const float* db = Dbuf + ((int64_t)b * heads + h) * N;
    for (int i0 = q0; i0 < q1; i0 += AT_QT) {
        __syncthreads();
        stage_rows<T, HD>(qb, ldq, i0, q1, AT_QT, Qs);
        stage_rows<T, HD>(gb, lddo, i0, q1, AT_QT, Gs);
        if (threadIdx.x < AT_QT) {
            const int i = i0 + threadIdx.x;
            Ls[threadIdx.x] = i < q1 ? lb[i] : 0.f;
            Ds[threadIdx.x] = i < q1 ? db[i] : 0.f;
        }
        __syncthreads();
        const int in = q1 - i0 < AT_QT ? q1 - i0 : AT_QT;
        for (int i = 0; i < in; ++i) {
            const float* qr = Qs + i * HD + part * 32;
            const float* gr = Gs + i * HD + part * 32;
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) { s = fmaf(qr[d], kr[d], s); dp = fmaf(gr[d], vr[d], dp); }
            s = part_sum<TPR>(s) * scale;
            dp = part_sum<TPR>(dp);
            const float p = kvld ? __expf(s - Ls[i]) : 0.f;
            const float ds = p * (dp - Ds[i]) * scale;
#pragma unroll
            for (int d = 0; d < 32; ++d) { dv[d] = fmaf(p, gr[d], dv[d]); dk[d] = fmaf(ds, qr[d], dk[d]); }
        }
    }
    if (kvld) {
        const int C = heads * HD;
        float* dst = slab + (((int64_t)blockIdx.y * B * Nkv + (int64_t)b * Nkv + kj) * 2) * C + h * HD + part * 32;
#pragma unroll
        for (int d = 0; d < 32; ++d) { dst[d] = dk[d]; dst[C + d] = dv[d]; }
    }
}

// dk[r][c] = sum_chunk slab[chunk][r][0][c]; dv likewise
template <typename T>
__global__ void attn_dkv_reduce_kernel(const float* __restrict__ slab, int nchunk, int64_t rows, int C, T* __restrict__ dk,
                                       int64_t lddk, T* __restrict__ dv, int64_t lddv) {
    const int64_t total = rows * 2 * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < nchunk; ++z) s += slab[(int64_t)z * total + i];
        const int64_t r = i / (2 * C);
        const int rem = (int)(i - r * 2 * C);
        if (rem < C) stf<T>(dk + r * lddk + rem, s);
        else stf<T>(dv + r * lddv + (rem - C), s);
    }
}

static inline int attn_vec_ok(int dt, const void* p, int64_t ld) {
    const int64_t esz = dt == SEGF_BF16 ? 2 : 4;
    return ((uintptr_t)p % 16 == 0) && ((ld * esz) % 16 == 0);
}
// Query chunks of the key-side backward (grid = key blocks x chunks x (batch, head); every workgroup stages its keys once and walks its
// chunk's queries; the chunks' partial dK / dV slabs are summed by attn_dkv_reduce_kernel).  r04's rule asked for >= 512 workgroups
// (two per CU) and stopped there: 192 images x 1 head gave 3 chunks = 576 workgroups = one full round + one of 64 -- the kernel cost 28 %
// more per image at batch 192 than at 256 (profiles/r05_batch_sweep.txt: why 192 was SLOWER than 128).  Now the count minimises
// rounds x (queries per chunk + a fixed per-workgroup cost), the model of a grid that runs in whole rounds of 512 resident workgroups:
// the same choices at power-of-two batches (256 x 1 head: 2 chunks = 512 workgroups), 8 chunks = three full rounds at 192.
static inline void attn_chunks(int B, int heads, int N, int Nkv, int hd, int& nchunk, int& qchunk) {
    const int tpr = hd / 32, kpb = AT_THREADS / tpr;
    const int kvtiles = (Nkv + kpb - 1) / kpb;
    const int64_t base = (int64_t)B * heads * kvtiles;            // workgroups per chunk
    int maxc = (N + 63) / 64;
    if (maxc > 64) maxc = 64;
    if (maxc < 1) maxc = 1;
    const int64_t cap = 512;                                      // resident workgroups (two per CU)
    const double fixed = 256.0;                                   // per-workgroup prologue (key staging) in units of one query's work
    double best = 0.0;
    int bc = 1;
    for (int c = 1; c <= maxc; ++c) {
        const int q = ((N + c - 1) / c + AT_QT - 1) / AT_QT * AT_QT;
        const int n = (N + q - 1) / q;
        if (n != c && c > 1) continue;                            // (the same partition as a smaller count)
        const int64_t wg = base * n;
        const int64_t rounds = (wg + cap - 1) / cap;
        // below one round the chip is not full: time falls with every further chunk until it is (the r04 rule); the 0.5 % per chunk
        // breaks ties towards fewer slabs to write and sum
        const double t = (double)rounds * ((double)q + fixed) * (1.0 + 0.005 * n);
        if (c == 1 || t < best) { best = t; bc = n; }
    }
    qchunk = ((N + bc - 1) / bc + AT_QT - 1) / AT_QT * AT_QT;
    nchunk = (N + qchunk - 1) / qchunk;
}

extern "C" int segf_attention_fwd(int dt, int B, int heads, int N, int Nkv, int hd, const void* q, int64_t ldq, const void* k,
                                  int64_t ldk, const void* v, int64_t ldv, float scale, void* o, int64_t ldo, float* lse,
                                  void* stream) {
    if (B <= 0 || heads <= 0 || N <= 0) return 0;
    if (Nkv <= 0 || (hd != 32 && hd != 64) || heads > 65535 || B > 65535) return SEGF_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const int vec = attn_vec_ok(dt, q, ldq) && attn_vec_ok(dt, k, ldk) && attn_vec_ok(dt, v, ldv) && attn_vec_ok(dt, o, ldo);
    if (dt == SEGF_BF16 && vec && !POL(attn_no_mfma))
        return attn_mfma_fwd(hd, B, heads, N, Nkv, q, ldq, k, ldk, v, ldv, scale, o, ldo, lse, st);
    if (dt == SEGF_F32 && vec && !POL(attn_f32_no_mfma)) {
        const dim3 gridm((unsigned)((N + 31) / 32), (unsigned)heads, (unsigned)B);
        if (hd == 32)
            hipLaunchKernelGGL((attn_f32_mfma_fwd_kernel<32>), gridm, dim3(64), 0, st, (const float*)q, ldq, (const float*)k, ldk,
                               (const float*)v, ldv, (float*)o, ldo, lse, heads, N, Nkv, scale);
        else
            hipLaunchKernelGGL((attn_f32_mfma_fwd_kernel<64>), gridm, dim3(64), 0, st, (const float*)q, ldq, (const float*)k, ldk,
                               (const float*)v, ldv, (float*)o, ldo, lse, heads, N, Nkv, scale);
        SEGF_CHECK_LAUNCH();
        return 0;
    }
    const int qpb = AT_THREADS / (hd / 32);
    dim3 grid((N + qpb - 1) / qpb, heads, B);
    SEGF_DISPATCH_DT(dt, T, {
        if (hd == 32)
            hipLaunchKernelGGL((attn_fwd_kernel<T, 32>), grid, dim3(AT_THREADS), 0, st, (const T*)q, ldq, (const T*)k, ldk,
                               (const T*)v, ldv, (T*)o, ldo, lse, heads, N, Nkv, scale, vec);
        else
            hipLaunchKernelGGL((attn_fwd_kernel<T, 64>), grid, dim3(AT_THREADS), 0, st, (const T*)q, ldq, (const T*)k, ldk,
                               (const T*)v, ldv, (T*)o, ldo, lse, heads, N, Nkv, scale, vec);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t segf_attention_bwd_ws(int B, int heads, int N, int Nkv, int hd) {
    if (hd != 32 && hd != 64) return 0;
    int nchunk, qchunk;
    attn_chunks(B, heads, N, Nkv, hd, nchunk, qchunk);
    return (int64_t)B * heads * N + (int64_t)nchunk * B * Nkv * 2 * heads * hd;
}

extern "C" int segf_attention_bwd(int dt, int B, int heads, int N, int Nkv, int hd, const void* q, int64_t ldq, const void* k,
                                  int64_t ldk, const void* v, int64_t ldv, float scale, const void* o, int64_t ldo,
                                  const void* d_o, int64_t lddo, const float* lse, void* dq, int64_t lddq, void* dk,
                                  int64_t lddk, void* dv, int64_t lddv, float* ws, void* stream) {
    if (B <= 0 || heads <= 0 || N <= 0) return 0;
    if (Nkv <= 0 || (hd != 32 && hd != 64) || heads > 65535 || B > 65535) return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int vec = attn_vec_ok(dt, q, ldq) && attn_vec_ok(dt, k, ldk) && attn_vec_ok(dt, v, ldv) && attn_vec_ok(dt, o, ldo) &&
                    attn_vec_ok(dt, d_o, lddo) && attn_vec_ok(dt, dq, lddq);
    int nchunk, qchunk;
    attn_chunks(B, heads, N, Nkv, hd, nchunk, qchunk);
    float* Dbuf = ws;
    float* slab = ws + (int64_t)B * heads * N;
    const int tpr = hd / 32;
    const int qpb = AT_THREADS / tpr, kpb = AT_THREADS / tpr;
    dim3 g1((N + qpb - 1) / qpb, heads, B);
    dim3 g2((Nkv + kpb - 1) / kpb, nchunk, B * heads);
    if (g2.z > 65535u) return SEGF_ERR_SHAPE;
    const int C = heads * hd;
    const int64_t rows = (int64_t)B * Nkv;
    const int rblocks = (int)imin64(cdiv64(rows * 2 * C, 256), 2048);
    if (dt == SEGF_BF16 && vec && attn_vec_ok(dt, dk, lddk) && attn_vec_ok(dt, dv, lddv) && !POL(attn_no_mfma)) {
        // the slab rows are written with 16-byte stores: 2*C*4 bytes per row is always a multiple of 16
        const int rc = attn_mfma_bwd(hd, B, heads, N, Nkv, q, ldq, k, ldk, v, ldv, scale, o, ldo, d_o, lddo, lse, dq, lddq, Dbuf,
                                     slab, nchunk, qchunk, st);
        if (rc) return rc;
        hipLaunchKernelGGL((attn_dkv_reduce_kernel<bf16_t>), dim3(rblocks), dim3(256), 0, st, slab, nchunk, rows, C, (bf16_t*)dk,
                           lddk, (bf16_t*)dv, lddv);
        SEGF_CHECK_LAUNCH();
        return 0;
    }
    SEGF_DISPATCH_DT(dt, T, {
        if (hd == 32) {
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, 32>), g1, dim3(AT_THREADS), 0, st, (const T*)q, ldq, (const T*)k, ldk,
                               (const T*)v, ldv, (const T*)o, ldo, (const T*)d_o, lddo, lse, (T*)dq, lddq, Dbuf, heads, N, Nkv, scale, vec);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 32>), g2, dim3(AT_THREADS), 0, st, (const T*)q, ldq, (const T*)k, ldk,
                               (const T*)v, ldv, (const T*)d_o, lddo, lse, Dbuf, slab, heads, N, Nkv, B, qchunk, scale, vec);
        } else {
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, 64>), g1, dim3(AT_THREADS), 0, st, (const T*)q, ldq, (const T*)k, ldk,
                               (const T*)v, ldv, (const T*)o, ldo, (const T*)d_o, lddo, lse, (T*)dq, lddq, Dbuf, heads, N, Nkv, scale, vec);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, 64>), g2, dim3(AT_THREADS), 0, st, (const T*)q, ldq, (const T*)k, ldk,
                               (const T*)v, ldv, (const T*)d_o, lddo, lse, Dbuf, slab, heads, N, Nkv, B, qchunk, scale, vec);
        }
        hipLaunchKernelGGL((attn_dkv_reduce_kernel<T>), dim3(rblocks), dim3(256), 0, st, slab, nchunk, rows, C, (T*)dk, lddk,
                           (T*)dv, lddv);
    })
    SEGF_CHECK_LAUNCH();
    return 0;
}
