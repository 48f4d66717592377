// Shared by loss.hip and loss_band.hip: geometry of the fused upsample + CE + Dice kernels and a few wave64 helpers.
#pragma once
#include "common.h"

#define LS_NBLK 128         // workgroups per image in the forward / eval kernels
#define LS_THREADS 256
#define LS_EPS 1e-6f
#define LS_TILE 8           // backward (tile kernels): low-res taps per workgroup tile edge
#define LS_LOG2E 1.4426950408889634f
#define LS_LN2 0.6931471805599453f

struct LossGeom { int B, C, h, w, H, W; int64_t ldl; };

typedef float lossf4 __attribute__((ext_vector_type(4)));
typedef __bf16 lossbf8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float tap_weight(int k, float ly, float lx) {      // k = 2 * (row tap) + (column tap)
    return ((k >> 1) ? ly : 1.f - ly) * ((k & 1) ? lx : 1.f - lx);
}
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v); v += dpp_mov<DPP_HALF_MIRROR>(v); v += dpp_mov<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float sel4(const float (&v)[4], int r) {
    return r == 0 ? v[0] : (r == 1 ? v[1] : (r == 2 ? v[2] : v[3]));
}

__device__ __forceinline__ void dice_coef_one(const float* __restrict__ stats, int b, int B, int C, int dice, int cls, float& gI, float& gP) {
    gI = 0.f; gP = 0.f;
    if (cls < C && dice) {
        const float* st = stats + (int64_t)b * (3 * C + 4);
        const float I = st[cls], P = st[C + cls], Tt = st[2 * C + cls];
        const float sets = P + Tt;
        if (sets != 0.f) {
            const float nbc = 1.f / (float)(B * C);
            gI = -nbc * 2.f / (sets + LS_EPS);
            gP = nbc * (2.f * I + LS_EPS) / ((sets + LS_EPS) * (sets + LS_EPS));
        }
    }
}

// loss_band.hip: band-sweep backward for ratio 4 / bf16; returns false when the configuration is not covered
// lse (nullable): per (image, cell, pixel) -log2(sum exp), written by the forward and consumed by the backward (LSE variant)
bool loss_band_fwd_covers(const bf16_t* logits, LossGeom g);
bool loss_band_bwd_launch(const bf16_t* logits, LossGeom g, const int64_t* target, int64_t ignore_index, const float* cw, int dice,
                          const float* stats, const float* grad_out, bf16_t* dlow, int64_t ldd, int* retry, const float* lse,
                          hipStream_t st);
bool loss_band_fwd_launch(const bf16_t* logits, LossGeom g, const int64_t* target, int64_t ignore_index, const float* cw,
                          float* partial, int* retry, float* lse, hipStream_t st);
