// BatchNorm backward of SegFormerHead's fuse ConvModule with the classifier's data gradient folded in
//   reference: heads/segformer.py:21-29,40,57-58  (linear_fuse.bn / activate -> Dropout2d -> linear_pred), backward of
//              a = relu(bn(x)) * drop;  y = a W^T + b
// The tiled path runs three launches over [B*h*w, 768] tensors: da = dy W (writes 3.2 GB at cfg2, batch 128), the column sums
// of BatchNorm's backward (reads x and da), and its apply pass (reads x and da again, writes dx): 19.8 GB of traffic.
// da[token][feature] = sum_class dy[token][class] W[class][feature] has only K = #classes (<= 192) terms, so it is cheaper to
// RECOMPUTE than to store: both BatchNorm passes below rebuild their da tile on the matrix pipe from the dy rows (0.64 GB) and
// a register-resident slice of W, and da is never written or read (10.9 GB).
// Structure = the streaming skinny GEMM of gemm.hip: a wave owns 32 output features, keeps their K x 32 weight fragments in
// registers, walks 16-token groups, computes the product transposed (C^T = W^T dy^T: the dy rows ARE the MFMA B operand as
// they lie in memory) with the feature rows permuted so that each lane ends up with 8 consecutive features of one token =
// one 16-byte access into x / dx.  The four waves of a workgroup take four neighbouring feature slices of the SAME tokens
// (their dy loads hit the same lines).  Per-feature BatchNorm constants sit in registers (column-fixed lanes).
//   pass 1: partial[blk][2][C] = per-workgroup sums of g and g * xhat  (g = da * drop * relu'(bn(x)));  finalize -> dbeta, dgamma
//   pass 2: dx = gamma rstd (g - mean(g) - xhat mean(g xhat))          (eval mode: dx = gamma rstd g)
// Deterministic: fixed token order per lane, DPP row sums, fixed-order finalize.
#include "colreduce.h"

typedef __attribute__((ext_vector_type(8))) __bf16 hf_bf16x8;
typedef __attribute__((ext_vector_type(4))) float hf_f32x4;
typedef __attribute__((ext_vector_type(8))) short hf_s16x8;

struct HeadFusedArgs {
    const bf16_t* dy; int64_t ldy;        // [M][>= 32 KS] class gradients (pad columns zero)
    const bf16_t* w; int64_t ldw;         // [32 KS][C] classifier weight, rows = classes (pad rows zero)
    const bf16_t* x;                      // [M][C] input of the BatchNorm
    const float *mean, *rstd, *gamma, *beta;
    const float* cscale;                  // [B][C] Dropout2d scale or nullptr
    const float* sums;                    // pass 2: [2][C] = {dbeta, dgamma} from pass 1
    bf16_t* dx;                           // pass 2
    float* partial;                       // pass 1: [gridDim.x][2][C]
    int64_t M; int C; int64_t rps; int groups_per_sample, chunks_per_sample; int act, eval_mode;
    const bf16_t* x1; int64_t ldx1;       // DW: [M][>= 32] second operand of the riding weight-gradient product
    float* dwpart;                        // DW: [gridDim.x / ny][C][DW_LD] per-workgroup partials of dx^T [x1 | 1]
    float* cwpart;                        // CW: [gridDim.x / ny][32 KS][C] per-workgroup partials of the classifier's weight gradient
};
#define DW_C1 32                          // channels of x1 (MiT-B0's stage-1 width)
#define DW_LD 40                          // row length of the product: 32 channels, the all-ones column (= column sums of dx), 7 x 0

// fragment of a row-major [token][column] bf16 LDS tile with the tokens as the contraction index: lane (i = lane & 15, g = lane >> 4)
// gets column cb + i of tokens 8 g .. 8 g + 7 (ds_read_b64_tr_b16, two 4-row blocks)
__device__ __forceinline__ hf_bf16x8 hf_frag_tok_tr(const unsigned char* tile, int rowbytes, int cb, int lane) {
    typedef __attribute__((ext_vector_type(4))) short hf_s16x4;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned char* a0 = tile + (8 * g + q) * rowbytes + (cb + 4 * p) * 2;
    const hf_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hf_s16x4*)a0);
    const hf_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hf_s16x4*)(a0 + 4 * rowbytes));
    const hf_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(hf_bf16x8, v);
}

__device__ __forceinline__ float hf_row_sum16(float v) {
    v += dpp_mov<DPP_XOR1>(v); v += dpp_mov<DPP_XOR2>(v); v += dpp_mov<DPP_HALF_MIRROR>(v); v += dpp_mov<DPP_MIRROR>(v);
    return v;
}

#ifndef HF_WAVES
#define HF_WAVES 8                 // waves per workgroup = 32-feature slices that share one staged dy tile (256 features)
#endif
#ifndef HF_OCC
#define HF_OCC 1
#endif
#ifndef HF_U
#define HF_U 4                     // 16-token groups per iteration (2 -> 4: twice the bytes in flight per CU, -5 %)
#endif
// DW (pass 2 only): the weight-gradient product of the NEXT backward step rides along.  dx is the gradient of the folded
// SegFormerHead's stride-4 map; its producer x1 G1^T (stage-1 tokens x1 [M][32]) needs dG1 = dx^T x1 and the column sums of dx --
// a [C x 32] product over all M tokens that, as its own launch, re-reads the 3.2 GB dx tensor (0.81 ms at cfg2, batch 128).
// Here each wave multiplies its 32 features of the finished dx tile (read back transposed from the staging tile `ot`: exactly the
// bf16 values that go to memory) with the tokens' x1 rows [x1 | 1 | 0] staged beside it: 2 x 3 accumulator tiles, 12 MFMAs per
// 64 tokens.  Per-workgroup partials [C][DW_LD] are summed in fixed order by colreduce_finalize.
// CW (pass 1 only): the CLASSIFIER's weight gradient rides along.  dW[class][feature] = sum_tokens dy[token][class] a[token][feature]
// with a = act(bn(x)) * drop is the product the operand-prologue GEMM (segf_gemm_pro, layout 2) formed in its own pass over x
// (1.03 ms at cfg2, batch 128): pass 1 already has the x tile and the dy tile on chip.  Each wave writes the normalised
// activations of its 32 features (bf16, the rounding the GEMM operand had) into its columns of a [TOK][256] LDS tile -- only the
// wave itself reads them back, so no workgroup barrier is added -- and multiplies dy^T (classes x tokens, transposed read of
// the staged dy tile) with it: 2 KS x 2 accumulator tiles per wave, 4 KS MFMAs per 32 tokens.  Per-workgroup partials
// [32 KS][C] are summed in fixed order by colreduce_finalize.
template <int PASS, int KS, bool DW = false, bool CW = false>
__global__ void __launch_bounds__(64 * HF_WAVES, HF_OCC) bn_cls_bwd_kernel(HeadFusedArgs a) {
    constexpr int NT = 2;                                   // 32 features per wave
    constexpr int TOK = 16 * HF_U;                          // tokens per iteration
    constexpr int RS = 32 * KS + 8;                         // LDS row stride in bf16 (+16 bytes: rows land on different banks)
    constexpr int NCH = TOK * 4 * KS;                       // 16-byte chunks of one dy tile
    constexpr int CPT = (NCH + 64 * HF_WAVES - 1) / (64 * HF_WAVES);
    // Every wave of the workgroup needs the same dy rows (its B operand): they are fetched ONCE per workgroup into LDS (double
    // buffered, one barrier per 32 tokens) instead of once per wave -- with per-wave global loads the dy rows crossed the
    // L2 -> CU path 24 times per pass and cost 1.5 of the 4.0 ms (measured by removing them).
    __shared__ __attribute__((aligned(16))) bf16_t tile[2][TOK][RS];
    // x (and dx in pass 2) also travel through LDS, cooperatively and in whole 512-byte row runs: a wave owns 32 features, so
    // its own accesses were 64-byte pieces of 16 rows per instruction; with contiguous accesses both passes measured 2.60 -> 1.78 ms
    // (upper bound, wrong layout), the staged form below is what that buys with the LDS round trip and the second barrier.
    constexpr int XW = 32 * HF_WAVES;                       // features per workgroup
    constexpr int XRS = XW + 8;                             // LDS row stride in bf16
    constexpr int XCH = TOK * (XW / 8);                     // 16-byte chunks of one x tile
    constexpr int XPT = XCH / (64 * HF_WAVES);              // per thread
    static_assert(XCH % (64 * HF_WAVES) == 0, "x tile chunks must divide evenly over the workgroup");
    __shared__ __attribute__((aligned(16))) bf16_t xt[2][TOK][XRS];
    __shared__ __attribute__((aligned(16))) bf16_t ot[PASS == 2 ? TOK : 1][XRS];
    static_assert(!DW || (PASS == 2 && HF_WAVES == 8 && TOK == 64), "the riding weight gradient belongs to pass 2");
    static_assert(!CW || (PASS == 1 && TOK % 32 == 0), "the classifier's weight gradient rides on pass 1");
    __shared__ __attribute__((aligned(16))) bf16_t at[CW ? TOK : 1][XRS];
    hf_f32x4 accc[CW ? 2 * KS : 1][CW ? 2 : 1];
#pragma unroll
    for (int i = 0; i < (CW ? 2 * KS : 1); ++i)
#pragma unroll
        for (int j = 0; j < (CW ? 2 : 1); ++j) accc[i][j] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int X1RS = DW_C1 + 16 + 8;                    // [x1 (32) | 1 | 0 x 15 | pad]: 112-byte rows
    __shared__ __attribute__((aligned(16))) bf16_t x1t[DW ? 2 : 1][DW ? TOK : 1][X1RS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mi = lane & 15, g = lane >> 4;
    // 1-D grid, XCD-aware logical id L = (token chunk, feature slice) with the slice fastest: the workgroups that cover the whole
    // row of the same tokens are dispatched together on one XCD (full 2C-byte runs of x / dx, dy rows shared in that L2)
    const unsigned L = xcd_block();
    const int ny = a.C / (32 * HF_WAVES);
    const int slice = (int)(L % (unsigned)ny), blk = (int)(L / (unsigned)ny);
    const int n0 = (slice * HF_WAVES + wave) * (16 * NT);
    // weight fragments (skinny-GEMM order): MFMA row i of tile nt carries feature n0 + 8 (i >> 2) + 4 nt + (i & 3), so the tile
    // pair gives lane group g the 8 consecutive features n0 + 8 g .. + 7
    hf_bf16x8 Wf[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + 8 * (mi >> 2) + 4 * nt + (mi & 3);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            hf_s16x8 wv;
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = (short)a.w[(int64_t)(32 * s + 8 * g + j) * a.ldw + n];
            Wf[nt][s] = __builtin_bit_cast(hf_bf16x8, wv);
        }
    }
    // per-feature constants of this lane's 8 features f = n0 + 8 g + j:
    //   z = x zA + zB (pre-activation), xhat = x hC + hD;  pass 2: dx = E gg - P - x Q
    const int f0 = n0 + 8 * g;
    float zA[8], zB[8], hC[8], hD[8], E[8], P[8], Q[8];
    const float invn = 1.f / (float)a.M;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float mu = a.mean[f0 + j], rs = a.rstd[f0 + j], ga = a.gamma[f0 + j], be = a.beta[f0 + j];
        zA[j] = ga * rs; zB[j] = be - mu * ga * rs;
        hC[j] = rs; hD[j] = -mu * rs;
        if (PASS == 2) {
            const float mg = a.eval_mode ? 0.f : a.sums[f0 + j] * invn, mgx = a.eval_mode ? 0.f : a.sums[a.C + f0 + j] * invn;
            E[j] = ga * rs;
            P[j] = ga * rs * (mg + hD[j] * mgx);
            Q[j] = ga * rs * hC[j] * mgx;
        }
    }
    const float zlo = a.act == 0 ? -INFINITY : 0.f, zhi = a.act == 2 ? 6.f : INFINITY;     // act(z) passes gradient iff zlo < z < zhi
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    // this workgroup's token groups: sample b, chunk c of its groups_per_sample 16-token groups
    const int b = blk / a.chunks_per_sample, chunk = blk % a.chunks_per_sample;
    const int gbeg = (int)((int64_t)a.groups_per_sample * chunk / a.chunks_per_sample);
    const int gend = (int)((int64_t)a.groups_per_sample * (chunk + 1) / a.chunks_per_sample);
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        cs[j] = a.cscale ? a.cscale[(int64_t)b * a.C + f0 + j] : 1.f;
        if (PASS == 2) E[j] *= cs[j];
        // CW (act = identity / ReLU only): the Dropout2d scale cs >= 0 commutes with the activation, so it is folded into the affine
        // map: z' = cs z, a = act(z'); the gradient mask z' > 0 is unchanged for cs > 0, and for cs = 0 the sums are multiplied by
        // cs = 0 at the end anyway.  (Frees eight registers in the loop: the riding product needs 80 accumulators.)
        if (CW) { zA[j] *= cs[j]; zB[j] *= cs[j]; }
    }
    const int64_t row0 = (int64_t)b * a.rps;
    const int64_t last_row = row0 + (int64_t)gend * 16 - 1;
    // cooperative dy tile fetch: chunk q of the tile = (token q / (4 KS), 16-byte column q % (4 KS)); at most two chunks per thread,
    // held in two NAMED registers (an indexed private array was demoted to LDS by the compiler and waited for at once)
    static_assert(CPT <= 3, "dy tile: at most three 16-byte chunks per thread");
    uint4 stg0 = make_uint4(0, 0, 0, 0), stg1 = make_uint4(0, 0, 0, 0), stg2 = make_uint4(0, 0, 0, 0);
    auto fetch_one = [&](int gp, int q) -> uint4 {
        // unconditional load from clamped chunk / row indices: a load inside a divergent `if` is followed by a full
        // s_waitcnt and the "prefetch" would wait out the HBM latency at the top of every iteration
        q = q < NCH ? q : NCH - 1;
        const int tk = q / (4 * KS), col = q % (4 * KS);
        int64_t m = row0 + (int64_t)gp * 16 + tk;
        m = m <= last_row ? m : last_row;                                       // odd tail: clamp (results of dead groups are dropped)
        return *reinterpret_cast<const uint4*>(a.dy + m * a.ldy + 8 * col);
    };
    auto fetch_tile = [&](int gp) {
        stg0 = fetch_one(gp, (int)threadIdx.x);
        if (CPT > 1) stg1 = fetch_one(gp, (int)threadIdx.x + 64 * HF_WAVES);
        if (CPT > 2) stg2 = fetch_one(gp, (int)threadIdx.x + 128 * HF_WAVES);
    };
    auto stash_one = [&](int bufi, int q, const uint4& v) {
        const int tk = q / (4 * KS), col = q % (4 * KS);
        if (q < NCH) *reinterpret_cast<uint4*>(&tile[bufi][tk][8 * col]) = v;
    };
    auto stash_tile = [&](int bufi) {
        stash_one(bufi, (int)threadIdx.x, stg0);
        if (CPT > 1) stash_one(bufi, (int)threadIdx.x + 64 * HF_WAVES, stg1);
        if (CPT > 2) stash_one(bufi, (int)threadIdx.x + 128 * HF_WAVES, stg2);
    };
    typedef uint32_t hf_u32x4 __attribute__((ext_vector_type(4)));
    hf_u32x4 xs[XPT];                                        // this thread's chunks of the NEXT x tile
    const int wg_f0 = slice * XW;                            // first feature of the workgroup
    auto fetch_x = [&](int gp) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int q = (int)threadIdx.x + 64 * HF_WAVES * i, tk = q / (XW / 8), c16 = q % (XW / 8);
            int64_t m = row0 + (int64_t)gp * 16 + tk;
            m = m <= last_row ? m : last_row;
            xs[i] = *reinterpret_cast<const hf_u32x4*>(a.x + m * a.C + wg_f0 + 8 * c16);
        }
    };
    auto stash_x = [&](int bufi) {
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int q = (int)threadIdx.x + 64 * HF_WAVES * i, tk = q / (XW / 8), c16 = q % (XW / 8);
            *reinterpret_cast<hf_u32x4*>(&xt[bufi][tk][8 * c16]) = xs[i];
        }
    };
    // x1 rows of the token tile: threads 0..255 carry one 16-byte chunk each (4 per token); threads 256..383 write the ones /
    // zero columns.  Rows of dead groups (odd tail of the chunk) are staged as zeros so that they drop out of the product.
    hf_u32x4 x1s = {0u, 0u, 0u, 0u};
    auto fetch_x1 = [&](int gp) {
        if (DW) {
            const int q = (int)threadIdx.x & 255, tk = q >> 2, c16 = q & 3;
            int64_t m = row0 + (int64_t)gp * 16 + tk;
            m = m <= last_row ? m : last_row;
            x1s = *reinterpret_cast<const hf_u32x4*>(a.x1 + m * a.ldx1 + 8 * c16);
        }
    };
    auto stash_x1 = [&](int bufi, int gp) {
        if (DW) {
            const int t = (int)threadIdx.x;
            if (t < 256) {
                const int tk = t >> 2, c16 = t & 3;
                const bool live = gp + (tk >> 4) < gend;
                const hf_u32x4 z = {0u, 0u, 0u, 0u};
                *reinterpret_cast<hf_u32x4*>(&x1t[bufi][tk][8 * c16]) = live ? x1s : z;
            } else if (t < 384) {
                const int r = t - 256, tk = r >> 1, half = r & 1;
                const bool live = gp + (tk >> 4) < gend;
                const hf_u32x4 v = {(half == 0 && live) ? 0x00003f80u : 0u, 0u, 0u, 0u};      // bf16 1.0 in column 32
                *reinterpret_cast<hf_u32x4*>(&x1t[bufi][tk][DW_C1 + 8 * half]) = v;
            }
        }
    };
    hf_f32x4 accw[DW ? 2 : 1][DW ? 3 : 1];
#pragma unroll
    for (int i = 0; i < (DW ? 2 : 1); ++i)
#pragma unroll
        for (int j = 0; j < (DW ? 3 : 1); ++j) accw[i][j] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
    if (gbeg < gend) { fetch_tile(gbeg); fetch_x(gbeg); fetch_x1(gbeg); stash_tile(0); stash_x(0); stash_x1(0, gbeg); }
    __syncthreads();
    int bufi = 0;
    for (int gp = gbeg; gp < gend; gp += HF_U) {
        const bool more = gp + HF_U < gend;
        if (more) { fetch_tile(gp + HF_U); fetch_x(gp + HF_U); fetch_x1(gp + HF_U); }
        constexpr int UNR = CW ? 1 : HF_U;                 // CW: 80 accumulator registers ride along -- one token group at a time
#pragma unroll UNR
        for (int u = 0; u < HF_U; ++u) {
            hf_f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = hf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const hf_bf16x8 yv = *reinterpret_cast<const hf_bf16x8*>(&tile[bufi][16 * u + mi][32 * s + 8 * g]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf[nt][s], yv, acc[nt], 0, 0, 0);
            }
            const bool live = gp + u < gend;
            const hf_u32x4 xv4 = *reinterpret_cast<const hf_u32x4*>(&xt[bufi][16 * u + mi][32 * wave + 8 * g]);
            const uint32_t xw[4] = {xv4[0], xv4[1], xv4[2], xv4[3]};
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float da = j < 4 ? acc[0][j] : acc[1][j - 4];                   // d loss / d a[token][f0 + j]
                const float xv = (j & 1) ? __uint_as_float(xw[j >> 1] & 0xffff0000u) : __uint_as_float(xw[j >> 1] << 16);
                const float z = fmaf(xv, zA[j], zB[j]);
                const bool on = live & (z > zlo) & (z < zhi);                          // branch-free: act as an open interval
                const float gm = on ? da : 0.f;                                        // the Dropout2d scale is per (sample, feature):
                if (PASS == 1) {                                                       // applied to the sums / folded into E below
                    s1[j] += gm;
                    s2[j] = fmaf(gm, xv, s2[j]);                                       // sum g x; xhat = x hC + hD is applied to the sums
                    if (CW) o[j] = live ? fmaxf(z, zlo) : 0.f;                         // a = act(bn(x)) * drop (cs folded into z; dead rows: zero)
                } else {
                    o[j] = fmaf(E[j], gm, -fmaf(xv, Q[j], P[j]));
                }
            }
            if (CW) {
                const hf_u32x4 av = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
                *reinterpret_cast<hf_u32x4*>(&at[16 * u + mi][32 * wave + 8 * g]) = av;
            }
            if (PASS == 2) {
                const hf_u32x4 ov = {pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
                *reinterpret_cast<hf_u32x4*>(&ot[16 * u + mi][32 * wave + 8 * g]) = ov;
            }
        }
        if (CW) {       // dWcls partial += dy_tile^T a_tile: tokens are the contraction index of both operands; this wave's columns only
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < TOK / 32; ++ks) {
                const unsigned char* ab = reinterpret_cast<const unsigned char*>(&at[32 * ks][0]);
                const unsigned char* yb = reinterpret_cast<const unsigned char*>(&tile[bufi][32 * ks][0]);
                const hf_bf16x8 fb0 = hf_frag_tok_tr(ab, XRS * 2, 32 * wave, lane), fb1 = hf_frag_tok_tr(ab, XRS * 2, 32 * wave + 16, lane);
#pragma unroll
                for (int ct = 0; ct < 2 * KS; ++ct) {
                    const hf_bf16x8 fa = hf_frag_tok_tr(yb, RS * 2, 16 * ct, lane);
                    accc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb0, accc[ct][0], 0, 0, 0);
                    accc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb1, accc[ct][1], 0, 0, 0);
                    if (ct & 1) __builtin_amdgcn_sched_barrier(0);      // at most two dy fragments live (the scheduler otherwise hoists all 2 KS reads: spills)
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // keep the LDS writes of the prefetched tile BELOW the arithmetic: hoisted above it (the compiler sees no dependence) they
        // wait for the global loads at the top of the iteration and expose the whole HBM latency
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        if (more) {
            stash_tile(bufi ^ 1);
            stash_x(bufi ^ 1);
            stash_x1(bufi ^ 1, gp + HF_U);
        }
        __syncthreads();                 // the next tiles are complete, and nobody still reads the buffers that get overwritten next
        if (DW) {                        // dG1 partial += dx_tile^T [x1 | 1]: tokens are the contraction index of both operands
#pragma unroll
            for (int ks = 0; ks < TOK / 32; ++ks) {
                const unsigned char* ob = reinterpret_cast<const unsigned char*>(&ot[32 * ks][0]);
                const unsigned char* xb = reinterpret_cast<const unsigned char*>(&x1t[bufi][32 * ks][0]);
                const hf_bf16x8 fa0 = hf_frag_tok_tr(ob, XRS * 2, 32 * wave, lane), fa1 = hf_frag_tok_tr(ob, XRS * 2, 32 * wave + 16, lane);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    const hf_bf16x8 fb = hf_frag_tok_tr(xb, X1RS * 2, 16 * nt, lane);
                    accw[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0, fb, accw[0][nt], 0, 0, 0);
                    accw[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1, fb, accw[1][nt], 0, 0, 0);
                }
            }
        }
        if (PASS == 2) {                 // the dx tile leaves in whole 512-byte row runs (rows of dead groups are dropped)
#pragma unroll
            for (int i = 0; i < XPT; ++i) {
                const int q = (int)threadIdx.x + 64 * HF_WAVES * i, tk = q / (XW / 8), c16 = q % (XW / 8);
                const hf_u32x4 ov = *reinterpret_cast<const hf_u32x4*>(&ot[tk][8 * c16]);
                if (gp + (tk >> 4) < gend)
                    *reinterpret_cast<hf_u32x4*>(a.dx + (row0 + (int64_t)gp * 16 + tk) * a.C + wg_f0 + 8 * c16) = ov;
            }
            __syncthreads();             // ot is free again
        }
        bufi ^= 1;
    }
    if (DW) {
        // accumulator (mt, nt): rows = features wg_f0 + 32 wave + 16 mt + 4 (lane >> 4) + r, column = 16 nt + (lane & 15)
        float* dst = a.dwpart + ((int64_t)blk * a.C + wg_f0 + 32 * wave) * DW_LD;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int cc = 16 * nt + mi;
                if (cc < DW_LD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(16 * mt + 4 * g + r) * DW_LD + cc] = accw[mt][nt][r];
                }
            }
    }
    if (CW) {
        // accumulator (ct, ft): rows = classes 16 ct + 4 (lane >> 4) + r, column = feature wg_f0 + 32 wave + 16 ft + (lane & 15)
        float* dst = a.cwpart + (int64_t)blk * (32 * KS) * a.C + wg_f0 + 32 * wave + mi;
#pragma unroll
        for (int ct = 0; ct < 2 * KS; ++ct)
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(int64_t)(16 * ct + 4 * g + r) * a.C + 16 * ft] = accc[ct][ft][r];
    }
    if (PASS == 1) {
        // tokens of a group sit in the 16 lanes of a row group: DPP row sums, then lane mi == 0 of each (wave, g) writes its 8 features
        float* dst = a.partial + (int64_t)blk * 2 * a.C + f0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float r1 = hf_row_sum16(s1[j]), r2 = hf_row_sum16(s2[j]);
            // sum g = cs sum gm;  sum g xhat = cs (hC sum gm x + hD sum gm)   (constants re-read here: not live across the loop)
            const float csj = a.cscale ? a.cscale[(int64_t)b * a.C + f0 + j] : 1.f;
            const float hCj = a.rstd[f0 + j], hDj = -a.mean[f0 + j] * hCj;
            if (mi == 0) { dst[j] = csj * r1; dst[a.C + j] = csj * fmaf(hCj, r2, hDj * r1); }
        }
    }
}

// out [2][C] = {dbeta, dgamma} -> separate arrays
__global__ void hf_split_kernel(const float* __restrict__ sums, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = sums[c];
    dgamma[c] = sums[C + c];
}

static int hf_chunks(int B, int groups_per_sample, int ny, bool cw = false) {
    // ~3000 workgroups in flight, at least 8 token groups per workgroup; with the classifier's weight gradient riding on pass 1
    // every workgroup leaves a [32 KS][256] fp32 partial: three rounds of workgroups (768) keep those at ~0.1 GB
    const int budget = cw ? 768 : 3072;
    int s = (budget / ny + B - 1) / B;
    if (s > groups_per_sample / 8) s = groups_per_sample / 8;
    return s < 1 ? 1 : s;
}

extern "C" int segf_bn_cls_bwd_supported(int dt, int64_t M, int C, int K, int64_t rows_per_sample) {
    if (dt != SEGF_BF16 || POL(no_head_fused)) return 0;
    if (K % 32 || K < 32 || K > 192 || C % (32 * HF_WAVES) || rows_per_sample <= 0 || rows_per_sample % 16 || M % rows_per_sample) return 0;
    if (M / rows_per_sample > 65535 || M < 16384) return 0;
    return 1;
}
extern "C" int64_t segf_bn_cls_bwd_ws(int64_t M, int C, int64_t rows_per_sample) {
    const int B = (int)(M / rows_per_sample);
    return (int64_t)B * hf_chunks(B, (int)(rows_per_sample / 16), C / (32 * HF_WAVES)) * 2 * C + 2 * C;
}

// everything that can ride: x1 / dG (pass 2, nullable pair) and dWcls fp32 [K][C] (pass 1, nullable)
extern "C" int64_t segf_bn_cls_bwd_full_ws(int64_t M, int C, int K, int64_t rows_per_sample) {
    // sized for the larger of the two workgroup budgets: with dwcls == NULL the launch uses the 3072-workgroup plan (up to 4x the chunks
    // of the 768-workgroup plan that carries the classifier's weight gradient), and x1 / dG may still ride on pass 2
    const int B = (int)(M / rows_per_sample);
    const int gps = (int)(rows_per_sample / 16), ny = C / (32 * HF_WAVES);
    const int64_t nb_cw = (int64_t)B * hf_chunks(B, gps, ny, true), nb_plain = (int64_t)B * hf_chunks(B, gps, ny, false);
    const int64_t nblk = nb_cw > nb_plain ? nb_cw : nb_plain;
    return nblk * 2 * C + 2 * C + nblk * C * DW_LD + nb_cw * (int64_t)K * C;
}

extern "C" int segf_bn_cls_bwd_dw_supported(int dt, int64_t M, int C, int K, int64_t rows_per_sample, int C1) {
    if (POL(no_head_fused_dw)) return 0;
    return segf_bn_cls_bwd_supported(dt, M, C, K, rows_per_sample) && C1 == DW_C1 && K <= 160;     // K = 192: the tiles pass 160 KB of LDS
}
extern "C" int64_t segf_bn_cls_bwd_dw_ws(int64_t M, int C, int64_t rows_per_sample) {
    const int B = (int)(M / rows_per_sample);
    return segf_bn_cls_bwd_ws(M, C, rows_per_sample) +
           (int64_t)B * hf_chunks(B, (int)(rows_per_sample / 16), C / (32 * HF_WAVES)) * C * DW_LD;
}

static int bn_cls_bwd_impl(int dt, int64_t M, int C, int K, const void* dy, int64_t ldy, const void* w, int64_t ldw, const void* x,
                           const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                           const float* chan_scale, int64_t rows_per_sample, int eval_mode, void* dx, float* dgamma,
                           float* dbeta, float* ws, const void* x1, int64_t ldx1, float* dG, void* stream, float* dwcls = nullptr);

extern "C" int segf_bn_cls_bwd_full(int dt, int64_t M, int C, int K, const void* dy, int64_t ldy, const void* w, int64_t ldw,
                                    const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                    const float* chan_scale, int64_t rows_per_sample, int eval_mode, void* dx, float* dgamma,
                                    float* dbeta, float* ws, const void* x1, int64_t ldx1, int C1, float* dG, float* dwcls,
                                    void* stream) {
    if (x1 && (!segf_bn_cls_bwd_dw_supported(dt, M, C, K, rows_per_sample, C1) || !dG || ldx1 < C1 || ldx1 % 8 || ((uintptr_t)x1 % 16)))
        return SEGF_ERR_SHAPE;
    if (dwcls && (K > 192 || act > 1)) return SEGF_ERR_SHAPE;      // the riding weight gradient rebuilds a = max(z, zlo): identity / ReLU only
    return bn_cls_bwd_impl(dt, M, C, K, dy, ldy, w, ldw, x, mean, rstd, gamma, beta, act, chan_scale, rows_per_sample, eval_mode, dx,
                           dgamma, dbeta, ws, x1, ldx1, x1 ? dG : nullptr, stream, dwcls);
}

extern "C" int segf_bn_cls_bwd(int dt, int64_t M, int C, int K, const void* dy, int64_t ldy, const void* w, int64_t ldw, const void* x,
                               const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                               const float* chan_scale, int64_t rows_per_sample, int eval_mode, void* dx, float* dgamma,
                               float* dbeta, float* ws, void* stream) {
    return bn_cls_bwd_impl(dt, M, C, K, dy, ldy, w, ldw, x, mean, rstd, gamma, beta, act, chan_scale, rows_per_sample, eval_mode, dx,
                           dgamma, dbeta, ws, nullptr, 0, nullptr, stream);
}
extern "C" int segf_bn_cls_bwd_dw(int dt, int64_t M, int C, int K, const void* dy, int64_t ldy, const void* w, int64_t ldw, const void* x,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                  const float* chan_scale, int64_t rows_per_sample, int eval_mode, void* dx, float* dgamma,
                                  float* dbeta, float* ws, const void* x1, int64_t ldx1, int C1, float* dG, void* stream) {
    if (!segf_bn_cls_bwd_dw_supported(dt, M, C, K, rows_per_sample, C1) || !x1 || !dG || ldx1 < C1 || ldx1 % 8 || ((uintptr_t)x1 % 16))
        return SEGF_ERR_SHAPE;
    return bn_cls_bwd_impl(dt, M, C, K, dy, ldy, w, ldw, x, mean, rstd, gamma, beta, act, chan_scale, rows_per_sample, eval_mode, dx,
                           dgamma, dbeta, ws, x1, ldx1, dG, stream);
}

static int bn_cls_bwd_impl(int dt, int64_t M, int C, int K, const void* dy, int64_t ldy, const void* w, int64_t ldw, const void* x,
                           const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                           const float* chan_scale, int64_t rows_per_sample, int eval_mode, void* dx, float* dgamma,
                           float* dbeta, float* ws, const void* x1, int64_t ldx1, float* dG, void* stream, float* dwcls) {
    if (!segf_bn_cls_bwd_supported(dt, M, C, K, rows_per_sample) || ldy < K || ldw < C || act < 0 || act > 2) return SEGF_ERR_SHAPE;
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) % 16 || (ldy % 8)) return SEGF_ERR_SHAPE;
    if (!ws) return SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int B = (int)(M / rows_per_sample), gps = (int)(rows_per_sample / 16), ny = C / (32 * HF_WAVES);
    const int chunks = hf_chunks(B, gps, ny, dwcls != nullptr);
    const int nblk = B * chunks;
    float* sums = ws + (int64_t)nblk * 2 * C;
    float* dwpart = sums + 2 * C;                       // (only with x1) [nblk][C][DW_LD]
    float* cwpart = dwpart + (int64_t)nblk * C * DW_LD; // (only with dwcls) [nblk][K][C]
    HeadFusedArgs a{(const bf16_t*)dy, ldy, (const bf16_t*)w, ldw, (const bf16_t*)x, mean, rstd, gamma, beta, chan_scale, sums,
                    (bf16_t*)dx, ws, M, C, rows_per_sample, gps, chunks, act, eval_mode, (const bf16_t*)x1, ldx1, dwpart, cwpart};
    const dim3 grid((unsigned)(nblk * ny));
#define HF_LAUNCH(PASS)                                                                                                  \
    do {                                                                                                                 \
        switch (K / 32) {                                                                                                \
        case 1: hipLaunchKernelGGL((bn_cls_bwd_kernel<PASS, 1>), grid, dim3(64 * HF_WAVES), 0, st, a); break;                      \
        case 2: hipLaunchKernelGGL((bn_cls_bwd_kernel<PASS, 2>), grid, dim3(64 * HF_WAVES), 0, st, a); break;                      \
        case 3: hipLaunchKernelGGL((bn_cls_bwd_kernel<PASS, 3>), grid, dim3(64 * HF_WAVES), 0, st, a); break;                      \
        case 4: hipLaunchKernelGGL((bn_cls_bwd_kernel<PASS, 4>), grid, dim3(64 * HF_WAVES), 0, st, a); break;                      \
        case 5: hipLaunchKernelGGL((bn_cls_bwd_kernel<PASS, 5>), grid, dim3(64 * HF_WAVES), 0, st, a); break;                      \
        default: hipLaunchKernelGGL((bn_cls_bwd_kernel<PASS, 6>), grid, dim3(64 * HF_WAVES), 0, st, a); break;                     \
        }                                                                                                                \
    } while (0)
    if (dwcls) {
        switch (K / 32) {
        case 1: hipLaunchKernelGGL((bn_cls_bwd_kernel<1, 1, false, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        case 2: hipLaunchKernelGGL((bn_cls_bwd_kernel<1, 2, false, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        case 3: hipLaunchKernelGGL((bn_cls_bwd_kernel<1, 3, false, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        case 4: hipLaunchKernelGGL((bn_cls_bwd_kernel<1, 4, false, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        case 5: hipLaunchKernelGGL((bn_cls_bwd_kernel<1, 5, false, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        default: hipLaunchKernelGGL((bn_cls_bwd_kernel<1, 6, false, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        }
        SEGF_CHECK_LAUNCH();
        colreduce_finalize_launch(cwpart, nblk, (int64_t)K * C, dwcls, st);
    } else {
        HF_LAUNCH(1);
    }
    SEGF_CHECK_LAUNCH();
    colreduce_finalize_launch(ws, nblk, 2 * (int64_t)C, sums, st);
    SEGF_CHECK_LAUNCH();
    if (x1) {
        switch (K / 32) {
        case 1: hipLaunchKernelGGL((bn_cls_bwd_kernel<2, 1, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        case 2: hipLaunchKernelGGL((bn_cls_bwd_kernel<2, 2, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        case 3: hipLaunchKernelGGL((bn_cls_bwd_kernel<2, 3, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        case 4: hipLaunchKernelGGL((bn_cls_bwd_kernel<2, 4, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        default: hipLaunchKernelGGL((bn_cls_bwd_kernel<2, 5, true>), grid, dim3(64 * HF_WAVES), 0, st, a); break;
        }
        SEGF_CHECK_LAUNCH();
        colreduce_finalize_launch(dwpart, nblk, (int64_t)C * DW_LD, dG, st);
    } else {
        HF_LAUNCH(2);
    }
    SEGF_CHECK_LAUNCH();
#undef HF_LAUNCH
    hipLaunchKernelGGL(hf_split_kernel, dim3((C + 255) / 256), dim3(256), 0, st, sums, C, dgamma, dbeta);
    SEGF_CHECK_LAUNCH();
    return 0;
}
