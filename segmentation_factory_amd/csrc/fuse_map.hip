// The folded SegFormerHead's stride-4 map in ONE pass, on the matrix pipe.
//
// Reference: heads/segformer.py:42-56  (Linear_i -> bilinear resize to stride 4 -> concat [c4,c3,c2,c1] -> 1x1 fuse conv).
// With the fold of functional.SegformerFoldedFuseFn the map is
//     fused[b, Y, X, :] = x1[b, Y, X, :] G1^T  +  sum_{i=2,3,4} bilinear_i( t_i )[b, Y, X, :],      t_i = x_i G_i^T + beta_i
// (the stage-1 bias rides in t_2: bilinear weights sum to one).  Until round 2 this took two launches that wrote / read a
// [B*H*W, C] tensor twice (gemm_skinny_rows: x1 G1^T -> HBM; upsample_add_248: read it back, add the three interpolated maps on
// the VALU, write `fused`): 0.63 + 1.78 ms at cfg2 / batch 128.  Here a workgroup owns an 8 x 8 block of stride-4 pixels and a
// 128-channel slice, and BOTH terms are matrix products accumulated in the same MFMA accumulators:
//   * x1 G1^T: K = C1 (32 or 64), the x1 rows are the B operand as they lie in memory (one 16-byte load per lane and K step);
//   * the three bilinear interpolations TOGETHER are one product  W_int [64 pixels x 64 sources] . T_src [64 sources x channels]:
//     the 8 x 8 block draws on 6 x 6 pixels of the 1/2 map, 4 x 4 of the 1/4 map and 3 x 3 of the 1/8 map (36 + 16 + 9 = 61 <= 64
//     source rows, K = 64 = two MFMA steps).  W_int depends only on the position inside the block -- border blocks use the same
//     weights on CLAMPED source addresses, which is exactly ATen's index clamping (area_pixel_compute_source_index) -- so it is
//     a per-wave constant in registers.  Every weight is (a/16)(b/16), a, b <= 15: exact in bf16; products are exact in fp32.
//     The source rows are staged in LDS as they lie in memory ([source][channel]) and read as MFMA operands through
//     ds_read_b64_tr_b16 (the contraction index is the ROW index of that image).
// The [16 pixels x 128 channels] result of a wave leaves through its own LDS slab as 256-byte row runs; the per-channel sum and
// sum of squares (BatchNorm statistics of the ConvModule that follows, heads/segformer.py:21-29) are accumulated per lane from
// the fp32 accumulators and reduced once per workgroup (deterministic partials + colreduce_finalize).
// HBM traffic: fused written once (3.2 GB at cfg2 / batch 128) + x1 and the three maps read (0.13 + 1.05 GB).
#include <stdlib.h>
#include "colreduce.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define FM_SLICE 128                 // channels per workgroup
#define FM_SROW 272                  // staged output row: 128 bf16 + 16 bytes of padding
#define FM_NSRC 61                   // live source rows: 6x6 (1/2) + 4x4 (1/4) + 3x3 (1/8)
#ifndef FM_OCC
#define FM_OCC 2                     // waves per SIMD the register budget is set for (3: 168 VGPRs, spills 5 dwords with the statistics)
#endif

struct FuseMapArgs {
    const bf16_t* x1; int64_t ldx1;
    const bf16_t* g1; int64_t ldg;                 // [C][C1]
    const bf16_t *src0, *src1, *src2; int64_t ld0, ld1, ld2;   // 1/2, 1/4, 1/8 maps, token-major [B*h*w][C]
    bf16_t* out; int64_t ldo;
    float* partial;                                // [streams][2][C] or nullptr
    int B, H, W, C;
    int tiles_x, tiles_per_img; int64_t ntiles;
    FastDivU32 div_tx, div_tpi;
    int nslice, nstream;
};

// same image as gemm.hip's reduction-major operand tile: [64 k rows][128 channels] bf16, 256-byte rows, 16-byte chunk c of row k
// stored at chunk c ^ swz(k): conflict-free for the 16-byte row writes and for the transposed fragment reads
__device__ __forceinline__ int fm_swz(int krow) { return 2 * ((krow & 3) + 4 * ((krow >> 3) & 1)); }
__device__ __forceinline__ bf16x8 fm_frag_tr(const unsigned char* tile, int cb, int s, int lane) {
    // lane (i = lane & 15, g = lane >> 4) receives T[k = 32 s + 8 g + j][channel cb + i], j = 0..7
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int u = (cb >> 2) + p;
    const int chunk = u >> 1, half = u & 1;
    s16x4 lo, hi;
    {
        const int krow = 32 * s + 8 * g + q;
        const unsigned char* a = tile + krow * 256 + ((chunk ^ fm_swz(krow)) << 4) + half * 8;
        lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    }
    {
        const int krow = 32 * s + 8 * g + 4 + q;
        const unsigned char* a = tile + krow * 256 + ((chunk ^ fm_swz(krow)) << 4) + half * 8;
        hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    }
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// source row r (0..63) of a block: level (0 = 1/2, 1 = 1/4, 2 = 1/8), offset (dy, dx) from the block's first source pixel
__device__ __forceinline__ void fm_source_of(int r, int& lvl, int& dy, int& dx) {
    if (r > FM_NSRC - 1) r = FM_NSRC - 1;          // rows 61..63 carry weight 0: any finite data
    if (r < 36) { lvl = 0; dy = r / 6; dx = r - 6 * dy; }
    else if (r < 52) { r -= 36; lvl = 1; dy = r >> 2; dx = r & 3; }
    else { r -= 52; lvl = 2; dy = r / 3; dx = r - 3 * dy; }
}
// bilinear weight (align_corners = False) of local source index r for local output pixel p (0..7) at ratio 2 / 4 / 8; the first
// source pixel of the block is (block origin / ratio) - 1, hence the + 1
__device__ __forceinline__ float fm_w1(int lvl, int p, int r) {
    const float ratio = lvl == 0 ? 0.5f : (lvl == 1 ? 0.25f : 0.125f);
    const float s = (p + 0.5f) * ratio - 0.5f + 1.f;
    const int i0 = (int)s;
    const float f = s - (float)i0;
    return r == i0 ? 1.f - f : (r == i0 + 1 ? f : 0.f);
}
__device__ __forceinline__ float fm_weight(int k, int py, int px) {
    if (k >= FM_NSRC) return 0.f;
    int lvl, dy, dx;
    fm_source_of(k, lvl, dy, dx);
    return fm_w1(lvl, py, dy) * fm_w1(lvl, px, dx);
}

template <int KS1, bool STATS>
__global__ void __launch_bounds__(256, FM_OCC) fuse_map_kernel(FuseMapArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char tsrc[64 * 256];
    __shared__ __attribute__((aligned(16))) unsigned char ostage[4][16 * FM_SROW];
    __shared__ __attribute__((aligned(16))) unsigned char g1lds[8 * KS1 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mi = lane & 15, g = lane >> 4;
    const unsigned L = xcd_block();
    const int slice = (int)(L % (unsigned)a.nslice);
    const int64_t stream = L / (unsigned)a.nslice;
    const int c0 = slice * FM_SLICE;

    // ---- per-workgroup constants -------------------------------------------------------------------------------------------
    // G1 slice in fragment order: fragment (nt, s), lane (mi, g) = G1[c0 + 16 nt + mi][32 s + 8 g .. + 7]
    for (int f = wave; f < 8 * KS1; f += 4) {
        const int s = f % KS1, nt = f / KS1;
        *reinterpret_cast<uint4*>(g1lds + f * 1024 + lane * 16) =
            *reinterpret_cast<const uint4*>(a.g1 + (int64_t)(c0 + 16 * nt + mi) * a.ldg + 32 * s + 8 * g);
    }
    // interpolation weights of this wave's 16 pixels (rows 2 wave, 2 wave + 1 of the block): B operand, lane (pixel mi, k group g)
    bf16x8 wint[2];
    {
        const int py = 2 * wave + (mi >> 3), px = mi & 7;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            s16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (short)f2bf(fm_weight(32 * s + 8 * g + j, py, px));
            wint[s] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // the four source rows this thread stages per block: rows r = tid / 16 + 16 i, 16-byte chunk ck = tid % 16 of the slice
    const int ck = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    int s_lvl[4], s_dy[4], s_dx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fm_source_of(r0 + 16 * i, s_lvl[i], s_dy[i], s_dx[i]);

    auto tile_coords = [&](int64_t t, int& b, int& ty, int& tx) {
        b = (int)fastdiv((uint32_t)t, a.div_tpi);
        const int rem = (int)t - b * a.tiles_per_img;
        ty = (int)fastdiv((uint32_t)rem, a.div_tx);
        tx = rem - ty * a.tiles_x;
    };
    auto load_sources = [&](int64_t t, u32x4 (&reg)[4]) {
        int b, ty, tx;
        tile_coords(t, b, ty, tx);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lv = s_lvl[i], sh = 2 - lv;                     // block origin in source pixels: (8 ty) >> (lv + 1) = ty << (2 - lv)
            int y = (ty << sh) - 1 + s_dy[i], x = (tx << sh) - 1 + s_dx[i];
            const int h = a.H >> (lv + 1), w = a.W >> (lv + 1);
            y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
            // (selects, not indexed kernel-argument arrays: those would be copied to scratch)
            const bf16_t* sp = lv == 0 ? a.src0 : (lv == 1 ? a.src1 : a.src2);
            const int64_t sl = lv == 0 ? a.ld0 : (lv == 1 ? a.ld1 : a.ld2);
            const bf16_t* p = sp + (((int64_t)b * h + y) * w + x) * sl + c0 + 8 * ck;
            reg[i] = *reinterpret_cast<const u32x4*>(p);
        }
    };
    auto pixel_of = [&](int64_t t, int m) -> int64_t {                // flat pixel index of this wave's pixel m (0..15) of block t
        int b, ty, tx;
        tile_coords(t, b, ty, tx);
        return ((int64_t)b * a.H + 8 * ty + 2 * wave + (m >> 3)) * a.W + 8 * tx + (m & 7);
    };
    auto load_x1 = [&](int64_t t, u32x4 (&reg)[KS1]) {
        const bf16_t* p = a.x1 + pixel_of(t, mi) * a.ldx1 + 8 * g;
#pragma unroll
        for (int s = 0; s < KS1; ++s) reg[s] = *reinterpret_cast<const u32x4*>(p + 32 * s);
    };

    f32x2_t p1[8][2], p2[8][2];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int h = 0; h < 2; ++h) { p1[nt][h] = f32x2_t{0.f, 0.f}; p2[nt][h] = f32x2_t{0.f, 0.f}; }

    u32x4 sreg[4], xa[KS1], xb[KS1];
    // Block order: stream-strided (concurrently running workgroups cover a contiguous band of the image).  FM_CHUNKED = 1 lets a
    // stream walk a contiguous range in raster order instead (the block below comes up w/8 iterations later in the same workgroup,
    // same XCD): measured no better on the MI355X (1.143 vs 1.116 ms stand-alone at cfg2 / batch 128; FETCH_SIZE reads 2.5 GB
    // against 1.2 GB algorithmic either way -- the halo re-fetches are Infinity-Cache hits, the launch is bound by its 3.2 GB of writes).
#ifndef FM_CHUNKED
#define FM_CHUNKED 0
#endif
    const int64_t chunk = (a.ntiles + a.nstream - 1) / a.nstream;
    int64_t t = FM_CHUNKED ? stream * chunk : stream;
    const int64_t tstep = FM_CHUNKED ? 1 : a.nstream;
    const int64_t tend = FM_CHUNKED ? (t + chunk < a.ntiles ? t + chunk : a.ntiles) : a.ntiles;
    if (t < tend) { load_sources(t, sreg); load_x1(t, xa); }
    unsigned char* ost = ostage[wave];
    for (; t < tend; t += tstep) {
        __syncthreads();                                              // every wave is done with the previous block's source image
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int krow = r0 + 16 * i;
            *reinterpret_cast<u32x4*>(tsrc + krow * 256 + ((ck ^ fm_swz(krow)) << 4)) = sreg[i];
        }
        __syncthreads();
        {   // next block's operands on their way while this one is multiplied (the last iteration re-reads its own)
            const int64_t tn = t + tstep < tend ? t + tstep : t;
            load_sources(tn, sreg);
            load_x1(tn, xb);
            SEGF_LOADS_ISSUED();
        }
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const bf16x8 wf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(g1lds + (nt * KS1 + s) * 1024 + lane * 16));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, xa[s]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm_frag_tr(tsrc, 16 * nt, s, lane), wint[s], acc, 0, 0, 0);
            // acc[r] = fused[pixel mi][channel c0 + 16 nt + 4 g + r]
            if (STATS) {
                const f32x2_t lo = {acc[0], acc[1]}, hi = {acc[2], acc[3]};
                p1[nt][0] += lo; p1[nt][1] += hi;
                p2[nt][0] += lo * lo; p2[nt][1] += hi * hi;
            }
            *reinterpret_cast<uint2*>(ost + mi * FM_SROW + (16 * nt + 4 * g) * 2) =
                make_uint2(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {   // 16 rows x 16 chunks of 16 bytes: lane -> (row, chunk), 256-byte runs per pixel
            const int64_t pix_lo = pixel_of(t, 0), pix_hi = pixel_of(t, 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i, row = idx >> 4, cc = idx & 15;
                const uint4 o = *reinterpret_cast<const uint4*>(ost + row * FM_SROW + 16 * cc);
                const int64_t pix = (row < 8 ? pix_lo : pix_hi) + (row & 7);
                *reinterpret_cast<uint4*>(a.out + pix * a.ldo + c0 + 8 * cc) = o;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            xa[s] = xb[s];
            asm volatile("" : "+v"(xa[s]));
        }
    }
    if (STATS) {
        // per-channel sums: over the wave's 16 pixel lanes (DPP row reduction), then over the four waves in fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(tsrc);                 // [wave][2][128]
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v1 = wave_sum(r < 2 ? p1[nt][0][r] : p1[nt][1][r - 2], 16);
                const float v2 = wave_sum(r < 2 ? p2[nt][0][r] : p2[nt][1][r - 2], 16);
                if (mi == 0) {
                    red[(wave * 2 + 0) * 128 + 16 * nt + 4 * g + r] = v1;
                    red[(wave * 2 + 1) * 128 + 16 * nt + 4 * g + r] = v2;
                }
            }
        __syncthreads();
        if (threadIdx.x < 256) {
            const int o = threadIdx.x >> 7, c = threadIdx.x & 127;
            const float v = ((red[(0 * 2 + o) * 128 + c] + red[(1 * 2 + o) * 128 + c]) + red[(2 * 2 + o) * 128 + c]) + red[(3 * 2 + o) * 128 + c];
            a.partial[(stream * 2 + o) * a.C + c0 + c] = v;
        }
    }
}

static int fuse_map_streams(int B, int H, int W, int C) {
    const int64_t ntiles = (int64_t)B * (H / 8) * (W / 8);
    const int nslice = C / FM_SLICE;
    int64_t streams = (256 * FM_OCC) / nslice;                       // FM_OCC workgroups per CU resident
    if (streams < 1) streams = 1;
    if (streams > ntiles) streams = ntiles;
    return (int)streams;
}

extern "C" int segf_fuse_map_248_supported(int dt, int B, int H, int W, int C, int C1) {
    return dt == SEGF_BF16 && B > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0 && C % FM_SLICE == 0 && (C1 == 32 || C1 == 64) &&
           (int64_t)B * (H / 8) * (W / 8) < (1ll << 31) && !POL(no_fuse_map);
}
extern "C" int64_t segf_fuse_map_248_ws(int B, int H, int W, int C) {
    return (int64_t)fuse_map_streams(B, H, W, C) * 2 * C;
}
extern "C" int segf_fuse_map_248(int B, int H, int W, int C, int C1, const void* x1, int64_t ldx1, const void* g1, int64_t ldg,
                                 const void* t2, int64_t ld2, const void* t3, int64_t ld3, const void* t4, int64_t ld4,
                                 void* out, int64_t ldo, float* sums, float* ws, void* stream) {
    if (!segf_fuse_map_248_supported(SEGF_BF16, B, H, W, C, C1)) return SEGF_ERR_SHAPE;
    const void* ptrs[6] = {x1, g1, t2, t3, t4, out};
    const int64_t lds[6] = {ldx1, ldg, ld2, ld3, ld4, ldo};
    const int64_t mins[6] = {C1, C1, C, C, C, C};
    for (int i = 0; i < 6; ++i)
        if (!ptrs[i] || ((uintptr_t)ptrs[i] % 16) || ((lds[i] * 2) % 16) || lds[i] < mins[i]) return SEGF_ERR_SHAPE;
    if (sums && !ws) return SEGF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    FuseMapArgs a;
    a.x1 = (const bf16_t*)x1; a.ldx1 = ldx1; a.g1 = (const bf16_t*)g1; a.ldg = ldg;
    a.src0 = (const bf16_t*)t2; a.src1 = (const bf16_t*)t3; a.src2 = (const bf16_t*)t4;
    a.ld0 = ld2; a.ld1 = ld3; a.ld2 = ld4;
    a.out = (bf16_t*)out; a.ldo = ldo; a.partial = sums ? ws : nullptr;
    a.B = B; a.H = H; a.W = W; a.C = C;
    a.tiles_x = W / 8; a.tiles_per_img = (H / 8) * (W / 8); a.ntiles = (int64_t)B * a.tiles_per_img;
    a.div_tx = fastdiv_make((uint32_t)a.tiles_x); a.div_tpi = fastdiv_make((uint32_t)a.tiles_per_img);
    a.nslice = C / FM_SLICE; a.nstream = fuse_map_streams(B, H, W, C);
    const dim3 grid((unsigned)(a.nstream * a.nslice));
    if (C1 == 32) {
        if (sums) hipLaunchKernelGGL((fuse_map_kernel<1, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((fuse_map_kernel<1, false>), grid, dim3(256), 0, st, a);
    } else {
        if (sums) hipLaunchKernelGGL((fuse_map_kernel<2, true>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((fuse_map_kernel<2, false>), grid, dim3(256), 0, st, a);
    }
    SEGF_CHECK_LAUNCH();
    if (sums) {
        colreduce_finalize_launch(ws, a.nstream, 2 * (int64_t)C, sums, st);
        SEGF_CHECK_LAUNCH();
    }
    return 0;
}

// =====================================================================================================================================
// The transposed map: the three transposed bilinear resizes of the folded head's backward (gradient of the stride-4 map -> gradients
// of the 1/2, 1/4, 1/8 maps; heads/segformer.py:44-50 backward) on the matrix pipe.
//   dT_i[src][c] = sum_pix W_i[pix][src] dy[pix][c]
// A workgroup owns a run of 8 x 8 pixel blocks along x (one block row, one 128-channel slice).  A block OWNS the sources whose
// support it centres: 4 x 4 pixels of the 1/2 map, 2 x 2 of the 1/4 map, 1 of the 1/8 map (21 sources); every gradient pixel that
// touches them lies in the 16 x 16 window around the block, so  dT^T [channels x sources] = dy^T [channels x 256 window pixels] .
// W [256 x 21]  is one MFMA product with K = 256 (8 steps of two window rows; the 1/2-map tile skips the two outer steps).
// W is separable, W[(wy, wx)][src] = wy[src_y][wy] * wx[src_x][wx], so a lane builds its B-operand fragments as the outer product of
// two 8-vectors (exact in bf16: every factor is a multiple of 1/16).  Image borders: ATen clamps the source index, i.e. an edge
// source also receives the weight of the virtual source beyond it, and window pixels outside the image do not exist -- both are
// properties of the 1-D factors (fmb_w1d), so border blocks run the same code with their own factors (rebuilt for the first,
// second and last block of a run; the window loads of outside pixels read clamped addresses and meet weight 0).
// The window lives in LDS as it lies in memory ([pixel][channel]) and is read through ds_read_b64_tr_b16; consecutive blocks of a
// run share half of their window, so only the 8 new columns are staged per block (the two column halves swap roles: lane group g
// reads the physical half g ^ phase).  Every gradient element passes the vector memory path twice (the VALU kernel it replaces,
// bilinear_bwd_248_kernel: four times, at 2.26 TB/s of algorithmic bytes = 0.28 of the HBM roofline).
struct FuseMapBwdArgs {
    const bf16_t* dy; int64_t ldo;
    bf16_t *d2, *d4, *d8;
    int B, H, W, C;
    int nslice;
};

__device__ __forceinline__ float fmb_tri(float d) { d = fabsf(d); return d < 1.f ? 1.f - d : 0.f; }
// 1-D factor: weight of window position wpos (0..15; pixel 8 t - 4 + wpos of nb * 8 pixels) for the owned source `own` (0 .. 8 / R - 1)
// of block t at ratio R = 2 << lvl, with ATen's index clamping folded in and pixels outside the image weighted 0
__device__ __forceinline__ float fmb_w1d(int lvl, int own, int wpos, int t, int nb) {
    const int P = 8 * t - 4 + wpos;
    if (P < 0 || P >= 8 * nb) return 0.f;
    const float R = (float)(2 << lvl);
    const int per = 4 >> lvl;                               // owned sources per block and axis
    const int S = per * t + own, smax = per * nb - 1;
    const float rel = ((float)P + 0.5f) / R - 0.5f;
    float w = fmb_tri(rel - (float)S);
    if (S == 0) w += fmb_tri(rel + 1.f);
    if (S == smax) w += fmb_tri(rel - (float)(smax + 1));
    return w;
}

__global__ void __launch_bounds__(256, 2) fuse_map_bwd_kernel(FuseMapBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char win[4 * 64 * 256];          // 4 tiles of [64 k rows][128 channels]
    __shared__ __attribute__((aligned(16))) unsigned char ost[2][16 * FM_SROW];       // [source tile][source][channel]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mi = lane & 15, g = lane >> 4;
    const unsigned L = xcd_block();
    const int slice = (int)(L % (unsigned)a.nslice);
    const unsigned run = L / (unsigned)a.nslice;
    const int h8 = a.H >> 3, w8 = a.W >> 3;
    const int ty = (int)(run % (unsigned)h8), b = (int)(run / (unsigned)h8);
    const int c0 = slice * FM_SLICE;
    const int h2 = a.H >> 1, w2 = a.W >> 1, h4 = a.H >> 2, w4 = a.W >> 2;

    // this lane's owned source in each source tile: tile 0 = the 4 x 4 sources of the 1/2 map; tile 1 = 2 x 2 of the 1/4 map
    // (n = 0..3), the 1/8 source (n = 4), nothing (n >= 5)
    const int l1 = mi < 4 ? 1 : 2, oy1 = mi < 4 ? (mi >> 1) : 0, ox1 = mi < 4 ? (mi & 1) : 0;
    const bool live1 = mi < 5;
    // y factors of the run: K step s, this lane's row 2 s + (g >> 1)
    float wy0[6], wy1[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int wy = 2 * s + (g >> 1);
        if (s >= 1 && s <= 6) wy0[s - 1] = fmb_w1d(0, mi >> 2, wy, ty, h8);
        wy1[s] = live1 ? fmb_w1d(l1, oy1, wy, ty, h8) : 0.f;
    }
    // B operand, lane (source n = mi, k group g), K step s: window pixels k = 32 s + 8 g + j = (row 2 s + (g >> 1), column 8 (g & 1) + j)
    bf16x8 wb0[6], wb1[8];
    auto build_weights = [&](int tx) {
        float wx0[8], wx1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int wx = 8 * (g & 1) + j;
            wx0[j] = fmb_w1d(0, mi & 3, wx, tx, w8);
            wx1[j] = live1 ? fmb_w1d(l1, ox1, wx, tx, w8) : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s >= 1 && s <= 6) {
                union { uint32_t u[4]; bf16x8 v; } f;
#pragma unroll
                for (int j = 0; j < 4; ++j) f.u[j] = pack2bf(wy0[s - 1] * wx0[2 * j], wy0[s - 1] * wx0[2 * j + 1]);
                wb0[s - 1] = f.v;
            }
            union { uint32_t u[4]; bf16x8 v; } f1;
#pragma unroll
            for (int j = 0; j < 4; ++j) f1.u[j] = pack2bf(wy1[s] * wx1[2 * j], wy1[s] * wx1[2 * j + 1]);
            wb1[s] = f1.v;
        }
    };
    // staging: 8 columns x 16 rows of 256-byte pixel rows = 2048 chunks of 16 bytes, 8 per thread: pixel p = tid / 16 + 16 i, chunk tid % 16
    const int ck = threadIdx.x & 15, p0 = threadIdx.x >> 4;
    const bf16_t* imgbase = a.dy + (int64_t)b * a.H * a.W * a.ldo + c0 + 8 * ck;
    int yoff[8];                                                   // clamped window rows of this thread's eight pixels (x W)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int y = 8 * ty - 4 + ((p0 + 16 * i) >> 3);
        y = y < 0 ? 0 : (y > a.H - 1 ? a.H - 1 : y);
        yoff[i] = y * a.W;
    }
    auto load_half = [&](int col0, u32x4 (&reg)[8]) {                  // window columns col0 .. col0 + 7 (global x, clamped), all 16 rows
        int x = col0 + (p0 & 7);
        x = x < 0 ? 0 : (x > a.W - 1 ? a.W - 1 : x);
#pragma unroll
        for (int i = 0; i < 8; ++i) reg[i] = *reinterpret_cast<const u32x4*>(imgbase + (int64_t)(yoff[i] + x) * a.ldo);
    };
    auto store_half = [&](int phys, const u32x4 (&reg)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = p0 + 16 * i, wy = p >> 3, wxh = p & 7;
            const int krow = 32 * ((wy & 3) >> 1) + 16 * (wy & 1) + 8 * phys + wxh;
            *reinterpret_cast<u32x4*>(win + (wy >> 2) * 16384 + krow * 256 + ((ck ^ fm_swz(krow)) << 4)) = reg[i];
        }
    };
    u32x4 sreg[8];
    load_half(-4, sreg);
    store_half(0, sreg);
    load_half(4, sreg);
    store_half(1, sreg);
    __syncthreads();
    for (int tx = 0; tx < w8; ++tx) {
        const int phase = tx & 1;
        {   // the next block's new columns (the last block re-reads its own: unconditional loads)
            const int txn = tx + 1 < w8 ? tx + 1 : tx;
            load_half(8 * txn + 4, sreg);
            SEGF_LOADS_ISSUED();
        }
        if (tx <= 1 || tx == w8 - 1) build_weights(tx);    // x factors change at the two image borders only (wave-uniform)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int cb = 16 * (2 * wave + ct);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                // A operand: dy^T, lane (channel cb + mi, k group g) -> physical k group g ^ phase (the column halves alternate)
                const unsigned char* tile = win + (s >> 1) * 16384;
                const int gp = g ^ phase, q = mi >> 2, pp = mi & 3;
                const int u = (cb >> 2) + pp, chunk = u >> 1, half = u & 1;
                const int kr0 = 32 * (s & 1) + 8 * gp + q, kr1 = kr0 + 4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(tile + kr0 * 256 + ((chunk ^ fm_swz(kr0)) << 4) + half * 8));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(tile + kr1 * 256 + ((chunk ^ fm_swz(kr1)) << 4) + half * 8));
                const bf16x8 fa = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                if (s >= 1 && s <= 6) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, wb0[s - 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, wb1[s], acc1, 0, 0, 0);
            }
            // acc[r] = dT^T[channel cb + 4 g + r][source mi]
            *reinterpret_cast<uint2*>(ost[0] + mi * FM_SROW + (cb + 4 * g) * 2) = make_uint2(pack2bf(acc0[0], acc0[1]), pack2bf(acc0[2], acc0[3]));
            *reinterpret_cast<uint2*>(ost[1] + mi * FM_SROW + (cb + 4 * g) * 2) = make_uint2(pack2bf(acc1[0], acc1[1]), pack2bf(acc1[2], acc1[3]));
        }
        __syncthreads();                                   // window reads done, output tiles complete
        store_half(phase, sreg);                           // the new right half replaces this block's left half
        {
            const int src = threadIdx.x >> 4;              // 16 sources of the 1/2 map x 16 chunks
            const uint4 o = *reinterpret_cast<const uint4*>(ost[0] + src * FM_SROW + 16 * ck);
            *reinterpret_cast<uint4*>(a.d2 + (((int64_t)b * h2 + 4 * ty + (src >> 2)) * w2 + 4 * tx + (src & 3)) * a.C + c0 + 8 * ck) = o;
            if (threadIdx.x < 80) {                        // 4 sources of the 1/4 map, 1 of the 1/8 map
                const uint4 o1 = *reinterpret_cast<const uint4*>(ost[1] + src * FM_SROW + 16 * ck);
                bf16_t* dst = src < 4 ? a.d4 + (((int64_t)b * h4 + 2 * ty + (src >> 1)) * w4 + 2 * tx + (src & 1)) * a.C
                                      : a.d8 + (((int64_t)b * h8 + ty) * w8 + tx) * a.C;
                *reinterpret_cast<uint4*>(dst + c0 + 8 * ck) = o1;
            }
        }
        __syncthreads();                                   // window complete for the next block, output tiles free
    }
}

int fuse_map_bwd_supported(int dt, int B, int H, int W, int C) {
    return dt == SEGF_BF16 && B > 0 && H % 8 == 0 && W % 8 == 0 && H >= 8 && W >= 8 && C % FM_SLICE == 0 &&
           (int64_t)H * W < (1ll << 30) && !POL(no_bwd248_mfma);
}
int fuse_map_bwd_launch(int B, int H, int W, int C, const void* dy, int64_t ldo, void* d2, void* d4, void* d8, hipStream_t st) {
    FuseMapBwdArgs a{(const bf16_t*)dy, ldo, (bf16_t*)d2, (bf16_t*)d4, (bf16_t*)d8, B, H, W, C, C / FM_SLICE};
    const int64_t wgs = (int64_t)B * (H / 8) * a.nslice;
    if (wgs <= 0 || wgs >= (1ll << 31)) return SEGF_ERR_SHAPE;
    hipLaunchKernelGGL(fuse_map_bwd_kernel, dim3((unsigned)wgs), dim3(256), 0, st, a);
    SEGF_CHECK_LAUNCH();
    return 0;
}
