/*
 * segfac.h -- C ABI of libsegfac_hip.so: the MI355X (gfx950) kernels underneath the
 * Segmentation_Factory plugin API (SegmentationModel / backbones / heads / engine.criterion /
 * util.metrics).  The reference has no native boundary on this path (it is eager PyTorch,
 * SURVEY.md section 2.2); each entry point below names the reference code it stands in for.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch caching allocator);
 *     kernels never allocate, free or keep pointers; workspaces are caller-provided.
 *   - activations are token-major / NHWC: element (row, c) at base[row * ld + c]; `ld` in elements.
 *   - dt: SEGF_F32 (0) or SEGF_BF16 (1) = storage type of activations; all arithmetic,
 *     statistics and accumulation are fp32.  Parameters and parameter gradients are fp32.
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue, never synchronise.
 *   - return 0 = enqueued; <0 = argument error (-1 shape/alignment, -2 dtype, -3 workspace);
 *     >0 = hipError_t from the launch.  Nothing is printed.
 */
#ifndef SEGFAC_H
#define SEGFAC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEGF_F32 0
#define SEGF_BF16 1
#define SEGF_ERR_SHAPE (-1)
#define SEGF_ERR_DTYPE (-2)
#define SEGF_ERR_WORKSPACE (-3)

/* library / build info: returns a static string "segfac-hip <version> gfx950" */
const char* segf_version(void);

/* ---- elementwise plumbing ------------------------------------------------------------------- */
/* dst[i] = (dst_dt) src[i] */
int segf_cast(const void* src, int src_dt, void* dst, int dst_dt, int64_t n, void* stream);
/* dst[r][c] = (dst_dt) src[r][c], r < rows, c < cols, with leading dimensions (elements): weight packing / column slices */
int segf_cast2d(const void* src, int src_dt, int64_t ld_src, void* dst, int dst_dt, int64_t ld_dst, int64_t rows, int64_t cols,
                void* stream);
/* out[a][c][b] = in[a][b][c]  (+ zero padding of the last output dim up to ldb_out):
 * OIHW -> O(HW)I weight re-layout for NHWC im2col GEMMs (models/backbones/mit.py:105 conv weights),
 * NCHW <-> NHWC at the plugin boundary (mit.py:198 permute).  */
int segf_permute021(const void* in, int in_dt, void* out, int out_dt, int64_t A, int64_t Bd, int64_t Cd,
                    int64_t ld_out, void* stream);
/* Several of the three jobs above in ONE launch (the small-batch step is bound by its launch count, train_gpu.py:71 default batch 4):
 *   op 0  dst[r][c] = (dst_dt) src[r][c]                       rows x cols, leading dimensions ld_src / ld_dst (segf_cast2d)
 *   op 1  dst[a][c][b] = (dst_dt) src[a][b][c], a < rows, b < pb, c < pc; output row length ld_dst >= pb, the rest zero (segf_permute021);
 *         cols > 0: element distance between consecutive dst[a] blocks (default pc * ld_dst)
 *   op 2  dst[r][c] = 0                                        rows x cols at ld_dst
 * Jobs of one call must not overlap in what they write / read from each other: they run concurrently. */
typedef struct SegfPrepItem {
    const void* src; void* dst;
    int64_t rows, cols, ld_src, ld_dst, pb, pc;
    int32_t op, src_dt, dst_dt, reserved;
} SegfPrepItem;
int segf_prep_grouped(int n, const SegfPrepItem* items, void* stream);
/* y[r][c] = x[r][c] * scale[r / rows_per_group]   (DropPath backward, models/layers/drop_path.py:18-25) */
int segf_scale_rows(int dt, const void* x, int64_t ldx, void* y, int64_t ldy, const float* scale,
                    int64_t rows, int64_t cols, int64_t rows_per_group, void* stream);
/* y = a + b  (residual / gradient fan-in), 2-D with leading dims */
int segf_add(int dt, const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy,
             int64_t rows, int64_t cols, void* stream);
/* out[c] = sum_r x[r][c] as fp32 (bias gradients of nn.Linear / 1x1 conv).  ws >= segf_colsum_ws(rows, cols) floats */
int64_t segf_colsum_ws(int64_t rows, int64_t cols);
int segf_colsum(int dt, const void* x, int64_t ldx, int64_t rows, int64_t cols, float* out, float* ws, void* stream);

/* zero-fill / counter increment / stochastic-layer scales as library kernels (no framework kernels inside the captured step).
 * segf_bernoulli_scale: out[i] = U_i < keep_prob[i / row_len] ? 1 / keep_prob[i / row_len] : 0 -- DropPath
 * (models/layers/drop_path.py:18-25) with rows = draws and row_len = batch, Dropout2d (heads/segformer.py:40) with one row of
 * B * C entries; state = {seed, launch counter} (uint64[2], advanced by the kernel: graph replays draw fresh numbers). */
int segf_zero(void* p, int64_t nbytes, void* stream);
int segf_add_i64(int64_t* p, int64_t v, void* stream);
/* Metrics.update's `self.hist += bincount(...)` (util/metrics.py:24-27): hist fp32 [n] += (float)counts int64 [n] (torch's promotion of
 * float32 += int64: each count rounded to fp32, then added); clear != 0 also zeroes the counts for the next batch. */
int segf_hist_accum(float* hist, int64_t* counts, int64_t n, int clear, void* stream);
/* test hook: a single wave that occupies `stream` for `us` microseconds (<= 200000), e.g. to delay a gradient in the data-parallel
 * ordering test (the event-ordered exchange of train_gpu.py:233-236's DDP replacement). */
int segf_debug_spin(int64_t us, void* stream);
/* Tuning switch of the eight-phase GEMM (gemm8.hip; the UPerHead / PPM 3x3 ConvModules of heads/upernet.py:26-31, modules/ppm.py:19 and
 * the ConvNeXt block MLPs on fp8 operands): what = 0 selects whether the fp8 kernels run their two wave groups one barrier apart
 * (value 1, default) or in lockstep (value 0); any other value only reads.  Returns the previous setting (or SEGF_ERR_SHAPE).  The host
 * layer times both on the device at hand once per process: devices differ in the clock they hold under the denser schedule. */
int segf_gemm8_option(int what, int value);
int segf_bernoulli_scale(uint64_t* state, const float* keep_prob, int64_t n, int64_t row_len, float* out, void* stream);

/* ---- FP8 (OCP e4m3fn) forward GEMM: BASELINE cfg5 "ConvNeXtV2-L + UPerNet, fp8 MFMA weights" (the pointwise linears of
 * convnextv2.py:90-95; no fp8 exists in the reference -- an option of this build, tolerance stated in the tests) ------------
 * segf_quant_rows_fp8: q[r][k] = e4m3(x[r][k] / scale[r]), scale[r] = amax_k |x[r][k]| / 448 (token rows of the activations,
 * output-channel rows of the weights); x of dtype dt, K % 8 == 0.
 * segf_gemm_fp8: C[M][N] (bf16) = (A[M][K] . B[N][K]^T) * scale_a[m] * scale_b[n] + bias[n], optionally
 * residual[m][n] + rscale[m / rows_per_group] * (...); A, B e4m3 bytes; v_mfma_scale_f32_16x16x128_f8f6f4, fp32 accumulate;
 * K % 128 == 0 (segf_gemm_fp8_supported). */
int segf_quant_rows_fp8(int dt, int64_t rows, int K, const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, void* stream);
int segf_gemm_fp8_supported(int64_t M, int64_t N, int64_t K);
int segf_gemm_fp8(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const float* scale_a, const void* B, int64_t ldb,
                  const float* scale_b, const float* bias, const void* residual, int64_t ldr, const float* rscale,
                  int64_t rows_per_group, void* C, int64_t ldc, void* stream);

/* fp8 3x3 convolutions (the UPerHead / PPM ConvModules, heads/upernet.py:26-31, modules/ppm.py:19: ~700 of cfg5's 2,050 GFLOP per
 * image).  segf_quant_tensor_fp8: ONE scale for a whole [rows][cols] tensor (the implicit-GEMM K axis gathers nine pixels, so
 * per-token scales do not factor out): q = cvt(x / scale), scale = amax |x| / FMAX; fmt 0 = e4m3fn (activations, FMAX 448),
 * 1 = e5m2 (gradients, FMAX 57344); amax_ws = 4 bytes of scratch; all on the stream, no host read (graph-safe).
 * segf_conv3x3_fp8: mode 0  y[pix][co] = sx * sw[co] * sum_{tap,ci} xq[pix+off(tap)][ci] wq[co][tap*Cin+ci]   (xq e4m3, wq e4m3 rows)
 *                   mode 1  dx[pix][ci] = sx * sw[ci] * sum_{tap,co} gq[pix-off(tap)][co] wq[ci][tap*Cout+co]  (gq e5m2, wq e4m3 rows)
 * bf16 output; Cin, Cout multiples of 16; ldx / ldw in bytes = elements, multiples of 16. */
int segf_quant_tensor_fp8(int dt, int fmt, int64_t rows, int cols, const void* x, int64_t ldx, void* q, int64_t ldq, float* scale,
                          void* amax_ws, void* stream);
int segf_conv3x3_fp8_supported(int mode, int B, int H, int W, int Cin, int Cout);
int segf_conv3x3_fp8(int mode, int B, int H, int W, int Cin, int Cout, const void* xq, int64_t ldx, const float* sx,
                     const void* wq, int64_t ldw, const float* sw, void* y, int64_t ldy, void* stream);
/* ... and the weight gradient on the SAME quantised tensors: dW[co][tap*Cin+ci] = sg * sx * sum_pix gq[pix][co] xq[pix+off(tap)][ci]
 * (gq: the e5m2 gradient of the data-gradient call, xq: the e4m3 input of the forward call; strides in bytes), fp32 [Cout][9*Cin];
 * split_k > 1 needs ws >= split_k * Cout * 9 * Cin floats.  Cin % 128 == 0, Cout % 256 == 0, B*H*W >= 65536. */
int segf_conv3x3_fp8_wgrad_supported(int B, int H, int W, int Cin, int Cout);
int segf_conv3x3_fp8_wgrad(int B, int H, int W, int Cin, int Cout, const void* xq, int64_t ldx, const float* sx, const void* gq,
                           int64_t ldg, const float* sg, float* dw, int64_t lddw, int split_k, float* ws, void* stream);

/* nn.Linear products on fp8 operands with one scale per activation / gradient TENSOR and one per weight row (the ConvNeXt block MLPs,
 * convnextv2.py:83-113), on the 256 x 256 tile kernel:
 *   mode 0  y[m][n] = sa * sb[n] * sum_k Aq[m][k] Bq[n][k] (+ bias[n]); with residual: y = residual + rscale[m / rows_per_group] * (..)
 *           Aq e4m3 [M][K] (segf_quant_tensor_fp8, fmt 0), Bq e4m3 [N][K] (segf_quant_rows_fp8 of the weight), bf16 out
 *   mode 1  the same with Aq in e5m2: the data gradient dx = dy W (Aq = the quantised gradient [M][N_out], Bq = W^T quantised per row)
 *   segf_linear_fp8_wgrad: dW[n][k] = sg * sx * sum_t gq[t][n] xq[t][k], gq e5m2 [T][N], xq e4m3 [T][K] -- the tensors mode 1 / mode 0
 *           were called with -- fp32 out; split_k > 1 needs ws >= split_k * N * K floats.
 * M (T), N multiples of 256, K of 128; strides in bytes = elements, multiples of 16.  segf_linear_fp8_supported(mode, M, N, K) with
 * mode 2 = the weight gradient (M = T tokens, N x K the weight). */
int segf_linear_fp8_supported(int mode, int64_t M, int64_t N, int64_t K);
int segf_linear_fp8(int mode, int64_t M, int64_t N, int64_t K, const void* Aq, int64_t lda, const float* sa, const void* Bq,
                    int64_t ldb, const float* sb, void* C, int64_t ldc, const float* bias, const void* residual, int64_t ldr,
                    const float* rscale, int64_t rows_per_group, void* stream);
int segf_linear_fp8_wgrad_splitk(int64_t N, int64_t K, int64_t T);      /* slices over the tokens (sizes ws) */
int segf_linear_fp8_wgrad(int64_t N, int64_t K, int64_t T, const void* gq, int64_t ldg, const float* sg, const void* xq,
                          int64_t ldx, const float* sx, float* dw, int64_t lddw, int split_k, float* ws, void* stream);

/* ---- stream ordering for the data-parallel exchange (train_gpu.py:233-236: DistributedDataParallel overlaps the gradient
 * all-reduce with backward through per-bucket hooks).  segf_event_record(.., external=1) during a stream capture adds an
 * EVENT-RECORD NODE to the hipGraph (hipEventRecordExternal); at each replay a stream outside the graph can
 * segf_stream_wait_event() on it: "this bucket of gradients is final".  `event` / `stream` are hipEvent_t / hipStream_t. */
int segf_event_create(void** event);
int segf_event_destroy(void* event);
int segf_event_record(void* event, void* stream, int external);
int segf_stream_wait_event(void* stream, void* event);

/* ---- GEMM: nn.Linear / 1x1 conv / im2col'd conv, forward and both backward products -------------
 * C[M,N] = epilogue( sum_k A(m,k) * B(k,n) ), fp32 accumulate.
 *   layout 0: A stored [M][K] (lda), B stored [N][K] (ldb)   y  = x W^T      (F.linear forward)
 *   layout 1: A stored [M][K],       B stored [K][N]         dx = dy W
 *   layout 2: A stored [K][M],       B stored [K][N]         dW = dy^T x     (K = token count)
 * epilogue: v = acc (+ bias[n]); if residual: v = residual[m][n] + (rscale ? rscale[m / rows_per_group] : 1) * v
 * c_dt may be SEGF_F32 while dt is SEGF_BF16 (parameter gradients).
 * split_k > 1 (layout 2 only) needs ws >= split_k * M * N floats; result is reduced deterministically.
 * bf16 uses v_mfma_f32_16x16x32_bf16; f32 uses an fp32 FMA kernel (exact-fp32 parity mode).       */
int segf_gemm(int dt, int layout, int64_t M, int64_t N, int64_t K,
              const void* A, int64_t lda, const void* B, int64_t ldb,
              void* C, int c_dt, int64_t ldc,
              const float* bias, const void* residual, int64_t ldr,
              const float* rscale, int64_t rows_per_group,
              int split_k, float* ws, void* stream);
int segf_gemm_pick_splitk(int64_t M, int64_t N, int64_t K);
/* Product whose ACTIVATION operand is normalised on the way into LDS: ConvModule's BatchNorm2d + ReLU and the Dropout2d
 * channel scale in front of SegFormerHead.linear_pred (heads/segformer.py:21-29,40,57-58), folded into per-(sample, channel)
 * tables by segf_bn_affine_table, so that the normalised [B*H*W, C] tensor is never written or re-read:
 *   layout 0:  C[M,N] = act(A s + t) B^T + bias      (A [M tokens][K], B [N][K]; bf16 out)
 *   layout 2:  C[M,N] = A^T act(B s + t)             (A = dy [K tokens][M], B = x [K tokens][N]; fp32 out, split-K)
 * s, t: fp32 [tokens / rows_per_group][features] (features = K in layout 0, N in layout 2); act 0 none / 1 ReLU / 2 ReLU6
 * (ReLU6 only with unit channel scale).  Implemented by the 256x256-tile kernel only: segf_gemm_pro_supported() tells
 * whether a shape qualifies (the caller otherwise materialises the normalised tensor with segf_bn_apply). */
int segf_gemm_pro_supported(int dt, int layout, int64_t M, int64_t N, int64_t K, int64_t rows_per_group);
int segf_gemm_pro(int dt, int layout, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                  void* C, int c_dt, int64_t ldc, const float* bias, int split_k, float* ws, const float* pro_scale,
                  const float* pro_shift, int64_t rows_per_group, int act, void* stream);
int segf_bn_affine_table(const float* mean, const float* rstd, const float* gamma, const float* beta, const float* chan_scale,
                         int groups, int C, float* scale, float* shift, void* stream);
/* Weight gradient and bias gradient of nn.Linear / 1x1 conv in ONE pass over dy (mit.py:45,52,58,98-99 backward):
 *   C[M,N] = sum_k A(k,m) B(k,n)  (layout 2: A = dy [K tokens][M], B = x [K tokens][N]),  dbias[m] = sum_k A(k,m)  (fp32).
 * The column sums ride on the matrix pipe (an all-ones operand) inside the GEMM; shapes that take the 256x256-tile or fp32
 * kernels run a separate column reduction behind the product.  ws >= segf_gemm_dw_db_ws(M, N, K, split_k) floats. */
int64_t segf_gemm_dw_db_ws(int64_t M, int64_t N, int64_t K, int split_k);
int segf_gemm_dw_db(int dt, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                    void* C, int c_dt, int64_t ldc, int split_k, float* ws, float* dbias, void* stream);
/* The same for SEVERAL layers in one call (the backward of a MiT / ConvNeXt block: q, kv, proj, fc1, fc2 of mit.py:43-59,98-99 -- their
 * weight gradients do not depend on each other).  Items that take the 128-tile split-K kernel are gathered into grouped launches (one
 * product launch + one reduce launch per up to 12 layers); the rest run exactly as segf_gemm_dw_db.  Every item's dw [M][lddw] fp32 and
 * db [M] are (with shared_split == 0) bitwise what segf_gemm_dw_db(dt, M, N, K, dy, lddy, x, ldx, dw, F32, lddw, split_k, ws, db) produces; ws per item >=
 * segf_gemm_dw_db_ws(M, N, K, split_k) floats, 16-byte aligned. */
typedef struct SegfDwItem {
    int64_t M, N, K;            /* dw = dy^T x: dy [K][M], x [K][N] (K = tokens) */
    const void* dy; int64_t lddy;
    const void* x; int64_t ldx;
    float* dw; int64_t lddw;
    float* db;
    float* ws;
    int split_k;
    int shared_split;           /* != 0: split_k is an UPPER BOUND -- inside a group the flagged members may run with fewer slices (one common
                                 * K range per slice, chosen so that the group as a whole fills the chip); 0: exactly split_k, bitwise
                                 * segf_gemm_dw_db */
} SegfDwItem;
int segf_gemm_dw_db_grouped(int dt, int n, const SegfDwItem* items, void* stream);

/* ---- LayerNorm over the last dim (nn.LayerNorm eps 1e-5 in mit.py:107,136-140,178-190; ConvNeXt's
 * channels-first LayerNorm, convnext.py:8-23, is the same kernel on NHWC rows) ---------------------- */
int segf_layernorm_fwd(int dt, int64_t rows, int C, const void* x, const float* gamma, const float* beta,
                       float eps, void* y, float* mean, float* rstd, void* stream);
int64_t segf_layernorm_bwd_ws(int64_t rows, int C);
int segf_layernorm_bwd(int dt, int64_t rows, int C, const void* x, const void* dy, const float* gamma,
                       const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                       float* ws, void* stream);
/* the same with two optional fan-in operands, so that a pre-norm residual block needs no separate gradient additions
 * (mit.py:143-146 `x + drop_path(f(norm(x)))`: dx = dres + LN_bwd(dy)) and a normalised map with two consumers (the stage
 * output feeding the head and the next patch embedding, mit.py:196-216) sums their gradients on load:
 *   dx = LN_bwd(dy + dy2) + dres;   dy2, dres nullable, [rows][C] of dtype dt. */
int segf_layernorm_bwd_fused(int dt, int64_t rows, int C, const void* x, const void* dy, const void* dy2, const void* dres,
                             const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                             float* ws, void* stream);
/* ... with a second output dxs[r][c] = (dt) dx[r][c] * rscale[r / rows_per_group] (NULL, NULL: none): when dx's consumer is the backward of
 * a residual branch x + DropPath(f(.)) (mit.py:143-146, drop_path.py:18-25), the first thing it does is this row scaling of dx --
 * segf_scale_rows on the stored dx, bitwise.  rows < 2^22. */
int segf_layernorm_bwd_scaled(int dt, int64_t rows, int C, const void* x, const void* dy, const void* dy2, const void* dres,
                              const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                              float* ws, const float* rscale, int64_t rows_per_group, void* dxs, void* stream);
/* The spatial-reduction convolution of MiT's attention (mit.py:20-22,47-48: Conv2d(dim, dim, sr, sr) on the LayerNorm output) without
 * im2col / col2im passes: segf_layernorm_fwd_patch also writes its output in the PATCH-MAJOR row order of that convolution's im2col matrix
 * (token (b, y, x) -> row ((b Ho + y / sr) Wo + x / sr), chunk (y % sr, x % sr)), so y2 viewed as [rows / sr^2][sr^2 C] is the matrix itself;
 * segf_layernorm_bwd_patch reads its fan-in operand dy2 in the same order -- the data gradient of the convolution as its product leaves it.
 * Map width W = 2^log2_w, sr = 2^log2_sr, rows % (W sr) == 0; log2_w < 0 in the backward: dy2 in token order (segf_layernorm_bwd_scaled). */
int segf_layernorm_fwd_patch(int dt, int64_t rows, int C, const void* x, const float* gamma, const float* beta, float eps, void* y,
                             float* mean, float* rstd, void* y2, int log2_w, int log2_sr, void* stream);
int segf_layernorm_bwd_patch(int dt, int64_t rows, int C, const void* x, const void* dy, const void* dy2, const void* dres,
                             const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                             float* ws, const float* rscale, int64_t rows_per_group, void* dxs, int log2_w, int log2_sr, void* stream);
/* Deferred finalize: with dgamma == NULL segf_layernorm_bwd_fused leaves its per-block partial sums [blocks][2 C] in ws
 * (blocks = segf_layernorm_bwd_blocks(rows, C)) and the caller finalizes SEVERAL such reductions in one launch later:
 * out[i] = sum_b partial[b][i], i < len, summed in the order of the single finalize (bitwise the same dgamma / dbeta). */
typedef struct SegfFinalizeItem { const float* partial; float* out; int64_t len; int nblk; int scatter_c; } SegfFinalizeItem;
/* scatter_c = C > 0 (len must be 10 C): the sums are the [10][C] partials of segf_dwconv3x3_gelu_bwd called with dw == NULL
 * (segf_dwconv3x3_bwd_blocks(dt, B, H, W, C) blocks in its ws) and land as out = dw[C][9] followed by db[C] (mit.py:62-71 backward). */
int segf_layernorm_bwd_blocks(int64_t rows, int C);
int segf_colreduce_finalize_grouped(int n, const SegfFinalizeItem* items, void* stream);

/* ---- BatchNorm2d (train: batch statistics) + ReLU/ReLU6 + Dropout2d, NHWC rows -------------------
 * ConvModule of heads/segformer.py:21-29, layers/conv_module.py:4-9, mobilenetv2.py:5-11.
 * stats: mean/var(biased) over rows; running stats updated with momentum and unbiased var.         */
int64_t segf_bn_ws(int64_t rows, int C);
int segf_bn_stats(int dt, int64_t rows, int C, const void* x, float* mean, float* rstd,
                  float* running_mean, float* running_var, float momentum, float eps, float* ws, void* stream);
/* y = act(gamma * (x-mean)*rstd + beta) * (chan_scale ? chan_scale[row / rows_per_sample][c] : 1); act 0 none, 1 relu, 2 relu6 */
int segf_bn_apply(int dt, int64_t rows, int C, const void* x, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, int act, const float* chan_scale,
                  int64_t rows_per_sample, void* y, void* stream);
/* training backward (batch statistics): dx, dgamma, dbeta.  eval_mode=1: statistics are constants. */
int segf_bn_bwd(int dt, int64_t rows, int C, const void* x, const void* dy, const float* mean, const float* rstd,
                const float* gamma, const float* beta, int act, const float* chan_scale, int64_t rows_per_sample,
                int eval_mode, void* dx, float* dgamma, float* dbeta, float* ws, void* stream);

/* BatchNorm backward of SegFormerHead's fuse ConvModule WITH the classifier's data gradient folded in (heads/segformer.py:21-29,
 * 40,57-58: linear_fuse.bn / activate -> Dropout2d -> linear_pred): instead of materialising da = dy W ([tokens, C]) and
 * streaming it twice through segf_bn_bwd, both BatchNorm passes recompute their da tile on the matrix pipe from the class
 * gradients dy [M][ldy >= K] (K = padded class count, a multiple of 32, pad columns zero) and w [K][ldw >= C] (classifier weight,
 * rows = classes, pad rows zero).  Same results as segf_gemm(layout 1) + segf_bn_bwd with da kept in fp32.
 * bf16 only; C % 128 == 0; rows_per_sample % 16 == 0: ask segf_bn_cls_bwd_supported.  ws >= segf_bn_cls_bwd_ws floats. */
int segf_bn_cls_bwd_supported(int dt, int64_t M, int C, int K, int64_t rows_per_sample);
int64_t segf_bn_cls_bwd_ws(int64_t M, int C, int64_t rows_per_sample);
int segf_bn_cls_bwd(int dt, int64_t M, int C, int K, const void* dy, int64_t ldy, const void* w, int64_t ldw, const void* x,
                    const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                    const float* chan_scale, int64_t rows_per_sample, int eval_mode, void* dx, float* dgamma, float* dbeta,
                    float* ws, void* stream);
/* The same with the NEXT backward step's weight-gradient product riding on pass 2: dx is the gradient of the folded SegFormerHead's
 * stride-4 map y = sum_i resize_i(x_i G_i^T) (heads/segformer.py:42-56); its stage-1 term needs dG_1 = dx^T x1 and colsum(dx), a
 * [C x C1] product over all M tokens.  dG: fp32 [C][C1 + 8] = [dx^T x1 | colsum(dx) | 0 x 7], accumulated from the bf16 dx tile each
 * workgroup has on chip (bit-identical operands to a separate product over the stored dx; saves one 2 M C byte pass).  C1 == 32,
 * K <= 160: ask segf_bn_cls_bwd_dw_supported.  ws >= segf_bn_cls_bwd_dw_ws floats. */
int segf_bn_cls_bwd_dw_supported(int dt, int64_t M, int C, int K, int64_t rows_per_sample, int C1);
int64_t segf_bn_cls_bwd_dw_ws(int64_t M, int C, int64_t rows_per_sample);
int segf_bn_cls_bwd_dw(int dt, int64_t M, int C, int K, const void* dy, int64_t ldy, const void* w, int64_t ldw, const void* x,
                       const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                       const float* chan_scale, int64_t rows_per_sample, int eval_mode, void* dx, float* dgamma,
                       float* dbeta, float* ws, const void* x1, int64_t ldx1, int C1, float* dG, void* stream);
/* The same with everything that can ride along, each part optional: (x1, dG) as in segf_bn_cls_bwd_dw (pass 2), and dwcls = the
 * CLASSIFIER's weight gradient fp32 [K][C] = dy^T act(bn(x)) * drop (heads/segformer.py:57-58 backward; act = 0 / 1 only),
 * formed in pass 1 from the x and dy tiles it already holds -- replaces the layout-2 segf_gemm_pro pass over x.
 * ws >= segf_bn_cls_bwd_full_ws(M, C, K, rows_per_sample) floats. */
int64_t segf_bn_cls_bwd_full_ws(int64_t M, int C, int K, int64_t rows_per_sample);
int segf_bn_cls_bwd_full(int dt, int64_t M, int C, int K, const void* dy, int64_t ldy, const void* w, int64_t ldw, const void* x,
                         const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                         const float* chan_scale, int64_t rows_per_sample, int eval_mode, void* dx, float* dgamma, float* dbeta,
                         float* ws, const void* x1, int64_t ldx1, int C1, float* dG, float* dwcls, void* stream);

/* ---- Global Response Normalization (ConvNeXtV2 GRN, convnextv2.py:68-80) on NHWC rows, B images of rows_per_sample rows:
 * y = gamma * (x * Nx) + beta + x, Nx = ||x||_2(H,W) / (mean_c ||x||_2 + 1e-6).  sumsq_out [B][C] is saved for the backward.
 * pre_gelu != 0: the input of the normalisation is gelu(x) of the stored tensor (nn.GELU between pwconv1 and grn, convnextv2.py:92-94):
 * the activation is applied on the way in and the backward returns the gradient of the PRE-activation, dx = dGRN * gelu'(x) -- gelu(x)
 * is never written or read. */
int64_t segf_grn_ws(int B, int64_t rows_per_sample, int C, int bwd);
int segf_grn_fwd(int dt, int B, int64_t rows_per_sample, int C, const void* x, const float* gamma, const float* beta, void* y,
                 float* a_scratch, float* sumsq_out, float* ws, int pre_gelu, void* stream);
int segf_grn_bwd(int dt, int B, int64_t rows_per_sample, int C, const void* x, const void* dy, const float* gamma,
                 const float* sumsq_saved, void* dx, float* dgamma, float* dbeta, float* ws, int pre_gelu, void* stream);

/* ---- MiT spatial-reduction attention core (mit.py:52-57): O = softmax(Q K^T * scale) V -----------
 * q: [B*N][ldq] with head h at columns h*hd; k, v likewise over B*Nkv rows; o: [B*N][ldo]; lse: [B][heads][N] */
int segf_attention_fwd(int dt, int B, int heads, int N, int Nkv, int hd, const void* q, int64_t ldq,
                       const void* k, int64_t ldk, const void* v, int64_t ldv, float scale,
                       void* o, int64_t ldo, float* lse, void* stream);
int64_t segf_attention_bwd_ws(int B, int heads, int N, int Nkv, int hd);
int segf_attention_bwd(int dt, int B, int heads, int N, int Nkv, int hd, const void* q, int64_t ldq,
                       const void* k, int64_t ldk, const void* v, int64_t ldv, float scale,
                       const void* o, int64_t ldo, const void* d_o, int64_t lddo, const float* lse,
                       void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                       float* ws, void* stream);

/* ---- depthwise 3x3 conv + bias + GELU(erf) on NHWC (mit.py:62-71,98-99 DWConv -> F.gelu) ----------- */
int segf_dwconv3x3_gelu_fwd(int dt, int B, int H, int W, int C, const void* x, const float* w /*[C][9]*/,
                            const float* bias, int apply_gelu, void* y, void* stream);
int64_t segf_dwconv3x3_bwd_ws(int B, int H, int W, int C);
/* du = dy * gelu'(conv(x)+b): `du` (same shape as x) is SCRATCH -- written by the three-pass forms, left untouched by the one-launch form of
 * small maps (H W <= 1024 in bf16 / 512 in fp32: the whole backward in LDS); dx = conv^T(du); dw[C][9], db[C] fp32 */
int segf_dwconv3x3_bwd_blocks(int dt, int B, int H, int W, int C);
int segf_dwconv3x3_gelu_bwd(int dt, int B, int H, int W, int C, const void* x, const float* w, const float* bias,
                            int apply_gelu, const void* dy, void* du, void* dx, float* dw, float* db,
                            float* ws, void* stream);

/* ---- depthwise 7x7 conv + bias on NHWC (ConvNeXt Block.dwconv, convnext.py:29,39; convnextv2.py:88,101) --------------
 * wt: fp32 [49][C] (the [C][1][7][7] parameter transposed with segf_permute021); C % 8 == 0.                    */
int segf_dwconv7x7_fwd(int dt, int B, int H, int W, int C, const void* x, const float* wt, const float* bias, void* y,
                       void* stream);
int64_t segf_dwconv7x7_bwd_ws(int B, int H, int W, int C);
/* dx = conv^T(dy) (skipped when dx == NULL); dw fp32 [C][49]; db fp32 [C] (nullable) */
int segf_dwconv7x7_bwd(int dt, int B, int H, int W, int C, const void* x, const float* wt, const void* dy, void* dx,
                       float* dw, float* db, float* ws, void* stream);

/* ---- 3x3 conv, stride 1, pad 1, NHWC, as an implicit MFMA GEMM (no im2col buffer): ConvModule(.., 3, 1, 1) of
 * heads/upernet.py:26,28, modules/ppm.py:19, heads/fpn.py:19.  bf16 only; P = B*H*W pixels.
 *   mode 0: y[P][ldy]  = conv(x[P][ldx], w[Cout][9*Cin])            (+ bias[Cout], nullable)
 *   mode 1: y = dx[P][ldy] = conv^T: x := dy[P][ldx], w := wt[Cin][9*Cout] (weights transposed to [ci][tap][co])
 *   mode 2: y = dw fp32 [Cout][9*Cin] (ldy): x[P][ldx], w := dy[P][ldw]; split_k slices P, ws >= split_k*Cout*9*Cin floats */
/* Modes 0 / 1 with FEW output tiles over a long reduction (PPM bottleneck 3840 -> 768 on 16 x 16, ppm.py:19; the 40 x 40 / 20 x 20 levels):
 * segf_conv3x3_fwd_splitk > 1 = run with that split_k and ws of split_k * P * N floats (bf16 output, no bias): K slices of the eight-phase
 * tile, fp32 partials, one reduce pass.  Any other split_k for modes 0 / 1 is ignored. */
int segf_conv3x3_fwd_splitk(int mode, int B, int H, int W, int Cin, int Cout);
/* split-K count for mode 2 of segf_conv3x3 (size ws with it): segf_gemm_pick_splitk for the shapes that take the generic rule, a few slices
 * of the eight-phase tile for large outputs over short reductions (the FPN levels of UPerHead, upernet.py:26-28) */
int segf_conv3x3_pick_splitk(int Cin, int Cout, int64_t P);
int segf_conv3x3(int mode, int B, int H, int W, int Cin, int Cout, const void* x, int64_t ldx, const void* w, int64_t ldw,
                 void* y, int y_dt, int64_t ldy, const float* bias, int split_k, float* ws, void* stream);

/* y = gelu_erf(u) (mode 0) or dy * gelu_erf'(u) (mode 1), flat, n % 8 == 0 (nn.GELU, convnext.py:32,43) */
int segf_gelu(int dt, int mode, const void* u, const void* dy, void* y, int64_t n, void* stream);
/* out[r] = sum_c a[r][c] b[r][c] (+ extra_a[r] extra_b[r]), fp32: d gamma of a layer scale folded into the weights */
int segf_rowdot(const float* a, int64_t lda, const float* b, int64_t ldb, const float* extra_a, const float* extra_b,
                float* out, int64_t rows, int64_t cols, void* stream);
/* nn.AdaptiveAvgPool2d(S) on NHWC (modules/ppm.py:13): bwd=0 in[B][H][W][C] -> out[B][S][S][C]; bwd=1 the transpose */
int segf_adaptive_avgpool(int dt, int bwd, int B, int H, int W, int C, int S, const void* in, void* out, void* stream);

/* ---- im2col / col2im for strided convs (PatchEmbed mit.py:105,127; sr conv mit.py:21,48) ---------
 * col[(b,oy,ox)][(ky,kx,ci)] with leading dim ldcol (>= kh*kw*Cin, pad columns are zeroed).
 * in_nchw_f32=1: input is the fp32 NCHW image (train_gpu.py tensor contract); else NHWC of dtype dt. */
int segf_im2col(int dt, int in_nchw_f32, int B, int H, int W, int Cin, int kh, int kw, int stride, int pad,
                int Ho, int Wo, const void* x, void* col, int64_t ldcol, void* stream);
int segf_col2im(int dt, int B, int H, int W, int Cin, int kh, int kw, int stride, int pad,
                int Ho, int Wo, const void* dcol, int64_t ldcol, void* dx, void* stream);

/* ---- bilinear resize on NHWC (F.interpolate mode='bilinear'; heads/segformer.py:48, ppm.py:24,
 * upernet.py:41,46, build_models.py:65); out may be a channel slice of a wider buffer (ldo) ---------- */
int segf_bilinear_fwd(int dt, int B, int h, int w, int C, const void* in, int64_t ldi,
                      int H, int W, void* out, int64_t ldo, int align_corners, void* stream);
int segf_bilinear_bwd(int dt, int B, int h, int w, int C, void* din, int64_t ldi,
                      int H, int W, const void* dout, int64_t ldo, int align_corners, void* stream);
/* The transposes of the x2 / x4 / x8 bilinear upsamplings (align_corners = 0) of ONE gradient map in one pass over it: the
 * backward of the folded SegFormerHead's accumulation step (segf_upsample_add with sources at 1/2, 1/4, 1/8 of the grid,
 * heads/segformer.py:44-56).  dout [B][H][W][ldo >= C] -> d2 [B][H/2][W/2][C], d4 [B][H/4][W/4][C], d8 [B][H/8][W/8][C];
 * H % 8 == 0, W % 8 == 0, C % 8 == 0.  Same values as three segf_bilinear_bwd calls. */
int segf_bilinear_bwd_248(int dt, int B, int H, int W, int C, const void* dout, int64_t ldo, void* d2, void* d4, void* d8,
                          void* stream);
/* out = base + sum_{k<nsrc} bilinear_up(src_k), all NHWC of dtype dt, C % 8 == 0: the accumulation step of the folded
 * SegFormerHead (heads/segformer.py:44-56: Linear -> resize -> concat -> 1x1 conv is affine, and bilinear resizing
 * commutes with affine maps, so the per-scale products are formed at native resolution and added here).  nsrc <= 3.  */
int segf_upsample_add(int dt, int B, int H, int W, int C, const void* base, int64_t ldb, int nsrc,
                      const void* src0, int h0, int w0, int64_t ld0, const void* src1, int h1, int w1, int64_t ld1,
                      const void* src2, int h2, int w2, int64_t ld2, void* out, int64_t ldo, int align_corners, void* stream);
/* The same, also returning sums[2][C] = per-channel (sum, sum of squares) of the stored result over all B*H*W rows: the
 * BatchNorm batch statistics of the ConvModule that consumes it (heads/segformer.py:21-29), so no separate pass over the
 * [B*H*W, C] tensor is needed; segf_bn_stats_from_sums turns them into mean / rstd / running statistics.  Fused only for the
 * folded SegFormerHead geometry (three sources at 1/2, 1/4, 1/8 of the grid, align_corners = 0): SEGF_ERR_SHAPE otherwise
 * (use segf_upsample_add + segf_bn_stats).  ws >= segf_upsample_add_stats_ws(B, H, W, C) floats. */
int64_t segf_upsample_add_stats_ws(int B, int H, int W, int C);
int segf_upsample_add_stats(int dt, int B, int H, int W, int C, const void* base, int64_t ldb, int nsrc,
                            const void* src0, int h0, int w0, int64_t ld0, const void* src1, int h1, int w1, int64_t ld1,
                            const void* src2, int h2, int w2, int64_t ld2, void* out, int64_t ldo, int align_corners,
                            float* sums, float* ws, void* stream);
/* The folded SegFormerHead's stride-4 map in one pass on the matrix pipe (heads/segformer.py:42-56 after the fold):
 *   out[b,Y,X,:] = x1[b,Y,X,:] G1^T + bilinear(t2) + bilinear(t3) + bilinear(t4): the stage-1 product (K = C1 = 32 or 64) and the
 *   three align_corners=False resizes of the 1/2, 1/4, 1/8 maps t2, t3, t4 ([B*(H/r)*(W/r)][C] bf16, every bias already added:
 *   bilinear weights sum to one) as ONE accumulated MFMA product per 8 x 8 pixel block -- replaces segf_gemm (x1 G1^T written to
 *   memory) + segf_upsample_add_stats (read back, interpolated on the VALU).  bf16 only; H, W multiples of 8; C a multiple of 128.
 *   sums (nullable): fp32 [2][C] per-channel sum / sum of squares of the fp32 results (BatchNorm statistics of the consumer,
 *   heads/segformer.py:21-29); ws >= segf_fuse_map_248_ws floats when sums is given.  g1: [C][C1] bf16, row stride ldg. */
int segf_fuse_map_248_supported(int dt, int B, int H, int W, int C, int C1);
int64_t segf_fuse_map_248_ws(int B, int H, int W, int C);
int segf_fuse_map_248(int B, int H, int W, int C, int C1, const void* x1, int64_t ldx1, const void* g1, int64_t ldg,
                      const void* t2, int64_t ld2, const void* t3, int64_t ld3, const void* t4, int64_t ld4,
                      void* out, int64_t ldo, float* sums, float* ws, void* stream);
int segf_bn_stats_from_sums(const float* sums, int64_t rows, int C, float* mean, float* rstd, float* running_mean,
                            float* running_var, float momentum, float eps, void* stream);
/* Nearest-neighbour resize on dense NHWC (F.interpolate mode='nearest': the top-down step of FPNHead, heads/fpn.py:31,35).
 * bwd=0: out[B][H][W][C] = in[B][src(Y)][src(X)][C] (+ base[B][H][W][C], nullable: the `out + lateral` of fpn.py:34 fused);
 * bwd=1: out[B][h][w][C] = sums of in[B][H][W][C] over the destinations of each source.  Integer ratios: src(Y) = Y / (H / h); any
 * other (h, H) -- inputs that are not multiples of 32 give 3 x 3 -> 5 x 6 and 10 x 12 -> 9 x 12 steps, fpn.py:30-31 -- follows ATen:
 * src(Y) = min((int)floorf(Y * ((float)h / H)), h - 1).  C % 8 == 0. */
int segf_nearest_up(int dt, int bwd, int B, int h, int w, int C, int H, int W, const void* in, const void* base, void* out,
                    void* stream);
/* out (fp32 NCHW [B][C][H][W]) = bilinear(in NHWC [B][h][w][ldi]) -- materialised logits for API parity */
int segf_bilinear_to_nchw_f32(int dt, int B, int h, int w, int C, const void* in, int64_t ldi,
                              int H, int W, float* out, void* stream);

/* ---- fused final-upsample + CrossEntropy + Dice (build_models.py:65 + engine.py:10-15 +
 * util/losses.py:126-177) ------------------------------------------------------------------------
 * logits: NHWC [B][h][w][ldl] (low-res head output; h==H,w==W means "already full-res");
 * target: int64 [B][H][W]; stats (opaque to the caller, segf_ce_dice_stats_floats(B, C) floats): per image
 * {I[C], P[C], T[C], ce_sum, w_sum, n_valid, bad_label_flag}, then the batch totals [4], then block partials;
 * loss: fp32 [3] {total, ce, dice_loss}.  class_weight nullable ([C]).  dice=0 -> CE only.  C <= 192.
 * When H/h == W/w is a power of two the upsample is fused in both directions (no full-resolution tensor).
 * pix_lse (nullable; segf_ce_dice_lse_floats(...) floats, 0 = this configuration has none): scratch that carries the
 * per-pixel log-sum-exp from the forward to the backward of the SAME logits, so the backward does not recompute the softmax
 * normalisation (bf16, ratio-4 fused path only).  Pass the same buffer to both calls, or NULL to both.     */
int64_t segf_ce_dice_stats_floats(int B, int C);
int64_t segf_ce_dice_lse_floats(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl);
int segf_ce_dice_fwd(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl,
                     const int64_t* target, int64_t ignore_index, const float* class_weight, int dice,
                     float* stats, float* loss, float* pix_lse, void* stream);
/* dlogits: NHWC [B][h][w][ldd] of dtype dt = grad_out[0] * d loss / d logits (the LOW-resolution head output; the
 * transposed bilinear resize is applied inside; columns [C, ldd) are zeroed).  ws: segf_ce_dice_bwd_ws floats
 * (0 on the fused power-of-two path; the generic path stages the full-resolution gradient there).        */
int64_t segf_ce_dice_bwd_ws(int dt, int B, int C, int h, int w, int H, int W);
int segf_ce_dice_bwd(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl,
                     const int64_t* target, int64_t ignore_index, const float* class_weight, int dice,
                     const float* stats, const float* grad_out, void* dlogits, int64_t ldd, float* ws,
                     const float* pix_lse, void* stream);

/* test hook: out[16] = column sums of in[64][16] through the loss kernels' transposing wave reduction */
int segf_debug_wave_reduce16(const float* in, float* out, void* stream);

/* ---- fused upsample + argmax + confusion matrix (engine.py:89-91; util/utils.py:99-109;
 * util/metrics.py:24-27).  mat: int64 [n][n] += counts where 0<=t<n; hist: int64 [n][n] += counts
 * where t != ignore_label (t>=n and != ignore is skipped and flagged in flag[0]).                  */
int segf_argmax_confmat(int dt, int B, int C, int h, int w, int H, int W, const void* logits, int64_t ldl,
                        const int64_t* target, int64_t ignore_label, int64_t* mat, int64_t* hist,
                        int32_t* flag, int64_t* pred_out /*nullable [B][H][W]*/, void* stream);

/* out[r] = argmax_c x[r][c] (lowest index on ties): the prediction step of single-image inference
 * (estimate_model.py:104-106, softmax(dim=1).argmax(dim=1) -- softmax is monotonic).  x: [rows][ld >= C], C <= 192. */
int segf_argmax_rows(int dt, int64_t rows, int C, const void* x, int64_t ld, int64_t* out, void* stream);

/* ConfusionMatrix.update(a, b) on explicit int64 (ground truth, prediction) pairs (util/utils.py:99-109) and
 * the Metrics.update bincount (util/metrics.py:24-27); mat / hist nullable.  flag bit0: label >= n that is not
 * ignore_label, bit1: prediction out of range.                                                      */
int segf_confmat_pairs(const int64_t* gt, const int64_t* pred, int64_t n, int C, int64_t ignore_label,
                       int64_t* mat, int64_t* hist, int32_t* flag, void* stream);

/* ---- optimizer step (engine.py:52-53 -> timm NativeScaler -> AGC clip -> AdamW) over flat fp32
 * param/grad/state buffers.  A "unit" is one dim-0 row of a >=2-D weight or a whole 1-D tensor:
 * unit_offset[u], unit_len[u] (elements), unit_flags[u] bit0 = apply weight decay, bit1 = the parameter got no gradient
 * this step: the unit is skipped entirely, as torch.optim.AdamW skips `p.grad is None` (train_gpu.py:269).  unit_step
 * (nullable): per-unit step counts kept on the device and advanced by the kernel for every unit it updates -- torch's per-parameter
 * state['step'], which sets the bias corrections; when null the host scalar `step` is used for all units.  clip_factor <= 0
 * disables AGC.  See oracle/optim.py for the restated arithmetic (timm 0.9.2; "parity unpinned").   */
int segf_agc_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   const int64_t* unit_offset, const int32_t* unit_len, const uint8_t* unit_flags, int32_t* unit_step,
                   int nunits, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   float clip_factor, float agc_eps, void* stream);

/* The other --clip-mode values of the reference (train_gpu.py:99-102 -> timm.utils.dispatch_clip_grad) on the flat gradient buffer:
 * mode 0 'norm' = torch.nn.utils.clip_grad_norm_(params, value, 2.0): g *= min(1, value / (||g||_2 + 1e-6));
 * mode 1 'value' = clip_grad_value_: g = clamp(g, -value, value).  ws: segf_clip_grad_ws() floats (mode 0).  No host read. */
int64_t segf_clip_grad_ws(void);
int segf_clip_grad(float* grad, int64_t n, int mode, float value, float* ws, void* stream);

/* ---- device-side input pipeline over decoded uint8 images resident in HBM ---------------------------
 * segf_input_train: ExtRandomCrop -> ExtColorJitter -> ExtRandomHorizontalFlip -> ExtToTensor -> ExtNormalize
 * (datasets/build_datasets.py:14-22; datasets/extra_transform.py:319-392, 426-509, 196-214, 259-281, 288-313) and the dataset
 * classes' label table + .long() (datasets/ade.py:122-124, cityscapes.py:159, coco_stuff.py:95-100) for a batch of B samples
 * cropped to H x W.  Pillow's uint8 arithmetic (ImagingBlend, rgb2l, ImageStat mean) bit for bit; the float tail
 * ((u8 / 255) / 255 - mean) / std keeps the reference's second / 255 (quirk Q11).  One segf_input_sample per image: packed RGB
 * rows (img_stride bytes apart) and uint8 label rows; the crop window starts at (top, left) and reads 0 outside the source
 * (Image.crop); `order` lists the jitter steps in application order, 2 bits each (1 brightness, 2 contrast, 3 saturation,
 * 0 end), factor[k] belongs to step k; the random values are drawn by the caller.  lsum_ws: B x uint64 (luma sums for the
 * contrast means); label_lut nullable (int64[256]; identity when null).  out_img fp32 [B][3][H][W], out_lbl int64 [B][H][W]. */
typedef struct {
    const uint8_t* img;
    const uint8_t* lbl;
    int64_t img_stride, lbl_stride;
    int32_t src_h, src_w, top, left;
    int32_t flip, order;
    float factor[3];
    int32_t reserved;
} segf_input_sample;                                                     /* 72 bytes */
int segf_input_train(const segf_input_sample* samples /*device [B]*/, int B, int H, int W, uint64_t* lsum_ws, const float* mean3,
                     const float* std3, const int64_t* label_lut, float* out_img, int64_t* out_lbl, void* stream);
/* segf_input_val: ExtResize -> ExtToTensor -> ExtNormalize (build_datasets.py:24-29; extra_transform.py:395-419) for ONE image:
 * Image.resize((out_w, out_h), BILINEAR) = ImagingResample's horizontal then vertical pass in 22-bit fixed point, each rounded to
 * uint8, and Image.resize(..., NEAREST) for the label (ImagingScaleAffine's accumulated source index); the caller computes
 * (out_h, out_w) (smaller edge -> size, other edge int(size * long / short)).  ws: segf_input_val_ws(...) BYTES, 16-byte aligned.
 * out_img fp32 [3][out_h][out_w], out_lbl int64 [out_h][out_w]. */
/* Single-image inference preprocessing (estimate_model.py:85-97 with torchvision 0.15.2, environment.yml:22): T.Resize((out_h, out_w)) of a
 * uint8 CHW TENSOR = bilinear, align_corners = False, no antialias, computed in float32 and rounded half-to-even back to uint8, then
 * x / 255 and Normalize(mean, std).  img: uint8 [3][src_h][src_w] on the device; out: fp32 [3][out_h][out_w]. */
int segf_infer_preprocess(const uint8_t* img, int src_h, int src_w, int out_h, int out_w, const float* mean3, const float* std3,
                          float* out, void* stream);
int64_t segf_input_val_ws(int src_h, int src_w, int out_h, int out_w);
int segf_input_val(const uint8_t* img, int64_t img_stride, const uint8_t* lbl, int64_t lbl_stride, int src_h, int src_w, int out_h,
                   int out_w, void* ws, const float* mean3, const float* std3, const int64_t* label_lut, float* out_img,
                   int64_t* out_lbl, void* stream);

/* ---- Dispatch policy and launch trace (csrc/policy.h, csrc/policy.hip) -------------------------------------------------------------
 * Every switch that can change WHICH kernel (or which form of a kernel) a call takes lives in one table, csrc/policy.h, and is read
 * from the environment ONCE, at first use, into one struct: nothing on the launch path calls getenv().  A set variable that is not a
 * number counts as 1; "0" equals unset for the on/off switches.  The table (the reference has no counterpart: it dispatches through
 * ATen; models/build_models.py:43-54 is where the BASELINE shapes these rules were measured on come from):
 *
 *   SEGFAC_GEMM_NO_BIG           never take the 256 x 256-tile kernel (gemm_bf16_big_kernel): everything on the 128-tile one
 *   SEGFAC_GEMM_NO_SKINNY        no streaming products for K, N <= 128 (gemm_skinny_kernel and its relatives)
 *   SEGFAC_GEMM_NO_SKINNY_ROWS   the [tokens x 32] -> 768 projection on gemm_skinny_kernel instead of the whole-row form
 *   SEGFAC_GEMM_NO_SKINNY_K      no K-split streaming product for 32- / 64-wide outputs (gemm_skinny_k_kernel)
 *   SEGFAC_GEMM_NO_DW_SKINNY     small-output weight gradients on the tiled split-K kernel instead of gemm_dw_skinny_kernel
 *   SEGFAC_GEMM_NO_NARROW        256-tile kernel: full 128 x 64 wave tiles also for outputs <= 160 wide (bit-identical results)
 *   SEGFAC_GEMM_NO_DEEP          256-tile forward: one K step of operand loads in flight instead of two
 *   SEGFAC_GEMM_NO_DEEP128       128-tile kernel: one K step in flight instead of two
 *   SEGFAC_GEMM_NO_FASTLOAD      guarded tile loads everywhere (no workgroup-uniform guard-free full-K-step loads)
 *   SEGFAC_GEMM_NO_TR            debugging: reduction-major fragments by scalar LDS reads instead of ds_read_b64_tr_b16; disables every kernel built on the transposed read
 *   SEGFAC_GEMM_NO_PRO           segf_gemm_pro_supported answers 0: BatchNorm + ReLU + Dropout2d are applied by their own pass
 *   SEGFAC_GEMM_NO_FUSED_DB      bias gradient as its own column-sum launch instead of riding on the weight-gradient product
 *   SEGFAC_NO_GROUPED_DW         segf_gemm_dw_db_grouped runs its members one by one
 *   SEGFAC_DW_NO_XCD_SLABS       split-K weight gradients (128-tile kernel) in hardware workgroup order instead of one K slab per XCD
 *   SEGFAC_DW_NO_SHARED_SPLIT    grouped weight gradients keep their per-layer slice counts (also read by the host layer)
 *   SEGFAC_NO_WIDE_REDUCE        split-K partials of large outputs summed by the 16 x 16 form instead of whole rows
 *   SEGFAC_NO_REDUCE4            split-K reduce: one output per thread instead of four (bitwise the same sums)
 *   SEGFAC_GEMM_F32_NO_MFMA      fp32 storage (exact-parity mode, evaluate): products on the vector FMA kernel instead of the f32 matrix instruction
 *   SEGFAC_GEMM8_LINEAR          0: plain nn.Linear products never take the eight-phase kernel (the 256 / 128 tile kernels as in r04)  (default 1)
 *   SEGFAC_GEMM8_LINEAR_MIN_TILES fewest 256 x 256 tiles for which a K >= 2048 nn.Linear product takes the eight-phase kernel (192 for shorter K)  (default 128)
 *   SEGFAC_GEMM8_LINEAR_MIN_FILL smallest share (percent) of the launched 256 x 256 tiles that must be output for an nn.Linear product with ragged last tiles to take the eight-phase kernel  (default 60)
 *   SEGFAC_GEMM8_LINEAR_MIN_K    shortest reduction for which an nn.Linear product takes the eight-phase kernel (its 12-load prologue and drain against K / 64 tiles)  (default 256)
 *   SEGFAC_GEMM8_DW              0: nn.Linear weight gradients never take the eight-phase kernel below 65536 tokens (the grouped 128-tile kernel as in r04)  (default 1)
 *   SEGFAC_GEMM8_DW_MIN_GFLOP    smallest nn.Linear weight gradient (GFLOP; both feature counts multiples of 256, >= 256 FLOP per operand byte) that takes the eight-phase kernel + a column-sum pass instead of the grouped 128-tile kernel  (default 100)
 *   SEGFAC_GEMM8_LINEAR_MIN_GFLOP smallest nn.Linear product (GFLOP, K >= 512; 100 for shorter K) that takes the eight-phase kernel  (default 36)
 *   SEGFAC_NO_GEMM8              no eight-phase kernel at all (gemm8_kernel): the two-phase 256-tile kernel everywhere
 *   SEGFAC_NO_GEMM8T             no eight-phase kernel for weight gradients (reduction-major operands)
 *   SEGFAC_CONV_NO_FWD_SPLIT     3 x 3 forward / data gradient with few output tiles: no split over the (channel block, tap) walk
 *   SEGFAC_G8_STAGGER            eight-phase kernel: 1 = wave groups one barrier apart, 0 = lockstep, -1 = staggered for bf16 and the per-device choice (segf_gemm8_option) for fp8  (default -1)
 *   SEGFAC_NO_FP8                segf_gemm_fp8_supported answers 0 (block-scaled 128-tile fp8 GEMM)
 *   SEGFAC_NO_FP8_CONV           segf_conv3x3_fp8*_supported answer 0
 *   SEGFAC_NO_FP8_WGRAD          fp8 3 x 3 convolutions keep a bf16 weight gradient
 *   SEGFAC_NO_FP8_LINEAR         segf_linear_fp8_supported answers 0
 *   SEGFAC_ATTN_NO_MFMA          attention on the VALU reference kernels (attention.hip) also in bf16
 *   SEGFAC_ATTN_F32_NO_MFMA      fp32 attention forward on the vector kernel (one query per lane) instead of the f32 matrix instruction
 *   SEGFAC_ATTN64_PRESCALE       head dim 64, >= 128 keys: scale log2(e) rides on the Q fragments (bf16(q c), one more rounding per q element) and -max / -lse are the score accumulators' initial values, instead of one multiply-add per score: forward + query-side backward, +8 % / +2 % per kernel, attention error x 1.2 - 2.3
 *   SEGFAC_ATTN64_DKV_ROWS       head dim 64, >= 128 keys, key-side backward: query rows per staged Q / dO tile and barrier (128, 64 or 32: the same arithmetic, bit for bit)  (default 128)
 *   SEGFAC_ATTN_NO_FUSED_BWD     head dim 32, <= 256 keys: query-side + key-side backward kernels instead of the one-kernel backward
 *   SEGFAC_DW_NO_WALK            depthwise 3 x 3: the round-1 strip kernels instead of the vertical-walk kernels
 *   SEGFAC_DW_WALK_ROWS          depthwise 3 x 3 walk: rows per segment (0 = chosen from the map size)
 *   SEGFAC_DW_NO_SMALL           depthwise 3 x 3 backward of small maps: three passes instead of the one-launch LDS form
 *   SEGFAC_DW_SMALL_ALWAYS       ... the one-launch form beyond one round of workgroups as well
 *   SEGFAC_NO_FUSE_MAP           folded SegFormerHead map: streaming product + VALU upsample-add instead of fuse_map_kernel
 *   SEGFAC_NO_BWD248_MFMA        transposed 1/2-1/4-1/8 resizes on the VALU kernel instead of fuse_map_bwd_kernel
 *   SEGFAC_UPADD_GENERIC         upsample-add: the generic per-source kernels also for the 2-4-8 pyramid
 *   SEGFAC_NO_HEAD_FUSED         segf_bn_cls_bwd_supported answers 0: classifier data gradient and BatchNorm backward as separate passes
 *   SEGFAC_NO_HEAD_FUSED_DW      the folded head's stage-1 weight gradient does not ride on pass 2 of the fused BatchNorm backward
 *   SEGFAC_LOSS_NO_BAND          CE / Dice forward and backward on the tile kernels instead of the band sweep
 *   SEGFAC_LOSS_NO_BAND_FWD      ... the forward only
 *   SEGFAC_LOSS_BAND_ROWS        band sweep: rows per segment (0 = 16)
 *   SEGFAC_LOSS_NO_MFMA          ratio-4 loss kernels on the VALU cells form (fp32 storage)
 *   SEGFAC_LOSS_NO_RETRY         no exact per-pixel retry pass behind the flagged cells
 *   SEGFAC_LOSS_NO_LSE           the backward recomputes the softmax normalisation instead of taking the forward's per-pixel log-sum
 *   SEGFAC_NO_WIDE_FINALIZE      column-reduction finalize: one output per thread instead of four (bitwise the same sums)
 *   SEGFAC_NO_LN_PATCH           [host] MiT blocks: im2col / col2im around the spatial-reduction conv instead of the patch-major LayerNorm output
 *   SEGFAC_NO_SCALED_LN_BWD      [host] DropPath backward as its own scale_rows launch instead of riding on the LayerNorm backward
 *   SEGFAC_NO_DEFERRED_DW        [host] weight gradients issued layer by layer inside the captured step (no grouped launches)
 *   SEGFAC_NO_DEFERRED_FINALIZE  [host] LayerNorm dgamma / dbeta finalizes issued one by one
 *   SEGFAC_NO_HEAD_FUSED_CW      [host] the classifier's weight gradient as its own product instead of riding on pass 1 of the fused BatchNorm backward
 *   SEGFAC_NO_BWD248             [host] folded head backward: three segf_bilinear_bwd launches instead of the one-pass segf_bilinear_bwd_248
 *   SEGFAC_NO_GELU_GRN           [host] ConvNeXtV2 blocks: GELU as its own pass in front of the GRN kernels
 *   SEGFAC_NO_WEIGHT_SHADOW      [host] captured step: weights cast to bf16 per use instead of one cast of the flat buffer
 *   SEGFAC_NO_DERIVED_WEIGHTS    [host] captured step: derived weight layouts built in front of each layer instead of one grouped launch
 *
 * segf_policy_count / segf_policy_describe enumerate the table (field name, environment name, current value, default, description);
 * segf_policy_get / segf_policy_set read / change one value in the running process by field or environment name (INT_MIN: no such
 * switch; set returns the previous value); segf_policy_reload re-reads the environment (tests, A/B scripts).
 *
 * Launch trace: between segf_trace_begin(dry_run) and segf_trace_end(buf, cap) every kernel launched by the library ON THE CALLING
 * THREAD is recorded by its instantiated name (template arguments included), one per line in buf; segf_trace_end returns the number of
 * launches and closes the trace.  With dry_run != 0 the launches themselves (and the launch-error checks) are skipped: an entry point
 * can then be called with any non-null, 16-byte-aligned pointer values on a machine WITHOUT a GPU to learn which kernels a shape takes
 * -- tests/test_host_cpu.py::test_dispatch_of_baseline_shapes pins the BASELINE shapes against tests/golden/dispatch_table.json, so a
 * dispatch edit shows up as a diff. */
int segf_policy_count(void);
int segf_policy_describe(int i, const char** field, const char** env, int* value, int* def, const char** doc);
int segf_policy_get(const char* name);
int segf_policy_set(const char* name, int value);
void segf_policy_reload(void);
void segf_trace_begin(int dry_run);
int segf_trace_end(char* buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
