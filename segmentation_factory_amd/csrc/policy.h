// Dispatch policy of libsegfac_hip.so: every switch that can change WHICH kernel (or which form of a kernel) a call takes, in ONE
// table, read from the environment ONCE (first use) into one struct.  Nothing on the launch path calls getenv().
//
//   SEGF_POLICY_TABLE(X):  X(field, "ENVIRONMENT_NAME", default, "what a non-default value does")
//
// A switch that is set to anything but a number counts as 1; "0" is the same as unset for the on/off switches.  Tests and A/B
// scripts change a value in a running process with segf_policy_set() or -- after changing the environment -- segf_policy_reload()
// (include/segfac.h).  Rows marked [host] are read by the Python layer above the C ABI (segmentation_factory_amd/hip.py: policy()),
// listed here so that there is one place that names every switch.
//
// Switches whose experiment lost are NOT here any more (r05): the alternative split-K roundings, the 256-tile fast loads for weight
// gradients, the tunable short-K threshold / group size bounds / workgroup budgets, the K / tile order toggles of the eight-phase
// GEMM, the three-wave and dense-LDS variants of the head-dim-64 attention kernels, the grouped upsample-add, the VALU border ring
// of the transposed resize, the gather im2col / generic col2im, the cell-based arg-max, the four-round LayerNorm backward.
#pragma once

#define SEGF_POLICY_TABLE(X)                                                                                                             \
    /* ---- GEMM family (gemm.hip) ---- */                                                                                               \
    X(gemm_no_big, "SEGFAC_GEMM_NO_BIG", 0, "never take the 256 x 256-tile kernel (gemm_bf16_big_kernel): everything on the 128-tile one") \
    X(gemm_no_skinny, "SEGFAC_GEMM_NO_SKINNY", 0, "no streaming products for K, N <= 128 (gemm_skinny_kernel and its relatives)")        \
    X(gemm_no_skinny_rows, "SEGFAC_GEMM_NO_SKINNY_ROWS", 0, "the [tokens x 32] -> 768 projection on gemm_skinny_kernel instead of the whole-row form") \
    X(gemm_no_skinny_k, "SEGFAC_GEMM_NO_SKINNY_K", 0, "no K-split streaming product for 32- / 64-wide outputs (gemm_skinny_k_kernel)")  \
    X(gemm_no_dw_skinny, "SEGFAC_GEMM_NO_DW_SKINNY", 0, "small-output weight gradients on the tiled split-K kernel instead of gemm_dw_skinny_kernel") \
    X(gemm_no_narrow, "SEGFAC_GEMM_NO_NARROW", 0, "256-tile kernel: full 128 x 64 wave tiles also for outputs <= 160 wide (bit-identical results)") \
    X(gemm_no_deep, "SEGFAC_GEMM_NO_DEEP", 0, "256-tile forward: one K step of operand loads in flight instead of two")                  \
    X(gemm_no_deep128, "SEGFAC_GEMM_NO_DEEP128", 0, "128-tile kernel: one K step in flight instead of two")                              \
    X(gemm_no_fastload, "SEGFAC_GEMM_NO_FASTLOAD", 0, "guarded tile loads everywhere (no workgroup-uniform guard-free full-K-step loads)") \
    X(gemm_no_tr, "SEGFAC_GEMM_NO_TR", 0, "debugging: reduction-major fragments by scalar LDS reads instead of ds_read_b64_tr_b16; disables every kernel built on the transposed read") \
    X(gemm_no_pro, "SEGFAC_GEMM_NO_PRO", 0, "segf_gemm_pro_supported answers 0: BatchNorm + ReLU + Dropout2d are applied by their own pass") \
    X(gemm_no_fused_db, "SEGFAC_GEMM_NO_FUSED_DB", 0, "bias gradient as its own column-sum launch instead of riding on the weight-gradient product") \
    X(no_grouped_dw, "SEGFAC_NO_GROUPED_DW", 0, "segf_gemm_dw_db_grouped runs its members one by one")                                   \
    X(dw_no_xcd_slabs, "SEGFAC_DW_NO_XCD_SLABS", 0, "split-K weight gradients (128-tile kernel) in hardware workgroup order instead of one K slab per XCD")                \
    X(dw_no_shared_split, "SEGFAC_DW_NO_SHARED_SPLIT", 0, "grouped weight gradients keep their per-layer slice counts (also read by the host layer)") \
    X(no_wide_reduce, "SEGFAC_NO_WIDE_REDUCE", 0, "split-K partials of large outputs summed by the 16 x 16 form instead of whole rows")   \
    X(no_reduce4, "SEGFAC_NO_REDUCE4", 0, "split-K reduce: one output per thread instead of four (bitwise the same sums)")               \
    X(gemm_f32_no_mfma, "SEGFAC_GEMM_F32_NO_MFMA", 0, "fp32 storage (exact-parity mode, evaluate): products on the vector FMA kernel instead of the f32 matrix instruction") \
    X(gemm8_linear, "SEGFAC_GEMM8_LINEAR", 1, "0: plain nn.Linear products never take the eight-phase kernel (the 256 / 128 tile kernels as in r04)") \
    X(gemm8_linear_min_tiles, "SEGFAC_GEMM8_LINEAR_MIN_TILES", 128, "fewest 256 x 256 tiles for which a K >= 2048 nn.Linear product takes the eight-phase kernel (192 for shorter K)") \
    X(gemm8_linear_min_fill, "SEGFAC_GEMM8_LINEAR_MIN_FILL", 60, "smallest share (percent) of the launched 256 x 256 tiles that must be output for an nn.Linear product with ragged last tiles to take the eight-phase kernel") \
    X(gemm8_linear_min_k, "SEGFAC_GEMM8_LINEAR_MIN_K", 256, "shortest reduction for which an nn.Linear product takes the eight-phase kernel (its 12-load prologue and drain against K / 64 tiles)") \
    X(gemm8_dw, "SEGFAC_GEMM8_DW", 1, "0: nn.Linear weight gradients never take the eight-phase kernel below 65536 tokens (the grouped 128-tile kernel as in r04)") \
    X(gemm8_dw_min_gflop, "SEGFAC_GEMM8_DW_MIN_GFLOP", 100, "smallest nn.Linear weight gradient (GFLOP; both feature counts multiples of 256, >= 256 FLOP per operand byte) that takes the eight-phase kernel + a column-sum pass instead of the grouped 128-tile kernel") \
    X(gemm8_linear_min_gflop, "SEGFAC_GEMM8_LINEAR_MIN_GFLOP", 36, "smallest nn.Linear product (GFLOP, K >= 512; 100 for shorter K) that takes the eight-phase kernel") \
    /* ---- implicit-GEMM 3 x 3 convolution, eight-phase kernel, fp8 (gemm.hip, gemm8.hip, fp8.hip) ---- */                             \
    X(no_gemm8, "SEGFAC_NO_GEMM8", 0, "no eight-phase kernel at all (gemm8_kernel): the two-phase 256-tile kernel everywhere")           \
    X(no_gemm8t, "SEGFAC_NO_GEMM8T", 0, "no eight-phase kernel for weight gradients (reduction-major operands)")                         \
    X(conv_no_fwd_split, "SEGFAC_CONV_NO_FWD_SPLIT", 0, "3 x 3 forward / data gradient with few output tiles: no split over the (channel block, tap) walk") \
    X(g8_stagger, "SEGFAC_G8_STAGGER", -1, "eight-phase kernel: 1 = wave groups one barrier apart, 0 = lockstep, -1 = staggered for bf16 and the per-device choice (segf_gemm8_option) for fp8") \
    X(no_fp8, "SEGFAC_NO_FP8", 0, "segf_gemm_fp8_supported answers 0 (block-scaled 128-tile fp8 GEMM)")                                  \
    X(no_fp8_conv, "SEGFAC_NO_FP8_CONV", 0, "segf_conv3x3_fp8*_supported answer 0")                                                      \
    X(no_fp8_wgrad, "SEGFAC_NO_FP8_WGRAD", 0, "fp8 3 x 3 convolutions keep a bf16 weight gradient")                                      \
    X(no_fp8_linear, "SEGFAC_NO_FP8_LINEAR", 0, "segf_linear_fp8_supported answers 0")                                                   \
    /* ---- attention (attention.hip, attention_mfma.hip) ---- */                                                                        \
    X(attn_no_mfma, "SEGFAC_ATTN_NO_MFMA", 0, "attention on the VALU reference kernels (attention.hip) also in bf16")                    \
    X(attn_f32_no_mfma, "SEGFAC_ATTN_F32_NO_MFMA", 0, "fp32 attention forward on the vector kernel (one query per lane) instead of the f32 matrix instruction") \
    X(attn64_prescale, "SEGFAC_ATTN64_PRESCALE", 0, "head dim 64, >= 128 keys: scale log2(e) rides on the Q fragments (bf16(q c), one more rounding per q element) and -max / -lse are the score accumulators' initial values, instead of one multiply-add per score: forward + query-side backward, +8 % / +2 % per kernel, attention error x 1.2 - 2.3") \
    X(attn64_dkv_rows, "SEGFAC_ATTN64_DKV_ROWS", 128, "head dim 64, >= 128 keys, key-side backward: query rows per staged Q / dO tile and barrier (128, 64 or 32: the same arithmetic, bit for bit)") \
    X(attn_no_fused_bwd, "SEGFAC_ATTN_NO_FUSED_BWD", 0, "head dim 32, <= 256 keys: query-side + key-side backward kernels instead of the one-kernel backward") \
    /* ---- depthwise / patch convolutions (conv.hip) ---- */                                                                            \
    X(dw_no_walk, "SEGFAC_DW_NO_WALK", 0, "depthwise 3 x 3: the round-1 strip kernels instead of the vertical-walk kernels")             \
    X(dw_walk_rows, "SEGFAC_DW_WALK_ROWS", 0, "depthwise 3 x 3 walk: rows per segment (0 = chosen from the map size)")                   \
    X(dw_no_small, "SEGFAC_DW_NO_SMALL", 0, "depthwise 3 x 3 backward of small maps: three passes instead of the one-launch LDS form")   \
    X(dw_small_always, "SEGFAC_DW_SMALL_ALWAYS", 0, "... the one-launch form beyond one round of workgroups as well")                    \
    /* ---- decode-head kernels (fuse_map.hip, head_fused.hip, resize.hip) ---- */                                                       \
    X(no_fuse_map, "SEGFAC_NO_FUSE_MAP", 0, "folded SegFormerHead map: streaming product + VALU upsample-add instead of fuse_map_kernel") \
    X(no_bwd248_mfma, "SEGFAC_NO_BWD248_MFMA", 0, "transposed 1/2-1/4-1/8 resizes on the VALU kernel instead of fuse_map_bwd_kernel")    \
    X(upadd_generic, "SEGFAC_UPADD_GENERIC", 0, "upsample-add: the generic per-source kernels also for the 2-4-8 pyramid")               \
    X(no_head_fused, "SEGFAC_NO_HEAD_FUSED", 0, "segf_bn_cls_bwd_supported answers 0: classifier data gradient and BatchNorm backward as separate passes") \
    X(no_head_fused_dw, "SEGFAC_NO_HEAD_FUSED_DW", 0, "the folded head's stage-1 weight gradient does not ride on pass 2 of the fused BatchNorm backward") \
    /* ---- loss / metrics (loss.hip, loss_band.hip) ---- */                                                                             \
    X(loss_no_band, "SEGFAC_LOSS_NO_BAND", 0, "CE / Dice forward and backward on the tile kernels instead of the band sweep")            \
    X(loss_no_band_fwd, "SEGFAC_LOSS_NO_BAND_FWD", 0, "... the forward only")                                                             \
    X(loss_band_rows, "SEGFAC_LOSS_BAND_ROWS", 0, "band sweep: rows per segment (0 = 16)")                                               \
    X(loss_no_mfma, "SEGFAC_LOSS_NO_MFMA", 0, "ratio-4 loss kernels on the VALU cells form (fp32 storage)")                              \
    X(loss_no_retry, "SEGFAC_LOSS_NO_RETRY", 0, "no exact per-pixel retry pass behind the flagged cells")                                \
    X(loss_no_lse, "SEGFAC_LOSS_NO_LSE", 0, "the backward recomputes the softmax normalisation instead of taking the forward's per-pixel log-sum") \
    /* ---- reductions (colreduce.h) ---- */                                                                                             \
    X(no_wide_finalize, "SEGFAC_NO_WIDE_FINALIZE", 0, "column-reduction finalize: one output per thread instead of four (bitwise the same sums)") \
    /* ---- [host] switches read by the Python layer ---- */                                                                             \
    X(no_ln_patch, "SEGFAC_NO_LN_PATCH", 0, "[host] MiT blocks: im2col / col2im around the spatial-reduction conv instead of the patch-major LayerNorm output") \
    X(no_scaled_ln_bwd, "SEGFAC_NO_SCALED_LN_BWD", 0, "[host] DropPath backward as its own scale_rows launch instead of riding on the LayerNorm backward") \
    X(no_deferred_dw, "SEGFAC_NO_DEFERRED_DW", 0, "[host] weight gradients issued layer by layer inside the captured step (no grouped launches)") \
    X(no_deferred_finalize, "SEGFAC_NO_DEFERRED_FINALIZE", 0, "[host] LayerNorm dgamma / dbeta finalizes issued one by one")              \
    X(no_head_fused_cw, "SEGFAC_NO_HEAD_FUSED_CW", 0, "[host] the classifier's weight gradient as its own product instead of riding on pass 1 of the fused BatchNorm backward") \
    X(no_bwd248, "SEGFAC_NO_BWD248", 0, "[host] folded head backward: three segf_bilinear_bwd launches instead of the one-pass segf_bilinear_bwd_248") \
    X(no_gelu_grn, "SEGFAC_NO_GELU_GRN", 0, "[host] ConvNeXtV2 blocks: GELU as its own pass in front of the GRN kernels")                \
    X(no_weight_shadow, "SEGFAC_NO_WEIGHT_SHADOW", 0, "[host] captured step: weights cast to bf16 per use instead of one cast of the flat buffer") \
    X(no_derived_weights, "SEGFAC_NO_DERIVED_WEIGHTS", 0, "[host] captured step: derived weight layouts built in front of each layer instead of one grouped launch")

struct SegfPolicy {
#define SEGF_POLICY_FIELD(field, env, def, doc) int field;
    SEGF_POLICY_TABLE(SEGF_POLICY_FIELD)
#undef SEGF_POLICY_FIELD
    int g8_stagger_fp8;          // per-device choice for the fp8 eight-phase kernel (segf_gemm8_option / hip.autotune_gemm8_fp8); not an environment switch
};

const SegfPolicy& segf_policy();          // policy.hip: parsed from the environment at first use
#define POL(field) (segf_policy().field)

// ---- launch trace (policy.hip): which kernels did a C-ABI call launch?  segf_trace_begin / segf_trace_end, include/segfac.h --------------
struct SegfTrace {
    int on;                      // record kernel names
    int dry;                     // ... and do not launch them (dispatch decisions without a GPU: nothing is dereferenced on the host)
    int n;                       // launches since segf_trace_begin
    const char* names[256];      // the kernel as the launch site spells it: "(gemm_bf16_big_kernel<0, bf16_t, false, true, 1, true>)"
    const char* where[256];      // __PRETTY_FUNCTION__ of the launching host function: carries the template arguments ("[L = 0, KS = 2]")
                                 // when the launch site spells the kernel with the template parameters of its enclosing function
};
SegfTrace& segf_trace();
void segf_trace_note(const char* kernel_text, const char* where);
