/* ucod_dpl.h -- C ABI of libucod_dpl.so: the MI355X (gfx950) hot path of UCOD-DPL.
 *
 * The reference (Heartfirey/UCOD-DPL) is pure Python/PyTorch: it has no FFI of its own.  The
 * boundary this library replaces is therefore the set of ATen / HuggingFace op call sites on the
 * path BASELINE.json:north_star names; every entry point below cites the reference lines whose
 * device arithmetic it takes over (paths relative to the reference repo root).  The Python host
 * side (ucod_dpl_amd/) binds these with ctypes and re-exposes the reference's module interface
 * (models/uscod.py::baseline, models/discriminator.py::Discriminator,
 * data/utils/feature_extractor.py::backbone, engine/runner/loop_UCOD_DPL.py::TrainLoop).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; nothing is allocated inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous,
 *     stream-ordered and re-entrant; workspace sizes come from the *_workspace_bytes helpers;
 *   - return value: 0 on success, UCOD_EINVAL (-1) for rejected arguments, otherwise a hipError_t;
 *   - "bf16" buffers are raw uint16 bfloat16; "f32" are IEEE float.
 *   - tensors are dense row-major; [B,C,H,W] is NCHW exactly as the reference holds them.
 *
 * Environment switches read by the PRODUCT library -- three, all of them change which (deterministic) summation order a GEMM launch takes, none is a
 * measurement knob; they are read once per process (ucod_gemm_reload_tuning re-reads them), never on the launch path:
 *   UCOD_GEMM_NO_PATCH=1       ucod_gemm_bf16*: every output through the tile path.  By default a launch that ends a few tiles past
 *                              one or two whole rounds of CUs computes those tiles as 16x32 patches on the side, which sums K in a
 *                              different order: results stay deterministic but a row's low f32 bits then depend on where the row
 *                              sits in the batch.  Set this for bitwise batch-position independence.
 *   UCOD_GEMM_PATCH_ROUNDS=n   largest number of whole rounds for which the patch mode is considered (default 2).
 *   UCOD_GEMM_NO_MIXED=1       no mixed-height launches (the one-shot large tile with a partly filled last round instead).
 * Measurement knobs (alternative kernel forms, grid sizes, store policies: UCOD_RESIZE_ELEMENTWISE, UCOD_LN_*, UCOD_STATS_STRIPS, UCOD_RESID16_NT,
 * UCOD_LN_FOLD_NO_PARTIALS, UCOD_GEMM_GROUP_M / _COL_FAST / _ST_AUX, UCOD_LORA_GRAD_*, UCOD_DGRAD_F32) exist only in builds made with
 * -DUCOD_LAB_KNOBS (`make -C ucod_dpl_amd/csrc knobs`, used by tools/); libucod_dpl.so / libucod_dpl_f16.so ignore them (csrc/common.h: lab_env).
 */
#ifndef UCOD_DPL_H
#define UCOD_DPL_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): ucod_vit_desc gained resid16 (shifts ucod_vit_train_desc), epilogues 9 / 10, ucod_resid16_overflow_*, ucod_gemm_reload_tuning;
 * the experiment variants of ucod_gemm_bf16 / ucod_attention_fwd left the product library.  native.load() refuses any other version. */
/* 3 (round 4): ucod_disc_params gained `nbt`; ucod_step_loss, ucod_disc_bce, the feature-branch discriminator's backward entry points and
 * the assembly attention variants (ucod_attention_fwd variant 64 / 32 / 5) were added. */
/* 4 (round 5): LayerNorm folded into its consumer GEMMs -- epilogues 11 - 14, ucod_gemm_lnfold, ucod_gemm_bf16_stats, ucod_cls_rows_h16_stats,
 * ucod_row_stats_h16, ucod_vit_desc.ln_fold, UCOD_VIT_LAYER_STRIDE 14 -> 16 (two column-sum slots per layer); ucod_zero_segments,
 * ucod_accumulators_prezeroed, the *_multi forms of the Look-Twice crop / paste; the assembly attention variants 64 / 32 of ucod_attention_fwd and the
 * UCOD_ATTN_ASM switch LEFT the product library (laboratory: ucod_attention_fwd_asm_lab of libucod_dpl_variants.so). */
/* 5 (round 6): the row partials of the LayerNorm fold hold (sum, M2 about the slot mean) and are merged with Chan's formula (was: raw sums of squares);
 * the split-operand (f32-equivalent) backbone pass: ucod_split_rows, ucod_layernorm_split, ucod_patch_im2col_split, ucod_qkv_split,
 * ucod_attention_split_fwd, ucod_vit_forward_split (+ their size helpers); ucod_clock_probe; the measurement knobs left the product build (UCOD_LAB_KNOBS). */
#define UCOD_ABI_VERSION 5
int ucod_abi_version(void);
/* 1 when a gfx950 device is visible to this process (hipGetDeviceProperties().gcnArchName) */
int ucod_device_is_gfx950(void);
/* 16-bit operand type this library was built for: "bf16" (libucod_dpl.so, BASELINE configs[1]) or "f16" (libucod_dpl_f16.so, built with
 * -DUCOD_HALF_F16: IEEE fp16 GEMM / attention operands, the arithmetic of the reference's fp16-autocast launcher,
 * scripts/launch_train_first_stage.sh; every "bf16" in the names and comments below then reads fp16; the backbone-backward entry points
 * (ucod_vit_forward_train, ucod_vit_backward, ucod_gemm_bf16_train, ucod_layernorm_*lora*, ucod_lora_*, ucod_attention_bwd, ...)
 * return UCOD_EINVAL there) */
const char* ucod_half_name(void);

/* Optional per-op timing with HIP events recorded on the launch stream around every launcher below (off by default;
 * bench.py turns it on over its timed region to price each kernel class against its roofline).
 * ucod_prof_collect synchronises the recorded events, fills total_ms[ncls] / count[ncls] and clears the log. */
int ucod_prof_enable(int on);
int ucod_prof_num_classes(void);
const char* ucod_prof_class_name(int cls);
int ucod_prof_collect(double* total_ms_host, long long* count_host);
/* One single-wave launch that stores (s_memtime, s_memrealtime) = (shader-clock cycle counter, constant 100 MHz counter) of the CU it runs on into
 * out_dev[0..1] (two uint64).  Two probes on the same stream bracket a region: (memtime_1 - memtime_0) / (memrealtime_1 - memrealtime_0) * 100 MHz is the
 * mean shader clock the chip HELD over the region (bench.py --sustain-s: the roofline fraction against MFMA rate x held clock). */
int ucod_clock_probe(unsigned long long* out_dev, void* stream);

/* ------------------------------------------------------------------ ViT backbone (rows B1-B8) */

/* GEMM epilogues: C[m][n] = sum_k A[m][k]*B[n][k], A:[M,K] B:[N,K] bf16, K contiguous (y = x W^T) */
enum {
  UCOD_EPI_BIAS_BF16 = 0,            /* out bf16[M,N] = C + bias[n]            QKV: modeling_dinov2.py:199-212, dino.py:103,110 */
  UCOD_EPI_BIAS_GELU_BF16 = 1,       /* out bf16[M,N] = gelu_erf(C + bias[n])  MLP fc1: modeling_dinov2.py:281-297, dino.py:77-93 */
  UCOD_EPI_BIAS_SCALE_RESID_F32 = 2, /* out f32[M,N] = resid + scale[n]*(C+bias[n])  out-proj/fc2 + LayerScale + residual:
                                        modeling_dinov2.py:238-253,272-278,361-381 (scale = ones for DINOv1, dino.py:139-140) */
  UCOD_EPI_PATCH_TOKENS_F32 = 3,     /* A = im2col patches [B*(tok-1),K]; out f32[B*tok,N] rows b*tok+1+p = C + bias[n] + pos[1+p][n]
                                        patch-embed conv + position embedding: modeling_dinov2.py:97-116,139-149, dino.py:144-159,223-235 */
  UCOD_EPI_KEY_NCHW_F32 = 4,         /* A = W_key [C,K], B = tokens [Bimg*tok,K]; out f32[Bimg,C,tok-1] = C + bias[m], CLS dropped:
                                        the key hook, data/utils/feature_extractor.py:42,46-47,55-58 */
  UCOD_EPI_BIAS_F32 = 5,             /* out f32[M,N] = C + bias[n] (final LayerNorm consumers / tests; dgrad GEMMs with bias NULL) */
  /* backbone-backward mode (row B9), ucod_gemm_bf16_train only: */
  UCOD_EPI_GELU_BWD_BF16 = 6,        /* out bf16[M,N] = C * gelu'(aux[m][n]), aux = saved fc1 pre-activation (fc2 dgrad) */
  UCOD_EPI_BIAS_GELU_SAVE_BF16 = 7,  /* out bf16 = gelu_erf(C + bias[n]) and out2 bf16 = C + bias[n] (training-mode fc1) */
  UCOD_EPI_BIAS_SCALE_RESID_H16 = 9, /* as UCOD_EPI_BIAS_SCALE_RESID_F32 with the residual stream kept in IEEE fp16: out f16[M,N] = sat(resid f16 +
                                        scale[n]*(C+bias[n])) (ucod_vit_desc.resid16; `resid` / `out` point at f16 rows).  sat = clamp to
                                        +-65504 (never inf); every clamp is counted: ucod_resid16_overflow_fetch.  Any shape. */
  UCOD_EPI_PATCH_TOKENS_H16 = 10,    /* as UCOD_EPI_PATCH_TOKENS_F32 with f16 token rows (same saturation) */
  UCOD_EPI_LNFOLD_BIAS_BF16 = 11,    /* ucod_gemm_lnfold only.  LayerNorm folded into the QKV projection (modeling_dinov2.py:348-353,199-212: norm1 -> query / key /
                                        value): A = the fp16 residual stream x itself, B = fp16(gamma (.) W), colsum[n] = sum_k B[n][k], bias = W beta + b,
                                        stats[m] = (rstd, -mean * rstd) of row m:  out 16-bit [M,N] = (stats[m][0] * C + stats[m][1] * colsum[n] + bias[n]) * scale[n] */
  UCOD_EPI_LNFOLD_GELU_BF16 = 12,    /* ucod_gemm_lnfold only.  The same fold for norm2 -> fc1 -> GELU (modeling_dinov2.py:365-373,281-297):
                                        out = gelu_erf(stats[m][0] * C + stats[m][1] * colsum[n] + bias[n]) */
  UCOD_EPI_BIAS_SCALE_RESID_H16_STATS = 13, /* ucod_gemm_bf16_stats only.  UCOD_EPI_BIAS_SCALE_RESID_H16 that also leaves, per output row and 64-column slot, the
                                        (sum S, M2 = sum of squared deviations from the slot's own mean S / 64) of the fp16 values it has just written:
                                        row_partials f32 [M][N/64][2].  The next ucod_gemm_lnfold merges a row's slots in its prologue (Chan's parallel-variance
                                        formula: nothing cancels, whatever |mean| / sigma) instead of reading `stats`: no statistics launch.  Large passes only
                                        (M >= 2048, N % 64 == 0); otherwise UCOD_EINVAL and nothing is launched */
  UCOD_EPI_PATCH_TOKENS_H16_STATS = 14,     /* ucod_gemm_bf16_stats only.  UCOD_EPI_PATCH_TOKENS_H16 with the same partials, indexed by OUTPUT token row (the CLS rows'
                                        partials come from ucod_cls_rows_h16_stats) */
  UCOD_EPI_BIAS_GELU_SPLIT2 = 15,    /* split-operand pass, two terms (csrc/split.hip): out bf16 [M, 3 N] = the A-side split operand (segments hi | hi | lo) of
                                        gelu_erf(C + bias[n]) -- fc1 + GELU + the split in ONE launch instead of UCOD_EPI_BIAS_F32 + ucod_split_rows(op 1): no f32
                                        round trip of the MLP hidden.  bf16 library; N % 8 == 0 */
  UCOD_EPI_QKV_FP8 = 8               /* QKV projection of the fp8 attention path (BASELINE configs[4]): out = e4m3 bytes
                                        [3 (q|k|v)][Bimg*heads][Npad][64], Npad = tokens rounded up to 64, value = clamp((C + bias[n]) *
                                        scale[n], +-448); N = 3*heads*64, M = Bimg*tokens_per_image; large-tile kernel only */
};
/* variant: 0 = auto, 1 = 128x128 register staging, 2 = 128x128 LDS-DMA, 12 = 64x64 tile with LDS-DMA (what auto picks when there are
 * fewer 128x128 tiles than CUs: batch-1 passes), 9 / 10 = 256x256 / 256x192 large tile (LDS-DMA in flight across barriers, staggered
 * wave groups, two barrier intervals per K-tile, leftover tiles as patches: what auto picks for one- and two-round launches),
 * 13 / 14 = the same loop with mixed-height tiles (whole rounds: what auto picks for three and more rounds).  3-8 (four intervals,
 * no stagger, persistent form) are laboratory variants: ucod_gemm_bf16_lab of libucod_dpl_variants.so (`make variants`), refused here.
 * Large-tile variants need N % 4 == 0 (N % 8 == 0 for bf16 output).  K % 64 == 0.  For UCOD_EPI_BIAS_BF16 a non-NULL `scale` [N] multiplies
 * (C + bias) per column before the bf16 rounding (used to fold the softmax scale into Q). */
int ucod_gemm_bf16(int epilogue, const void* A_bf16, const void* B_bf16, void* out, int M, int N, int K,
                   const float* bias, const float* scale, const float* resid, const float* pos,
                   int tokens_per_image, int variant, void* stream);
/* LayerNorm folded into its consumer GEMM (epilogues UCOD_EPI_LNFOLD_*; libucod_dpl_f16.so only -- an MFMA takes both operands in one type and
 * the A operand here is the IEEE fp16 residual stream itself; the bf16 build returns UCOD_EINVAL).  Replaces nn.LayerNorm + nn.Linear of
 * modeling_dinov2.py:348-381 (norm1 -> attention, norm2 -> mlp) and dino.py:127-131:
 *   LN(x) W^T + b = rstd[m] * (x W'^T - mean[m] * colsum[n]) + bias'[n],   W' = fp16(gamma (.) W),  colsum[n] = sum_k W'[n][k] (of the ROUNDED
 *   W', so that the subtraction cancels what the MFMA summed),  bias' = W beta + b.
 * x_f16 [M,K] fp16, w_folded [N,K] fp16, colsum / bias_folded f32 [N], stats f32 [M][2] = (rstd, -mean * rstd) per row (ucod_row_stats_h16),
 * scale: optional column scale (UCOD_EPI_LNFOLD_BIAS_BF16 only), out fp16 [M,N].  N % 8 == 0, K % 64 == 0.  variant as ucod_gemm_bf16
 * (the 192-wide forms 10 / 14 are taken as 9 / 13). */
int ucod_gemm_lnfold(int epilogue, const void* x_f16, const void* w_folded, void* out, int M, int N, int K, const float* bias_folded,
                     const float* colsum, const float* stats, const float* row_partials, int nslot, float eps, const float* scale,
                     int variant, void* stream);
/* `stats` may be NULL when `row_partials` f32 [M][nslot][2] is given (nslot = K / 64, even, <= 24): per 64-column slot (S_i, M2_i) = (sum, sum of squared
 * deviations from S_i / 64) of x's row, left by the producer of x (ucod_gemm_bf16_stats / ucod_cls_rows_h16_stats); the kernel's prologue forms
 *   mean = sum_i S_i / K,  var = (sum_i M2_i + 64 sum_i (S_i / 64 - mean)^2) / K,  rstd = rsqrt(var + eps)   in f32
 * (ABI 5; ABI 4 stored raw sums of squares and formed E[x^2] - mean^2, which lost the variance of rows with |mean| >> sigma).  Same value as
 * ucod_row_stats_h16's two-pass form to f32 rounding.  The large-tile kernels only (a small shape is then run on them too). */
/* The producers of the fp16 residual stream with row partials (epilogues UCOD_EPI_*_STATS; arguments as ucod_gemm_bf16; `resid` / `out` f16 rows;
 * nslot = N / 64).  UCOD_EINVAL (nothing launched) for shapes the large-tile kernels do not take: use the plain epilogue and ucod_row_stats_h16 then. */
int ucod_gemm_bf16_stats(int epilogue, const void* A_bf16, const void* B_bf16, void* out, int M, int N, int K, const float* bias, const float* scale,
                         const void* resid_f16, const float* pos, int tokens_per_image, float* row_partials, int nslot, void* stream);
/* ucod_cls_rows_h16 that also writes the CLS rows' partials, slot by slot like the other rows' (nslot = D / 64) */
int ucod_cls_rows_h16_stats(void* x_f16, const float* cls, const float* pos, float* row_partials, int nslot, int B, int tok, int D, void* stream);
/* Row statistics of the fp16 residual stream for the folded epilogues: stats[m] = (rstd, -mean * rstd), two-pass in f32 over the row held in
 * registers, biased variance + eps like nn.LayerNorm.  x f16 [rows,D], D % 256 == 0, D <= 1536.
 * Range of the fold: x W'^T and mean * colsum cancel in f32, which costs ~2^-24 sqrt(K) |mean| / sigma of the output scale (0.2 fp16 ulp at 100 sigma).  Both
 * statistics paths (this kernel and the prologue of ucod_gemm_lnfold on row partials) COUNT every row with |mean| > 256 sigma into the saturation counter of the
 * fp16 stream (ucod_resid16_overflow_*): such a checkpoint is reported by ViTEngine.check_overflow, never folded silently; use ln_fold = 0 for it. */
int ucod_row_stats_h16(const void* x_f16, float* stats, int rows, int D, float eps, void* stream);
/* The UCOD_GEMM_* tuning variables (csrc/gemm_bf16_plan.h) are read once per process; this re-reads them (tests, sweep tools). */
void ucod_gemm_reload_tuning(void);

/* Saturation counter of the f16 residual stream (one word per device): how many wave-lanes had to clamp a value of x to +-65504 since the
 * last reset.  fetch = asynchronous 4-byte copy into (pinned) host memory on `stream`; the caller orders its own read behind it (event /
 * stream sync).  A non-zero count means the f16 stream cannot hold this checkpoint's activations: use resid16 = 0 (ViTEngine(resid="f32")). */
int ucod_resid16_overflow_fetch(unsigned* host_dst, void* stream);
int ucod_resid16_overflow_reset(void* stream);
/* A counter of the caller's own (round 4): the launches issued NEXT FROM THIS HOST THREAD count into `device_counter` (one zero-initialised
 * device word, e.g. one per engine) instead of the per-device word, and fetch / reset act on it; NULL restores the per-device word.  The
 * binding is host-side state read at launch time: bind, issue the pass, unbind. */
int ucod_resid16_overflow_bind(unsigned* device_counter);

/* nn.LayerNorm over the last dim (modeling_dinov2.py:348,353,365,373,441; dino.py:127,131,184):
 * x f32 [rows,D] -> y bf16 [rows,D] (or f32 when out_f32 != 0).  D % 128 == 0. */
int ucod_layernorm(const float* x, const float* gamma, const float* beta, void* y, int rows, int D, float eps,
                   int out_f32, void* stream);
/* the same with the residual stream in IEEE fp16 (ucod_vit_desc.resid16): x f16 [rows,D] -> y bf16 [rows,D]; statistics in f32 */
int ucod_layernorm_h16(const void* x_f16, const float* gamma, const float* beta, void* y, int rows, int D, float eps, void* stream);

/* softmax(Q K^T * scale) V per (image, head), head_dim 64 (modeling_dinov2.py:153-179; dino.py:113-117).
 * qkv bf16 [B*tok, 3*heads*64] rows = [q | k | v], heads contiguous; out bf16 [B*tok, heads*64].
 * scale != 0: the generic kernel (Q as the reference holds it, the scale applied inside the softmax, running max per tile).
 * scale == 0 declares that Q already carries head_dim^-0.5 * log2(e) (ucod_fill_qscale + the QKV epilogue scale do that inside
 * ucod_vit_forward) and selects the product kernel: K/V by buffer loads to LDS, -m as the score accumulator's initial value, deferred
 * max, probabilities fed back as MFMA operands from registers, f32 row sums, 16-byte output stores; query rows and 32-key blocks past
 * the last token are not computed.  variant: 0 or 2 (the same kernels), 5 / 66 = attn_fwd_v5_kernel / attn_fwd_v6_kernel by name; every other number is a laboratory variant
 * (ucod_attention_fwd_lab of libucod_dpl_variants.so) and is refused.  tok * heads * 384 must fit 32 bits. */
int ucod_attention_fwd(const void* qkv_bf16, void* out_bf16, int B, int tok, int heads, float scale, int variant,
                       void* stream);

/* The same attention with Q, K, V and the probabilities quantised to OCP e4m3 and both products on the block-scaled CDNA4 matrix
 * instruction v_mfma_scale_f32_32x32x64_f8f6f4 (BASELINE.json configs[4], the "fp8 attention path"; same reference lines as above).
 * qkv is the QKV projection's 16-bit output with Q pre-scaled by head_dim^-0.5 * log2(e) (as for scale == 0 above).  A first kernel
 * writes Q8 / K8 [B*heads][Npad][64] and the per-tile transposed Vt8 into `workspace` (ucod_attention_fp8_workspace_bytes), each
 * tensor multiplied by 2^q_exp / 2^k_exp / 2^v_exp before rounding (clamped to +-448); the inverse powers of two ride on the
 * instruction's E8M0 block scales.  ucod_vit_forward takes this path for attn_variant == 8 with (q_exp, k_exp, v_exp) = (5, 3, 3).
 * Tolerance (tests/test_gpu_fp8_attention.py): exact on e4m3-representable inputs; relative L2 <= 1e-1 against the f32 softmax
 * attention on Gaussian inputs (7.2e-2 measured: the error of a 64-term e4m3 score in the exponent). */
size_t ucod_attention_fp8_workspace_bytes(int B, int tok, int heads);
/* Fused form: `q8k8v8` is what ucod_gemm_bf16(UCOD_EPI_QKV_FP8, ...) wrote (column scales 2^q_exp * head_dim^-0.5 * log2 e, 2^k_exp,
 * 2^v_exp folded into its `scale` vector), V row-major like K and transposed on the fly by ds_read_b64_tr_b8: no conversion pass and
 * half the QKV output bytes.  ucod_attention_fp8_zero_pad clears the rows tokens..Npad-1 of all three tensors (once per workspace:
 * nothing writes them afterwards; a NaN byte there would survive the multiplication by a zero probability). */
int ucod_attention_fp8_zero_pad(void* q8k8v8, int B, int tok, int heads, void* stream);
int ucod_attention_fwd_fp8_fused(const void* q8k8v8, void* out_bf16, int B, int tok, int heads, int q_exp, int k_exp, int v_exp, void* stream);
int ucod_attention_fwd_fp8(const void* qkv_bf16, void* out_bf16, void* workspace, size_t workspace_bytes, int B, int tok, int heads,
                           int q_exp, int k_exp, int v_exp, void* stream);

/* patch gather: img f32 [B,C,H,W] -> bf16 rows [B*(H/P)*(W/P), Kpad], k = c*P*P + py*P + px, zero padded
 * (the im2col view of the stride-P conv, modeling_dinov2.py:139-149 / dino.py:154-158).  Kpad % 64 == 0. */
int ucod_patch_im2col(const float* img, void* patches_bf16, int B, int C, int H, int W, int P, int Kpad, void* stream);

/* x f32 [B*tok, D]: row b*tok = cls + pos[0]  (modeling_dinov2.py:107-112; dino.py:227-232) */
int ucod_cls_rows(float* x, const float* cls, const float* pos, int B, int tok, int D, void* stream);
int ucod_cls_rows_h16(void* x_f16, const float* cls, const float* pos, int B, int tok, int D, void* stream);

/* v[0..D) = c, v[D..3D) = 1: the per-column factor of the fused QKV epilogue */
int ucod_fill_qscale(float* v, int D, float c, void* stream);
/* v[3*D] = cq for the q columns, ck for k, cv for v: the column scales of UCOD_EPI_QKV_FP8 (pre-scale and power-of-two range scales) */
int ucod_fill_qscale3(float* v, int D, float cq, float ck, float cv, void* stream);

/* f32 -> bf16 cast of n elements (weight preparation) */
int ucod_cast_f32_bf16(const float* src, void* dst_bf16, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Backbone-backward mode.  The reference describes it in models/modules/full_model.py:47-72,79-126
 * (peft LoRA r=2, lora_alpha=4 on query/key/value of every encoder layer, all other weights frozen, key hook of the
 * last layer feeds the decoder); that file is not importable, so parity is pinned on HuggingFace Dinov2Model autograd
 * with the LoRA forward restated (tests/golden/g12_lora_backbone.npz).  LoRA rides on the GEMMs as UCOD_LORA_AUG
 * extra K columns -- see ucod_dpl_amd/csrc/vit_train.hip for the operand formats.
 * One layer's LoRA parameters / gradients (f32): [A_q (r x D) | B_q (D x r) | A_k | B_k | A_v | B_v], 3r <= UCOD_LORA_AUG. */
#define UCOD_LORA_AUG 64

/* ucod_gemm_bf16 with the two training epilogues (large-tile kernels; N % 8 == 0, K >= 128).  For UCOD_EPI_BIAS_BF16 and
 * UCOD_EPI_BIAS_F32 (either entry point) a NULL bias means a plain product (K >= 128, N % 4 == 0, variant not 1/2). */
int ucod_gemm_bf16_train(int epilogue, const void* A, const void* B, void* out, int M, int N, int K, const float* bias,
                         const void* aux_bf16, void* out2_bf16, int variant, void* stream);

/* LoRA dropout (LoraConfig.lora_dropout, full_model.py:50: nn.Dropout on the input of every lora_A, an independent mask per target
 * module).  Counter-based (ABI 3: one mix per element for the three projections): with
 *   h = seed_lo ^ (idx * 0x9E3779B1);  h ^= seed_hi + layer * 0x85EBCA77;  h ^= h >> 16;  h *= 0x7FEB352D;  h ^= h >> 15;
 *   h *= 0x846CA68B;  h ^= h >> 16;      (32-bit wrap-around arithmetic, idx = row * D + col)
 * element (row, col) of projection p (0 q, 1 k, 2 v) in `layer` is dropped iff  (h >> (10 * p)) & 1023  <  T,  T = floor(p_drop * 1024);
 * kept elements are scaled by 1 / (1 - T / 1024) (the effective drop probability is T / 1024).  Forward and backward regenerate the
 * mask; nothing is stored.
 * Pass NULL (or p == 0) for no dropout. */
typedef struct {
  float p;
  unsigned long long seed;   /* change it every step */
  int layer;
} ucod_lora_dropout;

/* y_aug bf16 [rows, D+64] = [ LayerNorm(x) | dropout_q(LN(x)) A_q^T, dropout_k(LN(x)) A_k^T, dropout_v(LN(x)) A_v^T (3r values) | 0 ] */
int ucod_layernorm_lora(const float* x, const float* gamma, const float* beta, const float* lora_layer, int r, void* y_aug_bf16,
                        int rows, int D, float eps, const ucod_lora_dropout* dropout, void* stream);
/* the same from an IEEE fp16 residual stream (the no-grad LoRA pass with vit.resid16) */
int ucod_layernorm_lora_h16(const void* x_f16, const float* gamma, const float* beta, const float* lora_layer, int r, void* y_aug_bf16,
                            int rows, int D, float eps, const ucod_lora_dropout* dropout, void* stream);

/* LayerNorm backward w.r.t. its input (frozen gamma/beta), fused with the residual add and the next GEMM's A operand:
 * dx f32 [rows,D] = dres (nullable) + dLN(dy; x, gamma);  s bf16 [rows,D] = next_scale (nullable: ones) * dx.
 * dx may alias dres or dy; either output may be NULL. */
int ucod_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* dres, const float* next_scale, float* dx,
                       void* s_bf16, int rows, int D, float eps, void* stream);
/* The same with the dropout-masked LoRA branch added to dy first:  dy += sum_p mask_p/(1-p) * (t_p A_p),  t = aug columns of
 * dqkv_aug [rows, 3D+64].  Used for LayerNorm-1 when dropout is on (the A^T columns of WqkvT_aug are then zero, ucod_lora_pack). */
int ucod_layernorm_bwd_lora(const float* dy, const float* x, const float* gamma, const float* dres, const float* next_scale, float* dx,
                            void* s_bf16, int rows, int D, float eps, const void* dqkv_aug_bf16, const float* lora_layer, int r,
                            const ucod_lora_dropout* dropout, void* stream);
/* Both with 16-bit inputs (round 4): flags & UCOD_LNB_DY_BF16: dy is bf16 [rows,D] (what ucod_vit_backward's dgrad GEMMs write: half the bytes on the
 * GEMM's store-bound drain and on this kernel's read side; the upstream gradient is rounded to 8 mantissa bits once, the residual cotangent
 * stream dx stays f32); flags & UCOD_LNB_X_F16 (only together with DY_BF16): x is IEEE fp16 (the saved residual stream of a training pass
 * with vit.resid16).  flags = 0: the two functions above. */
#define UCOD_LNB_DY_BF16 1
#define UCOD_LNB_X_F16 2
int ucod_layernorm_bwd_ex(const void* dy, const void* x, int flags, const float* gamma, const float* dres, const float* next_scale, float* dx,
                          void* s_bf16, int rows, int D, float eps, void* stream);
int ucod_layernorm_bwd_lora_ex(const void* dy, const void* x, int flags, const float* gamma, const float* dres, const float* next_scale, float* dx,
                               void* s_bf16, int rows, int D, float eps, const void* dqkv_aug_bf16, const float* lora_layer, int r,
                               const ucod_lora_dropout* dropout, void* stream);

/* Attention forward that also returns the base-2 log-sum-exp of the scaled scores, lse f32 [B, heads, tok]
 * (Q must carry head_dim^-0.5 * log2(e), as for ucod_attention_fwd with scale == 0). */
int ucod_attention_fwd_lse(const void* qkv_bf16, void* out_bf16, float* lse, int B, int tok, int heads, void* stream);

/* Attention backward (eager_attention_forward of modeling_dinov2.py:153-179 differentiated): qkv as in the forward
 * (Q pre-scaled), out/dout bf16 [B*tok, D], lse from ucod_attention_fwd_lse; delta f32 [B, heads, tok] is scratch.
 * dqkv bf16 rows of length ld_dqkv (>= 3D) = gradient w.r.t. the UNSCALED q | k | v projections. */
int ucod_attention_bwd(const void* qkv_bf16, const void* out_bf16, const void* dout_bf16, const float* lse, float* delta,
                       void* dqkv_bf16, int ld_dqkv, int B, int tok, int heads, void* stream);

/* dkey f32 [B, D, tok-1] (cotangent of the key hook) -> rows of dqkv_aug bf16 [B*tok, 3D+64]: k third = dkey^T (CLS row 0),
 * q and v thirds and the aug columns zero. */
int ucod_key_grad_tokens(const float* dkey, void* dqkv_aug_bf16, int B, int tok, int D, void* stream);

/* Fill the aug columns of Wqkv_aug bf16 [3D, D+64] (alpha/r * B_q|B_k|B_v on the block diagonal) and of WqkvT_aug bf16
 * [D, 3D+64] (A_q^T|A_k^T|A_v^T; zeros when zero_a_columns != 0, the dropout case) from one layer's LoRA parameters.  Either
 * matrix may be NULL. */
int ucod_lora_pack(const float* lora_layer, int r, float scaling, void* w_aug_bf16, void* wt_aug_bf16, int D, int zero_a_columns,
                   void* stream);

/* One layer's LoRA gradients from dqkv_aug [rows, 3D+64] and h_aug [rows, D+64]; also writes t = alpha/r * dqkv B into the
 * aug columns of dqkv_aug (consumed by the dgrad GEMM).  grad_layer has the parameter layout; accumulate != 0 adds. */
size_t ucod_lora_grad_workspace_bytes(int D);
int ucod_lora_grad(void* dqkv_aug_bf16, const void* h_aug_bf16, const float* lora_layer, int r, float scaling, float* grad_layer,
                   int accumulate, void* workspace, size_t workspace_bytes, int rows, int D, const ucod_lora_dropout* dropout, void* stream);

/* Whole frozen backbone forward up to the last layer's key projection: one call enqueues every kernel.
 * Pointer table (HOST array of DEVICE pointers; "w" entries are bf16 [out,in], the rest f32):
 *   [0] patch_w bf16 [D,Kpad] (zero padded k)  [1] patch_b [D]  [2] cls [D]  [3] pos [tok,D] (already interpolated)
 *   layer l at base = 4 + UCOD_VIT_LAYER_STRIDE*l:
 *     +0 ln1_g  +1 ln1_b  +2 qkv_w bf16 [3D,D] (rows q|k|v)  +3 qkv_b [3D]  +4 proj_w bf16 [D,D]  +5 proj_b
 *     +6 ls1 [D] (LayerScale lambda1; ones for DINOv1)  +7 ln2_g  +8 ln2_b  +9 fc1_w bf16 [F,D]  +10 fc1_b [F]
 *     +11 fc2_w bf16 [D,F]  +12 fc2_b [D]  +13 ls2 [D]
 *     +14 qkv_colsum [3D]  +15 fc1_colsum [F]   (ln_fold only, else unused: with ln_fold the entries +2 / +3 / +9 / +10 of every layer BUT THE LAST
 *     of the pass hold the folded forms fp16(gamma (.) W) and W beta + b, and +14 / +15 the column sums of the folded weights; for
 *     attn_variant != 1 the Q rows of the folded qkv entries are also multiplied by head_dim^-0.5 * log2(e), the pre-scale the unfolded
 *     pass applies as a column scale; the last layer keeps plain weights -- its LayerNorm 1 runs as a kernel and feeds the key hook)
 * The last layer uses only ln1 and the K slice (rows D..2D-1) of qkv_w / qkv_b: that projection, bias included and
 * before the head split, is what the reference's forward hook captures (feature_extractor.py:42,46-47).
 * key_out f32 [B, D, H/P, W/P].  full_last_layer != 0 additionally runs the rest of the last layer exactly as the
 * reference does (its output is discarded there too); key_out is identical either way. */
#define UCOD_VIT_LAYER_STRIDE 16
typedef struct {
  int B, C, H, W, P;      /* images */
  int D, heads, F, L;     /* width, heads (head_dim = D/heads = 64), MLP width, layers */
  int Kpad;               /* padded C*P*P */
  float eps;              /* LayerNorm eps (1e-6 for DINOv2 / DINO) */
  int full_last_layer;
  int gemm_variant;       /* ucod_gemm_bf16's variant for every GEMM of the pass (0 = auto) */
  int attn_variant;       /* 0 (auto) / 2: pre-scaled-Q product kernel; 1: generic-scale kernel; 8: the fp8 path of BASELINE configs[4]; 5 / 66: the
                             two pre-scaled-Q kernels by name; anything else is refused (UCOD_EINVAL) */
  int resid16;            /* 1: the residual stream x lives in IEEE fp16 instead of f32 (11 significand bits: 8x finer than the bf16 GEMM
                             operands it feeds, so the bf16 build's accuracy is unchanged to its own rounding).  Halves the bytes of
                             LayerNorm's read and of the out-proj / fc2 read-modify-write epilogues.  Values saturate at +-65504 and every
                             saturation is counted (ucod_resid16_overflow_fetch): a checkpoint whose residual stream exceeds fp16's range
                             is reported, never silently turned into inf / NaN.  0: f32 (exact accumulation; what the reference holds).
                             Logit max-abs vs the f32 reference at full size, random-init weights: bf16 operands 3.2e-3 either way; fp16
                             operands 3.8e-4 (f32 stream) / 6.4e-4 (fp16 stream).  Works for any batch size (small passes take the
                             128 x 128 / 64 x 64 kernels with the same epilogue). */
  int ln_fold;            /* 1 (libucod_dpl_f16.so with resid16 = 1 only): LayerNorm 1 / 2 of every layer but the last are folded into the QKV / fc1
                             GEMMs (ucod_gemm_lnfold): the fp16 stream is the A operand, no LayerNorm output is written or rounded.  The table then
                             carries folded weights (see the table layout).  0: LayerNorm kernels. */
} ucod_vit_desc;
size_t ucod_vit_workspace_bytes(const ucod_vit_desc* d);
int ucod_vit_forward(const ucod_vit_desc* d, const void* const* table_host, const float* img, float* key_out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ split-operand (f32-equivalent) backbone pass (rows B1-B8 at the reference's fp32)
 * The reference builds its cached training features with the backbone in plain fp32 (data/datasets/base_dataset.py:124-138: no autocast; only the Look-Twice
 * extractor is autocast, engine/runner/loop_UCOD_DPL.py:289-290).  This pass computes every matrix product of the ViT with both f32 operands written as sums
 * of `terms` bf16 values -- 2: a0 b0 + a0 b1 + a1 b0 (16 significand bits per operand, 3x the MFMA work; logits within 3e-5 of the f32 oracle on trained-like
 * weights), 3: + a1 b1 + a0 b2 + a2 b0 (24 bits: f32-equivalent, 6x) -- each partial product exact in the MFMA's f32 accumulator, f32 residual stream, f32
 * two-pass LayerNorm, exact-erf GELU, f32 softmax.  The partial products ride on ucod_gemm_bf16 by concatenation along K: an [M, K] operand is stored as
 * bf16 [M, P K], P = ucod_split_products(terms) = 3 / 6, segment p of an A-side operand (role 0) holding term {0,0,1,1,0,2}[p] and of a B-side operand (role 1)
 * term {0,1,0,1,2,0}[p].  libucod_dpl.so only (bf16 MFMA); the fp16 build returns UCOD_EINVAL.  csrc/split.hip. */
int ucod_split_products(int terms);
/* in f32 [M, K] with row pitch ld_in floats -> out bf16 [M, P K].  op 0: the values; 1: exact-erf GELU of them (modeling_dinov2.py:289); 2: times alpha.  K % 8 == 0 */
int ucod_split_rows(const float* in, long ld_in, void* out_bf16, int M, int K, int terms, int role, int op, float alpha, void* stream);
/* nn.LayerNorm (two-pass f32) of f32 rows, written as the split operand of the next GEMM: out bf16 [rows, P D].  D % 128 == 0, D <= 1536 */
int ucod_layernorm_split(const float* x, const float* gamma, const float* beta, void* out_bf16, int rows, int D, float eps, int terms, int role, void* stream);
/* ucod_patch_im2col with split output: patches bf16 [B gh gw, P Kpad] (A side) */
int ucod_patch_im2col_split(const float* img, void* patches_bf16, int B, int C, int H, int W, int P, int Kpad, int terms, void* stream);
/* f32 qkv [B tok, 3 heads 64] (the QKV projection through UCOD_EPI_BIAS_F32) -> the attention kernel's operands in `operands` (ucod_attention_split_operand_bytes):
 * Qc | Kc bf16 [B heads][tok_pad][terms 64] (one 64-wide segment per term; Q times qscale before the split; tok_pad = tok rounded up to 32, pad rows zero)
 * and Vt bf16 [terms][B heads][64][tok_pad] */
size_t ucod_attention_split_operand_bytes(int B, int tok, int heads, int terms);
int ucod_qkv_split(const float* qkv, void* operands, int B, int tok, int heads, int terms, float qscale, void* stream);
/* softmax(Q K^T hd^-0.5) V (modeling_dinov2.py:153-179) on those operands, Q carrying head_dim^-0.5 log2 e; scores, softmax and accumulation in f32, the
 * probabilities split into `terms` bf16 values in registers.  out bf16 [B tok, P heads 64]: the A-side split operand of the out-projection. */
int ucod_attention_split_fwd(const void* operands, void* out_split_bf16, int B, int tok, int heads, int terms, void* stream);
/* The whole pass, image -> last-layer key map f32 [B, D, H/P, W/P] (data/utils/feature_extractor.py:42-59), key-minimal (d->full_last_layer must be 0; resid16,
 * ln_fold and attn_variant of the descriptor are ignored).  Table as ucod_vit_forward's with every weight matrix in its split form:
 *   +0 patch_w bf16 [D, P Kpad] (B side)  +1 patch_b  +2 cls  +3 pos;  layer l at 4 + UCOD_VIT_LAYER_STRIDE l:  +2 qkv_w [3D, P D]  +4 proj_w [D, P D]
 *   +9 fc1_w [F, P D]  +11 fc2_w [D, P F]  (B side),  +14 the K rows of qkv_w as an A-side operand [D, P D] (key hook; needed for the last layer of the pass),
 *   the f32 vectors (+0 +1 +3 +5 +6 +7 +8 +10 +12 +13) as in ucod_vit_forward. */
size_t ucod_vit_split_workspace_bytes(const ucod_vit_desc* d, int terms);
/* byte offset of the f32 residual stream x [B tok, D] inside that workspace: after the pass it holds the input of the pass's last layer (whose CLS rows the pseudo-label
 * generator's attention row is computed from, generate_pseudo_label.py:78-89); (size_t)-1 for a descriptor the pass does not take */
size_t ucod_vit_split_stream_offset(const ucod_vit_desc* d, int terms);
int ucod_vit_forward_split(const ucod_vit_desc* d, int terms, const void* const* table_host, const float* img, float* key_out, void* workspace,
                           size_t workspace_bytes, void* stream);

/* Backbone-backward mode, whole passes (row B9; operand formats in ucod_dpl_amd/csrc/vit_train.hip).
 * T = the table of ucod_vit_forward; TT = per-layer training table (HOST array of DEVICE pointers), layer l at
 * UCOD_VIT_TRAIN_STRIDE*l:
 *   +0 qkv_w_aug bf16 [3D, D+64]   +1 qkv_wT_aug bf16 [D, 3D+64]   (aug columns maintained by ucod_lora_pack)
 *   +2 proj_w^T bf16 [D,D]   +3 fc1_w^T bf16 [D,F]   +4 fc2_w^T bf16 [F,D]   (frozen; transposed once by the host)
 *   +5 LoRA parameters f32 [6*r*D]   +6 LoRA gradients f32 [6*r*D] (overwritten by ucod_vit_backward)
 * forward_train saves its activations in `workspace`; backward must be given the same, untouched workspace.
 * dkey f32 [B, D, H/P, W/P] = cotangent of key_out.  gemm_variant of the embedded desc applies; attention is the
 * pre-scaled kernel.  With lora_dropout > 0 the aug A^T columns of qkv_wT_aug must have been packed as zeros (ucod_lora_pack). */
#define UCOD_VIT_TRAIN_STRIDE 7
typedef struct {
  ucod_vit_desc vit;
  int lora_r;                    /* models/modules/full_model.py:48: r = 2 */
  float lora_scaling;            /* lora_alpha / r = 4 / 2 */
  float lora_dropout;            /* full_model.py:50: 0.05 in the reference config; 0 = off (eval mode) */
  unsigned long long seed;       /* dropout seed of THIS step: forward_train and backward must be given the same value */
} ucod_vit_train_desc;
size_t ucod_vit_train_workspace_bytes(const ucod_vit_train_desc* t);
int ucod_vit_forward_train(const ucod_vit_train_desc* t, const void* const* table_host, const void* const* train_table_host,
                           const float* img, float* key_out, void* workspace, size_t workspace_bytes, void* stream);
int ucod_vit_backward(const ucod_vit_train_desc* t, const void* const* table_host, const void* const* train_table_host,
                      const float* dkey, void* workspace, size_t workspace_bytes, void* stream);
/* No-grad pass of the LoRA backbone -- the EMA teacher of models/modules/full_model.py:84,108-111 (backbone_ema under torch.no_grad()):
 * the arithmetic of ucod_vit_forward_train (LoRA as aug columns, the same dropout masks for the same seed) with nothing saved for a
 * backward; t->vit.resid16 may be 1 (fp16 residual stream, as ucod_vit_forward).  Own, inference-sized workspace. */
size_t ucod_vit_lora_infer_workspace_bytes(const ucod_vit_train_desc* t);
int ucod_vit_forward_lora_infer(const ucod_vit_train_desc* t, const void* const* table_host, const void* const* train_table_host,
                                const float* img, float* key_out, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ decoder / APM path (rows A1-A8) */

/* F.interpolate(mode='bilinear', align_corners=False) on `planes` independent [ih,iw] maps
 * (engine/runner/loop_UCOD_DPL.py:153-154,236,241,305,315,356-358). */
int ucod_bilinear_resize(const float* in, float* out, int planes, int ih, int iw, int oh, int ow, void* stream);
/* Transpose of the above: gin [planes,ih,iw] = U^T gout [planes,oh,ow] (same taps and weights as the forward).  The
 * training step runs the 1x1 decoupling conv and its weight gradient on the backbone's native 37x37 grid and moves
 * d / gd across the resize instead of the 768-channel features (conv and resize commute). */
int ucod_bilinear_resize_adjoint(const float* gout, float* gin, int planes, int ih, int iw, int oh, int ow, void* stream);

/* 1x1 "decoupling" conv as an exact-f32 MFMA GEMM (models/modules/DBA.py:13,35):
 * d[b][n][p] = sum_c W[n][c]*x[b][c][p] + bias[n];  x [B,C,HW], W [Nout,C], d [B,Nout,HW].
 * Nout = 128 (one decoder) or 256 (student rows 0..127 | teacher rows 128..255 sharing the x read). */
int ucod_dba_project(const float* x, const float* W, const float* bias, float* d, int B, int C, int HW, int Nout,
                     void* stream);
/* The same contraction on the bf16 matrix pipe with every f32 operand split into three bf16 terms (24 significand bits) and the six
 * partial products of weight >= 2^-16 accumulated in f32: f32-equivalent (what is dropped is below 2^-24 |W x| per term, the rounding of
 * an f32 FMA chain), 2.7x less matrix-pipe time than v_mfma_f32_32x32x2_f32.  C % 16 == 0, Nout = 128 or 256.
 * ws: ucod_dba_project_split_workspace_bytes(C, Nout) bytes, 16-byte aligned (the pre-split W planes). */
size_t ucod_dba_project_split_workspace_bytes(int C, int Nout);
int ucod_dba_project_split(const float* x, const float* W, const float* bias, float* d, void* ws, size_t ws_bytes, int B, int C, int HW,
                           int Nout, void* stream);

/* L2 norm over the PIXEL axis of (d * emb) per (image, channel), clamped at 1e-12
 * (F.normalize(dim=1) on [B,HW,64], DBA.py:40-41).  d is a [B, ld_c, HW] buffer, channels c0..c0+127 used;
 * emb [2,64] -> channel c scales by emb[c/64][c%64].  norm out [B,128]. */
int ucod_dba_colnorm(const float* d, int ld_c, int c0, const float* emb, float* norm, int B, int HW, void* stream);

/* gate + heads (DBA.py:48-52): a = sigmoid(f*d)+d, fg = w_fg.a1 + b_fg, bg = w_bg.a2 + b_bg, f = d*emb/norm.
 * head_w [2,64] = (conv_out_fg.weight, conv_out_bg.weight), head_b [2].  fg,bg [B,HW] (bg may be NULL).
 * sdiag (may be NULL) [B] receives sum_p (f1_p . f2_p)^2, the diagonal term of the orthogonality loss. */
int ucod_dba_heads_fwd(const float* d, int ld_c, int c0, const float* emb, const float* norm, const float* head_w,
                       const float* head_b, float* fg, float* bg, float* sdiag, int B, int HW, void* stream);

/* Orthogonality loss in Gram form (DBA.py:25-29 rewritten):
 * gram [B,2,64,64] (G1,G2 of the normalised features), sdiag [B] = sum_i (f1_i.f2_i)^2 from ucod_dba_heads_fwd,
 * loss[0] = (sum_b tr(G1 G2) - sum_b sdiag[b]) / (B*HW*HW).  ws: ucod_orth_workspace_bytes. */
size_t ucod_orth_workspace_bytes(int B, int HW);
int ucod_orth_gram_fwd(const float* d, int ld_c, int c0, const float* emb, const float* norm, const float* sdiag,
                       float* gram, float* loss, void* ws, int B, int HW, void* stream);

/* Backward of heads + gate + HW-axis normalisation + Gram orthogonality loss w.r.t. d (closed form,
 * SURVEY.md section 7).  gfg,gbg [B,HW] upstream logit grads, gextra = dL/d(extra_loss).
 * Outputs: gd [B,128,HW]; g_head_w [2,64], g_head_b [2], g_dec_bias [128] (each zeroed inside, then accumulated).
 * The gradient of learnable_embedding is analytically zero (its scale cancels under the HW-axis normalisation;
 * the reference's autograd value is f32 cancellation noise) and is not produced. */
size_t ucod_dba_bwd_workspace_bytes(int B, int HW);
int ucod_dba_bwd(const float* d, int ld_c, int c0, const float* emb, const float* norm, const float* head_w,
                 const float* gram, const float* gfg, const float* gbg, float gextra, float* gd, float* g_head_w,
                 float* g_head_b, float* g_dec_bias, void* ws, int B, int HW, void* stream);

/* weight gradient of the decoupling conv: gW[n][c] += sum_{b,p} gd[b][n][p]*x[b][c][p]  (f32 MFMA, split-K
 * over (b, pixel chunks) with f32 atomics into gW [128,C], which is zeroed inside first). */
int ucod_dba_wgrad(const float* gd, const float* x, float* gW, int B, int C, int HW, void* stream);
/* The same weight gradient on the bf16 matrix pipe, both operands split into three bf16 terms while staged (see ucod_dba_project_split:
 * f32-equivalent); split-K over (image, pixel chunk) with f32 atomics, gW zeroed inside first. */
int ucod_dba_wgrad_split(const float* gd, const float* x, float* gW, int B, int C, int HW, void* stream);

/* APM discriminator forward, always train-mode BatchNorm (models/discriminator.py:60-70,86-95;
 * dis_use_features=False).  params (f32, reference state_dict order):
 *   w1 [32,1,3,3] g1 b1 [32]  w2 [16,32,3,3] g2 b2 [16]  w3 [8,16,3,3] g3 b3 [8]  lin_w [8*ceil(fs/4)^2] lin_b [1]
 * running = (rm1,rv1,rm2,rv2,rm3,rv3) updated in place with momentum 0.1 when update_running != 0.
 * mask [B,1,fs,fs] -> prob [B].  `saved` (ucod_disc_saved_bytes) keeps pre-BN activations + batch statistics
 * for ucod_disc_bwd. */
typedef struct {
  const float *w1, *g1, *b1, *w2, *g2, *b2, *w3, *g3, *b3, *lin_w, *lin_b;
  float *rm1, *rv1, *rm2, *rv2, *rm3, *rv3;
  long long* nbt; /* ABI 3: the three num_batches_tracked counters as ONE int64[3] (or NULL): += 1 per call with update_running */
} ucod_disc_params;
size_t ucod_disc_saved_bytes(int B, int fs);
int ucod_disc_fwd(const float* mask, const ucod_disc_params* p_host, float* prob, void* saved, int B, int fs,
                  int update_running, void* stream);
/* gradient of sum_b gprob[b]*prob[b] w.r.t. the 11 parameter tensors, written (not accumulated unless
 * accumulate != 0) into grads laid out in the same order/shapes as ucod_disc_params' const members.
 * (engine/runner/loop_UCOD_DPL.py:244-251) */
typedef struct { float *w1, *g1, *b1, *w2, *g2, *b2, *w3, *g3, *b3, *lin_w, *lin_b; } ucod_disc_grads;
size_t ucod_disc_bwd_workspace_bytes(int B, int fs);
int ucod_disc_bwd(const float* mask, const ucod_disc_params* p_host, const void* saved, const float* gprob,
                  const ucod_disc_grads* g_host, int accumulate, void* ws, int B, int fs, void* stream);

/* APM fusion + both BCE-with-logits losses + their logit gradients in one pass
 * (engine/runner/loop_UCOD_DPL.py:257-272 and :161-173):
 *   w[b] = clamp(0.5*(1+cos(pi*|p_s-p_p|)) + epoch_frac, 0, 1);  merged = pl*(1-w) + (sigmoid(teacher)>0.5)*w
 *   losses[0] = mean BCEWithLogits(fg, merged), losses[1] = mean BCEWithLogits(bg, 1-merged),
 *   losses[2] = dis_loss = mean BCE(p_s, 0);  gfg = (sigmoid(fg)-merged)/(B*HW)*gscale, gbg likewise.
 * pl, teacher, fg, bg, merged, gfg, gbg: [B,HW]; p_s, p_p, w: [B].  losses [4] (zeroed inside; [3] unused). */
int ucod_apm_bce(const float* pl, const float* teacher, const float* fg, const float* bg, const float* p_s,
                 const float* p_p, float epoch_frac, float gscale, float* w, float* merged, float* gfg, float* gbg,
                 float* losses, int B, int HW, void* stream);
/* out[i] = (sigmoid(x[i]) > 0.5) when logits != 0, else (x[i] > 0.5), as 0/1 floats (loop_UCOD_DPL.py:240-241,258-261) */
int ucod_binarize(const float* x, float* out, size_t n, int logits, void* stream);

/* torch.optim.AdamW step + EMA teacher update over a flat f32 arena
 * (engine/runner/runner.py:282-298; loop_UCOD_DPL.py:178,186-191).  ema may be NULL. */
/* loss of TrainLoop._process_batch (loop_UCOD_DPL.py:161-169): out[0] = losses[0] + losses[1] + extra[0] (- losses[2] unless finetune), from
 * the four scalars ucod_apm_bce left in `losses` and the orthogonality loss -- one launch instead of three elementwise adds (ABI 3). */
int ucod_step_loss(const float* losses, const float* extra, int finetune, float* out, void* stream);
/* dst[q][0..n[q]) = src[q][0..n[q]) for q < count <= 4, one launch (host arrays of device pointers / element counts; f32).  What
 * loop_UCOD_DPL.py's refresh of the shared student | teacher projection needs per step instead of four device-to-device copies. */
int ucod_copy_segments(float* const* dst, const float* const* src, const size_t* n, int count, void* stream);
/* One launch that zeroes up to eight device regions (HOST arrays of DEVICE pointers and byte counts, 4-byte aligned): the accumulating outputs of a
 * whole step at once.  Together with ucod_accumulators_prezeroed it replaces the memset each accumulating entry point otherwise issues in front of
 * its kernel (round 4: seven rocclr fills among the ~125 launches of a student step). */
int ucod_zero_segments(void* const* dst, const size_t* bytes, int count, void* stream);
/* on != 0: for the calls issued NEXT FROM THIS HOST THREAD the caller guarantees that these outputs are already zero (ucod_zero_segments), and the entry
 * points skip their own memset: `sdiag` of ucod_dba_heads_fwd, `losses` of ucod_apm_bce, g_head_w / g_head_b / g_dec_bias of ucod_dba_bwd, `gW` of
 * ucod_dba_wgrad / ucod_dba_wgrad_split, and the last 112 doubles of ucod_disc_fwd's `saved` buffer (its BatchNorm sums -- which every ucod_disc_fwd call
 * also LEAVES zero, so a saved buffer that was allocated zeroed stays valid call after call).  Host-side state, like ucod_resid16_overflow_bind: set it,
 * issue the step, clear it. */
int ucod_accumulators_prezeroed(int on);
int ucod_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float ema_alpha, void* stream);

/* Discriminator WITH the feature branch (models/discriminator.py:77-95, dis_use_features=True; no shipped config enables it): forward only, from
 * generic pieces.  A ConvBlock(Cin, Cout, 3, stride, 1) is  ucod_unfold3x3 -> ucod_dba_project(W.reshape(Cout, Cin*9) zero-padded to Kpad columns,
 * zero bias) -> ucod_bn_lrelu_train.
 * ucod_unfold3x3: x f32 [B,C,H,W] -> out f32 [B,Kpad,Ho*Wo], row c*9 + ky*3 + kx = the input shifted by (ky-1, kx-1) with zero padding (F.unfold
 *   order), rows >= C*9 zero; Ho = (H-1)/stride + 1; stride 1 or 2; Kpad % 16 == 0 for the GEMM.
 * ucod_bn_lrelu_train: y f32 [B,C,HW] in place: training-mode nn.BatchNorm2d (batch statistics, biased variance; running buffers updated with
 *   `momentum` and the unbiased variance when update_running != 0) followed by LeakyReLU(slope).  Statistics in f64, deterministic.
 * ucod_linear_sigmoid: out[b] = sigmoid(x[b,:] . w + bias[0]). */
int ucod_unfold3x3(const float* x, float* out, int B, int C, int H, int W, int stride, int Kpad, void* stream);
size_t ucod_bn_lrelu_workspace_bytes(int C);
int ucod_bn_lrelu_train(float* y, const float* gamma, const float* beta, float* running_mean, float* running_var, int B, int C, int HW, float eps,
                        float momentum, float slope, int update_running, void* workspace, size_t workspace_bytes, void* stream);
int ucod_linear_sigmoid(const float* x, const float* w, const float* bias, float* out, int B, int K, void* stream);
/* Training that discriminator (the discriminator phase, engine/runner/loop_UCOD_DPL.py:230-255, with dis_use_features=True): the backward of the
 * same pieces.  A ConvBlock's backward is  ucod_bn_lrelu_bwd -> ucod_conv_wgrad_f32 (weight gradient from the saved unfold) -> [input gradient:
 * ucod_dba_project with W^T on the gradient, then ucod_fold3x3].
 * ucod_bn_lrelu_train_save: as ucod_bn_lrelu_train, out of place (y_pre is kept for the backward) and with the batch statistics left in `stats`
 *   (ucod_bn_lrelu_workspace_bytes(C) bytes, owned by the caller until the backward has run).
 * ucod_bn_lrelu_bwd: gout = dL/d(block output) -> gy_pre = dL/d(conv output); dgamma / dbeta written, or added to when accumulate != 0 (the phase
 *   calls the discriminator twice per step).  f64 sums in a fixed order.
 * ucod_fold3x3: the adjoint of ucod_unfold3x3 (col2im): gcols [B,Kpad,Ho*Wo] -> gx [B,C,H,W].
 * ucod_conv_wgrad_f32: gW [Nout,C] (+)= sum_{b,p} gd[b][n][p] * cols[b][c][p]  (exact-f32 MFMA, split over pixel chunks, f32 atomics).
 * ucod_linear_sigmoid_bwd: gprob = dL/dprob -> gx [B,K], gw [K], gb [1] (written or accumulated).
 * ucod_disc_bce: nn.BCELoss(cat(probs_student, probs_pseudo), [0..0, 1..1]) (mean over 2B, logs clamped at -100 like torch) -> loss[0], and its
 *   gradient times `inv` / (the mean's 1 / 2B is part of inv = 1 / (2 B world)) with torch's max(p (1 - p), 1e-12) denominator. */
int ucod_bn_lrelu_train_save(const float* y_pre, float* y_out, const float* gamma, const float* beta, float* running_mean, float* running_var, int B, int C,
                             int HW, float eps, float momentum, float slope, int update_running, void* stats, size_t stats_bytes, void* stream);
int ucod_bn_lrelu_bwd(const float* y_pre, const float* gout, float* gy_pre, const void* stats, const float* gamma, const float* beta, float* dgamma,
                      float* dbeta, int B, int C, int HW, float eps, float slope, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
int ucod_fold3x3(const float* gcols, float* gx, int B, int C, int H, int W, int stride, int Kpad, void* stream);
int ucod_conv_wgrad_f32(const float* gd, const float* cols, float* gW, int B, int C, int HW, int Nout, int accumulate, void* stream);
int ucod_linear_sigmoid_bwd(const float* x, const float* w, const float* prob, const float* gprob, float* gx, float* gw, float* gb, int B, int K,
                            int accumulate, void* stream);
int ucod_disc_bce(const float* probs_student, const float* probs_pseudo, float* g_student, float* g_pseudo, float* loss, int B, float inv, void* stream);

/* ------------------------------------------------------------------ Look-Twice (rows L1-L3) */

/* 8-connected component labelling of a HOST uint8 [H,W] mask (non-zero = foreground) into HOST int32 labels
 * (0 = background, k = k-th component in OpenCV's numbering: raster order of the components' first 2 x 2 blocks -- the order in which
 * the block-based labelling of cv2.connectedComponents(connectivity=8) creates and flattens its labels); returns the number of labels
 * INCLUDING the background, like cv2.connectedComponents (engine/runner/loop_UCOD_DPL.py:366).  < 0 on error. */
int ucod_ccl8_host(const uint8_t* mask_host, int H, int W, int32_t* labels_host);

/* Pillow Image.resize on an 8-bit single-channel HOST image: antialiased separable resample with 22-bit fixed-point
 * coefficients, horizontal then vertical pass (filter 0 = BILINEAR, 1 = BICUBIC, the Pillow default used for the
 * 'L' mask at loop_UCOD_DPL.py:350).  Bit-identical to Pillow. */
int ucod_pil_resize_u8_host(const uint8_t* src_host, int h, int w, uint8_t* dst_host, int oh, int ow, int filter);

/* Pseudo-label generator (generate_pseudo_label.py:71-94, data/utils/found_bkg_mask.py:4-86).
 * ucod_vit_last_ln1_offset: byte offset, inside the workspace of ucod_vit_forward (full_last_layer == 0), of the last layer's
 * LayerNorm-1 output bf16 [B*tok, D]; valid until the next pass on that workspace.
 * ucod_cls_qk: q_cls, k_cls f32 [B,D] = that layer's query / key projection of token 0 (qkv_w bf16 [3D,D], qkv_b f32 [3D]).
 * ucod_cls_attention: att f32 [B,heads,hw] = outputs.attentions[-1][:, :, 0, 1:] -- softmax of the CLS query over CLS + all
 * patch keys, patch columns only -- with the patch keys taken from key_map f32 [B, D, hw] (the backbone's NCHW output).
 * ucod_bkg_seg: found_bkg_mask.py:31-86 with up_size = grid: bkg_mask, sim_map f32 [B,hw] (sim_map already multiplied by
 * 1 - bkg_mask and normalised by the batch-wide maximum, :81-82), cos_row f32 [B,hw], seed int32 [B], beta f32 [B,heads];
 * scratch4 = 4 bytes of device scratch. */
size_t ucod_vit_last_ln1_offset(const ucod_vit_desc* d);
int ucod_cls_qk(const void* h_ln1_bf16, const void* qkv_w_bf16, const float* qkv_b, float* q_cls, float* k_cls, int B, int tok, int D, void* stream);
int ucod_cls_attention(const float* q_cls, const float* k_cls, const float* key_map, float* att, int B, int heads, int hw, float scale, void* stream);
int ucod_bkg_seg(const float* att, const float* key_map, float th_bkg, float epsilon, int apply_weights, float* bkg_mask, float* sim_map,
                 float* cos_row, int* seed, float* beta, void* scratch4, int B, int heads, int hw, void* stream);

/* GPU Look-Twice tail.
 * ucod_ccl8_components: 8-connected components of a DEVICE uint8 [H,W] mask (non-zero = foreground) -> a DEVICE table of
 * `*count` rows {root, area, xmin, xmax, ymin, ymax, order} (7 x int32; at most `capacity` rows are written, *count may exceed it),
 * in arbitrary order.  root = linear index of the component's first pixel in raster order; order = raster index of its first 2 x 2
 * block: sorting rows by `order` yields cv2.connectedComponents' label order (labels 1..n, see ucod_ccl8_host) --
 * loop_UCOD_DPL.py:366-384 needs only area and bounding box per label. */
size_t ucod_ccl8_workspace_bytes(int H, int W);
int ucod_ccl8_components(const uint8_t* mask_dev, int H, int W, int32_t* table_dev, int capacity, int32_t* count_dev, void* workspace,
                         size_t workspace_bytes, void* stream);

/* For each box i (HOST int32 [nbox,4] = x,y,w,h in canvas pixels, pasted in order): Pillow-BICUBIC resize of the DEVICE
 * uint8 mask i ([nbox, sh, sw]) to (w,h) and paste into the DEVICE uint8 canvas [CH,CW], clipped to the canvas
 * (Image.resize + Image.paste, loop_UCOD_DPL.py:346-352).  Bit-identical to Pillow.  w,h <= 0 is rejected like PIL does. */
size_t ucod_paste_workspace_bytes(int nbox, int max_w, int max_h, int sh, int sw);
int ucod_paste_resized_u8(const uint8_t* masks_dev, int nbox, int sh, int sw, const int32_t* boxes_host, uint8_t* canvas_dev, int CH, int CW,
                          void* workspace, size_t workspace_bytes, void* stream);

/* The same for boxes that belong to SEVERAL canvases `canvases_dev` u8 [ncanvas,CH,CW] (the batched Look-Twice validation of BASELINE configs[3]: every
 * image of a validation batch in one call): box_canvas_host int32 [nbox] names each box's canvas (NULL: all on canvas 0); boxes are pasted in the order
 * given, so one canvas's boxes keep their order. */
int ucod_paste_resized_u8_multi(const uint8_t* masks_dev, int nbox, int sh, int sw, const int32_t* boxes_host, const int32_t* box_canvas_host,
                                uint8_t* canvases_dev, int ncanvas, int CH, int CW, void* workspace, size_t workspace_bytes, void* stream);

/* Batched crop + Pillow-BILINEAR resize + ToTensor + ImageNet Normalize on the GPU
 * (PIL crop + torchvision Resize/ToTensor/Normalize, loop_UCOD_DPL.py:282-286,341-342).
 * img u8 [H,W,3] (HWC, device); boxes_host int32 [nbox,4] = (x,y,w,h) in source pixels (regions outside the image read
 * as zero, as PIL's crop does); out f32 [nbox,3,oh,ow].  The 8-bit intermediate is bit-identical to Pillow's. */
size_t ucod_crop_workspace_bytes(int nbox, int max_crop_h, int max_crop_w, int oh, int ow);
int ucod_crop_resize_norm(const uint8_t* img, int H, int W, const int32_t* boxes_host, int nbox, float* out, int oh, int ow,
                          void* workspace, size_t workspace_bytes, void* stream);

/* The same for boxes cut from SEVERAL source images in one launch pair ("batch all crops of all images"): imgs_host = HOST array of nimg
 * DEVICE pointers to u8 [H_i,W_i,3] images, hw_host int32 [nimg,2] = (H_i, W_i), box_image_host int32 [nbox] = source image of each box. */
int ucod_crop_resize_norm_multi(const uint8_t* const* imgs_host, const int32_t* hw_host, int nimg, const int32_t* box_image_host,
                                const int32_t* boxes_host, int nbox, float* out, int oh, int ow, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ CORAL SparseRefiner (rows R1-R4), inference */

/* Cross-attention core of nn.MultiheadAttention with head_dim 96 (models/modules/mlp.py:122,143; 8 heads on C=768):
 * softmax(Q K^T) V per (window, head).  q bf16 [B*Nq, ldq] (head h at columns h*96.., PRE-SCALED by 96^-0.5*log2(e));
 * k, v bf16 rows of stride ldkv (may point into one [k|v] buffer); out bf16 [B*Nq, heads*96]. */
int ucod_cross_attention96_fwd(const void* q_bf16, int ldq, const void* k_bf16, const void* v_bf16, int ldkv, void* out_bf16,
                               int B, int Nq, int Nk, int heads, void* stream);

/* out[(i*HW + p)*C + c] = src[(src_idx[i]*C + c)*HW + p]: window gather (models/modules/ASR.py:13-20) fused with
 * CSF._BCHW_to_BLC (models/modules/CSF.py:29-31).  src f32 [*,C,HW], src_idx int32 [n] (device), out f32 [n*HW, C]. */
int ucod_gather_tokens(const float* src, const int* src_idx, float* out, int n, int C, int HW, void* stream);

/* EntropySelector scores (models/modules/ASR.py:42-48): entropy = -p*log(max(p,1e-5)) with p = preds (use_sigmoid == 0) or
 * sigmoid(preds); scores = adaptive_avg_pool2d(entropy, (ws,ws)).  preds, entropy f32 [B,1,H,W]; scores f32 [B,ws,ws]. */
int ucod_entropy_scores(const float* preds, int use_sigmoid, float* entropy, float* scores, int B, int H, int W, int ws, void* stream);

/* depthwise 7x7 conv (padding 3, bias) + 1x1 mask head (models/modules/CSF.py:13-25,41-42) on token-major activations:
 * x f32 [n,H,W,C]; dw_tapmajor f32 [49,C] (= depthwise_conv.weight [C,1,7,7] transposed), dw_bias [C], w1 [C], b1;
 * out f32 [n,1,H,W]. */
int ucod_dwconv7_maskdec(const float* x_tokens, const float* dw_tapmajor, const float* dw_bias, const float* w1, float b1, float* out,
                         int n, int H, int W, int C, void* stream);

/* HRE.concate_windows (models/modules/HRE.py:18-39): out f32 [B,1,ws*H,ws*W] = 0, then window i [H,W] placed at
 * (coords[i][0]*H, coords[i][1]*W) of image win_img[i], divided by (1 + 1e-6).  coords int32 [n,2], win_img int32 [n] (device). */
int ucod_window_scatter(const float* windows, const int* coords, const int* win_img, float* out, int n, int B, int H, int W, int ws,
                        void* stream);

/* SparseRefiner.cal_ex_loss, training mode (models/UDLR.py:52-75): the IoU-weighted window loss.  window_preds f32 [n,1,H,W] (logits of
 * the selected windows), h_targets f32 [B*ws*ws,1,H,W] (per-window high-resolution targets, all windows), win_flat int32 [n] = b*ws*ws + j of
 * each selected window (raster order, as mask.flatten() selects), l_up f32 [B,1,ws*H,ws*W] = the first-stage logits resized bilinearly.
 * Per window: l = sigmoid(l_up block) > 0.5; iou of (targets > 0.5 -- after a sigmoid when targets_are_logits, binary_iou's "max > 1"
 * heuristic which the caller evaluates over all selected targets) with l; w = clamp(1.5 iou, 0, 1);
 * part[i] = sum_pixels w BCEWithLogits(x, t) + (1 - w) BCEWithLogits(x, l);  loss_out[0] = sum(part) / (2 n H W) (f64 sum, fixed order).
 * iou_out [n] optional. */
int ucod_window_loss(const float* window_preds, const float* h_targets, const int* win_flat, const float* l_up, int targets_are_logits,
                     float* part, float* iou_out, float* loss_out, int n, int B, int H, int W, int ws, void* stream);

/* GatedEnsembler (models/modules/GE_pix_level.py:16-25) after the bilinear resize of l1: 19x19 zero-padded average of
 * sigmoid(l1), its entropy normalised by the maximum over the WHOLE batch, gate = ((1 - en/en_max) + mean sigmoid(l1))/2,
 * y = l1*gate + l2*(1-gate), out = fuser(y) (1x1 conv 1->64, ReLU, 1x1 conv 64->1).  All maps f32 [B,1,h,w]. */
size_t ucod_gated_ensemble_workspace_bytes(int B, int h, int w);
int ucod_gated_ensemble(const float* l1_up, const float* l2, const float* fuser0_w, const float* fuser0_b, const float* fuser2_w,
                        float fuser2_b, float* out, float* weight_out, void* workspace, int B, int h, int w, void* stream);

/* ------------------------------------------------------------------ COD measures (row N4: engine/utils/metrics/metric.py::statistics)
 * pred, gt f32 [B,H,W] (any range: the measures min-max normalise the prediction and threshold the normalised ground truth at 0.5,
 * _prepare_data :125-133).  out f64 [B][UCOD_COD_RECORD], per image:
 *   [0] MAE  [1] ACC  [2] IoU  [3] S-measure  [4] weighted F  [5] adaptive E  [6] adaptive F  [7] ground-truth foreground pixels,
 *   [8..263] E-measure over the 256 thresholds, [264..519] F-measure curve (beta^2 = 0.3), [520..775] precision, [776..1031] recall
 * -- exactly what one statistics.step() appends to its seven measure objects; get_result() is means over images of these.
 * float64 arithmetic throughout, fixed reduction trees (deterministic); agreement with the reference's numpy is to summation order.
 * A constant prediction follows the reference's integer branch (pred.astype(int)); constants outside [0, 2) are not supported. */
#define UCOD_COD_RECORD 1032
size_t ucod_cod_metrics_workspace_bytes(int B, int H, int W);
int ucod_cod_metrics(const float* pred, const float* gt, int B, int H, int W, double* out, void* workspace, size_t workspace_bytes,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
