// Host-side driver of the frozen ViT backbone forward (data/utils/feature_extractor.py:49-59 ->
// transformers Dinov2Model / models/backbones/dino.py): one C call enqueues every kernel of the pass on the
// caller's stream, from the NCHW image to the [B,C,h,w] last-layer key map.  No allocation, no sync: it can be
// captured into a hipGraph by the caller.
#include <cstdlib>
#include "common.h"
#include "gemm_bf16_plan.h"
#include "../../include/ucod_dpl.h"

#ifdef UCOD_HALF_F16
#define UCOD_HALF_IS_F16 1
#else
#define UCOD_HALF_IS_F16 0
#endif

namespace {

struct Plan {
  size_t off_x, off_h, off_qkv, off_a, off_g, off_patch, off_qscale, off_f8, f8_bytes, off_stats, off_part, total;
  int M, tok;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

Plan make_plan(const ucod_vit_desc* d) {
  Plan p;
  const int gh = d->H / d->P, gw = d->W / d->P;
  p.tok = gh * gw + 1;
  p.M = d->B * p.tok;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = align_up(o + bytes, 256); return r; };
  p.off_x = take((size_t)p.M * d->D * (d->resid16 ? 2 : 4));
  p.off_h = take((size_t)p.M * d->D * 2);
  p.off_qkv = take((size_t)p.M * 3 * d->D * 2);
  p.off_a = take((size_t)p.M * d->D * 2);
  p.off_g = take((size_t)p.M * d->F * 2);
  p.off_patch = take((size_t)d->B * gh * gw * d->Kpad * 2);
  p.off_qscale = take((size_t)3 * d->D * 4);
  p.f8_bytes = d->attn_variant == 8 ? ucod_attention_fp8_workspace_bytes(d->B, p.tok, d->heads) : 0;   // Q8 | K8 | Vt8 of the fp8 attention path
  p.off_f8 = take(p.f8_bytes);
  p.off_stats = take(d->ln_fold ? (size_t)p.M * 8 : 0);            // (rstd, -mean * rstd) per token row of the folded LayerNorms (small passes)
  p.off_part = take(d->ln_fold ? (size_t)p.M * (d->D / 64) * 8 : 0);   // per-row partial (sum, M2 about the slot mean) per 64-column slot (large passes)
  p.total = o;
  return p;
}

// attn_variant of the pass: 0 (auto) and 2 = the pre-scaled-Q product kernel, 1 = the generic-scale kernel (Q as the reference holds it,
// the scale applied inside the softmax), 8 = the fp8 path of BASELINE configs[4].  Laboratory variants (variants/attention_lab.hip) are
// not reachable from the ViT driver: bench / tools that want to time one call ucod_attention_fwd_lab directly.
// 5 / 66: attn_fwd_v5_kernel / attn_fwd_v6_kernel by name (ucod_attention_fwd's variant)
inline bool attn_variant_known(int av) { return av == 0 || av == 1 || av == 2 || av == 8 || av == 5 || av == 66; }
inline bool attn_variant_takes_prescaled_q(int av) { return av != 1; }

bool valid(const ucod_vit_desc* d) {
  return d && d->B > 0 && d->C > 0 && d->P > 0 && d->H > 0 && d->W > 0 && d->H % d->P == 0 && d->W % d->P == 0 && d->D > 0 &&
         d->heads > 0 && d->D == d->heads * 64 && d->D % 128 == 0 && d->F % 128 == 0 && d->L >= 1 && d->Kpad % 64 == 0 &&
         d->Kpad >= d->C * d->P * d->P && (d->resid16 == 0 || d->resid16 == 1) && attn_variant_known(d->attn_variant) &&
         (d->ln_fold == 0 || (d->ln_fold == 1 && d->resid16 == 1 && d->attn_variant != 8 && UCOD_HALF_IS_F16 && d->D % 256 == 0 && d->D <= 1536));
}

}  // namespace

#define RUN(call)                \
  do {                           \
    int rc__ = (call);           \
    if (rc__ != 0) return rc__;  \
  } while (0)

extern "C" size_t ucod_vit_workspace_bytes(const ucod_vit_desc* d) { return valid(d) ? make_plan(d).total : 0; }
extern "C" size_t ucod_vit_last_ln1_offset(const ucod_vit_desc* d) { return valid(d) ? make_plan(d).off_h : (size_t)-1; }

extern "C" int ucod_vit_forward(const ucod_vit_desc* d, const void* const* T, const float* img, float* key_out, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!valid(d) || !T || !img || !key_out || !workspace) return UCOD_EINVAL;
  const Plan p = make_plan(d);
  if (workspace_bytes < p.total) return UCOD_ENOMEM;
  char* ws = (char*)workspace;
  float* x = (float*)(ws + p.off_x);
  void* h = ws + p.off_h;
  void* qkv = ws + p.off_qkv;
  void* a = ws + p.off_a;
  void* g = ws + p.off_g;
  void* patches = ws + p.off_patch;
  const int M = p.M, tok = p.tok, D = d->D, F = d->F, gv = d->gemm_variant, av = d->attn_variant;
  const float scale = 0.125f;  // head_dim^-0.5, head_dim = 64
  // attn_variant 0 / 2: fold head_dim^-0.5 * log2(e) into the Q third of the QKV epilogue (before its 16-bit rounding)
  // attn_variant 8: the fp8 (e4m3, block-scaled MFMA) attention path of BASELINE configs[4]; same pre-scaled Q
  const bool prescale = attn_variant_takes_prescaled_q(av);
  float* qscale = (float*)(ws + p.off_qscale);
  // (with ln_fold the softmax pre-scale rides on the folded Q rows; only an unfolded QKV projection -- the last layer's, when it runs at all -- needs the vector)
  const bool needs_qscale = prescale && !(d->ln_fold && !d->full_last_layer);
  if (needs_qscale && av != 8) RUN(ucod_fill_qscale(qscale, D, scale * 1.4426950408889634f, stream));
  // fp8 path: the QKV epilogue writes e4m3 Q8 | K8 | V8 itself; its column scales carry 2^q_exp (times the pre-scale), 2^k_exp, 2^v_exp
  constexpr int QE = 5, KE = 3, VE = 3;
  if (av == 8) {
    RUN(ucod_fill_qscale3(qscale, D, scale * 1.4426950408889634f * 32.f, 8.f, 8.f, stream));
    RUN(ucod_attention_fp8_zero_pad(ws + p.off_f8, d->B, tok, d->heads, stream));
  }

  // residual stream x: f32, or IEEE fp16 with resid16 (half the bytes of LayerNorm's read and of the out-proj / fc2 epilogues)
  const bool r16 = d->resid16 != 0;
  const int epi_patch = r16 ? UCOD_EPI_PATCH_TOKENS_H16 : UCOD_EPI_PATCH_TOKENS_F32;
  const int epi_resid = r16 ? UCOD_EPI_BIAS_SCALE_RESID_H16 : UCOD_EPI_BIAS_SCALE_RESID_F32;
  auto layernorm = [&](const float* g, const float* b) {
    return r16 ? ucod_layernorm_h16(x, g, b, h, M, D, d->eps, stream) : ucod_layernorm(x, g, b, h, M, D, d->eps, 0, stream);
  };
  // embeddings: patch conv as GEMM (+bias +pos), CLS rows
  // ln_fold: the producers of x (patch embedding + CLS rows, out-projection, fc2) leave per-row partial sums for the folded consumer that follows
  // (UCOD_EPI_*_STATS: large passes; no statistics launch at all).  `have_part` = `part` describes the current x; a producer that cannot take the
  // shape returns UCOD_EINVAL without launching, the plain epilogue runs instead and ucod_row_stats_h16 supplies `stats` (small passes).
  float* const stats = (float*)(ws + p.off_stats);
  float* const part = (float*)(ws + p.off_part);
  const int nslot = D / 64;
  static const bool no_part = ucod::lab_env("UCOD_LN_FOLD_NO_PARTIALS") != nullptr;      // measurement knob: always the statistics kernel
  bool have_part = false;
  RUN(ucod_patch_im2col(img, patches, d->B, d->C, d->H, d->W, d->P, d->Kpad, stream));
  // (the patch embedding's *_STATS form runs without the leftover-as-patches mode: worth it only when its 256 x 256 tiles come out as nearly whole
  // rounds of the chip; at 32 x 1369 rows -- 516 tiles on 256 CUs -- the plain launch plus one statistics launch is 11 us faster)
  const int patch_tiles = ucod::cdiv((long)d->B * (tok - 1), 256) * ucod::cdiv(D, 256), n_cu = ucod::device_cus();
  const bool patch_whole_rounds = patch_tiles >= n_cu && (patch_tiles % n_cu == 0 || patch_tiles % n_cu >= n_cu / 2);
  if (d->ln_fold && d->L > 1 && !no_part && patch_whole_rounds) {
    const int rc = ucod_gemm_bf16_stats(UCOD_EPI_PATCH_TOKENS_H16_STATS, patches, T[0], x, d->B * (tok - 1), D, d->Kpad, (const float*)T[1], nullptr, nullptr,
                                        (const float*)T[3], tok, part, nslot, stream);
    if (rc == UCOD_OK) have_part = true;
    else if (rc != UCOD_EINVAL) return rc;
  }
  if (have_part) {
    RUN(ucod_cls_rows_h16_stats(x, (const float*)T[2], (const float*)T[3], part, nslot, d->B, tok, D, stream));
  } else {
    RUN(ucod_gemm_bf16(epi_patch, patches, T[0], x, d->B * (tok - 1), D, d->Kpad, (const float*)T[1], nullptr, nullptr,
                       (const float*)T[3], tok, gv, stream));
    if (r16) RUN(ucod_cls_rows_h16(x, (const float*)T[2], (const float*)T[3], d->B, tok, D, stream));
    else RUN(ucod_cls_rows(x, (const float*)T[2], (const float*)T[3], d->B, tok, D, stream));
  }
  // out-projection / fc2 (+ LayerScale + residual) into x; `want_part`: a folded consumer reads x next
  auto resid_gemm = [&](const void* act, const void* w, const float* b, const float* ls, int K, bool want_part) -> int {
    have_part = false;
    if (want_part && !no_part) {
      const int rc = ucod_gemm_bf16_stats(UCOD_EPI_BIAS_SCALE_RESID_H16_STATS, act, w, x, M, D, K, b, ls, x, nullptr, tok, part, nslot, stream);
      if (rc == UCOD_OK) { have_part = true; return UCOD_OK; }
      if (rc != UCOD_EINVAL) return rc;
    }
    return ucod_gemm_bf16(epi_resid, act, w, x, M, D, K, b, ls, x, nullptr, tok, gv, stream);
  };

  for (int l = 0; l < d->L; ++l) {
    const void* const* W = T + 4 + UCOD_VIT_LAYER_STRIDE * l;
    const bool last = (l == d->L - 1);
    // ln_fold: LayerNorm 1 / 2 of every layer but the last live in the QKV / fc1 epilogues (ucod_gemm_lnfold: the fp16 stream x is the A operand,
    // W[2] / W[9] hold fp16(gamma (.) W), W[3] / W[10] the folded bias, W[14] / W[15] the column sums); only the row statistics are computed here
    const bool fold = d->ln_fold != 0 && !last;
    const bool next_fold = d->ln_fold != 0 && l + 1 < d->L - 1;    // LayerNorm 1 of the next layer is folded too
    if (fold) { if (!have_part) RUN(ucod_row_stats_h16(x, stats, M, D, d->eps, stream)); }
    else RUN(layernorm((const float*)W[0], (const float*)W[1]));
    if (last) {
      // key hook: only the K slice (rows D..2D-1) of the fused qkv weight; output written as [B,D,h,w]
      const char* wk = (const char*)W[2] + (size_t)D * D * 2;
      const float* bk = (const float*)W[3] + D;
      RUN(ucod_gemm_bf16(UCOD_EPI_KEY_NCHW_F32, wk, h, key_out, D, M, D, bk, nullptr, nullptr, nullptr, tok, gv, stream));
      if (!d->full_last_layer) break;
    }
    if (av == 8) {
      RUN(ucod_gemm_bf16(UCOD_EPI_QKV_FP8, h, W[2], ws + p.off_f8, M, 3 * D, D, (const float*)W[3], qscale, nullptr, nullptr, tok, gv, stream));
      RUN(ucod_attention_fwd_fp8_fused(ws + p.off_f8, a, d->B, tok, d->heads, QE, KE, VE, stream));
    } else {
      if (fold) RUN(ucod_gemm_lnfold(UCOD_EPI_LNFOLD_BIAS_BF16, x, W[2], qkv, M, 3 * D, D, (const float*)W[3], (const float*)W[14], have_part ? nullptr : stats,
                                     have_part ? part : nullptr, nslot, d->eps, nullptr /* the folded Q rows carry the softmax pre-scale */, gv, stream));
      else RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_BF16, h, W[2], qkv, M, 3 * D, D, (const float*)W[3], prescale ? qscale : nullptr, nullptr, nullptr, tok, gv, stream));
      RUN(ucod_attention_fwd(qkv, a, d->B, tok, d->heads, prescale ? 0.f : scale, (av == 5 || av == 66) ? av : 0, stream));
    }
    RUN(resid_gemm(a, W[4], (const float*)W[5], (const float*)W[6], D, fold));
    if (fold) {
      if (!have_part) RUN(ucod_row_stats_h16(x, stats, M, D, d->eps, stream));
      RUN(ucod_gemm_lnfold(UCOD_EPI_LNFOLD_GELU_BF16, x, W[9], g, M, F, D, (const float*)W[10], (const float*)W[15], have_part ? nullptr : stats,
                           have_part ? part : nullptr, nslot, d->eps, nullptr, gv, stream));
    } else {
      RUN(layernorm((const float*)W[7], (const float*)W[8]));
      RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_GELU_BF16, h, W[9], g, M, F, D, (const float*)W[10], nullptr, nullptr, nullptr, tok, gv, stream));
    }
    RUN(resid_gemm(g, W[11], (const float*)W[12], (const float*)W[13], F, next_fold));
  }
  return UCOD_OK;
}

extern "C" const char* ucod_half_name(void) { return UCOD_HALF_NAME; }
