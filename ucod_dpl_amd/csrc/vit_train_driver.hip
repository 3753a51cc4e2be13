// Host-side driver of the backbone-backward mode (models/modules/full_model.py:79-126: LoRA-wrapped
// backbone -> last-layer key hook -> decoder).  Two C calls, each enqueues a whole pass on the caller's stream:
//   ucod_vit_forward_train   image -> key map, saving what the backward needs (per layer: the two residual-stream states in
//                            f32, LN1 output + LoRA down-projection, qkv, attention output, log-sum-exp, fc1 pre-activation)
//   ucod_vit_backward        cotangent of the key map -> LoRA gradients of every layer (all other weights are frozen, so no
//                            weight-gradient GEMMs: dgrad only, ~2x the forward FLOPs with the recomputed attention scores)
// No allocation, no sync.  Operand formats of the LoRA "aug" columns: vit_train.hip.
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace {

struct TPlan {
  int M, tok, L;
  size_t x_in, x_mid, h_aug, qkv, att, lse, pre;           // per-layer arrays: base offset; stride below
  size_t s_x, s_h, s_qkv, s_att, s_lse, s_pre;
  size_t h2, g, patch, qscale;                             // forward transients
  size_t dx, dh, s, da, dqkv, delta, lgw;                  // backward scratch (dpre aliases g, s aliases h2)
  size_t total;
};

inline size_t up(size_t v) { return (v + 255) / 256 * 256; }

bool valid(const ucod_vit_train_desc* t) {
  if (!t) return false;
  const ucod_vit_desc* d = &t->vit;
  return d->B > 0 && d->C > 0 && d->P > 0 && d->H > 0 && d->W > 0 && d->H % d->P == 0 && d->W % d->P == 0 && d->D > 0 && d->heads > 0 &&
         d->D == d->heads * 64 && d->D % 128 == 0 && d->F % 128 == 0 && d->L >= 1 && d->Kpad % 64 == 0 && d->Kpad >= d->C * d->P * d->P &&
         t->lora_r >= 1 && 3 * t->lora_r <= UCOD_LORA_AUG && t->lora_dropout >= 0.f && t->lora_dropout < 1.f &&
         (d->resid16 == 0 || d->resid16 == 1);                   // resid16: the saved residual stream is IEEE fp16 (round 4: LayerNorm backward reads it)
}

TPlan make_plan(const ucod_vit_train_desc* t) {
  const ucod_vit_desc* d = &t->vit;
  TPlan p;
  const int gh = d->H / d->P, gw = d->W / d->P;
  p.tok = gh * gw + 1;
  p.M = d->B * p.tok;
  p.L = d->L;
  const size_t M = p.M, D = d->D, F = d->F, L = d->L;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += up(bytes); return r; };
  p.s_x = up(M * D * (d->resid16 ? 2 : 4));
  p.s_h = up(M * (D + UCOD_LORA_AUG) * 2);
  p.s_qkv = up(M * 3 * D * 2);
  p.s_att = up(M * D * 2);
  p.s_lse = up((size_t)d->B * d->heads * p.tok * 4);
  p.s_pre = up(M * F * 2);
  p.x_in = take(p.s_x * L);
  p.x_mid = take(p.s_x * (L - 1));
  p.h_aug = take(p.s_h * L);
  p.qkv = take(p.s_qkv * (L - 1));
  p.att = take(p.s_att * (L - 1));
  p.lse = take(p.s_lse * (L - 1));
  p.pre = take(p.s_pre * (L - 1));
  p.h2 = take(M * D * 2);
  p.g = take(M * F * 2);
  p.patch = take((size_t)d->B * gh * gw * d->Kpad * 2);
  p.qscale = take(3 * D * 4);
  p.dx = take(M * D * 4);
  p.dh = take(M * D * 4);
  p.s = p.h2;
  p.da = take(M * D * 2);
  p.dqkv = take(M * (3 * D + UCOD_LORA_AUG) * 2);
  p.delta = take((size_t)d->B * d->heads * p.tok * 4);
  p.lgw = take(ucod_lora_grad_workspace_bytes(d->D));
  p.total = o;
  return p;
}

}  // namespace

#define RUN(call)                \
  do {                           \
    int rc__ = (call);           \
    if (rc__ != 0) return rc__;  \
  } while (0)

extern "C" size_t ucod_vit_train_workspace_bytes(const ucod_vit_train_desc* t) { return valid(t) ? make_plan(t).total : 0; }

extern "C" int ucod_vit_forward_train(const ucod_vit_train_desc* t, const void* const* T, const void* const* TT, const float* img,
                                      float* key_out, void* workspace, size_t workspace_bytes, void* stream) {
  UCOD_BF16_ONLY();
  if (!valid(t) || !T || !TT || !img || !key_out || !workspace) return UCOD_EINVAL;
  const ucod_vit_desc* d = &t->vit;
  const TPlan p = make_plan(t);
  if (workspace_bytes < p.total) return UCOD_ENOMEM;
  char* ws = (char*)workspace;
  const int M = p.M, tok = p.tok, D = d->D, F = d->F, gv = d->gemm_variant, KA = D + UCOD_LORA_AUG;
  float* qscale = (float*)(ws + p.qscale);
  void* h2 = ws + p.h2;
  void* g = ws + p.g;
  void* patches = ws + p.patch;
  RUN(ucod_fill_qscale(qscale, D, 0.125f * 1.4426950408889634f, stream));

  const bool r16 = d->resid16 != 0;                               // fp16 residual stream (saved as such: half the bytes forward and backward)
  const int epi_patch = r16 ? UCOD_EPI_PATCH_TOKENS_H16 : UCOD_EPI_PATCH_TOKENS_F32;
  const int epi_resid = r16 ? UCOD_EPI_BIAS_SCALE_RESID_H16 : UCOD_EPI_BIAS_SCALE_RESID_F32;
  float* x0 = (float*)(ws + p.x_in);
  RUN(ucod_patch_im2col(img, patches, d->B, d->C, d->H, d->W, d->P, d->Kpad, stream));
  RUN(ucod_gemm_bf16(epi_patch, patches, T[0], x0, d->B * (tok - 1), D, d->Kpad, (const float*)T[1], nullptr, nullptr,
                     (const float*)T[3], tok, gv, stream));
  if (r16) RUN(ucod_cls_rows_h16(x0, (const float*)T[2], (const float*)T[3], d->B, tok, D, stream));
  else RUN(ucod_cls_rows(x0, (const float*)T[2], (const float*)T[3], d->B, tok, D, stream));

  for (int l = 0; l < d->L; ++l) {
    const void* const* W = T + 4 + UCOD_VIT_LAYER_STRIDE * l;
    const void* const* X = TT + UCOD_VIT_TRAIN_STRIDE * l;
    const bool last = (l == d->L - 1);
    float* x_in = (float*)(ws + p.x_in + p.s_x * l);
    void* h_aug = ws + p.h_aug + p.s_h * l;
    const ucod_lora_dropout drop{t->lora_dropout, t->seed, l};
    if (r16) RUN(ucod_layernorm_lora_h16(x_in, (const float*)W[0], (const float*)W[1], (const float*)X[5], t->lora_r, h_aug, M, D, d->eps,
                                         t->lora_dropout > 0.f ? &drop : nullptr, stream));
    else RUN(ucod_layernorm_lora(x_in, (const float*)W[0], (const float*)W[1], (const float*)X[5], t->lora_r, h_aug, M, D, d->eps,
                                 t->lora_dropout > 0.f ? &drop : nullptr, stream));
    if (last) {   // key hook: K rows of the augmented qkv weight; [B,D,h,w] out
      const char* wk = (const char*)X[0] + (size_t)D * KA * 2;
      RUN(ucod_gemm_bf16(UCOD_EPI_KEY_NCHW_F32, wk, h_aug, key_out, D, M, KA, (const float*)W[3] + D, nullptr, nullptr, nullptr, tok, gv, stream));
      break;
    }
    float* x_mid = (float*)(ws + p.x_mid + p.s_x * l);
    float* x_next = (float*)(ws + p.x_in + p.s_x * (l + 1));
    void* qkv = ws + p.qkv + p.s_qkv * l;
    void* att = ws + p.att + p.s_att * l;
    float* lse = (float*)(ws + p.lse + p.s_lse * l);
    void* pre = ws + p.pre + p.s_pre * l;
    RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_BF16, h_aug, X[0], qkv, M, 3 * D, KA, (const float*)W[3], qscale, nullptr, nullptr, tok, gv, stream));
    RUN(ucod_attention_fwd_lse(qkv, att, lse, d->B, tok, d->heads, stream));
    RUN(ucod_gemm_bf16(epi_resid, att, W[4], x_mid, M, D, D, (const float*)W[5], (const float*)W[6], x_in, nullptr, tok, gv, stream));
    if (r16) RUN(ucod_layernorm_h16(x_mid, (const float*)W[7], (const float*)W[8], h2, M, D, d->eps, stream));
    else RUN(ucod_layernorm(x_mid, (const float*)W[7], (const float*)W[8], h2, M, D, d->eps, 0, stream));
    RUN(ucod_gemm_bf16_train(UCOD_EPI_BIAS_GELU_SAVE_BF16, h2, W[9], g, M, F, D, (const float*)W[10], nullptr, pre, gv, stream));
    RUN(ucod_gemm_bf16(epi_resid, g, W[11], x_next, M, D, F, (const float*)W[12], (const float*)W[13], x_mid, nullptr, tok, gv, stream));
  }
  return UCOD_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// No-grad pass of the LoRA backbone: what the EMA teacher of models/modules/full_model.py:84,108-111 runs (backbone_ema under
// torch.no_grad()).  Same arithmetic as ucod_vit_forward_train -- LoRA as 64 extra K columns of the QKV / key GEMMs, dropout masks
// from the same counter hash -- but nothing is saved: one residual buffer updated in place, plain GELU epilogue, no log-sum-exp, and
// the residual stream may be IEEE fp16 (vit.resid16) like the frozen-backbone pass.  Workspace ~ the inference pass's.
namespace {
struct IPlan {
  size_t x, h_aug, h2, qkv, a, g, patch, qscale, total;
  int M, tok;
};
bool valid_infer(const ucod_vit_train_desc* t) {
  if (!t) return false;
  const ucod_vit_desc* d = &t->vit;
  return d->B > 0 && d->C > 0 && d->P > 0 && d->H > 0 && d->W > 0 && d->H % d->P == 0 && d->W % d->P == 0 && d->D > 0 && d->heads > 0 &&
         d->D == d->heads * 64 && d->D % 128 == 0 && d->F % 128 == 0 && d->L >= 1 && d->Kpad % 64 == 0 && d->Kpad >= d->C * d->P * d->P &&
         t->lora_r >= 1 && 3 * t->lora_r <= UCOD_LORA_AUG && t->lora_dropout >= 0.f && t->lora_dropout < 1.f && (d->resid16 == 0 || d->resid16 == 1);
}
IPlan make_iplan(const ucod_vit_train_desc* t) {
  const ucod_vit_desc* d = &t->vit;
  IPlan p;
  const int gh = d->H / d->P, gw = d->W / d->P;
  p.tok = gh * gw + 1;
  p.M = d->B * p.tok;
  const size_t M = p.M, D = d->D, F = d->F;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += up(bytes); return r; };
  p.x = take(M * D * (d->resid16 ? 2 : 4));
  p.h_aug = take(M * (D + UCOD_LORA_AUG) * 2);
  p.h2 = take(M * D * 2);
  p.qkv = take(M * 3 * D * 2);
  p.a = take(M * D * 2);
  p.g = take(M * F * 2);
  p.patch = take((size_t)d->B * gh * gw * d->Kpad * 2);
  p.qscale = take(3 * D * 4);
  p.total = o;
  return p;
}
}  // namespace

extern "C" size_t ucod_vit_lora_infer_workspace_bytes(const ucod_vit_train_desc* t) { return valid_infer(t) ? make_iplan(t).total : 0; }

extern "C" int ucod_vit_forward_lora_infer(const ucod_vit_train_desc* t, const void* const* T, const void* const* TT, const float* img,
                                           float* key_out, void* workspace, size_t workspace_bytes, void* stream) {
  UCOD_BF16_ONLY();
  if (!valid_infer(t) || !T || !TT || !img || !key_out || !workspace) return UCOD_EINVAL;
  const ucod_vit_desc* d = &t->vit;
  const IPlan p = make_iplan(t);
  if (workspace_bytes < p.total) return UCOD_ENOMEM;
  char* ws = (char*)workspace;
  const int M = p.M, tok = p.tok, D = d->D, F = d->F, gv = d->gemm_variant, KA = D + UCOD_LORA_AUG;
  const bool r16 = d->resid16 != 0;
  float* x = (float*)(ws + p.x);
  void* h_aug = ws + p.h_aug;
  void* h2 = ws + p.h2;
  void* qkv = ws + p.qkv;
  void* a = ws + p.a;
  void* g = ws + p.g;
  void* patches = ws + p.patch;
  float* qscale = (float*)(ws + p.qscale);
  const int epi_patch = r16 ? UCOD_EPI_PATCH_TOKENS_H16 : UCOD_EPI_PATCH_TOKENS_F32;
  const int epi_resid = r16 ? UCOD_EPI_BIAS_SCALE_RESID_H16 : UCOD_EPI_BIAS_SCALE_RESID_F32;
  RUN(ucod_fill_qscale(qscale, D, 0.125f * 1.4426950408889634f, stream));
  RUN(ucod_patch_im2col(img, patches, d->B, d->C, d->H, d->W, d->P, d->Kpad, stream));
  RUN(ucod_gemm_bf16(epi_patch, patches, T[0], x, d->B * (tok - 1), D, d->Kpad, (const float*)T[1], nullptr, nullptr, (const float*)T[3], tok, gv, stream));
  if (r16) RUN(ucod_cls_rows_h16(x, (const float*)T[2], (const float*)T[3], d->B, tok, D, stream));
  else RUN(ucod_cls_rows(x, (const float*)T[2], (const float*)T[3], d->B, tok, D, stream));
  for (int l = 0; l < d->L; ++l) {
    const void* const* W = T + 4 + UCOD_VIT_LAYER_STRIDE * l;
    const void* const* X = TT + UCOD_VIT_TRAIN_STRIDE * l;
    const ucod_lora_dropout drop{t->lora_dropout, t->seed, l};
    const ucod_lora_dropout* dp = t->lora_dropout > 0.f ? &drop : nullptr;
    if (r16) RUN(ucod_layernorm_lora_h16(x, (const float*)W[0], (const float*)W[1], (const float*)X[5], t->lora_r, h_aug, M, D, d->eps, dp, stream));
    else RUN(ucod_layernorm_lora(x, (const float*)W[0], (const float*)W[1], (const float*)X[5], t->lora_r, h_aug, M, D, d->eps, dp, stream));
    if (l == d->L - 1) {
      const char* wk = (const char*)X[0] + (size_t)D * KA * 2;
      RUN(ucod_gemm_bf16(UCOD_EPI_KEY_NCHW_F32, wk, h_aug, key_out, D, M, KA, (const float*)W[3] + D, nullptr, nullptr, nullptr, tok, gv, stream));
      break;
    }
    RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_BF16, h_aug, X[0], qkv, M, 3 * D, KA, (const float*)W[3], qscale, nullptr, nullptr, tok, gv, stream));
    RUN(ucod_attention_fwd(qkv, a, d->B, tok, d->heads, 0.f, 0, stream));
    RUN(ucod_gemm_bf16(epi_resid, a, W[4], x, M, D, D, (const float*)W[5], (const float*)W[6], x, nullptr, tok, gv, stream));
    if (r16) RUN(ucod_layernorm_h16(x, (const float*)W[7], (const float*)W[8], h2, M, D, d->eps, stream));
    else RUN(ucod_layernorm(x, (const float*)W[7], (const float*)W[8], h2, M, D, d->eps, 0, stream));
    RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_GELU_BF16, h2, W[9], g, M, F, D, (const float*)W[10], nullptr, nullptr, nullptr, tok, gv, stream));
    RUN(ucod_gemm_bf16(epi_resid, g, W[11], x, M, D, F, (const float*)W[12], (const float*)W[13], x, nullptr, tok, gv, stream));
  }
  return UCOD_OK;
}

extern "C" int ucod_vit_backward(const ucod_vit_train_desc* t, const void* const* T, const void* const* TT, const float* dkey,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  UCOD_BF16_ONLY();
  if (!valid(t) || !T || !TT || !dkey || !workspace) return UCOD_EINVAL;
  const ucod_vit_desc* d = &t->vit;
  const TPlan p = make_plan(t);
  if (workspace_bytes < p.total) return UCOD_ENOMEM;
  char* ws = (char*)workspace;
  const int M = p.M, tok = p.tok, D = d->D, F = d->F, gv = d->gemm_variant, KQ = 3 * D + UCOD_LORA_AUG, r = t->lora_r;
  float* dx = (float*)(ws + p.dx);
  float* dh = (float*)(ws + p.dh);
  void* s = ws + p.s;
  void* dpre = ws + p.g;
  void* da = ws + p.da;
  void* dqkv = ws + p.dqkv;
  float* delta = (float*)(ws + p.delta);
  void* lgw = ws + p.lgw;
  const size_t lgw_bytes = ucod_lora_grad_workspace_bytes(D);

  // dgrad outputs that feed a LayerNorm backward (dh) are written as bf16 (UCOD_DGRAD_F32=1: f32, the round-3 path): the GEMM's drain and the
  // LayerNorm backward's read side move half the bytes; the residual cotangent stream (dx) stays f32
  static const bool dgrad16_env = !(ucod::lab_env("UCOD_DGRAD_F32") && ucod::lab_env("UCOD_DGRAD_F32")[0] == '1');
  const bool r16 = d->resid16 != 0;                               // the saved residual stream is fp16 (then the dgrad outputs are bf16 whatever the variable says)
  const bool dgrad16 = dgrad16_env || r16;
  const int epi_dh = dgrad16 ? UCOD_EPI_BIAS_BF16 : UCOD_EPI_BIAS_F32;
  const int lnb_flags = (dgrad16 ? UCOD_LNB_DY_BF16 : 0) | (r16 ? UCOD_LNB_X_F16 : 0);
  auto ln_bwd_plain = [&](const void* dy, const void* x, const float* gam, const float* dres, const float* next_scale) -> int {
    return ucod_layernorm_bwd_ex(dy, x, lnb_flags, gam, dres, next_scale, dx, s, M, D, d->eps, stream);
  };
  auto qkv_side = [&](int l) -> int {   // dqkv_aug (k/q/v thirds filled) -> LoRA grads of layer l, dh = d LN1 output
    const void* const* X = TT + UCOD_VIT_TRAIN_STRIDE * l;
    const void* h_aug = ws + p.h_aug + p.s_h * l;
    const ucod_lora_dropout drop{t->lora_dropout, t->seed, l};
    RUN(ucod_lora_grad(dqkv, h_aug, (const float*)X[5], r, t->lora_scaling, (float*)X[6], 0, lgw, lgw_bytes, M, D,
                       t->lora_dropout > 0.f ? &drop : nullptr, stream));
    RUN(ucod_gemm_bf16(epi_dh, dqkv, X[1], dh, M, D, KQ, nullptr, nullptr, nullptr, nullptr, tok, gv, stream));
    return UCOD_OK;
  };

  // LayerNorm-1 backward of layer l: with dropout on, the masked LoRA branch t A is added here (it cannot ride on the dgrad GEMM)
  auto ln1_bwd = [&](int l, const void* dy, const void* x, const float* gam, const float* dres, const float* next_scale) -> int {
    if (t->lora_dropout > 0.f) {
      const ucod_lora_dropout drop{t->lora_dropout, t->seed, l};
      const float* lora_l = (const float*)(TT + UCOD_VIT_TRAIN_STRIDE * l)[5];
      return ucod_layernorm_bwd_lora_ex(dy, x, lnb_flags, gam, dres, next_scale, dx, s, M, D, d->eps, dqkv, lora_l, r, &drop, stream);
    }
    return ln_bwd_plain(dy, x, gam, dres, next_scale);
  };

  // last layer: only the key projection reaches the loss
  const int last = d->L - 1;
  RUN(ucod_key_grad_tokens(dkey, dqkv, d->B, tok, D, stream));
  RUN(qkv_side(last));
  if (last == 0) return UCOD_OK;
  {
    const void* const* W = T + 4 + UCOD_VIT_LAYER_STRIDE * last;
    const void* const* Wp = T + 4 + UCOD_VIT_LAYER_STRIDE * (last - 1);
    RUN(ln1_bwd(last, dh, (const float*)(ws + p.x_in + p.s_x * last), (const float*)W[0], nullptr, (const float*)Wp[13]));
  }
  for (int l = last - 1; l >= 0; --l) {
    const void* const* W = T + 4 + UCOD_VIT_LAYER_STRIDE * l;
    const void* const* X = TT + UCOD_VIT_TRAIN_STRIDE * l;
    const float* x_in = (const float*)(ws + p.x_in + p.s_x * l);
    const float* x_mid = (const float*)(ws + p.x_mid + p.s_x * l);
    const void* qkv = ws + p.qkv + p.s_qkv * l;
    const void* att = ws + p.att + p.s_att * l;
    const float* lse = (const float*)(ws + p.lse + p.s_lse * l);
    const void* pre = ws + p.pre + p.s_pre * l;
    // MLP branch: s = ls2 * dx  ->  fc2 dgrad (x gelu')  ->  fc1 dgrad  ->  LN2 backward + residual
    RUN(ucod_gemm_bf16_train(UCOD_EPI_GELU_BWD_BF16, s, X[4], dpre, M, F, D, nullptr, pre, nullptr, gv, stream));
    RUN(ucod_gemm_bf16(epi_dh, dpre, X[3], dh, M, D, F, nullptr, nullptr, nullptr, nullptr, tok, gv, stream));
    RUN(ln_bwd_plain(dh, x_mid, (const float*)W[7], dx, (const float*)W[6]));
    // attention branch: s = ls1 * dx  ->  out-proj dgrad  ->  attention backward  ->  LoRA grads + qkv dgrad  ->  LN1 backward
    RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_BF16, s, X[2], da, M, D, D, nullptr, nullptr, nullptr, nullptr, tok, gv, stream));
    RUN(ucod_attention_bwd(qkv, att, da, lse, delta, dqkv, KQ, d->B, tok, d->heads, stream));
    RUN(qkv_side(l));
    if (l > 0) {
      const void* const* Wp = T + 4 + UCOD_VIT_LAYER_STRIDE * (l - 1);
      RUN(ln1_bwd(l, dh, x_in, (const float*)W[0], dx, (const float*)Wp[13]));
    }
  }
  return UCOD_OK;
}
