// Discriminator with the feature branch (models/discriminator.py:77-95 with dis_use_features=True: featureConv 768 -> 768 3x3 + BN + LeakyReLU,
// concatenated with the mask branch, two stride-2 ConvBlocks on 800 / 400 channels, Linear on 200 * ((fs + 3) / 4)^2, sigmoid).  No shipped
// config enables it (configs/uscod/UCOD-DPL_dinov2.py:33)): forward (what TrainLoop.merge_pseudo_label needs from a frozen discriminator,
// loop_UCOD_DPL.py:257-272) and backward (the discriminator phase, :230-255), built from generic pieces instead of the fused small-channel kernels of disc.hip:
//   ucod_unfold3x3        im2col of a 3x3 / pad 1 / stride s convolution: [B,C,H,W] -> [B, Kpad, Ho*Wo], row c*9 + ky*3 + kx (F.unfold order), rows
//                         C*9 .. Kpad-1 zero, so that the convolution IS ucod_dba_project (the exact-f32 MFMA GEMM) on W.reshape(O, C*9)
//   ucod_bn_lrelu_train   nn.BatchNorm2d in training mode (batch statistics over (B, H, W), eps 1e-5, running buffers updated with momentum 0.1 and
//                         the unbiased variance) + LeakyReLU(slope), in place on [B,C,HW]; statistics in f64, fixed reduction order
//   ucod_linear_sigmoid   sigmoid(x [B,K] . w [K] + b)
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {
namespace {

__global__ __launch_bounds__(256) void unfold3x3_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W, int Ho, int Wo, int stride,
                                                        int Kpad) {
  // grid (cdiv(Ho*Wo, 256), Kpad, B): one output row (c, ky, kx) of one image per blockIdx.y
  const int p = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y, b = blockIdx.z;
  if (p >= Ho * Wo) return;
  float v = 0.f;
  if (k < C * 9) {
    const int c = k / 9, t = k - c * 9, ky = t / 3, kx = t - ky * 3;
    const int oy = p / Wo, ox = p - oy * Wo;
    const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * C + c) * H + iy) * W + ix];
  }
  out[((size_t)b * Kpad + k) * (Ho * Wo) + p] = v;
}

// per-channel sum and sum of squares over (B, HW): one workgroup per channel, f64, fixed order
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, double* __restrict__ acc, int B, int C, int HW) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* row = y + ((size_t)b * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
      const double v = row[i];
      s += v;
      q += v * v;
    }
  }
  s = block_sum(s, red);
  q = block_sum(q, red);
  if (threadIdx.x == 0) {
    acc[c] = s;
    acc[C + c] = q;
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ yin, float* __restrict__ y, const double* __restrict__ acc,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
                                                       float* __restrict__ rvar, int B, int C, int HW, float eps, float momentum, float slope, int update) {
  // grid (cdiv(HW, 256), C, B)
  const int c = blockIdx.y, b = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  const double n = (double)B * HW;
  const double mean = acc[c] / n;
  double var = acc[C + c] / n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  if (i < HW) {
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const size_t at = ((size_t)b * C + c) * HW + i;
    const float v = (yin[at] - (float)mean) * rstd * gamma[c] + beta[c];
    y[at] = v > 0.f ? v : v * slope;
  }
  if (update && b == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

__global__ __launch_bounds__(256) void linear_sigmoid_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ out, int K) {
  __shared__ double red[16];
  const int b = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < K; i += 256) acc += (double)x[(size_t)b * K + i] * (double)w[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[b] = sigmoid_acc((float)acc + bias[0]);
}

// ---- backward pieces (the discriminator phase with dis_use_features=True, loop_UCOD_DPL.py:230-255) ------------------------------------------
// BatchNorm2d (training mode) + LeakyReLU backward.  With xh = (y - mean) * rstd, v = gamma * xh + beta, g' = g * (v > 0 ? 1 : slope):
//   dbeta = sum g',  dgamma = sum g' * xh,  dy = gamma * rstd * (g' - dbeta / n - xh * dgamma / n).   Sums in f64, fixed order.
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ y, const float* __restrict__ g, const double* __restrict__ acc,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, double* __restrict__ red_out,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C, int HW, float eps, float slope,
                                                           int accumulate) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  const double n = (double)B * HW;
  const double mean = acc[c] / n;
  double var = acc[C + c] / n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps)), gm = gamma[c], bt = beta[c], mf = (float)mean;
  double sb = 0.0, sg = 0.0;
  for (int b = 0; b < B; ++b) {
    const size_t row = ((size_t)b * C + c) * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float xh = (y[row + i] - mf) * rstd;
      const float v = gm * xh + bt;
      const float gp = g[row + i] * (v > 0.f ? 1.f : slope);
      sb += gp;
      sg += (double)gp * xh;
    }
  }
  sb = block_sum(sb, red);
  sg = block_sum(sg, red);
  if (threadIdx.x == 0) {
    red_out[c] = sb;
    red_out[C + c] = sg;
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sb;
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sg;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ g, float* __restrict__ gy,
                                                           const double* __restrict__ acc, const double* __restrict__ red, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, int B, int C, int HW, float eps, float slope) {
  const int c = blockIdx.y, b = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  const double n = (double)B * HW;
  const double mean = acc[c] / n;
  double var = acc[C + c] / n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps)), gm = gamma[c];
  const size_t at = ((size_t)b * C + c) * HW + i;
  const float xh = (y[at] - (float)mean) * rstd;
  const float v = gm * xh + beta[c];
  const float gp = g[at] * (v > 0.f ? 1.f : slope);
  gy[at] = gm * rstd * (gp - (float)(red[c] / n) - xh * (float)(red[C + c] / n));
}

// col2im of the 3x3 / pad 1 / stride s unfold: gx[b][c][iy][ix] = sum over the (ky, kx, oy, ox) with oy * s + ky - 1 == iy, ox * s + kx - 1 == ix
__global__ __launch_bounds__(256) void fold3x3_kernel(const float* __restrict__ gcols, float* __restrict__ gx, int C, int H, int W, int Ho, int Wo, int stride,
                                                      int Kpad) {
  const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (p >= H * W) return;
  const int iy = p / W, ix = p - iy * W;
  float acc = 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || (ty % stride) != 0) continue;
    const int oy = ty / stride;
    if (oy >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || (tx % stride) != 0) continue;
      const int ox = tx / stride;
      if (ox >= Wo) continue;
      acc += gcols[((size_t)b * Kpad + c * 9 + ky * 3 + kx) * (Ho * Wo) + oy * Wo + ox];
    }
  }
  gx[((size_t)b * C + c) * (H * W) + p] = acc;
}

// Linear + sigmoid backward: gz[b] = gprob[b] * p (1 - p);  gx[b][k] = gz[b] w[k];  gw[k] (+)= sum_b gz[b] x[b][k];  gb (+)= sum_b gz[b]
__global__ __launch_bounds__(256) void linear_sigmoid_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ prob,
                                                                 const float* __restrict__ gprob, float* __restrict__ gx, float* __restrict__ gw,
                                                                 float* __restrict__ gb, int B, int K, int accumulate) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  float sw = 0.f, sb = 0.f;
  for (int b = 0; b < B; ++b) {
    const float p = prob[b];
    const float gz = gprob[b] * p * (1.f - p);
    sb += gz;
    if (k < K) {
      gx[(size_t)b * K + k] = gz * w[k];
      sw += gz * x[(size_t)b * K + k];
    }
  }
  if (k < K) gw[k] = (accumulate ? gw[k] : 0.f) + sw;
  if (k == 0) gb[0] = (accumulate ? gb[0] : 0.f) + sb;
}

// nn.BCELoss(cat(probs_student, probs_pseudo), [0 .. 0, 1 .. 1]) of the discriminator phase (loop_UCOD_DPL.py:246-247): the mean over 2B and its
// gradient with respect to the probabilities.  torch clamps the logarithms at -100 and divides by max(p (1 - p), 1e-12) in the backward
// (aten/src/ATen/native/cuda/Loss.cu); `inv` = 1 / (2 B world).  One workgroup.
__global__ __launch_bounds__(256) void disc_bce_kernel(const float* __restrict__ ps, const float* __restrict__ pp, float* __restrict__ g_s,
                                                       float* __restrict__ g_p, float* __restrict__ loss, int B, float inv) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float s = ps[b], p = pp[b];
    g_s[b] = s / fmaxf(s * (1.f - s), 1e-12f) * inv;
    g_p[b] = (p - 1.f) / fmaxf(p * (1.f - p), 1e-12f) * inv;
    acc -= (double)fmaxf(logf(1.f - s), -100.f) + (double)fmaxf(logf(p), -100.f);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = (float)(acc / (2.0 * B));
}

}  // namespace
}  // namespace ucod

using namespace ucod;

extern "C" int ucod_unfold3x3(const float* x, float* out, int B, int C, int H, int W, int stride, int Kpad, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || Kpad < C * 9 || Kpad > 65535) return UCOD_EINVAL;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  hipLaunchKernelGGL(unfold3x3_kernel, dim3(cdiv((long)Ho * Wo, 256), Kpad, B), dim3(256), 0, (hipStream_t)stream, x, out, C, H, W, Ho, Wo, stride, Kpad);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" size_t ucod_bn_lrelu_workspace_bytes(int C) { return C > 0 ? (size_t)2 * C * sizeof(double) : 0; }

extern "C" int ucod_bn_lrelu_train(float* y, const float* gamma, const float* beta, float* running_mean, float* running_var, int B, int C, int HW, float eps,
                                   float momentum, float slope, int update_running, void* workspace, size_t workspace_bytes, void* stream) {
  if (!y || !gamma || !beta || !running_mean || !running_var || !workspace || B <= 0 || C <= 0 || C > 65535 || HW <= 0) return UCOD_EINVAL;
  if (workspace_bytes < ucod_bn_lrelu_workspace_bytes(C)) return UCOD_ENOMEM;
  double* acc = (double*)workspace;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, y, acc, B, C, HW);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(cdiv(HW, 256), C, B), dim3(256), 0, (hipStream_t)stream, y, y, acc, gamma, beta, running_mean, running_var, B, C, HW, eps,
                     momentum, slope, update_running);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_bn_lrelu_train_save(const float* y_pre, float* y_out, const float* gamma, const float* beta, float* running_mean, float* running_var, int B,
                                        int C, int HW, float eps, float momentum, float slope, int update_running, void* stats, size_t stats_bytes, void* stream) {
  if (!y_pre || !y_out || !gamma || !beta || !running_mean || !running_var || !stats || B <= 0 || C <= 0 || C > 65535 || HW <= 0) return UCOD_EINVAL;
  if (stats_bytes < ucod_bn_lrelu_workspace_bytes(C)) return UCOD_ENOMEM;
  double* acc = (double*)stats;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, y_pre, acc, B, C, HW);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(cdiv(HW, 256), C, B), dim3(256), 0, (hipStream_t)stream, y_pre, y_out, acc, gamma, beta, running_mean, running_var, B, C, HW,
                     eps, momentum, slope, update_running);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_bn_lrelu_bwd(const float* y_pre, const float* gout, float* gy_pre, const void* stats, const float* gamma, const float* beta, float* dgamma,
                                 float* dbeta, int B, int C, int HW, float eps, float slope, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (!y_pre || !gout || !gy_pre || !stats || !gamma || !beta || !dgamma || !dbeta || !workspace || B <= 0 || C <= 0 || C > 65535 || HW <= 0) return UCOD_EINVAL;
  if (workspace_bytes < ucod_bn_lrelu_workspace_bytes(C)) return UCOD_ENOMEM;
  double* red = (double*)workspace;
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, y_pre, gout, (const double*)stats, gamma, beta, red, dgamma, dbeta, B, C, HW, eps,
                     slope, accumulate);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cdiv(HW, 256), C, B), dim3(256), 0, (hipStream_t)stream, y_pre, gout, gy_pre, (const double*)stats, red, gamma, beta, B, C,
                     HW, eps, slope);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_fold3x3(const float* gcols, float* gx, int B, int C, int H, int W, int stride, int Kpad, void* stream) {
  if (!gcols || !gx || B <= 0 || C <= 0 || C > 65535 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || Kpad < C * 9) return UCOD_EINVAL;
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  hipLaunchKernelGGL(fold3x3_kernel, dim3(cdiv((long)H * W, 256), C, B), dim3(256), 0, (hipStream_t)stream, gcols, gx, C, H, W, Ho, Wo, stride, Kpad);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_linear_sigmoid_bwd(const float* x, const float* w, const float* prob, const float* gprob, float* gx, float* gw, float* gb, int B, int K,
                                       int accumulate, void* stream) {
  if (!x || !w || !prob || !gprob || !gx || !gw || !gb || B <= 0 || K <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(linear_sigmoid_bwd_kernel, dim3(cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, x, w, prob, gprob, gx, gw, gb, B, K, accumulate);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_disc_bce(const float* probs_student, const float* probs_pseudo, float* g_student, float* g_pseudo, float* loss, int B, float inv, void* stream) {
  if (!probs_student || !probs_pseudo || !g_student || !g_pseudo || !loss || B <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(disc_bce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, probs_student, probs_pseudo, g_student, g_pseudo, loss, B, inv);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_linear_sigmoid(const float* x, const float* w, const float* bias, float* out, int B, int K, void* stream) {
  if (!x || !w || !bias || !out || B <= 0 || K <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(linear_sigmoid_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, K);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
