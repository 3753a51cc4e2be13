// Discriminator with the feature branch (models/discriminator.py:77-95 with dis_use_features=True: featureConv 768 -> 768 3x3 + BN + LeakyReLU,
// concatenated with the mask branch, two stride-2 ConvBlocks on 800 / 400 channels, Linear on 200 * ((fs + 3) / 4)^2, sigmoid).  No shipped
// config enables it (configs/uscod/UCOD-DPL_dinov2.py:33), so this is the forward path only -- what TrainLoop.merge_pseudo_label needs from a frozen
// discriminator (loop_UCOD_DPL.py:257-272) -- built from generic pieces instead of the fused small-channel kernels of disc.hip:
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

__global__ __launch_bounds__(256) void bn_apply_kernel(float* __restrict__ y, const double* __restrict__ acc, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, int B, int C,
                                                       int HW, float eps, float momentum, float slope, int update) {
  // grid (cdiv(HW, 256), C, B)
  const int c = blockIdx.y, b = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  const double n = (double)B * HW;
  const double mean = acc[c] / n;
  double var = acc[C + c] / n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  if (i < HW) {
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    float* p = y + ((size_t)b * C + c) * HW + i;
    const float v = (*p - (float)mean) * rstd * gamma[c] + beta[c];
    *p = v > 0.f ? v : v * slope;
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
  hipLaunchKernelGGL(bn_apply_kernel, dim3(cdiv(HW, 256), C, B), dim3(256), 0, (hipStream_t)stream, y, acc, gamma, beta, running_mean, running_var, B, C, HW, eps,
                     momentum, slope, update_running);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_linear_sigmoid(const float* x, const float* w, const float* bias, float* out, int B, int K, void* stream) {
  if (!x || !w || !bias || !out || B <= 0 || K <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(linear_sigmoid_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, w, bias, out, K);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
