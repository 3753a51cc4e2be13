// Small streaming kernels of the decoder / APM path: bilinear resize (A1), APM fusion + both BCE losses + their
// gradients in one pass (A5+A6), thresholding, fused AdamW + EMA over a flat parameter arena (A7).
// These move < 2 MB per step (except the feature resize) and are launch-latency bound: each is one launch.
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

// ATen upsample_bilinear2d, align_corners=False: src = scale*(dst+0.5)-0.5 evaluated as ONE fma (both the ATen
// CPU build and the GPU compilers contract it; it decides lambda==0.5 ties under the >0.5 thresholds of
// loop_UCOD_DPL.py:241,261), clamped at 0; i1 = i0 + (i0 < in-1).
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = fmaf(scale, (float)dst + 0.5f, -0.5f);
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = i0 < in_size - 1 ? i0 : in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, long planes, int ih,
                                                       int iw, int oh, int ow, float sh, float sw) {
  const long total = planes * oh * ow;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int x = (int)(idx % ow);
    const int y = (int)((idx / ow) % oh);
    const long pl = idx / ((long)ow * oh);
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(y, sh, ih, y0, y1, ly);
    src_index(x, sw, iw, x0, x1, lx);
    const float* p = in + pl * ih * iw;
    const float v00 = p[y0 * iw + x0], v01 = p[y0 * iw + x1], v10 = p[y1 * iw + x0], v11 = p[y1 * iw + x1];
    const float top = v00 * (1.f - lx) + v01 * lx;
    const float bot = v10 * (1.f - lx) + v11 * lx;
    out[idx] = top * (1.f - ly) + bot * ly;
  }
}

// Adjoint (transpose) of bilinear_kernel: gin[plane][iy][ix] = sum over the output pixels whose taps touch (iy,ix) of
// weight * gout.  Gather form (no atomics): the candidate output range of a source index is scanned with the SAME
// src_index() as the forward, so forward and adjoint agree tap for tap.  Used to pull the decoder's gradient back from the
// 68x68 grid to the backbone's native 37x37 grid, where the 1x1-conv weight gradient is 3.4x cheaper (the conv and the
// resize commute: both are linear, one acts on channels, the other on pixels).
// taps of source index i along one axis: the (at most MAXT) output indices whose bilinear footprint touches i, with
// their weights.  Scans the candidate range with the SAME src_index() as the forward.
constexpr int MAXT = 8;
__device__ __forceinline__ int adjoint_taps(int i, float scale, int in_size, int out_size, int (&idx)[MAXT], float (&w)[MAXT]) {
  int lo = (int)floorf((float)(i - 1) / scale) - 1, hi = (int)ceilf((float)(i + 1) / scale) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > out_size - 1 ? out_size - 1 : hi;
  int n = 0;
  for (int o = lo; o <= hi; ++o) {
    int i0, i1;
    float l1;
    src_index(o, scale, in_size, i0, i1, l1);
    float wt = 0.f;
    if (i0 == i) wt += 1.f - l1;
    if (i1 == i) wt += l1;
    if (wt != 0.f && n < MAXT) { idx[n] = o; w[n] = wt; ++n; }
  }
  return n;
}

// grid (cdiv(ih*iw,256), plane-chunks): each thread owns one source pixel, computes its taps ONCE and then sweeps planes.
__global__ __launch_bounds__(256) void bilinear_adjoint_kernel(const float* __restrict__ gout, float* __restrict__ gin, long planes,
                                                               int ih, int iw, int oh, int ow, float sh, float sw) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= ih * iw) return;
  const int iy = p / iw, ix = p - iy * iw;
  int yi[MAXT], xi[MAXT];
  float yw[MAXT], xw[MAXT];
  const int ny = adjoint_taps(iy, sh, ih, oh, yi, yw);
  const int nx = adjoint_taps(ix, sw, iw, ow, xi, xw);
  for (long pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    const float* g = gout + pl * oh * ow;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < MAXT; ++a) {
      if (a < ny) {
        float row = 0.f;
#pragma unroll
        for (int b = 0; b < MAXT; ++b)
          if (b < nx) row = fmaf(xw[b], g[yi[a] * ow + xi[b]], row);
        acc = fmaf(yw[a], row, acc);
      }
    }
    gin[pl * ih * iw + p] = acc;
  }
}

__global__ __launch_bounds__(256) void binarize_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n, int logits) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i];
    out[i] = (logits ? (sigmoid_acc(v) > 0.5f) : (v > 0.5f)) ? 1.f : 0.f;
  }
}

__device__ __forceinline__ float bce_logits(float x, float t) {
  return (1.f - t) * x + fmaxf(-x, 0.f) + log1pf(expf(-fabsf(x)));
}

// grid (cdiv(HW,256), B)
__global__ __launch_bounds__(256) void apm_bce_kernel(const float* __restrict__ pl, const float* __restrict__ teacher,
                                                      const float* __restrict__ fg, const float* __restrict__ bg,
                                                      const float* __restrict__ p_s, const float* __restrict__ p_p, float epoch_frac,
                                                      float gscale, float* __restrict__ wout, float* __restrict__ merged,
                                                      float* __restrict__ gfg, float* __restrict__ gbg, float* __restrict__ losses,
                                                      int B, int HW) {
  __shared__ float red[16];
  const int b = blockIdx.y, tid = threadIdx.x;
  const float ps = p_s[b], pp = p_p[b];
  float w = 0.5f * (1.f + cosf(fabsf(ps - pp) * 3.14159265358979323846f)) + epoch_frac;
  w = fminf(fmaxf(w, 0.f), 1.f);
  const int p = blockIdx.x * 256 + tid;
  float l1 = 0.f, l2 = 0.f;
  const float inv_n = 1.f / ((float)B * (float)HW);
  if (p < HW) {
    const long i = (long)b * HW + p;
    const float pt = sigmoid_acc(teacher[i]) > 0.5f ? 1.f : 0.f;
    const float t = pl[i] * (1.f - w) + pt * w;
    merged[i] = t;
    const float xf = fg[i], xb = bg[i];
    l1 = bce_logits(xf, t);
    l2 = bce_logits(xb, 1.f - t);
    gfg[i] = (sigmoid_acc(xf) - t) * inv_n * gscale;
    gbg[i] = (sigmoid_acc(xb) - (1.f - t)) * inv_n * gscale;
  }
  l1 = block_sum(l1, red);
  l2 = block_sum(l2, red);
  if (tid == 0) {
    atomicAdd(&losses[0], l1 * inv_n);
    atomicAdd(&losses[1], l2 * inv_n);
    if (blockIdx.x == 0) {
      wout[b] = w;
      atomicAdd(&losses[2], -fmaxf(logf(1.f - ps), -100.f) / (float)B);  // BCELoss(p_s, 0), torch clamps log at -100
    }
  }
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ ema, size_t n, float decay,
                                                        float beta1, float beta2, float step_size, float bc2_sqrt, float eps,
                                                        float alpha) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i];
    float pi = p[i] * decay;
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);            // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;         // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
    if (ema) ema[i] = ema[i] * alpha + (1.f - alpha) * pi;           // loop_UCOD_DPL.py:189
  }
}

static inline int nblocks(size_t n, int cap) { return (int)((n + 255) / 256 < (size_t)cap ? (n + 255) / 256 : (size_t)cap); }

}  // namespace ucod

using namespace ucod;

extern "C" int ucod_bilinear_resize(const float* in, float* out, int planes, int ih, int iw, int oh, int ow, void* stream) {
  if (!in || !out || planes <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0) return UCOD_EINVAL;
  const size_t total = (size_t)planes * oh * ow;
  const float sh = (float)ih / (float)oh, sw = (float)iw / (float)ow;   // area_pixel_compute_scale<float>
  UCOD_PROF(PROF_BILINEAR, stream);
  hipLaunchKernelGGL(bilinear_kernel, dim3(nblocks(total, 16384)), dim3(256), 0, (hipStream_t)stream, in, out, (long)planes, ih, iw, oh, ow, sh, sw);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_bilinear_resize_adjoint(const float* gout, float* gin, int planes, int ih, int iw, int oh, int ow, void* stream) {
  if (!gout || !gin || planes <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0) return UCOD_EINVAL;
  const size_t total = (size_t)planes * ih * iw;
  const float sh = (float)ih / (float)oh, sw = (float)iw / (float)ow;
  UCOD_PROF(PROF_BILINEAR, stream);
  (void)total;
  if (ceilf(2.f / fminf(sh, sw)) + 1.f > (float)MAXT) return UCOD_EINVAL;   // more than MAXT taps per axis (upsampling beyond ~3.5x)
  const int py = planes < 1024 ? planes : 1024;
  hipLaunchKernelGGL(bilinear_adjoint_kernel, dim3(cdiv((long)ih * iw, 256), py), dim3(256), 0, (hipStream_t)stream, gout, gin, (long)planes, ih, iw, oh, ow, sh, sw);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_binarize(const float* x, float* out, size_t n, int logits, void* stream) {
  if (!x || !out) return UCOD_EINVAL;
  if (n == 0) return UCOD_OK;
  UCOD_PROF(PROF_BINARIZE, stream);
  hipLaunchKernelGGL(binarize_kernel, dim3(nblocks(n, 4096)), dim3(256), 0, (hipStream_t)stream, x, out, n, logits);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_apm_bce(const float* pl, const float* teacher, const float* fg, const float* bg, const float* p_s,
                            const float* p_p, float epoch_frac, float gscale, float* w, float* merged, float* gfg, float* gbg,
                            float* losses, int B, int HW, void* stream) {
  if (!pl || !teacher || !fg || !bg || !p_s || !p_p || !w || !merged || !gfg || !gbg || !losses || B <= 0 || HW <= 0) return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  UCOD_PROF(PROF_APM, s);
  hipError_t e = hipMemsetAsync(losses, 0, 4 * sizeof(float), s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(apm_bce_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, s, pl, teacher, fg, bg, p_s, p_p, epoch_frac, gscale, w, merged, gfg, gbg, losses, B, HW);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float ema_alpha, void* stream) {
  if (!p || !g || !m || !v || step <= 0) return UCOD_EINVAL;
  if (n == 0) return UCOD_OK;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  UCOD_PROF(PROF_ADAMW, stream);
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(nblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, n, decay, beta1, beta2, step_size, bc2_sqrt, eps, ema_alpha);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
