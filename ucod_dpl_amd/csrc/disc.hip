// APM discriminator (models/discriminator.py:60-70,86-95; dis_use_features=False):
//   ConvBlock = conv3x3(no bias) + BatchNorm2d (ALWAYS train mode: batch statistics, running buffers mutate on
//   every call -- loop_UCOD_DPL.py:136 never puts it in eval) + LeakyReLU(0.1);
//   mask[B,1,fs,fs] -> 32@fs -> 16@fs/2 (stride 2) -> 8@fs/4 (stride 2) -> flatten -> Linear -> sigmoid.
// 14 MFLOP per image: launch-latency bound.  Direct convolution, one thread per output pixel computing every
// output channel (weights broadcast from LDS); BatchNorm's batch-wide barrier is the kernel boundary: each
// conv writes its pre-BN output, a statistics kernel reduces per channel in f64, and the NEXT kernel applies
// normalise + affine + LeakyReLU on load.  The pre-BN activations and statistics are the `saved` state the
// backward pass (discriminator phase, loop_UCOD_DPL.py:230-255) consumes.
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

constexpr float BN_EPS = 1e-5f, BN_MOM = 0.1f, LRELU = 0.1f;

struct DiscDims {
  int B, s1, s2, s3;          // spatial sizes of the three conv outputs
  size_t n1, n2, n3;          // element counts of y1,y2,y3
};
static inline DiscDims disc_dims(int B, int fs) {
  DiscDims d;
  d.B = B;
  d.s1 = fs;
  d.s2 = (fs - 1) / 2 + 1;
  d.s3 = (d.s2 - 1) / 2 + 1;
  d.n1 = (size_t)B * 32 * d.s1 * d.s1;
  d.n2 = (size_t)B * 16 * d.s2 * d.s2;
  d.n3 = (size_t)B * 8 * d.s3 * d.s3;
  return d;
}
// saved layout (floats): y1 | y2 | y3 | stats[2*(32+16+8)] (mean, rstd per channel, layer after layer)
static inline size_t stats_off(const DiscDims& d) { return d.n1 + d.n2 + d.n3; }

__device__ __forceinline__ float bn_lrelu(float y, float mean, float rstd, float g, float b) {
  const float h = (y - mean) * rstd * g + b;
  return h >= 0.f ? h : h * LRELU;
}

// in: [B,CIN,IH,IW] (pre-BN output of the previous block, or the raw mask when !BN_IN); out: [B,COUT,OH,OW] pre-BN
template <int CIN, int COUT, int STRIDE, bool BN_IN>
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                      const float* __restrict__ in_stats, const float* __restrict__ in_g,
                                                      const float* __restrict__ in_b, float* __restrict__ out, int B, int IH,
                                                      int OH) {
  __shared__ __attribute__((aligned(16))) float ws[CIN * 9 * COUT];  // [ci][tap][co]
  __shared__ float sc[CIN > 1 ? CIN : 1], sh[CIN > 1 ? CIN : 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < CIN * 9 * COUT; i += 256) {
    const int co = i % COUT, rest = i / COUT, tap = rest % 9, ci = rest / 9;
    ws[i] = w[(co * CIN + ci) * 9 + tap];
  }
  if (BN_IN && tid < CIN) {
    const float mean = in_stats[tid], rstd = in_stats[CIN + tid];
    sc[tid] = rstd * in_g[tid];
    sh[tid] = in_b[tid] - mean * rstd * in_g[tid];
  }
  __syncthreads();
  const long total = (long)B * OH * OH;
  const long idx = (long)blockIdx.x * 256 + tid;
  if (idx >= total) return;
  const int ox = (int)(idx % OH), oy = (int)((idx / OH) % OH), b = (int)(idx / ((long)OH * OH));
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  const float* ib = in + (long)b * CIN * IH * IH;
  for (int ci = 0; ci < CIN; ++ci) {
    float v[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy * STRIDE - 1 + ky, ix = ox * STRIDE - 1 + kx;
        float t = 0.f;
        if (iy >= 0 && iy < IH && ix >= 0 && ix < IH) {
          t = ib[((long)ci * IH + iy) * IH + ix];
          if (BN_IN) {
            t = fmaf(t, sc[ci], sh[ci]);
            t = t >= 0.f ? t : t * LRELU;
          }
        }
        v[ky * 3 + kx] = t;
      }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* wr = &ws[(ci * 9 + tap) * COUT];
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] = fmaf(wr[co], v[tap], acc[co]);
    }
  }
  float* ob = out + (long)b * COUT * OH * OH + (long)oy * OH + ox;
#pragma unroll
  for (int co = 0; co < COUT; ++co) ob[(long)co * OH * OH] = acc[co];
}

// one workgroup per channel: mean / biased var in f64 over (B, H, W); writes mean, rstd; updates running buffers
__global__ __launch_bounds__(1024) void bn_stats_kernel(const float* __restrict__ y, int C, int B, int HW, float* __restrict__ stats,
                                                        float* __restrict__ rmean, float* __restrict__ rvar, int update) {
  __shared__ double red[16];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* p = y + ((long)b * C + c) * HW;
    for (int i = tid; i < HW; i += 1024) {
      const double v = (double)p[i];
      s += v;
      q += v * v;
    }
  }
  s = block_sum(s, red);
  q = block_sum(q, red);
  if (tid == 0) {
    const double n = (double)B * HW;
    const double mean = s / n;
    double var = q / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
    if (update) {
      const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
      rmean[c] = (1.f - BN_MOM) * rmean[c] + BN_MOM * (float)mean;
      rvar[c] = (1.f - BN_MOM) * rvar[c] + BN_MOM * (float)unb;
    }
  }
}

// one workgroup per image: BN3 + LeakyReLU, flatten (c, y, x), Linear, sigmoid
__global__ __launch_bounds__(256) void disc_head_kernel(const float* __restrict__ y3, const float* __restrict__ stats,
                                                        const float* __restrict__ g, const float* __restrict__ bta,
                                                        const float* __restrict__ lw, const float* __restrict__ lb,
                                                        float* __restrict__ prob, int HW3) {
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x, n = 8 * HW3;
  float acc = 0.f;
  for (int i = tid; i < n; i += 256) {
    const int c = i / HW3;
    acc = fmaf(lw[i], bn_lrelu(y3[(long)b * n + i], stats[c], stats[8 + c], g[c], bta[c]), acc);
  }
  acc = block_sum(acc, red);
  if (tid == 0) prob[b] = sigmoid_acc(acc + lb[0]);
}

}  // namespace ucod

using namespace ucod;

extern "C" size_t ucod_disc_saved_bytes(int B, int fs) {
  const DiscDims d = disc_dims(B, fs);
  return (stats_off(d) + 2 * (32 + 16 + 8)) * sizeof(float);
}

extern "C" int ucod_disc_fwd(const float* mask, const ucod_disc_params* p, float* prob, void* saved, int B, int fs,
                             int update_running, void* stream) {
  if (!mask || !p || !prob || !saved || B <= 0 || fs < 4) return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const DiscDims d = disc_dims(B, fs);
  float* y1 = (float*)saved;
  float* y2 = y1 + d.n1;
  float* y3 = y2 + d.n2;
  float* st1 = y1 + stats_off(d);
  float* st2 = st1 + 64;
  float* st3 = st2 + 32;
  hipLaunchKernelGGL((conv3x3_kernel<1, 32, 1, false>), dim3(cdiv((long)B * d.s1 * d.s1, 256)), dim3(256), 0, s, mask, p->w1, nullptr, nullptr, nullptr, y1, B, fs, d.s1);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(32), dim3(1024), 0, s, y1, 32, B, d.s1 * d.s1, st1, p->rm1, p->rv1, update_running);
  hipLaunchKernelGGL((conv3x3_kernel<32, 16, 2, true>), dim3(cdiv((long)B * d.s2 * d.s2, 256)), dim3(256), 0, s, y1, p->w2, st1, p->g1, p->b1, y2, B, d.s1, d.s2);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(16), dim3(1024), 0, s, y2, 16, B, d.s2 * d.s2, st2, p->rm2, p->rv2, update_running);
  hipLaunchKernelGGL((conv3x3_kernel<16, 8, 2, true>), dim3(cdiv((long)B * d.s3 * d.s3, 256)), dim3(256), 0, s, y2, p->w3, st2, p->g2, p->b2, y3, B, d.s2, d.s3);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(8), dim3(1024), 0, s, y3, 8, B, d.s3 * d.s3, st3, p->rm3, p->rv3, update_running);
  hipLaunchKernelGGL(disc_head_kernel, dim3(B), dim3(256), 0, s, y3, st3, p->g3, p->b3, p->lin_w, p->lin_b, prob, d.s3 * d.s3);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
