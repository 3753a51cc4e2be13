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
// saved layout (floats): y1 | y2 | y3 | stats[2*(32+16+8)] (mean, rstd per channel, layer after layer) | f64 sums[2*(32+16+8)]
static inline size_t stats_off(const DiscDims& d) { return d.n1 + d.n2 + d.n3; }
static inline size_t acc_off_bytes(const DiscDims& d) { return ((stats_off(d) + 112) * sizeof(float) + 7) / 8 * 8; }

__device__ __forceinline__ float bn_lrelu(float y, float mean, float rstd, float g, float b) {
  const float h = (y - mean) * rstd * g + b;
  return h >= 0.f ? h : h * LRELU;
}

// in: [B,CIN,IH,IW] (pre-BN output of the previous block, or the raw mask when !BN_IN); out: [B,COUT,OH,OW] pre-BN
template <int CIN, int COUT, int STRIDE, bool BN_IN, int CPT = COUT>
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                      const float* __restrict__ in_stats, const float* __restrict__ in_g,
                                                      const float* __restrict__ in_b, float* __restrict__ out,
                                                      double* __restrict__ gacc /* [2*COUT]: sum, sum of squares */, int B, int IH,
                                                      int OH) {
  // Round 3.  Weights: with CIN >= 16 they are read straight from `w` with wave-uniform indices (scalar loads, the FMA takes the SGPR
  // operand); they used to be transposed into LDS and re-read per FMA -- 4 608 LDS reads per thread in the 32 -> 16 layer, 63.7 us for
  // 341 MFLOP.  The single-input-channel layer keeps its 288 weights in LDS (they do not fit the scalar registers at once).  Taps: the nine
  // clamped offsets and validity flags are computed once, not per input channel.  The accumulation order per output channel -- (ci, tap)
  // lexicographic -- is unchanged, so results are bitwise the same as before.
  // CPT output channels per thread (blockIdx.y selects the group): the 32 -> 16 layer has 145 workgroups' worth of pixels at batch 32, not
  // one wave per SIMD, and every input-channel step is a load latency followed by its FMAs; four channel groups put four times the waves
  // on the chip.  (The statistics partial of a workgroup is then summed over 256 / CPT instead of 256 / COUT threads per channel.)
  static_assert(COUT % CPT == 0 && 256 % CPT == 0, "channel groups");
  const int c0 = blockIdx.y * CPT;
  constexpr bool SW = CIN >= 16 || CIN * CPT * 9 <= 96;   // (scalar weights: a step's CPT x 9, or all of them, fit the scalar registers)
  __shared__ __attribute__((aligned(16))) float ws[SW ? 1 : CIN * 9 * COUT];  // [ci][tap][co]
  __shared__ float sc[CIN > 1 ? CIN : 1], sh[CIN > 1 ? CIN : 1];
  constexpr int PARTS = 256 / CPT;                      // threads cooperating on one channel in the statistics pass
  __shared__ float red[256 * (CPT + 1)];
  __shared__ float psum[PARTS][CPT], psq[PARTS][CPT];
  const int tid = threadIdx.x;
  if constexpr (!SW) {
    for (int i = tid; i < CIN * 9 * COUT; i += 256) {
      const int co = i % COUT, rest = i / COUT, tap = rest % 9, ci = rest / 9;
      ws[i] = w[(co * CIN + ci) * 9 + tap];
    }
  }
  if (BN_IN && tid < CIN) {
    const float mean = in_stats[tid], rstd = in_stats[CIN + tid];
    sc[tid] = rstd * in_g[tid];
    sh[tid] = in_b[tid] - mean * rstd * in_g[tid];
  }
  __syncthreads();
  const long total = (long)B * OH * OH;
  const long idx0 = (long)blockIdx.x * 256 + tid;
  const bool valid = idx0 < total;
  const long idx = valid ? idx0 : total - 1;
  const int ox = (int)(idx % OH), oy = (int)((idx / OH) % OH), b = (int)(idx / ((long)OH * OH));
  float acc[CPT];
#pragma unroll
  for (int co = 0; co < CPT; ++co) acc[co] = 0.f;
  int off[9];
  bool ok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * STRIDE - 1 + ky, ix = ox * STRIDE - 1 + kx;
      ok[ky * 3 + kx] = iy >= 0 && iy < IH && ix >= 0 && ix < IH;
      const int cy = iy < 0 ? 0 : (iy >= IH ? IH - 1 : iy), cx = ix < 0 ? 0 : (ix >= IH ? IH - 1 : ix);
      off[ky * 3 + kx] = cy * IH + cx;
    }
  const float* ib = in + (long)b * CIN * IH * IH;
  for (int ci = 0; ci < CIN; ++ci) {
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float t = ib[off[k]];                                       // (clamped: always in range)
      if (BN_IN) {
        t = fmaf(t, sc[ci], sh[ci]);
        t = t >= 0.f ? t : t * LRELU;
      }
      v[k] = ok[k] ? t : 0.f;
    }
    ib += IH * IH;
    if constexpr (SW) {
#pragma unroll
      for (int co = 0; co < CPT; ++co) {
        const float* wr = w + ((c0 + co) * CIN + ci) * 9;         // (uniform address: scalar loads)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc[co] = fmaf(wr[tap], v[tap], acc[co]);
      }
    } else {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float* wr = &ws[(ci * 9 + tap) * COUT + c0];
#pragma unroll
        for (int co = 0; co < CPT; ++co) acc[co] = fmaf(wr[co], v[tap], acc[co]);
      }
    }
  }
  float* ob = out + ((long)b * COUT + c0) * OH * OH + (long)oy * OH + ox;
  // pre-BN output + this workgroup's share of the batch statistics: transpose through LDS (thread-major -> channel-major),
  // PARTS threads per channel sum 256/PARTS pixels each, then one f64 atomic per channel per workgroup.
#pragma unroll
  for (int co = 0; co < CPT; ++co) {
    const float v = valid ? acc[co] : 0.f;
    if (valid) ob[(long)co * OH * OH] = v;
    red[tid * (CPT + 1) + co] = v;
  }
  __syncthreads();
  {
    const int ch = tid % CPT, part = tid / CPT;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int r = part; r < 256; r += PARTS) {
      const float v = red[r * (CPT + 1) + ch];
      s1 += v;
      s2 = fmaf(v, v, s2);
    }
    psum[part][ch] = s1;
    psq[part][ch] = s2;
  }
  __syncthreads();
  if (tid < CPT) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < PARTS; ++q) { s1 += psum[q][tid]; s2 += psq[q][tid]; }
    atomicAdd(&gacc[c0 + tid], (double)s1);
    atomicAdd(&gacc[COUT + c0 + tid], (double)s2);
  }
}

// mean / biased variance from the f64 sums the conv kernel accumulated; writes (mean, rstd), updates the running buffers
// (the sums are consumed here and nowhere else: they are left ZERO for the next call, so that a caller that keeps its `saved` buffer -- allocated
// zeroed -- under ucod_accumulators_prezeroed never needs a memset in front of ucod_disc_fwd)
__global__ void bn_finalize_kernel(double* __restrict__ gacc, int C, double n, float* __restrict__ stats,
                                   float* __restrict__ rmean, float* __restrict__ rvar, int update) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const double mean = gacc[c] / n;
  double var = gacc[C + c] / n - mean * mean;
  gacc[c] = 0.0;
  gacc[C + c] = 0.0;
  var = var > 0.0 ? var : 0.0;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
  if (update) {
    const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
    rmean[c] = (1.f - BN_MOM) * rmean[c] + BN_MOM * (float)mean;
    rvar[c] = (1.f - BN_MOM) * rvar[c] + BN_MOM * (float)unb;
  }
}

// one workgroup per image: BN3 + LeakyReLU, flatten (c, y, x), Linear, sigmoid
__global__ __launch_bounds__(256) void disc_head_kernel(const float* __restrict__ y3, const float* __restrict__ stats,
                                                        const float* __restrict__ g, const float* __restrict__ bta,
                                                        const float* __restrict__ lw, const float* __restrict__ lb,
                                                        float* __restrict__ prob, int HW3, long long* __restrict__ nbt) {
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x, n = 8 * HW3;
  float acc = 0.f;
  for (int i = tid; i < n; i += 256) {
    const int c = i / HW3;
    acc = fmaf(lw[i], bn_lrelu(y3[(long)b * n + i], stats[c], stats[8 + c], g[c], bta[c]), acc);
  }
  acc = block_sum(acc, red);
  if (tid == 0) prob[b] = sigmoid_acc(acc + lb[0]);
  // nn.BatchNorm2d.num_batches_tracked of the three blocks (one training-mode call = +1 each): done here, in the pass's last launch, so
  // that the host issues no elementwise kernel for it
  if (nbt && b == 0 && tid < 3) nbt[tid] += 1;
}

}  // namespace ucod

using namespace ucod;

extern "C" size_t ucod_disc_saved_bytes(int B, int fs) {
  const DiscDims d = disc_dims(B, fs);
  return acc_off_bytes(d) + 112 * sizeof(double);
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
  UCOD_PROF(PROF_DISC_FWD, s);
  double* ac1 = (double*)((char*)saved + acc_off_bytes(d));
  double* ac2 = ac1 + 64;
  double* ac3 = ac2 + 32;
  if (!accumulators_prezeroed()) {
    hipError_t e = hipMemsetAsync(ac1, 0, 112 * sizeof(double), s);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((conv3x3_kernel<1, 32, 1, false, 8>), dim3(cdiv((long)B * d.s1 * d.s1, 256), 4), dim3(256), 0, s, mask, p->w1, nullptr, nullptr, nullptr, y1, ac1, B, fs, d.s1);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, s, ac1, 32, (double)B * d.s1 * d.s1, st1, p->rm1, p->rv1, update_running);
  hipLaunchKernelGGL((conv3x3_kernel<32, 16, 2, true, 4>), dim3(cdiv((long)B * d.s2 * d.s2, 256), 4), dim3(256), 0, s, y1, p->w2, st1, p->g1, p->b1, y2, ac2, B, d.s1, d.s2);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, s, ac2, 16, (double)B * d.s2 * d.s2, st2, p->rm2, p->rv2, update_running);
  hipLaunchKernelGGL((conv3x3_kernel<16, 8, 2, true, 2>), dim3(cdiv((long)B * d.s3 * d.s3, 256), 4), dim3(256), 0, s, y2, p->w3, st2, p->g2, p->b2, y3, ac3, B, d.s2, d.s3);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, s, ac3, 8, (double)B * d.s3 * d.s3, st3, p->rm3, p->rv3, update_running);
  hipLaunchKernelGGL(disc_head_kernel, dim3(B), dim3(256), 0, s, y3, st3, p->g3, p->b3, p->lin_w, p->lin_b, prob, d.s3 * d.s3, update_running ? p->nbt : nullptr);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

// =====================================================================================================
// Backward (discriminator phase, engine/runner/loop_UCOD_DPL.py:230-255): gradient of sum_b gprob[b]*prob[b]
// w.r.t. every parameter.  Per block, from the top:  head -> [BN+LeakyReLU backward: per-channel sums of gh and
// gh*xhat (f64), then gy = gamma*rstd*(gh - mean(gh) - xhat*mean(gh*xhat))] -> conv weight gradient
// (recomputing the activated input on the fly) -> conv input gradient (-> gh of the block below).
// =====================================================================================================
namespace ucod {

// head: gz = gprob*p*(1-p); g_lin_w += gz*a3; g_lin_b += gz; gh3 = gz*lin_w*lrelu'(h3)
__global__ __launch_bounds__(256) void disc_head_bwd_kernel(const float* __restrict__ y3, const float* __restrict__ stats,
                                                            const float* __restrict__ g, const float* __restrict__ bta,
                                                            const float* __restrict__ lw, const float* __restrict__ lb,
                                                            const float* __restrict__ gprob, float* __restrict__ gh3,
                                                            float* __restrict__ g_lw, float* __restrict__ g_lb, int HW3) {
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x, n = 8 * HW3;
  float acc = 0.f;
  for (int i = tid; i < n; i += 256) {
    const int c = i / HW3;
    acc = fmaf(lw[i], bn_lrelu(y3[(long)b * n + i], stats[c], stats[8 + c], g[c], bta[c]), acc);
  }
  acc = block_sum(acc, red);
  const float p = sigmoid_acc(acc + lb[0]);
  const float gz = gprob[b] * p * (1.f - p);
  for (int i = tid; i < n; i += 256) {
    const int c = i / HW3;
    const float h = (y3[(long)b * n + i] - stats[c]) * stats[8 + c] * g[c] + bta[c];
    const float a = h >= 0.f ? h : h * LRELU;
    atomicAdd(&g_lw[i], gz * a);
    gh3[(long)b * n + i] = gz * lw[i] * (h >= 0.f ? 1.f : LRELU);
  }
  if (tid == 0) atomicAdd(g_lb, gz);
}

// one workgroup per channel: S1 = sum gh, S2 = sum gh*xhat (f64) -> g_beta, g_gamma (accumulated) and sums[c], sums[C+c]
__global__ __launch_bounds__(1024) void bn_bwd_stats_kernel(const float* __restrict__ y, const float* __restrict__ gh, int C, int B,
                                                            int HW, const float* __restrict__ stats, float* __restrict__ sums,
                                                            float* __restrict__ g_gamma, float* __restrict__ g_beta) {
  __shared__ double red[16];
  const int c = blockIdx.x, tid = threadIdx.x;
  const float mean = stats[c], rstd = stats[C + c];
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < B; ++b) {
    const long off = ((long)b * C + c) * HW;
    for (int i = tid; i < HW; i += 1024) {
      const double gv = (double)gh[off + i];
      s1 += gv;
      s2 += gv * (double)((y[off + i] - mean) * rstd);
    }
  }
  s1 = block_sum(s1, red);
  s2 = block_sum(s2, red);
  if (tid == 0) {
    const double n = (double)B * HW;
    sums[c] = (float)(s1 / n);
    sums[C + c] = (float)(s2 / n);
    g_beta[c] += (float)s1;
    g_gamma[c] += (float)s2;
  }
}

// gh -> gy in place
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ y, float* __restrict__ gh, int C, int HW, long total,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ sums) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)((i / HW) % C);
    const float rstd = stats[C + c];
    const float xh = (y[i] - stats[c]) * rstd;
    gh[i] = gamma[c] * rstd * (gh[i] - sums[c] - xh * sums[C + c]);
  }
}

// weight gradient: workgroup = (co, ci) x image chunk; partial[chunk][co][ci][tap] = sum over the chunk's (b,oy,ox) of
// gy[b,co,oy,ox] * act_in[b,ci,iy,ix], summed over chunks in a fixed order by conv3x3_wgrad_reduce_kernel (deterministic).
// (One workgroup per (co, ci) over ALL images was the whole discriminator phase: the 1->32 first layer has only 32 such pairs,
// 32 workgroups on a 256-CU chip each walking 148 k pixels: 0.98 ms per call.)
template <int CIN, int COUT, int STRIDE, bool BN_IN>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ in_stats,
                                                            const float* __restrict__ in_g, const float* __restrict__ in_b,
                                                            const float* __restrict__ gy, float* __restrict__ partial, int B, int IH, int OH) {
  __shared__ float red[16];
  const int co = blockIdx.x / CIN, ci = blockIdx.x % CIN, tid = threadIdx.x;
  const int b_lo = (int)((long)B * blockIdx.y / gridDim.y), b_hi = (int)((long)B * (blockIdx.y + 1) / gridDim.y);
  float sc = 1.f, sh = 0.f;
  if (BN_IN) {
    const float mean = in_stats[ci], rstd = in_stats[CIN + ci];
    sc = rstd * in_g[ci];
    sh = in_b[ci] - mean * sc;
  }
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  // 32-bit pixel index inside one image (the 64-bit div / mod per pixel and the branch around every tap load -- each with its own
  // vmcnt(0) -- were most of this kernel's time); the pixels of an image are dealt to the threads per image now, so the sums add in a
  // different (still fixed) order
  const int npix = OH * OH;
  for (int b = b_lo; b < b_hi; ++b) {
    const float* gyb = gy + ((long)b * COUT + co) * npix;
    const float* ib = in + ((long)b * CIN + ci) * IH * IH;
    for (int p = tid; p < npix; p += 256) {
      const int oy = p / OH, ox = p - oy * OH;
      const float gv = gyb[p];
      float t9[9];
      bool ok9[9];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = oy * STRIDE - 1 + ky, ix = ox * STRIDE - 1 + kx;
          ok9[ky * 3 + kx] = iy >= 0 && iy < IH && ix >= 0 && ix < IH;
          const int cy = iy < 0 ? 0 : (iy >= IH ? IH - 1 : iy), cx = ix < 0 ? 0 : (ix >= IH ? IH - 1 : ix);
          t9[ky * 3 + kx] = ib[cy * IH + cx];
        }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        float t = t9[k];
        if (BN_IN) {
          t = fmaf(t, sc, sh);
          t = t >= 0.f ? t : t * LRELU;
        }
        if (ok9[k]) acc[k] = fmaf(gv, t, acc[k]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float s = block_sum(acc[t], red);
    if (tid == 0) partial[((long)blockIdx.y * COUT * CIN + blockIdx.x) * 9 + t] = s;
  }
}

__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw, int n, int chunks) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += partial[(long)c * n + i];
  gw[i] += s;
}

// input gradient of a conv block, fused with the LeakyReLU' of the block BELOW: thread = input pixel (b,iy,ix),
// gh_in[b,ci,iy,ix] = lrelu'(h_in) * sum_{co,ky,kx} gy[b,co,oy,ox] * w[co][ci][ky][kx],  oy = (iy+1-ky)/STRIDE
template <int CIN, int COUT, int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                            const float* __restrict__ y_in, const float* __restrict__ in_stats,
                                                            const float* __restrict__ in_g, const float* __restrict__ in_b,
                                                            float* __restrict__ gh_in, int B, int IH, int OH) {
  __shared__ float ws[COUT * 9 * CIN];  // [co][tap][ci]
  const int tid = threadIdx.x;
  for (int i = tid; i < COUT * 9 * CIN; i += 256) {
    const int ci = i % CIN, rest = i / CIN, tap = rest % 9, co = rest / 9;
    ws[i] = w[(co * CIN + ci) * 9 + tap];
  }
  __syncthreads();
  const long total = (long)B * IH * IH;
  const long idx = (long)blockIdx.x * 256 + tid;
  if (idx >= total) return;
  const int ix = (int)(idx % IH), iy = (int)((idx / IH) % IH), b = (int)(idx / ((long)IH * IH));
  float acc[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
  // which of the nine taps reach an output pixel, and where, does not depend on the output channel: computed once; the loads are
  // unconditional (clamped) and the tap is skipped by a wave-level test -- a branch around each load cost a vmcnt(0) per tap
  int toff[9];
  bool tok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ty = iy + 1 - ky, tx = ix + 1 - kx;
      const int oy = ty / STRIDE, ox = tx / STRIDE;
      const bool ok = ty >= 0 && tx >= 0 && (ty % STRIDE) == 0 && (tx % STRIDE) == 0 && oy < OH && ox < OH;
      tok[ky * 3 + kx] = ok;
      toff[ky * 3 + kx] = ok ? oy * OH + ox : 0;
    }
  const float* gyb = gy + (long)b * COUT * OH * OH;
  for (int co = 0; co < COUT; ++co) {
    float gv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) gv[k] = gyb[co * OH * OH + toff[k]];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float g = tok[k] ? gv[k] : 0.f;
      const float* wr = &ws[(co * 9 + k) * CIN];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) acc[ci] = fmaf(g, wr[ci], acc[ci]);
    }
  }
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) {
    const long o = (((long)b * CIN + ci) * IH + iy) * IH + ix;
    const float h = (y_in[o] - in_stats[ci]) * in_stats[CIN + ci] * in_g[ci] + in_b[ci];
    gh_in[o] = acc[ci] * (h >= 0.f ? 1.f : LRELU);
  }
}

}  // namespace ucod

// weight-gradient partials: chunks x (co*ci) x 9, chunks = min(B, 2048 / pairs) -> at most 2048 x 9 floats per layer
constexpr int WGRAD_MAX_PARTIAL = 2048 * 9 + 512 * 9;

extern "C" size_t ucod_disc_bwd_workspace_bytes(int B, int fs) {
  const DiscDims d = disc_dims(B, fs);
  return (d.n1 + d.n2 + d.n3 + 2 * (32 + 16 + 8) + (size_t)WGRAD_MAX_PARTIAL) * sizeof(float);
}

extern "C" int ucod_disc_bwd(const float* mask, const ucod_disc_params* p, const void* saved, const float* gprob,
                             const ucod_disc_grads* g, int accumulate, void* ws, int B, int fs, void* stream) {
  if (!mask || !p || !saved || !gprob || !g || !ws || B <= 0 || fs < 4) return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const DiscDims d = disc_dims(B, fs);
  const float* y1 = (const float*)saved;
  const float* y2 = y1 + d.n1;
  const float* y3 = y2 + d.n2;
  const float* st1 = y1 + stats_off(d);
  const float* st2 = st1 + 64;
  const float* st3 = st2 + 32;
  float* gh1 = (float*)ws;
  float* gh2 = gh1 + d.n1;
  float* gh3 = gh2 + d.n2;
  float* su1 = gh3 + d.n3;
  float* su2 = su1 + 64;
  float* su3 = su2 + 32;
  float* wpart = su3 + 16;
  const int hw1 = d.s1 * d.s1, hw2 = d.s2 * d.s2, hw3 = d.s3 * d.s3;
  auto chunks_for = [&](int pairs) { const int c = 2048 / pairs; return c < 1 ? 1 : (c > B ? B : c); };
  UCOD_PROF(PROF_DISC_BWD, s);
  if (!accumulate) {
    hipError_t e = hipSuccess;
    auto z = [&](float* ptr, size_t n) { if (e == hipSuccess) e = hipMemsetAsync(ptr, 0, n * sizeof(float), s); };
    z(g->w1, 32 * 9); z(g->g1, 32); z(g->b1, 32); z(g->w2, 16 * 32 * 9); z(g->g2, 16); z(g->b2, 16);
    z(g->w3, 8 * 16 * 9); z(g->g3, 8); z(g->b3, 8); z(g->lin_w, (size_t)8 * hw3); z(g->lin_b, 1);
    if (e != hipSuccess) return (int)e;
  }
  auto blocks = [](size_t n) { return dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)); };
  hipLaunchKernelGGL(disc_head_bwd_kernel, dim3(B), dim3(256), 0, s, y3, st3, p->g3, p->b3, p->lin_w, p->lin_b, gprob, gh3, g->lin_w, g->lin_b, hw3);
  // block 3
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(8), dim3(1024), 0, s, y3, gh3, 8, B, hw3, st3, su3, g->g3, g->b3);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, blocks(d.n3), dim3(256), 0, s, y3, gh3, 8, hw3, (long)d.n3, st3, p->g3, su3);
  {
    const int ch = chunks_for(8 * 16);
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<16, 8, 2, true>), dim3(8 * 16, ch), dim3(256), 0, s, y2, st2, p->g2, p->b2, gh3, wpart, B, d.s2, d.s3);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(cdiv(8 * 16 * 9, 256)), dim3(256), 0, s, wpart, g->w3, 8 * 16 * 9, ch);
  }
  hipLaunchKernelGGL((conv3x3_dgrad_kernel<16, 8, 2>), dim3(cdiv((long)B * hw2, 256)), dim3(256), 0, s, gh3, p->w3, y2, st2, p->g2, p->b2, gh2, B, d.s2, d.s3);
  // block 2
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(16), dim3(1024), 0, s, y2, gh2, 16, B, hw2, st2, su2, g->g2, g->b2);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, blocks(d.n2), dim3(256), 0, s, y2, gh2, 16, hw2, (long)d.n2, st2, p->g2, su2);
  {
    const int ch = chunks_for(16 * 32);
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<32, 16, 2, true>), dim3(16 * 32, ch), dim3(256), 0, s, y1, st1, p->g1, p->b1, gh2, wpart, B, d.s1, d.s2);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(cdiv(16 * 32 * 9, 256)), dim3(256), 0, s, wpart, g->w2, 16 * 32 * 9, ch);
  }
  hipLaunchKernelGGL((conv3x3_dgrad_kernel<32, 16, 2>), dim3(cdiv((long)B * hw1, 256)), dim3(256), 0, s, gh2, p->w2, y1, st1, p->g1, p->b1, gh1, B, d.s1, d.s2);
  // block 1
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(32), dim3(1024), 0, s, y1, gh1, 32, B, hw1, st1, su1, g->g1, g->b1);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, blocks(d.n1), dim3(256), 0, s, y1, gh1, 32, hw1, (long)d.n1, st1, p->g1, su1);
  {
    const int ch = chunks_for(32);
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<1, 32, 1, false>), dim3(32, ch), dim3(256), 0, s, mask, nullptr, nullptr, nullptr, gh1, wpart, B, fs, d.s1);
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(cdiv(32 * 9, 256)), dim3(256), 0, s, wpart, g->w1, 32 * 9, ch);
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
