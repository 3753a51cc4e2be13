// Small streaming kernels of the decoder / APM path: bilinear resize (A1), APM fusion + both BCE losses + their
// gradients in one pass (A5+A6), thresholding, fused AdamW + EMA over a flat parameter arena (A7).
// These move < 2 MB per step (except the feature resize) and are launch-latency bound: each is one launch.
#include <cstdlib>
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

// The four-tap blend with its roundings pinned: per axis the product with the SECOND weight is rounded, the one with the first is
// fused -- w0*a + w1*b = fma(a, w0, rn(b*w1)).  That is the contraction ATen's CPU kernel was built with: on the step's geometries
// (37 -> 68 and 16 -> 68) this reproduces F.interpolate(mode="bilinear") bit for bit (tools/resize_rounding.py tries the nine
// candidate patterns), which is what keeps thresholds taken downstream (binarize at 0.5, loop_UCOD_DPL.py:241,261) on the
// reference's side of a tie.  Left to -ffp-contract, hipcc picked different patterns in the unrolled and the remainder loop.
__device__ __forceinline__ float lerp2(float v00, float v01, float v10, float v11, float lx, float ly) {
  const float wx = 1.f - lx, wy = 1.f - ly;
  const float top = fmaf(v00, wx, v01 * lx);
  const float bot = fmaf(v10, wx, v11 * lx);
  return fmaf(top, wy, bot * ly);
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
    out[idx] = lerp2(p[y0 * iw + x0], p[y0 * iw + x1], p[y1 * iw + x0], p[y1 * iw + x1], lx, ly);
  }
}

// Upsampling form used by the step (256-channel d: 37x37 -> 68x68, 196 MB): PB source planes are staged in LDS by 16-byte
// loads, a thread owns one group of four output columns (its x taps are computed once) and walks (plane, row) pairs, one
// 16-byte store per output group.  The element-per-thread kernel above spends its time in 64-bit index divisions and 4-byte
// stores (store-issue bound at ~2 TB/s); same arithmetic here, so the two agree bit for bit.
template <int PB>
__global__ __launch_bounds__(256) void bilinear_up4_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int ih, int iw,
                                                           int oh, int ow, float sh, float sw) {
  extern __shared__ __attribute__((aligned(16))) float sp[];           // PB x ih x iw
  const int tid = threadIdx.x;
  const int plane0 = blockIdx.x * PB;
  const int np = (planes - plane0) < PB ? (planes - plane0) : PB;
  const int src_elems = np * ih * iw;
  const float* src = in + (size_t)plane0 * ih * iw;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    for (int i = tid * 4; i + 3 < src_elems; i += 1024) *reinterpret_cast<f32x4*>(sp + i) = *reinterpret_cast<const f32x4*>(src + i);
    for (int i = (src_elems & ~3) + tid; i < src_elems; i += 256) sp[i] = src[i];
  } else {
    for (int i = tid; i < src_elems; i += 256) sp[i] = src[i];
  }
  const int G = ow >> 2, S = 256 / G;                                  // column groups per row, row slots per pass
  const int xg = tid % G, slot = tid / G;
  int x0[4], x1[4];
  float lx[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) src_index(4 * xg + e, sw, iw, x0[e], x1[e], lx[e]);
  __syncthreads();
  if (slot >= S) return;
  for (int r = slot; r < np * oh; r += S) {
    const int pl = r / oh, y = r - pl * oh;
    int y0, y1;
    float ly;
    src_index(y, sh, ih, y0, y1, ly);
    const float* p = sp + pl * ih * iw;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = lerp2(p[y0 * iw + x0[e]], p[y0 * iw + x1[e]], p[y1 * iw + x0[e]], p[y1 * iw + x1[e]], lx[e], ly);
    }
    *reinterpret_cast<f32x4*>(out + ((size_t)(plane0 + pl) * oh + y) * ow + 4 * xg) = o;
  }
}

// Adjoint (transpose) of bilinear_kernel: gin[plane][iy][ix] = sum over the output pixels whose taps touch (iy,ix) of
// weight * gout.  Gather form (no atomics): the candidate output range of a source index is scanned with the SAME
// src_index() as the forward, so forward and adjoint agree tap for tap.  Used to pull the decoder's gradient back from the
// 68x68 grid to the backbone's native 37x37 grid, where the 1x1-conv weight gradient is 3.4x cheaper (the conv and the
// resize commute: both are linear, one acts on channels, the other on pixels).
// taps of source index i along one axis: the (at most MAXT) output indices whose bilinear footprint touches i, with
// their weights.  Scans the candidate range with the SAME src_index() as the forward.
// MAXT (template parameter of the two kernels): 8 covers up-sampling up to 3.5x (the step's 37 -> 68: 5 taps) with the loops unrolled to eight
// predicated taps; 16 takes the rare coarser grids (a 224-pixel image gives 16 x 16 keys: 16 -> 68 is 4.25x, 10 taps) up to 7.5x.
template <int MAXT>
__device__ __forceinline__ int adjoint_taps(int i, float scale, int in_size, int out_size, int (&idx)[MAXT], float (&w)[MAXT]) {
  // outputs o whose source coordinate (o + 0.5) * scale - 0.5 lies in (i - 1, i + 1), one more on each side for the rounding of the bound
  // (round 5: the bounds (i -+ 1) / scale -+ 1 of the first version lose taps once 0.5 / scale - 0.5 exceeds that margin, i.e. beyond ~3.5x)
  int lo = (int)floorf(((float)i - 0.5f) / scale - 0.5f) - 1, hi = (int)ceilf(((float)i + 1.5f) / scale - 0.5f) + 1;
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
template <int MAXT>
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

// Separable LDS form of the adjoint (what the step uses: gd 4096 planes 68x68 -> 37x37).  The element kernel above already sums
// "x taps inside, y taps outside"; here the inner sums R[oy][ix] are formed ONCE per output row instead of once per (iy, tap):
// a workgroup walks `ppw` planes; per plane it stages gout in LDS (16-byte loads), pass 1 reduces along x with the thread's own
// column taps held in registers, pass 2 reduces along y with the row taps read from an LDS table.  Same products, same order:
// bit-identical to the element kernel, without its ~20 global gathers per output (it was bound by the gather rate).
template <int MAXT>
__global__ __launch_bounds__(256) void bilinear_adjoint_sep_kernel(const float* __restrict__ gout, float* __restrict__ gin, int planes, int ih,
                                                                   int iw, int oh, int ow, float sh, float sw, int ppw) {
  extern __shared__ __attribute__((aligned(16))) float sm[];           // G [oh*ow (padded to 4)] | R [oh*iw] | y taps
  const int g_elems = (oh * ow + 3) & ~3;
  float* G = sm;
  float* R = G + g_elems;
  int* yt_i = reinterpret_cast<int*>(R + oh * iw);                     // [ih][MAXT]
  float* yt_w = reinterpret_cast<float*>(yt_i + ih * MAXT);             // [ih][MAXT]
  int* yt_n = reinterpret_cast<int*>(yt_w + ih * MAXT);                 // [ih]
  const int tid = threadIdx.x;
  const int S = 256 / iw, ix = tid % iw, slot = tid / iw;
  const bool active = slot < S;
  int xi[MAXT];
  float xw[MAXT];
  const int nx = adjoint_taps(ix, sw, iw, ow, xi, xw);
  for (int a = tid; a < ih; a += 256) {
    int idx[MAXT];
    float w[MAXT];
    const int n = adjoint_taps(a, sh, ih, oh, idx, w);
    yt_n[a] = n;
#pragma unroll
    for (int e = 0; e < MAXT; ++e) { yt_i[a * MAXT + e] = e < n ? idx[e] : 0; yt_w[a * MAXT + e] = e < n ? w[e] : 0.f; }
  }
  const int n_in = oh * ow, hw = ih * iw;
  const int pl_end = (blockIdx.x + 1) * ppw < planes ? (blockIdx.x + 1) * ppw : planes;
  // Planes of up to 5120 elements whose rows are 16-byte aligned (the step's 68 x 68): a plane travels through five float4 registers per
  // thread, requested for plane pl + 1 BEFORE plane pl is reduced and written to LDS once pass 1 has released G -- the load latency of a
  // plane is covered by the previous plane's arithmetic (round 3: the copy loop waited for each pair of loads, five round trips per plane).
  const bool piped = n_in <= 5120 && (n_in & 3) == 0 && (reinterpret_cast<uintptr_t>(gout) & 15) == 0;
  f32x4 nxt[5];
  auto request = [&](int pl) {
    const float* src = gout + (size_t)pl * n_in;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int i = tid * 4 + j * 1024;
      nxt[j] = *reinterpret_cast<const f32x4*>(src + (i < n_in ? i : 0));
    }
  };
  auto deposit = [&]() {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int i = tid * 4 + j * 1024;
      if (i < n_in) *reinterpret_cast<f32x4*>(G + i) = nxt[j];
    }
  };
  if (piped && (int)(blockIdx.x * ppw) < pl_end) request(blockIdx.x * ppw);
  for (int pl = blockIdx.x * ppw; pl < pl_end; ++pl) {
    const float* src = gout + (size_t)pl * n_in;
    if (piped) {
      deposit();                                                       // (G is free: the barrier after pass 1 of the previous plane)
      if (pl + 1 < pl_end) request(pl + 1);
      __builtin_amdgcn_sched_barrier(0);
    } else if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
      for (int i = tid * 4; i + 3 < n_in; i += 1024) *reinterpret_cast<f32x4*>(G + i) = *reinterpret_cast<const f32x4*>(src + i);
      for (int i = (n_in & ~3) + tid; i < n_in; i += 256) G[i] = src[i];
    } else {
      for (int i = tid; i < n_in; i += 256) G[i] = src[i];
    }
    __syncthreads();                                                   // G (and, first time, the y table) ready; R free: everyone is past pass 2
    if (active) {
      for (int oy = slot; oy < oh; oy += S) {
        const float* grow = G + oy * ow;
        float row = 0.f;
#pragma unroll
        for (int b2 = 0; b2 < MAXT; ++b2)
          if (b2 < nx) row = fmaf(xw[b2], grow[xi[b2]], row);
        R[oy * iw + ix] = row;
      }
    }
    __syncthreads();                                                   // R ready; G free for the next plane
    if (active) {
      float* dst = gin + (size_t)pl * hw;
      for (int iy = slot; iy < ih; iy += S) {
        const int ny = yt_n[iy];
        float acc = 0.f;
#pragma unroll
        for (int a2 = 0; a2 < MAXT; ++a2)
          if (a2 < ny) acc = fmaf(yt_w[iy * MAXT + a2], R[yt_i[iy * MAXT + a2] * iw + ix], acc);
        dst[iy * iw + ix] = acc;
      }
    }
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

__global__ void step_loss_kernel(const float* __restrict__ losses, const float* __restrict__ extra, int finetune, float* __restrict__ out) {
  float l = losses[0] + losses[1] + extra[0];
  if (!finetune) l -= losses[2];
  out[0] = l;
}

static inline int nblocks(size_t n, int cap) { return (int)((n + 255) / 256 < (size_t)cap ? (n + 255) / 256 : (size_t)cap); }

}  // namespace ucod

using namespace ucod;

extern "C" int ucod_bilinear_resize(const float* in, float* out, int planes, int ih, int iw, int oh, int ow, void* stream) {
  if (!in || !out || planes <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0) return UCOD_EINVAL;
  const size_t total = (size_t)planes * oh * ow;
  const float sh = (float)ih / (float)oh, sw = (float)iw / (float)ow;   // area_pixel_compute_scale<float>
  UCOD_PROF(PROF_BILINEAR, stream);
  constexpr int PB = 4;
  const size_t lds = (size_t)PB * ih * iw * sizeof(float);
  if (!ucod::lab_env("UCOD_RESIZE_ELEMENTWISE") && (ow & 3) == 0 && ow >= 4 && ow <= 1024 && lds <= 48 * 1024 && planes >= 64 && oh * ow >= ih * iw) {
    hipLaunchKernelGGL(bilinear_up4_kernel<PB>, dim3(cdiv(planes, PB)), dim3(256), lds, (hipStream_t)stream, in, out, planes, ih, iw, oh, ow, sh, sw);
  } else {
    hipLaunchKernelGGL(bilinear_kernel, dim3(nblocks(total, 16384)), dim3(256), 0, (hipStream_t)stream, in, out, (long)planes, ih, iw, oh, ow, sh, sw);
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_bilinear_resize_adjoint(const float* gout, float* gin, int planes, int ih, int iw, int oh, int ow, void* stream) {
  if (!gout || !gin || planes <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0) return UCOD_EINVAL;
  const size_t total = (size_t)planes * ih * iw;
  const float sh = (float)ih / (float)oh, sw = (float)iw / (float)ow;
  UCOD_PROF(PROF_BILINEAR, stream);
  (void)total;
  const float taps = ceilf(2.f / fminf(sh, sw)) + 1.f;
  if (taps > 16.f) return UCOD_EINVAL;                                       // more than 16 taps per axis (up-sampling beyond 7.5x)
  const int MAXT = taps > 8.f ? 16 : 8;
  const size_t lds = ((((size_t)oh * ow + 3) & ~(size_t)3) + (size_t)oh * iw + (size_t)ih * (2 * MAXT + 1)) * sizeof(float);
  if (!ucod::lab_env("UCOD_RESIZE_ELEMENTWISE") && lds <= 60 * 1024 && planes >= 64 && iw <= 128) {
    // planes per workgroup: amortises the tap set-up, but a workgroup's planes are strictly sequential (load, barrier, pass 1, barrier,
    // pass 2) and the latencies are hidden by the other 4 workgroups on the CU -- 4096 planes: 4 per workgroup 40 us, 8: 72 us, 1: 50 us
    const int ppw = planes >= 2048 ? 4 : (planes >= 512 ? 2 : 1);
    if (MAXT == 8) hipLaunchKernelGGL(bilinear_adjoint_sep_kernel<8>, dim3(cdiv(planes, ppw)), dim3(256), lds, (hipStream_t)stream, gout, gin, planes, ih, iw, oh, ow, sh, sw, ppw);
    else hipLaunchKernelGGL(bilinear_adjoint_sep_kernel<16>, dim3(cdiv(planes, ppw)), dim3(256), lds, (hipStream_t)stream, gout, gin, planes, ih, iw, oh, ow, sh, sw, ppw);
  } else {
    const int py = planes < 1024 ? planes : 1024;
    if (MAXT == 8) hipLaunchKernelGGL(bilinear_adjoint_kernel<8>, dim3(cdiv((long)ih * iw, 256), py), dim3(256), 0, (hipStream_t)stream, gout, gin, (long)planes, ih, iw, oh, ow, sh, sw);
    else hipLaunchKernelGGL(bilinear_adjoint_kernel<16>, dim3(cdiv((long)ih * iw, 256), py), dim3(256), 0, (hipStream_t)stream, gout, gin, (long)planes, ih, iw, oh, ow, sh, sw);
  }
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
  if (!ucod::accumulators_prezeroed()) {
    hipError_t e = hipMemsetAsync(losses, 0, 4 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(apm_bce_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, s, pl, teacher, fg, bg, p_s, p_p, epoch_frac, gscale, w, merged, gfg, gbg, losses, B, HW);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_step_loss(const float* losses, const float* extra, int finetune, float* out, void* stream) {
  if (!losses || !extra || !out) return UCOD_EINVAL;
  hipLaunchKernelGGL(step_loss_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, losses, extra, finetune, out);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

// Up to four f32 segment copies in ONE launch (loop_UCOD_DPL.py's shared student | teacher projection: two weight blocks, two bias blocks per step;
// four hipMemcpyAsync = four rocclr copy kernels before round 4).  Segments with n == 0 are skipped.
namespace ucod {
struct CopySegs { float* dst[4]; const float* src[4]; size_t n[4]; };
__global__ __launch_bounds__(256) void copy_segments_kernel(const CopySegs s, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    size_t k = i;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (k < s.n[q]) { s.dst[q][k] = s.src[q][k]; break; }
      k -= s.n[q];
    }
  }
}
}  // namespace ucod

namespace ucod {
static thread_local bool t_prezeroed = false;
bool accumulators_prezeroed() { return t_prezeroed; }
struct ZeroSegs { unsigned* dst[8]; size_t n[8]; };                  // n in 4-byte words
__global__ __launch_bounds__(256) void zero_segments_kernel(const ZeroSegs s, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    size_t k = i;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (k < s.n[q]) { s.dst[q][k] = 0u; break; }
      k -= s.n[q];
    }
  }
}
}  // namespace ucod

extern "C" int ucod_accumulators_prezeroed(int on) {
  ucod::t_prezeroed = on != 0;
  return UCOD_OK;
}

extern "C" int ucod_zero_segments(void* const* dst, const size_t* bytes, int count, void* stream) {
  if (!dst || !bytes || count < 1 || count > 8) return UCOD_EINVAL;
  ucod::ZeroSegs s{};
  size_t total = 0;
  for (int q = 0; q < count; ++q) {
    if (bytes[q] && (!dst[q] || (bytes[q] & 3) || ((size_t)dst[q] & 3))) return UCOD_EINVAL;
    s.dst[q] = (unsigned*)dst[q];
    s.n[q] = bytes[q] / 4;
    total += s.n[q];
  }
  if (total == 0) return UCOD_OK;
  hipLaunchKernelGGL(ucod::zero_segments_kernel, dim3(ucod::nblocks(total, 1024)), dim3(256), 0, (hipStream_t)stream, s, total);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_copy_segments(float* const* dst, const float* const* src, const size_t* n, int count, void* stream) {
  if (!dst || !src || !n || count < 1 || count > 4) return UCOD_EINVAL;
  ucod::CopySegs s{};
  size_t total = 0;
  for (int q = 0; q < count; ++q) {
    if (n[q] && (!dst[q] || !src[q])) return UCOD_EINVAL;
    s.dst[q] = dst[q]; s.src[q] = src[q]; s.n[q] = n[q];
    total += n[q];
  }
  if (total == 0) return UCOD_OK;
  hipLaunchKernelGGL(ucod::copy_segments_kernel, dim3(ucod::nblocks(total, 1024)), dim3(256), 0, (hipStream_t)stream, s, total);
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
