// CORAL SparseRefiner support kernels (models/UDLR.py, models/modules/{ASR,CSF,HRE,GE_pix_level}.py), inference path.
// The heavy parts (LayerNorms, q / kv / out / MLP projections, cross-attention) reuse the backbone's kernels; this file holds
// the refiner-specific data movement and small spatial ops.  All f32.
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

// ------------------------------------------------------------------ gather + NCHW -> token-major transpose
// out[(i*HW + p)*C + c] = src[(src_idx[i]*C + c)*HW + p]   (CSF._BCHW_to_BLC, CSF.py:29-31, fused with the window gather of
// EntropySelector.window_sets, ASR.py:13-20: h_inputs[mask] / repeat_interleave(input_features)).  64x64 LDS tiles.
__global__ __launch_bounds__(256) void gather_tokens_kernel(const float* __restrict__ src, const int* __restrict__ src_idx,
                                                            float* __restrict__ out, int C, int HW) {
  __shared__ float tile[64][65];
  const int i = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const float* s = src + (size_t)src_idx[i] * C * HW;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, p = p0 + tx;
    tile[r][tx] = (c < C && p < HW) ? s[(size_t)c * HW + p] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 64; r += 4) {
    const int p = p0 + r, c = c0 + tx;
    if (p < HW && c < C) out[((size_t)i * HW + p) * C + c] = tile[tx][r];
  }
}

// ------------------------------------------------------------------ entropy + adaptive average pooling (ASR.py:42-48)
// entropy = -p*log(max(p,1e-5)), p = preds or sigmoid(preds); scores[b][i][j] = mean over the adaptive_avg_pool2d bin.
__global__ __launch_bounds__(256) void entropy_scores_kernel(const float* __restrict__ preds, int use_sigmoid, float* __restrict__ entropy,
                                                             float* __restrict__ scores, int H, int W, int ws) {
  __shared__ float red[16];
  const int b = blockIdx.z, bi = blockIdx.y, bj = blockIdx.x;
  const int y0 = (bi * H) / ws, y1 = ((bi + 1) * H + ws - 1) / ws;     // floor / ceil bin edges
  const int x0 = (bj * W) / ws, x1 = ((bj + 1) * W + ws - 1) / ws;
  const int bw = x1 - x0, n = (y1 - y0) * bw;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int y = y0 + i / bw, x = x0 + i % bw;
    const float v = preds[((size_t)b * H + y) * W + x];
    const float p = use_sigmoid ? sigmoid_acc(v) : v;
    const float e = -p * logf(fmaxf(p, 1e-5f));
    entropy[((size_t)b * H + y) * W + x] = e;                          // bins may overlap by one row/col: same value rewritten
    acc += e;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) scores[((size_t)b * ws + bi) * ws + bj] = acc / (float)n;
}

// ------------------------------------------------------------------ depthwise 7x7 conv + 1x1 mask head (CSF.py:13-25,41-42)
// x token-major [Nw, H, W, C]; dwT [49][C] (tap-major copy of depthwise_conv.weight), dwb [C], w1 [C], b1.
// one workgroup per output pixel, lane = channel (coalesced); out[i][y][x] = b1 + sum_c w1[c]*(dwb[c] + sum_taps dw*x).
__global__ __launch_bounds__(256) void dwconv7_maskdec_kernel(const float* __restrict__ x, const float* __restrict__ dwT,
                                                              const float* __restrict__ dwb, const float* __restrict__ w1, float b1,
                                                              float* __restrict__ out, int H, int W, int C) {
  __shared__ float red[16];
  const int i = blockIdx.z, oy = blockIdx.y, ox = blockIdx.x;
  const float* xi = x + (size_t)i * H * W * C;
  float total = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = dwb[c];
    for (int ky = 0; ky < 7; ++ky) {
      const int yy = oy + ky - 3;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        const int xx = ox + kx - 3;
        if (xx < 0 || xx >= W) continue;
        acc = fmaf(dwT[(ky * 7 + kx) * C + c], xi[((size_t)yy * W + xx) * C + c], acc);
      }
    }
    total = fmaf(w1[c], acc, total);
  }
  total = block_sum(total, red);
  if (threadIdx.x == 0) out[((size_t)i * H + oy) * W + ox] = total + b1;
}

// ------------------------------------------------------------------ window scatter-average (HRE.py:18-39)
// out [B,1,ws*H,ws*W] zero-filled, then window i placed at (coords[i].y*H, coords[i].x*W) of image win_img[i], / (1 + 1e-6).
__global__ __launch_bounds__(256) void window_scatter_kernel(const float* __restrict__ win, const int* __restrict__ coords,
                                                             const int* __restrict__ win_img, float* __restrict__ out, int H, int W, int ws) {
  const int i = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  const int Y = coords[2 * i] * H + y, X = coords[2 * i + 1] * W + x;
  out[((size_t)win_img[i] * ws * H + Y) * (ws * W) + X] = win[(size_t)i * H * W + p] / (1.0f + 1e-6f);
}

// ------------------------------------------------------------------ gated ensembling (GE_pix_level.py:16-25)
// pass 0: p = sigmoid(l1) once per pixel (round 3: the entropy pass evaluated the sigmoid of all 361 neighbours of every pixel), psum[b] += p.
// pass 1: en = -m*log(max(m,1e-5)), m = 19x19 average of p (zero padded, /361), summed in the same (dy, dx) order as before; enmax = max en.
__global__ __launch_bounds__(256) void ge_prob_kernel(const float* __restrict__ l1, float* __restrict__ pm, float* __restrict__ psum, int hw) {
  __shared__ float red[16];
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  float pv = 0.f;
  if (p < hw) {
    pv = sigmoid_acc(l1[(size_t)b * hw + p]);
    pm[(size_t)b * hw + p] = pv;
  }
  const float bs = block_sum(pv, red);
  if (threadIdx.x == 0) atomicAdd(&psum[b], bs);
}

__global__ __launch_bounds__(256) void ge_stats_kernel(const float* __restrict__ pm, float* __restrict__ en, unsigned* __restrict__ enmax_bits,
                                                       int h, int w) {
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  float ev = 0.f;
  if (p < h * w) {
    const int y = p / w, x = p - y * w;
    const float* pb = pm + (size_t)b * h * w;
    float s = 0.f;
    for (int dy = -9; dy <= 9; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= h) continue;                            // (wave-uniform only by luck; no load in this branch's condition)
      const float* prow = pb + yy * w;
      float t19[19];
#pragma unroll
      for (int dx = -9; dx <= 9; ++dx) {
        const int xx = x + dx;
        t19[dx + 9] = prow[xx < 0 ? 0 : (xx >= w ? w - 1 : xx)];   // clamped: the 19 loads of a row go out together
      }
#pragma unroll
      for (int dx = -9; dx <= 9; ++dx) {
        const int xx = x + dx;
        if (xx >= 0 && xx < w) s += t19[dx + 9];
      }
    }
    const float m = s / 361.0f;
    ev = -m * logf(fmaxf(m, 1e-5f));
    en[(size_t)b * h * w + p] = ev;
  }
  float mx = ev;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(enmax_bits, __float_as_uint(fmaxf(mx, 0.f)));   // en >= 0: uint order == float order
}

// pass 2: weight = ((1 - en/enmax) + mean_p)/2; y = l1*w + l2*(1-w); out = fuser(y) (1 -> 64 ReLU -> 1, 1x1 convs)
__global__ __launch_bounds__(256) void ge_fuse_kernel(const float* __restrict__ l1, const float* __restrict__ l2, const float* __restrict__ en,
                                                      const float* __restrict__ psum, const unsigned* __restrict__ enmax_bits,
                                                      const float* __restrict__ f0w, const float* __restrict__ f0b, const float* __restrict__ f2w,
                                                      float f2b, float* __restrict__ out, float* __restrict__ wout, int h, int w) {
  __shared__ float a[64], c[64], d[64];
  if (threadIdx.x < 64) { a[threadIdx.x] = f0w[threadIdx.x]; c[threadIdx.x] = f0b[threadIdx.x]; d[threadIdx.x] = f2w[threadIdx.x]; }
  __syncthreads();
  const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= h * w) return;
  const size_t i = (size_t)b * h * w + p;
  const float enmax = __uint_as_float(*enmax_bits);
  const float wt = ((1.f - en[i] / enmax) + psum[b] / (float)(h * w)) * 0.5f;
  const float y = l1[i] * wt + l2[i] * (1.f - wt);
  float o = f2b;
#pragma unroll 8
  for (int k = 0; k < 64; ++k) o = fmaf(d[k], fmaxf(fmaf(a[k], y, c[k]), 0.f), o);
  out[i] = o;
  wout[i] = wt;
}


// SparseRefiner.cal_ex_loss (models/UDLR.py:52-75), training mode: per selected window i (one workgroup each)
//   l = (sigmoid(l_up) > 0.5) of that window  (l_up = the first-stage logits, bilinear-resized to [B,1,ws*H,ws*W]; a window is the
//       (wy, wx) = (j / ws, j % ws) block of image b, j = the window's raster index: the unfold of UDLR.py:67),
//   t = h_targets[win_flat[i]]  (the window's own high-resolution target),
//   iou = |bin(t) & l| / (|bin(t) | l| + 1e-6), bin(t) = (t' > 0.5) with t' = sigmoid(t) if `targets_are_logits` (binary_iou's
//       "preds.max() > 1" heuristic, decided by the caller over ALL selected targets) else t;  w = clamp(1.5 * iou, 0, 1)
//   part[i] = sum_pixels  w * BCEWithLogits(x, t) + (1 - w) * BCEWithLogits(x, l),   x = window_preds[i]
// and the caller divides the sum of part by (2 * n * H * W)  (the reference's .mean() / 2).  Fixed reduction trees: deterministic.
__global__ __launch_bounds__(256) void window_loss_kernel(const float* __restrict__ win_preds, const float* __restrict__ h_targets,
                                                          const int* __restrict__ win_flat, const float* __restrict__ l_up, int targets_are_logits,
                                                          float* __restrict__ part, float* __restrict__ iou_out, int H, int W, int ws) {
  __shared__ float red[16];
  const int i = blockIdx.x, HW = H * W;
  const int flat = win_flat[i], b = flat / (ws * ws), j = flat - b * (ws * ws);
  const int wy = j / ws, wx = j - wy * ws;
  const float* x = win_preds + (size_t)i * HW;
  const float* t = h_targets + (size_t)flat * HW;
  const float* l = l_up + ((size_t)b * ws * H + (size_t)wy * H) * (ws * W) + (size_t)wx * W;
  float inter = 0.f, uni = 0.f;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, xx = p - y * W;
    const float tv = t[p];
    const bool tb = (targets_are_logits ? sigmoid_acc(tv) : tv) > 0.5f;
    const bool lb = sigmoid_acc(l[(size_t)y * (ws * W) + xx]) > 0.5f;
    inter += (tb && lb) ? 1.f : 0.f;
    uni += (tb || lb) ? 1.f : 0.f;
  }
  inter = block_sum(inter, red);
  uni = block_sum(uni, red);
  const float iou = inter / (uni + 1e-6f);
  const float w = fminf(fmaxf(iou * 1.5f, 0.f), 1.f);
  float acc = 0.f;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / W, xx = p - y * W;
    const float xv = x[p], tv = t[p];
    const float lv = sigmoid_acc(l[(size_t)y * (ws * W) + xx]) > 0.5f ? 1.f : 0.f;
    // BCEWithLogits(x, z) = max(x, 0) - x z + log1p(exp(-|x|))   (ATen's stable form)
    const float sp = fmaxf(xv, 0.f) + log1pf(expf(-fabsf(xv)));
    acc += w * (sp - xv * tv) + (1.f - w) * (sp - xv * lv);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) {
    part[i] = acc;
    if (iou_out) iou_out[i] = iou;
  }
}

// loss = (sum_i part[i]) * scale, summed in f64 in a fixed order (one wave)
__global__ void window_loss_finish_kernel(const float* __restrict__ part, float* __restrict__ loss, int n, double scale) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) acc += (double)part[i];
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) loss[0] = (float)(acc * scale);
}

}  // namespace ucod

using namespace ucod;

extern "C" int ucod_gather_tokens(const float* src, const int* src_idx, float* out, int n, int C, int HW, void* stream) {
  if (!src || !src_idx || !out || n <= 0 || C <= 0 || HW <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(gather_tokens_kernel, dim3(cdiv(HW, 64), cdiv(C, 64), n), dim3(256), 0, (hipStream_t)stream, src, src_idx, out, C, HW);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_entropy_scores(const float* preds, int use_sigmoid, float* entropy, float* scores, int B, int H, int W, int ws, void* stream) {
  if (!preds || !entropy || !scores || B <= 0 || H <= 0 || W <= 0 || ws <= 0 || ws > H || ws > W) return UCOD_EINVAL;
  hipLaunchKernelGGL(entropy_scores_kernel, dim3(ws, ws, B), dim3(256), 0, (hipStream_t)stream, preds, use_sigmoid, entropy, scores, H, W, ws);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_dwconv7_maskdec(const float* x_tokens, const float* dw_tapmajor, const float* dw_bias, const float* w1, float b1, float* out,
                                    int n, int H, int W, int C, void* stream) {
  if (!x_tokens || !dw_tapmajor || !dw_bias || !w1 || !out || n <= 0 || H <= 0 || W <= 0 || C <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(dwconv7_maskdec_kernel, dim3(W, H, n), dim3(256), 0, (hipStream_t)stream, x_tokens, dw_tapmajor, dw_bias, w1, b1, out, H, W, C);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_window_scatter(const float* windows, const int* coords, const int* win_img, float* out, int n, int B, int H, int W, int ws,
                                   void* stream) {
  if (!out || B <= 0 || H <= 0 || W <= 0 || ws <= 0 || n < 0) return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * ws * H * ws * W, s);
  if (e != hipSuccess) return (int)e;
  if (n == 0) return UCOD_OK;
  if (!windows || !coords || !win_img) return UCOD_EINVAL;
  hipLaunchKernelGGL(window_scatter_kernel, dim3(cdiv(H * W, 256), n), dim3(256), 0, s, windows, coords, win_img, out, H, W, ws);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_gated_ensemble(const float* l1_up, const float* l2, const float* f0w, const float* f0b, const float* f2w, float f2b,
                                   float* out, float* weight_out, void* ws, int B, int h, int w, void* stream) {
  if (!l1_up || !l2 || !f0w || !f0b || !f2w || !out || !weight_out || !ws || B <= 0 || h <= 0 || w <= 0) return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* en = (float*)ws;
  float* psum = en + (size_t)B * h * w;
  unsigned* enmax = (unsigned*)(psum + B);
  float* pm = psum + B + 1;                                       // sigmoid(l1), B*h*w
  hipError_t e = hipMemsetAsync(psum, 0, sizeof(float) * (B + 1), s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(ge_prob_kernel, dim3(cdiv(h * w, 256), B), dim3(256), 0, s, l1_up, pm, psum, h * w);
  hipLaunchKernelGGL(ge_stats_kernel, dim3(cdiv(h * w, 256), B), dim3(256), 0, s, pm, en, enmax, h, w);
  hipLaunchKernelGGL(ge_fuse_kernel, dim3(cdiv(h * w, 256), B), dim3(256), 0, s, l1_up, l2, en, psum, enmax, f0w, f0b, f2w, f2b, out, weight_out, h, w);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" size_t ucod_gated_ensemble_workspace_bytes(int B, int h, int w) { return ((size_t)2 * B * h * w + B + 1) * sizeof(float); }

extern "C" int ucod_window_loss(const float* window_preds, const float* h_targets, const int* win_flat, const float* l_up, int targets_are_logits,
                                float* part, float* iou_out, float* loss_out, int n, int B, int H, int W, int ws, void* stream) {
  if (!window_preds || !h_targets || !win_flat || !l_up || !part || !loss_out || n <= 0 || B <= 0 || H <= 0 || W <= 0 || ws <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(ucod::window_loss_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, window_preds, h_targets, win_flat, l_up, targets_are_logits,
                     part, iou_out, H, W, ws);
  hipLaunchKernelGGL(ucod::window_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, loss_out, n, 1.0 / (2.0 * n * H * W));
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
