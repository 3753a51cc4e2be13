// Pseudo-label generator on the GPU: the upstream producer of the `pseudo_label` tensors the training step
// consumes (generate_pseudo_label.py:71-94 -> data/utils/found_bkg_mask.py:4-86).
//   ucod_cls_qk         q and k projections of the CLS token of the LAST layer (the only query row the generator reads from
//                       outputs.attentions[-1], found_bkg_mask.py:23) from that layer's LN1 output, which ucod_vit_forward leaves
//                       in its workspace;
//   ucod_cls_attention  softmax over ALL keys (CLS + patches) of that query per head, patch columns kept -- from the NCHW key map
//                       the backbone pass already produced (one token-contiguous row per channel: coalesced);
//   ucod_bkg_seg        CroW sparsity weights, seed = least-attended patch, cosine similarity of every patch's (weighted) key
//                       descriptor to the seed's, background mask and similarity map.  Only the seed's row of the HW x HW
//                       similarity matrix is formed (the reference builds all of it and reads one row).  One workgroup per image.
// Everything is exact f32.  up_size (the optional bilinear upsampling of found_bkg_mask.py:26,50) is not built: the generator
// calls it with up_size = grid (generate_pseudo_label.py:83-89).
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {
namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) s = fmaxf(s, red[i]);
  return s;
}

// one block per image; wave per output row of [Wq ; Wk]
__global__ __launch_bounds__(256) void cls_qk_kernel(const bf16_raw* __restrict__ h, const bf16_raw* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ q, float* __restrict__ k, int tok, int D) {
  extern __shared__ float x[];                               // LN1(x_cls) as f32
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bf16_raw* hr = h + (size_t)b * tok * D;
  for (int i = threadIdx.x; i < D; i += 256) x[i] = h_to_f32(hr[i]);
  __syncthreads();
  for (int n = wv; n < 2 * D; n += 4) {
    const bf16_raw* wr = w + (size_t)n * D;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) s += h_to_f32(wr[i]) * x[i];
    s = wave_sum(s) + bias[n];
    if (lane == 0) {
      if (n < D) q[(size_t)b * D + n] = s;
      else k[(size_t)b * D + n - D] = s;
    }
  }
}

// grid (B, heads); att[b][h][j] for patches j (CLS column dropped after the softmax)
__global__ __launch_bounds__(256) void cls_attention_kernel(const float* __restrict__ q, const float* __restrict__ kc, const float* __restrict__ key,
                                                            float* __restrict__ att, int heads, int hw, float scale) {
  extern __shared__ float sc[];                              // hw scores, then 8 floats of reduction space
  float* red = sc + hw;
  const int b = blockIdx.x, hd = blockIdx.y, D = heads * 64;
  const float* qh = q + (size_t)b * D + hd * 64;
  const float* kb = key + ((size_t)b * D + hd * 64) * hw;
  float mx = -3.0e38f;
  for (int j = threadIdx.x; j < hw; j += 256) {
    float s = 0.f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) s += qh[d] * kb[(size_t)d * hw + j];
    s *= scale;
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  float s_cls = 0.f;
  for (int d = 0; d < 64; ++d) s_cls += qh[d] * kc[(size_t)b * D + hd * 64 + d];
  s_cls *= scale;
  mx = fmaxf(block_max(mx, red), s_cls);
  float sum = 0.f;
  for (int j = threadIdx.x; j < hw; j += 256) {
    const float e = __expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = block_sum(sum, red) + __expf(s_cls - mx);
  const float inv = 1.0f / sum;
  float* out = att + ((size_t)b * heads + hd) * hw;
  for (int j = threadIdx.x; j < hw; j += 256) out[j] = sc[j] * inv;
}

// one block per image.  cosr[b][j] = cosine similarity to the seed patch; mask = cos > th; simr = 1 - cos (normalised by the
// batch-wide maximum in bkg_finalize_kernel, as `sim_map.max()` is over the whole batch, found_bkg_mask.py:81).
__global__ __launch_bounds__(256) void bkg_seg_kernel(const float* __restrict__ att, const float* __restrict__ key, float* __restrict__ mask,
                                                      float* __restrict__ simr, float* __restrict__ cosr, int* __restrict__ seed, float* __restrict__ beta_out,
                                                      unsigned* __restrict__ gmax, int heads, int hw, float th, float eps, int apply_w) {
  extern __shared__ float sm[];
  float* red = sm;                                           // 8
  float* beta = sm + 8;                                      // heads
  float* qn = beta + heads;                                  // heads (sparsity counts)
  const int b = blockIdx.x, D = heads * 64;
  const float* ab = att + (size_t)b * heads * hw;
  const float* kb = key + (size_t)b * D * hw;
  // threshold = mean attention of the image
  float s = 0.f;
  for (int i = threadIdx.x; i < heads * hw; i += 256) s += ab[i];
  const float thr = block_sum(s, red) / (float)(heads * hw);
  for (int h = 0; h < heads; ++h) {
    float c = 0.f;
    for (int j = threadIdx.x; j < hw; j += 256) c += ab[(size_t)h * hw + j] > thr ? 1.f : 0.f;
    c = block_sum(c, red);
    if (threadIdx.x == 0) qn[h] = c / (float)hw;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int h = 0; h < heads; ++h) tot += qn[h] + eps;
    for (int h = 0; h < heads; ++h) {
      beta[h] = logf(tot / (qn[h] + eps));
      beta_out[(size_t)b * heads + h] = beta[h];
    }
  }
  __syncthreads();
  // seed: first patch with the smallest (weighted) total attention
  float best = 3.0e38f;
  int bi = 0x7fffffff;
  for (int j = threadIdx.x; j < hw; j += 256) {
    float t = 0.f;
    for (int h = 0; h < heads; ++h) t += apply_w ? ab[(size_t)h * hw + j] * beta[h] : ab[(size_t)h * hw + j];
    if (t < best) { best = t; bi = j; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  __shared__ float wb[4];
  __shared__ int wi[4];
  if ((threadIdx.x & 63) == 0) { wb[threadIdx.x >> 6] = best; wi[threadIdx.x >> 6] = bi; }
  __syncthreads();
  best = wb[0];
  bi = wi[0];
  for (int w = 1; w < 4; ++w)
    if (wb[w] < best || (wb[w] == best && wi[w] < bi)) { best = wb[w]; bi = wi[w]; }
  const int ref = bi;
  if (threadIdx.x == 0) seed[b] = ref;
  // cosine similarity of every patch descriptor to the seed's
  float lmax = 0.f;
  for (int j = threadIdx.x; j < hw; j += 256) {
    float dot = 0.f, nj = 0.f, nr = 0.f;
    for (int c = 0; c < D; ++c) {
      const float w = apply_w ? beta[c >> 6] : 1.f;
      const float vj = kb[(size_t)c * hw + j] * w, vr = kb[(size_t)c * hw + ref] * w;
      dot += vj * vr;
      nj += vj * vj;
      nr += vr * vr;
    }
    const float cs = dot / (fmaxf(sqrtf(nj), 1e-12f) * fmaxf(sqrtf(nr), 1e-12f));
    cosr[(size_t)b * hw + j] = cs;
    mask[(size_t)b * hw + j] = cs > th ? 1.f : 0.f;
    const float sv = 1.f - cs;
    simr[(size_t)b * hw + j] = sv;
    lmax = fmaxf(lmax, sv);
  }
  lmax = block_max(lmax, red);
  if (threadIdx.x == 0) atomicMax(gmax, __float_as_uint(fmaxf(lmax, 0.f)));
}

__global__ __launch_bounds__(256) void bkg_finalize_kernel(const float* __restrict__ mask, float* __restrict__ sim, const unsigned* __restrict__ gmax, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float m = __uint_as_float(*gmax);
  sim[i] = sim[i] / (m + 1e-10f) * (1.f - mask[i]);
}

}  // namespace
}  // namespace ucod

using namespace ucod;

extern "C" int ucod_cls_qk(const void* h_ln1_bf16, const void* qkv_w_bf16, const float* qkv_b, float* q_cls, float* k_cls, int B, int tok, int D,
                           void* stream) {
  if (!h_ln1_bf16 || !qkv_w_bf16 || !qkv_b || !q_cls || !k_cls || B <= 0 || tok <= 0 || D <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(cls_qk_kernel, dim3(B), dim3(256), (size_t)D * sizeof(float), (hipStream_t)stream, (const bf16_raw*)h_ln1_bf16,
                     (const bf16_raw*)qkv_w_bf16, qkv_b, q_cls, k_cls, tok, D);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_cls_attention(const float* q_cls, const float* k_cls, const float* key_map, float* att, int B, int heads, int hw, float scale,
                                  void* stream) {
  if (!q_cls || !k_cls || !key_map || !att || B <= 0 || heads <= 0 || hw <= 0 || hw > 12000) return UCOD_EINVAL;
  hipLaunchKernelGGL(cls_attention_kernel, dim3(B, heads), dim3(256), (size_t)(hw + 8) * sizeof(float), (hipStream_t)stream, q_cls, k_cls, key_map, att,
                     heads, hw, scale);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_bkg_seg(const float* att, const float* key_map, float th_bkg, float epsilon, int apply_weights, float* bkg_mask, float* sim_map,
                            float* cos_row, int* seed, float* beta, void* scratch4, int B, int heads, int hw, void* stream) {
  if (!att || !key_map || !bkg_mask || !sim_map || !cos_row || !seed || !beta || !scratch4 || B <= 0 || heads <= 0 || heads > 64 || hw <= 0)
    return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const hipError_t e = hipMemsetAsync(scratch4, 0, 4, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(bkg_seg_kernel, dim3(B), dim3(256), (size_t)(8 + 2 * heads) * sizeof(float), s, att, key_map, bkg_mask, sim_map, cos_row, seed, beta,
                     (unsigned*)scratch4, heads, hw, th_bkg, epsilon, apply_weights);
  const size_t n = (size_t)B * hw;
  hipLaunchKernelGGL(bkg_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, bkg_mask, sim_map, (const unsigned*)scratch4, n);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
