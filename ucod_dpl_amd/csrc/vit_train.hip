// Backbone-backward mode (models/modules/full_model.py:47-72,79-126: LoRA r=2, alpha=4 on the query /
// key / value projections of every layer, everything else frozen).  Row-wise kernels of the training pass; the GEMMs and
// the attention kernels live in gemm_bf16.hip / attention.hip / attention_bwd.hip, the pass itself in vit_train_driver.hip.
//
// LoRA rides on the big GEMMs as 64 extra K columns ("aug" columns), so no GEMM kernel knows about it:
//   forward   h_aug  [M, D+64]  = [ LN1(x) | u = LN1(x) A^T (3r values: q,k,v) | 0 ]        (ucod_layernorm_lora)
//             Wqkv_aug [3D, D+64] = [ Wqkv | alpha/r * B on the block diagonal | 0 ]         (ucod_lora_pack)
//             qkv = h_aug Wqkv_aug^T  ==  LN1(x) Wqkv^T + alpha/r * (LN1(x) A^T) B^T
//   backward  dqkv_aug [M, 3D+64] = [ dqkv | t = alpha/r * dqkv B (3r values) | 0 ]          (ucod_lora_grad)
//             WqkvT_aug [D, 3D+64] = [ Wqkv^T | A^T | 0 ]                                     (ucod_lora_pack)
//             dLN1 = dqkv_aug WqkvT_aug^T  ==  dqkv Wqkv + t A
//             dA = t^T LN1(x),  dB = alpha/r * dqkv^T u                                        (ucod_lora_grad)
// Parameter layout of one layer in the flat LoRA arena (f32): [A_q (r x D) | B_q (D x r) | A_k | B_k | A_v | B_v].
// With LoRA dropout on, u is computed from the dropped LN1(x) (one mask per projection), dA from the same dropped input, and the
// t A term of dLN1 is added -- masked -- by the LayerNorm-1 backward instead of the dgrad GEMM (A^T columns packed as zeros).
#include <type_traits>
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

constexpr int AUG = UCOD_LORA_AUG;

// LoRA dropout (LoraConfig.lora_dropout of models/modules/full_model.py:50,63: nn.Dropout on the input of every lora_A, one
// independent mask per target module).  Counter-based: the keep decision of element (row, col) of projection p in layer l is a
// pure function of (seed, 3*l + p, row*D + col), so forward and backward regenerate the same mask and nothing is stored.
// ABI 3 (round 4): ONE 32-bit mix per element serves the three projections -- bits 0-9 decide query, 10-19 key, 20-29 value, each against
// floor(p * 1024) -- instead of one mix per projection: 3 instead of 9 quarter-rate integer multiplies per element in ln_lora / ln_bwd
// (the hash was half of ln_lora's time).  The drop probability is therefore p_eff = floor(1024 p) / 1024 and the kept
// elements are scaled by 1 / (1 - p_eff), so the mask stays unbiased.
struct Drop {
  unsigned seed_lo, seed_hi, thresh;     // drop iff the projection's 10-bit field < thresh  (thresh = floor(p * 1024))
  int key0;                              // layer
  float inv_keep;                        // 1 / (1 - thresh / 1024); 0 thresh = dropout off
};
__host__ __device__ __forceinline__ unsigned drop_hash(unsigned seed_lo, unsigned seed_hi, unsigned key, unsigned idx) {
  unsigned h = seed_lo ^ (idx * 0x9E3779B1u);
  h ^= seed_hi + key * 0x85EBCA77u;
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float drop_scale(const Drop& d, int p, unsigned idx) {
  // (the compiler shares the mix between the three projections of one element: a pure function of (d, idx))
  return ((drop_hash(d.seed_lo, d.seed_hi, (unsigned)d.key0, idx) >> (10 * p)) & 1023u) < d.thresh ? 0.f : d.inv_keep;
}
static Drop make_drop(const ucod_lora_dropout* dd) {
  Drop d{0u, 0u, 0u, 0, 1.f};
  if (dd && dd->p > 0.f) {
    d.seed_lo = (unsigned)(dd->seed & 0xFFFFFFFFull);
    d.seed_hi = (unsigned)(dd->seed >> 32);
    const double t = (double)dd->p * 1024.0;
    d.thresh = t >= 1023.0 ? 1023u : (unsigned)t;
    d.key0 = dd->layer;
    d.inv_keep = 1.0f / (1.0f - (float)d.thresh / 1024.0f);
  }
  return d;
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm forward + LoRA down-projection.  One wave per row, D = 128*NCH, row in registers (as layernorm_kernel).
// ---------------------------------------------------------------------------------------------------------------------
// XH16: the residual stream is IEEE fp16 (the no-grad pass of the EMA teacher, ucod_vit_forward_lora_infer with resid16) instead of f32
template <int NCH, bool XH16 = false>
__global__ __launch_bounds__(256) void ln_lora_kernel(const void* __restrict__ x_, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ lora,
                                                      int r, bf16_raw* __restrict__ y, int rows, int D, float eps, Drop drop) {
  // the 3r LoRA A rows live in LDS for the whole block (they were re-read from L2 for every row: 92 us against 33 us for the plain
  // LayerNorm); each wave walks rows block-stride, two rows in flight (as layernorm_kernel)
  extern __shared__ float a_lds[];                               // [3r][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 3 * r * D; i += 256) {
    const int p = i / (r * D), rem = i - p * r * D;               // A_p at lora + p*2*r*D
    a_lds[i] = lora[(size_t)p * 2 * r * D + rem];
  }
  __syncthreads();
  constexpr int R = 2;
  const float2* g2 = reinterpret_cast<const float2*>(gamma);
  const float2* b2 = reinterpret_cast<const float2*>(beta);
  for (int row0 = (blockIdx.x * 4 + wave) * R; row0 < rows; row0 += gridDim.x * 4 * R) {
    float2 v[R][NCH];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int row = (row0 + q) < rows ? (row0 + q) : rows - 1;
      if constexpr (XH16) {
        const unsigned* xr = reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned short*>(x_) + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < NCH; ++i) unpack_f16x2(xr[lane + 64 * i], v[q][i].x, v[q][i].y);
      } else {
        const float2* xr = reinterpret_cast<const float2*>(reinterpret_cast<const float*>(x_) + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < NCH; ++i) v[q][i] = xr[lane + 64 * i];
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int row = row0 + q;
      if (row >= rows) break;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) s += v[q][i].x + v[q][i].y;
      const float mean = wave_sum(s) / (float)D;
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const float a = v[q][i].x - mean, b = v[q][i].y - mean;
        qq += a * a + b * b;
      }
      const float rstd = rsqrtf(wave_sum(qq) / (float)D + eps);
      bf16_raw* yr = y + (size_t)row * (D + AUG);
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const float2 g = g2[lane + 64 * i], b = b2[lane + 64 * i];
        v[q][i].x = (v[q][i].x - mean) * rstd * g.x + b.x;
        v[q][i].y = (v[q][i].y - mean) * rstd * g.y + b.y;
        reinterpret_cast<unsigned*>(yr)[lane + 64 * i] = pack_bf16x2(v[q][i].x, v[q][i].y);
      }
      // u[j] = <dropout_p(LN(x)), A[j]>, j = p*r + rank
      float mine = 0.f;
      for (int p = 0; p < 3; ++p) {
        float2 vm[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          vm[i] = v[q][i];
          if (drop.thresh) {
            const unsigned idx = (unsigned)row * (unsigned)D + 2u * (unsigned)(lane + 64 * i);
            vm[i].x *= drop_scale(drop, p, idx);
            vm[i].y *= drop_scale(drop, p, idx + 1u);
          }
        }
        for (int jr = 0; jr < r; ++jr) {
          const int j = p * r + jr;
          const float2* a2 = reinterpret_cast<const float2*>(a_lds + (size_t)j * D);
          float d = 0.f;
#pragma unroll
          for (int i = 0; i < NCH; ++i) {
            const float2 a = a2[lane + 64 * i];
            d += vm[i].x * a.x + vm[i].y * a.y;
          }
          d = wave_sum(d);
          if (lane == j) mine = d;
        }
      }
      yr[D + lane] = f32_to_bf16(lane < 3 * r ? mine : 0.f);
    }
  }
}

// The same for D % 256 == 0 (ViT-B 768, ViT-L 1024) with a lane owning FOUR consecutive columns per chunk (round 4): 16-byte (f32) / 8-byte (fp16)
// loads, 8-byte stores instead of 4-byte ones, and the dropout mix of an element computed ONCE for its three masks (in the form above the compiler
// shared it between only some of the projections: 177 v_mul_lo_u32 per two rows instead of 72).
template <int NV, bool XH16>
__global__ __launch_bounds__(256) void ln_lora4_kernel(const void* __restrict__ x_, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ lora,
                                                       int r, bf16_raw* __restrict__ y, int rows, int D, float eps, Drop drop) {
  extern __shared__ float a_lds[];                               // [3r][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 3 * r * D; i += 256) {
    const int p = i / (r * D), rem = i - p * r * D;
    a_lds[i] = lora[(size_t)p * 2 * r * D + rem];
  }
  __syncthreads();
  constexpr int R = 2;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4* g4 = reinterpret_cast<const f4*>(gamma);
  const f4* b4 = reinterpret_cast<const f4*>(beta);
  f4 gm[NV], bt[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { gm[i] = g4[lane + 64 * i]; bt[i] = b4[lane + 64 * i]; }
  for (int row0 = (blockIdx.x * 4 + wave) * R; row0 < rows; row0 += gridDim.x * 4 * R) {
    f4 v[R][NV];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int row = (row0 + q) < rows ? (row0 + q) : rows - 1;
      if constexpr (XH16) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4* xr = reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(x_) + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const h4 h = xr[lane + 64 * i];
          v[q][i] = (f4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        }
      } else {
        const f4* xr = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(x_) + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[q][i] = xr[lane + 64 * i];
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int row = row0 + q;
      if (row >= rows) break;
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) s += (v[q][i][0] + v[q][i][1]) + (v[q][i][2] + v[q][i][3]);
      const float mean = wave_sum(s) / (float)D;
      float qq = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = v[q][i][e] - mean;
          qq += a * a;
        }
      const float rstd = rsqrtf(wave_sum(qq) / (float)D + eps);
      bf16_raw* yr = y + (size_t)row * (D + AUG);
      unsigned hsh[NV][4];                                        // the element's one mix: bits 0-9 query, 10-19 key, 20-29 value
#pragma unroll
      for (int i = 0; i < NV; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[q][i][e] = (v[q][i][e] - mean) * rstd * gm[i][e] + bt[i][e];
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        reinterpret_cast<u2*>(yr)[lane + 64 * i] = (u2){pack_bf16x2(v[q][i][0], v[q][i][1]), pack_bf16x2(v[q][i][2], v[q][i][3])};
        if (drop.thresh) {
          const unsigned idx = (unsigned)row * (unsigned)D + 4u * (unsigned)(lane + 64 * i);
#pragma unroll
          for (int e = 0; e < 4; ++e) hsh[i][e] = drop_hash(drop.seed_lo, drop.seed_hi, (unsigned)drop.key0, idx + (unsigned)e);
        }
      }
      float mine = 0.f;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        f4 vm[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          vm[i] = v[q][i];
          if (drop.thresh) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vm[i][e] *= ((hsh[i][e] >> (10 * p)) & 1023u) < drop.thresh ? 0.f : drop.inv_keep;
          }
        }
        for (int jr = 0; jr < r; ++jr) {
          const int j = p * r + jr;
          const f4* a4 = reinterpret_cast<const f4*>(a_lds + (size_t)j * D);
          float d = 0.f;
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const f4 a = a4[lane + 64 * i];
            d += (vm[i][0] * a[0] + vm[i][1] * a[1]) + (vm[i][2] * a[2] + vm[i][3] * a[3]);
          }
          d = wave_sum(d);
          if (lane == j) mine = d;
        }
      }
      yr[D + lane] = f32_to_bf16(lane < 3 * r ? mine : 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm backward w.r.t. its input (gamma / beta are frozen), fused with the residual add and with the column scale +
// bf16 cast the NEXT dgrad GEMM wants as its A operand:
//   g = dy * gamma;  dx_ln = rstd * (g - mean(g) - xhat * mean(g * xhat));   dx = dres + dx_ln;   s = bf16(scale * dx)
// ---------------------------------------------------------------------------------------------------------------------
// LORA: dy additionally receives the dropout-masked LoRA branch  sum_p mask_p/keep * (t_p A_p)  (t = aug columns of dqkv_aug); with
// dropout off that term rides on the dgrad GEMM instead (A^T columns of WqkvT_aug) and this kernel is launched with LORA = false.
// One wave per row; a lane owns NCH chunks of VW consecutive floats (VW = 4: 16-byte accesses, rows of a multiple of 256 floats; VW = 2 for
// the other widths).  Everything the row needs from memory -- x, dy, the residual cotangent -- is requested before the first reduction, so
// that one latency covers all three streams (round 3: 103 -> see profiles/r03_ln_bwd.txt).
// DYB: dy arrives as bf16 (the dgrad GEMMs of the backward driver write their output in the 16-bit type: half the bytes on both sides).
// XH: x (the saved LayerNorm input = the residual stream) is IEEE fp16 (training pass with vit.resid16).
template <int NCH, int VW, bool LORA, bool DYB, bool XH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_, const void* __restrict__ x_,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     const float* __restrict__ scale, float* __restrict__ dx,
                                                     bf16_raw* __restrict__ sout, int rows, int D, float eps,
                                                     const bf16_raw* __restrict__ tq, int ldt, const float* __restrict__ lora, int r, Drop drop) {
  typedef float vecf __attribute__((ext_vector_type(VW)));
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const vecf* xr = reinterpret_cast<const vecf*>(reinterpret_cast<const float*>(x_) + (size_t)row * D);
  typedef unsigned short vech __attribute__((ext_vector_type(VW)));
  const vecf* dyr = reinterpret_cast<const vecf*>(reinterpret_cast<const float*>(dy_) + (size_t)row * D);
  const vech* dyh = reinterpret_cast<const vech*>(reinterpret_cast<const bf16_raw*>(dy_) + (size_t)row * D);
  const vecf* rr = dres ? reinterpret_cast<const vecf*>(dres + (size_t)row * D) : nullptr;
  const vecf* g2 = reinterpret_cast<const vecf*>(gamma);
  vecf v[NCH], g[NCH], res[NCH];
  if constexpr (XH) {
    typedef _Float16 vecx __attribute__((ext_vector_type(VW)));
    const vecx* xh = reinterpret_cast<const vecx*>(reinterpret_cast<const _Float16*>(x_) + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const vecx h = xh[lane + 64 * i];
#pragma unroll
      for (int e = 0; e < VW; ++e) v[i][e] = (float)h[e];
    }
  } else {
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = xr[lane + 64 * i];
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if constexpr (DYB) {
      const vech h = dyh[lane + 64 * i];
#pragma unroll
      for (int e = 0; e < VW; ++e) g[i][e] = bf16_to_f32(h[e]);
    } else {
      g[i] = dyr[lane + 64 * i];
    }
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) res[i] = rr ? rr[lane + 64 * i] : (vecf)(0.f);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    vecf d = g[i];
    if constexpr (LORA) {
      const unsigned idx = (unsigned)row * (unsigned)D + (unsigned)VW * (unsigned)(lane + 64 * i);
      auto branch = [&](auto rc) {                                // rank known at compile time: the 3 r coefficient / A-row loads go out together
        constexpr int R = decltype(rc)::value;
        vecf av[3][R];
        float t[3][R];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int jr = 0; jr < R; ++jr) {
            t[p][jr] = bf16_to_f32(tq[(size_t)row * ldt + p * R + jr]);
            av[p][jr] = reinterpret_cast<const vecf*>(lora + (size_t)p * 2 * R * D + (size_t)jr * D)[lane + 64 * i];
          }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          vecf acc = (vecf)(0.f);
#pragma unroll
          for (int jr = 0; jr < R; ++jr) acc += t[p][jr] * av[p][jr];
#pragma unroll
          for (int e = 0; e < VW; ++e) d[e] += acc[e] * drop_scale(drop, p, idx + (unsigned)e);
        }
      };
      if (r == 2) branch(std::integral_constant<int, 2>{});
      else if (r == 4) branch(std::integral_constant<int, 4>{});
      else if (r == 1) branch(std::integral_constant<int, 1>{});
      else {
        for (int p = 0; p < 3; ++p) {
          vecf acc = (vecf)(0.f);
          for (int jr = 0; jr < r; ++jr) {
            const float t = bf16_to_f32(tq[(size_t)row * ldt + p * r + jr]);
            const vecf av = reinterpret_cast<const vecf*>(lora + (size_t)p * 2 * r * D + (size_t)jr * D)[lane + 64 * i];
            acc += t * av;
          }
#pragma unroll
          for (int e = 0; e < VW; ++e) d[e] += acc[e] * drop_scale(drop, p, idx + (unsigned)e);
        }
      }
    }
    g[i] = d * g2[lane + 64 * i];
#pragma unroll
    for (int e = 0; e < VW; ++e) s += v[i][e];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    v[i] -= mean;
#pragma unroll
    for (int e = 0; e < VW; ++e) q += v[i][e] * v[i][e];
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    v[i] *= rstd;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      sg += g[i][e];
      sgx += g[i][e] * v[i][e];
    }
  }
  const float mg = wave_sum(sg) / (float)D, mgx = wave_sum(sgx) / (float)D;
  const vecf* sc2 = scale ? reinterpret_cast<const vecf*>(scale) : nullptr;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    vecf o;
#pragma unroll
    for (int e = 0; e < VW; ++e) o[e] = rstd * (g[i][e] - mg - v[i][e] * mgx) + res[i][e];
    if (dx) reinterpret_cast<vecf*>(dx + (size_t)row * D)[lane + 64 * i] = o;
    if (sout) {
      const vecf c = sc2 ? sc2[lane + 64 * i] : (vecf)(1.f);
      typedef unsigned vecu __attribute__((ext_vector_type(VW / 2)));
      vecu w;
#pragma unroll
      for (int e = 0; e < VW / 2; ++e) w[e] = pack_bf16x2(o[2 * e] * c[2 * e], o[2 * e + 1] * c[2 * e + 1]);
      reinterpret_cast<vecu*>(sout + (size_t)row * D)[lane + 64 * i] = w;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Cotangent of the key hook, [B, D, h*w] f32 (CLS dropped, NCHW) -> token-major rows of dqkv_aug: the k third gets the
// transposed values (CLS row 0), the q and v thirds are zero (the last layer's q / v never reach the loss).
// Block = (image, 32 tokens); 32x32 LDS transposes over the channel axis.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void key_grad_tokens_kernel(const float* __restrict__ dkey, bf16_raw* __restrict__ dqkv, int tok,
                                                              int D) {
  __shared__ float tile[32][33];
  const int b = blockIdx.y, t0 = blockIdx.x * 32;
  const int hw = tok - 1, ld = 3 * D + AUG;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
  bf16_raw* base = dqkv + (size_t)b * tok * ld;
  // zero q, v (and aug) of these rows
  for (int rr = 0; rr < 32; ++rr) {
    const int t = t0 + rr;
    if (t >= tok) break;
    bf16_raw* rowp = base + (size_t)t * ld;
    for (int c = threadIdx.x; c < D; c += 256) { rowp[c] = 0; rowp[2 * D + c] = 0; }
    if (threadIdx.x < AUG) rowp[3 * D + threadIdx.x] = 0;
  }
  for (int c0 = 0; c0 < D; c0 += 32) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + ty + 8 * k, t = t0 + tx;                 // read: tokens contiguous
      float v = 0.f;
      if (t >= 1 && t < tok) v = dkey[((size_t)b * D + c) * hw + (t - 1)];
      tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = t0 + ty + 8 * k, c = c0 + tx;                 // write: channels contiguous
      if (t < tok) base[(size_t)t * ld + D + c] = f32_to_bf16(tile[tx][ty + 8 * k]);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// LoRA pack: aug columns of Wqkv_aug [3D, D+64] and WqkvT_aug [D, 3D+64] from one layer's LoRA parameters.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lora_pack_kernel(const float* __restrict__ lora, int r, float scaling, bf16_raw* __restrict__ w_aug,
                                                        bf16_raw* __restrict__ wt_aug, int D, int zero_at) {
  const int idx = blockIdx.x * 256 + threadIdx.x;                // one thread per (row, aug column) of either matrix
  const int n_fwd = 3 * D * AUG;
  if (idx < n_fwd) {
    if (!w_aug) return;
    const int n = idx / AUG, j = idx - n * AUG;                  // row n of Wqkv (which of q/k/v: p), aug column j
    const int p = n / D, nn = n - p * D;
    float v = 0.f;
    if (j >= p * r && j < (p + 1) * r) v = scaling * lora[(size_t)p * 2 * r * D + (size_t)r * D + (size_t)nn * r + (j - p * r)];   // B_p[nn][j']
    w_aug[(size_t)n * (D + AUG) + D + j] = f32_to_bf16(v);
    return;
  }
  const int k = idx - n_fwd;
  if (k >= D * AUG || !wt_aug) return;
  const int d = k / AUG, j = k - d * AUG;
  float v = 0.f;
  if (j < 3 * r && !zero_at) {                                   // zero_at: dropout on -- the (masked) A term is added by ln_bwd instead
    const int p = j / r, jj = j - p * r;
    v = lora[(size_t)p * 2 * r * D + (size_t)jj * D + d];        // A_p[jj][d]
  }
  wt_aug[(size_t)d * (3 * D + AUG) + 3 * D + j] = f32_to_bf16(v);
}

// ---------------------------------------------------------------------------------------------------------------------
// LoRA gradients of one layer, ranks [j0, j0+RW) of each of q, k, v in one pass over dqkv and LN1(x):
//   t[m][p][j]  = scaling * sum_n dqkv[m][pD+n] * B_p[n][j]          -> aug columns of dqkv_aug (bf16)
//   dA_p[j][d] += t[m][p][j] * h[m][d]
//   dB_p[n][j] += scaling * dqkv[m][pD+n] * u[m][p][j]                (u = aug columns of h_aug)
// Block = 6 waves = 2 row streams x 3 projections: wave (p, s) walks rows s, s+2*gridDim, ... and touches only the p-th
// third of dqkv, so its state is 2*RW*NCH*2 accumulators + the lane's B values (~110 VGPRs: 4 waves per SIMD, where the
// first version -- one wave for all three thirds, 332 VGPRs, one wave per SIMD -- sat on the load latency of every row).
// The next row's loads are issued before the current row's arithmetic.  Lane owns columns {2*lane, 2*lane+1} + 128*i of
// its D-wide slice.  Per-block partials, summed in a fixed order by lora_grad_reduce_kernel (deterministic).
// ---------------------------------------------------------------------------------------------------------------------
// VW = columns per lane and chunk: 2 (4-byte loads) or, for D % 256 == 0, 4 (8-byte loads: half the load instructions per row -- round 4)
// NS = row streams per block (block = 3 NS waves): the per-block combine and the partials the reduce kernel walks are paid once per NS streams
template <int NCH, int RW, int VW, int NS>
__global__ __launch_bounds__(192 * NS) void lora_grad_kernel(bf16_raw* __restrict__ dqkv, const bf16_raw* __restrict__ h,
                                                        const float* __restrict__ lora, int r, int j0, float scaling,
                                                        float* __restrict__ partial, int rows, int D, Drop drop) {
  extern __shared__ float lds[];                                  // block reduction buffer [3][RW][D] | [3][D][RW]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = wave % 3, strm = wave / 3;
  const int ldq = 3 * D + AUG, ldh = D + AUG;
  // this lane's B_p[n][j0 + j] (n = its 2*NCH columns)
  float bw[NCH][VW][RW];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < VW; ++e)
#pragma unroll
      for (int j = 0; j < RW; ++j) {
        const int n = VW * (lane + 64 * i) + e;
        bw[i][e][j] = (j0 + j < r) ? lora[(size_t)p * 2 * r * D + (size_t)r * D + (size_t)n * r + j0 + j] : 0.f;
      }
  float accA[RW][NCH][VW], accB[NCH][VW][RW];
#pragma unroll
  for (int j = 0; j < RW; ++j)
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < VW; ++e) accA[j][i][e] = accB[i][e][j] = 0.f;
  const int step = gridDim.x * NS;
  int row = blockIdx.x * NS + strm;
  typedef unsigned uvw __attribute__((ext_vector_type(VW / 2)));
  // PF rows of this wave in flight beside the one being reduced.  Measured (round 4, ViT-B, 32 images): PF = 1 -> 153-158 us per launch, PF = 4 -> 161-169
  // (185 VGPRs: two waves per SIMD), 4-byte or 8-byte loads alike: the kernel is bound neither by load latency nor by load width.
  constexpr int PF = 1;
  uvw dw[PF][NCH], hw_[PF][NCH];
  unsigned uw[PF];
  auto issue = [&](int k, int rw) {
    const uvw* dq = reinterpret_cast<const uvw*>(dqkv + (size_t)rw * ldq + (size_t)p * D);
    const uvw* hr = reinterpret_cast<const uvw*>(h + (size_t)rw * ldh);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      dw[k][i] = dq[lane + 64 * i];
      hw_[k][i] = hr[lane + 64 * i];
    }
    // u[p][j0 .. j0+RW): bf16 pair read as one dword when aligned, else two halves (wave-uniform address: broadcast)
    const bf16_raw* up = h + (size_t)rw * ldh + D + p * r + j0;
    const unsigned lo = up[0], hi = (RW > 1 && j0 + 1 < r) ? up[1] : 0u;
    uw[k] = lo | (hi << 16);
  };
#pragma unroll
  for (int k = 0; k < PF; ++k)
    if (row + k * step < rows) issue(k, row + k * step);
  while (row < rows) {
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      if (row >= rows) break;
      float dv[NCH][VW], hv[NCH][VW], u[RW], t[RW];
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int w = 0; w < VW / 2; ++w) {
          dv[i][2 * w] = __uint_as_float(dw[k][i][w] << 16);
          dv[i][2 * w + 1] = __uint_as_float(dw[k][i][w] & 0xFFFF0000u);
          hv[i][2 * w] = __uint_as_float(hw_[k][i][w] << 16);
          hv[i][2 * w + 1] = __uint_as_float(hw_[k][i][w] & 0xFFFF0000u);
        }
      u[0] = __uint_as_float(uw[k] << 16);
      if (RW > 1) u[1] = __uint_as_float(uw[k] & 0xFFFF0000u);
      const int cur = row;
      if (drop.thresh) {                                             // dA sees the SAME dropped input the forward's lora_A saw
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const unsigned idx = (unsigned)cur * (unsigned)D + (unsigned)VW * (unsigned)(lane + 64 * i);
#pragma unroll
          for (int e = 0; e < VW; ++e) hv[i][e] *= drop_scale(drop, p, idx + (unsigned)e);
        }
      }
      if (row + PF * step < rows) issue(k, row + PF * step);         // this slot's next row, PF rows ahead
      row += step;
#pragma unroll
      for (int j = 0; j < RW; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            s += dv[i][e] * bw[i][e][j];
            accB[i][e][j] += dv[i][e] * u[j];
          }
        t[j] = bf16_to_f32(f32_to_bf16(scaling * wave_sum(s)));      // the dgrad GEMM sees the bf16 value: use it for dA too
        if (lane == 0 && j0 + j < r) dqkv[(size_t)cur * ldq + 3 * D + p * r + j0 + j] = f32_to_bf16(t[j]);
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
          for (int e = 0; e < VW; ++e) accA[j][i][e] += t[j] * hv[i][e];
      }
    }
  }
  // block combine: [3][RW][D] (dA window) then [3][D][RW] (dB window); stream 0 writes, stream 1 adds
  const int nA = 3 * RW * D, nB = 3 * D * RW;
  float* out = partial + (size_t)blockIdx.x * (nA + nB);
  for (int s2 = 0; s2 < NS; ++s2) {
    if (strm == s2) {
#pragma unroll
      for (int j = 0; j < RW; ++j)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            const int d = VW * (lane + 64 * i) + e;
            const int ia = (p * RW + j) * D + d, ib = nA + (p * D + d) * RW + j;
            if (s2 == 0) {
              lds[ia] = accA[j][i][e];
              lds[ib] = scaling * accB[i][e][j];
            } else {
              lds[ia] += accA[j][i][e];
              lds[ib] += scaling * accB[i][e][j];
            }
          }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nA + nB; i += 192 * NS) out[i] = lds[i];
}

// grad layout = parameter layout of the layer: [A_q | B_q | A_k | B_k | A_v | B_v]; accumulate = add to existing content
template <int RW>
__global__ __launch_bounds__(256) void lora_grad_reduce_kernel(const float* __restrict__ partial, int nblk, int r, int j0, float* __restrict__ grad,
                                                               int D, int accumulate) {
  // 32 elements x 8 slices of the block list per workgroup (a single thread walking all 512 partials was latency-bound)
  __shared__ float red[8][33];
  const int nA = 3 * RW * D, nB = 3 * D * RW;
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;
  float s = 0.f;
  if (i < nA + nB) {
#pragma unroll 4
    for (int b = sl; b < nblk; b += 8) s += partial[(size_t)b * (nA + nB) + i];
  }
  red[sl][e] = s;
  __syncthreads();
  if (sl != 0 || i >= nA + nB) return;
  s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += red[k][e];
  int p, j, d;
  size_t dst;
  if (i < nA) {
    p = i / (RW * D);
    j = (i - p * RW * D) / D;
    d = i - p * RW * D - j * D;
    if (j0 + j >= r) return;
    dst = (size_t)p * 2 * r * D + (size_t)(j0 + j) * D + d;
  } else {
    const int k = i - nA;
    p = k / (D * RW);
    d = (k - p * D * RW) / RW;
    j = k - p * D * RW - d * RW;
    if (j0 + j >= r) return;
    dst = (size_t)p * 2 * r * D + (size_t)r * D + (size_t)d * r + j0 + j;
  }
  grad[dst] = accumulate ? grad[dst] + s : s;
}

constexpr int LORA_GRAD_BLOCKS = 512;                            // (1024 measured: 153 -> 183 us per launch: more partials to combine, no more rows in flight per SIMD)

}  // namespace ucod

using namespace ucod;

static int launch_ln_lora(const void* x, bool x_h16, const float* gamma, const float* beta, const float* lora, int r, void* y_aug, int rows, int D, float eps,
                          const ucod_lora_dropout* dropout, void* stream) {
  if (!x || !gamma || !beta || !lora || !y_aug || rows <= 0 || D <= 0 || (D % 128) != 0 || r < 1 || 3 * r > AUG) return UCOD_EINVAL;
  if (dropout && (dropout->p < 0.f || dropout->p >= 1.f)) return UCOD_EINVAL;
  UCOD_PROF(PROF_LN, stream);
  const Drop drop = make_drop(dropout);
  static const int lnl_env = [] { const char* e = ucod::lab_env("UCOD_LN_LORA_NBLK"); return e ? atoi(e) : 0; }();          // measurement knob
  const int lnl_max = lnl_env > 0 ? lnl_env : 2048;
  const int nblk = cdiv(rows, 8) < lnl_max ? cdiv(rows, 8) : lnl_max;   // block-stride over rows: the A rows are staged once per block
  dim3 grid(nblk), block(256);
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)3 * r * D * sizeof(float);
  static const bool narrow = ucod::lab_env("UCOD_LN_LORA_NARROW") != nullptr;     // measurement knob: the 8-byte / 4-byte form
  if ((D % 256) == 0 && D <= 1536 && !narrow) {
    switch (D / 256) {
#define C4(n)                                                                                                                                        \
  case n:                                                                                                                                            \
    if (x_h16) hipLaunchKernelGGL((ln_lora4_kernel<n, true>), grid, block, lds, s, x, gamma, beta, lora, r, (bf16_raw*)y_aug, rows, D, eps, drop);     \
    else hipLaunchKernelGGL((ln_lora4_kernel<n, false>), grid, block, lds, s, x, gamma, beta, lora, r, (bf16_raw*)y_aug, rows, D, eps, drop);          \
    break;
      C4(1) C4(2) C4(3) C4(4) C4(5) C4(6)
#undef C4
    }
    UCOD_CHECK_LAUNCH();
    return UCOD_OK;
  }
  switch (D / 128) {
#define C(n)                                                                                                                                         \
  case n:                                                                                                                                            \
    if (x_h16) hipLaunchKernelGGL((ln_lora_kernel<n, true>), grid, block, lds, s, x, gamma, beta, lora, r, (bf16_raw*)y_aug, rows, D, eps, drop);      \
    else hipLaunchKernelGGL((ln_lora_kernel<n, false>), grid, block, lds, s, x, gamma, beta, lora, r, (bf16_raw*)y_aug, rows, D, eps, drop);           \
    break;
    C(1) C(2) C(3) C(4) C(5) C(6) C(8) C(10) C(12)
#undef C
    default: return UCOD_EINVAL;
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_layernorm_lora(const float* x, const float* gamma, const float* beta, const float* lora, int r, void* y_aug, int rows,
                                   int D, float eps, const ucod_lora_dropout* dropout, void* stream) {
  UCOD_BF16_ONLY();
  return launch_ln_lora(x, false, gamma, beta, lora, r, y_aug, rows, D, eps, dropout, stream);
}

extern "C" int ucod_layernorm_lora_h16(const void* x_f16, const float* gamma, const float* beta, const float* lora, int r, void* y_aug, int rows,
                                       int D, float eps, const ucod_lora_dropout* dropout, void* stream) {
  UCOD_BF16_ONLY();
  return launch_ln_lora(x_f16, true, gamma, beta, lora, r, y_aug, rows, D, eps, dropout, stream);
}

static int launch_ln_bwd(const void* dy, bool dy_bf16, const void* x, bool x_f16, const float* gamma, const float* dres, const float* next_scale, float* dx, void* s_bf16,
                         int rows, int D, float eps, const bf16_raw* tq, int ldt, const float* lora, int r, const Drop& drop, hipStream_t s) {
  dim3 grid(cdiv(rows, 4)), block(256);
  const bool lo = tq != nullptr;
  if (x_f16 && !dy_bf16) return UCOD_EINVAL;                // (the fp16 stream comes with bf16 dgrad outputs: the driver's only use)
#define L(n, vw, lora_, dyb_, xh_) hipLaunchKernelGGL((ln_bwd_kernel<n, vw, lora_, dyb_, xh_>), grid, block, 0, s, dy, x, gamma, dres, next_scale, dx, (bf16_raw*)s_bf16, rows, D, eps, tq, ldt, lora, r, drop)
#define C(n, vw)                                                                                                                               \
  case n:                                                                                                                                      \
    if (lo && x_f16) L(n, vw, true, true, true);                                                                                               \
    else if (lo && dy_bf16) L(n, vw, true, true, false);                                                                                       \
    else if (lo) L(n, vw, true, false, false);                                                                                                 \
    else if (x_f16) L(n, vw, false, true, true);                                                                                               \
    else if (dy_bf16) L(n, vw, false, true, false);                                                                                            \
    else L(n, vw, false, false, false);                                                                                                        \
    break;
  if (D % 256 == 0) {                                      // 16-byte accesses
    switch (D / 256) {
      C(1, 4) C(2, 4) C(3, 4) C(4, 4) C(5, 4) C(6, 4)
      default: return UCOD_EINVAL;
    }
  } else {
    switch (D / 128) {
      C(1, 2) C(3, 2) C(5, 2) C(7, 2) C(9, 2) C(11, 2)
      default: return UCOD_EINVAL;
    }
  }
#undef C
#undef L
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* dres, const float* next_scale, float* dx,
                                  void* s_bf16, int rows, int D, float eps, void* stream) {
  UCOD_BF16_ONLY();
  if (!dy || !x || !gamma || (!dx && !s_bf16) || rows <= 0 || D <= 0 || (D % 128) != 0) return UCOD_EINVAL;
  UCOD_PROF(PROF_LN_BWD, stream);
  return launch_ln_bwd(dy, false, x, false, gamma, dres, next_scale, dx, s_bf16, rows, D, eps, nullptr, 0, nullptr, 0, make_drop(nullptr), (hipStream_t)stream);
}

extern "C" int ucod_layernorm_bwd_ex(const void* dy, const void* x, int flags, const float* gamma, const float* dres, const float* next_scale, float* dx,
                                     void* s_bf16, int rows, int D, float eps, void* stream) {
  UCOD_BF16_ONLY();
  if (!dy || !x || !gamma || (!dx && !s_bf16) || rows <= 0 || D <= 0 || (D % 128) != 0 || (flags & ~3)) return UCOD_EINVAL;
  UCOD_PROF(PROF_LN_BWD, stream);
  return launch_ln_bwd(dy, (flags & UCOD_LNB_DY_BF16) != 0, x, (flags & UCOD_LNB_X_F16) != 0, gamma, dres, next_scale, dx, s_bf16, rows, D, eps, nullptr, 0, nullptr, 0,
                       make_drop(nullptr), (hipStream_t)stream);
}

extern "C" int ucod_layernorm_bwd_lora(const float* dy, const float* x, const float* gamma, const float* dres, const float* next_scale, float* dx,
                                       void* s_bf16, int rows, int D, float eps, const void* dqkv_aug, const float* lora_layer, int r,
                                       const ucod_lora_dropout* dropout, void* stream) {
  UCOD_BF16_ONLY();
  if (!dy || !x || !gamma || (!dx && !s_bf16) || rows <= 0 || D <= 0 || (D % 128) != 0 || !dqkv_aug || !lora_layer || r < 1 || 3 * r > AUG ||
      !dropout || dropout->p < 0.f || dropout->p >= 1.f)
    return UCOD_EINVAL;
  UCOD_PROF(PROF_LN_BWD, stream);
  return launch_ln_bwd(dy, false, x, false, gamma, dres, next_scale, dx, s_bf16, rows, D, eps, (const bf16_raw*)dqkv_aug + 3 * D, 3 * D + AUG, lora_layer, r,
                       make_drop(dropout), (hipStream_t)stream);
}

extern "C" int ucod_layernorm_bwd_lora_ex(const void* dy, const void* x, int flags, const float* gamma, const float* dres, const float* next_scale, float* dx,
                                          void* s_bf16, int rows, int D, float eps, const void* dqkv_aug, const float* lora_layer, int r,
                                          const ucod_lora_dropout* dropout, void* stream) {
  UCOD_BF16_ONLY();
  if (!dy || !x || !gamma || (!dx && !s_bf16) || rows <= 0 || D <= 0 || (D % 128) != 0 || !dqkv_aug || !lora_layer || r < 1 || 3 * r > AUG ||
      !dropout || dropout->p < 0.f || dropout->p >= 1.f || (flags & ~3))
    return UCOD_EINVAL;
  UCOD_PROF(PROF_LN_BWD, stream);
  return launch_ln_bwd(dy, (flags & UCOD_LNB_DY_BF16) != 0, x, (flags & UCOD_LNB_X_F16) != 0, gamma, dres, next_scale, dx, s_bf16, rows, D, eps,
                       (const bf16_raw*)dqkv_aug + 3 * D, 3 * D + AUG, lora_layer, r, make_drop(dropout), (hipStream_t)stream);
}

extern "C" int ucod_key_grad_tokens(const float* dkey, void* dqkv_aug, int B, int tok, int D, void* stream) {
  UCOD_BF16_ONLY();
  if (!dkey || !dqkv_aug || B <= 0 || tok < 2 || D <= 0 || (D % 32) != 0) return UCOD_EINVAL;
  UCOD_PROF(PROF_LORA, stream);
  hipLaunchKernelGGL(key_grad_tokens_kernel, dim3(cdiv(tok, 32), B), dim3(256), 0, (hipStream_t)stream, dkey, (bf16_raw*)dqkv_aug, tok, D);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_lora_pack(const float* lora_layer, int r, float scaling, void* w_aug, void* wt_aug, int D, int zero_a_columns, void* stream) {
  UCOD_BF16_ONLY();
  if (!lora_layer || (!w_aug && !wt_aug) || r < 1 || 3 * r > AUG || D <= 0) return UCOD_EINVAL;
  UCOD_PROF(PROF_LORA, stream);
  const int n = 3 * D * AUG + D * AUG;
  hipLaunchKernelGGL(lora_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, lora_layer, r, scaling, (bf16_raw*)w_aug,
                     (bf16_raw*)wt_aug, D, zero_a_columns);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" size_t ucod_lora_grad_workspace_bytes(int D) { return (size_t)LORA_GRAD_BLOCKS * (size_t)(12 * D) * sizeof(float); }

extern "C" int ucod_lora_grad(void* dqkv_aug, const void* h_aug, const float* lora_layer, int r, float scaling, float* grad_layer,
                              int accumulate, void* workspace, size_t workspace_bytes, int rows, int D, const ucod_lora_dropout* dropout,
                              void* stream) {
  UCOD_BF16_ONLY();
  if (!dqkv_aug || !h_aug || !lora_layer || !grad_layer || !workspace || r < 1 || 3 * r > AUG || rows <= 0 || D <= 0 || (D % 128) != 0)
    return UCOD_EINVAL;
  if (dropout && (dropout->p < 0.f || dropout->p >= 1.f)) return UCOD_EINVAL;
  if (workspace_bytes < ucod_lora_grad_workspace_bytes(D)) return UCOD_ENOMEM;
  UCOD_PROF(PROF_LORA, stream);
  const Drop drop = make_drop(dropout);
  hipStream_t s = (hipStream_t)stream;
  constexpr int RW = 2;                                           // ranks per pass (the reference's r = 2 is one pass)
  static const int ns_env = [] { const char* e = ucod::lab_env("UCOD_LORA_GRAD_STREAMS"); return e ? atoi(e) : 0; }();   // measurement knob: 2 or 4 row streams per block
  const int NSr = ns_env == 2 ? 2 : ns_env == 5 ? 5 : 4;
  static const int nb_env = [] { const char* e = ucod::lab_env("UCOD_LORA_GRAD_NBLK"); return e ? atoi(e) : 0; }();       // measurement knob (<= LORA_GRAD_BLOCKS)
  // default: one block per CU (12 waves of 4 row streams x 3 projections), a whole round -- round 4: 512 blocks of 6 waves 158 us -> 109 us per launch at
  // ViT-B / 32 images (fewer partials to combine and to reduce; 384 blocks = 1.5 rounds: 140 us)
  static const int n_cu = [] { hipDeviceProp_t pr; int dv = 0; (void)hipGetDevice(&dv); return hipGetDeviceProperties(&pr, dv) == hipSuccess ? pr.multiProcessorCount : 256; }();
  const int nb_max = nb_env > 0 && nb_env <= LORA_GRAD_BLOCKS ? nb_env : (n_cu < LORA_GRAD_BLOCKS ? n_cu : LORA_GRAD_BLOCKS);
  const int nblk = rows < nb_max * NSr ? cdiv(rows, NSr) : nb_max;
  const size_t lds_bytes = (size_t)6 * D * RW * sizeof(float);      // B window (3*D*RW) <= block reduction buffer (6*D*RW)
  for (int j0 = 0; j0 < r; j0 += RW) {
    static const bool narrow = ucod::lab_env("UCOD_LORA_GRAD_WIDE") == nullptr;      // default: the 4-byte-load form (88 VGPRs; the 8-byte one: 148, 4 % slower)
#define LG(n, vw)                                                                                                                                   \
  do {                                                                                                                                              \
    if (NSr == 2) hipLaunchKernelGGL((lora_grad_kernel<n, RW, vw, 2>), dim3(nblk), dim3(384), lds_bytes, s, (bf16_raw*)dqkv_aug, (const bf16_raw*)h_aug, lora_layer, r, j0, scaling, (float*)workspace, rows, D, drop); \
    else if (NSr == 5) hipLaunchKernelGGL((lora_grad_kernel<n, RW, vw, 5>), dim3(nblk), dim3(960), lds_bytes, s, (bf16_raw*)dqkv_aug, (const bf16_raw*)h_aug, lora_layer, r, j0, scaling, (float*)workspace, rows, D, drop); \
    else hipLaunchKernelGGL((lora_grad_kernel<n, RW, vw, 4>), dim3(nblk), dim3(768), lds_bytes, s, (bf16_raw*)dqkv_aug, (const bf16_raw*)h_aug, lora_layer, r, j0, scaling, (float*)workspace, rows, D, drop);     \
  } while (0)
    if ((D % 256) == 0 && D <= 1024 && !narrow) {
      switch (D / 256) {
        case 1: LG(1, 4); break;
        case 2: LG(2, 4); break;
        case 3: LG(3, 4); break;
        case 4: LG(4, 4); break;
      }
    } else {
      switch (D / 128) {
#define C(n) case n: LG(n, 2); break;
        C(1) C(2) C(3) C(4) C(5) C(6) C(8)
#undef C
        default: return UCOD_EINVAL;
      }
    }
#undef LG
    UCOD_CHECK_LAUNCH();
    hipLaunchKernelGGL((lora_grad_reduce_kernel<RW>), dim3(cdiv(6 * RW * D, 32)), dim3(256), 0, s, (const float*)workspace, nblk, r, j0,
                       grad_layer, D, accumulate);
    UCOD_CHECK_LAUNCH();
  }
  return UCOD_OK;
}
