// Memory-bound helpers of the ViT backbone: LayerNorm, patch gather (im2col), CLS rows, casts.
// All are HBM-bound streaming kernels: one wave per row with wave-shuffle reductions (LayerNorm),
// vector loads/stores, no LDS.  SURVEY.md 8a rows B1,B2,B3.
#include <cstdlib>
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

// One wave per row, row resident in registers (NCH float2 chunks per lane, D = 128*NCH), two-pass
// mean / variance like ATen's LayerNorm (modeling_dinov2.py:348,353; eps from the checkpoint config).
template <int NCH, bool OUT_F32>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ y, int rows,
                                                        int D, float eps) {
  // R rows per wave, all their loads issued before the first reduction: twice the bytes in flight per wave (the kernel is
  // pure HBM streaming; one row per wave left the memory pipe waiting on the two shuffle reductions of every row)
  constexpr int R = 2;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  float2 v[R][NCH];
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = (row0 + r) < rows ? (row0 + r) : rows - 1;
    const float2* xr = reinterpret_cast<const float2*>(x + (size_t)row * D);
    s[r] = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[r][i] = xr[lane + 64 * i];
  }
  const float2* g2 = reinterpret_cast<const float2*>(gamma);
  const float2* b2 = reinterpret_cast<const float2*>(beta);
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) s[r] += v[r][i].x + v[r][i].y;
    const float mean = wave_sum(s[r]) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const float a = v[r][i].x - mean, b = v[r][i].y - mean;
      q += a * a + b * b;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    const int row = row0 + r;
    if (row >= rows) break;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const float2 g = g2[lane + 64 * i], b = b2[lane + 64 * i];
      const float o0 = (v[r][i].x - mean) * rstd * g.x + b.x;
      const float o1 = (v[r][i].y - mean) * rstd * g.y + b.y;
      if constexpr (OUT_F32) {
        reinterpret_cast<float2*>(reinterpret_cast<float*>(y) + (size_t)row * D)[lane + 64 * i] = make_float2(o0, o1);
      } else {
        reinterpret_cast<unsigned*>(reinterpret_cast<bf16_raw*>(y) + (size_t)row * D)[lane + 64 * i] = pack_h2(o0, o1);
      }
    }
  }
}

// float4 form for D % 256 == 0 (ViT-B 768, ViT-L 1024): half the load / store instructions of the float2 form at equal bytes.
template <int NV, bool OUT_F32>
__global__ __launch_bounds__(256) void layernorm4_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, void* __restrict__ y, int rows,
                                                         int D, float eps) {
  constexpr int R = 2;
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  float4 v[R][NV];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = (row0 + r) < rows ? (row0 + r) : rows - 1;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[r][i] = xr[lane + 64 * i];
  }
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float a = v[r][i].x - mean, b = v[r][i].y - mean, c = v[r][i].z - mean, d = v[r][i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    const int row = row0 + r;
    if (row >= rows) break;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 g = g4[lane + 64 * i], b = b4[lane + 64 * i];
      const float o0 = (v[r][i].x - mean) * rstd * g.x + b.x, o1 = (v[r][i].y - mean) * rstd * g.y + b.y;
      const float o2 = (v[r][i].z - mean) * rstd * g.z + b.z, o3 = (v[r][i].w - mean) * rstd * g.w + b.w;
      if constexpr (OUT_F32) {
        reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + (size_t)row * D)[lane + 64 * i] = make_float4(o0, o1, o2, o3);
      } else {
        u32x2 w;
        w[0] = pack_h2(o0, o1);
        w[1] = pack_h2(o2, o3);
        reinterpret_cast<u32x2*>(reinterpret_cast<bf16_raw*>(y) + (size_t)row * D)[lane + 64 * i] = w;
      }
    }
  }
}

// 16-byte form of the fp16-stream LayerNorm (D % 256 == 0): a wave owns a STRIP of two consecutive rows = D / 4 chunks of 16 bytes that are
// contiguous in memory on both sides (8 f16 in, 8 operand halves out), lane l takes chunks l, l + 64, ... of the strip, so a lane may
// hold pieces of both rows and keeps one set of statistics per row.  Why: the 8-byte stores of the form below run at the per-CU
// store-issue rate (~7 B/clk/CU, MI355X_MICROARCH.md), 263 KB of output per CU = 22 of the launch's 26 us; 16-byte stores issue 2.4x
// the bytes per clock.
template <int NC>                                          // NC = D / 256 chunks per lane
__global__ __launch_bounds__(256) void layernorm_h16_strip_kernel(const u32x4* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, u32x4* __restrict__ y, int rows, int D, float eps) {
  // A wave walks the strips  w, w + (waves of the grid), ...  with gamma / beta held in registers: per strip they are TWICE the bytes of the
  // strip's own fp16 rows (2 x D x 4 B of f32 against 2 x D x 2 B), all of it L2 -> CU traffic when every strip re-reads them (round 4: the
  // launcher sizes the grid so that a wave takes 2 - 4 strips; measured in profiles/r04_layernorm_strips.txt).
  const int lane = threadIdx.x & 63;
  const int cpr = D >> 3;                                  // chunks per row
  bool second[NC];
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float4 ga[NC], gb[NC], ba[NC], bb[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + 64 * i;
    second[i] = c >= cpr;
    const int col = second[i] ? c - cpr : c;
    ga[i] = g4[2 * col];
    gb[i] = g4[2 * col + 1];
    ba[i] = b4[2 * col];
    bb[i] = b4[2 * col + 1];
  }
  const int nstrips = (rows + 1) >> 1, stride = gridDim.x * 4;
  for (int strip = blockIdx.x * 4 + (threadIdx.x >> 6); strip < nstrips; strip += stride) {
    const int row0 = strip * 2;
    const bool two = row0 + 1 < rows;
    const u32x4* xs = x + (size_t)row0 * cpr;
    float v[NC][8];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      const u32x4 w = xs[(second[i] && !two) ? c - cpr : c];  // an odd last row: re-read row 0 (never used)
#pragma unroll
      for (int e = 0; e < 4; ++e) unpack_f16x2(w[e], v[i][2 * e], v[i][2 * e + 1]);
    }
    // (LLVM sinks loads into the conditional block that uses them -- the odd-last-row skip below -- unless the values are opaque here)
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      asm volatile("" : "+v"(ga[i].x), "+v"(ga[i].y), "+v"(ga[i].z), "+v"(ga[i].w), "+v"(gb[i].x), "+v"(gb[i].y), "+v"(gb[i].z), "+v"(gb[i].w));
      asm volatile("" : "+v"(ba[i].x), "+v"(ba[i].y), "+v"(ba[i].z), "+v"(ba[i].w), "+v"(bb[i].x), "+v"(bb[i].y), "+v"(bb[i].z), "+v"(bb[i].w));
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const float t = ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
      s0 += second[i] ? 0.f : t;
      s1 += second[i] ? t : 0.f;
    }
    const float inv_d = 1.0f / (float)D;
    const float mean0 = wave_sum(s0) * inv_d, mean1 = wave_sum(s1) * inv_d;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const float m = second[i] ? mean1 : mean0;
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] -= m;
        t += v[i][e] * v[i][e];
      }
      q0 += second[i] ? 0.f : t;
      q1 += second[i] ? t : 0.f;
    }
    const float rstd0 = rsqrtf(wave_sum(q0) * inv_d + eps), rstd1 = rsqrtf(wave_sum(q1) * inv_d + eps);
    u32x4* ys = y + (size_t)row0 * cpr;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      if (second[i] && !two) continue;
      const float rstd = second[i] ? rstd1 : rstd0;
      u32x4 w;
      w[0] = pack_h2(v[i][0] * rstd * ga[i].x + ba[i].x, v[i][1] * rstd * ga[i].y + ba[i].y);
      w[1] = pack_h2(v[i][2] * rstd * ga[i].z + ba[i].z, v[i][3] * rstd * ga[i].w + ba[i].w);
      w[2] = pack_h2(v[i][4] * rstd * gb[i].x + bb[i].x, v[i][5] * rstd * gb[i].y + bb[i].y);
      w[3] = pack_h2(v[i][6] * rstd * gb[i].z + bb[i].z, v[i][7] * rstd * gb[i].w + bb[i].w);
      ys[c] = w;
    }
  }
}

// Row statistics of the fp16 residual stream for the LayerNorm-folded GEMM epilogues (ucod_gemm_lnfold): the strip walk of the kernel above
// without gamma / beta and without the output -- half the bytes of a LayerNorm launch.  stats[row] = (rstd, -mean * rstd); same two-pass f32
// arithmetic on the row held in registers.
template <int NC>
__global__ __launch_bounds__(256) void row_stats_h16_kernel(const u32x4* __restrict__ x, float2* __restrict__ stats, int rows, int D, float eps, unsigned* __restrict__ ovf) {
  const int lane = threadIdx.x & 63;
  const int cpr = D >> 3;
  bool second[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) second[i] = lane + 64 * i >= cpr;
  const int nstrips = (rows + 1) >> 1, stride = gridDim.x * 4;
  const float inv_d = 1.0f / (float)D;
  for (int strip = blockIdx.x * 4 + (threadIdx.x >> 6); strip < nstrips; strip += stride) {
    const int row0 = strip * 2;
    const bool two = row0 + 1 < rows;
    const u32x4* xs = x + (size_t)row0 * cpr;
    float v[NC][8];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = lane + 64 * i;
      const u32x4 w = xs[(second[i] && !two) ? c - cpr : c];
#pragma unroll
      for (int e = 0; e < 4; ++e) unpack_f16x2(w[e], v[i][2 * e], v[i][2 * e + 1]);
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const float t = ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])) + ((v[i][4] + v[i][5]) + (v[i][6] + v[i][7]));
      s0 += second[i] ? 0.f : t;
      s1 += second[i] ? t : 0.f;
    }
    const float mean0 = wave_sum(s0) * inv_d, mean1 = wave_sum(s1) * inv_d;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const float m = second[i] ? mean1 : mean0;
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - m;
        t += d * d;
      }
      q0 += second[i] ? 0.f : t;
      q1 += second[i] ? t : 0.f;
    }
    const float rstd0 = rsqrtf(wave_sum(q0) * inv_d + eps), rstd1 = rsqrtf(wave_sum(q1) * inv_d + eps);
    if (lane == 0) stats[row0] = make_float2(rstd0, -mean0 * rstd0);
    if (lane == 1 && two) stats[row0 + 1] = make_float2(rstd1, -mean1 * rstd1);
    // the fold's range guard (gemm_bf16_epilogue.h: fold_finish): a row with |mean| > 256 sigma is counted, mean^2 rstd^2 = mean^2 / (var + eps)
    if (ovf && ((lane == 0 && mean0 * mean0 * rstd0 * rstd0 > 65536.0f) || (lane == 1 && two && mean1 * mean1 * rstd1 * rstd1 > 65536.0f))) atomicAdd(ovf, 1u);
  }
}

// LayerNorm of an IEEE-fp16 residual stream (ucod_vit_desc.resid16): the row arrives as 8-byte (4 x f16, D % 256 == 0) or 4-byte
// (2 x f16) chunks per lane, is widened to f32 in registers, and the same two-pass f32 statistics follow; output = operand type.
template <int NV, int W, int R = 2>                       // NV chunks of W f16 per lane: D = 64 * NV * W; R rows per wave
__device__ __forceinline__ void layernorm_h16_body(const unsigned* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, bf16_raw* __restrict__ y, int rows, int D, float eps);

template <int NV, int W, int R = 2>
__global__ __launch_bounds__(256) void layernorm_h16_kernel(const unsigned* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_raw* __restrict__ y, int rows, int D, float eps) {
  layernorm_h16_body<NV, W, R>(x, gamma, beta, y, rows, D, eps);
}

template <int NV, int W, int R>
__device__ __forceinline__ void layernorm_h16_body(const unsigned* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_raw* __restrict__ y, int rows, int D, float eps) {
  constexpr int PW = W / 2;                                // packed dwords per chunk
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  float v[R][NV][W];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = (row0 + r) < rows ? (row0 + r) : rows - 1;
    const unsigned* xr = x + (size_t)row * (D / 2);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      unsigned w[PW];
      if constexpr (PW == 2) {
        const u32x2 t = reinterpret_cast<const u32x2*>(xr)[lane + 64 * i];
        w[0] = t[0];
        w[1] = t[1];
      } else {
        w[0] = xr[lane + 64 * i];
      }
#pragma unroll
      for (int e = 0; e < PW; ++e) unpack_f16x2(w[e], v[r][i][2 * e], v[r][i][2 * e + 1]);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < W; ++e) s += v[r][i][e];
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const float a = v[r][i][e] - mean;
        q += a * a;
      }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    const int row = row0 + r;
    if (row >= rows) break;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c0 = (lane + 64 * i) * W;
      unsigned o[PW];
#pragma unroll
      for (int e = 0; e < PW; ++e) {
        const float o0 = (v[r][i][2 * e] - mean) * rstd * gamma[c0 + 2 * e] + beta[c0 + 2 * e];
        const float o1 = (v[r][i][2 * e + 1] - mean) * rstd * gamma[c0 + 2 * e + 1] + beta[c0 + 2 * e + 1];
        o[e] = pack_h2(o0, o1);
      }
      unsigned* yr = reinterpret_cast<unsigned*>(y + (size_t)row * D);
      if constexpr (PW == 2) reinterpret_cast<u32x2*>(yr)[lane + 64 * i] = (u32x2){o[0], o[1]};
      else yr[lane + 64 * i] = o[0];
    }
  }
}

template <bool OUT_F32>
static int launch_ln(const float* x, const float* g, const float* b, void* y, int rows, int D, float eps, hipStream_t s) {
  dim3 grid(cdiv(rows, 8)), block(256);
  if ((D % 256) == 0 && D / 256 <= 6) {
    switch (D / 256) {
#define LN4_CASE(n) \
  case n: hipLaunchKernelGGL((layernorm4_kernel<n, OUT_F32>), grid, block, 0, s, x, g, b, y, rows, D, eps); break;
      LN4_CASE(1) LN4_CASE(2) LN4_CASE(3) LN4_CASE(4) LN4_CASE(5) LN4_CASE(6)
#undef LN4_CASE
    }
    UCOD_CHECK_LAUNCH();
    return UCOD_OK;
  }
  switch (D / 128) {
#define LN_CASE(n) \
  case n: hipLaunchKernelGGL((layernorm_kernel<n, OUT_F32>), grid, block, 0, s, x, g, b, y, rows, D, eps); break;
    LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(5) LN_CASE(6) LN_CASE(8) LN_CASE(10) LN_CASE(12)
#undef LN_CASE
    default: return UCOD_EINVAL;
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

// img [B,C,H,W] f32 -> patches bf16 [B*gh*gw, Kpad]; one thread per output pair (k, k+1).
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ img, bf16_raw* __restrict__ out, int B, int C,
                                                     int H, int W, int P, int Kpad, int gh, int gw) {
  const int kp = Kpad >> 1;
  const size_t total = (size_t)B * gh * gw * kp;
  const int K = C * P * P;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % kp) * 2;
    const size_t m = idx / kp;
    const int px = (int)(m % gw);
    const int py = (int)((m / gw) % gh);
    const int b = (int)(m / ((size_t)gw * gh));
    float v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = k + j;
      if (kk < K) {
        const int c = kk / (P * P), r = kk - c * P * P;
        const int dy = r / P, dx = r - dy * P;
        v[j] = img[(((size_t)b * C + c) * H + (py * P + dy)) * W + (px * P + dx)];
      } else {
        v[j] = 0.f;
      }
    }
    reinterpret_cast<unsigned*>(out)[idx] = pack_h2(v[0], v[1]);
  }
}

// Patch-row form of the gather: one workgroup owns one (image, patch row, channel), stages the P image rows it needs in LDS with
// coalesced loads and writes that channel's P*P-wide slice of every patch of the row as 8-byte (4 x bf16) stores; the zero padding
// past K is written by the last channel's workgroups.  The element-per-thread kernel above spends its time in 64-bit index divisions
// and 4-byte stores (2 TB/s on a 159 MB op).  Needs P*P % 4 == 0, Kpad % 4 == 0 and P*W floats of LDS.
__global__ __launch_bounds__(256) void im2col_rows_kernel(const float* __restrict__ img, bf16_raw* __restrict__ out, int C, int H, int W, int P,
                                                          int Kpad, int gh, int gw) {
  extern __shared__ __attribute__((aligned(16))) float rows[];           // [P][W] | lut [P*P]
  const int tid = threadIdx.x;
  const int c = blockIdx.x % C, py = (blockIdx.x / C) % gh, b = blockIdx.x / (C * gh);
  int* lut = reinterpret_cast<int*>(rows + P * W);                       // k within the channel -> LDS offset dy * W + dx
  const float* src = img + (((size_t)b * C + c) * H + (size_t)py * P) * W;
  for (int i = tid; i < P * W; i += 256) rows[i] = src[i];               // P consecutive image rows are contiguous
  const int PP = P * P;
  for (int k = tid; k < PP; k += 256) lut[k] = (k / P) * W + (k % P);
  __syncthreads();
  const int K = C * PP, chunks = PP >> 2;                                // 8-byte chunks of this channel's slice per patch
  bf16_raw* obase = out + ((size_t)(b * gh + py) * gw) * Kpad + (size_t)c * PP;
  for (int q = tid; q < gw * chunks; q += 256) {
    const int px = q / chunks, j = q - px * chunks;
    const float* pr = rows + px * P;
    u32x2 w;
    w[0] = pack_h2(pr[lut[4 * j]], pr[lut[4 * j + 1]]);
    w[1] = pack_h2(pr[lut[4 * j + 2]], pr[lut[4 * j + 3]]);
    *reinterpret_cast<u32x2*>(obase + (size_t)px * Kpad + 4 * j) = w;
  }
  if (c == C - 1) {                                                      // zero padding K .. Kpad-1 of every patch of the row
    const int pad4 = (Kpad - K) >> 2;
    bf16_raw* pbase = out + ((size_t)(b * gh + py) * gw) * Kpad + K;
    for (int q = tid; q < gw * pad4; q += 256) {
      const int px = q / pad4, j = q - px * pad4;
      *reinterpret_cast<u32x2*>(pbase + (size_t)px * Kpad + 4 * j) = (u32x2){0u, 0u};
    }
  }
}

__global__ void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos, int B,
                                int tok, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, j = i - b * D;
  x[(size_t)b * tok * D + j] = cls[j] + pos[j];
}

__global__ void cls_rows_h16_kernel(unsigned short* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos, int B,
                                    int tok, int D, unsigned* __restrict__ ovf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, j = i - b * D;
  const float o = cls[j] + pos[j];
  if (beyond_f16(o)) atomicAdd(ovf, 1u);
  x[(size_t)b * tok * D + j] = __builtin_bit_cast(unsigned short, (_Float16)clamp_f16(o));
}

// CLS rows of the fp16 stream WITH their row partials for the LayerNorm-folded consumer (gemm_bf16_epilogue.h: kStats): one workgroup per image.  A wave's 64
// lanes hold exactly one 64-column slot per iteration (columns wave * 64 + 256 i ..): slot (sum, M2 about the slot's own mean) of the rounded values, the
// format row_partial8 writes for the other rows (D % 64 == 0: the launcher checks).
__global__ __launch_bounds__(256) void cls_rows_h16_stats_kernel(unsigned short* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                                                 float2* __restrict__ part, int nslot, int tok, int D, unsigned* __restrict__ ovf) {
  const int b = blockIdx.x;
  bool sat = false;
  float2* row = part + (size_t)b * tok * nslot;
  for (int j = threadIdx.x; j < D; j += 256) {                    // (D % 64 == 0: whole waves)
    const float o = cls[j] + pos[j];
    sat |= beyond_f16(o);
    const _Float16 h = (_Float16)clamp_f16(o);
    x[(size_t)b * tok * D + j] = __builtin_bit_cast(unsigned short, h);
    const float r = (float)h;
    const float s = wave_sum(r);
    const float d = r - s * (1.0f / 64.0f);
    const float m2 = wave_sum(d * d);
    if ((threadIdx.x & 63) == 0) row[j >> 6] = make_float2(s, m2);
  }
  if (sat) atomicAdd(ovf, 1u);
}

// Saturation counter of the f16 residual stream: every kernel that rounds the stream to fp16 (patch / out-proj / fc2 epilogues, CLS rows)
// clamps to +-65504 and adds the number of wave-lanes that had to.  One word per device, polled by the host (ucod_resid16_overflow_*).
__device__ unsigned g_resid16_overflow = 0;

__global__ void cast_kernel(const float* __restrict__ s, bf16_raw* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = f32_to_h(s[i]);
}

// A caller that wants its OWN counter (one per engine instead of one per device) binds a device word for the launches it issues next on
// this host thread (ucod_resid16_overflow_bind); NULL restores the per-device word.
static thread_local unsigned* t_bound_counter = nullptr;

unsigned* resid16_overflow_counter() {
  if (t_bound_counter) return t_bound_counter;
  static unsigned* table[64] = {};                                   // one address per device of this process
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = dev < 0 || dev >= 64 ? 0 : dev;
  if (!table[dev]) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_resid16_overflow)) != hipSuccess) return nullptr;
    table[dev] = (unsigned*)p;
  }
  return table[dev];
}

}  // namespace ucod

extern "C" int ucod_resid16_overflow_bind(unsigned* device_counter) {
  ucod::t_bound_counter = device_counter;
  return UCOD_OK;
}

extern "C" int ucod_resid16_overflow_fetch(unsigned* host_dst, void* stream) {
  unsigned* c = ucod::resid16_overflow_counter();
  if (!host_dst || !c) return UCOD_EINVAL;
  return (int)hipMemcpyAsync(host_dst, c, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
}

extern "C" int ucod_resid16_overflow_reset(void* stream) {
  unsigned* c = ucod::resid16_overflow_counter();
  if (!c) return UCOD_EINVAL;
  return (int)hipMemsetAsync(c, 0, sizeof(unsigned), (hipStream_t)stream);
}

extern "C" int ucod_layernorm(const float* x, const float* gamma, const float* beta, void* y, int rows, int D, float eps,
                              int out_f32, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || D <= 0 || (D % 128) != 0) return UCOD_EINVAL;
  UCOD_PROF(ucod::PROF_LN, stream);
  return out_f32 ? ucod::launch_ln<true>(x, gamma, beta, y, rows, D, eps, (hipStream_t)stream)
                 : ucod::launch_ln<false>(x, gamma, beta, y, rows, D, eps, (hipStream_t)stream);
}

extern "C" int ucod_layernorm_h16(const void* x, const float* gamma, const float* beta, void* y, int rows, int D, float eps, void* stream) {
  using namespace ucod;
  if (!x || !gamma || !beta || !y || rows <= 0 || D <= 0 || (D % 128) != 0 || D > 1536) return UCOD_EINVAL;
  UCOD_PROF(PROF_LN, stream);
  dim3 grid(cdiv(rows, 8)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const unsigned* xp = (const unsigned*)x;
  bf16_raw* yp = (bf16_raw*)y;
  static const bool no_strip = ucod::lab_env("UCOD_LN_NO_STRIP") != nullptr;     // measurement knob: the 8-byte form
  // strips (row pairs) per wave of the 16-byte form: gamma / beta stay in registers over them.  UCOD_LN_STRIPS (read once): 0 = one strip per
  // wave (the round-3 launch); default 2 below D = 1024, 4 from there (profiles/r04_layernorm_strips.txt: 24.3 -> 23.1 us at 43840 x 768,
  // 18.6 -> 17.9 us = 0.63 of the HBM peak at 21920 x 1024, BASELINE configs[3]; 8 and more lose to the shorter tail)
  static const int strips_env = [] { const char* e = ucod::lab_env("UCOD_LN_STRIPS"); return e ? atoi(e) : -1; }();
  // (round 4, late) below D = 1024: 3, not the isolated launch's optimum of 2 -- in the pipelined step, where this kernel runs beside the other stream's
  // large-tile GEMM, 3 and 4 strips give 3 381-3 398 images/s against 3 356-3 374 with 2 (three alternating runs each, one box); 3 keeps the launch at
  // 26.0 us (0.65 of the HBM peak; 2: 25.3 us, 4: 27.0 us)
  const int strips_per_wave = strips_env >= 0 ? strips_env : (D >= 1024 ? 4 : 3);
  if ((D % 256) == 0 && !no_strip) {
    const int nstrips = (rows + 1) / 2;
    const dim3 sgrid(strips_per_wave > 1 ? (unsigned)(cdiv(cdiv(nstrips, strips_per_wave), 4) > 256 ? cdiv(cdiv(nstrips, strips_per_wave), 4) : 256) : (unsigned)cdiv(nstrips, 4));
    switch (D / 256) {
#define LNS_CASE(n) \
  case n: hipLaunchKernelGGL(layernorm_h16_strip_kernel<n>, sgrid, block, 0, s, (const u32x4*)x, gamma, beta, (u32x4*)y, rows, D, eps); break;
      LNS_CASE(1) LNS_CASE(2) LNS_CASE(3) LNS_CASE(4) LNS_CASE(5) LNS_CASE(6)
#undef LNS_CASE
      default: return UCOD_EINVAL;
    }
  } else if ((D % 256) == 0) {
    switch (D / 256) {
#define LNH_CASE(n) \
  case n: hipLaunchKernelGGL((layernorm_h16_kernel<n, 4>), grid, block, 0, s, xp, gamma, beta, yp, rows, D, eps); break;
      LNH_CASE(1) LNH_CASE(2) LNH_CASE(3) LNH_CASE(4) LNH_CASE(5) LNH_CASE(6)
#undef LNH_CASE
      default: return UCOD_EINVAL;
    }
  } else {
    switch (D / 128) {
#define LNH_CASE(n) \
  case n: hipLaunchKernelGGL((layernorm_h16_kernel<n, 2>), grid, block, 0, s, xp, gamma, beta, yp, rows, D, eps); break;
      LNH_CASE(1) LNH_CASE(3) LNH_CASE(5) LNH_CASE(7) LNH_CASE(9) LNH_CASE(11)
#undef LNH_CASE
      default: return UCOD_EINVAL;
    }
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_row_stats_h16(const void* x, float* stats, int rows, int D, float eps, void* stream) {
  using namespace ucod;
  if (!x || !stats || rows <= 0 || D <= 0 || (D % 256) != 0 || D > 1536) return UCOD_EINVAL;
  UCOD_PROF(PROF_ROW_STATS, stream);
  hipStream_t s = (hipStream_t)stream;
  static const int strips_env = [] { const char* e = ucod::lab_env("UCOD_STATS_STRIPS"); return e ? atoi(e) : -1; }();
  const int per_wave = strips_env > 0 ? strips_env : 4;
  const int nstrips = (rows + 1) / 2;
  const int want = cdiv(cdiv(nstrips, per_wave), 4);
  const dim3 grid((unsigned)(want > 256 ? want : (cdiv(nstrips, 4) < 256 ? cdiv(nstrips, 4) : 256))), block(256);
  switch (D / 256) {
#define RS_CASE(n) \
  case n: hipLaunchKernelGGL(row_stats_h16_kernel<n>, grid, block, 0, s, (const u32x4*)x, (float2*)stats, rows, D, eps, ucod::resid16_overflow_counter()); break;
    RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6)
#undef RS_CASE
    default: return UCOD_EINVAL;
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_cls_rows_h16(void* x, const float* cls, const float* pos, int B, int tok, int D, void* stream) {
  if (!x || !cls || !pos || B <= 0 || tok <= 0 || D <= 0) return UCOD_EINVAL;
  UCOD_PROF(ucod::PROF_CLS, stream);
  hipLaunchKernelGGL(ucod::cls_rows_h16_kernel, dim3(ucod::cdiv((long)B * D, 256)), dim3(256), 0, (hipStream_t)stream, (unsigned short*)x, cls, pos, B, tok, D, ucod::resid16_overflow_counter());
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_cls_rows_h16_stats(void* x, const float* cls, const float* pos, float* row_partials, int nslot, int B, int tok, int D, void* stream) {
  if (!x || !cls || !pos || !row_partials || nslot <= 0 || nslot > 256 || B <= 0 || tok <= 0 || D <= 0 || (D % 64) != 0 || nslot != D / 64) return UCOD_EINVAL;
  UCOD_PROF(ucod::PROF_CLS, stream);
  hipLaunchKernelGGL(ucod::cls_rows_h16_stats_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (unsigned short*)x, cls, pos, (float2*)row_partials, nslot, tok, D,
                     ucod::resid16_overflow_counter());
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_patch_im2col(const float* img, void* patches, int B, int C, int H, int W, int P, int Kpad, void* stream) {
  if (!img || !patches || B <= 0 || C <= 0 || P <= 0 || H % P || W % P || Kpad < C * P * P || (Kpad % 64) != 0) return UCOD_EINVAL;
  const int gh = H / P, gw = W / P;
  const size_t total = (size_t)B * gh * gw * (Kpad / 2);
  UCOD_PROF(ucod::PROF_IM2COL, stream);
  const int K = C * P * P;
  const size_t lds = ((size_t)P * W + (size_t)P * P) * sizeof(float);
  if (((P * P) & 3) == 0 && (K & 3) == 0 && (Kpad & 3) == 0 && lds <= 60 * 1024) {
    hipLaunchKernelGGL(ucod::im2col_rows_kernel, dim3(B * gh * C), dim3(256), lds, (hipStream_t)stream, img, (bf16_raw*)patches, C, H, W, P, Kpad, gh, gw);
  } else {
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(ucod::im2col_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, (bf16_raw*)patches, B, C, H, W, P,
                       Kpad, gh, gw);
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_cls_rows(float* x, const float* cls, const float* pos, int B, int tok, int D, void* stream) {
  if (!x || !cls || !pos || B <= 0 || tok <= 0 || D <= 0) return UCOD_EINVAL;
  UCOD_PROF(ucod::PROF_CLS, stream);
  hipLaunchKernelGGL(ucod::cls_rows_kernel, dim3(ucod::cdiv((long)B * D, 256)), dim3(256), 0, (hipStream_t)stream, x, cls, pos, B, tok, D);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_cast_f32_bf16(const float* src, void* dst, size_t n, void* stream) {
  if (!src || !dst) return UCOD_EINVAL;
  if (n == 0) return UCOD_OK;
  UCOD_PROF(ucod::PROF_CAST, stream);
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(ucod::cast_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_raw*)dst, n);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

__global__ void fill_qscale_kernel(float* __restrict__ v, int D, float c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * D) v[i] = i < D ? c : 1.f;
}
extern "C" int ucod_fill_qscale(float* v, int D, float c, void* stream) {
  if (!v || D <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(fill_qscale_kernel, dim3(ucod::cdiv(3L * D, 256)), dim3(256), 0, (hipStream_t)stream, v, D, c);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

__global__ void fill_qscale3_kernel(float* v, int D, float cq, float ck, float cv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * D) v[i] = i < D ? cq : (i < 2 * D ? ck : cv);
}
extern "C" int ucod_fill_qscale3(float* v, int D, float cq, float ck, float cv, void* stream) {
  if (!v || D <= 0) return UCOD_EINVAL;
  hipLaunchKernelGGL(fill_qscale3_kernel, dim3(ucod::cdiv(3L * D, 256)), dim3(256), 0, (hipStream_t)stream, v, D, cq, ck, cv);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_abi_version(void) { return UCOD_ABI_VERSION; }

extern "C" int ucod_device_is_gfx950(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  const char* a = p.gcnArchName;
  return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
}
