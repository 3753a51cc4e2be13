// bf16 MFMA GEMM with fused epilogues for the ViT backbone (SURVEY.md 8a rows B1,B4,B5,B7,B8; training epilogues: row B9).
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        A:[M,K]  B:[N,K]  both row-major, K contiguous (bf16)
//
// which is exactly nn.Linear (y = x W^T) with A = activations, B = weight -- and, with the operands
// swapped (A = W_key, B = tokens), the last layer's key projection written straight into the
// [B,C,h,w] map the reference's hook produces (data/utils/feature_extractor.py:46-47,55-58).
//
// Three kernels (variant numbers of ucod_gemm_bf16 in brackets):
//   * gemm_bf16_kernel       [1,2]    128x128x64 tile, 4 waves (2x2), 2 workgroups per CU: small shapes.
//   * gemm_bf16_big_kernel   [3-6,9,10] 256 x 256|192 x 64 tile, 8 waves (2x4), ONE workgroup per CU; LDS-DMA operands that stay in
//                                     flight across raw s_barriers, staggered wave groups, 4 or 2 barrier phases per K-tile; what
//                                     `auto` picks for every large shape (9/10).  Section comment below.
//   * gemm_bf16_pers_kernel  [7,8]    persistent form of the large tile (next tile's first K-tile under the epilogue).
// All share the XCD-aware tile order (blocks b and b+8 share an XCD/L2) and, for the hot epilogues, `big_epilogue`: bias as the
// accumulator's initial value, column scale in the MFMA layout, drain through wave-private LDS into 16-byte buffer stores with no
// load between two stores, f32 residual / saved pre-activation double-buffered across passes.
#include <cstdlib>
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

constexpr int BM = 128, BN = 128, BK = 64;
// cache policy of the large-tile epilogue's output stores (aux bits of buffer_store: 0 default, 2 nt, 16 sc1 = write-through, the line
// is dropped from the XCD's L2 instead of displacing operand panels)
#ifndef UCOD_ST_AUX
#define UCOD_ST_AUX 0
#endif


// Cache policy of the operand LDS-DMA loads (experiment builds: make variant NAME=.. DEFS=-DUCOD_LD_AUX_A=2): 0 default, 2 = nt
#ifndef UCOD_LD_AUX_A
#define UCOD_LD_AUX_A 0
#endif
#ifndef UCOD_LD_AUX_B
#define UCOD_LD_AUX_B 0
#endif

struct GemmArgs {
  unsigned long long* stamps;   // diagnostic builds only (UCOD_GEMM_STAMPS): per-workgroup segment cycle sums, never read by kernels
  const bf16_raw* A;
  const bf16_raw* B;
  void* out;
  const float* bias;
  const float* scale;
  const float* resid;
  const float* pos;
  const void* aux;    // GELU_BWD: bf16 [M,N] pre-activation of the forward fc1
  void* out2;         // BIAS_GELU_SAVE: bf16 [M,N] pre-activation output
  int M, N, K;
  int tok;   // tokens per image incl. CLS (PATCH / KEY epilogues)
  int tiles_m, tiles_n;
  int main_tiles;      // large-tile kernel, leftover-as-patches mode (see patch_phase): workgroups launched = whole tiles computed; 0 = off
  int patches_per_wg;  // 16 x 32 patches of the remaining tiles each workgroup computes on the side
  int group_m;         // large-tile kernels: row-tiles per group of the tile order inside an XCD's chunk (see tile_of)
  int col_fast;        // 1: column-tile fastest inside a group (one A panel's N-sweep back to back), 0: row-tile fastest
};

// Tile order of the large-tile kernels inside one XCD's contiguous chunk of the grid: groups of `group_m` row-tiles x all column-tiles.
// row-tile fastest (col_fast = 0): the 32 workgroups resident on an XCD share group_m A panels and 32/group_m B panels;
// column-tile fastest (col_fast = 1): they share 32/tiles_n A panels and ALL B panels, which then stay hot in the XCD's L2 while the A
// panels stream through once -- the better order when the whole weight matrix fits beside the streaming panels (4 MiB L2 per XCD).
__device__ __forceinline__ void tile_of(const GemmArgs& a, int wg, int& tm, int& tn) {
  const int gm = a.group_m;
  const int per_group = gm * a.tiles_n;
  const int grp = wg / per_group, first_m = grp * gm;
  const int gsz = (a.tiles_m - first_m) < gm ? (a.tiles_m - first_m) : gm;
  const int in_grp = wg - grp * per_group;
  if (a.col_fast) {
    tm = first_m + in_grp / a.tiles_n;
    tn = in_grp - (in_grp / a.tiles_n) * a.tiles_n;
  } else {
    tm = first_m + in_grp % gsz;
    tn = in_grp / gsz;
  }
}

// 16-byte chunk swizzle inside a 128-byte (64 x bf16) tile row: conflict-free ds_read_b128 for the
// 16x16x32 fragment pattern (rows l&15, chunk l>>4) under the 64-bank / 16-lane-group rule.
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <bool GLDS, int NI = 4>
__device__ __forceinline__ void stage_tile(const bf16_raw* __restrict__ G, int rows_total, int row0, int K, int k0,
                                           char* lds_tile, int wave, int lane, u32x4 (&regs)[NI]) {
  // 32*NI rows x 8 chunks; wave-instruction i covers rows (i*4+wave)*8 .. +7, lane -> (row l>>3, phys chunk l&7)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int r = (i * 4 + wave) * 8 + (lane >> 3);
    const int c = swz(r, lane & 7);
    int gr = row0 + r;
    gr = gr < rows_total ? gr : rows_total - 1;
    const bf16_raw* src = G + (size_t)gr * K + k0 + c * 8;
    if constexpr (GLDS) {
      char* dst = lds_tile + (i * 4 + wave) * 1024;  // wave-uniform base; HW adds lane*16
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    } else {
      regs[i] = *reinterpret_cast<const u32x4*>(src);
    }
  }
}

template <int NI = 4>
__device__ __forceinline__ void write_tile(char* lds_tile, int wave, int lane, const u32x4 (&regs)[NI]) {
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    *reinterpret_cast<u32x4*>(lds_tile + (i * 4 + wave) * 1024 + lane * 16) = regs[i];
  }
}

// exact-erf GELU (transformers ACT2FN["gelu"], modeling_dinov2.py:289), two elements per call so the polynomial runs
// on v_pk_fma_f32.  With a = |x|:  0.5*erfc(a/sqrt2) = exp2(-(1 + a*(d1 + d2 a + d3 a^2 + d4 a^3 + d5 a^4)))  (weighted
// minimax fit of -log2 erfc, |erfc err| <= 5e-6 and RELATIVE in the tail), and  gelu(x) = max(x,0) - a * 0.5*erfc(a/sqrt2).
// Max |gelu err| = 7.1e-7 over [-30,30] in fp32 -- the same as the Abramowitz-Stegun 7.1.26 form it replaces, at one
// transcendental and ~9 issue slots per element instead of two and ~22 (the fc1 epilogue runs it 134 M times per
// launch and was VALU-bound: 6.3 k of its 18.1 k cycles per 256x256 tile).  d5 > 0, so large |x| underflows to t = 0.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 a = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  f32x2 p = a * 4.881021588e-04f + (-7.198718842e-03f);
  p = p * a + 5.214663086e-02f;
  p = p * a + 4.595958292e-01f;
  p = p * a + 1.151000509e+00f;
  p = p * a + 1.0f;
  const f32x2 t = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
  const f32x2 pos = {__builtin_fmaxf(x[0], 0.f), __builtin_fmaxf(x[1], 0.f)};
  return pos - a * t;
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2((f32x2){x, x})[0]; }
// d/dx gelu(x) = Phi(x) + x * phi(x), Phi from the same 0.5*erfc fit (backbone-backward mode, fc1 dgrad epilogue)
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x) {
  const f32x2 a = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  f32x2 p = a * 4.881021588e-04f + (-7.198718842e-03f);
  p = p * a + 5.214663086e-02f;
  p = p * a + 4.595958292e-01f;
  p = p * a + 1.151000509e+00f;
  p = p * a + 1.0f;
  const f32x2 t = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};          // 0.5 * erfc(|x| / sqrt 2)
  const f32x2 cdf = {x[0] >= 0.f ? 1.f - t[0] : t[0], x[1] >= 0.f ? 1.f - t[1] : t[1]};
  const f32x2 xx = x * x * (-0.72134752044448170f);                                         // -x^2/2 * log2(e)
  const f32x2 pdf = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
  return cdf + x * pdf * 0.39894228040143268f;
}

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& a, int m, int n, float v) {
  if (m >= a.M || n >= a.N) return;
  if constexpr (EPI == UCOD_EPI_BIAS_BF16) {
    reinterpret_cast<bf16_raw*>(a.out)[(size_t)m * a.N + n] = f32_to_h((v + a.bias[n]) * (a.scale ? a.scale[n] : 1.f));
  } else if constexpr (EPI == UCOD_EPI_BIAS_GELU_BF16) {
    reinterpret_cast<bf16_raw*>(a.out)[(size_t)m * a.N + n] = f32_to_h(gelu_erf(v + a.bias[n]));
  } else if constexpr (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32) {
    const size_t i = (size_t)m * a.N + n;
    reinterpret_cast<float*>(a.out)[i] = a.resid[i] + a.scale[n] * (v + a.bias[n]);
  } else if constexpr (EPI == UCOD_EPI_PATCH_TOKENS_F32) {
    // row m = b*(tok-1)+p  ->  token row b*tok + 1 + p ; + bias + position embedding of token 1+p
    const int np = a.tok - 1;
    const int b = m / np, p = m - b * np;
    reinterpret_cast<float*>(a.out)[((size_t)b * a.tok + 1 + p) * a.N + n] = v + a.bias[n] + a.pos[(size_t)(1 + p) * a.N + n];
  } else if constexpr (EPI == UCOD_EPI_PATCH_TOKENS_H16) {
    const int np = a.tok - 1;
    const int b = m / np, p = m - b * np;
    reinterpret_cast<unsigned short*>(a.out)[((size_t)b * a.tok + 1 + p) * a.N + n] =
        __builtin_bit_cast(unsigned short, (_Float16)(v + a.bias[n] + a.pos[(size_t)(1 + p) * a.N + n]));
  } else if constexpr (EPI == UCOD_EPI_KEY_NCHW_F32) {
    // m = channel, n = global token index; drop CLS, write [B, C, tok-1]
    const int b = n / a.tok, t = n - b * a.tok;
    if (t == 0) return;
    reinterpret_cast<float*>(a.out)[((size_t)b * a.M + m) * (a.tok - 1) + (t - 1)] = v + a.bias[m];
  } else if constexpr (EPI == UCOD_EPI_BIAS_F32) {
    reinterpret_cast<float*>(a.out)[(size_t)m * a.N + n] = v + a.bias[n];
  }
}

// Four consecutive columns n..n+3 of output row m (n % 4 == 0, N % 4 == 0): vector loads / stores.
template <int EPI>
__device__ __forceinline__ void epilogue_store4(const GemmArgs& a, int m, int n, f32x4 v) {
  if (m >= a.M || n >= a.N) return;
  if constexpr (EPI == UCOD_EPI_KEY_NCHW_F32) {
    // four consecutive tokens of one image, none of them CLS: one dword-aligned 16-byte store into [B, C, tok-1] (row starts are
    // only 4-byte aligned there: tok-1 is odd); groups that touch a CLS token or straddle two images go token by token
    const int b = n / a.tok, t = n - b * a.tok;
    if (t >= 1 && t + 3 < a.tok && n + 3 < a.N) {
      typedef f32x4 f32x4_u __attribute__((aligned(4)));
      const float bm = a.bias[m];
      *reinterpret_cast<f32x4_u*>(reinterpret_cast<float*>(a.out) + ((size_t)b * a.M + m) * (a.tok - 1) + (t - 1)) = v + (f32x4){bm, bm, bm, bm};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) epilogue_store<EPI>(a, m, n + e, v[e]);
    }
  } else {
    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + n);
    if constexpr (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16) {
      f32x4 o = v + b;
      if constexpr (EPI == UCOD_EPI_BIAS_BF16) {
        if (a.scale) o = o * *reinterpret_cast<const f32x4*>(a.scale + n);
      }
      if constexpr (EPI == UCOD_EPI_BIAS_GELU_BF16) {
        const f32x2 g0 = gelu_erf2((f32x2){o[0], o[1]}), g1 = gelu_erf2((f32x2){o[2], o[3]});
        o = (f32x4){g0[0], g0[1], g1[0], g1[1]};
      }
      u32x2 w;
      w[0] = pack_h2(o[0], o[1]);
      w[1] = pack_h2(o[2], o[3]);
      *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_raw*>(a.out) + (size_t)m * a.N + n) = w;
    } else if constexpr (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32) {
      const size_t i = (size_t)m * a.N + n;
      const f32x4 r = *reinterpret_cast<const f32x4*>(a.resid + i);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + i) = r + sc * (v + b);
    } else if constexpr (EPI == UCOD_EPI_PATCH_TOKENS_F32 || EPI == UCOD_EPI_PATCH_TOKENS_H16) {
      const int np = a.tok - 1;
      const int bi = m / np, p = m - bi * np;
      const f32x4 ps = *reinterpret_cast<const f32x4*>(a.pos + (size_t)(1 + p) * a.N + n);
      const f32x4 o = v + b + ps;
      if constexpr (EPI == UCOD_EPI_PATCH_TOKENS_H16) {
        u32x2 w;
        w[0] = pack_f16x2(o[0], o[1]);
        w[1] = pack_f16x2(o[2], o[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(a.out) + ((size_t)bi * a.tok + 1 + p) * a.N + n) = w;
      } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + ((size_t)bi * a.tok + 1 + p) * a.N + n) = o;
      }
    } else {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + (size_t)m * a.N + n) = v + b;
    }
  }
}

// Eight consecutive columns of a bf16-output row: ONE 16-byte store per lane.  The per-CU store path is issue-bound
// (~7 B/clk/CU with 8-byte stores; measured 12 us to drain a 256x256 bf16 tile): halving the instruction count at equal
// bytes halves the drain time.
template <int EPI>
__device__ __forceinline__ void epilogue_store8_bf16(const GemmArgs& a, int m, int n, f32x4 v0, f32x4 v1) {
  if (m >= a.M || n >= a.N) return;
  f32x4 o0 = v0 + *reinterpret_cast<const f32x4*>(a.bias + n);
  f32x4 o1 = v1 + *reinterpret_cast<const f32x4*>(a.bias + n + 4);
  if constexpr (EPI == UCOD_EPI_BIAS_BF16) {
    if (a.scale) {
      o0 = o0 * *reinterpret_cast<const f32x4*>(a.scale + n);
      o1 = o1 * *reinterpret_cast<const f32x4*>(a.scale + n + 4);
    }
  } else {
    const f32x2 g0 = gelu_erf2((f32x2){o0[0], o0[1]}), g1 = gelu_erf2((f32x2){o0[2], o0[3]});
    const f32x2 g2 = gelu_erf2((f32x2){o1[0], o1[1]}), g3 = gelu_erf2((f32x2){o1[2], o1[3]});
    o0 = (f32x4){g0[0], g0[1], g1[0], g1[1]};
    o1 = (f32x4){g2[0], g2[1], g3[0], g3[1]};
  }
  u32x4 w;
  w[0] = pack_h2(o0[0], o0[1]);
  w[1] = pack_h2(o0[2], o0[3]);
  w[2] = pack_h2(o1[0], o1[1]);
  w[3] = pack_h2(o1[2], o1[3]);
  *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_raw*>(a.out) + (size_t)m * a.N + n) = w;
}

// Epilogue of one wave's RxWCOLS f32 sub-tile through a wave-private LDS region: accumulators are written with the
// MFMA C layout (lane -> column), read back row-major 16/32 B per lane, so global traffic is whole row segments moved
// by 16-byte-per-lane instructions (4-8x fewer, wider instructions than storing straight from the accumulator layout).
template <int EPI, int WCOLS, int ROWS>
__device__ __forceinline__ void drain_rows(const GemmArgs& a, const char* wbase, int m_first, int n_first, int lane) {
  constexpr bool BF16_OUT = (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16);
  if constexpr (BF16_OUT && (WCOLS % 8) == 0 && (ROWS * (WCOLS / 8)) % 64 == 0) {
    constexpr int CH = WCOLS / 8;                     // 32-byte (8 x f32) chunks per row -> 16-byte bf16 stores
    if ((a.N & 7) == 0) {
#pragma unroll
      for (int it = 0; it < ROWS * CH / 64; ++it) {
        const int idx = it * 64 + lane, r = idx / CH, c = idx - r * CH;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(wbase + r * (WCOLS * 4) + c * 32);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(wbase + r * (WCOLS * 4) + c * 32 + 16);
        epilogue_store8_bf16<EPI>(a, m_first + r, n_first + c * 8, v0, v1);
      }
      return;
    }
  }
  constexpr int CH = WCOLS / 4;                       // 16-byte chunks per row
  static_assert((ROWS * CH) % 64 == 0, "whole wave instructions");
#pragma unroll
  for (int it = 0; it < ROWS * CH / 64; ++it) {
    const int idx = it * 64 + lane, r = idx / CH, c = idx - r * CH;
    const f32x4 v = *reinterpret_cast<const f32x4*>(wbase + r * (WCOLS * 4) + c * 16);
    epilogue_store4<EPI>(a, m_first + r, n_first + c * 4, v);
  }
}

// T = 128: the 128 x 128 tile (each wave 64 x 64).  T = 64: a 64 x 64 tile (each wave 32 x 32) for launches whose 128-tiles would
// leave most CUs idle -- a batch-1 backbone pass (Look-Twice, validation) has 66 tiles of proj / fc2 on 256 CUs.
template <int EPI, bool GLDS, int T = 128>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs a) {
  constexpr int TB = T * BK * 2, NI = T / 32, FI = T / 32;            // bytes per operand per stage; DMA instructions per wave; 16-row fragments per wave
  __shared__ __attribute__((aligned(16))) char smem[4 * TB];          // [stage][A|B]; reused as the epilogue staging (4 waves x (T/2)^2 f32 = 4 * TB / 4 bytes... <= 4 * TB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware bijective remap of the 1-D grid, then tn fastest (neighbours share the A row panel)
  const int nwg = a.tiles_m * a.tiles_n;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
  const int tm = wg / a.tiles_n, tn = wg - tm * a.tiles_n;
  const int m0 = tm * T, n0 = tn * T;

  f32x4 acc[FI][FI];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = a.K / BK;
  u32x4 ra[NI], rb[NI];
  stage_tile<GLDS, NI>(a.A, a.M, m0, a.K, 0, smem, wave, lane, ra);
  stage_tile<GLDS, NI>(a.B, a.N, n0, a.K, 0, smem + TB, wave, lane, rb);
  if constexpr (!GLDS) {
    write_tile<NI>(smem, wave, lane, ra);
    write_tile<NI>(smem + TB, wave, lane, rb);
  }

  for (int t = 0; t < nt; ++t) {
    __syncthreads();  // tile t visible (the fence drains the LDS-DMA); everyone is done with the other stage
    char* curA = smem + (t & 1) * 2 * TB;
    char* curB = curA + TB;
    char* nxtA = smem + ((t + 1) & 1) * 2 * TB;
    const bool more = (t + 1 < nt);
    if (more) {
      stage_tile<GLDS, NI>(a.A, a.M, m0, a.K, (t + 1) * BK, nxtA, wave, lane, ra);
      stage_tile<GLDS, NI>(a.B, a.N, n0, a.K, (t + 1) * BK, nxtA + TB, wave, lane, rb);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      hx8 fa[FI], fb[FI];
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const int r = wr * (T / 2) + i * 16 + (lane & 15);
        fa[i] = *reinterpret_cast<const hx8*>(curA + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
      }
#pragma unroll
      for (int j = 0; j < FI; ++j) {
        const int r = wc * (T / 2) + j * 16 + (lane & 15);
        fb[j] = *reinterpret_cast<const hx8*>(curB + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
      }
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FI; ++j) acc[i][j] = UCOD_MFMA16(fa[i], fb[j], acc[i][j]);
    }
    if constexpr (!GLDS) {
      if (more) {
        write_tile<NI>(nxtA, wave, lane, ra);
        write_tile<NI>(nxtA + TB, wave, lane, rb);
      }
    }
  }

  // C/D map of v_mfma_f32_16x16x32: col = lane&15, row = (lane>>4)*4 + reg.  Drain through LDS (see drain_rows).
  __syncthreads();
  if ((a.N & 3) == 0) {
    constexpr int WT = T / 2;                                      // the wave's square sub-tile
    char* wbase = smem + wave * (WT * WT * 4);
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FI; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *reinterpret_cast<float*>(wbase + (i * 16 + (lane >> 4) * 4 + rg) * (WT * 4) + (j * 16 + (lane & 15)) * 4) = acc[i][j][rg];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    drain_rows<EPI, WT, WT>(a, wbase, m0 + wr * WT, n0 + wc * WT, lane);
  } else {
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FI; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          epilogue_store<EPI>(a, m0 + wr * (T / 2) + i * 16 + (lane >> 4) * 4 + rg, n0 + wc * (T / 2) + j * 16 + (lane & 15), acc[i][j][rg]);
  }
}

// ---- large-tile epilogue (one-shot and persistent kernels) ------------------------------------------------------
// s_memtime stamps (tools/gemm_stamps.py) showed the old epilogue costing 12 k (bf16 out) to 40 k (f32 residual) cycles per
// 256-wide tile INDEPENDENT of how many CUs were active: not bandwidth, but a latency chain -- every 16-byte store was
// preceded by bias / scale / residual loads whose `s_waitcnt vmcnt(0)` also drained the stores issued just before (vmcnt
// counts stores on gfx9), i.e. one ~700-cycle store round trip per store instruction.  So, for the three hot epilogues:
//   * the bias is the accumulator's INITIAL value and the per-column scale (Q pre-scale, LayerScale gamma) is applied in the
//     MFMA C layout, where a lane owns one column per 16-wide tile: NT + NT registers, loaded once per output tile;
//   * GELU runs in the C layout too, so the row-major drain of a bf16 tile is ds_read -> cvt -> 16-byte store, no loads;
//   * the f32 residual is double buffered: the loads of pass p+1 are issued BEFORE the stores of pass p, and vmcnt retires
//     in order, so the wait for them leaves pass p's stores in flight.  (out may alias resid: passes touch disjoint rows.)
template <int EPI>
constexpr bool kColFused = (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16 || EPI == UCOD_EPI_BIAS_SCALE_RESID_F32 ||
                            EPI == UCOD_EPI_BIAS_F32 || EPI == UCOD_EPI_GELU_BWD_BF16 || EPI == UCOD_EPI_BIAS_GELU_SAVE_BF16 ||
                            EPI == UCOD_EPI_QKV_FP8 || EPI == UCOD_EPI_BIAS_SCALE_RESID_H16);
template <int EPI>
constexpr bool kF32Out = (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32 || EPI == UCOD_EPI_BIAS_F32);

template <int EPI, int NT>
__device__ __forceinline__ void load_col_consts(const GemmArgs& a, int ncol0, float (&cb)[NT], float (&cs)[NT]) {
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    cb[j] = 0.f;
    cs[j] = 1.f;
    if constexpr (kColFused<EPI>) {
      int n = ncol0 + j * 16;
      n = n < a.N ? n : a.N - 1;
      if constexpr (EPI == UCOD_EPI_BIAS_F32 || EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_QKV_FP8) {
        cb[j] = (a.bias ? a.bias : reinterpret_cast<const float*>(a.B))[n];   // NULL bias = plain product (dgrad GEMMs): selected in finish_col_consts
      } else if constexpr (EPI != UCOD_EPI_GELU_BWD_BF16) {
        cb[j] = a.bias[n];
      }
      if constexpr (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32 || EPI == UCOD_EPI_BIAS_SCALE_RESID_H16) cs[j] = a.scale[n];
      // optional scale: unconditional load now (a branch here costs a vmcnt(0) at the join, ahead of the operand DMAs),
      // select at the point of use (finish_col_consts) so nothing waits on the load before the DMAs are out
      if constexpr (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_QKV_FP8) cs[j] = (a.scale ? a.scale : reinterpret_cast<const float*>(a.B))[n];
    }
  }
}

template <int EPI, int NT>
__device__ __forceinline__ void finish_col_consts(const GemmArgs& a, float (&cb)[NT], float (&cs)[NT]) {
  if constexpr (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_QKV_FP8) {
#pragma unroll
    for (int j = 0; j < NT; ++j) cs[j] = a.scale ? cs[j] : 1.f;
  }
  if constexpr (EPI == UCOD_EPI_BIAS_F32 || EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_QKV_FP8) {
#pragma unroll
    for (int j = 0; j < NT; ++j) cb[j] = a.bias ? cb[j] : 0.f;
  }
}

// acc: the wave's 128 x 16*NT tile (8 row-tiles x NT column-tiles, C layout col = lane&15, row = (lane>>4)*4 + reg), bias
// already inside for the fused epilogues; cs = per-column scale.  wbase: wave-private 32 x WCOLS f32 staging area.  Four passes of 32 rows.
template <int EPI, int NT, int NI = 8, int AUX = UCOD_ST_AUX>
__device__ __forceinline__ void big_epilogue(const GemmArgs& a, f32x4 (&acc)[NI][NT], const float (&cs)[NT], char* wbase,
                                             int m_first, int n_first, int lane) {
  constexpr int WCOLS = 16 * NT, PR = 32;
  constexpr int NP = (NI + 1) / 2;                    // passes of 32 rows; with NI odd the last pass holds 16 rows (rows 16..31 masked off)
  static_assert(NI == 8 || kColFused<EPI>, "odd row-tile counts only in the column-fused epilogues");
  auto rows_in = [&](int pass) { return (NI - 2 * pass) >= 2 ? 32 : 16; };
  auto stage = [&](int pass) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (pass * 2 + i >= NI) continue;
        f32x4 v = acc[pass * 2 + i < NI ? pass * 2 + i : 0][j];
        if constexpr (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_SCALE_RESID_F32 || EPI == UCOD_EPI_QKV_FP8 ||
                      EPI == UCOD_EPI_BIAS_SCALE_RESID_H16) v = v * cs[j];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *reinterpret_cast<float*>(wbase + (i * 16 + (lane >> 4) * 4 + rg) * (WCOLS * 4) + (j * 16 + (lane & 15)) * 4) = v[rg];
      }
  };
  if constexpr (!kColFused<EPI>) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      stage(pass);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      drain_rows<EPI, WCOLS, PR>(a, wbase, m_first + pass * PR, n_first, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else {
    // Row / column guards without branches (a branch per access makes hipcc fall back to vmcnt(0) before every store):
    // buffer descriptors over [first row of this wave's tile, end of the matrix) -- rows past M fail the range check and
    // are dropped (loads return 0) -- and columns past N get an offset beyond any descriptor.
    constexpr unsigned OOB = 0xFFFFFFF0u;
    constexpr int ELT = kF32Out<EPI> ? 4 : 2;
    const long rows_left = (long)a.M - m_first;
    const unsigned long left = rows_left > 0 ? (unsigned long)rows_left * a.N * ELT : 0ul;
    const unsigned records = left > 0xFFFFFFFFul ? 0xFFFFFFFFu : (unsigned)left;
    const size_t base = (size_t)(m_first < a.M ? m_first : 0) * a.N * ELT;
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + base, 0, records, 0x00020000);
    const unsigned row_bytes = (unsigned)a.N * ELT;
    const unsigned pass_bytes = PR * row_bytes;
    if constexpr (EPI == UCOD_EPI_QKV_FP8) {
      // The wave's 64 columns are one head of q, k or v (n_first is a multiple of 64): e4m3 rows of 64 bytes into
      // [q|k|v][image * heads + head][Npad][64].  Lane -> (row, 16-column chunk): one 16-byte store per 16 outputs.
      static_assert(NT == 4, "one head per wave");
      const int Dm = a.N / 3, heads = Dm >> 6, tok = a.tok, npad = ((tok + 63) >> 6) << 6;
      const int region = n_first / Dm, head = (n_first - region * Dm) >> 6;
      const size_t npairs = (size_t)(a.M / tok) * heads;
      char* dst0 = reinterpret_cast<char*>(a.out) + (size_t)region * npairs * npad * 64;
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) {
        stage(pass);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int idx = it * 64 + lane, r = idx >> 2, c = idx & 3;
          const int m = m_first + pass * PR + r;
          u32x4 w;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wbase + r * (WCOLS * 4) + c * 64 + e * 16);
            int p = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[0], -448.f), 448.f), fminf(fmaxf(v[1], -448.f), 448.f), 0, false);
            p = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[2], -448.f), 448.f), fminf(fmaxf(v[3], -448.f), 448.f), p, true);
            w[e] = (unsigned)p;
          }
          if (m < a.M && r < rows_in(pass) && n_first < a.N) {
            const int bimg = m / tok, t = m - bimg * tok;
            *reinterpret_cast<u32x4*>(dst0 + (((size_t)bimg * heads + head) * npad + t) * 64 + c * 16) = w;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else if constexpr (kF32Out<EPI>) {
      constexpr bool RESID = (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32);
      constexpr int CH = WCOLS / 4, ITS = PR * CH / 64;          // 16-byte chunks per row; wave instructions per pass
      static_assert((PR * CH) % 64 == 0, "whole wave instructions");
      const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(RESID ? (const void*)a.resid : (const void*)a.out)) + base, 0, RESID ? records : 0u, 0x00020000);
      unsigned off[ITS];                                          // byte offset of (row, chunk) of pass 0; + pass * 32 rows
      int lrow[ITS], lchk[ITS];
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int idx = it * 64 + lane;
        lrow[it] = idx / CH;
        lchk[it] = idx - lrow[it] * CH;
        const int n = n_first + lchk[it] * 4;
        off[it] = n < a.N ? (unsigned)lrow[it] * row_bytes + (unsigned)n * 4u : OOB;
      }
      // (the pass offset goes into the VGPR offset, not soffset: the range check covers only voffset + inst_offset)
      auto at = [&](int it, int pass) { return (off[it] == OOB || lrow[it] >= rows_in(pass)) ? OOB : off[it] + (unsigned)pass * pass_bytes; };
      u32x4 rb[2][ITS];
      if constexpr (RESID) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) rb[0][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off[it], 0, 0);
      }
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) {
        stage(pass);
        if constexpr (RESID) {
          if (pass + 1 < NP) {
#pragma unroll
            for (int it = 0; it < ITS; ++it)
              rb[(pass + 1) & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, at(it, pass + 1), 0, 0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          f32x4 o = *reinterpret_cast<const f32x4*>(wbase + lrow[it] * (WCOLS * 4) + lchk[it] * 16);
          if constexpr (RESID) o = o + __builtin_bit_cast(f32x4, rb[pass & 1][it]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_out, at(it, pass), 0, AUX);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {                                                      // bf16 out: 16-byte stores (launch() guarantees N % 8 == 0)
      constexpr bool GBWD = (EPI == UCOD_EPI_GELU_BWD_BF16), SAVE = (EPI == UCOD_EPI_BIAS_GELU_SAVE_BF16);
      constexpr bool RH16 = (EPI == UCOD_EPI_BIAS_SCALE_RESID_H16);   // second matrix = the f16 residual stream (may alias out)
      constexpr int CH = WCOLS / 8, ITS = PR * CH / 64;
      static_assert((PR * CH) % 64 == 0, "whole wave instructions");
      // second bf16 [M,N] matrix with the same geometry: the saved pre-activation, read (GELU_BWD) or written (GELU_SAVE)
      const void* second = GBWD ? a.aux : (SAVE ? (const void*)a.out2 : (RH16 ? (const void*)a.resid : (const void*)a.out));
      const auto rs_2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(second)) + base, 0,
                                                          (GBWD || SAVE || RH16) ? records : 0u, 0x00020000);
      // (row, chunk) of wave instruction `it`: recomputed where needed -- index arrays cost registers the persistent kernel lacks
      auto lrow = [&](int it) { return (it * 64 + lane) / CH; };
      auto lchk = [&](int it) { return (it * 64 + lane) - lrow(it) * CH; };
      auto at = [&](int it, int pass) {
        const int n = n_first + lchk(it) * 8;
        return (n < a.N && lrow(it) < rows_in(pass)) ? (unsigned)(pass * PR + lrow(it)) * row_bytes + (unsigned)n * 2u : OOB;
      };
      u32x4 pre[2][ITS];
      if constexpr (GBWD || RH16) {
#pragma unroll
        for (int it = 0; it < ITS; ++it) pre[0][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_2, at(it, 0), 0, 0);
      }
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) {
        stage(pass);
        if constexpr (GBWD || RH16) {
          if (pass + 1 < NP) {
#pragma unroll
            for (int it = 0; it < ITS; ++it) pre[(pass + 1) & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_2, at(it, pass + 1), 0, 0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          f32x4 v0 = *reinterpret_cast<const f32x4*>(wbase + lrow(it) * (WCOLS * 4) + lchk(it) * 32);
          f32x4 v1 = *reinterpret_cast<const f32x4*>(wbase + lrow(it) * (WCOLS * 4) + lchk(it) * 32 + 16);
          if constexpr (SAVE) {                                   // pre-activation out first
            u32x4 w;
            w[0] = pack_h2(v0[0], v0[1]);
            w[1] = pack_h2(v0[2], v0[3]);
            w[2] = pack_h2(v1[0], v1[1]);
            w[3] = pack_h2(v1[2], v1[3]);
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_2, at(it, pass), 0, 0);
          }
          if constexpr (SAVE || EPI == UCOD_EPI_BIAS_GELU_BF16) {  // GELU in the row-major layout (fewer live registers than in the C layout)
            const f32x2 g0 = gelu_erf2((f32x2){v0[0], v0[1]}), g1 = gelu_erf2((f32x2){v0[2], v0[3]});
            const f32x2 g2 = gelu_erf2((f32x2){v1[0], v1[1]}), g3 = gelu_erf2((f32x2){v1[2], v1[3]});
            v0 = (f32x4){g0[0], g0[1], g1[0], g1[1]};
            v1 = (f32x4){g2[0], g2[1], g3[0], g3[1]};
          }
          if constexpr (GBWD) {
            const u32x4 pw = pre[pass & 1][it];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              f32x2 x;
              {
                float x0, x1;
                unpack_h2(pw[e], x0, x1);
                x = (f32x2){x0, x1};
              }
              const f32x2 g = gelu_grad2(x);
              if (e < 2) { v0[2 * e] *= g[0]; v0[2 * e + 1] *= g[1]; }
              else { v1[2 * (e - 2)] *= g[0]; v1[2 * (e - 2) + 1] *= g[1]; }
            }
          }
          u32x4 w;
          if constexpr (RH16) {                                   // x_new = x_old + lambda (acc + b), all in IEEE fp16 storage
            const u32x4 pw = pre[pass & 1][it];
            float r[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) unpack_f16x2(pw[e], r[2 * e], r[2 * e + 1]);
            w[0] = pack_f16x2(v0[0] + r[0], v0[1] + r[1]);
            w[1] = pack_f16x2(v0[2] + r[2], v0[3] + r[3]);
            w[2] = pack_f16x2(v1[0] + r[4], v1[1] + r[5]);
            w[3] = pack_f16x2(v1[2] + r[6], v1[3] + r[7]);
          } else {
            w[0] = pack_h2(v0[0], v0[1]);
            w[1] = pack_h2(v0[2], v0[3]);
            w[2] = pack_h2(v1[0], v1[1]);
            w[3] = pack_h2(v1[2], v1[3]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, at(it, pass), 0, AUX);
          __builtin_amdgcn_sched_barrier(0);                      // keep chunks in order: hoisting every ds_read/cvt of a pass spills in the persistent kernel
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
}

// ---- leftover tiles as patches ------------------------------------------------------------------------------------
// One large-tile workgroup fills a CU, so a launch runs in rounds of n_cu tiles and the backbone's shapes all land just past
// a whole number of rounds (32 x 1370 rows: 516 = 2 x 256 + 4 tiles for proj / fc2): the last 4 tiles ran alone on 4 CUs
// while 252 idled -- 18 % of fc2, 11 % of proj (tools/gemm_tail_probe.py: M = 43520 vs 43840).  In this mode the launch has
// exactly rounds x n_cu workgroups, and the outputs of the remaining L tiles are cut into 16 x 32 patches that the
// workgroups compute on the side, one or two each, BEFORE their own tile: the patch's operand loads are in flight together
// with the tile's first K-tile DMAs (a latency every workgroup pays anyway), the K range is dealt round-robin to the 8 waves
// (v_mfma_f32_16x16x32_bf16 straight from global registers), partial sums meet in the LDS slot the main loop touches last.
// Deterministic: a patch is summed by one workgroup in a fixed order.  Result bits differ from the tile path only by the
// order of the f32 adds over K.
template <int C> struct PatchC { static constexpr int value = C; };

template <int EPI>
constexpr bool kPatchPrefetch = (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16 || EPI == UCOD_EPI_BIAS_SCALE_RESID_F32 ||
                                 EPI == UCOD_EPI_BIAS_F32);

template <int EPI, int BN_>
__device__ __forceinline__ void patch_phase(const GemmArgs& a, char* scratch /* 16 KiB */, int orig, int wave, int lane) {
  constexpr int PC = BN_ / 32, PPT = 16 * PC;                   // patches per leftover tile
  const int total = a.tiles_m * a.tiles_n;
  const int npatch = (total - a.main_tiles) * PPT;
  const int K = a.K, steps = K >> 5;
  const int l15 = lane & 15, q = lane >> 4;
  for (int pi = 0; pi < a.patches_per_wg; ++pi) {
    const int p = orig * a.patches_per_wg + pi;
    if (p >= npatch) break;
    const int wg = a.main_tiles + p / PPT, rem = p % PPT;
    int ptm, ptn;
    tile_of(a, wg, ptm, ptn);
    const int r0 = ptm * 256 + (rem / PC) * 16;
    const int c0 = ptn * BN_ + (rem % PC) * 32;
    if (r0 >= a.M || c0 >= a.N) continue;                       // ragged last row / column tile: nothing there
    // this thread's output of the patch (one of 16 x 32) and its epilogue operands, requested before the operand loads so that
    // nothing is left to fetch once the partial sums meet
    const int idx = wave * 64 + lane, om = r0 + (idx >> 5), on = c0 + (idx & 31);
    const bool live = om < a.M && on < a.N;
    const int cm = om < a.M ? om : a.M - 1, cn = on < a.N ? on : a.N - 1;
    float e_bias = 0.f, e_scale = 1.f, e_resid = 0.f;
    if constexpr (kPatchPrefetch<EPI>) {
      if (a.bias) e_bias = a.bias[cn];
      if constexpr (EPI == UCOD_EPI_BIAS_BF16) { if (a.scale) e_scale = a.scale[cn]; }
      if constexpr (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32) {
        e_scale = a.scale[cn];
        e_resid = a.resid[(size_t)cm * a.N + cn];
      }
    }
    int ar = r0 + l15, br0 = c0 + l15, br1 = c0 + 16 + l15;
    ar = ar < a.M ? ar : a.M - 1;
    br0 = br0 < a.N ? br0 : a.N - 1;
    br1 = br1 < a.N ? br1 : a.N - 1;
    const bf16_raw* pa = a.A + (size_t)ar * K + q * 8;
    const bf16_raw* pb0 = a.B + (size_t)br0 * K + q * 8;
    const bf16_raw* pb1 = a.B + (size_t)br1 * K + q * 8;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // this wave's k-steps: wave, wave + 8, ...; loaded in the largest chunks that fit (every load is a real one: the patch is
    // bound by the 64 B/clk/CU of the vector-memory path, 48 rows x K x 2 bytes per patch)
    auto chunk = [&](int s0, auto cnt) {
      constexpr int C = decltype(cnt)::value;
      hx8 fa[C], f0[C], f1[C];
#pragma unroll
      for (int i = 0; i < C; ++i) {
        const int st = s0 + 8 * i;
        fa[i] = *reinterpret_cast<const hx8*>(pa + st * 32);
        f0[i] = *reinterpret_cast<const hx8*>(pb0 + st * 32);
        f1[i] = *reinterpret_cast<const hx8*>(pb1 + st * 32);
      }
#pragma unroll
      for (int i = 0; i < C; ++i) {
        acc0 = UCOD_MFMA16(fa[i], f0[i], acc0);
        acc1 = UCOD_MFMA16(fa[i], f1[i], acc1);
      }
    };
    {
      int s0 = wave, left = (steps - wave + 7) >> 3;            // wave-uniform
      for (; left >= 12; left -= 12, s0 += 96) chunk(s0, PatchC<12>{});
      if (left >= 6) { chunk(s0, PatchC<6>{}); left -= 6; s0 += 48; }
      if (left >= 3) { chunk(s0, PatchC<3>{}); left -= 3; s0 += 24; }
      for (; left > 0; --left, s0 += 8) chunk(s0, PatchC<1>{});
    }
    // partial sums [wave][16 rows][32 cols]; C layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    float* sc = reinterpret_cast<float*>(scratch) + wave * 512;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      sc[(4 * q + rg) * 32 + l15] = acc0[rg];
      sc[(4 * q + rg) * 32 + 16 + l15] = acc1[rg];
    }
    __syncthreads();
    {
      const float* rd = reinterpret_cast<const float*>(scratch) + idx;
      float v = rd[0];
#pragma unroll
      for (int w = 1; w < 8; ++w) v += rd[w * 512];
      if constexpr (kPatchPrefetch<EPI>) {
        if (live) {
          const size_t o = (size_t)om * a.N + on;
          if constexpr (EPI == UCOD_EPI_BIAS_BF16) reinterpret_cast<bf16_raw*>(a.out)[o] = f32_to_h((v + e_bias) * e_scale);
          else if constexpr (EPI == UCOD_EPI_BIAS_GELU_BF16) reinterpret_cast<bf16_raw*>(a.out)[o] = f32_to_h(gelu_erf(v + e_bias));
          else if constexpr (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32) reinterpret_cast<float*>(a.out)[o] = e_resid + e_scale * (v + e_bias);
          else reinterpret_cast<float*>(a.out)[o] = v + e_bias;
        }
      } else {
        epilogue_store<EPI>(a, om, on, v);
      }
    }
    if (pi + 1 < a.patches_per_wg) __syncthreads();            // scratch is reused by the next patch
  }
}

// =====================================================================================================
// Large-tile kernel: 256 x (64*NT) x 64 block tile, 8 waves (2 in M x 4 in N), one workgroup per CU.
//   * per wave 128 x 16*NT outputs; a K-tile is consumed in FOUR phases of 32 rows each (2 x NT tiles x 2 k-steps
//     = 4*NT MFMAs per phase); the wave's B fragments are read once per K-tile (phase 1) and stay in registers;
//   * LDS = two K-tile buffers {A0 | A1 | B}; operands arrive by 16-byte LDS-DMA that stays IN FLIGHT across the
//     phase barriers: phase 1/2 stage A0/A1 of tile t+1 into the other buffer, phase 3/4 stage B of tile t+2 into
//     THIS buffer (its B slot is dead after phase 1), and the only wait is a counted `s_waitcnt vmcnt(BN/64)` at
//     phase 4 that leaves exactly the B(t+2) DMAs outstanding; raw s_barrier (a __syncthreads would drain vmcnt);
//   * 256-row tiles halve the L2->LDS traffic per FLOP of the 128x128 kernel, which is L2-bandwidth bound
//     (2 WGs/CU x 32 KB per 1024 MFMA cycles ~ 39 TB/s chip-wide, above the ~34.5 TB/s L2 ceiling).
// Hazards: RAW -- every wave waits for its own DMAs (vmcnt) BEFORE the phase-4 barrier, reads happen after it;
//          WAR -- B slot of buffer b: last ds_read in phase 1 (retired before its MFMAs), first restaged in phase 3;
//                 A slots of buffer b^1: last read in phase 4 of tile t-1, first restaged in phase 1 of tile t,
//                 with the phase-4 barrier in between.
// =====================================================================================================
constexpr int SLOT_A = 128 * 128;  // bytes: 128 rows x 64 bf16

template <int NT>
struct BigCfg {
  static constexpr int BN_ = 64 * NT;
  static constexpr int NB = BN_ / 64;                 // LDS-DMA instructions per thread for the B tile
  static constexpr int BUF = 2 * SLOT_A + BN_ * 128;  // bytes per K-tile buffer
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

template <int EPI, int NT, bool STAGGER, int NPH = 4>
__global__ __launch_bounds__(512) void gemm_bf16_big_kernel(const GemmArgs a) {
  // NPH phases of 128/NPH rows per K-tile and wave group.  NPH = 2 halves the number of barrier intervals per MFMA (two
  // 32-MFMA intervals instead of four 16-MFMA ones per K-tile and group) at the price of 16 more fragment registers.
  constexpr int IT = 8 / NPH;                                     // 16-row i-tiles per phase
  using Cfg = BigCfg<NT>;
  __shared__ __attribute__((aligned(16))) char smem[2 * Cfg::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int nwg = a.main_tiles > 0 ? a.main_tiles : a.tiles_m * a.tiles_n;   // leftover-as-patches mode: the first main_tiles tiles of the order
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
  // grouped order inside each XCD's chunk: GROUP_M row-tiles x all column-tiles, row-tile fastest -- the workgroups that are
  // resident together on an XCD then share a few B (weight) panels and GROUP_M A panels that fit its 4 MiB L2, instead of
  // every row-tile streaming the whole weight matrix through L2 (FETCH_SIZE was 5x the algorithmic bytes on fc1).
  int tm, tn;
  tile_of(a, wg, tm, tn);
  const int m0 = tm * 256, n0 = tn * Cfg::BN_;
  const int K = a.K, nt = K / BK;

  // per-thread LDS-DMA source rows (fixed for the whole K loop): A0,A1 -> 2 instructions each; B -> NB instructions
  const bf16_raw* srcA[2][2];
  const bf16_raw* srcB[Cfg::NB];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (i * 8 + wave) * 8 + (lane >> 3);
      int gr = m0 + h * 128 + r;
      gr = gr < a.M ? gr : a.M - 1;
      srcA[h][i] = a.A + (size_t)gr * K + swz(r, lane & 7) * 8;
    }
#pragma unroll
  for (int i = 0; i < Cfg::NB; ++i) {
    const int r = (i * 8 + wave) * 8 + (lane >> 3);
    int gr = n0 + r;
    gr = gr < a.N ? gr : a.N - 1;
    srcB[i] = a.B + (size_t)gr * K + swz(r, lane & 7) * 8;
  }
  auto dmaA = [&](const bf16_raw* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, UCOD_LD_AUX_A);
  };
  auto dmaB = [&](const bf16_raw* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, UCOD_LD_AUX_B);
  };
  auto stageA = [&](int t, int h) {
    char* slot = smem + (t & 1) * Cfg::BUF + h * SLOT_A;
#pragma unroll
    for (int i = 0; i < 2; ++i) dmaA(srcA[h][i] + t * BK, slot + (i * 8 + wave) * 1024);
  };
  auto stageB = [&](int t, int i0, int i1) {
    char* slot = smem + (t & 1) * Cfg::BUF + 2 * SLOT_A;
#pragma unroll
    for (int i = 0; i < Cfg::NB; ++i)
      if (i >= i0 && i < i1) dmaB(srcB[i] + t * BK, slot + (i * 8 + wave) * 1024);
  };
  constexpr int B_SPLIT = Cfg::NB >= 2 ? 2 : 1;   // phase 3 issues [0,B_SPLIT), phase 4 the rest

  // per-column epilogue constants first (oldest in the vmcnt queue: landed long before the accumulators are initialised)
  float cb[NT], cs[NT];
  load_col_consts<EPI, NT>(a, n0 + wn * 16 * NT + (lane & 15), cb, cs);

  // prologue: tile 0 complete, B of tile 1 in flight
  stageA(0, 0);
  stageA(0, 1);
  stageB(0, 0, Cfg::NB);
  if (nt > 1) stageB(1, 0, Cfg::NB);
  if constexpr (EPI != UCOD_EPI_GELU_BWD_BF16 && EPI != UCOD_EPI_BIAS_GELU_SAVE_BF16) {
    // scratch: the A0 slot of buffer 1, first written by the DMAs of K-tile 1 after the barrier below.  vmcnt retires in order,
    // so the patch's stores (older than every later DMA) never disturb the counted waits of the main loop.
    if (a.patches_per_wg > 0) patch_phase<EPI, Cfg::BN_>(a, smem + Cfg::BUF, orig, wave, lane);
  }
  if (nt > 1) wait_vmcnt<Cfg::NB>(); else wait_vmcnt<0>();
  finish_col_consts<EPI, NT>(a, cb, cs);
  f32x4 acc[8][NT];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){cb[j], cb[j], cb[j], cb[j]};
  __builtin_amdgcn_s_barrier();
  // STAGGER: the wm==1 waves run one barrier interval behind the wm==0 waves, so on every SIMD one wave is in its
  // MFMA interval while its partner is in its LDS-read / DMA-issue interval (two barriers per phase: R | M).
  // All waves execute the same number of barriers (extra one here for wm==1, extra one after the loop for wm==0).
  if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();

  for (int t = 0; t < nt; ++t) {
    const char* bufA = smem + (t & 1) * Cfg::BUF + wm * SLOT_A;
    const char* bufB = smem + (t & 1) * Cfg::BUF + 2 * SLOT_A;
    const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
    hx8 fb[NT][2];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      if constexpr (NPH == 4) {
        if (ph == 0 && more1) stageA(t + 1, 0);
        if (ph == 1 && more1) stageA(t + 1, 1);
        if (ph == 2 && more2) stageB(t + 2, 0, B_SPLIT);
        if (ph == 3 && more2) stageB(t + 2, B_SPLIT, Cfg::NB);
      } else {
        if (ph == 0 && more1) { stageA(t + 1, 0); stageA(t + 1, 1); }
        if (ph == 1 && more2) stageB(t + 2, 0, Cfg::NB);
      }
      if (ph == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int r = wn * 16 * NT + j * 16 + (lane & 15);
            fb[j][ks] = *reinterpret_cast<const hx8*>(bufB + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
          }
      }
      hx8 fa[IT][2];
#pragma unroll
      for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int r = ph * (128 / NPH) + i * 16 + (lane & 15);
          fa[i][ks] = *reinterpret_cast<const hx8*>(bufA + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
        }
      if constexpr (STAGGER) {
        // RAW: every wave retires its tile-(t+1) DMAs BEFORE the barrier that precedes the leading group's first read
        if (ph == NPH - 1) {
          if (more2) wait_vmcnt<Cfg::NB>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[ph * IT + i][j] = UCOD_MFMA16(fa[i][ks], fb[j][ks], acc[ph * IT + i][j]);
      __builtin_amdgcn_s_setprio(0);
      if constexpr (!STAGGER) {
        if (ph == NPH - 1) {
          if (more2) wait_vmcnt<Cfg::NB>(); else wait_vmcnt<0>();
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (STAGGER && wm == 0) __builtin_amdgcn_s_barrier();

  // epilogue through a wave-private LDS region (operand tiles are dead: last barrier passed)
  big_epilogue<EPI, NT>(a, acc, cs, smem + wave * (32 * 16 * NT * 4), m0 + wm * 128, n0 + wn * 16 * NT, lane);
}

// =====================================================================================================
// Mixed-height launch of the large-tile kernel: whole rounds instead of a nearly empty last one.
// One large-tile workgroup fills a CU, so a launch runs in rounds of n_cu tiles, and 32 x 1370 rows put the backbone's shapes just
// past a whole number of rounds: QKV 172 x 9 = 1548 tiles = 6.05 rounds, fc1 172 x 12 = 2064 = 8.06 -- the 7th / 9th round runs on 12
// / 16 CUs.  Here the M axis is cut into tm' = floor(rounds * n_cu / tiles_n) row-tiles instead: n_tall of them 288 rows high (nine
// 16-row MFMA tiles per wave group instead of eight), the others 256, tall ones spread evenly over the grid (every `stride`-th
// row-tile) so that each XCD gets its share.  QKV: 160 x 256 + 10 x 288 rows = 43 840, 170 x 9 = 1530 tiles <= 6 x 256: six rounds, 90 of
// the tiles 12.5 % longer.  The tall body is a second instantiation of the same loop (three barrier intervals of 24 MFMAs per K-tile
// and group instead of two of 32, so that its A fragments fit the register budget), selected by one wave-uniform branch at the top.
// =====================================================================================================
template <int NT, int XT>
struct MixCfg {
  static constexpr int RG = 128 + 16 * XT;            // rows per wave group
  static constexpr int NI = 8 + XT;                   // 16-row MFMA tiles per wave group
  static constexpr int NPH = XT ? 3 : 2;              // barrier intervals per K-tile and group
  static constexpr int IT = NI / NPH;
  static constexpr int SLOT = RG * 128;               // bytes of one group's A slot (64 bf16 per row)
  static constexpr int BN_ = 64 * NT, NB = BN_ / 64;
  static constexpr int BUF = 2 * SLOT + BN_ * 128;
  static_assert(NI % NPH == 0, "whole phases");
};

template <int EPI, int NT, int XT, int AUX>
__device__ __forceinline__ void mixed_body(const GemmArgs& a, char* smem, int m0, int n0, int wave, int lane) {
  using Cfg = MixCfg<NT, XT>;
  constexpr int NPH = Cfg::NPH, IT = Cfg::IT, NI = Cfg::NI;
  const int wm = wave >> 2, wn = wave & 3;
  const int K = a.K, nt = K / BK;
  // per-thread LDS-DMA sources as 32-bit byte offsets into buffer descriptors over A and B (two instructions cover 128 rows of a
  // group's slot, a third -- waves 0 and 1 only -- the 16 extra rows; rows past the end of the matrix fail the range check and arrive
  // as zeros; the K-tile offset rides in the scalar offset).  Half the address registers of per-lane 64-bit pointers: the kernel has
  // to stay at 224 VGPRs so that 64 per SIMD lane remain for the small kernels of a concurrent stream (decoder step, LayerNorm).
  const unsigned long bytesA = (unsigned long)a.M * K * 2ul, bytesB = (unsigned long)a.N * K * 2ul;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(a.A), 0, bytesA > 0xFFFFFFFFul ? 0xFFFFFFFFu : (unsigned)bytesA, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(a.B), 0, bytesB > 0xFFFFFFFFul ? 0xFFFFFFFFu : (unsigned)bytesB, 0x00020000);
  unsigned offA[2][2 + XT], offB[Cfg::NB];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2 + XT; ++i) {
      const int r = (i * 8 + wave) * 8 + (lane >> 3);
      offA[h][i] = ((unsigned)(m0 + h * Cfg::RG + r) * (unsigned)K + (unsigned)swz(r, lane & 7) * 8u) * 2u;
    }
#pragma unroll
  for (int i = 0; i < Cfg::NB; ++i) {
    const int r = (i * 8 + wave) * 8 + (lane >> 3);
    offB[i] = ((unsigned)(n0 + r) * (unsigned)K + (unsigned)swz(r, lane & 7) * 8u) * 2u;
  }
  auto stageA = [&](int t, int h) {
    char* slot = smem + (t & 1) * Cfg::BUF + h * Cfg::SLOT;
    const unsigned kt = (unsigned)t * (BK * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(slot + (i * 8 + wave) * 1024), 16, offA[h][i], kt, 0, UCOD_LD_AUX_A);
    if constexpr (XT == 1) {
      if (wave < 2)                                          // rows 128..143 (wave-uniform: `wave` is an SGPR)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(slot + (16 + wave) * 1024), 16, offA[h][2], kt, 0, UCOD_LD_AUX_A);
    }
  };
  auto stageB = [&](int t) {
    char* slot = smem + (t & 1) * Cfg::BUF + 2 * Cfg::SLOT;
    const unsigned kt = (unsigned)t * (BK * 2);
#pragma unroll
    for (int i = 0; i < Cfg::NB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(slot + (i * 8 + wave) * 1024), 16, offB[i], kt, 0, UCOD_LD_AUX_B);
  };

  float cb[NT], cs[NT];
  load_col_consts<EPI, NT>(a, n0 + wn * 16 * NT + (lane & 15), cb, cs);
  stageA(0, 0);
  stageA(0, 1);
  stageB(0);
  if (nt > 1) stageB(1);
  if (nt > 1) wait_vmcnt<Cfg::NB>(); else wait_vmcnt<0>();
  finish_col_consts<EPI, NT>(a, cb, cs);
  f32x4 acc[NI][NT];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){cb[j], cb[j], cb[j], cb[j]};
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();            // staggered wave groups (see gemm_bf16_big_kernel)

  for (int t = 0; t < nt; ++t) {
    const char* bufA = smem + (t & 1) * Cfg::BUF + wm * Cfg::SLOT;
    const char* bufB = smem + (t & 1) * Cfg::BUF + 2 * Cfg::SLOT;
    const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
    hx8 fb[NT][2];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      if constexpr (NPH == 2) {
        if (ph == 0 && more1) { stageA(t + 1, 0); stageA(t + 1, 1); }
        if (ph == 1 && more2) stageB(t + 2);
      } else {
        if (ph == 0 && more1) stageA(t + 1, 0);
        if (ph == 1 && more1) stageA(t + 1, 1);
        if (ph == 2 && more2) stageB(t + 2);
      }
      if (ph == 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int r = wn * 16 * NT + j * 16 + (lane & 15);
            fb[j][ks] = *reinterpret_cast<const hx8*>(bufB + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
          }
      }
      hx8 fa[IT][2];
#pragma unroll
      for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int r = ph * (IT * 16) + i * 16 + (lane & 15);
          fa[i][ks] = *reinterpret_cast<const hx8*>(bufA + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
        }
      if (ph == NPH - 1) {                               // RAW: every wave retires its tile-(t+1) DMAs before the barrier ahead of the first read
        if (more2) wait_vmcnt<Cfg::NB>(); else wait_vmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[ph * IT + i][j] = UCOD_MFMA16(fa[i][ks], fb[j][ks], acc[ph * IT + i][j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  big_epilogue<EPI, NT, NI, AUX>(a, acc, cs, smem + wave * (32 * 16 * NT * 4), m0 + wm * Cfg::RG, n0 + wn * 16 * NT, lane);
}

// first row of row-tile tm when every `stride`-th row-tile (n_tall of them in all) is 32 rows taller
__device__ __forceinline__ int mixed_row0(int tm, int n_tall, int stride, bool& tall) {
  const int before = tm / stride + (tm % stride ? 1 : 0);       // tall row-tiles among 0 .. tm-1 are 0, stride, 2*stride, ...
  const int nb = before < n_tall ? before : n_tall;
  tall = (tm % stride) == 0 && (tm / stride) < n_tall;
  return tm * 256 + nb * 32;
}

template <int EPI, int NT, int AUX = 0>
__global__ __launch_bounds__(512) void gemm_bf16_mixed_kernel(const GemmArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * MixCfg<NT, 1>::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = a.tiles_m * a.tiles_n;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
  int tm, tn;
  tile_of(a, wg, tm, tn);
  bool tall;
  const int m0 = mixed_row0(tm, a.main_tiles /* n_tall */, a.patches_per_wg /* stride */, tall);
  const int n0 = tn * MixCfg<NT, 0>::BN_;
  if (tall) mixed_body<EPI, NT, 1, AUX>(a, smem, m0, n0, wave, lane);
  else mixed_body<EPI, NT, 0, AUX>(a, smem, m0, n0, wave, lane);
}

// =====================================================================================================
// Persistent form of the large-tile kernel: one workgroup per CU walks tiles vt = blockIdx.x, +gridDim.x, ...
// What it buys: the first K-tile of the NEXT output tile (A0|A1|B, 9-11 LDS-DMAs per thread) is issued BEFORE the epilogue of
// the current tile, into the K-tile buffer the main loop has just vacated, so the ~3 us of first-tile HBM/L2 latency that every
// tile of the one-shot kernel pays up front (13 % of a K=768 tile) hides under the epilogue's stores.  The epilogue stages
// through the OTHER buffer (4 passes of 32 rows, 8 KB per wave) so the two never touch the same LDS bytes.
// Hazards on top of the one-shot kernel's:
//   * next-tile DMAs target buffer free_buf = (last K-tile's buffer)^1, last read during K-tile nt-2: dead long before;
//   * epilogue staging lives in last_buf, whose operand reads all retired before the stagger-out barrier;
//   * after the epilogue: every wave `vmcnt(0)` (its DMAs landed; also its stores) -> barrier -> only then may B(1) of the next
//     tile be DMA'd into last_buf (it overlaps other waves' staging areas) and the next main loop read free_buf.
// =====================================================================================================
template <int EPI, int NT>
__global__ __launch_bounds__(512) void gemm_bf16_pers_kernel(const GemmArgs a) {
  using Cfg = BigCfg<NT>;
  __shared__ __attribute__((aligned(16))) char smem[2 * Cfg::BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int ntiles = a.tiles_m * a.tiles_n;
  const int K = a.K, nt = K / BK;
  constexpr int WCOLS = 16 * NT;

  auto decode = [&](int vt, int& m0, int& n0) {
    const int q = ntiles >> 3, r8 = ntiles & 7, xcd = vt & 7;
    const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (vt >> 3);
    int tm, tn;
    tile_of(a, wg, tm, tn);
    m0 = tm * 256;
    n0 = tn * Cfg::BN_;
  };
  // DMA source rows as 32-bit element offsets from the tile's first A / B row (64-bit per-tile bases stay in SGPRs): the
  // persistent kernel keeps the next tile's sources live across the epilogue, and 64-bit pointers there spilled VGPRs.
  unsigned srcA[2][2], srcB[Cfg::NB];
  const bf16_raw *baseA, *baseB;
  auto set_src = [&](int m0, int n0) {
    baseA = a.A + (size_t)m0 * K;
    baseB = a.B + (size_t)n0 * K;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        int lr = h * 128 + r;
        lr = (m0 + lr) < a.M ? lr : a.M - 1 - m0;
        srcA[h][i] = (unsigned)lr * (unsigned)K + swz(r, lane & 7) * 8;
      }
#pragma unroll
    for (int i = 0; i < Cfg::NB; ++i) {
      const int r = (i * 8 + wave) * 8 + (lane >> 3);
      int lr = (n0 + r) < a.N ? r : a.N - 1 - n0;
      srcB[i] = (unsigned)lr * (unsigned)K + swz(r, lane & 7) * 8;
    }
  };
  auto dma = [&](const bf16_raw* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto stageA = [&](int t, int h, int pb) {
    char* slot = smem + ((t + pb) & 1) * Cfg::BUF + h * SLOT_A;
#pragma unroll
    for (int i = 0; i < 2; ++i) dma(baseA + t * BK + srcA[h][i], slot + (i * 8 + wave) * 1024);
  };
  auto stageB = [&](int t, int i0, int i1, int pb) {
    char* slot = smem + ((t + pb) & 1) * Cfg::BUF + 2 * SLOT_A;
#pragma unroll
    for (int i = 0; i < Cfg::NB; ++i)
      if (i >= i0 && i < i1) dma(baseB + t * BK + srcB[i], slot + (i * 8 + wave) * 1024);
  };
  constexpr int B_SPLIT = Cfg::NB >= 2 ? 2 : 1;

#ifdef UCOD_GEMM_STAMPS
  unsigned long long st_loop = 0, st_epi = 0, st_wait = 0, st_tiles = 0, t0s, t1s, t2s, t3s;
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#else
#define STAMP(v)
#endif
  int vt = blockIdx.x, pb = 0, m0, n0;
  decode(vt, m0, n0);
  float cb[NT], cs[NT];
  load_col_consts<EPI, NT>(a, n0 + wn * WCOLS + (lane & 15), cb, cs);
  set_src(m0, n0);
  stageA(0, 0, pb);
  stageA(0, 1, pb);
  stageB(0, 0, Cfg::NB, pb);
  if (nt > 1) {
    stageB(1, 0, Cfg::NB, pb);
    wait_vmcnt<Cfg::NB>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();

  while (true) {
    finish_col_consts<EPI, NT>(a, cb, cs);
    f32x4 acc[8][NT];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){cb[j], cb[j], cb[j], cb[j]};
    STAMP(t0s);
    if (wm == 1) __builtin_amdgcn_s_barrier();               // stagger in (see the one-shot kernel)

    for (int t = 0; t < nt; ++t) {
      const char* bufA = smem + ((t + pb) & 1) * Cfg::BUF + wm * SLOT_A;
      const char* bufB = smem + ((t + pb) & 1) * Cfg::BUF + 2 * SLOT_A;
      const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
      hx8 fb[NT][2];
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        if (ph == 0 && more1) stageA(t + 1, 0, pb);
        if (ph == 1 && more1) stageA(t + 1, 1, pb);
        if (ph == 2 && more2) stageB(t + 2, 0, B_SPLIT, pb);
        if (ph == 3 && more2) stageB(t + 2, B_SPLIT, Cfg::NB, pb);
        if (ph == 0) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const int r = wn * 16 * NT + j * 16 + (lane & 15);
              fb[j][ks] = *reinterpret_cast<const hx8*>(bufB + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
            }
        }
        hx8 fa[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int r = ph * 32 + i * 16 + (lane & 15);
            fa[i][ks] = *reinterpret_cast<const hx8*>(bufA + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
          }
        if (ph == 3) {
          if (more2) wait_vmcnt<Cfg::NB>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[ph * 2 + i][j] = UCOD_MFMA16(fa[i][ks], fb[j][ks], acc[ph * 2 + i][j]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();               // stagger out: every wave is past its last LDS operand read
    STAMP(t1s);

    const int last_buf = (nt - 1 + pb) & 1, free_buf = last_buf ^ 1;
    const int vnext = vt + gridDim.x;
    const bool has_next = vnext < ntiles;
    const int em0 = m0, en0 = n0;
    if (has_next) {                                           // first K-tile of the next tile, in flight under the epilogue
      decode(vnext, m0, n0);
      set_src(m0, n0);
      stageA(0, 0, free_buf);
      stageA(0, 1, free_buf);
      stageB(0, 0, Cfg::NB, free_buf);
    }
    big_epilogue<EPI, NT>(a, acc, cs, smem + last_buf * Cfg::BUF + wave * (32 * WCOLS * 4), em0 + wm * 128, en0 + wn * WCOLS, lane);
    if (has_next) load_col_consts<EPI, NT>(a, n0 + wn * WCOLS + (lane & 15), cb, cs);   // retired by the vmcnt(0) below, with the stores
    STAMP(t2s);
#ifdef UCOD_GEMM_STAMPS
    wait_vmcnt<0>();
    STAMP(t3s);
    st_loop += t1s - t0s; st_epi += t2s - t1s; st_wait += t3s - t2s; st_tiles += 1;
    if (!has_next) {
      if (tid == 0 && a.stamps) { a.stamps[blockIdx.x * 4 + 0] = st_loop; a.stamps[blockIdx.x * 4 + 1] = st_epi; a.stamps[blockIdx.x * 4 + 2] = st_wait; a.stamps[blockIdx.x * 4 + 3] = st_tiles; }
      break;
    }
#else
    if (!has_next) break;
    wait_vmcnt<0>();
#endif
    __builtin_amdgcn_s_barrier();
    vt = vnext;
    pb = free_buf;
    if (nt > 1) stageB(1, 0, Cfg::NB, pb);
  }
}

// variant: 0 auto | 1 128^2 register staging | 2 128^2 LDS-DMA | 3 256x256 | 4 256x192 | 5,6 = 3,4 with staggered wave groups
// Large-tile launch plan for a tile width: whole rounds of n_cu tiles, and whether the tiles past the last whole round are few
// enough to be computed as patches on the side (patch_phase) instead of as a nearly empty extra round.
struct BigPlan {
  int total, rounds, left, ppt;
  bool patches;
  double cost;        // makespan model, fitted to tools/gemm_bench.py on MI355X: a tile costs a fixed part (A-panel DMA, prologue,
};                    // epilogue set-up) plus a part proportional to its width; a patch ~2 % of a tile per round
static int device_cus() {
  static const int n_cu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
  return n_cu;
}
static BigPlan big_plan(int M, int N, int K, int bn, bool patch_epi) {
  // UCOD_GEMM_NO_PATCH=1 (read per call): every output through the tile path, whose f32 sum over K has one fixed order -- results
  // are then bitwise independent of where a row sits in the batch; a patch sums K in 8 interleaved partials
  const char* no_patch = getenv("UCOD_GEMM_NO_PATCH");
  const bool off = no_patch && no_patch[0] != '0';
  const char* mr = getenv("UCOD_GEMM_PATCH_ROUNDS");
  const int max_rounds = mr ? atoi(mr) : 2;             // 3-4 rounds measured: no gain alone (ViT-L QKV), -2 % in the two-stream step (the other stream fills those tails)
  const int n_cu = device_cus();
  BigPlan p;
  p.total = cdiv(M, 256) * cdiv(N, bn);
  p.rounds = p.total / n_cu;
  p.left = p.total - p.rounds * n_cu;
  p.ppt = 16 * (bn / 32);
  // (more rounds dilute the tail below what a patch costs every workgroup -- QKV / fc1 of ViT-B: 6 and 8 rounds -- the cost model decides)
  p.patches = patch_epi && !off && p.rounds >= 1 && p.rounds <= max_rounds && p.left > 0 && (long)p.left * p.ppt <= 2L * p.rounds * n_cu && (K & 31) == 0;
  // Makespan in tile units.  A last, partly filled round is cheaper than a full one (its tiles run on an otherwise idle chip: measured
  // 0.42 of a round at 1.6 % fill, fc2 2 rounds 183 us -> 2.016 rounds 221 us): 0.4 + 0.6 * fill.  A patch costs every workgroup ~8 % of
  // its tile (3.4 us of 41 at K = 768, 7.4 of 91 at K = 3072).  With these two numbers the model reproduces the measured choices: patches
  // for ViT-B proj / fc2 (553 vs 617 units = the measured 198 vs 221 us), plain 256-wide tiles for ViT-L's N = 1024 (1.34 rounds).
  const double tile = 0.45 * 256 + 0.55 * bn;
  const double fill = (double)p.left / n_cu;
  const double plain = (p.rounds + (p.left ? 0.4 + 0.6 * fill : 0.0)) * tile;
  const double patched = p.rounds * tile * 1.08;
  if (p.patches && patched >= plain) p.patches = false;
  p.cost = p.patches ? patched : plain;
  return p;
}

// Mixed-height plan (see gemm_bf16_mixed_kernel): row-tiles, how many of them tall, and their spacing; feasible = false when the shape
// already fills whole rounds or when 32 extra rows on every row-tile would not be enough.
struct MixedPlan { bool feasible; int tiles_m, n_tall, stride, rounds; };
static MixedPlan mixed_plan(int M, int N, int bn) {
  MixedPlan p{false, 0, 0, 1, 0};
  const int n_cu = device_cus(), tiles_n = cdiv(N, bn), t0 = cdiv(M, 256) * tiles_n;
  const int rounds = t0 / n_cu;
  if (rounds < 1 || t0 == rounds * n_cu) return p;
  const int tm = (rounds * n_cu) / tiles_n;                     // row-tiles that fit `rounds` whole rounds
  const long short_rows = (long)M - 256L * tm;
  if (tm < 1 || short_rows <= 0) return p;
  const int n_tall = (int)cdiv(short_rows, 32L);
  if (n_tall > tm) return p;
  p.feasible = true;
  p.tiles_m = tm;
  p.n_tall = n_tall;
  p.stride = tm / n_tall;
  p.rounds = rounds;
  return p;
}

template <int EPI>
static int launch(GemmArgs a, int variant, hipStream_t s) {
  constexpr bool kTrainEpi = (EPI == UCOD_EPI_GELU_BWD_BF16 || EPI == UCOD_EPI_BIAS_GELU_SAVE_BF16);
  constexpr bool kPatchEpi = !kTrainEpi;
  const bool auto_small = variant == 0;                       // only `auto` may pick the 64 x 64 tile by itself
  if (variant == 0) {
    variant = 2;
    // large tiles when either dimension is long enough to fill the chip with 256-row tiles (the key hook has M = channels = 768
    // but N = all tokens: 3 x 172 tiles)
    const bool big_enough = a.M >= 2048 || (a.M >= 512 && (long)a.M * a.N >= (1L << 24));
    if (big_enough && a.K >= 128 && (a.N & 3) == 0 && (!(EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16) || (a.N & 7) == 0)) {
      // two 32-MFMA barrier intervals per K-tile (variants 9/10) beat four 16-MFMA ones (5/6) by 2-4 % and the persistent
      // form (7/8) by 1-5 % on every backbone shape (tools/gemm_bench.py); the width with the shorter modelled makespan
      variant = (big_plan(a.M, a.N, a.K, 192, kPatchEpi).cost < big_plan(a.M, a.N, a.K, 256, kPatchEpi).cost) ? 10 : 9;
      // three or more rounds with a nearly empty last one (QKV 6.05, fc1 8.06): mixed-height tiles make it whole rounds (-7 % / -8 %,
      // tools/gemm_order_sweep.py); at one or two rounds the patches above already do that at the same cost
      const MixedPlan mp = mixed_plan(a.M, a.N, 256);
      if (kColFused<EPI> && mp.feasible && mp.rounds >= 3 && !getenv("UCOD_GEMM_NO_MIXED")) variant = 13;
    }
  }
  if (kTrainEpi || ((EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_F32) && !a.bias)) {   // large-tile kernels only
    if (kTrainEpi && ((a.N & 7) != 0 || a.K < 128)) return UCOD_EINVAL;
    if (variant < 3) variant = (big_plan(a.M, a.N, a.K, 192, kPatchEpi).cost < big_plan(a.M, a.N, a.K, 256, kPatchEpi).cost) ? 10 : 9;
  }
  constexpr bool kBf16Out = (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16 || kTrainEpi);
  if (variant >= 3 && variant <= 10 && ((a.N & 3) != 0 || (kBf16Out && (a.N & 7) != 0))) return UCOD_EINVAL;   // 16-byte row stores
  if constexpr (kColFused<EPI>) {
    if (variant == 13 || variant == 14) {                     // mixed-height tiles; falls back to 9 / 10 when the plan is not feasible
      const MixedPlan mp = mixed_plan(a.M, a.N, variant == 13 ? 256 : 192);
      const bool fits32 = (long)a.M * a.K * 2 < (1L << 32) && (long)a.N * a.K * 2 < (1L << 32);   // the kernel addresses its operands with 32-bit byte offsets
      if (!mp.feasible || !fits32 || (a.N & 3) != 0 || (kBf16Out && (a.N & 7) != 0) || a.K < 128) {
        variant -= 4;
      } else {
        a.tiles_m = mp.tiles_m;
        a.tiles_n = cdiv(a.N, variant == 13 ? 256 : 192);
        a.main_tiles = mp.n_tall;                               // (the two fields are free in this mode: no patches)
        a.patches_per_wg = mp.stride;
        a.col_fast = a.tiles_n <= 4;
        if (const char* e = getenv("UCOD_GEMM_GROUP_M")) a.group_m = atoi(e) > 0 ? atoi(e) : a.tiles_m;
        if (const char* e = getenv("UCOD_GEMM_COL_FAST")) a.col_fast = atoi(e);
        // bf16 outputs of the two forward epilogues (qkv, MLP hidden) leave with the non-temporal policy: they are read once, by the
        // next kernel, and displace less of what the running launch re-reads (in the step: QKV 153.5 -> 145.8 us, fc1 223.7 -> 216.7)
        int aux = 2;
        if (const char* e = getenv("UCOD_GEMM_ST_AUX")) aux = atoi(e);
        dim3 grid(a.tiles_m * a.tiles_n), block(512);
        if constexpr (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16) {   // store-policy experiment builds exist for the two bf16 forward epilogues
          if (aux == 2 && variant == 13) { hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 4, 2>), grid, block, 0, s, a); UCOD_CHECK_LAUNCH(); return UCOD_OK; }
          if (aux == 16 && variant == 13) { hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 4, 16>), grid, block, 0, s, a); UCOD_CHECK_LAUNCH(); return UCOD_OK; }
        }
        if (variant == 13) hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 4>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 3>), grid, block, 0, s, a);
        UCOD_CHECK_LAUNCH();
        return UCOD_OK;
      }
    }
  } else {
    if (variant == 13 || variant == 14) variant -= 4;
  }
  if (variant >= 3 && variant <= 10) {
    const bool wide = (variant == 3 || variant == 5 || variant == 7 || variant == 9);
    a.tiles_m = cdiv(a.M, 256);
    a.tiles_n = cdiv(a.N, wide ? 256 : 192);
    // few column tiles (proj / fc2: N = 768): sweep one A panel's columns back to back (proj 75 -> 70 us, L2 fetch 263 -> 230 MB);
    // many (QKV 9, fc1 12): row-tile fastest in groups of 8 (fc1 is 3-6 % slower column-fastest: its weight matrix alone exceeds the L2)
    a.col_fast = a.tiles_n <= 4;
    if (const char* e = getenv("UCOD_GEMM_GROUP_M")) a.group_m = atoi(e) > 0 ? atoi(e) : a.tiles_m;   // tuning knobs (tools/gemm_order_sweep.py)
    if (const char* e = getenv("UCOD_GEMM_COL_FAST")) a.col_fast = atoi(e);
    dim3 grid(a.tiles_m * a.tiles_n), block(512);
    if (variant != 7 && variant != 8) {
      const BigPlan pl = big_plan(a.M, a.N, a.K, wide ? 256 : 192, kPatchEpi);
      if (pl.patches) {                                        // leftover-as-patches: exactly rounds x n_cu workgroups
        a.main_tiles = pl.rounds * device_cus();
        a.patches_per_wg = cdiv((long)pl.left * pl.ppt, a.main_tiles);
        grid.x = a.main_tiles;
      }
    }
    switch (variant) {
      case 3: hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, 4, false>), grid, block, 0, s, a); break;
      case 4: hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, 3, false>), grid, block, 0, s, a); break;
      case 5: hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, 4, true>), grid, block, 0, s, a); break;
      case 6: hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, 3, true>), grid, block, 0, s, a); break;
      case 9: hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, 4, true, 2>), grid, block, 0, s, a); break;
      case 10: hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, 3, true, 2>), grid, block, 0, s, a); break;
      default: {                                             // 7, 8: persistent, one workgroup per CU
        const int n_cu = device_cus();
        const int ntiles = a.tiles_m * a.tiles_n;
        dim3 pgrid(ntiles < n_cu ? ntiles : n_cu);
#ifdef UCOD_GEMM_STAMPS
        if (const char* g = getenv("UCOD_PERS_GRID")) pgrid.x = atoi(g) < ntiles ? atoi(g) : ntiles;   // diagnostic: fewer active CUs
#endif
        if (variant == 7) hipLaunchKernelGGL((gemm_bf16_pers_kernel<EPI, 4>), pgrid, block, 0, s, a);
        else hipLaunchKernelGGL((gemm_bf16_pers_kernel<EPI, 3>), pgrid, block, 0, s, a);
      }
    }
  } else {
    // 128 x 128 tiles (two workgroups per CU), or 64 x 64 when there are fewer 128-tiles than CUs: a batch-1 backbone pass has 66 tiles of
    // proj / fc2 (29 -> 19 us per launch with the small tile; 264 tiles of fc1 are already better off with 128 x 128).  Variant 12 forces
    // the small tile, 1 / 2 the large one.
    const int t128 = a.tiles_m * a.tiles_n;
    const bool small = variant == 12 || (variant == 2 && auto_small && t128 < device_cus());
    dim3 block(256);
    if (small) {
      a.tiles_m = cdiv(a.M, 64);
      a.tiles_n = cdiv(a.N, 64);
      hipLaunchKernelGGL((gemm_bf16_kernel<EPI, true, 64>), dim3(a.tiles_m * a.tiles_n), block, 0, s, a);
    } else if (variant == 1) {
      hipLaunchKernelGGL((gemm_bf16_kernel<EPI, false>), dim3(t128), block, 0, s, a);
    } else {
      hipLaunchKernelGGL((gemm_bf16_kernel<EPI, true>), dim3(t128), block, 0, s, a);
    }
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

}  // namespace ucod

// QKV projection with e4m3 output (UCOD_EPI_QKV_FP8): always the mixed-height large-tile kernel, 256 wide (a wave's 64 columns are one
// head), with tall tiles where that makes whole rounds and without them otherwise.
static int launch_qkv_fp8(ucod::GemmArgs a, hipStream_t s) {
  using namespace ucod;
  if (a.N % 192 != 0 || a.K < 128 || a.tok < 1 || a.M % a.tok != 0 || !a.bias) return UCOD_EINVAL;
  if ((long)a.M * a.K * 2 >= (1L << 32) || (long)a.N * a.K * 2 >= (1L << 32)) return UCOD_EINVAL;      // 32-bit operand offsets
  const MixedPlan mp = mixed_plan(a.M, a.N, 256);
  a.tiles_n = cdiv(a.N, 256);
  if (mp.feasible) {
    a.tiles_m = mp.tiles_m;
    a.main_tiles = mp.n_tall;
    a.patches_per_wg = mp.stride;
  } else {
    a.tiles_m = cdiv(a.M, 256);
    a.main_tiles = 0;                                           // no tall row-tiles
    a.patches_per_wg = 1 << 30;
  }
  a.col_fast = a.tiles_n <= 4;
  hipLaunchKernelGGL((gemm_bf16_mixed_kernel<UCOD_EPI_QKV_FP8, 4>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, s, a);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

// Out-projection / fc2 with the f16 residual stream (UCOD_EPI_BIAS_SCALE_RESID_H16): the mixed-height large-tile kernel, 256 wide.
static int launch_resid_h16(ucod::GemmArgs a, hipStream_t s) {
  using namespace ucod;
  if ((a.N & 7) != 0 || a.K < 128 || !a.bias || !a.scale || !a.resid) return UCOD_EINVAL;
  if ((long)a.M * a.K * 2 >= (1L << 32) || (long)a.N * a.K * 2 >= (1L << 32)) return UCOD_EINVAL;      // 32-bit operand offsets
  const MixedPlan mp = mixed_plan(a.M, a.N, 256);
  a.tiles_n = cdiv(a.N, 256);
  if (mp.feasible) {
    a.tiles_m = mp.tiles_m;
    a.main_tiles = mp.n_tall;
    a.patches_per_wg = mp.stride;
  } else {
    a.tiles_m = cdiv(a.M, 256);
    a.main_tiles = 0;
    a.patches_per_wg = 1 << 30;
  }
  a.col_fast = a.tiles_n <= 4;
  if (const char* e = getenv("UCOD_GEMM_GROUP_M")) a.group_m = atoi(e) > 0 ? atoi(e) : a.tiles_m;
  if (const char* e = getenv("UCOD_GEMM_COL_FAST")) a.col_fast = atoi(e);
  hipLaunchKernelGGL((gemm_bf16_mixed_kernel<UCOD_EPI_BIAS_SCALE_RESID_H16, 4>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, s, a);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

static int gemm_entry(int epilogue, const void* A, const void* B, void* out, int M, int N, int K, const float* bias,
                      const float* scale, const float* resid, const float* pos, int tokens_per_image, int variant,
                      void* stream, const void* aux, void* out2) {
  using namespace ucod;
  if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0) return UCOD_EINVAL;
  GemmArgs a;
  a.aux = aux;
  a.out2 = out2;
  a.stamps = nullptr;
#ifdef UCOD_GEMM_STAMPS
  a.stamps = (unsigned long long*)pos;   // diagnostic build: the (otherwise unused here) `pos` argument carries the stamp buffer
#endif
  a.A = (const bf16_raw*)A;
  a.B = (const bf16_raw*)B;
  a.out = out;
  a.bias = bias;
  a.scale = scale;
  a.resid = resid;
  a.pos = pos;
  a.M = M;
  a.N = N;
  a.K = K;
  a.tok = tokens_per_image;
  a.tiles_m = cdiv(M, BM);
  a.tiles_n = cdiv(N, BN);
  a.main_tiles = 0;
  a.patches_per_wg = 0;
  a.group_m = 8;
  a.col_fast = 0;
  hipStream_t s = (hipStream_t)stream;
  UCOD_PROF(epilogue == UCOD_EPI_QKV_FP8 ? 0 : epilogue == UCOD_EPI_BIAS_SCALE_RESID_H16 ? 2 : epilogue == UCOD_EPI_PATCH_TOKENS_H16 ? 3 : (epilogue >= 0 && epilogue <= 5 ? epilogue : (epilogue == UCOD_EPI_GELU_BWD_BF16 ? PROF_GEMM_EPI6 : PROF_GEMM_EPI7)), s);
  switch (epilogue) {
    case UCOD_EPI_BIAS_BF16:                                   // NULL bias (plain product) only in the large-tile kernels
      if (!bias && (variant == 1 || variant == 2 || K < 128 || (N & 3))) return UCOD_EINVAL;
      return launch<UCOD_EPI_BIAS_BF16>(a, variant, s);
    case UCOD_EPI_GELU_BWD_BF16: if (!aux) return UCOD_EINVAL; return launch<UCOD_EPI_GELU_BWD_BF16>(a, variant, s);
    case UCOD_EPI_BIAS_GELU_SAVE_BF16: if (!bias || !out2) return UCOD_EINVAL; return launch<UCOD_EPI_BIAS_GELU_SAVE_BF16>(a, variant, s);
    case UCOD_EPI_BIAS_GELU_BF16: if (!bias) return UCOD_EINVAL; return launch<UCOD_EPI_BIAS_GELU_BF16>(a, variant, s);
    case UCOD_EPI_BIAS_SCALE_RESID_F32:
      if (!bias || !scale || !resid) return UCOD_EINVAL;
      return launch<UCOD_EPI_BIAS_SCALE_RESID_F32>(a, variant, s);
    case UCOD_EPI_PATCH_TOKENS_F32:
      if (!bias || !pos || tokens_per_image < 2) return UCOD_EINVAL;
      return launch<UCOD_EPI_PATCH_TOKENS_F32>(a, variant, s);
    case UCOD_EPI_KEY_NCHW_F32:
      if (!bias || tokens_per_image < 2) return UCOD_EINVAL;
      return launch<UCOD_EPI_KEY_NCHW_F32>(a, variant, s);
    case UCOD_EPI_BIAS_F32:
      if (!bias && (variant == 1 || variant == 2 || K < 128 || (N & 3))) return UCOD_EINVAL;
      return launch<UCOD_EPI_BIAS_F32>(a, variant, s);
    case UCOD_EPI_QKV_FP8: return launch_qkv_fp8(a, s);
    case UCOD_EPI_BIAS_SCALE_RESID_H16: return launch_resid_h16(a, s);
    case UCOD_EPI_PATCH_TOKENS_H16:
      if (!bias || !pos || tokens_per_image < 2) return UCOD_EINVAL;
      return launch<UCOD_EPI_PATCH_TOKENS_H16>(a, variant, s);
    default: return UCOD_EINVAL;
  }
}

extern "C" int ucod_gemm_bf16(int epilogue, const void* A, const void* B, void* out, int M, int N, int K, const float* bias,
                              const float* scale, const float* resid, const float* pos, int tokens_per_image, int variant,
                              void* stream) {
  if (epilogue == UCOD_EPI_GELU_BWD_BF16 || epilogue == UCOD_EPI_BIAS_GELU_SAVE_BF16) return UCOD_EINVAL;   // need ucod_gemm_bf16_train
  return gemm_entry(epilogue, A, B, out, M, N, K, bias, scale, resid, pos, tokens_per_image, variant, stream, nullptr, nullptr);
}

extern "C" int ucod_gemm_bf16_train(int epilogue, const void* A, const void* B, void* out, int M, int N, int K, const float* bias,
                                    const void* aux_bf16, void* out2_bf16, int variant, void* stream) {
  UCOD_BF16_ONLY();
  return gemm_entry(epilogue, A, B, out, M, N, K, bias, nullptr, nullptr, nullptr, 0, variant, stream, aux_bf16, out2_bf16);
}
