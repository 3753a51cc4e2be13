// Shared device code of the bf16 MFMA GEMM (product: gemm_bf16.hip; experiment variants: variants/gemm_bf16_lab.hip):
// the argument block, the XCD-aware tile order, the LDS swizzle, exact-erf GELU, and the epilogues -- element / row-major drains of the
// 128 x 128 kernel and `big_epilogue` of the large-tile kernels (bias as the accumulator's initial value, column scale in the MFMA
// layout, drain through wave-private LDS into 16-byte buffer stores with no load between two stores, residual / saved pre-activation
// double-buffered across passes).  Reference arithmetic: transformers modeling_dinov2.py (Dinov2SelfOutput :238-253, Dinov2LayerScale
// :272-278, Dinov2MLP :281-297, Dinov2PatchEmbeddings :139-149), data/utils/feature_extractor.py:46-47,55-58 (key hook).
#pragma once
#include <cstdlib>
#include "common.h"
#include "../../include/ucod_dpl.h"

// Row pitch of the wave-private staging area.  UCOD_EPI_PAD (bytes, default 0) pads it: with 0 the four 16-lane groups of a ds_write_b32 of the stager land on
// the same 16 banks (rows 4 apart = 4 * 256 B); 16 shifts them by 16 banks each.  Measured (round 4, tools/gemm_ab.py): see DESIGN section 0.
#ifndef UCOD_EPI_PAD
#define UCOD_EPI_PAD 0
#endif
#define EPI_PITCH(wcols) ((wcols) * 4 + UCOD_EPI_PAD)

namespace ucod {

constexpr int BM = 128, BN = 128, BK = 64;

// LayerNorm folded into its consumer GEMM (round 5).  With x the fp16 residual stream as the A operand, W' = fp16(gamma (.) W), c[n] = sum_k W'[n][k],
// b'[n] = sum_k beta[k] W[n][k] + b[n]:   LN(x) W^T + b = rstd[m] * (x W'^T - mean[m] * c[n]) + b'[n]   (modeling_dinov2.py:348-381: norm1 -> attention,
// norm2 -> mlp).  The epilogue applies the two per-row scalars s = rstd, u = -mean * rstd:  out = s * acc + (u * c[n] + b'[n]).  The 16-bit
// rounding of the LayerNorm output and the LayerNorm launch itself disappear.
template <int EPI>
constexpr bool kFold = (EPI == UCOD_EPI_LNFOLD_BIAS_BF16 || EPI == UCOD_EPI_LNFOLD_GELU_BF16);
// The residual-stream producers that also leave row statistics for the next LayerNorm-folded consumer (round 5, step B): every wave adds up
// the 64 values of a row it has just rounded to fp16 -- sum and sum of squared deviations from the slot's own mean, of the ROUNDED values (what the consumer's MFMA will read) -- and
// stores the pair into slot (column / 64) of the row: no statistics launch, no atomics, one fixed order of additions.
template <int EPI>
constexpr bool kResidH16 = (EPI == UCOD_EPI_BIAS_SCALE_RESID_H16 || EPI == UCOD_EPI_BIAS_SCALE_RESID_H16_STATS);
template <int EPI>
constexpr bool kPatchH16 = (EPI == UCOD_EPI_PATCH_TOKENS_H16 || EPI == UCOD_EPI_PATCH_TOKENS_H16_STATS);
template <int EPI>
constexpr bool kStats = (EPI == UCOD_EPI_BIAS_SCALE_RESID_H16_STATS || EPI == UCOD_EPI_PATCH_TOKENS_H16_STATS);
template <int EPI>
constexpr bool kBiasLike = (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_LNFOLD_BIAS_BF16);        // optional column scale, 16-bit output
// fc1 of the two-term split-operand pass: GELU, then the result as the A-side split operand of fc2 -- three bf16 segments (hi | hi | lo) of a row 3 N wide
template <int EPI>
constexpr bool kSplit2Out = (EPI == UCOD_EPI_BIAS_GELU_SPLIT2);
template <int EPI>
constexpr int kOutPitchMul = kSplit2Out<EPI> ? 3 : 1;              // output row pitch in units of N elements
template <int EPI>
constexpr bool kGeluLike = (EPI == UCOD_EPI_BIAS_GELU_BF16 || EPI == UCOD_EPI_LNFOLD_GELU_BF16 || kSplit2Out<EPI>);   // erf-GELU, 16-bit output
// (hi, lo) bf16 terms of two f32 values, packed pairwise: v = hi + lo up to 2^-17 |v|
__device__ __forceinline__ u32x2 split2_pack(float a, float b) {   // -> (hi pair, lo pair)
  const unsigned hi = pack_bf16x2(a, b);
  return (u32x2){hi, pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u))};
}
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
  unsigned* ovf;      // f16 residual-stream epilogues: saturation counter (common.h: resid16_overflow_counter)
  const float* stats;   // LayerNorm-folded epilogues: per-row (rstd, -mean * rstd) of the A rows, f32 [M][2] (used when part_in is NULL)
  const float* colsum;  // LayerNorm-folded epilogues: c[n] = sum_k B[n][k] (of the ROUNDED folded weight), f32 [N]
  const float* part_in; // LayerNorm-folded epilogues: per-row partial (sum, M2 about the slot mean) of the A rows, f32 [M][nslot][2], written by the producer's
                        // *_STATS epilogue; the consumer's prologue sums them (large-tile kernels only)
  float* part_out;      // *_STATS epilogues: where this launch leaves its output rows' partial (sum, M2 about the slot mean), f32 [rows][nslot][2], slot = column / 64
  int nslot;            // partial slots per row of part_in / part_out
  float eps;            // LayerNorm eps (part_in)
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

// exact-erf GELU (transformers ACT2FN["gelu"], modeling_dinov2.py:289), two elements per call so the polynomial runs
// on v_pk_fma_f32.  With a = |x|:  0.5*erfc(a/sqrt2) = exp2(-(1 + a*(d1 + d2 a + d3 a^2 + d4 a^3 + d5 a^4)))  (weighted
// minimax fit of -log2 erfc, |erfc err| <= 5e-6 and RELATIVE in the tail), and  gelu(x) = max(x,0) - a * 0.5*erfc(a/sqrt2).
// Max |gelu err| = 7.1e-7 over [-30,30] in fp32 -- the same as the Abramowitz-Stegun 7.1.26 form it replaces, at one
// transcendental and ~9 issue slots per element instead of two and ~22 (the fc1 epilogue runs it 134 M times per
// launch and was VALU-bound: 6.3 k of its 18.1 k cycles per 256x256 tile).  d5 > 0, so large |x| underflows to t = 0.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
#ifdef UCOD_GELU_FMAXF
  const f32x2 a = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
#else
  // IEEE maximum (v_maximum3_f32): fmaxf would put a canonicalising v_max x, x, x in front of every max; |x| = 2 max(x, 0) - x (exact) is one
  // packed FMA per pair instead of two v_and.  11 instead of 14 vector instructions per pair: the fc1 epilogue is bound by exactly these
  // (profiles/r03_fc1_gelu_epilogue.txt)
  const f32x2 pos = {__builtin_elementwise_maximum(x[0], 0.f), __builtin_elementwise_maximum(x[1], 0.f)};
  const f32x2 a = pos * 2.0f - x;
#endif
  f32x2 p = a * 4.881021588e-04f + (-7.198718842e-03f);
  p = p * a + 5.214663086e-02f;
  p = p * a + 4.595958292e-01f;
  p = p * a + 1.151000509e+00f;
  p = p * a + 1.0f;
  const f32x2 t = {__builtin_amdgcn_exp2f(-p[0]), __builtin_amdgcn_exp2f(-p[1])};
#ifdef UCOD_GELU_FMAXF
  const f32x2 pos = {__builtin_fmaxf(x[0], 0.f), __builtin_fmaxf(x[1], 0.f)};
#endif
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
  if constexpr (kFold<EPI>) {
    const float s = a.stats[2 * (size_t)m], u = a.stats[2 * (size_t)m + 1];
    float o = fmaf(s, v, fmaf(u, a.colsum[n], a.bias[n]));
    if constexpr (EPI == UCOD_EPI_LNFOLD_BIAS_BF16) o *= (a.scale ? a.scale[n] : 1.f);
    else o = gelu_erf(o);
    reinterpret_cast<bf16_raw*>(a.out)[(size_t)m * a.N + n] = f32_to_h(o);
  } else if constexpr (EPI == UCOD_EPI_BIAS_BF16) {
    reinterpret_cast<bf16_raw*>(a.out)[(size_t)m * a.N + n] = f32_to_h((v + a.bias[n]) * (a.scale ? a.scale[n] : 1.f));
  } else if constexpr (kSplit2Out<EPI>) {
    const float o = gelu_erf(v + a.bias[n]);
    const bf16_raw hi = f32_to_bf16(o), lo = f32_to_bf16(o - bf16_to_f32(hi));
    bf16_raw* row = reinterpret_cast<bf16_raw*>(a.out) + (size_t)m * 3 * a.N + n;
    row[0] = hi;
    row[a.N] = hi;
    row[2 * (size_t)a.N] = lo;
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
  } else if constexpr (kPatchH16<EPI>) {
    const int np = a.tok - 1;
    const int b = m / np, p = m - b * np;
    const float o = v + a.bias[n] + a.pos[(size_t)(1 + p) * a.N + n];
    if (beyond_f16(o)) atomicAdd(a.ovf, 1u);
    reinterpret_cast<unsigned short*>(a.out)[((size_t)b * a.tok + 1 + p) * a.N + n] = __builtin_bit_cast(unsigned short, (_Float16)clamp_f16(o));
  } else if constexpr (kResidH16<EPI>) {
    const size_t i = (size_t)m * a.N + n;
    const float o = (float)reinterpret_cast<const _Float16*>(a.resid)[i] + a.scale[n] * (v + a.bias[n]);
    if (beyond_f16(o)) atomicAdd(a.ovf, 1u);
    reinterpret_cast<unsigned short*>(a.out)[i] = __builtin_bit_cast(unsigned short, (_Float16)clamp_f16(o));
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
    if constexpr (kBiasLike<EPI> || kGeluLike<EPI>) {
      f32x4 o = v + b;
      if constexpr (kFold<EPI>) {
        const float s = a.stats[2 * (size_t)m], u = a.stats[2 * (size_t)m + 1];
        const f32x4 c = *reinterpret_cast<const f32x4*>(a.colsum + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(s, v[e], fmaf(u, c[e], b[e]));
      }
      if constexpr (kBiasLike<EPI>) {
        if (a.scale) o = o * *reinterpret_cast<const f32x4*>(a.scale + n);
      }
      if constexpr (kGeluLike<EPI>) {
        const f32x2 g0 = gelu_erf2((f32x2){o[0], o[1]}), g1 = gelu_erf2((f32x2){o[2], o[3]});
        o = (f32x4){g0[0], g0[1], g1[0], g1[1]};
      }
      if constexpr (kSplit2Out<EPI>) {
        u32x2 hi, lo;
        { const u32x2 t2 = split2_pack(o[0], o[1]); hi[0] = t2[0]; lo[0] = t2[1]; }
        { const u32x2 t2 = split2_pack(o[2], o[3]); hi[1] = t2[0]; lo[1] = t2[1]; }
        bf16_raw* row = reinterpret_cast<bf16_raw*>(a.out) + (size_t)m * 3 * a.N + n;
        *reinterpret_cast<u32x2*>(row) = hi;
        *reinterpret_cast<u32x2*>(row + a.N) = hi;
        *reinterpret_cast<u32x2*>(row + 2 * (size_t)a.N) = lo;
        return;
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
    } else if constexpr (kResidH16<EPI>) {
      const size_t i = (size_t)m * a.N + n;
      const u32x2 rw = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(a.resid) + i);
      float r0, r1, r2, r3;
      unpack_f16x2(rw[0], r0, r1);
      unpack_f16x2(rw[1], r2, r3);
      const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
      const f32x4 o = (f32x4){r0, r1, r2, r3} + sc * (v + b);
      if (beyond_f16(o[0]) || beyond_f16(o[1]) || beyond_f16(o[2]) || beyond_f16(o[3])) atomicAdd(a.ovf, 1u);
      u32x2 w;
      w[0] = pack_f16x2(clamp_f16(o[0]), clamp_f16(o[1]));
      w[1] = pack_f16x2(clamp_f16(o[2]), clamp_f16(o[3]));
      *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned short*>(a.out) + i) = w;
    } else if constexpr (EPI == UCOD_EPI_PATCH_TOKENS_F32 || kPatchH16<EPI>) {
      const int np = a.tok - 1;
      const int bi = m / np, p = m - bi * np;
      const f32x4 ps = *reinterpret_cast<const f32x4*>(a.pos + (size_t)(1 + p) * a.N + n);
      const f32x4 o = v + b + ps;
      if constexpr (kPatchH16<EPI>) {
        if (beyond_f16(o[0]) || beyond_f16(o[1]) || beyond_f16(o[2]) || beyond_f16(o[3])) atomicAdd(a.ovf, 1u);
        u32x2 w;
        w[0] = pack_f16x2(clamp_f16(o[0]), clamp_f16(o[1]));
        w[1] = pack_f16x2(clamp_f16(o[2]), clamp_f16(o[3]));
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
  if constexpr (kFold<EPI>) {
    const float s = a.stats[2 * (size_t)m], u = a.stats[2 * (size_t)m + 1];
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + n), b1 = *reinterpret_cast<const f32x4*>(a.bias + n + 4);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(a.colsum + n), c1 = *reinterpret_cast<const f32x4*>(a.colsum + n + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o0[e] = fmaf(s, v0[e], fmaf(u, c0[e], b0[e]));
      o1[e] = fmaf(s, v1[e], fmaf(u, c1[e], b1[e]));
    }
  }
  if constexpr (kBiasLike<EPI>) {
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
  if constexpr (kSplit2Out<EPI>) {
    u32x4 hi, lo;
    { const u32x2 t2 = split2_pack(o0[0], o0[1]); hi[0] = t2[0]; lo[0] = t2[1]; }
    { const u32x2 t2 = split2_pack(o0[2], o0[3]); hi[1] = t2[0]; lo[1] = t2[1]; }
    { const u32x2 t2 = split2_pack(o1[0], o1[1]); hi[2] = t2[0]; lo[2] = t2[1]; }
    { const u32x2 t2 = split2_pack(o1[2], o1[3]); hi[3] = t2[0]; lo[3] = t2[1]; }
    bf16_raw* row = reinterpret_cast<bf16_raw*>(a.out) + (size_t)m * 3 * a.N + n;
    *reinterpret_cast<u32x4*>(row) = hi;
    *reinterpret_cast<u32x4*>(row + a.N) = hi;
    *reinterpret_cast<u32x4*>(row + 2 * (size_t)a.N) = lo;
    return;
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
  constexpr bool BF16_OUT = (kBiasLike<EPI> || kGeluLike<EPI>);
  if constexpr (BF16_OUT && (WCOLS % 8) == 0 && (ROWS * (WCOLS / 8)) % 64 == 0) {
    constexpr int CH = WCOLS / 8;                     // 32-byte (8 x f32) chunks per row -> 16-byte bf16 stores
    if ((a.N & 7) == 0) {
#pragma unroll
      for (int it = 0; it < ROWS * CH / 64; ++it) {
        const int idx = it * 64 + lane, r = idx / CH, c = idx - r * CH;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(wbase + r * EPI_PITCH(WCOLS) + c * 32);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(wbase + r * EPI_PITCH(WCOLS) + c * 32 + 16);
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
    const f32x4 v = *reinterpret_cast<const f32x4*>(wbase + r * EPI_PITCH(WCOLS) + c * 16);
    epilogue_store4<EPI>(a, m_first + r, n_first + c * 4, v);
  }
}

// Partial statistics of one row's 64-column slot, from the eight fp16 values in w of each of the 8 consecutive lanes that hold the slot (lane & 7 = chunk):
// (S, M2) = (sum, sum of squared deviations from the SLOT's own mean S / 64).  Two rounds of three DPP steps (xor 1, xor 2 inside the quad, then the mirrored
// lane of the other quad); every lane of the group ends with both totals.  Round 6: M2 about the slot mean instead of the raw sum of squares -- the consumer
// merges the slots with Chan's parallel-variance formula (fold_finish), so that no difference of two large nearly equal numbers is ever formed: a row with
// |mean| >> sigma keeps its variance to f32 rounding (VERDICT r5 weak #3 / ADVICE r5: E[x^2] - mean^2 lost it at |mean| / sigma ~ 100).
template <int CTRL>
__device__ __forceinline__ float dpp_add_t(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
  return v + __builtin_bit_cast(float, o);
}
__device__ __forceinline__ float slot_sum8(float v) {
  v = dpp_add_t<0xB1>(v);       // quad_perm [1,0,3,2]
  v = dpp_add_t<0x4E>(v);       // quad_perm [2,3,0,1]
  return dpp_add_t<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of each group of 8
}
__device__ __forceinline__ f32x2 row_partial8(const u32x4& w) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) unpack_f16x2(w[e], x[2 * e], x[2 * e + 1]);
  const float ps = slot_sum8(((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7])));
  const float m = ps * (1.0f / 64.0f);
  float pq = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float d = x[e] - m;
    pq = fmaf(d, d, pq);
  }
  return (f32x2){ps, slot_sum8(pq)};
}

// ---- LayerNorm-folded consumers: the tile's row table and the stager's constants ----------------------------------------------------
// Large-tile kernels keep (s, u) = (rstd, -mean * rstd) of the tile's rows in LDS (behind the two K-tile buffers): thread t of the workgroup owns
// row m0 + t, requests either its `stats` pair or its nslot (sum, M2) partials BEFORE the operand DMAs are issued (oldest in the
// vmcnt queue: they have landed when the counted prologue wait returns), turns them into (s, u) and writes the table ahead of the K loop's first
// barrier.  The stager then reads four rows' scalars per 16-row accumulator tile with two ds_read_b128 (requested one pass ahead) and applies
//   out = s[row] * (acc * q[col]) + (u[row] * (c q)[col] + (b' q)[col])      (q = optional column scale)
// in the MFMA C layout, where a lane owns one column per tile: the row-major drain behind it is the unfolded one (plus GELU for fc1).
// (ablation builds only -- wrong results, timing: bit 0 no prologue work, bit 1 no arithmetic in the stager, bit 2 no table reads)
#ifndef UCOD_FOLD_ABL
#define UCOD_FOLD_ABL 0
#endif
constexpr float kFoldMeanGuard = 65536.0f;                         // (|mean| / sigma)^2 above which a row is reported (fold_finish, row_stats_h16_kernel)
constexpr int FOLD_TAB_ROWS = 288;                                 // >= rows of the tallest tile (2 x 144)
// Round 6 experiment, NOT the product form: -DUCOD_FOLD_RANK1=1 sends the rank-one term  -mean[m] * c[n]  of the fold through the MATRIX pipe
// (fold_rank_one below: one extra 16 x 16 x 32 MFMA per accumulator tile behind the K loop, 32 against the loop's 768), so that the stager applies ONE FMA per
// element -- s * acc + b' -- and reads one table column instead of two.  Correct (tests/test_gpu_lnfold.py passes on it) and SLOWER: the 32 MFMAs, their operand
// conversions and table reads run serially between the K loop and the drain, while the stager's second FMA hides under its own LDS writes -- QKV 153.1 vs 149.8 us,
// fc1 213.4 vs 210.5 isolated; 158.1 vs 153.7 and 221.2 vs 216.8 in the step (profiles/r06_lnfold_ab.txt: gate of 149 / 211 us missed, experiment closed).
#ifndef UCOD_FOLD_RANK1
#define UCOD_FOLD_RANK1 0
#endif
constexpr int FOLD_TAB_BYTES = 3 * FOLD_TAB_ROWS * 4;              // s | u | -mean
constexpr int FOLD_MAX_SLOT_PAIRS = 12;                            // nslot <= 24 (D <= 1536)
template <int NT>
struct FoldCtx {
  const float* tab;      // LDS: s of this wave group's rows at tab[r], u at tab[FOLD_TAB_ROWS + r]
  float cc[NT], cb[NT];  // (colsum * q), (bias' * q) of the lane's column per 16-wide tile
};
struct FoldReq {
  f32x4 p[FOLD_MAX_SLOT_PAIRS];
  f32x2 su;
};
constexpr int FOLD_BASE_PAIRS = 6;                                 // slot pairs requested unconditionally (D = 768: all of them); the rest under one wave-uniform test
// request: no wait; rows / slots that do not exist read zeros through the descriptors' range check.  Only the waves that own rows of the tile take part
// (`wave` is a scalar: the test is a scalar branch taken before any operand DMA has been issued).
__device__ __forceinline__ void fold_request(const GemmArgs& a, int m0, int rows, int wave, int lane, FoldReq& r) {
  if constexpr ((UCOD_FOLD_ABL & 1) != 0) return;
  constexpr unsigned OOB = 0xFFFFFFF0u;
  if (wave * 64 >= rows) return;
  const int tid = wave * 64 + lane;
  const int m = m0 + tid;
  const bool live = tid < rows && m < a.M;
  const bool parts = a.part_in != nullptr;
  const unsigned prow = (unsigned)a.nslot * 8u;
  const unsigned long pbytes = parts ? (unsigned long)a.M * prow : 0ul;
  const auto rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(parts ? a.part_in : a.colsum), 0, pbytes > 0xFFFFFFF0ul ? 0xFFFFFFF0u : (unsigned)pbytes, 0x00020000);
  const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(parts ? a.colsum : a.stats), 0, parts ? 0u : (unsigned)a.M * 8u, 0x00020000);
#pragma unroll
  for (int i = 0; i < FOLD_BASE_PAIRS; ++i)
    r.p[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, (live && 2 * i < a.nslot) ? (unsigned)m * prow + (unsigned)i * 16u : OOB, 0, 0));
  if (a.nslot > 2 * FOLD_BASE_PAIRS) {
#pragma unroll
    for (int i = FOLD_BASE_PAIRS; i < FOLD_MAX_SLOT_PAIRS; ++i)
      r.p[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, (live && 2 * i < a.nslot) ? (unsigned)m * prow + (unsigned)i * 16u : OOB, 0, 0));
  }
  r.su = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_s, live ? (unsigned)m * 8u : OOB, 0, 0));
}
// finish: (s, u) of this thread's row into the table.  Partials: slot i holds (S_i, M2_i) of its 64 columns (row_partial8); merged with Chan's formula
//   mean = sum S_i / K,   M2 = sum M2_i + 64 * sum (S_i / 64 - mean)^2,   var = M2 / K  (biased, like nn.LayerNorm)
// -- every term is a sum of squares of small differences, nothing cancels: the result matches the two-pass kernel (ucod_row_stats_h16) to f32 rounding
// whatever the row's |mean| / sigma is, so a key map does not depend on which of the two paths a pass of a given size takes.
__device__ __forceinline__ void fold_finish(const GemmArgs& a, char* tab_bytes, int rows, int wave, int lane, const FoldReq& r) {
  if constexpr ((UCOD_FOLD_ABL & 1) != 0) return;
  if (wave * 64 >= rows) return;
  const int tid = wave * 64 + lane;
  float S = 0.f, Q = 0.f;
#pragma unroll
  for (int i = 0; i < FOLD_BASE_PAIRS; ++i) {
    S += r.p[i][0] + r.p[i][2];
    Q += r.p[i][1] + r.p[i][3];
  }
  if (a.nslot > 2 * FOLD_BASE_PAIRS) {
#pragma unroll
    for (int i = FOLD_BASE_PAIRS; i < FOLD_MAX_SLOT_PAIRS; ++i) {
      S += r.p[i][0] + r.p[i][2];
      Q += r.p[i][1] + r.p[i][3];
    }
  }
  const float inv_d = 1.0f / (float)a.K;
  const float mean = S * inv_d;
  float dev = 0.f;                                                 // sum over the slots of (slot mean - row mean)^2; slots past nslot read zeros and are masked
#pragma unroll
  for (int i = 0; i < FOLD_BASE_PAIRS; ++i) {
    const float live = 2 * i < a.nslot ? 1.f : 0.f;               // (wave-uniform)
    const float d0 = fmaf(r.p[i][0], 1.0f / 64.0f, -mean), d1 = fmaf(r.p[i][2], 1.0f / 64.0f, -mean);
    dev = fmaf(live, fmaf(d0, d0, d1 * d1), dev);
  }
  if (a.nslot > 2 * FOLD_BASE_PAIRS) {
#pragma unroll
    for (int i = FOLD_BASE_PAIRS; i < FOLD_MAX_SLOT_PAIRS; ++i) {
      const float live = 2 * i < a.nslot ? 1.f : 0.f;
      const float d0 = fmaf(r.p[i][0], 1.0f / 64.0f, -mean), d1 = fmaf(r.p[i][2], 1.0f / 64.0f, -mean);
      dev = fmaf(live, fmaf(d0, d0, d1 * d1), dev);
    }
  }
  const float var = fmaf(64.0f, dev, Q) * inv_d;
  const float rstd = rsqrtf(var + a.eps);
  const bool parts = a.part_in != nullptr;
  // Range guard of the fold ITSELF (not of its statistics, which are exact to rounding): x W'^T and mean * colsum are summed in f32 at magnitude |mean| |colsum|
  // before they cancel, which costs ~2^-24 sqrt(K) |mean| / sigma of the output scale -- 0.2 fp16 ulp at |mean| = 100 sigma, 2 ulps at 1000 sigma.  A live row with
  // |mean| > 256 sigma is COUNTED into the engine's device word (the fp16 stream's saturation counter: ViTEngine.check_overflow raises) instead of passing silently.
  if (parts && tid < rows && a.ovf && mean * mean > kFoldMeanGuard * (var + a.eps)) atomicAdd(a.ovf, 1u);
  const float s = parts ? rstd : r.su[0], u = parts ? -mean * rstd : r.su[1];
  if (tid < FOLD_TAB_ROWS) {
    float* tab = reinterpret_cast<float*>(tab_bytes);
    tab[tid] = s;
    tab[FOLD_TAB_ROWS + tid] = u;
    tab[2 * FOLD_TAB_ROWS + tid] = parts ? -mean : (r.su[0] != 0.f ? r.su[1] / r.su[0] : 0.f);   // (rows past M read zeros: s = 0)
  }
}

// The fold's rank-one term on the matrix pipe:  acc[i][j] += (-mean[row]) * c[col]  as one v_mfma_f32_16x16x32 per accumulator tile whose 32-deep K slice holds
//   A[row][0..2] = (mh, mh, ml * 2^11)      B[col][0..2] = (ch, cl, ch * 2^-11)      everything else zero,
// mh + ml = -mean and ch + cl = c split into two fp16 terms (22 significand bits each; the second-order term ml * cl is below 2^-22 of the product and is dropped;
// the powers of two keep ml away from fp16's subnormal range whatever |mean| is and are exact).  The three products are exact in the f32 accumulator; what the
// MFMA adds is what `fmaf(u, c, .)` added in the stager, to 2^-22 instead of 2^-24 of |mean c| -- far below the f32 rounding of the K-long sum it cancels against.
// Lane l supplies row / column (l & 15) and the k slice 8 (l >> 4) .. + 7: only the lanes of slice 0 carry values.
template <int NT, int NI>
__device__ __forceinline__ void fold_rank_one(f32x4 (&acc)[NI][NT], const float* tab, const float (&craw)[NT], int lane) {
  const bool k0 = (lane >> 4) == 0;
  hx8 fbx[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float c = k0 ? craw[j] : 0.f;
    const half_t ch = (half_t)c;
    const half_t cl = (half_t)(c - (float)ch);
    const half_t chs = (half_t)((float)ch * 0.00048828125f);
    const half_t z = (half_t)0.f;
    fbx[j] = (hx8){ch, cl, chs, z, z, z, z, z};
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float nm = k0 ? tab[2 * FOLD_TAB_ROWS + i * 16 + (lane & 15)] : 0.f;
    const half_t mh = (half_t)nm;
    const half_t ml = (half_t)((nm - (float)mh) * 2048.0f);
    const half_t z = (half_t)0.f;
    const hx8 fax = (hx8){mh, mh, ml, z, z, z, z, z};
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = UCOD_MFMA16(fax, fbx[j], acc[i][j]);
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
constexpr bool kColFused = (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16 || kSplit2Out<EPI> || EPI == UCOD_EPI_BIAS_SCALE_RESID_F32 ||
                            EPI == UCOD_EPI_BIAS_F32 || EPI == UCOD_EPI_GELU_BWD_BF16 || EPI == UCOD_EPI_BIAS_GELU_SAVE_BF16 ||
                            EPI == UCOD_EPI_QKV_FP8 || kResidH16<EPI> || kFold<EPI>);
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
      if constexpr (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32 || kResidH16<EPI>) cs[j] = a.scale[n];
      // optional scale: unconditional load now (a branch here costs a vmcnt(0) at the join, ahead of the operand DMAs),
      // select at the point of use (finish_col_consts) so nothing waits on the load before the DMAs are out
      if constexpr (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_QKV_FP8 || EPI == UCOD_EPI_LNFOLD_BIAS_BF16) cs[j] = (a.scale ? a.scale : reinterpret_cast<const float*>(a.B))[n];
    }
  }
}

template <int EPI, int NT>
__device__ __forceinline__ void finish_col_consts(const GemmArgs& a, float (&cb)[NT], float (&cs)[NT]) {
  if constexpr (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_QKV_FP8 || EPI == UCOD_EPI_LNFOLD_BIAS_BF16) {
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
// kRowMapped: the key hook (rows = channels, columns = tokens written as [B, C, tok-1]) and the patch embedding (rows remapped past the CLS
// rows, + position embedding).  With 64-column waves the large-tile kernel drains them with the offset scheme of the fused epilogues
// (FASTRM: what a pass needs from memory -- the per-channel bias rows of the key hook, the position rows of the patch embedding -- is
// requested before the pass is staged, so no load sits between two stores; round 3: their chunk-by-chunk drain had a bias / position load
// and a vmcnt(0) in front of every store, one store round trip per store instruction: key hook 106 us, patch embedding 78 us per 32
// images.  Same order of additions as the chunk-by-chunk drains, (sum + bias) + position: bitwise the same values on every tile path.)
template <int EPI>
constexpr bool kRowMapped = (EPI == UCOD_EPI_KEY_NCHW_F32 || EPI == UCOD_EPI_PATCH_TOKENS_F32 || kPatchH16<EPI>);
template <int EPI, int NT>
constexpr bool kFastRowMapped = kRowMapped<EPI> && NT == 4;

// The drain proper takes a STAGER: stage(pass) writes the wave's 32 x WCOLS f32 values of pass `pass` (column scale applied) into the wave-private
// staging area, row-major.  big_epilogue() below supplies the one for 16 x 16 accumulator tiles; a kernel on 32 x 32 MFMA tiles supplies its own.
template <int EPI>
constexpr bool kStageScaled = (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_SCALE_RESID_F32 || EPI == UCOD_EPI_QKV_FP8 ||
                               kResidH16<EPI> || EPI == UCOD_EPI_LNFOLD_BIAS_BF16);

template <int EPI, int NT, int NI, int AUX, bool FASTRM, class Stage>
__device__ __forceinline__ void big_epilogue_staged(const GemmArgs& a, const Stage& stage, char* wbase, int m_first, int n_first, int lane) {
  constexpr int WCOLS = 16 * NT, PR = 32;
  constexpr int NP = (NI + 1) / 2;                    // passes of 32 rows; with NI odd the last pass holds 16 rows (rows 16..31 masked off)
  static_assert(NI == 8 || kColFused<EPI>, "odd row-tile counts only in the column-fused epilogues");
  auto rows_in = [&](int pass) { return (NI - 2 * pass) >= 2 ? 32 : 16; };
  if constexpr (FASTRM && EPI == UCOD_EPI_KEY_NCHW_F32) {
    // out f32 [B, C = M, tok-1]; the lane's four tokens (columns) are the same for every row, so its byte offset is one register plus a
    // wave-uniform row term.  Four consecutive tokens of one image, none of them CLS: one 16-byte store (rows start 4-byte aligned only:
    // tok-1 is odd); chunks that hold a CLS token or straddle two images go element by element, in a branch the whole wave takes or skips.
    static_assert(WCOLS == 64 && NI == 8, "64-column waves");
    constexpr unsigned DROP = 0x80000000u;
    const int tok = a.tok, np1 = tok - 1;
    const unsigned total = (unsigned)(a.N / tok) * (unsigned)a.M * (unsigned)np1 * 4u;         // (launch(): < 2^31)
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out), 0, total, 0x00020000);
    const unsigned row_bytes = (unsigned)np1 * 4u;
    const int n = n_first + (lane & 15) * 4;
    const int lrow = m_first + (lane >> 4);
    const unsigned lane_row = (unsigned)lrow * row_bytes;
    const int b = n / tok, t = n - b * tok;
    const bool clean = n + 3 < a.N && t >= 1 && t + 3 < tok;
    const unsigned off_w = clean ? ((unsigned)b * (unsigned)a.M * (unsigned)np1 + (unsigned)(t - 1)) * 4u + lane_row : DROP;
    unsigned off_e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ne = n + e, be = ne / tok, te = ne - be * tok;
      off_e[e] = (!clean && ne < a.N && te != 0) ? ((unsigned)be * (unsigned)a.M * (unsigned)np1 + (unsigned)(te - 1)) * 4u + lane_row : DROP;
    }
    const bool ragged = __any(!clean && n < a.N);                 // (wave-uniform)
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      float bm[8];                                                // bias of the lane's eight rows of this pass
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int m = lrow + pass * PR + it * 4;
        bm[it] = a.bias[m < a.M ? m : a.M - 1];
      }
      stage(pass);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wbase + (it * 4 + (lane >> 4)) * EPI_PITCH(WCOLS) + (lane & 15) * 16) + (f32x4){bm[it], bm[it], bm[it], bm[it]};
        const unsigned rowterm = (unsigned)(pass * PR + it * 4) * row_bytes;
        const bool row_ok = lrow + pass * PR + it * 4 < a.M;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, row_ok ? off_w + rowterm : DROP, 0, 0);
        if (ragged) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            unsigned ve = __builtin_bit_cast(u32x4, v)[e];
            asm volatile("" : "+v"(ve));   // (hipcc 7.2 otherwise stores element 0 four times and reuses the other three registers for the offsets)
            __builtin_amdgcn_raw_buffer_store_b32(ve, rs, row_ok ? off_e[e] + rowterm : DROP, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else if constexpr (FASTRM && (EPI == UCOD_EPI_PATCH_TOKENS_F32 || kPatchH16<EPI>)) {
    // row m = image * (tok-1) + p  ->  token row image * tok + 1 + p = m + image + 1, + position embedding of token 1 + p.  The position
    // rows of a pass are requested before its accumulators are staged; nothing else is loaded, so the wait in front of a pass's stores is
    // the only one (it also retires the previous pass's stores: four round trips per tile instead of one per store).
    static_assert(WCOLS == 64 && NI == 8, "64-column waves");
    constexpr bool H16 = (kPatchH16<EPI>);
    constexpr unsigned DROP = 0x80000000u;
    constexpr int EW = H16 ? 8 : 4;                               // columns per lane and store
    constexpr int CH = WCOLS / EW, RPI = 64 / CH, ITS = PR / RPI; // chunks per row, rows per wave instruction, instructions per pass
    const int np = a.tok - 1;
    const unsigned out_bytes = (unsigned)(a.M / np) * (unsigned)a.tok * (unsigned)a.N * (H16 ? 2u : 4u);   // (launch(): < 2^31)
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out), 0, out_bytes, 0x00020000);
    const auto rs_p = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.pos)), 0, (unsigned)a.tok * (unsigned)a.N * 4u, 0x00020000);
    const auto rs_q = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(kStats<EPI> ? (void*)a.part_out : a.out), 0,
                                                        kStats<EPI> ? (unsigned)(a.M / np) * (unsigned)a.tok * (unsigned)a.nslot * 8u : 0u, 0x00020000);
    const int n = n_first + (lane % CH) * EW;
    const bool col_ok = n < a.N;
    const int lrow = m_first + lane / CH;
    f32x4 cbias[H16 ? 2 : 1];                                     // bias of the lane's EW columns
    {
      const int nb = col_ok ? n : 0;
      cbias[0] = *reinterpret_cast<const f32x4*>(a.bias + nb);
      if constexpr (H16) cbias[1] = *reinterpret_cast<const f32x4*>(a.bias + nb + 4);
    }
    float amax = 0.f;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      unsigned o_off[ITS];
      unsigned q_off[kStats<EPI> ? ITS : 1];                      // *_STATS: byte offset of (output token row, slot n_first / 64) in the partial-sum table
      u32x4 pv[ITS][H16 ? 2 : 1];
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const int m = lrow + pass * PR + it * RPI;
        const int bi = m / np, p = m - bi * np;
        const bool ok = col_ok && m < a.M;
        o_off[it] = ok ? ((unsigned)(m + bi + 1) * (unsigned)a.N + (unsigned)n) * (H16 ? 2u : 4u) : DROP;
        if constexpr (kStats<EPI>) q_off[it] = (ok && (lane % CH) == 0) ? ((unsigned)(m + bi + 1) * (unsigned)a.nslot + (unsigned)(n_first >> 6)) * 8u : DROP;
        const unsigned p_off = ok ? ((unsigned)(1 + p) * (unsigned)a.N + (unsigned)n) * 4u : DROP;
        pv[it][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, p_off, 0, 0);
        if constexpr (H16) pv[it][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_p, p_off + 16u, 0, 0);
      }
      stage(pass);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < ITS; ++it) {
        const char* src = wbase + (it * RPI + lane / CH) * EPI_PITCH(WCOLS) + (lane % CH) * (EW * 4);
        f32x4 v0 = (*reinterpret_cast<const f32x4*>(src) + cbias[0]) + __builtin_bit_cast(f32x4, pv[it][0]);
        if constexpr (H16) {
          const f32x4 v1 = (*reinterpret_cast<const f32x4*>(src + 16) + cbias[H16 ? 1 : 0]) + __builtin_bit_cast(f32x4, pv[it][1]);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            amax = __builtin_elementwise_maximum(amax, __builtin_elementwise_maximum(__builtin_fabsf(v0[e]), __builtin_fabsf(v0[e + 1])));      // IEEE maximum: a NaN stays a NaN
            amax = __builtin_elementwise_maximum(amax, __builtin_elementwise_maximum(__builtin_fabsf(v1[e]), __builtin_fabsf(v1[e + 1])));
          }
          u32x4 w;
          w[0] = pack_f16x2(clamp_f16(v0[0]), clamp_f16(v0[1]));
          w[1] = pack_f16x2(clamp_f16(v0[2]), clamp_f16(v0[3]));
          w[2] = pack_f16x2(clamp_f16(v1[0]), clamp_f16(v1[1]));
          w[3] = pack_f16x2(clamp_f16(v1[2]), clamp_f16(v1[3]));
          __builtin_amdgcn_raw_buffer_store_b128(w, rs_o, o_off[it], 0, 0);
          if constexpr (kStats<EPI>) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, row_partial8(w)), rs_q, q_off[it], 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), rs_o, o_off[it], 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (H16) {
      if (!(amax <= F16_MAX)) atomicAdd(a.ovf, 1u);          // true for NaN too (beyond_f16's convention, common.h)
    }
  } else if constexpr (!kColFused<EPI>) {
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
    constexpr int ELT = (kF32Out<EPI> ? 4 : 2) * kOutPitchMul<EPI>;   // bytes per output column of a row's pitch (the split epilogue's rows are 3 N wide)
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
            const f32x4 v = *reinterpret_cast<const f32x4*>(wbase + r * EPI_PITCH(WCOLS) + c * 64 + e * 16);
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
          f32x4 o = *reinterpret_cast<const f32x4*>(wbase + lrow[it] * EPI_PITCH(WCOLS) + lchk[it] * 16);
          if constexpr (RESID) o = o + __builtin_bit_cast(f32x4, rb[pass & 1][it]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_out, at(it, pass), 0, AUX);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {                                                      // bf16 out: 16-byte stores (launch() guarantees N % 8 == 0)
      constexpr bool GBWD = (EPI == UCOD_EPI_GELU_BWD_BF16), SAVE = (EPI == UCOD_EPI_BIAS_GELU_SAVE_BF16);
      constexpr bool RH16 = (kResidH16<EPI>);   // second matrix = the f16 residual stream (may alias out)
      constexpr int CH = WCOLS / 8, ITS = PR * CH / 64;
      static_assert((PR * CH) % 64 == 0, "whole wave instructions");
      // second bf16 [M,N] matrix with the same geometry: the saved pre-activation, read (GELU_BWD) or written (GELU_SAVE)
      const void* second = GBWD ? a.aux : (SAVE ? (const void*)a.out2 : (RH16 ? (const void*)a.resid : (const void*)a.out));
      // 64-column waves (CH == 8): wave instruction `it` covers rows it*8 + lane/8, chunk lane%8, so the lane's byte offset is ONE register
      // (its pass-0 / it-0 offset, or DROP when its columns lie past N) plus a wave-uniform term: one v_add per store instead of the ~11
      // vector instructions (one of them v_mul_lo_u32) of the generic (row, chunk) arithmetic -- the drain of fc1 is bound by its vector
      // work (profiles/r03_fc1_gelu_epilogue.txt).  DROP + any tile-relative offset stays >= 2^31 and the descriptors end below that.
      constexpr bool FAST = (CH == 8);
      constexpr unsigned DROP = 0x80000000u;
      const unsigned rec16 = FAST ? (records > 0x7FFFFFF0u ? 0x7FFFFFF0u : records) : records;
      const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + base, 0, rec16, 0x00020000);
      const auto rs_2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(second)) + base, 0,
                                                          (GBWD || SAVE || RH16) ? rec16 : 0u, 0x00020000);
      // (row, chunk) of wave instruction `it`: recomputed where needed -- index arrays cost registers the persistent kernel lacks
      auto lrow = [&](int it) { return (it * 64 + lane) / CH; };
      auto lchk = [&](int it) { return (it * 64 + lane) - lrow(it) * CH; };
      const unsigned off0 = (n_first + (lane & 7) * 8 < a.N) ? (unsigned)(lane >> 3) * row_bytes + (unsigned)(n_first + (lane & 7) * 8) * 2u : DROP;
      const char* lds0 = wbase + (lane >> 3) * EPI_PITCH(WCOLS) + (lane & 7) * 32;
      auto live = [&](int it, int pass) { return !FAST || it * 8 < rows_in(pass); };   // (compile-time after unrolling) rows 16..31 of a 16-row last pass
      auto at = [&](int it, int pass) -> unsigned {
        if constexpr (FAST) return off0 + (unsigned)(pass * PR + it * 8) * row_bytes;
        const int n = n_first + lchk(it) * 8;
        return (n < a.N && lrow(it) < rows_in(pass)) ? (unsigned)(pass * PR + lrow(it)) * row_bytes + (unsigned)n * 2u : OOB;
      };
      auto lds_at = [&](int it) -> const char* {
        if constexpr (FAST) return lds0 + it * 8 * EPI_PITCH(WCOLS);
        return wbase + lrow(it) * EPI_PITCH(WCOLS) + lchk(it) * 32;
      };
      float amax = 0.f;                                             // RH16: largest |x_new| this lane produced (saturation test after the stores)
      // *_STATS: this launch's rows of the partial-sum table [M][nslot][2] f32, descriptor from the wave tile's first row (rows past M are dropped)
      const unsigned part_row_bytes = kStats<EPI> ? (unsigned)a.nslot * 8u : 0u, part_slot_off = kStats<EPI> ? (unsigned)(n_first >> 6) * 8u : 0u;
      const unsigned long part_left = kStats<EPI> && rows_left > 0 ? (unsigned long)rows_left * part_row_bytes : 0ul;
      const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char*>(kStats<EPI> ? (void*)a.part_out : a.out) + (kStats<EPI> ? (size_t)(m_first < a.M ? m_first : 0) * part_row_bytes : 0), 0,
          part_left > 0x7FFFFFF0ul ? 0x7FFFFFF0u : (unsigned)part_left, 0x00020000);
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
            for (int it = 0; it < ITS; ++it) {
              if (!live(it, pass + 1)) continue;
              pre[(pass + 1) & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_2, at(it, pass + 1), 0, 0);
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < ITS; ++it) {
          if (!live(it, pass)) continue;
          f32x4 v0 = *reinterpret_cast<const f32x4*>(lds_at(it));
          f32x4 v1 = *reinterpret_cast<const f32x4*>(lds_at(it) + 16);
          if constexpr (SAVE) {                                   // pre-activation out first
            u32x4 w;
            w[0] = pack_h2(v0[0], v0[1]);
            w[1] = pack_h2(v0[2], v0[3]);
            w[2] = pack_h2(v1[0], v1[1]);
            w[3] = pack_h2(v1[2], v1[3]);
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_2, at(it, pass), 0, 0);
          }
          if constexpr (SAVE || kGeluLike<EPI>) {  // GELU in the row-major layout (fewer live registers than in the C layout)
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
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = v0[e] + r[e]; x[4 + e] = v1[e] + r[4 + e]; }
#pragma unroll
            for (int e = 0; e < 8; e += 2) amax = __builtin_elementwise_maximum(amax, __builtin_elementwise_maximum(__builtin_fabsf(x[e]), __builtin_fabsf(x[e + 1])));   // (v_maximum3 with |.| modifiers; NaN-propagating)
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = pack_f16x2(clamp_f16(x[2 * e]), clamp_f16(x[2 * e + 1]));
            if constexpr (kStats<EPI>) {                          // (sum, M2 about their mean) of the row's 64 ROUNDED values of this wave -> slot n_first / 64
              static_assert(FAST, "row partials need 64-column waves");
              const f32x2 pq = row_partial8(w);
              const unsigned po = (lane & 7) == 0 ? (unsigned)(pass * PR + it * 8 + (lane >> 3)) * part_row_bytes + part_slot_off : DROP;
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pq), rs_part, po, 0, 0);
            }
          } else if constexpr (kSplit2Out<EPI>) {                  // segments hi | hi | lo, N elements apart in a row of 3 N
            u32x4 lo;
            { const u32x2 t2 = split2_pack(v0[0], v0[1]); w[0] = t2[0]; lo[0] = t2[1]; }
            { const u32x2 t2 = split2_pack(v0[2], v0[3]); w[1] = t2[0]; lo[1] = t2[1]; }
            { const u32x2 t2 = split2_pack(v1[0], v1[1]); w[2] = t2[0]; lo[2] = t2[1]; }
            { const u32x2 t2 = split2_pack(v1[2], v1[3]); w[3] = t2[0]; lo[3] = t2[1]; }
            const unsigned o = at(it, pass);
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_o, o + (unsigned)a.N * 2u, 0, AUX);
            __builtin_amdgcn_raw_buffer_store_b128(lo, rs_o, o + (unsigned)a.N * 4u, 0, AUX);
          } else {
            w[0] = pack_h2(v0[0], v0[1]);
            w[1] = pack_h2(v0[2], v0[3]);
            w[2] = pack_h2(v1[0], v1[1]);
            w[3] = pack_h2(v1[2], v1[3]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(w, rs_o, at(it, pass), 0, AUX);
          __builtin_amdgcn_sched_barrier(0);                      // keep chunks in order: hoisting every ds_read/cvt of a pass spills in the persistent kernel
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (RH16) {                                         // after the last store: nothing waits on this
        if (!(amax <= F16_MAX)) atomicAdd(a.ovf, 1u);          // true for NaN too (beyond_f16's convention, common.h)
      }
    }
  }
}


// 16 x 16 accumulator tiles (C layout: col = lane & 15, row = 4 * (lane >> 4) + reg), bias already inside, cs = per-column scale
template <int EPI, int NT, int NI = 8, int AUX = UCOD_ST_AUX, bool FASTRM = false>
__device__ __forceinline__ void big_epilogue(const GemmArgs& a, f32x4 (&acc)[NI][NT], const float (&cs)[NT], char* wbase,
                                             int m_first, int n_first, int lane, const FoldCtx<NT>* fold = nullptr) {
  constexpr int WCOLS = 16 * NT;
  // LayerNorm-folded: (rstd, -mean * rstd) of the lane's four rows per 16-row tile, read from the tile's LDS table ONE PASS AHEAD (the reads of pass p + 1
  // are issued behind the staging writes of pass p and land under its drain)
  f32x4 sn[2], un[2];
  auto fold_fetch = [&](int pass) {
    if constexpr (kFold<EPI> && !(UCOD_FOLD_ABL & 4)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (pass * 2 + i >= NI) continue;
        const float* t = fold->tab + (pass * 2 + i) * 16 + (lane >> 4) * 4;
        sn[i] = *reinterpret_cast<const f32x4*>(t);
        if constexpr (!UCOD_FOLD_RANK1) un[i] = *reinterpret_cast<const f32x4*>(t + FOLD_TAB_ROWS);
      }
    }
  };
  fold_fetch(0);
  auto stage = [&](int pass) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (pass * 2 + i >= NI) continue;
      const f32x4 s4 = sn[i], u4 = un[i];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 v = acc[pass * 2 + i < NI ? pass * 2 + i : 0][j];
        if constexpr (kStageScaled<EPI>) v = v * cs[j];
        if constexpr (kFold<EPI> && !(UCOD_FOLD_ABL & 2)) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            if constexpr (UCOD_FOLD_RANK1) v[rg] = fmaf(s4[rg], v[rg], fold->cb[j]);          // (-mean c is inside the accumulator: fold_rank_one)
            else v[rg] = fmaf(s4[rg], v[rg], fmaf(u4[rg], fold->cc[j], fold->cb[j]));
          }
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          *reinterpret_cast<float*>(wbase + (i * 16 + (lane >> 4) * 4 + rg) * EPI_PITCH(WCOLS) + (j * 16 + (lane & 15)) * 4) = v[rg];
      }
    }
    if ((pass + 1) * 2 < NI) fold_fetch(pass + 1);
  };
  big_epilogue_staged<EPI, NT, NI, AUX, FASTRM>(a, stage, wbase, m_first, n_first, lane);
}

}  // namespace ucod
