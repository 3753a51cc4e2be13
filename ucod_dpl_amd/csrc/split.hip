// Split-operand ("f32-equivalent") backbone forward: every matrix product of the ViT with its f32 operands written as a sum of bf16 terms,
//   v = v0 + v1 (+ v2),   v0 = bf16(v), v1 = bf16(v - v0), v2 = bf16(v - v0 - v1)       (the subtractions are exact in f32)
// and the product as the sum of the partial products of weight >= 2^-8 (two terms: a0 b0 + a0 b1 + a1 b0, 16 significand bits per operand) or >= 2^-16
// (three terms: + a1 b1 + a0 b2 + a2 b0, 24 bits: what an f32 FMA chain keeps anyway), each exact in the f32 accumulator of the bf16 MFMA.
//
// The partial products ride on the EXISTING bf16 GEMM kernels (gemm_bf16.hip) by concatenation along K: with P = 3 (6) products an [M, K] operand becomes
// [M, P K] bf16, segment p of the A side holding term A_TERM[p] and segment p of the B side term B_TERM[p], so that one launch with K' = P K accumulates all
// of them in f32.  Everything between the GEMMs is f32: the residual stream (UCOD_EPI_*_F32 epilogues), LayerNorm (two-pass f32, output written straight into
// the next GEMM's split A operand), exact-erf GELU, and an attention kernel on split Q, K, V and split probabilities (attn_split_kernel below).
//
// Why it exists (VERDICT r5 missing #1): the reference builds its training features with the backbone in plain fp32
// (/root/reference/data/datasets/base_dataset.py:124-138, no autocast); on trained-like weights every 16-bit op class of the fast engines carries ~1e-3 of logit
// error (profiles/r06_error_budget_f16.json) and no 16-bit configuration meets the 1e-3 bar there.  Two-term split: logits within 3e-5 of the f32 oracle at
// 3x the MFMA work; three-term: f32-equivalent at 6x.  Reference arithmetic: transformers modeling_dinov2.py:38-149,153-235,238-297,342-381 and
// data/utils/feature_extractor.py:42-59 (key hook).  bf16 build only (the f16 build refuses: its MFMAs take fp16).
#include <cmath>
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

typedef __attribute__((ext_vector_type(8))) __bf16 sb16x8;

// term held by segment p of the A-side / B-side concatenation: A {0, 0, 1, 1, 0, 2}, B {0, 1, 0, 1, 2, 0}; the first three entries are the two-term form
#define A_TERM(p) ((p) == 0 ? 0 : (p) == 1 ? 0 : (p) == 2 ? 1 : (p) == 3 ? 1 : (p) == 4 ? 0 : 2)
#define B_TERM(p) ((p) == 0 ? 0 : (p) == 1 ? 1 : (p) == 2 ? 0 : (p) == 3 ? 1 : (p) == 4 ? 2 : 0)
constexpr int products_of(int terms) { return terms == 2 ? 3 : 6; }

__device__ __forceinline__ float bf16_round(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// v -> TERMS bf16 values (as bf16_raw) whose sum is v up to 2^-17 |v| (two terms) / 2^-25 |v| (three)
template <int TERMS>
__device__ __forceinline__ void split_terms(float v, bf16_raw (&t)[TERMS]) {
  float r = v;
#pragma unroll
  for (int s = 0; s < TERMS; ++s) {
    t[s] = f32_to_bf16(r);
    r -= bf16_to_f32(t[s]);
  }
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }   // transformers ACT2FN["gelu"], modeling_dinov2.py:289

// 8 consecutive f32 values -> one 16-byte bf16 store per segment
template <int TERMS>
__device__ __forceinline__ void store_split8(const float (&v)[8], bf16_raw* __restrict__ seg0, long seg_stride, int role) {
  constexpr int P = products_of(TERMS);
  unsigned w[TERMS][4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    bf16_raw a[TERMS], b[TERMS];
    split_terms<TERMS>(v[e], a);
    split_terms<TERMS>(v[e + 1], b);
#pragma unroll
    for (int s = 0; s < TERMS; ++s) w[s][e >> 1] = (unsigned)a[s] | ((unsigned)b[s] << 16);
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const int term = role ? B_TERM(p) : A_TERM(p);
    u32x4 o = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int s = 0; s < TERMS; ++s)
      if (s == term) o = (u32x4){w[s][0], w[s][1], w[s][2], w[s][3]};
    *reinterpret_cast<u32x4*>(seg0 + (long)p * seg_stride) = o;
  }
}

// 8 consecutive f32 values -> one 16-byte bf16 store per TERM, segments in term order (the attention operands: the kernel pairs the terms itself, so a term is
// stored once -- the K-concatenated GEMM layout above repeats term 0 two or three times)
template <int TERMS>
__device__ __forceinline__ void store_terms8(const float (&v)[8], bf16_raw* __restrict__ seg0, long seg_stride) {
  unsigned w[TERMS][4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    bf16_raw a[TERMS], b[TERMS];
    split_terms<TERMS>(v[e], a);
    split_terms<TERMS>(v[e + 1], b);
#pragma unroll
    for (int s = 0; s < TERMS; ++s) w[s][e >> 1] = (unsigned)a[s] | ((unsigned)b[s] << 16);
  }
#pragma unroll
  for (int s = 0; s < TERMS; ++s) *reinterpret_cast<u32x4*>(seg0 + (long)s * seg_stride) = (u32x4){w[s][0], w[s][1], w[s][2], w[s][3]};
}

// ---- f32 [M, K] (row pitch ld_in) -> bf16 [M, P K]; op 0: the values, 1: exact-erf GELU of them, 2: times `alpha`
template <int TERMS>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ in, long ld_in, bf16_raw* __restrict__ out, int M, int K, int role, int op, float alpha) {
  constexpr int P = products_of(TERMS);
  const int k8 = K >> 3;
  const long total = (long)M * k8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / k8;
    const int c = (int)(i - m * k8) * 8;
    const float4* src = reinterpret_cast<const float4*>(in + m * ld_in + c);
    const float4 a = src[0], b = src[1];
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (op == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = gelu_exact(v[e]);
    } else if (op == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= alpha;
    }
    store_split8<TERMS>(v, out + m * (long)P * K + c, K, role);
  }
}

// ---- LayerNorm (two-pass f32, biased variance + eps: nn.LayerNorm, modeling_dinov2.py:348-381) of an f32 row, written as the split operand of the next GEMM.
// One wave per row, the row in registers; lane l holds columns 2 l + 128 i (D % 128 == 0).
template <int TERMS, int NCH>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              bf16_raw* __restrict__ out, int rows, int D, float eps, int role) {
  constexpr int P = products_of(TERMS);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float2* xr = reinterpret_cast<const float2*>(x + (size_t)row * D);
  float2 v[NCH];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    v[i] = xr[lane + 64 * i];
    s += v[i].x + v[i].y;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean;
    q += a * a + b * b;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  const float2* g2 = reinterpret_cast<const float2*>(gamma);
  const float2* b2 = reinterpret_cast<const float2*>(beta);
  bf16_raw* orow = out + (size_t)row * P * D;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const float2 g = g2[lane + 64 * i], b = b2[lane + 64 * i];
    const float o0 = (v[i].x - mean) * rstd * g.x + b.x, o1 = (v[i].y - mean) * rstd * g.y + b.y;
    bf16_raw t0[TERMS], t1[TERMS];
    split_terms<TERMS>(o0, t0);
    split_terms<TERMS>(o1, t1);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int term = role ? B_TERM(p) : A_TERM(p);
      unsigned w = 0;
#pragma unroll
      for (int sI = 0; sI < TERMS; ++sI)
        if (sI == term) w = (unsigned)t0[sI] | ((unsigned)t1[sI] << 16);
      reinterpret_cast<unsigned*>(orow + (size_t)p * D)[lane + 64 * i] = w;
    }
  }
}

// ---- img [B,C,H,W] f32 -> split patches bf16 [B gh gw, P Kpad] (A side); one thread per (patch, k pair); k >= C P P is zero padding
template <int TERMS>
__global__ __launch_bounds__(256) void im2col_split_kernel(const float* __restrict__ img, bf16_raw* __restrict__ out, int B, int C, int H, int W, int Pp, int Kpad, int gh, int gw) {
  constexpr int P = products_of(TERMS);
  const int kp = Kpad >> 1;
  const size_t total = (size_t)B * gh * gw * kp;
  const int K = C * Pp * Pp;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % kp) * 2;
    const size_t m = idx / kp;
    const int px = (int)(m % gw), py = (int)((m / gw) % gh), b = (int)(m / ((size_t)gw * gh));
    float v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = k + j;
      if (kk < K) {
        const int c = kk / (Pp * Pp), r = kk - c * Pp * Pp;
        const int dy = r / Pp, dx = r - dy * Pp;
        v[j] = img[(((size_t)b * C + c) * H + (py * Pp + dy)) * W + (px * Pp + dx)];
      } else {
        v[j] = 0.f;
      }
    }
    bf16_raw t0[TERMS], t1[TERMS];
    split_terms<TERMS>(v[0], t0);
    split_terms<TERMS>(v[1], t1);
    bf16_raw* orow = out + m * (size_t)P * Kpad + k;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int term = A_TERM(p);
      unsigned w = 0;
#pragma unroll
      for (int s = 0; s < TERMS; ++s)
        if (s == term) w = (unsigned)t0[s] | ((unsigned)t1[s] << 16);
      *reinterpret_cast<unsigned*>(orow + (size_t)p * Kpad) = w;
    }
  }
}

// ---- f32 qkv [B tok, 3 D] -> the attention kernel's operands, per (image, head) and padded to tok_pad = 32-row blocks (pad rows are zeros):
//   Qc [BH][tok_pad][TERMS 64]  the terms of Q times head_dim^-0.5 log2 e (scaled in f32, before the split), one 64-wide segment per term
//   Kc [BH][tok_pad][TERMS 64]  the terms of K likewise (the attention kernel forms the P partial products of S^T = K Q^T from them)
//   Vt [TERMS][BH][64][tok_pad]  V transposed, one plane per term (the A side of O^T = V^T P^T takes 8 consecutive keys of one channel)
// One workgroup per (32-token block, image * head).
template <int TERMS>
__global__ __launch_bounds__(256) void qkv_split_kernel(const float* __restrict__ qkv, bf16_raw* __restrict__ Qc, bf16_raw* __restrict__ Kc, bf16_raw* __restrict__ Vt,
                                                        int tok, int tok_pad, int heads, int D, float qscale) {
  __shared__ float vs[32][65];
  const int tb = blockIdx.x, bh = blockIdx.y;
  const int b = bh / heads, hd = bh - b * heads;
  const int tid = threadIdx.x;
  const int tl = tid >> 3, d8 = (tid & 7) * 8;
  const int t = tb * 32 + tl;
  const bool live = t < tok;
  float q[8], k[8], v[8];
  if (live) {
    const float* row = qkv + ((size_t)b * tok + t) * 3 * D + hd * 64 + d8;
    const float4 q0 = *reinterpret_cast<const float4*>(row), q1 = *reinterpret_cast<const float4*>(row + 4);
    const float4 k0 = *reinterpret_cast<const float4*>(row + D), k1 = *reinterpret_cast<const float4*>(row + D + 4);
    const float4 v0 = *reinterpret_cast<const float4*>(row + 2 * D), v1 = *reinterpret_cast<const float4*>(row + 2 * D + 4);
    const float qq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w}, kk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
    const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) { q[e] = qq[e] * qscale; k[e] = kk[e]; v[e] = vv[e]; }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) q[e] = k[e] = v[e] = 0.f;
  }
  const size_t orow = ((size_t)bh * tok_pad + t) * (TERMS * 64) + d8;
  store_terms8<TERMS>(q, Qc + orow, 64);
  store_terms8<TERMS>(k, Kc + orow, 64);
#pragma unroll
  for (int e = 0; e < 8; ++e) vs[tl][d8 + e] = v[e];
  __syncthreads();
  const int d = tid >> 2, tq = (tid & 3) * 8;
  unsigned w[TERMS][4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    bf16_raw a[TERMS], c[TERMS];
    split_terms<TERMS>(vs[tq + e][d], a);
    split_terms<TERMS>(vs[tq + e + 1][d], c);
#pragma unroll
    for (int s = 0; s < TERMS; ++s) w[s][e >> 1] = (unsigned)a[s] | ((unsigned)c[s] << 16);
  }
  const size_t plane = (size_t)gridDim.y * 64 * tok_pad;
#pragma unroll
  for (int s = 0; s < TERMS; ++s)
    *reinterpret_cast<u32x4*>(Vt + s * plane + ((size_t)bh * 64 + d) * tok_pad + tb * 32 + tq) = (u32x4){w[s][0], w[s][1], w[s][2], w[s][3]};
}

// ---- attention on split operands (eager_attention_forward, modeling_dinov2.py:153-179: softmax(Q K^T hd^-0.5) V), f32-equivalent.
// A wave owns 32 queries and walks the keys in blocks of 32 with an online softmax; the four waves of a workgroup (128 queries of one (image, head)) SHARE every
// K / V^T block through LDS: the 256 threads copy block kb + 1 from global memory into registers while block kb is being multiplied, store it into the other LDS
// stage behind the MFMAs and meet at one barrier per block (round 6, second form: the first read K / V^T straight from global memory in every wave -- four
// copies of every request and a load -> MFMA latency chain per block: 1.74 ms per launch at 32 x 12 x 1370 tokens).
//   S^T[key][query] = sum over the P 64 concatenated columns  Kc[key][.] Qc[query][.]        v_mfma_f32_32x32x16_bf16, A = Kc rows (LDS), B = Qc rows (registers)
// Both operands take their 16-element k slices from the same places of a row (lane half h of MFMA step 2 j + u reads elements 32 j + 16 h + 8 u .. + 7: 32
// contiguous bytes per lane and chunk), which is a permutation of the contraction index common to A and B.  The key rows of a block are read in the order
// pi(m) = m with bits 2 and 3 swapped, so that the C layout (lane (n, h), register i -> row 8 (i / 4) + 4 h + i % 4) hands lane half h of PV step s the EIGHT
// CONSECUTIVE keys 16 s + 8 h .. + 7 in registers 8 s .. 8 s + 7: the probabilities are split into bf16 terms where they sit and become the B operand of
//   O^T[d][query] += sum_key Vt[d][key] P^T[key][query]
// whose A operand is one 16-byte LDS read of a V^T row.  Softmax statistics per query = per lane column (+ one exchange with the other lane half).
// LDS rows are padded by 16 bytes (K rows: TERMS 128 + 16 B, V^T rows: 64 + 16 B): the 16 lanes of a ds_read_b128 group then start at 16 different multiples of
// four banks (strides of 68 / 100 and 20 banks) -- conflict-free by the 64-bank / 16-lane-group rule.  K and Q are held as TERMS segments (a term once), the
// kernel pairs them: S^T = sum over the P (A term, B term) pairs of K_a Q_b^T -- a third / half fewer K bytes staged and read than the GEMM layout's P segments.
template <int V>
struct IntTag { static constexpr int value = V; };

template <int TERMS>
struct AttnSplitLds {
  static constexpr int P = products_of(TERMS);
  static constexpr int KROW = TERMS * 64 + 8;                    // bf16 elements per staged K row (one 64-wide segment per TERM + 16 bytes of padding)
  static constexpr int VROW = 32 + 8;                            // bf16 elements per staged V^T row (32 keys)
  static constexpr int K_ELEMS = 32 * KROW, V_ELEMS = TERMS * 64 * VROW;
  static constexpr int STAGE = K_ELEMS + V_ELEMS;                // bf16 elements per stage
  static constexpr int K_PIECES = 32 * TERMS * 8, V_PIECES = TERMS * 64 * 4;   // 16-byte pieces per block
  static constexpr int KPT = K_PIECES / 256, VPT = V_PIECES / 256;         // per thread
};

// QT = 32-query tiles per wave; the product uses ONE.  Two tiles (64 queries per wave, 256 per workgroup) halve the LDS reads per MFMA -- every K / V^T fragment
// feeds two independent chains -- but need 332 registers, i.e. one wave per SIMD instead of two, and the softmax between the two MFMA groups then has no other
// wave to hide behind: measured 992 us against 869 us per launch before the software pipeline below, 1 006 against 813 with it (two-term form, 32 x 12 x 1370 tokens).
template <int TERMS, int QT>
__global__ __launch_bounds__(256, (TERMS == 2 && QT == 1) ? 2 : 1) void attn_split_kernel(const bf16_raw* __restrict__ Qc, const bf16_raw* __restrict__ Kc, const bf16_raw* __restrict__ Vt,
                                                         bf16_raw* __restrict__ out, int tok, int tok_pad, int heads, int D) {
  using L = AttnSplitLds<TERMS>;
  constexpr int P = L::P;
  constexpr int NCH = TERMS * 2;                                 // 32-element chunks of a TERMS * 64 row: chunk 2 t + c = half c of term t
  static_assert(L::K_PIECES % 256 == 0 && L::V_PIECES % 256 == 0, "whole pieces per thread");
  extern __shared__ __attribute__((aligned(16))) bf16_raw lds[];  // two stages of [K block | V^T block]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = (blockIdx.x * 4 + wave) * 32 * QT;
  const bool active = q0 < tok_pad;                              // (a wave past the padded queries still copies its share of every block and meets the barriers)
  const int bh = blockIdx.y;
  const int b = bh / heads, hd = bh - b * heads;
  const int n = lane & 31, h = lane >> 5;
  const size_t rowlen = (size_t)TERMS * 64;
  const size_t plane = (size_t)gridDim.y * 64 * tok_pad;
  // ---- this thread's pieces of a block: global source offsets (block 0) and LDS destinations
  const bf16_raw* ksrc[L::KPT];
  int kdst[L::KPT];
#pragma unroll
  for (int i = 0; i < L::KPT; ++i) {
    const int piece = tid + 256 * i, row = piece / (TERMS * 8), c = piece - row * (TERMS * 8);
    ksrc[i] = Kc + ((size_t)bh * tok_pad + row) * rowlen + c * 8;
    kdst[i] = row * L::KROW + c * 8;
  }
  const bf16_raw* vsrc[L::VPT];
  int vdst[L::VPT];
#pragma unroll
  for (int i = 0; i < L::VPT; ++i) {
    const int piece = tid + 256 * i, row = piece >> 2, c = piece & 3;        // row = term * 64 + d
    const int t = row >> 6, d = row & 63;
    vsrc[i] = Vt + t * plane + ((size_t)bh * 64 + d) * tok_pad + c * 8;
    vdst[i] = L::K_ELEMS + row * L::VROW + c * 8;
  }
  // Software pipeline (round 6, third form): iteration kb issues the score MFMAs of block kb + 1 AND the P V MFMAs of block kb - 1 -- three independent accumulator
  // chains, none of which depends on this iteration's vector work -- around the softmax of block kb (maximum, 17 v_exp_f32, row sum, term split), so that the matrix
  // pipe works under the wave's own exponentials instead of waiting for them.  An LDS stage therefore holds [K(kb + 1) | V^T(kb - 1)]; one basic block per iteration
  // (the last block's key mask is a select, the rescale is unconditional) with scheduling hints that spread the MFMAs through the vector instructions.
  u32x4 kreg[L::KPT], vreg[L::VPT];
  const int nkb = tok_pad >> 5;
  auto fetch = [&](int kk, int vk) {                             // K block kk (if it exists) and V^T block vk (if >= 0) into registers
    if (kk < nkb) {
#pragma unroll
      for (int i = 0; i < L::KPT; ++i) kreg[i] = *reinterpret_cast<const u32x4*>(ksrc[i] + (size_t)kk * 32 * rowlen);
    }
    if (vk >= 0) {
#pragma unroll
      for (int i = 0; i < L::VPT; ++i) vreg[i] = *reinterpret_cast<const u32x4*>(vsrc[i] + (size_t)vk * 32);
    }
  };
  auto stash = [&](int stage, bool with_v) {
    bf16_raw* base = lds + stage * L::STAGE;
#pragma unroll
    for (int i = 0; i < L::KPT; ++i) *reinterpret_cast<u32x4*>(base + kdst[i]) = kreg[i];
    if (with_v) {
#pragma unroll
      for (int i = 0; i < L::VPT; ++i) *reinterpret_cast<u32x4*>(base + vdst[i]) = vreg[i];
    }
  };
  // Q operand of this wave's queries: registers for the whole pass (a tile that starts past the padded rows re-reads the last padded row: never stored)
  sb16x8 qreg[QT][NCH][2];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    int qr = q0 + 32 * t + n;
    qr = qr < tok_pad ? qr : tok_pad - 1;
    const bf16_raw* qrow = Qc + ((size_t)bh * tok_pad + qr) * rowlen + h * 16;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      qreg[t][j][0] = *reinterpret_cast<const sb16x8*>(qrow + j * 32);
      qreg[t][j][1] = *reinterpret_cast<const sb16x8*>(qrow + j * 32 + 8);
    }
  }
  const int pin = (n & ~12) | ((n & 4) << 1) | ((n & 8) >> 1);   // pi(n)
  const int koff = pin * L::KROW + h * 16;                       // this lane's K row in a stage
  const int voff = L::K_ELEMS + n * L::VROW + 8 * h;             // this lane's V^T row (term 0, channel n) in a stage
  f32x16 o0[QT], o1[QT];
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    o0[t] = (f32x16){0};
    o1[t] = (f32x16){0};
    m_run[t] = -INFINITY;
    l_run[t] = 0.f;
  }
  auto scores = [&](const bf16_raw* stage_base, f32x16 (&s)[QT]) {
#pragma unroll
    for (int t = 0; t < QT; ++t) s[t] = (f32x16){0};
    const bf16_raw* krow = stage_base + koff;
    sb16x8 ka[NCH][2];                                           // every K term's fragments once; each feeds the products of all the Q terms it is paired with
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      ka[j][0] = *reinterpret_cast<const sb16x8*>(krow + j * 32);
      ka[j][1] = *reinterpret_cast<const sb16x8*>(krow + j * 32 + 8);
    }
#pragma unroll
    for (int pr = 0; pr < P; ++pr) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[2 * A_TERM(pr) + c][0], qreg[t][2 * B_TERM(pr) + c][0], s[t], 0, 0, 0);
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[2 * A_TERM(pr) + c][1], qreg[t][2 * B_TERM(pr) + c][1], s[t], 0, 0, 0);
        }
      }
    }
  };
  auto pv = [&](const bf16_raw* stage_base, const sb16x8 (&pp)[QT][TERMS][2]) {   // o += V^T(block in the stage) P^T(pp)
    const bf16_raw* vrow = stage_base + voff;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      sb16x8 va[TERMS][2];
#pragma unroll
      for (int tt = 0; tt < TERMS; ++tt) {
        va[tt][0] = *reinterpret_cast<const sb16x8*>(vrow + (tt * 64) * L::VROW + 16 * st);
        va[tt][1] = *reinterpret_cast<const sb16x8*>(vrow + (tt * 64 + 32) * L::VROW + 16 * st);
      }
#pragma unroll
      for (int pr = 0; pr < P; ++pr) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          o0[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[A_TERM(pr)][0], pp[t][B_TERM(pr)][st], o0[t], 0, 0, 0);
          o1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[A_TERM(pr)][1], pp[t][B_TERM(pr)][st], o1[t], 0, 0, 0);
        }
      }
    }
  };
  // Two sets of score accumulators and probability operands, swapped by unrolling the block loop by two (a copy per iteration would be 32 vector moves); the key
  // mask of the last block lives in a separate instantiation of the step (a select per score in every block would be 32 more): profiles/r06_split_attn_pmc.txt
  // counted 8.8 vector instructions per MFMA before these two changes -- the kernel was bound by them, not by the matrix pipe.
  f32x16 sA[QT], sB[QT];
  sb16x8 pA[QT][TERMS][2], pB[QT][TERMS][2];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int tt = 0; tt < TERMS; ++tt) pA[t][tt][0] = pA[t][tt][1] = pB[t][tt][0] = pB[t][tt][1] = __builtin_bit_cast(sb16x8, (u32x4){0u, 0u, 0u, 0u});
  fetch(0, -1);
  stash(1, false);                                               // K(0) alone, in the K area of stage 1
  fetch(1, -1);
  stash(0, false);                                               // stage 0 = [K(1) | zeros: there is no block -1]
  for (int i = tid; i < L::V_ELEMS / 8; i += 256) *reinterpret_cast<u32x4*>(lds + L::K_ELEMS + i * 8) = (u32x4){0u, 0u, 0u, 0u};
  __syncthreads();
  if (active) scores(lds + L::STAGE, sA);
  __syncthreads();                                               // every wave has read K(0) before iteration 0 overwrites stage 1
  // one block: scores of block kb + 1 into sn, P V of block kb - 1 with pp, softmax of block kb (scores in sc) into pn
  auto step = [&](int kb, f32x16 (&sc)[QT], f32x16 (&sn)[QT], const sb16x8 (&pp)[QT][TERMS][2], sb16x8 (&pn)[QT][TERMS][2], auto masked) {
    const bf16_raw* st_base = lds + (kb & 1) * L::STAGE;
    fetch(kb + 2, kb);                                           // for the next iteration's stage [K(kb + 2) | V^T(kb)]; in flight under this block's MFMAs
    if (active) {
      scores(st_base, sn);                                       // K(kb + 1) (in the last iteration: a stale block, result unused)
      pv(st_base, pp);                                           // V^T(kb - 1) with the previous block's probabilities
      float alpha[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        // register i of lane (n, h) = key kb * 32 + 16 (i / 8) + 8 h + (i % 8)
        if constexpr (decltype(masked)::value) {
          const int lim = tok - kb * 32;                         // keys of this block that exist
#pragma unroll
          for (int i = 0; i < 16; ++i) sc[t][i] = (16 * (i >> 3) + 8 * h + (i & 7) < lim) ? sc[t][i] : -INFINITY;
        }
        float mx = sc[t][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sc[t][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[t], mx);
        alpha[t] = __builtin_amdgcn_exp2f(m_run[t] - m_new);
        float p[16], rs = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          p[i] = __builtin_amdgcn_exp2f(sc[t][i] - m_new);
          rs += p[i];
        }
        rs += __shfl_xor(rs, 32, 64);
        l_run[t] = l_run[t] * alpha[t] + rs;
        m_run[t] = m_new;
        // probabilities -> TERMS bf16 operands per PV step (consumed by the NEXT step's pv)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          unsigned w[TERMS][4];
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            bf16_raw a[TERMS], c[TERMS];
            split_terms<TERMS>(p[8 * st + e], a);
            split_terms<TERMS>(p[8 * st + e + 1], c);
#pragma unroll
            for (int tt = 0; tt < TERMS; ++tt) w[tt][e >> 1] = (unsigned)a[tt] | ((unsigned)c[tt] << 16);
          }
#pragma unroll
          for (int tt = 0; tt < TERMS; ++tt) pn[t][tt][st] = __builtin_bit_cast(sb16x8, (u32x4){w[tt][0], w[tt][1], w[tt][2], w[tt][3]});
        }
        // hints: one MFMA, then a handful of vector instructions, over the iteration's 8 P MFMAs (LLVM's IGroupLP; groups it cannot fill are skipped)
#pragma unroll
        for (int g = 0; g < 8 * P; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        // everything above was independent of this iteration's MFMAs; the rescale is not
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[t][i] *= alpha[t]; o1[t][i] *= alpha[t]; }
      }
    }
    stash((kb + 1) & 1, true);                                   // the other stage = [K(kb + 2) | V^T(kb)]: every wave left it at the previous barrier
    __syncthreads();
  };
  using No = IntTag<0>;
  using Yes = IntTag<1>;
  if constexpr (TERMS == 2) {
    int kb = 0;
    for (; kb + 2 <= nkb - 1; kb += 2) {
      step(kb, sA, sB, pA, pB, No{});
      step(kb + 1, sB, sA, pB, pA, No{});
    }
    if (kb < nkb - 1) {                                          // one more unmasked block, then the last one
      step(kb, sA, sB, pA, pB, No{});
      step(kb + 1, sB, sA, pB, pA, Yes{});
      if (active) pv(lds + (nkb & 1) * L::STAGE, pA);            // the last block's P V (its V^T was staged by the last step)
    } else {
      step(kb, sA, sB, pA, pB, Yes{});
      if (active) pv(lds + (nkb & 1) * L::STAGE, pB);
    }
  } else {
    // three terms: two live sets of accumulators AND operands do not fit beside the 96-register Q operand (measured: 482 registers, 1 629 us against 1 346): one
    // loop body, the sets handed over by copies
    auto hand_over = [&]() {
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        sA[t] = sB[t];
#pragma unroll
        for (int tt = 0; tt < TERMS; ++tt) { pA[t][tt][0] = pB[t][tt][0]; pA[t][tt][1] = pB[t][tt][1]; }
      }
    };
    for (int kb = 0; kb < nkb - 1; ++kb) {
      step(kb, sA, sB, pA, pB, No{});
      hand_over();
    }
    step(nkb - 1, sA, sB, pA, pB, Yes{});
    if (active) pv(lds + (nkb & 1) * L::STAGE, pB);
  }
  if (!active) return;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int q = q0 + 32 * t + n;
    if (q >= tok) continue;
    const float inv = 1.0f / l_run[t];
    // O^T register i of lane (n, h): channel 32 dt + 8 (i / 4) + 4 h + i % 4 of query q -> the A-side split operand of the out-projection, row b tok + q
    bf16_raw* orow = out + ((size_t)b * tok + q) * (size_t)P * D + hd * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        unsigned w[TERMS][2];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const float x0 = (dt ? o1[t][4 * g + e] : o0[t][4 * g + e]) * inv, x1 = (dt ? o1[t][4 * g + e + 1] : o0[t][4 * g + e + 1]) * inv;
          bf16_raw a[TERMS], c[TERMS];
          split_terms<TERMS>(x0, a);
          split_terms<TERMS>(x1, c);
#pragma unroll
          for (int tt = 0; tt < TERMS; ++tt) w[tt][e >> 1] = (unsigned)a[tt] | ((unsigned)c[tt] << 16);
        }
        const int d = 32 * dt + 8 * g + 4 * h;
#pragma unroll
        for (int pr = 0; pr < P; ++pr) {
          u32x2 ow = {0u, 0u};
#pragma unroll
          for (int tt = 0; tt < TERMS; ++tt)
            if (tt == A_TERM(pr)) ow = (u32x2){w[tt][0], w[tt][1]};
          *reinterpret_cast<u32x2*>(orow + (size_t)pr * D + d) = ow;
        }
      }
    }
  }
}

inline bool terms_ok(int terms) { return terms == 2 || terms == 3; }
inline int blocks_for(long total) { const long b = (total + 255) / 256; return (int)(b < 65536 ? (b > 0 ? b : 1) : 65536); }

}  // namespace ucod

using namespace ucod;

extern "C" int ucod_split_products(int terms) { return terms_ok(terms) ? products_of(terms) : 0; }

extern "C" int ucod_split_rows(const float* in, long ld_in, void* out, int M, int K, int terms, int role, int op, float alpha, void* stream) {
  UCOD_BF16_ONLY();
  if (!in || !out || M <= 0 || K <= 0 || (K & 7) != 0 || ld_in < K || (ld_in & 3) != 0 || !terms_ok(terms) || (role != 0 && role != 1) || op < 0 || op > 2) return UCOD_EINVAL;
  UCOD_PROF(PROF_SPLIT, stream);
  const int blocks = blocks_for((long)M * (K >> 3));
  if (terms == 2) hipLaunchKernelGGL(split_rows_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, ld_in, (bf16_raw*)out, M, K, role, op, alpha);
  else hipLaunchKernelGGL(split_rows_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, ld_in, (bf16_raw*)out, M, K, role, op, alpha);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_layernorm_split(const float* x, const float* gamma, const float* beta, void* out, int rows, int D, float eps, int terms, int role, void* stream) {
  UCOD_BF16_ONLY();
  if (!x || !gamma || !beta || !out || rows <= 0 || D <= 0 || (D % 128) != 0 || !terms_ok(terms) || (role != 0 && role != 1)) return UCOD_EINVAL;
  UCOD_PROF(PROF_LN_SPLIT, stream);
  dim3 grid(cdiv(rows, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  bf16_raw* o = (bf16_raw*)out;
#define LNS_CASE(n)                                                                                                              \
  case n:                                                                                                                        \
    if (terms == 2) hipLaunchKernelGGL((layernorm_split_kernel<2, n>), grid, block, 0, s, x, gamma, beta, o, rows, D, eps, role); \
    else hipLaunchKernelGGL((layernorm_split_kernel<3, n>), grid, block, 0, s, x, gamma, beta, o, rows, D, eps, role);           \
    break;
  switch (D / 128) {
    LNS_CASE(1) LNS_CASE(2) LNS_CASE(3) LNS_CASE(4) LNS_CASE(5) LNS_CASE(6) LNS_CASE(8) LNS_CASE(10) LNS_CASE(12)
    default: return UCOD_EINVAL;
  }
#undef LNS_CASE
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_patch_im2col_split(const float* img, void* patches, int B, int C, int H, int W, int P, int Kpad, int terms, void* stream) {
  UCOD_BF16_ONLY();
  if (!img || !patches || B <= 0 || C <= 0 || P <= 0 || H % P || W % P || Kpad < C * P * P || (Kpad % 64) != 0 || !terms_ok(terms)) return UCOD_EINVAL;
  const int gh = H / P, gw = W / P;
  UCOD_PROF(PROF_IM2COL, stream);
  const int blocks = blocks_for((long)B * gh * gw * (Kpad / 2));
  if (terms == 2) hipLaunchKernelGGL(im2col_split_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, (bf16_raw*)patches, B, C, H, W, P, Kpad, gh, gw);
  else hipLaunchKernelGGL(im2col_split_kernel<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, (bf16_raw*)patches, B, C, H, W, P, Kpad, gh, gw);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" size_t ucod_attention_split_operand_bytes(int B, int tok, int heads, int terms) {
  if (B <= 0 || tok <= 0 || heads <= 0 || !terms_ok(terms)) return 0;
  const size_t tok_pad = (size_t)(tok + 31) / 32 * 32, bh = (size_t)B * heads;
  const size_t qk = bh * tok_pad * terms * 64 * 2, v = (size_t)terms * bh * 64 * tok_pad * 2;
  return 2 * qk + v;
}

extern "C" int ucod_qkv_split(const float* qkv, void* operands, int B, int tok, int heads, int terms, float qscale, void* stream) {
  UCOD_BF16_ONLY();
  if (!qkv || !operands || B <= 0 || tok <= 0 || heads <= 0 || !terms_ok(terms)) return UCOD_EINVAL;
  const int tok_pad = (tok + 31) / 32 * 32, bh = B * heads;
  if (bh > 65535) return UCOD_EINVAL;
  const size_t qk = (size_t)bh * tok_pad * terms * 64;
  bf16_raw* Qc = (bf16_raw*)operands;
  bf16_raw* Kc = Qc + qk;
  bf16_raw* Vt = Kc + qk;
  UCOD_PROF(PROF_SPLIT, stream);
  dim3 grid(tok_pad / 32, bh), block(256);
  if (terms == 2) hipLaunchKernelGGL(qkv_split_kernel<2>, grid, block, 0, (hipStream_t)stream, qkv, Qc, Kc, Vt, tok, tok_pad, heads, heads * 64, qscale);
  else hipLaunchKernelGGL(qkv_split_kernel<3>, grid, block, 0, (hipStream_t)stream, qkv, Qc, Kc, Vt, tok, tok_pad, heads, heads * 64, qscale);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_attention_split_fwd(const void* operands, void* out_split, int B, int tok, int heads, int terms, void* stream) {
  UCOD_BF16_ONLY();
  if (!operands || !out_split || B <= 0 || tok <= 0 || heads <= 0 || !terms_ok(terms)) return UCOD_EINVAL;
  const int tok_pad = (tok + 31) / 32 * 32, bh = B * heads;
  if (bh > 65535) return UCOD_EINVAL;
  const size_t qk = (size_t)bh * tok_pad * terms * 64;
  const bf16_raw* Qc = (const bf16_raw*)operands;
  const bf16_raw* Kc = Qc + qk;
  const bf16_raw* Vt = Kc + qk;
  UCOD_PROF(PROF_ATTN_SPLIT, stream);
  dim3 block(256);
  if (terms == 2) {
    constexpr size_t lds = 2 * AttnSplitLds<2>::STAGE * sizeof(bf16_raw);
    hipLaunchKernelGGL((attn_split_kernel<2, 1>), dim3(cdiv(tok_pad, 128), bh), block, lds, (hipStream_t)stream, Qc, Kc, Vt, (bf16_raw*)out_split, tok, tok_pad, heads, heads * 64);
  } else {
    constexpr size_t lds = 2 * AttnSplitLds<3>::STAGE * sizeof(bf16_raw);
    static const int once = [] { return (int)hipFuncSetAttribute((const void*)attn_split_kernel<3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }();
    if (once != 0) return once;
    hipLaunchKernelGGL((attn_split_kernel<3, 1>), dim3(cdiv(tok_pad, 128), bh), block, lds, (hipStream_t)stream, Qc, Kc, Vt, (bf16_raw*)out_split, tok, tok_pad, heads, heads * 64);
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

// ------------------------------------------------------------------------------------------------ the pass
namespace {
struct SplitPlan {
  size_t off_x, off_h, off_qkv, off_att, off_a, off_f1, off_g, off_patch, total;
  int M, tok, P;
};
inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
SplitPlan split_plan(const ucod_vit_desc* d, int terms) {
  SplitPlan p;
  const int gh = d->H / d->P, gw = d->W / d->P;
  p.tok = gh * gw + 1;
  p.M = d->B * p.tok;
  p.P = products_of(terms);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o = up256(o + bytes); return r; };
  p.off_x = take((size_t)p.M * d->D * 4);
  p.off_h = take((size_t)p.M * p.P * d->D * 2);
  p.off_qkv = take((size_t)p.M * 3 * d->D * 4);
  p.off_att = take(ucod_attention_split_operand_bytes(d->B, p.tok, d->heads, terms));
  p.off_a = take((size_t)p.M * p.P * d->D * 2);
  p.off_f1 = take(terms == 2 ? 0 : (size_t)p.M * d->F * 4);       // (two terms: fc1 + GELU + split in one launch, no f32 copy of the MLP hidden)
  p.off_g = take((size_t)p.M * p.P * d->F * 2);
  p.off_patch = take((size_t)d->B * gh * gw * p.P * d->Kpad * 2);
  p.total = o;
  return p;
}
bool split_valid(const ucod_vit_desc* d, int terms) {
  return d && terms_ok(terms) && d->B > 0 && d->C > 0 && d->P > 0 && d->H > 0 && d->W > 0 && d->H % d->P == 0 && d->W % d->P == 0 && d->D > 0 && d->heads > 0 &&
         d->D == d->heads * 64 && d->D % 128 == 0 && d->D <= 1536 && d->F % 128 == 0 && d->L >= 1 && d->Kpad % 64 == 0 && d->Kpad >= d->C * d->P * d->P &&
         d->full_last_layer == 0 && (long)d->B * d->heads <= 65535;
}
}  // namespace

#define RUN(call)                \
  do {                           \
    int rc__ = (call);           \
    if (rc__ != 0) return rc__;  \
  } while (0)

extern "C" size_t ucod_vit_split_workspace_bytes(const ucod_vit_desc* d, int terms) { return split_valid(d, terms) ? split_plan(d, terms).total : 0; }
// byte offset, inside the workspace, of the f32 residual stream x [B tok, D]: after a key-minimal pass it holds the INPUT of the pass's last layer (the last layer
// only runs LayerNorm 1 and the key hook), which is what the CLS-attention row of the pseudo-label generator is computed from
extern "C" size_t ucod_vit_split_stream_offset(const ucod_vit_desc* d, int terms) { return split_valid(d, terms) ? split_plan(d, terms).off_x : (size_t)-1; }

extern "C" int ucod_vit_forward_split(const ucod_vit_desc* d, int terms, const void* const* T, const float* img, float* key_out, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  UCOD_BF16_ONLY();
  if (!split_valid(d, terms) || !T || !img || !key_out || !workspace) return UCOD_EINVAL;
  const SplitPlan p = split_plan(d, terms);
  if (workspace_bytes < p.total) return UCOD_ENOMEM;
  char* ws = (char*)workspace;
  float* x = (float*)(ws + p.off_x);
  void* h = ws + p.off_h;
  float* qkv = (float*)(ws + p.off_qkv);
  void* att = ws + p.off_att;
  void* a = ws + p.off_a;
  float* f1 = (float*)(ws + p.off_f1);
  void* g = ws + p.off_g;
  void* patches = ws + p.off_patch;
  const int M = p.M, tok = p.tok, D = d->D, F = d->F, P = p.P, gv = d->gemm_variant;
  RUN(ucod_patch_im2col_split(img, patches, d->B, d->C, d->H, d->W, d->P, d->Kpad, terms, stream));
  RUN(ucod_gemm_bf16(UCOD_EPI_PATCH_TOKENS_F32, patches, T[0], x, d->B * (tok - 1), D, P * d->Kpad, (const float*)T[1], nullptr, nullptr, (const float*)T[3], tok, gv, stream));
  RUN(ucod_cls_rows(x, (const float*)T[2], (const float*)T[3], d->B, tok, D, stream));
  for (int l = 0; l < d->L; ++l) {
    const void* const* W = T + 4 + UCOD_VIT_LAYER_STRIDE * l;
    const bool last = (l == d->L - 1);
    RUN(ucod_layernorm_split(x, (const float*)W[0], (const float*)W[1], h, M, D, d->eps, terms, last ? 1 : 0, stream));
    if (last) {
      // key hook (feature_extractor.py:42,46-47,55-58): rows = channels (A = the K rows of the split QKV weight, A side), columns = tokens (B side)
      if (!W[14]) return UCOD_EINVAL;
      RUN(ucod_gemm_bf16(UCOD_EPI_KEY_NCHW_F32, W[14], h, key_out, D, M, P * D, (const float*)W[3] + D, nullptr, nullptr, nullptr, tok, gv, stream));
      break;
    }
    RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_F32, h, W[2], qkv, M, 3 * D, P * D, (const float*)W[3], nullptr, nullptr, nullptr, tok, gv, stream));
    RUN(ucod_qkv_split(qkv, att, d->B, tok, d->heads, terms, 0.125f * 1.4426950408889634f, stream));
    RUN(ucod_attention_split_fwd(att, a, d->B, tok, d->heads, terms, stream));
    RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_SCALE_RESID_F32, a, W[4], x, M, D, P * D, (const float*)W[5], (const float*)W[6], x, nullptr, tok, gv, stream));
    RUN(ucod_layernorm_split(x, (const float*)W[7], (const float*)W[8], h, M, D, d->eps, terms, 0, stream));
    if (terms == 2) {
      // two terms: fc1 + GELU + the split of its result in ONE launch (UCOD_EPI_BIAS_GELU_SPLIT2: the epilogue's minimax erf-GELU, |err| <= 7.1e-7, is an order of
      // magnitude below the 2^-17 of a two-term operand); three terms keep the exact-erf kernel behind an f32 round trip
      RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_GELU_SPLIT2, h, W[9], g, M, F, P * D, (const float*)W[10], nullptr, nullptr, nullptr, tok, gv, stream));
    } else {
      RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_F32, h, W[9], f1, M, F, P * D, (const float*)W[10], nullptr, nullptr, nullptr, tok, gv, stream));
      RUN(ucod_split_rows(f1, F, g, M, F, terms, 0, 1, 1.f, stream));
    }
    RUN(ucod_gemm_bf16(UCOD_EPI_BIAS_SCALE_RESID_F32, g, W[11], x, M, D, P * F, (const float*)W[12], (const float*)W[13], x, nullptr, tok, gv, stream));
  }
  return UCOD_OK;
}
