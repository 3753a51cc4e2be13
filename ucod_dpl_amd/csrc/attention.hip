// Fused multi-head attention forward for the ViT backbone, head_dim 64, no mask, no dropout
// (transformers modeling_dinov2.py:153-179 eager_attention_forward; models/backbones/dino.py:113-117).
//
// gfx950 structure (wave64, v_mfma_f32_32x32x16_bf16):
//   * one workgroup = 4 waves = 128 query rows of one (image, head); each wave owns 32 query rows;
//   * scores are computed TRANSPOSED, S^T = K Q^T, so a lane holds one query column: the online-softmax
//     running max / sum and the O rescale are lane-local (one cross-half exchange with lane^32);
//   * the S^T accumulator registers are converted to bf16 in place and fed as the B operand of
//     O^T += V^T P^T (accumulator-as-operand: no LDS round trip for P);
//   * K and V tiles (64 keys) are staged through LDS once per workgroup, double buffered, one barrier per tile:
//     attn_fwd_kernel through registers (issue-early / write-late), attn_fwd_v5_kernel (the product path: Q arrives pre-scaled
//     from the QKV epilogue) by buffer LDS-DMA with the swizzle applied to the source chunk; K is XOR-swizzled for conflict-free
//     ds_read_b128, V is consumed through ds_read_b64_tr_b16 (hardware transpose).
//   * N (=1370 tokens) is not a tile multiple: out-of-range keys read as zero and are masked to -inf in the last tile only.
// This file holds the PRODUCT kernels only: the generic-scale kernel (any caller-supplied scale), the pre-scaled-Q kernel the ViT
// driver uses, and the head_dim-96 cross-attention of the CORAL refiner.  Every experiment variant measured on the way (register-staged
// v2, 8-wave ping-pong, in-wave ping-pong, one-decision, persistent -m block, asm transpose reads) lives in
// variants/attention_lab.hip, built by `make variants` into libucod_dpl_variants.so and never loaded by the product path.
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

constexpr int HD = 64;        // head dim
constexpr int QT = 128;       // query rows per workgroup
constexpr int KT = 64;        // keys per tile
constexpr int KV_BYTES = KT * HD * 2;  // 8 KiB

__device__ __forceinline__ int swz_k(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int swz_v(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_raw* __restrict__ qkv, bf16_raw* __restrict__ out, int N,
                                                       int heads, float c /* scale*log2(e) */) {
  constexpr int VB = KV_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * (KV_BYTES + VB)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int D = heads * HD, ld = 3 * D;
  const int q0 = blockIdx.x * QT + wave * 32;
  const bf16_raw* base = qkv + (size_t)b * N * ld + head * HD;

  // Q^T B-operand fragments: lane (q = l31, half h5) holds Q[q][16s + 8*h5 .. +7]
  hx8 qf[4];
  {
    int qr = q0 + l31;
    qr = qr < N ? qr : N - 1;
    const bf16_raw* qp = base + (size_t)qr * ld + 8 * h5;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const hx8*>(qp + 16 * s);
  }

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (N + KT - 1) / KT;
  u32x4 rk[2], rv[2];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, ch = idx & 7;
      int kr = t * KT + row;
      kr = kr < N ? kr : N - 1;
      const bf16_raw* p = base + (size_t)kr * ld + ch * 8;
      rk[i] = *reinterpret_cast<const u32x4*>(p + D);
      rv[i] = *reinterpret_cast<const u32x4*>(p + 2 * D);
    }
  };
  auto lwrite = [&](int buf) {
    char* kb = smem + buf * (KV_BYTES + VB);
    char* vb = kb + KV_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, ch = idx & 7;
      *reinterpret_cast<u32x4*>(kb + row * 128 + swz_k(row, ch) * 16) = rk[i];
      *reinterpret_cast<u32x4*>(vb + row * 128 + swz_v(row, ch) * 16) = rv[i];
    }
  };

  gload(0);
  lwrite(0);
  for (int t = 0; t < nt; ++t) {
    __syncthreads();
    const bool more = (t + 1 < nt);
    if (more) gload(t + 1);
    const char* kb = smem + (t & 1) * (KV_BYTES + VB);
    const char* vb = kb + KV_BYTES;

    // ---- S^T = K Q^T  (keys on rows/registers, queries on lanes)
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kt][i] = 0.f;
      const int row = kt * 32 + l31;
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        const hx8 kf = *reinterpret_cast<const hx8*>(kb + row * 128 + swz_k(row, 2 * sd + h5) * 16);
        s[kt] = UCOD_MFMA32(kf, qf[sd], s[kt]);
      }
    }
    if (t == nt - 1) {  // wave-uniform: mask keys >= N
      const int kbase = t * KT + 4 * h5;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= N) s[kt][r] = -1e30f;
        }
    }

    // ---- online softmax (per query = per lane; the other 32 keys of this query live in lane^32)
    float mloc = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[1][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    m_run = m_new;
    float psum = 0.f;
    hx8 pb[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][8 * ks + j], c, -mc));
          psum += p;
          pb[kt][ks][j] = (half_t)p;
        }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }

    // ---- O^T += V^T P^T : A = V^T fragment, element j of half h5 <-> key 16ks + 8(j>>2) + 4*h5 + (j&3)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int key0 = kt * 32 + ks * 16 + 4 * h5;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          // ds_read_b64_tr_b16: lane i of each 16-lane group addresses row (i>>2), columns 4*(i&3)..+3 of a
          // 4 x 16 block and receives column i of the 4 rows.
          const int i16 = lane & 15, g1 = (lane >> 4) & 1;
          const int key = key0 + (i16 >> 2);
          const int dst = dt * 32 + g1 * 16 + 4 * (i16 & 3);  // first d column this lane addresses
          const int ch = dst >> 3, sub = (dst & 7) * 2;
          const char* p0 = vb + key * 128 + swz_v(key, ch) * 16 + sub;
          const char* p1 = vb + (key + 8) * 128 + swz_v(key + 8, ch) * 16 + sub;
          const hx4 lo = UCOD_TR16(p0);
          const hx4 hi = UCOD_TR16(p1);
          const hx8 vf = (hx8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[dt] = UCOD_MFMA32(vf, pb[kt][ks], o[dt]);
        }
      }

    if (more) lwrite((t + 1) & 1);
  }

  // ---- normalise and store: lane = query, registers = d (row map of the 32x32 accumulator)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < N) {
    bf16_raw* op = out + ((size_t)b * N + q) * D + head * HD + 4 * h5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w[0] = pack_h2(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        w[1] = pack_h2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * g) = w;
      }
  }
}

// two f32 -> one packed bf16x2 register: the vector convert lowers to a single v_cvt_pk_bf16_f32.  (No inline asm: the
// compiler inserts no VALU->MFMA-operand wait states behind an asm statement, and the first MFMA that consumes P would
// read stale registers.)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef hx2 bf16x2_t;        // (name kept: the packed pair of the build's 16-bit operand type, bf16 by default)
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

constexpr float DEFER_THR = 8.0f;

// row-sum accumulation of a pair of probabilities: two v_add_f32, not one v_pk_add_f32 (MI355X_MICROARCH.md: packed f32 VALU is dearer
// than two plain instructions beside MFMAs; measured here: v5 216 -> 210 us).  The file is built with -fno-slp-vectorize (Makefile),
// otherwise hipcc re-packs the two adds.
#ifndef UCOD_ATTN_SCALAR_SUM
#define UCOD_ATTN_SCALAR_SUM 1
#endif
// Round 4: the accumulators are two SCALARS -- with a two-element vector accumulator the DAG combiner still re-forms a v_pk_add_f32 for the
// pairs whose registers happen to be adjacent (2 per 32-key block), and a v_pk_*_f32 does not overlap with an MFMA in flight: it costs ~10
// cycles of matrix-pipe time (tools/probes/acc_transpose_probe.hip, "MFMA + 2 v_pk_add_f32": 438.7 -> 721.7 ns per 32 MFMAs; four plain
// v_add_f32 overlap completely).
__device__ __forceinline__ void sum_pair(float& acc0, float& acc1, const f32x2_t& e) {
#if UCOD_ATTN_SCALAR_SUM
  acc0 += e[0];
  acc1 += e[1];
#else
  f32x2_t a = {acc0, acc1};
  a += e;
  acc0 = a[0];
  acc1 = a[1];
#endif
}

template <int V> struct IntC { static constexpr int value = V; };

__device__ __forceinline__ float xhalf_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0]: lanes 0..31's value everywhere, r[1]: lanes 32..63's
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// =====================================================================================================
// The product kernel (ucod_attention_fwd with scale == 0, ucod_attention_fwd_lse): Q arrives PRE-SCALED by head_dim^-0.5 * log2(e)
// (folded into the QKV GEMM epilogue before its 16-bit rounding), so a probability is a bare v_exp_f32 of the score.
//   * the score accumulator is initialised with -m (running max, per query = per lane): S' = S - m leaves the MFMA chain ready;
//   * deferred max: the running max moves only when a 32-key block's max exceeds it by more than 2^DEFER_THR; until then there is no
//     O rescale (P <= 2^THR, f32 accumulation).  The per-block max is a v_maximum3_f32 chain over the lane's own 16 keys; the
//     cross-half exchange (one v_permlane32_swap) happens only inside the rare rescale branch;
//   * P -> 16-bit by v_cvt_pk and fed back as the MFMA B operand (accumulator-as-operand, no LDS round trip);
//   * the denominator is the per-lane f32 sum of the unrounded probabilities, two plain v_add_f32 per pair (sum_pair);
//   * K/V tiles arrive by 16-byte BUFFER loads to LDS: loop-carried 32-bit offsets, rows past the last token fail the range check
//     and read as zero; the tile loop is unrolled by two so every LDS address is a loop-invariant register plus an immediate.
// Work that cannot contribute is not executed (round 3; N = 1370 = 42.8 blocks of 32): a wave whose 32 query rows all lie past the
// last token only stages its share of K/V and keeps the barriers (1 of 44 waves per (image, head)); the last tile runs its own body with
// the key mask, which skips the second 32-key block's Q K^T, exponentials and P V when that block lies entirely past the last key
// (1370: keys 1376..1407; 1 of 44 blocks).  The main-loop body carries neither the mask nor a branch for it.
// Epilogue: lanes l and l+32 hold adjacent 4-column groups of one output row; one v_permlane32_swap per register pair turns two
// 8-byte stores at a 1.5 KB row stride into one 16-byte store (the store tail of such kernels is issue-bound per instruction).
// =====================================================================================================
__global__ __launch_bounds__(256, 2) void attn_fwd_v5_kernel(const bf16_raw* __restrict__ qkv, bf16_raw* __restrict__ out, int N, int heads,
                                                              int npairs, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_BYTES];       // [buffer][K | V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5, l31 = lane & 31;
  const int nq = (N + QT - 1) / QT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot / nq) * 8 + xcd, qt = slot - (slot / nq) * nq;
  if (pair >= npairs) return;
  const int head = pair % heads, b = pair / heads;
  const int D = heads * HD, ld = 3 * D;
  const int q0 = qt * QT + wave * 32;
  const bool live = q0 < N;                              // wave-uniform: this wave owns at least one real query row
  const bf16_raw* base = qkv + (size_t)b * N * ld + head * HD;

  hx8 qf[4];
  {
    int qr = q0 + l31;
    qr = qr < N ? qr : N - 1;
    const bf16_raw* qp = base + (size_t)qr * ld + 8 * h5;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const hx8*>(qp + 16 * s);
  }

  // this image's qkv rows as one buffer: byte offsets fit 32 bits, a key row >= N is out of range and reads as zero
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(qkv + (size_t)b * N * ld), 0, (unsigned)N * (unsigned)ld * 2u, 0x00020000);
  unsigned sk[2], sv[2];                                 // loop-carried source offsets of this lane's four DMA chunks
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3), ch = lane & 7;
    sk[i] = (unsigned)(row * ld + D + head * HD + swz_k(row, ch) * 8) * 2u;
    sv[i] = (unsigned)(row * ld + 2 * D + head * HD + swz_v(row, ch) * 8) * 2u;
  }
  const unsigned tile_step = (unsigned)(KT * ld) * 2u;
  auto stage = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* dst = smem + BUF * (2 * KV_BYTES) + (i * 32 + wave * 8) * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, sk[i], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + KV_BYTES), 16, sv[i], 0, 0, 0);
      sk[i] += tile_step;
      sv[i] += tile_step;
    }
  };

  // loop-invariant LDS byte offsets: K fragment chunk per 16-wide d step, V fragment per 32-wide d half
  int koff[4], voff[2];
#pragma unroll
  for (int sd = 0; sd < 4; ++sd) koff[sd] = l31 * 128 + swz_k(l31, 2 * sd + h5) * 16;      // +4096 per 32 keys keeps the swizzle
  {
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int key = 4 * h5 + (i16 >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int dst = dt * 32 + g1 * 16 + 4 * (i16 & 3);
      voff[dt] = KV_BYTES + key * 128 + swz_v(key, dst >> 3) * 16 + (dst & 7) * 2;          // +8/16/32 keys keep the swizzle
    }
  }

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = 0.f;
  float lsum0 = 0.f, lsum1 = 0.f;
  const int nt = (N + KT - 1) / KT, nfull = nt - 1;
  const bool half_dead = nfull * KT + 32 >= N;           // (uniform) the keys of the last tile's second 32-key block all lie past the last token

  auto tile = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    dma_landed_barrier();                                // this wave's DMAs of tile t have landed; everyone is done reading tile t-1
    if (t + 1 < nt) stage(IntC<BUF ^ 1>{});
    if (!live) return;                                   // (wave-uniform) nothing to compute for rows past the last token
    const char* kb = smem + BUF * (2 * KV_BYTES);

    f32x16 s[2];
    const float neg_m = -m_run;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (t == nfull && kt == 1 && half_dead) break;     // (wave-uniform; last tile only) no real key in the second 32-key block
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kt][i] = neg_m;
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        const hx8 kf = *reinterpret_cast<const hx8*>(kb + kt * 4096 + koff[sd]);
        s[kt] = UCOD_MFMA32(kf, qf[sd], s[kt]);
      }
    }
    if (t == nfull && (N & (KT - 1)) != 0) {
      const int kbase = t * KT + 4 * h5;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= N) s[kt][r] = -1e30f;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (t == nfull && kt == 1 && half_dead) break;
      // v_maximum3_f32 (IEEE maximum: no operand canonicalisation), two scores per instruction.  The lane's 16 keys are enough
      // for the wave-wide "does any score run away" test; the other half's keys are fetched only when the rescale fires.
      float mloc = __builtin_elementwise_maximum(s[kt][0], s[kt][1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[kt][r]), s[kt][r + 1]);
      const bool first = (t == 0 && kt == 0);
      if (first || __any(mloc > DEFER_THR)) {
        mloc = xhalf_max(mloc);
        const float delta = first ? mloc : fmaxf(mloc, 0.f);
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[kt][i] -= delta;
          if (kt == 0) s[1][i] -= delta;
          o[0][i] *= alpha;
          o[1][i] *= alpha;
        }
        lsum0 *= alpha;
        lsum1 *= alpha;
      }
      hx8 pb[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 w;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x2_t e = {__builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj + 1])};
          sum_pair(lsum0, lsum1, e);
          w[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
        }
        pb[ks] = __builtin_bit_cast(hx8, w);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* p0 = kb + (kt * 32 + ks * 16) * 128 + voff[dt];
          const hx4 lo = UCOD_TR16(p0);
          const hx4 hi = UCOD_TR16(p0 + 8 * 128);
          const hx8 vf = (hx8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[dt] = UCOD_MFMA32(vf, pb[ks], o[dt]);
        }
    }
  };

  // two tiles per iteration: the LDS buffer index is a compile-time constant
  stage(IntC<0>{});
  for (int t = 0; t < nt; t += 2) {
    tile(t, IntC<0>{});
    if (t + 1 < nt) tile(t + 1, IntC<1>{});
  }
  if (!live) return;

  const float lane_sum = lsum0 + lsum1;
  const float denom = lane_sum + __shfl_xor(lane_sum, 32, 64);
  const float inv = 1.0f / denom;
  const int q = q0 + l31;
  if (lse && h5 == 0 && q < N) lse[((size_t)b * heads + head) * N + q] = m_run + __builtin_amdgcn_logf(denom);
  // o[dt][4g .. 4g+3] = columns dt*32 + 8g + 4*h5 .. +3 of row q.  For each register pair (g, g+1) one half exchange leaves lanes 0..31
  // with columns 8g .. 8g+7 ([own g | upper's g]) and lanes 32..63 with 8(g+1) .. 8(g+1)+7 ([lower's g+1 | own g+1]): one 16-byte store each.
  const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(out + ((size_t)b * N) * D, 0, (unsigned)N * (unsigned)D * 2u, 0x00020000);
  // rows past N: an offset that stays beyond the buffer after the per-store constants are added (records = N * D * 2 < 2^31), so the
  // range check drops the store -- 0xFFFFFFF0 would WRAP to the image's first row
  const unsigned row_off = q < N ? ((unsigned)q * (unsigned)D + (unsigned)(head * HD + 8 * h5)) * 2u : 0x80000000u;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int g = 2 * gp;
      unsigned a0 = cvt_pk_bf16(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv), a1 = cvt_pk_bf16(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
      unsigned b0 = cvt_pk_bf16(o[dt][4 * g + 4] * inv, o[dt][4 * g + 5] * inv), b1 = cvt_pk_bf16(o[dt][4 * g + 6] * inv, o[dt][4 * g + 7] * inv);
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      const u32x4 w = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
      __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, row_off + (unsigned)(dt * 32 + 16 * gp) * 2u, 0, 0);
    }
}

// =====================================================================================================
// v6 (round 4, late): the v5 kernel with 64 query rows per wave (two 32-row blocks) and 256 rows per workgroup, two workgroups per CU =
// two waves per SIMD.  Why: with 32 rows per wave and four waves per SIMD every wave re-reads the whole K / V tile from LDS -- 256 KB per
// 64 keys and CU = the matrix time of those keys at 128 B/clk -- so v5 is bound by LDS as much as by its vector work; here a K or V
// fragment read serves two MFMAs (128 KB per 64 keys and CU) and the two resident workgroups run unsynchronised, so one wave's
// exponentials sit beside the other's MFMAs.  Same arithmetic per element as v5 (deferred max per 32-row block, scalar row sums).
// =====================================================================================================
__global__ __launch_bounds__(256, 2) void attn_fwd_v6_kernel(const bf16_raw* __restrict__ qkv, bf16_raw* __restrict__ out, int N, int heads,
                                                              int npairs, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_BYTES];       // [buffer][K | V]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5, l31 = lane & 31;
  constexpr int QT6 = 256, RB = 2;                       // 4 waves x 64 query rows = two 32-row blocks per wave
  const int nq = (N + QT6 - 1) / QT6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot / nq) * 8 + xcd, qt = slot - (slot / nq) * nq;
  if (pair >= npairs) return;
  const int head = pair % heads, b = pair / heads;
  const int D = heads * HD, ld = 3 * D;
  const int q0 = qt * QT6 + wave * 64;
  const bool live = q0 < N;                              // wave-uniform: this wave owns at least one real query row
  const bool live1 = q0 + 32 < N;                        // ... and its second 32-row block too
  const bf16_raw* base = qkv + (size_t)b * N * ld + head * HD;

  hx8 qf[RB][4];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int qr = q0 + rb * 32 + l31;
    qr = qr < N ? qr : N - 1;
    const bf16_raw* qp = base + (size_t)qr * ld + 8 * h5;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[rb][s] = *reinterpret_cast<const hx8*>(qp + 16 * s);
  }

  // this image's qkv rows as one buffer: byte offsets fit 32 bits, a key row >= N is out of range and reads as zero
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(qkv + (size_t)b * N * ld), 0, (unsigned)N * (unsigned)ld * 2u, 0x00020000);
  unsigned sk[2], sv[2];                                 // loop-carried source offsets of this lane's four DMA chunks
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3), ch = lane & 7;
    sk[i] = (unsigned)(row * ld + D + head * HD + swz_k(row, ch) * 8) * 2u;
    sv[i] = (unsigned)(row * ld + 2 * D + head * HD + swz_v(row, ch) * 8) * 2u;
  }
  const unsigned tile_step = (unsigned)(KT * ld) * 2u;
  auto stage = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* dst = smem + BUF * (2 * KV_BYTES) + (i * 32 + wave * 8) * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, sk[i], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + KV_BYTES), 16, sv[i], 0, 0, 0);
      sk[i] += tile_step;
      sv[i] += tile_step;
    }
  };

  // loop-invariant LDS byte offsets: K fragment chunk per 16-wide d step, V fragment per 32-wide d half
  int koff[4], voff[2];
#pragma unroll
  for (int sd = 0; sd < 4; ++sd) koff[sd] = l31 * 128 + swz_k(l31, 2 * sd + h5) * 16;      // +4096 per 32 keys keeps the swizzle
  {
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int key = 4 * h5 + (i16 >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int dst = dt * 32 + g1 * 16 + 4 * (i16 & 3);
      voff[dt] = KV_BYTES + key * 128 + swz_v(key, dst >> 3) * 16 + (dst & 7) * 2;          // +8/16/32 keys keep the swizzle
    }
  }

  f32x16 o[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[rb][0][i] = 0.f; o[rb][1][i] = 0.f; }
  float m_run[RB] = {0.f, 0.f};
  float lsum0[RB] = {0.f, 0.f}, lsum1[RB] = {0.f, 0.f};
  const int nt = (N + KT - 1) / KT, nfull = nt - 1;
  const bool half_dead = nfull * KT + 32 >= N;           // (uniform) the keys of the last tile's second 32-key block all lie past the last token

  auto tile = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    dma_landed_barrier();                                // this wave's DMAs of tile t have landed; everyone is done reading tile t-1
    if (t + 1 < nt) stage(IntC<BUF ^ 1>{});
    if (!live) return;                                   // (wave-uniform) nothing to compute for rows past the last token
    const char* kb = smem + BUF * (2 * KV_BYTES);

    // one 32-key block at a time: its K fragments serve both row blocks, its V fragments too (half the LDS reads per MFMA of the 32-row kernel)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (t == nfull && kt == 1 && half_dead) break;     // (wave-uniform; last tile only) no real key in the second 32-key block
      f32x16 s[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < 16; ++i) s[rb][i] = -m_run[rb];
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        const hx8 kf = *reinterpret_cast<const hx8*>(kb + kt * 4096 + koff[sd]);
        s[0] = UCOD_MFMA32(kf, qf[0][sd], s[0]);
        if (live1) s[1] = UCOD_MFMA32(kf, qf[1][sd], s[1]);
      }
      if (t == nfull && (N & (KT - 1)) != 0) {
        const int kbase = t * KT + kt * 32 + 4 * h5;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kbase + (r & 3) + 8 * (r >> 2);
            if (key >= N) s[rb][r] = -1e30f;
          }
      }
      hx8 pb[RB][2];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if (rb == 1 && !live1) break;
        float mloc = __builtin_elementwise_maximum(s[rb][0], s[rb][1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[rb][r]), s[rb][r + 1]);
        const bool first = (t == 0 && kt == 0);
        if (first || __any(mloc > DEFER_THR)) {
          mloc = xhalf_max(mloc);
          const float delta = first ? mloc : fmaxf(mloc, 0.f);
          const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
          m_run[rb] += delta;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            s[rb][i] -= delta;
            o[rb][0][i] *= alpha;
            o[rb][1][i] *= alpha;
          }
          lsum0[rb] *= alpha;
          lsum1[rb] *= alpha;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 w;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x2_t e = {__builtin_amdgcn_exp2f(s[rb][8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(s[rb][8 * ks + 2 * jj + 1])};
            sum_pair(lsum0[rb], lsum1[rb], e);
            w[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
          }
          pb[rb][ks] = __builtin_bit_cast(hx8, w);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* p0 = kb + (kt * 32 + ks * 16) * 128 + voff[dt];
          const hx4 lo = UCOD_TR16(p0);
          const hx4 hi = UCOD_TR16(p0 + 8 * 128);
          const hx8 vf = (hx8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[0][dt] = UCOD_MFMA32(vf, pb[0][ks], o[0][dt]);
          if (live1) o[1][dt] = UCOD_MFMA32(vf, pb[1][ks], o[1][dt]);
        }
    }
  };

  // two tiles per iteration: the LDS buffer index is a compile-time constant
  stage(IntC<0>{});
  for (int t = 0; t < nt; t += 2) {
    tile(t, IntC<0>{});
    if (t + 1 < nt) tile(t + 1, IntC<1>{});
  }
  if (!live) return;

  const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(out + ((size_t)b * N) * D, 0, (unsigned)N * (unsigned)D * 2u, 0x00020000);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    if (rb == 1 && !live1) break;
    const float lane_sum = lsum0[rb] + lsum1[rb];
    const float denom = lane_sum + __shfl_xor(lane_sum, 32, 64);
    const float inv = 1.0f / denom;
    const int q = q0 + rb * 32 + l31;
    if (lse && h5 == 0 && q < N) lse[((size_t)b * heads + head) * N + q] = m_run[rb] + __builtin_amdgcn_logf(denom);
    const unsigned row_off = q < N ? ((unsigned)q * (unsigned)D + (unsigned)(head * HD + 8 * h5)) * 2u : 0x80000000u;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int g = 2 * gp;
        const f32x16& oo = o[rb][dt];
        unsigned a0 = cvt_pk_bf16(oo[4 * g + 0] * inv, oo[4 * g + 1] * inv), a1 = cvt_pk_bf16(oo[4 * g + 2] * inv, oo[4 * g + 3] * inv);
        unsigned b0 = cvt_pk_bf16(oo[4 * g + 4] * inv, oo[4 * g + 5] * inv), b1 = cvt_pk_bf16(oo[4 * g + 6] * inv, oo[4 * g + 7] * inv);
        const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        const u32x4 w = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
        __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, row_off + (unsigned)(dt * 32 + 16 * gp) * 2u, 0, 0);
      }
  }
}

// =====================================================================================================
// Cross-attention, head_dim 96 (CORAL refiner: nn.MultiheadAttention with 8 heads on C=768, models/modules/mlp.py:122,143).
// Same arithmetic as the pre-scaled-Q kernel above (pre-scaled Q, accumulator initialised with -m, per-half deferred max, denominator on
// the matrix pipe, V through ds_read_b64_tr_b16) with separate query / key-value sources and lengths.  LDS rows are padded
// to 256 B (16 slots of 16 B, 12 used): K slot = chunk ^ (row & 15), V slot = chunk ^ ((row & 3) << 2).
// =====================================================================================================
constexpr int HDX = 96;
constexpr int XROW = 256;                 // padded LDS row bytes
constexpr int XTILE = KT * XROW;          // 16 KiB per K or V tile

__global__ __launch_bounds__(256, 2) void attn_cross96_kernel(const bf16_raw* __restrict__ qp_, int ldq, const bf16_raw* __restrict__ kp_,
                                                               const bf16_raw* __restrict__ vp_, int ldkv, bf16_raw* __restrict__ out,
                                                               int Nq, int Nk, int heads) {
  __shared__ __attribute__((aligned(16))) char smem[4 * XTILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int D = heads * HDX;
  const int q0 = blockIdx.x * QT + wave * 32;
  const bf16_raw* qb = qp_ + (size_t)b * Nq * ldq + head * HDX;
  const bf16_raw* kbase = kp_ + (size_t)b * Nk * ldkv + head * HDX;
  const bf16_raw* vbase = vp_ + (size_t)b * Nk * ldkv + head * HDX;

  hx8 qf[6];
  {
    int qr = q0 + l31;
    qr = qr < Nq ? qr : Nq - 1;
    const bf16_raw* qp = qb + (size_t)qr * ldq + 8 * h5;
#pragma unroll
    for (int s = 0; s < 6; ++s) qf[s] = *reinterpret_cast<const hx8*>(qp + 16 * s);
  }
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const hx8 ones = __builtin_bit_cast(hx8, ones_u);

  f32x16 o[3], osum;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; o[2][i] = 0.f; osum[i] = 0.f; }
  float m_run = 0.f;

  const int nt = (Nk + KT - 1) / KT;
  u32x4 rk[3], rv[3];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + 256 * i, row = idx / 12, ch = idx - row * 12;
      int kr = t * KT + row;
      kr = kr < Nk ? kr : Nk - 1;
      rk[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)kr * ldkv + ch * 8);
      rv[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)kr * ldkv + ch * 8);
    }
  };
  auto lwrite = [&](int buf) {
    char* kb = smem + buf * (2 * XTILE);
    char* vb = kb + XTILE;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + 256 * i, row = idx / 12, ch = idx - row * 12;
      *reinterpret_cast<u32x4*>(kb + row * XROW + (ch ^ (row & 15)) * 16) = rk[i];
      *reinterpret_cast<u32x4*>(vb + row * XROW + (ch ^ ((row & 3) << 2)) * 16) = rv[i];
    }
  };
  int koff[2][6], voff[2][2][3];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int sd = 0; sd < 6; ++sd) {
      const int row = kt * 32 + l31;
      koff[kt][sd] = row * XROW + ((2 * sd + h5) ^ (row & 15)) * 16;
    }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dt = 0; dt < 3; ++dt) {
        const int i16 = lane & 15, g1 = (lane >> 4) & 1;
        const int key = kt * 32 + ks * 16 + 4 * h5 + (i16 >> 2);
        const int dst = dt * 32 + g1 * 16 + 4 * (i16 & 3);
        voff[kt][ks][dt] = XTILE + key * XROW + (((dst >> 3)) ^ ((key & 3) << 2)) * 16 + (dst & 7) * 2;   // key+8: same (key&3) -> +8*XROW
      }

  gload(0);
  lwrite(0);
  for (int t = 0; t < nt; ++t) {
    __syncthreads();
    const bool more = (t + 1 < nt);
    if (more) gload(t + 1);
    const char* kb = smem + (t & 1) * (2 * XTILE);
    f32x16 s[2];
    const float neg_m = -m_run;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kt][i] = neg_m;
#pragma unroll
      for (int sd = 0; sd < 6; ++sd) {
        const hx8 kf = *reinterpret_cast<const hx8*>(kb + koff[kt][sd]);
        s[kt] = UCOD_MFMA32(kf, qf[sd], s[kt]);
      }
    }
    if (t == nt - 1 && (Nk & (KT - 1)) != 0) {
      const int kbase_i = t * KT + 4 * h5;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase_i + kt * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= Nk) s[kt][r] = -1e30f;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float mloc = s[kt][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[kt][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const bool first = (t == 0 && kt == 0);
      if (first || __any(mloc > DEFER_THR)) {
        const float delta = first ? mloc : fmaxf(mloc, 0.f);
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[kt][i] -= delta;
          if (kt == 0) s[1][i] -= delta;
          o[0][i] *= alpha;
          o[1][i] *= alpha;
          o[2][i] *= alpha;
        }
        osum[0] *= alpha;
      }
      hx8 pb[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4_t w;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          w[jj] = cvt_pk_bf16(__builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj + 1]));
        pb[ks] = __builtin_bit_cast(hx8, w);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        osum = UCOD_MFMA32(ones, pb[ks], osum);
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
          const char* p0 = kb + voff[kt][ks][dt];
          const char* p1 = p0 + 8 * XROW;
          const hx4 lo = UCOD_TR16(p0);
          const hx4 hi = UCOD_TR16(p1);
          const hx8 vf = (hx8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[dt] = UCOD_MFMA32(vf, pb[ks], o[dt]);
        }
      }
    }
    if (more) lwrite((t + 1) & 1);
  }
  const float inv = 1.0f / osum[0];
  const int q = q0 + l31;
  if (q < Nq) {
    bf16_raw* op = out + ((size_t)b * Nq + q) * D + head * HDX + 4 * h5;
#pragma unroll
    for (int dt = 0; dt < 3; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w[0] = cvt_pk_bf16(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        w[1] = cvt_pk_bf16(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * g) = w;
      }
  }
}

}  // namespace ucod

// variant: 0 / 2 = the product kernels (generic-scale when scale != 0, pre-scaled-Q when scale == 0).  Every other number is an
// experiment variant of variants/attention_lab.hip (libucod_dpl_variants.so: ucod_attention_fwd_lab) and is refused here.
extern "C" int ucod_attention_fwd(const void* qkv, void* out, int B, int tok, int heads, float scale, int variant, void* stream) {
  using namespace ucod;
  if (!qkv || !out || B <= 0 || tok <= 0 || heads <= 0) return UCOD_EINVAL;
  if (variant != 0 && variant != 2 && variant != 5 && variant != 66) return UCOD_EINVAL;      // 5 = attn_fwd_v5_kernel by name (what 0 / 2 select with scale == 0)
  if ((size_t)tok * heads * HD * 3 * 2 >= (1ull << 32)) return UCOD_EINVAL;       // per-image qkv rows are addressed with 32-bit byte offsets
  UCOD_PROF(PROF_ATTN, stream);
  if (variant == 66 && scale != 0.f) return UCOD_EINVAL;         // 66 = attn_fwd_v6_kernel by name (64 query rows per wave)
  if (scale == 0.f && variant == 66) {
    const int npairs = B * heads, nq6 = cdiv(tok, 256);
    hipLaunchKernelGGL(attn_fwd_v6_kernel, dim3(cdiv(npairs, 8) * 8 * nq6), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads,
                       npairs, (float*)nullptr);
  } else if (scale == 0.f) {   // Q pre-scaled by head_dim^-0.5 * log2(e)
    const int npairs = B * heads, nq = cdiv(tok, QT);
    hipLaunchKernelGGL(attn_fwd_v5_kernel, dim3(cdiv(npairs, 8) * 8 * nq), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads,
                       npairs, (float*)nullptr);
  } else {
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(tok, QT), heads, B), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads,
                       scale * 1.4426950408889634f);
  }
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_cross_attention96_fwd(const void* q, int ldq, const void* k, const void* v, int ldkv, void* out, int B, int Nq, int Nk,
                                          int heads, void* stream) {
  using namespace ucod;
  if (!q || !k || !v || !out || B <= 0 || Nq <= 0 || Nk <= 0 || heads <= 0 || (ldq % 8) || (ldkv % 8)) return UCOD_EINVAL;
  UCOD_PROF(PROF_ATTN, stream);
  dim3 grid(cdiv(Nq, QT), heads, B), block(256);
  hipLaunchKernelGGL(attn_cross96_kernel, grid, block, 0, (hipStream_t)stream, (const bf16_raw*)q, ldq, (const bf16_raw*)k, (const bf16_raw*)v,
                     ldkv, (bf16_raw*)out, Nq, Nk, heads);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_attention_fwd_lse(const void* qkv, void* out, float* lse, int B, int tok, int heads, void* stream) {
  using namespace ucod;
  if (!qkv || !out || !lse || B <= 0 || tok <= 0 || heads <= 0) return UCOD_EINVAL;
  if ((size_t)tok * heads * HD * 3 * 2 >= (1ull << 32)) return UCOD_EINVAL;
  UCOD_PROF(PROF_ATTN, stream);
  const int npairs = B * heads, nq = cdiv(tok, QT);
  hipLaunchKernelGGL(attn_fwd_v5_kernel, dim3(cdiv(npairs, 8) * 8 * nq), dim3(256), 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs,
                     lse);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
