// LABORATORY FILE -- not part of the product library.  Built only by `make -C ucod_dpl_amd/csrc variants` into
// ../_native/libucod_dpl_variants.so (entry: ucod_attention_fwd_lab), loaded only by tools/ and by tests marked `variants`.
// It keeps every attention-forward experiment of rounds 1-2 runnable for A/B measurements against the product kernel of ../attention.hip
// (DESIGN.md section 4 has the measurements): variant 1 explicit-transpose V image, 3/4/6 the v2 kernel (register staging / LDS-DMA /
// f32 row sums), 2 the ROUND-2 product kernel as it was (baseline of the round-3 comparison), 7 one rescale decision per tile, 9/12 the
// 8-wave ping-pong, 13 persistent -m block, 14 in-wave ping-pong at one wave per SIMD, 15 asm transpose reads.
//
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
//     attn_fwd_kernel and variant 3 of the v2 kernel through registers (issue-early / write-late), the default v2 kernel
//     by LDS-DMA with the swizzle applied to the source chunk; K is XOR-swizzled for conflict-free ds_read_b128, V is
//     consumed through ds_read_b64_tr_b16 (hardware transpose).
//   * N (=1370 tokens) is not a tile multiple: out-of-range keys are clamped on load and masked to -inf.
#include "../common.h"
#include "../../../include/ucod_dpl.h"

namespace ucod {

constexpr int HD = 64;        // head dim
constexpr int QT = 128;       // query rows per workgroup
constexpr int KT = 64;        // keys per tile
constexpr int KV_BYTES = KT * HD * 2;  // 8 KiB
constexpr int VT_STRIDE = 72;          // bf16 elements per row of the explicit-transpose V image (VMODE 1)
constexpr int VT_BYTES = HD * VT_STRIDE * 2;

__device__ __forceinline__ int swz_k(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int swz_v(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }

template <int VMODE>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16_raw* __restrict__ qkv, bf16_raw* __restrict__ out, int N,
                                                       int heads, float c /* scale*log2(e) */) {
  constexpr int VB = (VMODE == 0) ? KV_BYTES : VT_BYTES;
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
      if constexpr (VMODE == 0) {
        *reinterpret_cast<u32x4*>(vb + row * 128 + swz_v(row, ch) * 16) = rv[i];
      } else {
        bf16_raw* vt = reinterpret_cast<bf16_raw*>(vb);
        const unsigned w[4] = {rv[i][0], rv[i][1], rv[i][2], rv[i][3]};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bf16_raw val = (bf16_raw)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
          vt[(ch * 8 + e) * VT_STRIDE + row] = val;
        }
      }
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
          hx8 vf;
          if constexpr (VMODE == 0) {
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
            vf = (hx8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          } else {
            const bf16_raw* vt = reinterpret_cast<const bf16_raw*>(vb) + (dt * 32 + l31) * VT_STRIDE + key0;
            const hx4 lo = *reinterpret_cast<const hx4*>(vt);
            const hx4 hi = *reinterpret_cast<const hx4*>(vt + 8);
            vf = (hx8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
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


// =====================================================================================================
// v2: VALU-lean variant (the first kernel spends 2/3 of its issue slots on softmax VALU work, 25 % MFMA busy).
//   * Q arrives PRE-SCALED by head_dim^-0.5 * log2(e) (folded into the QKV GEMM epilogue before its bf16
//     rounding), so probabilities are a bare v_exp_f32 of the score;
//   * the score accumulator is initialised with -m (running max, per query = per lane), so S' = S - m leaves the
//     MFMA chain ready: no per-element subtract;
//   * deferred max: the running max moves only when a tile's max exceeds it by more than THR (log2 units); until then
//     there is no O rescale and no exponent bookkeeping (P <= 2^THR, f32 accumulation);
//   * the softmax denominator is accumulated on the matrix pipe: O_sum += 1^T P^T (an all-ones A fragment), 4 extra
//     MFMAs per tile instead of 32 v_add (MFMA has slack, VALU does not);
//   * P is packed to bf16 with v_cvt_pk_bf16_f32 (one instruction per register pair).
// =====================================================================================================
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
__device__ __forceinline__ void sum_pair(f32x2_t& acc, const f32x2_t& e) {
#if UCOD_ATTN_SCALAR_SUM
  acc[0] += e[0];
  acc[1] += e[1];
#else
  acc += e;
#endif
}


template <bool DMA, bool VSUM = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_v2_kernel(const bf16_raw* __restrict__ qkv, bf16_raw* __restrict__ out, int N,
                                                              int heads, int npairs, float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) char smem[2 * (KV_BYTES + KV_BYTES)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h5 = lane >> 5, l31 = lane & 31;
  // XCD-aware 1-D grid: blocks i and i+8 share an XCD (and its 4 MiB L2), so all query tiles of one (image, head) are
  // given to ONE XCD back to back -- its K/V (350 KB at N=1370) is then re-read from that L2 instead of the fabric
  // (rocprofv3 FETCH_SIZE was 5.6x the algorithmic bytes with the natural (q-tile, head, image) order).
  const int nq = (N + QT - 1) / QT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot / nq) * 8 + xcd, qt = slot - (slot / nq) * nq;
  if (pair >= npairs) return;
  const int head = pair % heads, b = pair / heads;
  const int D = heads * HD, ld = 3 * D;
  const int q0 = qt * QT + wave * 32;
  const bf16_raw* base = qkv + (size_t)b * N * ld + head * HD;

  hx8 qf[4];
  {
    int qr = q0 + l31;
    qr = qr < N ? qr : N - 1;
    const bf16_raw* qp = base + (size_t)qr * ld + 8 * h5;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const hx8*>(qp + 16 * s);
  }
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const u32x4_t ones_u = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  const hx8 ones = __builtin_bit_cast(hx8, ones_u);

  f32x16 o[2], osum;
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; osum[i] = 0.f; }
  float m_run = 0.f;
  float lsum = 0.f;                                    // VSUM: this lane's share of the denominator (f32 adds instead of the all-ones MFMA)

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
    char* kb = smem + buf * (2 * KV_BYTES);
    char* vb = kb + KV_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, ch = idx & 7;
      *reinterpret_cast<u32x4*>(kb + row * 128 + swz_k(row, ch) * 16) = rk[i];
      *reinterpret_cast<u32x4*>(vb + row * 128 + swz_v(row, ch) * 16) = rv[i];
    }
  };

  // DMA form: K/V tiles go HBM -> LDS by 16-byte LDS-DMA (no staging registers, no ds_write, no VALU): a wave instruction fills
  // 8 rows (64 lanes x 16 B, lane-linear in LDS), so the bank swizzle is applied to the SOURCE chunk each lane fetches (both
  // swizzles are XORs, hence their own inverses).  Wave w stages rows 8w..8w+7 and 32+8w..32+8w+7 of K and of V.
  auto stage = [&](int t, int buf) {
    char* kb = smem + buf * (2 * KV_BYTES);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 32 + wave * 8 + (lane >> 3), slot = lane & 7;
      int kr = t * KT + row;
      kr = kr < N ? kr : N - 1;
      const bf16_raw* g = base + (size_t)kr * ld;
      const bf16_raw* gk = g + D + swz_k(row, slot) * 8;
      const bf16_raw* gv = g + 2 * D + swz_v(row, slot) * 8;
      char* dst = kb + (i * 32 + wave * 8) * 128;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gk, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gv, (__attribute__((address_space(3))) void*)(dst + KV_BYTES), 16, 0,
                                       0);
    }
  };

  // loop-invariant LDS byte offsets of this lane's fragments (relative to the tile buffer)
  int koff[2][4], voff[2][2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int sd = 0; sd < 4; ++sd) {
      const int row = kt * 32 + l31;
      koff[kt][sd] = row * 128 + swz_k(row, 2 * sd + h5) * 16;
    }
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int i16 = lane & 15, g1 = (lane >> 4) & 1;
        const int key = kt * 32 + ks * 16 + 4 * h5 + (i16 >> 2);
        const int dst = dt * 32 + g1 * 16 + 4 * (i16 & 3);
        voff[kt][ks][dt] = KV_BYTES + key * 128 + swz_v(key, dst >> 3) * 16 + (dst & 7) * 2;   // key+8 keeps the swizzle: +1024
      }

  if constexpr (DMA) {
    stage(0, 0);
  } else {
    gload(0);
    lwrite(0);
  }
  for (int t = 0; t < nt; ++t) {
    if constexpr (DMA) dma_landed_barrier();             // this wave's DMAs of tile t, then everyone's
    else __syncthreads();
    const bool more = (t + 1 < nt);
    if constexpr (DMA) {
      if (more) stage(t + 1, (t + 1) & 1);               // the other buffer: every wave is past its reads of tile t-1
    } else {
      if (more) gload(t + 1);
    }
    const char* kb = smem + (t & 1) * (2 * KV_BYTES);

    // S' = K Q^T - m_run  (accumulator initialised with the row constant)
    f32x16 s[2];
    const float neg_m = -m_run;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kt][i] = neg_m;
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        const hx8 kf = *reinterpret_cast<const hx8*>(kb + koff[kt][sd]);
        s[kt] = UCOD_MFMA32(kf, qf[sd], s[kt]);
      }
    }
    if (t == nt - 1 && (N & (KT - 1)) != 0) {
      const int kbase = t * KT + 4 * h5;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= N) s[kt][r] = -1e30f;
        }
    }
    // The 64-key tile is consumed as two independent 32-key halves, each with its own (cheap) deferred-max check:
    // while the VALU runs max/exp/cvt of half 0 the matrix pipe is still executing the QK^T MFMAs of half 1, and while it
    // runs half 1's softmax the pipe executes half 0's PV products -- MFMA || VALU overlap inside ONE wave.
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float mloc = s[kt][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, s[kt][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      // rescale only when some query's max ran away by more than THR (always on the very first half: m_run is a guess)
      const bool first = (t == 0 && kt == 0);
      if (first || __any(mloc > DEFER_THR)) {
        const float delta = first ? mloc : fmaxf(mloc, 0.f);
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[kt][i] -= delta;
          if (kt == 0) s[1][i] -= delta;       // half 1 was accumulated against the old running max
          o[0][i] *= alpha;
          o[1][i] *= alpha;
        }
        osum[0] *= alpha;
        lsum *= alpha;
      }
      hx8 pb[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4_t w;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const float e0 = __builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj]), e1 = __builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj + 1]);
          if constexpr (VSUM) lsum += e0 + e1;
          w[jj] = cvt_pk_bf16(e0, e1);
        }
        pb[ks] = __builtin_bit_cast(hx8, w);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (!VSUM) osum = UCOD_MFMA32(ones, pb[ks], osum);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* p0 = kb + voff[kt][ks][dt];
          const char* p1 = p0 + 8 * 128;
          const hx4 lo = UCOD_TR16(p0);
          const hx4 hi = UCOD_TR16(p1);
          const hx8 vf = (hx8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[dt] = UCOD_MFMA32(vf, pb[ks], o[dt]);
        }
      }
    }
    if constexpr (!DMA) {
      if (more) lwrite((t + 1) & 1);
    }
  }

  if constexpr (VSUM) osum[0] = lsum + __shfl_xor(lsum, 32, 64);
  const float inv = 1.0f / osum[0];
  const int q = q0 + l31;
  if (q < N) {
    // training mode: base-2 log-sum-exp of the scaled scores, consumed by ucod_attention_bwd
    if (lse && h5 == 0) lse[((size_t)b * heads + head) * N + q] = m_run + __builtin_amdgcn_logf(osum[0]);
    bf16_raw* op = out + ((size_t)b * N + q) * D + head * HD + 4 * h5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w[0] = cvt_pk_bf16(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        w[1] = cvt_pk_bf16(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * g) = w;
      }
  }
}


// =====================================================================================================
// v5: the v2 kernel with its VALU issue stream trimmed (rocprofv3 PMC: v2 is VALU-issue-bound, 0.56 matrix-pipe busy; of
// its ~190 VALU instructions per 64-key tile and wave about a third were address arithmetic, not softmax):
//   * K/V tiles staged by BUFFER loads to LDS: the per-lane byte offset is loop-carried (one v_add per DMA and tile), rows
//     past the last token fail the descriptor's range check and arrive as zeros -- no clamp, no 64-bit pointer math;
//   * the tile loop is unrolled by two so the LDS buffer index is a compile-time constant: every ds_read address is one of
//     six loop-invariant registers plus an immediate;
//   * the wave index is made scalar (M0 of the DMA comes from SALU, not v_readfirstlane);
//   * the per-tile max is a v_maximum3_f32 chain over the lane's own 16 keys; the cross-half exchange (one v_permlane32_swap)
//     happens only inside the rare rescale branch;
//   * the denominator is accumulated pairwise (v_pk_add_f32).
// Same arithmetic as v2 <DMA, VSUM> except for the order of the denominator's f32 adds.
// =====================================================================================================
template <int V> struct IntC { static constexpr int value = V; };

__device__ __forceinline__ float xhalf_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // r[0]: lanes 0..31's value everywhere, r[1]: lanes 32..63's
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// ds_read_b64_tr_b16 issued from an asm statement.  Why: hipcc's wait-count pass cannot tell which LDS bytes the transpose-read builtin
// touches and therefore puts an `s_waitcnt vmcnt(0)` in front of the first one of every tile -- i.e. it force-completes the NEXT tile's
// K/V LDS-DMA (issued at the top of the tile) in the middle of the current tile, a stall of several hundred cycles per tile and wave
// that the double buffer exists to avoid.  An asm read is invisible to that pass; tr16_join() is the explicit lgkmcnt wait in front of
// the consuming MFMA (tied to the fragment registers so that neither the read nor the MFMA can cross it).
__device__ __forceinline__ hx4 tr16_issue(unsigned lds_addr, int imm) {
  hx4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "i"(imm));
  return r;
}
__device__ __forceinline__ void tr16_join(hx4& a, hx4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
// counted form: YOUNGER = the asm reads issued after (a, b).  LDS operations retire in order and lgkmcnt counts every one of them, so
// reads the compiler issues in between only make the wait stricter, never too short.
__device__ __forceinline__ hx8 lds_b128_issue(unsigned lds_addr, int imm) {
  hx8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "i"(imm));
  return r;
}
template <int YOUNGER>
__device__ __forceinline__ void lds_b128_join_counted(hx8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(YOUNGER)); }
template <int YOUNGER>
__device__ __forceinline__ void tr16_join_counted(hx4& a, hx4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(YOUNGER)); }

// ONE_DECISION: one deferred-rescale decision per 64-key tile (maximum over both 32-key blocks) instead of one per block: one
// branch less per tile, and the second block's exponentials share a basic block with the first block's P V products, so the
// compiler can overlap them inside a wave (VALU beside MFMA) instead of leaving all overlap to the other waves of the SIMD.
// NEGM_BLOCK: -m lives in a persistent 16-register block that is the C operand of each 32-key block's first MFMA (no 32 v_mov per
// tile to initialise the score accumulators), and the two 32-key blocks of a tile are processed one after the other so that only
// one score tile is live and the block fits the 128-register budget of 4 waves per SIMD.
template <bool ONE_DECISION, bool NEGM_BLOCK = false, bool ASM_TR = false>
__global__ __launch_bounds__(256, NEGM_BLOCK ? 4 : 2) void attn_fwd_v5_kernel(const bf16_raw* __restrict__ qkv, bf16_raw* __restrict__ out, int N, int heads,
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

  const unsigned vaddr[2] = {(unsigned)(uintptr_t)smem + (unsigned)voff[0], (unsigned)(uintptr_t)smem + (unsigned)voff[1]};   // ASM_TR: LDS byte addresses
  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = 0.f;
  f32x2_t lsum = {0.f, 0.f};
  const int nt = (N + KT - 1) / KT;

  auto tile = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    dma_landed_barrier();                                // this wave's DMAs of tile t have landed; everyone is done reading tile t-1
    if (t + 1 < nt) stage(IntC<BUF ^ 1>{});
    const char* kb = smem + BUF * (2 * KV_BYTES);

    f32x16 s[2];
    const float neg_m = -m_run;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kt][i] = neg_m;
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        const hx8 kf = *reinterpret_cast<const hx8*>(kb + kt * 4096 + koff[sd]);
        s[kt] = UCOD_MFMA32(kf, qf[sd], s[kt]);
      }
    }
    if (t == nt - 1 && (N & (KT - 1)) != 0) {
      const int kbase = t * KT + 4 * h5;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= N) s[kt][r] = -1e30f;
        }
    }
    if constexpr (ONE_DECISION) {
      float mloc = __builtin_elementwise_maximum(s[0][0], s[0][1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[0][r]), s[0][r + 1]);
#pragma unroll
      for (int r = 0; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[1][r]), s[1][r + 1]);
      const bool first = (t == 0);
      if (first || __any(mloc > DEFER_THR)) {
        mloc = xhalf_max(mloc);
        const float delta = first ? mloc : fmaxf(mloc, 0.f);
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[0][i] -= delta;
          s[1][i] -= delta;
          o[0][i] *= alpha;
          o[1][i] *= alpha;
        }
        lsum *= alpha;
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      // v_maximum3_f32 (IEEE maximum: no operand canonicalisation), two scores per instruction.  The lane's 16 keys are enough
      // for the wave-wide "does any score run away" test; the other half's keys are fetched only when the rescale fires.
      float mloc = ONE_DECISION ? 0.f : __builtin_elementwise_maximum(s[kt][0], s[kt][1]);
      if constexpr (!ONE_DECISION) {
#pragma unroll
        for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[kt][r]), s[kt][r + 1]);
      }
      const bool first = (t == 0 && kt == 0);
      if (!ONE_DECISION && (first || __any(mloc > DEFER_THR))) {
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
        lsum *= alpha;
      }
      hx8 pb[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 w;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x2_t e = {__builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj + 1])};
          sum_pair(lsum, e);
          w[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
        }
        pb[ks] = __builtin_bit_cast(hx8, w);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          hx4 lo, hi;
          if constexpr (ASM_TR) {
            const int imm = BUF * (2 * KV_BYTES) + (kt * 32 + ks * 16) * 128;
            lo = tr16_issue(vaddr[dt], imm);
            hi = tr16_issue(vaddr[dt], imm + 8 * 128);
            tr16_join(lo, hi);
          } else {
            const char* p0 = kb + (kt * 32 + ks * 16) * 128 + voff[dt];
            lo = UCOD_TR16(p0);
            hi = UCOD_TR16(p0 + 8 * 128);
          }
          const hx8 vf = (hx8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[dt] = UCOD_MFMA32(vf, pb[ks], o[dt]);
        }
    }
  };

  f32x16 nm;
#pragma unroll
  for (int i = 0; i < 16; ++i) nm[i] = 0.f;
  const unsigned half_step = (unsigned)(32 * ld) * 2u;    // the swizzles repeat every 16 rows: chunk i = 1 is chunk 0 plus 32 rows
  auto stage_seq = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* dst = smem + BUF * (2 * KV_BYTES) + (i * 32 + wave * 8) * 128;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, sk[0] + i * half_step, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + KV_BYTES), 16, sv[0] + i * half_step, 0, 0, 0);
    }
    sk[0] += tile_step;
    sv[0] += tile_step;
  };
  auto tile_seq = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    dma_landed_barrier();
    if (t + 1 < nt) stage_seq(IntC<BUF ^ 1>{});
    const char* kb = smem + BUF * (2 * KV_BYTES);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x16 s = nm;
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        const hx8 kf = *reinterpret_cast<const hx8*>(kb + kt * 4096 + koff[sd]);
        s = UCOD_MFMA32(kf, qf[sd], s);
      }
      if (t == nt - 1 && (N & (KT - 1)) != 0) {
        const int kbase = t * KT + 4 * h5 + kt * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + (r & 3) + 8 * (r >> 2) >= N) s[r] = -1e30f;
      }
      float mloc = __builtin_elementwise_maximum(s[0], s[1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[r]), s[r + 1]);
      const bool first = (t == 0 && kt == 0);
      if (first || __any(mloc > DEFER_THR)) {
        mloc = xhalf_max(mloc);
        const float delta = first ? mloc : fmaxf(mloc, 0.f);
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[i] -= delta;
          nm[i] -= delta;                                  // = -m_run bit for bit (negation commutes with rounding)
          o[0][i] *= alpha;
          o[1][i] *= alpha;
        }
        lsum *= alpha;
      }
      hx8 pb[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 w;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x2_t e = {__builtin_amdgcn_exp2f(s[8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(s[8 * ks + 2 * jj + 1])};
          sum_pair(lsum, e);
          w[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
        }
        pb[ks] = __builtin_bit_cast(hx8, w);
      }
__builtin_amdgcn_sched_barrier(0);                  // the score tile dies here: the next block's Q K^T must not start above this point (one live score tile)
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

  if constexpr (NEGM_BLOCK) stage_seq(IntC<0>{});
  else stage(IntC<0>{});
  for (int t = 0; t < nt; t += 2) {
    if constexpr (NEGM_BLOCK) {
      tile_seq(t, IntC<0>{});
      if (t + 1 < nt) tile_seq(t + 1, IntC<1>{});
    } else {
      tile(t, IntC<0>{});
      if (t + 1 < nt) tile(t + 1, IntC<1>{});
    }
  }

  const float lane_sum = lsum[0] + lsum[1];
  const float denom = lane_sum + __shfl_xor(lane_sum, 32, 64);
  const float inv = 1.0f / denom;
  const int q = q0 + l31;
  if (q < N) {
    if (lse && h5 == 0) lse[((size_t)b * heads + head) * N + q] = m_run + __builtin_amdgcn_logf(denom);
    bf16_raw* op = out + ((size_t)b * N + q) * D + head * HD + 4 * h5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w[0] = cvt_pk_bf16(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        w[1] = cvt_pk_bf16(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * g) = w;
      }
  }
}


// =====================================================================================================
// pp ("ping-pong"): the v5 arithmetic in an 8-wave workgroup whose two wave groups run half a tile apart.
// One workgroup = 256 query rows = two 128-row groups A (waves 0-3) and B (waves 4-7); wave w and wave w + 4 share a SIMD.  A tile
// is processed in two segments separated by workgroup barriers:
//     M(t): O^T += V(t-1)^T P(t-1)   then   S(t)^T = K(t) Q^T - m      16 MFMAs, (almost) no VALU
//     V(t): deferred max, 32 x exp2, bf16 pack, row sums                no MFMA
// and B runs one segment behind A: while A's wave on a SIMD is in M(t) its partner is in V(t-1), so the matrix pipe and the vector
// ALU of every SIMD are both busy by construction instead of by chance (with four independent 4-wave workgroups per CU the v5
// kernel leaves that overlap to the hardware's arbitration: matrix pipe 0.51 busy, vector issue 0.69 busy,
// profiles/r02_attention_pmc.txt).  K / V tiles are DMA'd once per 256 rows (half the LDS-DMA traffic per FLOP of v5).
// Segment clock s (one barrier each): A runs M(t) at s = 2t, V(t) at 2t + 1; B runs M(t) at 2t + 1, V(t) at 2t + 2.
//   DMA batch u = {K(u+1), V(u)} is issued by ALL waves at the top of s = 2u (A: start of M(u); B: start of V(u-1)) and every wave
//   waits for its own part (vmcnt(0)) before the barrier that ends s = 2u + 1; first read at s = 2u + 2 (A's M(u+1)).
//   WAR: K(u+1) lands in the buffer of K(u-1) and V(u) in the buffer of V(u-2), both last read by B's M(u-1) at s = 2u - 1.
// =====================================================================================================
constexpr int QT2 = 256;

template <bool ONE_PRIO>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(const bf16_raw* __restrict__ qkv, bf16_raw* __restrict__ out, int N, int heads,
                                                              int npairs) {
  __shared__ __attribute__((aligned(16))) char smem[4 * KV_BYTES];       // K ring [2] | V ring [2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                             // 0 = A (leads), 1 = B (one segment behind)
  const int h5 = lane >> 5, l31 = lane & 31;
  const int nq = (N + QT2 - 1) / QT2;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot / nq) * 8 + xcd, qt = slot - (slot / nq) * nq;
  if (pair >= npairs) return;
  const int head = pair % heads, b = pair / heads;
  const int D = heads * HD, ld = 3 * D;
  const int q0 = qt * QT2 + wave * 32;                   // group A: rows 0..127 of the block, group B: 128..255
  const bf16_raw* base = qkv + (size_t)b * N * ld + head * HD;

  hx8 qf[4];
  {
    int qr = q0 + l31;
    qr = qr < N ? qr : N - 1;
    const bf16_raw* qp = base + (size_t)qr * ld + 8 * h5;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const hx8*>(qp + 16 * s);
  }
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(qkv + (size_t)b * N * ld), 0, (unsigned)N * (unsigned)ld * 2u, 0x00020000);
  // one 16-byte chunk of K and one of V per thread and tile: row = wave * 8 + lane / 8 (0..63), chunk = lane & 7
  unsigned sk, sv;
  {
    const int row = wave * 8 + (lane >> 3), ch = lane & 7;
    sk = (unsigned)(row * ld + D + head * HD + swz_k(row, ch) * 8) * 2u;
    sv = (unsigned)(row * ld + 2 * D + head * HD + swz_v(row, ch) * 8) * 2u;
  }
  const unsigned tile_step = (unsigned)(KT * ld) * 2u;
  char* const kring = smem;
  char* const vring = smem + 2 * KV_BYTES;
  auto dma_k = [&](int buf, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(kring + buf * KV_BYTES + wave * 1024), 16, off, 0, 0, 0);
  };
  auto dma_v = [&](int buf, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(vring + buf * KV_BYTES + wave * 1024), 16, off, 0, 0, 0);
  };
  const int nt = (N + KT - 1) / KT;
  // batch u = {K(u+1), V(u)}; rows past the last token fail the range check and arrive as zeros
  auto batch = [&](int u) {
    if (u + 1 < nt) dma_k((u + 1) & 1, sk + (unsigned)(u + 1) * tile_step);
    if (u < nt) dma_v(u & 1, sv + (unsigned)u * tile_step);
  };

  int koff[4], voff[2];
#pragma unroll
  for (int sd = 0; sd < 4; ++sd) koff[sd] = l31 * 128 + swz_k(l31, 2 * sd + h5) * 16;
  {
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int key = 4 * h5 + (i16 >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int dst = dt * 32 + g1 * 16 + 4 * (i16 & 3);
      voff[dt] = key * 128 + swz_v(key, dst >> 3) * 16 + (dst & 7) * 2;
    }
  }

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = 0.f;
  f32x2_t lsum = {0.f, 0.f};
  f32x16 s[2];
  hx8 pb[2][2];                                          // P(t) as the B operand of the next M segment: [32-key block][16-key step]

  auto wait_dma = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  auto seg_m = [&](int t) {                              // P V of tile t-1, then Q K^T of tile t
    if (t > 0) {
      // V fragments through asm reads (tr16_issue: a transpose-read BUILTIN right behind this group's DMA batch gets an
      // `s_waitcnt vmcnt(0)` from hipcc, i.e. the whole DMA round trip at the head of every M segment of group A, with group B
      // waiting at the barrier for it); ring of three, two products ahead, counted waits
      const unsigned vb0 = (unsigned)(uintptr_t)(vring + ((t - 1) & 1) * KV_BYTES) + (unsigned)voff[0];
      const unsigned vb1 = (unsigned)(uintptr_t)(vring + ((t - 1) & 1) * KV_BYTES) + (unsigned)voff[1];
      hx4 lo[3], hi[3];
      auto issue = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int imm = ((I >> 2) * 32 + ((I >> 1) & 1) * 16) * 128;
        lo[I % 3] = tr16_issue((I & 1) ? vb1 : vb0, imm);
        hi[I % 3] = tr16_issue((I & 1) ? vb1 : vb0, imm + 8 * 128);
      };
      auto product = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        if constexpr (I + 2 < 8) issue(IntC<I + 2>{});
        constexpr int ahead = (7 - I) < 2 ? (7 - I) : 2;
        tr16_join_counted<2 * ahead>(lo[I % 3], hi[I % 3]);
        const hx8 vf = (hx8){lo[I % 3][0], lo[I % 3][1], lo[I % 3][2], lo[I % 3][3], hi[I % 3][0], hi[I % 3][1], hi[I % 3][2], hi[I % 3][3]};
        o[I & 1] = UCOD_MFMA32(vf, pb[I >> 2][(I >> 1) & 1], o[I & 1]);
      };
      issue(IntC<0>{});
      issue(IntC<1>{});
      product(IntC<0>{}); product(IntC<1>{}); product(IntC<2>{}); product(IntC<3>{});
      product(IntC<4>{}); product(IntC<5>{}); product(IntC<6>{}); product(IntC<7>{});
    }
    if (t < nt) {
      const char* kb = kring + (t & 1) * KV_BYTES;
      const float neg_m = -m_run;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt][i] = neg_m;
#pragma unroll
        for (int sd = 0; sd < 4; ++sd) {
          const hx8 kf = *reinterpret_cast<const hx8*>(kb + kt * 4096 + koff[sd]);
          s[kt] = UCOD_MFMA32(kf, qf[sd], s[kt]);
        }
      }
    }
  };
  auto seg_v = [&](int t) {                              // softmax of tile t: s -> pb, lsum, (rare) rescale of o
    if (t == nt - 1 && (N & (KT - 1)) != 0) {
      const int kbase = t * KT + 4 * h5;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kt * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= N) s[kt][r] = -1e30f;
        }
    }
    float mloc = __builtin_elementwise_maximum(s[0][0], s[0][1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[0][r]), s[0][r + 1]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[1][r]), s[1][r + 1]);
    const bool first = (t == 0);
    if (first || __any(mloc > DEFER_THR)) {
      mloc = xhalf_max(mloc);
      const float delta = first ? mloc : fmaxf(mloc, 0.f);
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m_run += delta;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[0][i] -= delta;
        s[1][i] -= delta;
        o[0][i] *= alpha;
        o[1][i] *= alpha;
      }
      lsum *= alpha;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 w;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x2_t e = {__builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj + 1])};
          sum_pair(lsum, e);
          w[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
        }
        pb[kt][ks] = __builtin_bit_cast(hx8, w);
      }
  };

  // prologue: K(0) for everyone, then the segment clock starts
  dma_k(0, sk);
  wait_dma();
  bar();
  if (ONE_PRIO && grp == 1) __builtin_amdgcn_s_setprio(1);  // the later-dispatched half loses VALU arbitration by age: static priority
  if (grp == 0) {
    for (int u = 0; u < nt; ++u) {
      batch(u);                                            // s = 2u
      seg_m(u);
      bar();
      seg_v(u);                                            // s = 2u + 1
      wait_dma();
      bar();
    }
    seg_m(nt);                                             // s = 2 nt: the last P V
    bar();
  } else {
    batch(0);                                              // s = 0
    bar();
    for (int u = 0; u < nt; ++u) {
      seg_m(u);                                            // s = 2u + 1
      wait_dma();
      bar();
      batch(u + 1);                                        // s = 2u + 2
      seg_v(u);
      bar();
    }
    seg_m(nt);
  }

  const float lane_sum = lsum[0] + lsum[1];
  const float denom = lane_sum + __shfl_xor(lane_sum, 32, 64);
  const float inv = 1.0f / denom;
  const int q = q0 + l31;
  if (q < N) {
    bf16_raw* op = out + ((size_t)b * N + q) * D + head * HD + 4 * h5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w[0] = cvt_pk_bf16(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv);
        w[1] = cvt_pk_bf16(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * g) = w;
      }
  }
}


// =====================================================================================================
// w64: ping-pong INSIDE a wave.  One workgroup = 2 waves = 128 query rows; each wave owns 64 rows = two 32-row blocks A and B that run
// half a tile apart in ONE instruction stream:
//     block 1 of tile t:  S_B(t) = K(t) Q_B^T - m_B   (8 MFMAs)   beside   P_A(t) = exp2(S_A(t)), row sums   (VALU)
//                         O_A += V(t)^T P_A(t)        (8 MFMAs)   beside   running-max check of S_B(t)       (VALU)
//     block 2 of tile t:  S_A(t+1) = K(t+1) Q_A^T - m_A           beside   P_B(t) ...;   O_B += V(t)^T P_B(t)  beside  max check of S_A(t+1)
// so the matrix instructions of one block always have the other block's softmax to run beside, by construction and without the
// workgroup barriers the 8-wave ping-pong kernel pays for the same pairing (it lost to v5 on exactly those).  -m is a persistent
// 16-register C operand per block (no accumulator initialisation), the rescale branch is the only branch of the hot loop and sits at
// the end of each block; the key mask of a partial last tile lives in the peeled last two iterations.  K ring of three tiles (block A
// reads K(t+1) in the iteration in which block B reads K(t)) and of three V tiles, DMA two tiles ahead: 48 KB, one barrier per tile between two waves.
// =====================================================================================================
constexpr int QT3 = 128;
#ifndef W64_DEBUG
#define W64_DEBUG 0      // timing ablations of the w64 kernel (wrong results): 1 no barrier, 2 no V-fragment waits, 4 no exp, 8 no P V MFMAs, 16 no Q K^T MFMAs, 32 no max chain
#endif

// O^T += V^T P^T with the accumulator pinned to the AccVGPR half of the register file: the w64 kernel owns more than 256 registers per
// wave (one wave per SIMD) and O is touched by nothing but these MFMAs inside the tile loop, so the 64 O registers must not compete
// with the softmax operands for architectural VGPRs (left to itself hipcc puts the SCORE tiles there and pays a v_accvgpr_read per
// exponential).  The operands come from compiler-scheduled ds_read / v_cvt_pk (their waits are the compiler's); s_nop 1 covers the
// VALU-write -> MFMA-read wait states the hazard recogniser cannot see through an asm statement.
__device__ __forceinline__ void pv_mfma_acc(f32x16& acc, const hx8& a, const hx8& b) {
  asm volatile("s_nop 1\n\t" UCOD_MFMA32_ASM " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// O *= alpha in place in the AccVGPRs (rare rescale path).  Written as asm on "+a" operands so that O never changes register class at
// the join behind the branch: with a plain `o *= alpha` hipcc keeps the merged value in VGPRs and moves all 64 O registers of the block
// out of and back into the AccVGPRs on EVERY tile.
__device__ __forceinline__ void acc_scale(float& a, float alpha) {
  float t;
  asm volatile("v_accvgpr_read_b32 %1, %0\n\ts_nop 0\n\tv_mul_f32 %1, %1, %2\n\ts_nop 0\n\tv_accvgpr_write_b32 %0, %1" : "+a"(a), "=&v"(t) : "v"(alpha));
}
// before the compiler reads O out of the AccVGPRs (rescale branch, epilogue): the last asm MFMA (16 passes) must have retired
__device__ __forceinline__ void pv_mfma_fence(f32x16& a0, f32x16& a1) {
  asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a0), "+a"(a1));
}

__global__ __launch_bounds__(128) void attn_fwd_w64_kernel(const bf16_raw* __restrict__ qkv, bf16_raw* __restrict__ out, int N, int heads,
                                                           int npairs) {
  __shared__ __attribute__((aligned(16))) char smem[6 * KV_BYTES];       // K ring [3] | V ring [3]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5, l31 = lane & 31;
  const int nq = (N + QT3 - 1) / QT3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot / nq) * 8 + xcd, qt = slot - (slot / nq) * nq;
  if (pair >= npairs) return;
  const int head = pair % heads, b = pair / heads;
  const int D = heads * HD, ld = 3 * D;
  const int q0 = qt * QT3 + wave * 64;
  const bf16_raw* base = qkv + (size_t)b * N * ld + head * HD;

  hx8 qf[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    int qr = q0 + 32 * x + l31;
    qr = qr < N ? qr : N - 1;
    const bf16_raw* qp = base + (size_t)qr * ld + 8 * h5;
#pragma unroll
    for (int sd = 0; sd < 4; ++sd) qf[x][sd] = *reinterpret_cast<const hx8*>(qp + 16 * sd);
  }

  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(qkv + (size_t)b * N * ld), 0, (unsigned)N * (unsigned)ld * 2u, 0x00020000);
  // a 64-row tile = 4 DMA instructions per wave (1 KiB each: 8 rows); instruction i covers rows 16 i + 8 wave .. + 7 (the swizzles repeat every 16 rows)
  unsigned sk, sv;
  {
    const int row = wave * 8 + (lane >> 3), ch = lane & 7;
    sk = (unsigned)(row * ld + D + head * HD + swz_k(row, ch) * 8) * 2u;
    sv = (unsigned)(row * ld + 2 * D + head * HD + swz_v(row, ch) * 8) * 2u;
  }
  const unsigned tile_step = (unsigned)(KT * ld) * 2u, step16 = (unsigned)(16 * ld) * 2u;
  int kissue = 0, vissue = 0;
  auto stageK = [&]() {
    char* dst = smem + kissue * KV_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + i * 2048), 16, sk + i * step16, 0, 0, 0);
    sk += tile_step;
    kissue = kissue == 2 ? 0 : kissue + 1;
  };
  auto stageV = [&]() {
    char* dst = smem + (3 + vissue) * KV_BYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + i * 2048), 16, sv + i * step16, 0, 0, 0);
    sv += tile_step;
    vissue = vissue == 2 ? 0 : vissue + 1;
  };

  int koff[4], voff[2];
#pragma unroll
  for (int sd = 0; sd < 4; ++sd) koff[sd] = l31 * 128 + swz_k(l31, 2 * sd + h5) * 16;
  {
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int key = 4 * h5 + (i16 >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int dst = dt * 32 + g1 * 16 + 4 * (i16 & 3);
      voff[dt] = 3 * KV_BYTES + key * 128 + swz_v(key, dst >> 3) * 16 + (dst & 7) * 2;
    }
  }

  f32x16 o[2][2], s[2][2], nm[2];
  f32x2_t lsum[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    lsum[x] = (f32x2_t){0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[x][0][i] = 0.f; o[x][1][i] = 0.f; nm[x][i] = 0.f; }
  }
  const int nt = (N + KT - 1) / KT;

  // K fragments of the NEXT block's Q K^T are fetched into registers during the current block (one wave per SIMD: nothing else would
  // cover the LDS latency in front of the first product)
  hx8 kfr[2][4];
  auto loadk = [&](int kslot) {
    const char* kb = smem + kslot * KV_BYTES;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        if constexpr (!(W64_DEBUG & 64)) kfr[kt][sd] = *reinterpret_cast<const hx8*>(kb + kt * 4096 + koff[sd]);
        else kfr[kt][sd] = qf[kt][sd];
      }
  };
  // S_X(t)^T = K(t) Q_X^T - m_X from the fragments in kfr
  auto qk = [&](auto xc, auto maskc, int t) {
    constexpr int X = decltype(xc)::value;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      s[X][kt] = nm[X];
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        if constexpr (!(W64_DEBUG & 16)) s[X][kt] = UCOD_MFMA32(kfr[kt][sd], qf[X][sd], s[X][kt]);
        else s[X][kt][sd] += h_to_f32(kfr[kt][sd][0]);
      }
    }
    if constexpr (decltype(maskc)::value) {
      if (t == nt - 1 && (N & (KT - 1)) != 0) {
        const int kbase = t * KT + 4 * h5;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kbase + kt * 32 + (r & 3) + 8 * (r >> 2) >= N) s[X][kt][r] = -1e30f;
      }
    }
  };
  // deferred running max of block X: rescale only when a score runs away from -m by more than 2^DEFER_THR
  auto rescale_if = [&](auto xc, float mloc, bool first) {
    constexpr int X = decltype(xc)::value;
    if (first || __any(mloc > DEFER_THR)) {
      mloc = xhalf_max(mloc);
      const float delta = first ? mloc : fmaxf(mloc, 0.f);
      const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
      pv_mfma_fence(o[X][0], o[X][1]);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[X][0][i] -= delta;
        s[X][1][i] -= delta;
        nm[X][i] -= delta;
      }
      if (!first) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float a0 = o[X][0][i], a1 = o[X][1][i];
          acc_scale(a0, alpha);
          acc_scale(a1, alpha);
          o[X][0][i] = a0;
          o[X][1][i] = a1;
        }
      }
      lsum[X] *= alpha;
    }
  };
  auto maxfix = [&](auto xc, bool first) {                 // prologue only: the loop interleaves this chain with the P V products
    constexpr int X = decltype(xc)::value;
    float mloc = __builtin_elementwise_maximum(s[X][0][0], s[X][0][1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[X][0][r]), s[X][0][r + 1]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[X][1][r]), s[X][1][r + 1]);
    rescale_if(xc, mloc, first);
  };
  // V fragments of the eight P V products (kt, ks, dt) of a block: a ring of five, four products ahead of their use; the first four are
  // requested by pv_begin() at the TOP of the block (asm statements keep their order), i.e. a whole Q K^T ahead
  constexpr int VAHEAD = 4, VRING = VAHEAD + 1;
  hx4 lo[VRING], hi[VRING];
  unsigned vbase[2];
  auto issue = [&](auto ic) {
    constexpr int I = decltype(ic)::value;
    constexpr int imm = ((I >> 2) * 32 + ((I >> 1) & 1) * 16) * 128;
    if constexpr (!(W64_DEBUG & 128)) {
      lo[I % VRING] = tr16_issue(vbase[I & 1], imm);
      hi[I % VRING] = tr16_issue(vbase[I & 1], imm + 8 * 128);
    } else {
      lo[I % VRING] = (hx4){qf[0][0][0], qf[0][0][1], qf[0][0][2], qf[0][0][3]};
      hi[I % VRING] = lo[I % VRING];
    }
  };
  auto pv_begin = [&](int vslot) {
    vbase[0] = (unsigned)(uintptr_t)smem + (unsigned)(vslot * KV_BYTES + voff[0]);
    vbase[1] = (unsigned)(uintptr_t)smem + (unsigned)(vslot * KV_BYTES + voff[1]);
    issue(IntC<0>{});
    issue(IntC<1>{});
    issue(IntC<2>{});
    issue(IntC<3>{});
  };
  // One block:  S_Y = K Q_Y^T - m_Y  (fragments already in kfr)  beside  P_X = exp2(S_X), row sums;  the next block's K fragments;
  //             O_X^T += V^T P_X^T  beside  the running-max chain of S_Y;  rescale check of Y.   HAS_Y = 0: the very last block.
  auto block = [&](auto xc, auto yc, auto hasyc, auto maskc, int vslot, int knext_slot, int ty, bool first_y) {
    constexpr int X = decltype(xc)::value, Y = decltype(yc)::value;
    constexpr bool HAS_Y = decltype(hasyc)::value != 0;
    pv_begin(vslot);
    if constexpr (HAS_Y) qk(yc, maskc, ty);
    hx8 pb[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 w;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const f32x2_t e = (W64_DEBUG & 4) ? (f32x2_t){s[X][kt][8 * ks + 2 * jj], s[X][kt][8 * ks + 2 * jj + 1]}
                                            : (f32x2_t){__builtin_amdgcn_exp2f(s[X][kt][8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(s[X][kt][8 * ks + 2 * jj + 1])};
          sum_pair(lsum[X], e);
          w[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
        }
        pb[kt][ks] = __builtin_bit_cast(hx8, w);
      }
    if constexpr (HAS_Y) loadk(knext_slot);
    float mloc = 0.f;
    auto product = [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      if constexpr (I + VAHEAD < 8) issue(IntC<I + VAHEAD>{});
      constexpr int ahead = (7 - I) < VAHEAD ? (7 - I) : VAHEAD;
      constexpr int R = I % VRING;
      if constexpr (!(W64_DEBUG & 2)) tr16_join_counted<2 * ahead>(lo[R], hi[R]);
      const hx8 vf = (hx8){lo[R][0], lo[R][1], lo[R][2], lo[R][3], hi[R][0], hi[R][1], hi[R][2], hi[R][3]};
      if constexpr (!(W64_DEBUG & 8)) pv_mfma_acc(o[X][I & 1], vf, pb[I >> 2][(I >> 1) & 1]);
      else asm volatile("" :: "v"(vf), "v"(pb[I >> 2][(I >> 1) & 1]));
      if constexpr (HAS_Y && !(W64_DEBUG & 32)) {            // two links of Y's max chain behind every product
        constexpr int kt = I >> 2, r = (I & 3) * 4;
        if constexpr (I == 0) mloc = __builtin_elementwise_maximum(s[Y][0][0], s[Y][0][1]);
        else mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[Y][kt][r]), s[Y][kt][r + 1]);
        mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[Y][kt][r + 2]), s[Y][kt][r + 3]);
      }
    };
    product(IntC<0>{}); product(IntC<1>{}); product(IntC<2>{}); product(IntC<3>{});
    product(IntC<4>{}); product(IntC<5>{}); product(IntC<6>{}); product(IntC<7>{});
    if constexpr (HAS_Y) rescale_if(yc, mloc, first_y);
  };

  // DMA batches: batch(t) = {K(t+1), V(t)} is what iteration t needs at its top; it is issued at the top of iteration t-2 (two tiles of
  // flight: with one wave per SIMD nobody else covers a late tile) and waited for with vmcnt(8) = "all but the youngest batch"
  stageK();                                                // K(0)
  if (nt > 1) stageK();                                    // batch(0)
  stageV();
  if (nt > 2) stageK();                                    // batch(1)
  if (nt > 1) stageV();
  dma_landed_barrier();
  loadk(0);
  qk(IntC<0>{}, IntC<1>{}, 0);
  maxfix(IntC<0>{}, true);
  loadk(0);                                                // block 1 of tile 0 reads K(0) again, for B
  int kcur = 0, vcur = 0;
  auto iter = [&](int t, auto maskc) {
    constexpr bool TAIL = decltype(maskc)::value != 0;
    // K(t+1), V(t) have landed (batch(t+1) may still fly); everyone is done with tile t-1; then batch(t+2) -> the slots of K(t), V(t-1)
    if constexpr (TAIL) {
      dma_landed_barrier();
    } else {
      if constexpr (!(W64_DEBUG & 1)) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();
      }
    }
    if constexpr (!(W64_DEBUG & 256)) {
      if (!TAIL || t + 3 < nt) stageK();
      if (!TAIL || t + 2 < nt) stageV();
    }
    const int knext = kcur == 2 ? 0 : kcur + 1;
    // block 1: S_B(t) beside softmax_A(t), O_A += ..; then the fragments of K(t+1) for block 2
    block(IntC<0>{}, IntC<1>{}, IntC<1>{}, maskc, vcur, knext, t, t == 0);
    // block 2: S_A(t+1) beside softmax_B(t), O_B += ..; then the fragments of K(t+1) again, for block 1 of the next tile
    if (!TAIL || t + 1 < nt) block(IntC<1>{}, IntC<0>{}, IntC<1>{}, maskc, vcur, knext, t + 1, false);
    else block(IntC<1>{}, IntC<0>{}, IntC<0>{}, maskc, vcur, knext, t + 1, false);
    kcur = knext;
    vcur = vcur == 2 ? 0 : vcur + 1;
  };
  int t = 0;
  for (; t + 3 < nt; ++t) iter(t, IntC<0>{});
  for (; t < nt; ++t) iter(t, IntC<1>{});

#pragma unroll
  for (int x = 0; x < 2; ++x) {
    pv_mfma_fence(o[x][0], o[x][1]);
    const float lane_sum = lsum[x][0] + lsum[x][1];
    const float denom = lane_sum + __shfl_xor(lane_sum, 32, 64);
    const float inv = 1.0f / denom;
    const int q = q0 + 32 * x + l31;
    if (q < N) {
      bf16_raw* op = out + ((size_t)b * N + q) * D + head * HD + 4 * h5;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 w;
          w[0] = cvt_pk_bf16(o[x][dt][4 * g + 0] * inv, o[x][dt][4 * g + 1] * inv);
          w[1] = cvt_pk_bf16(o[x][dt][4 * g + 2] * inv, o[x][dt][4 * g + 3] * inv);
          *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * g) = w;
        }
    }
  }
}



// =====================================================================================================
// r3: the ROUND-3 product kernel (dead-wave / dead-block skip, 16-byte stores) with three more experiment flags, all measured SLOWER than
// the plain form (profiles/r03_attention_onescol_msum.txt, r03_attention_seq_occupancy.txt; variants 30 = plain, 31 ONESCOL, 32 MSUM, 33 both,
// 34 SEQ at five waves per SIMD, 35 SEQ at four):
// Two ways of moving softmax bookkeeping from the vector issue stream (the kernel's bound: ~9.7 vector instructions per MFMA, matrix pipe
// 0.54 busy; profiles/r03_attention_pmc.txt) onto the matrix pipe, measured as template flags (tools/attn_ab.py, variants 31-33 of the
// experiment build):
//   ONESCOL  S' = S - m as a FIFTH k-step of the score product instead of 16 v_mov per 32-key block: A fragment = a constant column of
//            ones (k = 0 of the step), B fragment = -m of the lane's query in that k (m kept representable in the 16-bit operand type so
//            that the subtraction is exact); the first k-step then starts from the inline constant 0.
//   MSUM     the softmax denominator as one more MFMA per 16 keys (all-ones A fragment times the P fragment that feeds P V: the row
//            sums of the ROUNDED probabilities) instead of 32 v_add_f32 per tile.
//   SEQ      one 32-key block at a time (scores, softmax, P V, then the next block): one live score tile -> 96 registers -> five waves per SIMD
template <bool ONESCOL, bool MSUM, int SEQ = 0, int ABL = 0, int SCHED = 0>
__global__ __launch_bounds__(256, SEQ == 1 ? 5 : (SCHED >= 3 ? 4 : 2)) void attn_fwd_r3_kernel(const bf16_raw* __restrict__ qkv, bf16_raw* __restrict__ out, int N, int heads,
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
  f32x2_t lsum = {0.f, 0.f};
  f32x16 osum;                                           // MSUM: every row of this tile = the lane's denominator
#pragma unroll
  for (int i = 0; i < 16; ++i) osum[i] = 0.f;
  // ONESCOL operands: element (half 0, j = 0) <-> k = 0 of the extra k-step.  kone = 1 there, qm = -m there, zeros elsewhere.
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
  const unsigned one_h = (unsigned)__builtin_bit_cast(unsigned short, (half_t)1.0f);
  const hx8 kone = __builtin_bit_cast(hx8, (u32x4_t){h5 == 0 ? one_h : 0u, 0u, 0u, 0u});
  const hx8 ones = __builtin_bit_cast(hx8, (u32x4_t){one_h * 0x10001u, one_h * 0x10001u, one_h * 0x10001u, one_h * 0x10001u});
  hx8 qm = __builtin_bit_cast(hx8, (u32x4_t){0u, 0u, 0u, 0u});
  const int nt = (N + KT - 1) / KT, nfull = nt - 1;
  const bool half_dead = nfull * KT + 32 >= N;           // (uniform) the keys of the last tile's second 32-key block all lie past the last token

  // ABL (timing only, results are wrong): 1 = no MFMA (operands and accumulator kept alive, accumulator opaque), 2 = no softmax vector
  // work (max / exp / sum / convert), 4 = no LDS fragment reads, 8 = no K/V staging and no barrier
  // SCHED 5 / 6 / 7 (bit 0 of SCHED - 4: NEGM, bit 1: DOT2) -- vector instructions taken OUT of the tile body:
  //   NEGM  a persistent 16-register block holding -m (rewritten only inside the rare rescale branch) as the C operand of each block's first
  //         MFMA, instead of 16 v_mov per tile re-creating that splat;
  //   DOT2  the denominator as one v_dot2c_f32_bf16 (pair of ROUNDED probabilities times (1, 1)) per pair instead of two v_add_f32.
  constexpr bool NEGM = SCHED >= 5 && ((SCHED - 4) & 1), DOT2 = SCHED >= 5 && ((SCHED - 4) & 2);
  f32x16 negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  if constexpr (NEGM) asm volatile("" : "+v"(negm));     // opaque: the compiler must not know it is a splat (it would re-materialise it per tile)
  auto MF = [&](const hx8& a, const hx8& bq, f32x16 c) -> f32x16 {
    if constexpr (ABL & 1) {
      asm volatile("" : "+v"(c) : "v"(a), "v"(bq));
      return c;
    } else return UCOD_MFMA32(a, bq, c);
  };
  auto tile = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    if constexpr (!(ABL & 8)) {
      dma_landed_barrier();                              // this wave's DMAs of tile t have landed; everyone is done reading tile t-1
      if (t + 1 < nt) stage(IntC<BUF ^ 1>{});
    }
    if (!live) return;                                   // (wave-uniform) nothing to compute for rows past the last token
    const char* kb = smem + BUF * (2 * KV_BYTES);

    if constexpr (SEQ != 0) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        if (t == nfull && kt == 1 && half_dead) break;
        f32x16 sc;
        const float neg_m = -m_run;
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = neg_m;
#pragma unroll
        for (int sd = 0; sd < 4; ++sd) {
          const hx8 kf = *reinterpret_cast<const hx8*>(kb + kt * 4096 + koff[sd]);
          sc = UCOD_MFMA32(kf, qf[sd], sc);
        }
        if (t == nfull && (N & (KT - 1)) != 0) {
          const int kbase = t * KT + 4 * h5 + kt * 32;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (kbase + (r & 3) + 8 * (r >> 2) >= N) sc[r] = -1e30f;
        }
        float mloc = __builtin_elementwise_maximum(sc[0], sc[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, sc[r]), sc[r + 1]);
        const bool first = (t == 0 && kt == 0);
        if (first || __any(mloc > DEFER_THR)) {
          mloc = xhalf_max(mloc);
          const float delta = first ? mloc : fmaxf(mloc, 0.f);
          const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
          m_run += delta;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            sc[i] -= delta;
            o[0][i] *= alpha;
            o[1][i] *= alpha;
          }
          lsum *= alpha;
        }
        hx8 pb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 w;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x2_t e = {__builtin_amdgcn_exp2f(sc[8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(sc[8 * ks + 2 * jj + 1])};
            sum_pair(lsum, e);
            w[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
          }
          pb[ks] = __builtin_bit_cast(hx8, w);
        }
        __builtin_amdgcn_sched_barrier(0);                 // the score tile dies here: the next block's Q K^T must not start above this point
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
      return;
    }
    if constexpr (SCHED == 3 || SCHED == 4) {
      // max / rescale test of the first 32-key block BEFORE the second block's Q K^T, so that the first block's exponentials, converts and
      // sums share a basic block with those four MFMAs; the second block's max shares one with the first block's P V.
      const bool do1 = !(t == nfull && half_dead);
      const bool masked = t == nfull && (N & (KT - 1)) != 0;
      f32x16 s0, s1;
      auto qk = [&](f32x16& sc, int kt) {
        const float nm = -m_run;
#pragma unroll
        for (int i = 0; i < 16; ++i) sc[i] = nm;
#pragma unroll
        for (int sd = 0; sd < 4; ++sd) {
          const hx8 kf = *reinterpret_cast<const hx8*>(kb + kt * 4096 + koff[sd]);
          sc = UCOD_MFMA32(kf, qf[sd], sc);
        }
      };
      auto mask = [&](f32x16& sc, int kt) {
        const int kbase = t * KT + 4 * h5 + kt * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbase + (r & 3) + 8 * (r >> 2) >= N) sc[r] = -1e30f;
      };
      auto maxfix = [&](f32x16& sc, bool first) {
        float mloc = __builtin_elementwise_maximum(sc[0], sc[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, sc[r]), sc[r + 1]);
        if (first || __any(mloc > DEFER_THR)) {
          mloc = xhalf_max(mloc);
          const float delta = first ? mloc : fmaxf(mloc, 0.f);
          const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
          m_run += delta;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            sc[i] -= delta;
            o[0][i] *= alpha;
            o[1][i] *= alpha;
          }
          lsum *= alpha;
        }
      };
      auto probs = [&](const f32x16& sc, hx8 (&pb)[2]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 w;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const f32x2_t e = {__builtin_amdgcn_exp2f(sc[8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(sc[8 * ks + 2 * jj + 1])};
            sum_pair(lsum, e);
            w[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
          }
          pb[ks] = __builtin_bit_cast(hx8, w);
        }
      };
      auto pv = [&](const hx8 (&pb)[2], int kt) {
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
      };
      hx8 p0[2], p1[2];
      qk(s0, 0);
      if (masked) mask(s0, 0);
      maxfix(s0, t == 0);
      if (do1) {
        qk(s1, 1);                                         // four MFMAs ...
        probs(s0, p0);                                     // ... beside 16 exp + 8 convert + 16 add of the first block
        if constexpr (SCHED == 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
          }
        }
        if (masked) mask(s1, 1);
        pv(p0, 0);
        maxfix(s1, false);
        probs(s1, p1);
        pv(p1, 1);
      } else {
        probs(s0, p0);
        pv(p0, 0);
      }
      return;
    }
    f32x16 s[2];
    const float neg_m = -m_run;
    if constexpr (SCHED == 1) __builtin_amdgcn_s_setprio(1);
    if constexpr (SCHED == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (t == nfull && kt == 1 && half_dead) break;     // (wave-uniform; last tile only) no real key in the second 32-key block
      if constexpr (!NEGM) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt][i] = ONESCOL ? 0.f : neg_m;
      }
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        const hx8 kf = (ABL & 4) ? qf[(sd + 1) & 3] : *reinterpret_cast<const hx8*>(kb + kt * 4096 + koff[sd]);
        s[kt] = MF(kf, qf[sd], (NEGM && sd == 0) ? negm : s[kt]);
      }
      if constexpr (ONESCOL) s[kt] = UCOD_MFMA32(kone, qm, s[kt]);
    }
    if constexpr (SCHED == 1) __builtin_amdgcn_s_setprio(0);
    if constexpr (SCHED == 2) __builtin_amdgcn_s_setprio(1);
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
      if constexpr (!(ABL & 2)) {
#pragma unroll
        for (int r = 2; r < 16; r += 2) mloc = __builtin_elementwise_maximum(__builtin_elementwise_maximum(mloc, s[kt][r]), s[kt][r + 1]);
      }
      const bool first = (t == 0 && kt == 0);
      if (!(ABL & 2) && (first || __any(mloc > DEFER_THR))) {
        mloc = xhalf_max(mloc);
        float delta = first ? mloc : fmaxf(mloc, 0.f);
        if constexpr (ONESCOL) {                           // the new maximum must be exact in the operand type: delta = what m really moves by
          const float m_new = (float)(half_t)(m_run + delta);
          delta = m_new - m_run;
          const unsigned nm = (unsigned)__builtin_bit_cast(unsigned short, (half_t)(-m_new));
          qm = __builtin_bit_cast(hx8, (u32x4_t){h5 == 0 ? nm : 0u, 0u, 0u, 0u});
        }
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);
        m_run += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[kt][i] -= delta;
          if (kt == 0) s[1][i] -= delta;
          o[0][i] *= alpha;
          o[1][i] *= alpha;
        }
        lsum *= alpha;
        if constexpr (MSUM) osum[0] *= alpha;
        if constexpr (NEGM) {
          const float nm = -m_run;
#pragma unroll
          for (int i = 0; i < 16; ++i) negm[i] = nm;
          asm volatile("" : "+v"(negm));
        }
      }
      hx8 pb[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u32x4 w;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if constexpr (ABL & 2) {                         // no exp / sum / convert: the raw score bits feed the second product
            w[jj] = (__builtin_bit_cast(unsigned, s[kt][8 * ks + 2 * jj]) >> 16) | (__builtin_bit_cast(unsigned, s[kt][8 * ks + 2 * jj + 1]) & 0xffff0000u);
          } else {
          const f32x2_t e = {__builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj]), __builtin_amdgcn_exp2f(s[kt][8 * ks + 2 * jj + 1])};
          if constexpr (!MSUM && !DOT2) sum_pair(lsum, e);
          w[jj] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, bf16x2_t));
          if constexpr (DOT2) {
#ifdef UCOD_HALF_F16
            lsum[jj & 1] = __builtin_amdgcn_fdot2(__builtin_bit_cast(hx2, w[jj]), __builtin_bit_cast(hx2, 0x3c003c00u), lsum[jj & 1], false);
#else
            lsum[jj & 1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hx2, w[jj]), __builtin_bit_cast(hx2, 0x3f803f80u), lsum[jj & 1], false);
#endif
          }
          }
        }
        pb[ks] = __builtin_bit_cast(hx8, w);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (MSUM) osum = UCOD_MFMA32(ones, pb[ks], osum);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const char* p0 = kb + (kt * 32 + ks * 16) * 128 + voff[dt];
          hx8 vf;
          if constexpr (ABL & 4) vf = qf[(2 * ks + dt) & 3];
          else {
            const hx4 lo = UCOD_TR16(p0);
            const hx4 hi = UCOD_TR16(p0 + 8 * 128);
            vf = (hx8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
          o[dt] = MF(vf, pb[ks], o[dt]);
        }
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

  const float lane_sum = lsum[0] + lsum[1];
  const float denom = MSUM ? osum[0] : lane_sum + __shfl_xor(lane_sum, 32, 64);   // (the all-ones product already sums both halves' keys)
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


}  // namespace ucod

extern "C" int ucod_attention_fwd_lab(const void* qkv, void* out, int B, int tok, int heads, float scale, int variant, void* stream) {
  using namespace ucod;
  if (!qkv || !out || B <= 0 || tok <= 0 || heads <= 0) return UCOD_EINVAL;
  dim3 grid(cdiv(tok, QT), heads, B), block(256);
  const float c = scale * 1.4426950408889634f;
  if (scale == 0.f) {   // Q pre-scaled by head_dim^-0.5 * log2(e): the VALU-lean kernel
    const int npairs = B * heads, nq = cdiv(tok, QT);
    dim3 grid1(cdiv(npairs, 8) * 8 * nq);
    // variant 3: K/V staged through registers, 4: LDS-DMA, both with the denominator on the matrix pipe (all-ones MFMA);
    // 6: LDS-DMA + denominator as f32 adds of the unrounded probabilities (4 % faster than 4 at 4 waves per SIMD)
    if (variant >= 30 && variant <= 59) {
      const float* nol = nullptr;
#define UCOD_ABL_CASE(V, A) if (variant == V) hipLaunchKernelGGL((attn_fwd_r3_kernel<false, false, 0, A>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nol);
      UCOD_ABL_CASE(40, 1) UCOD_ABL_CASE(41, 2) UCOD_ABL_CASE(42, 4) UCOD_ABL_CASE(43, 8) UCOD_ABL_CASE(44, 14) UCOD_ABL_CASE(45, 13) UCOD_ABL_CASE(46, 12) UCOD_ABL_CASE(47, 6)
#undef UCOD_ABL_CASE
#define UCOD_SCHED_CASE(V, S) if (variant == V) hipLaunchKernelGGL((attn_fwd_r3_kernel<false, false, 0, 0, S>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nol);
      UCOD_SCHED_CASE(50, 1) UCOD_SCHED_CASE(51, 2) UCOD_SCHED_CASE(52, 3) UCOD_SCHED_CASE(53, 4) UCOD_SCHED_CASE(54, 5) UCOD_SCHED_CASE(55, 6) UCOD_SCHED_CASE(56, 7)
#undef UCOD_SCHED_CASE
      if (variant == 30) hipLaunchKernelGGL((attn_fwd_r3_kernel<false, false, 0>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nol);
      if (variant == 31) hipLaunchKernelGGL((attn_fwd_r3_kernel<true, false, 0>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nol);
      if (variant == 32) hipLaunchKernelGGL((attn_fwd_r3_kernel<false, true, 0>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nol);
      if (variant == 33) hipLaunchKernelGGL((attn_fwd_r3_kernel<true, true, 0>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nol);
      if (variant == 34) hipLaunchKernelGGL((attn_fwd_r3_kernel<false, false, 1>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nol);
      if (variant == 35) hipLaunchKernelGGL((attn_fwd_r3_kernel<false, false, 2>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nol);
    } else if (variant == 4)
      hipLaunchKernelGGL((attn_fwd_v2_kernel<true, false>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nullptr);
    else if (variant == 3)
      hipLaunchKernelGGL((attn_fwd_v2_kernel<false>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nullptr);
    else if (variant == 6)
      hipLaunchKernelGGL((attn_fwd_v2_kernel<true, true>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nullptr);
    else if (variant == 9 || variant == 12) {              // 8-wave ping-pong (256 query rows per workgroup); 12 = with static priority for group B
      const int nq2 = cdiv(tok, QT2);
      dim3 grid2(cdiv(npairs, 8) * 8 * nq2), block2(512);
      if (variant == 9) hipLaunchKernelGGL((attn_fwd_pp_kernel<false>), grid2, block2, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs);
      else hipLaunchKernelGGL((attn_fwd_pp_kernel<true>), grid2, block2, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs);
    } else if (variant == 14) {                            // in-wave ping-pong: 2 waves x 64 query rows per workgroup
      const int nq3 = cdiv(tok, QT3);
      hipLaunchKernelGGL(attn_fwd_w64_kernel, dim3(cdiv(npairs, 8) * 8 * nq3), dim3(128), 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs);
    } else if (variant == 13)                              // v5 with -m as a persistent C-operand block, 32-key blocks in sequence
      hipLaunchKernelGGL((attn_fwd_v5_kernel<false, true>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nullptr);
    else if (variant == 15)                                // v5 with the V transpose-reads issued from asm (no vmcnt(0) in the tile)
      hipLaunchKernelGGL((attn_fwd_v5_kernel<false, false, true>), grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nullptr);
    else if (variant == 7)                                 // v5 with one rescale decision per 64-key tile
      hipLaunchKernelGGL(attn_fwd_v5_kernel<true>, grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nullptr);
    else                                                 // 0 / 2 / 5: the trimmed-issue kernel (buffer DMA, constant LDS offsets), 7-10 % faster than 6
      hipLaunchKernelGGL(attn_fwd_v5_kernel<false>, grid1, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, npairs, (float*)nullptr);
    UCOD_CHECK_LAUNCH();
    return UCOD_OK;
  }
  if (variant == 1)
    hipLaunchKernelGGL((attn_fwd_kernel<1>), grid, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, c);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<0>), grid, block, 0, (hipStream_t)stream, (const bf16_raw*)qkv, (bf16_raw*)out, tok, heads, c);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

