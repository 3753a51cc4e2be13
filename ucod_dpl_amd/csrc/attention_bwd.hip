// Multi-head attention backward, head_dim 64 (backbone-backward mode, SURVEY.md 8a row B9: eager_attention_forward of
// transformers modeling_dinov2.py:153-179 differentiated; dropout 0, no mask).
//
// Same transposed-score scheme as attention.hip: v_mfma_f32_32x32x16_bf16, a lane owns one column of every 32x32 tile, the
// C registers are converted to bf16 in place and reused as the B operand of the next product.  With Q' = Q * hd^-0.5 * log2(e)
// (what the forward stored), L = base-2 log-sum-exp of the scaled scores, P = exp2(Q'K^T - L), dP = dO V^T,
// delta = rowsum(dO * O), dS = P * (dP - delta):
//     dQ = hd^-0.5 * dS K          dK = ln(2) * dS^T Q'          dV = P^T dO
// Two kernels, no atomics (bitwise reproducible):
//   attn_bwd_dq_kernel   a wave owns 32 QUERIES (lane = query), walks the key tiles: 8 MFMAs for S^T and dP^T, 4 for dQ^T
//                        per 32 keys; L and delta are per-lane scalars folded into the accumulator initial values; also
//                        writes delta for the second kernel;
//   attn_bwd_dkv_kernel  a wave owns 32 KEYS (lane = key), walks the query tiles: S and dP (A operand = Q' / dO rows from
//                        LDS, B operand = the wave's K / V fragments in registers), then dV^T += dO^T P and
//                        dK^T += Q'^T dS with dO^T / Q'^T read through ds_read_b64_tr_b16; L[q] and delta[q] vary
//                        along the accumulator ROWS here, so they are staged in LDS and loaded as the initial values.
// Tiles that are needed both row-major (A operand of the score products) and transposed (A operand of the gradient
// products) are staged twice, once per bank swizzle.
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {
namespace {

constexpr int HD = 64;
constexpr int WT = 128;                 // rows (queries or keys) owned by one workgroup: 4 waves x 32
constexpr int ST = 64;                  // rows per streamed tile
constexpr int TILE = ST * HD * 2;       // 8 KiB

__device__ __forceinline__ int swz_k(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int swz_v(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// C registers 8*ks .. 8*ks+7 of a 32x32 tile -> bf16 B operand of the next product (k index = the C row permutation)
__device__ __forceinline__ bf16x8 pack_b(const f32x16& s, int ks) {
  u32x4_t w;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) w[jj] = pack_bf16x2(s[8 * ks + 2 * jj], s[8 * ks + 2 * jj + 1]);
  return __builtin_bit_cast(bf16x8, w);
}

// A operand X^T[32 d of block dt][16 rows of block (half, ks) in the C row permutation] from a swz_v-staged [64][64] tile
__device__ __forceinline__ bf16x8 read_tr(const char* tile, int off) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tile + off + 8 * 128));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// The same transposed fragment with its two ds_read_b64_tr_b16 issued from asm, and the wait by hand.  Why: hipcc puts `s_waitcnt vmcnt(0)`
// in front of the first transpose-read builtin of every tile (it cannot tell that the LDS-DMA in flight targets the other stage), which
// force-completes the NEXT tile's K / V / dO DMA a third of the way into the current tile; at the 2-3 waves per SIMD of these kernels
// that latency is not covered.  tr_issue() requests a fragment, tr_take<YOUNGER>() waits for it (YOUNGER = asm reads issued after it: LDS
// operations retire in order, so reads the compiler issues in between only make the wait stricter).
struct TrFrag { bf16x4 lo, hi; };
__device__ __forceinline__ TrFrag tr_issue(const char* tile, int off) {
  TrFrag f;
  const unsigned a = (unsigned)(uintptr_t)(tile + off);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(f.hi) : "v"(a));
  return f;
}
template <int YOUNGER>
__device__ __forceinline__ bf16x8 tr_take(TrFrag& f) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.lo), "+v"(f.hi) : "n"(YOUNGER));
  return (bf16x8){f.lo[0], f.lo[1], f.lo[2], f.lo[3], f.hi[0], f.hi[1], f.hi[2], f.hi[3]};
}

// Row-per-lane epilogue: lanes l and l+32 hold adjacent 4-column groups of one output row (columns 8g + 4*h5 .. +3 of each 8-column
// group g); one v_permlane32_swap per register pair leaves lanes 0..31 with columns 8g .. 8g+7 and lanes 32..63 with 8(g+1) .. 8(g+1)+7:
// one 16-byte store instead of two 8-byte ones (the store tail of such kernels is issue-bound per instruction).  `off` = byte offset of
// the row's 64-column head block + 16 * h5 (rows that must not be written: an offset past the buffer), `scale` applied before rounding.
__device__ __forceinline__ void store_rows64(const f32x16 (&acc)[2], float scale, __amdgpu_buffer_rsrc_t rs, unsigned off) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int g = 2 * gp;
      const unsigned a0 = pack_bf16x2(acc[dt][4 * g + 0] * scale, acc[dt][4 * g + 1] * scale), a1 = pack_bf16x2(acc[dt][4 * g + 2] * scale, acc[dt][4 * g + 3] * scale);
      const unsigned b0 = pack_bf16x2(acc[dt][4 * g + 4] * scale, acc[dt][4 * g + 5] * scale), b1 = pack_bf16x2(acc[dt][4 * g + 6] * scale, acc[dt][4 * g + 7] * scale);
      const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      const u32x4 w = {(unsigned)r0[0], (unsigned)r1[0], (unsigned)r0[1], (unsigned)r1[1]};
      __builtin_amdgcn_raw_buffer_store_b128(w, rs, off + (unsigned)(dt * 32 + 16 * gp) * 2u, 0, 0);
    }
}
constexpr unsigned ROW_DROPPED = 0x80000000u;          // (+ the per-store constants stays beyond any buffer of < 2 GiB; 0xFFFFFFF0 would wrap)

struct Map {            // XCD-aware 1-D grid -> (image, head, row block): all blocks of one (image, head) on one XCD
  int b, head, blk;
  bool ok;
};
__device__ __forceinline__ Map decode(int N, int heads, int npairs) {
  const int nb = (N + WT - 1) / WT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot / nb) * 8 + xcd;
  Map m;
  m.blk = slot - (slot / nb) * nb;
  m.ok = pair < npairs;
  m.head = pair % heads;
  m.b = pair / heads;
  return m;
}

}  // namespace

// Staging (both kernels): 16-byte buffer loads straight to LDS, the bank swizzle applied to the SOURCE chunk a lane fetches; the
// per-lane byte offsets are loop-carried (one v_add per DMA and tile) and a row past the last token fails the descriptor's range
// check and arrives as zeros.  The tile loop is unrolled by two, so the LDS buffer index is a compile-time constant and every
// ds_read address is a loop-invariant register plus an immediate.  (The first version staged through registers: 4 global loads
// with 64-bit address arithmetic and 6-8 ds_write_b128 per thread and tile -- the LDS store path alone was 40 % of the kernel's
// LDS cycles, on a kernel whose LDS and VALU/MFMA cycles per tile are about equal.)
template <int V> struct IntC { static constexpr int value = V; };
#define LDS_DMA16(rs, dst, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rs), (__attribute__((address_space(3))) void*)(dst), 16, (voff), (soff), 0, 0)

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16_raw* __restrict__ qkv, const bf16_raw* __restrict__ out,
                                                              const bf16_raw* __restrict__ dout, const float* __restrict__ lse,
                                                              float* __restrict__ delta, bf16_raw* __restrict__ dqkv, int ldd, int N,
                                                              int heads, int npairs, float qscale) {
  constexpr int STAGE = 3 * TILE;                                        // K (swz_k) | K (swz_v) | V (swz_k)
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5, l31 = lane & 31;
  const Map mp = decode(N, heads, npairs);
  if (!mp.ok) return;
  const int D = heads * HD, ld = 3 * D;
  const int q0 = mp.blk * WT + wave * 32;
  const bool live = q0 < N;                              // (wave-uniform) a wave whose 32 queries all lie past the last token only stages and keeps the barriers
  const bf16_raw* base = qkv + (size_t)mp.b * N * ld + mp.head * HD;
  const int q = q0 + l31, qc = q < N ? q : N - 1;

  bf16x8 qf[4], dof[4];
  float dl = 0.f;
  {
    const bf16_raw* qp = base + (size_t)qc * ld + 8 * h5;
    const bf16_raw* op = out + ((size_t)mp.b * N + qc) * D + mp.head * HD + 8 * h5;
    const bf16_raw* dp = dout + ((size_t)mp.b * N + qc) * D + mp.head * HD + 8 * h5;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
      dof[s] = *reinterpret_cast<const bf16x8*>(dp + 16 * s);
      const bf16x8 of = *reinterpret_cast<const bf16x8*>(op + 16 * s);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)dof[s][e] * (float)of[e];
    }
  }
  dl += __shfl_xor(dl, 32, 64);
  const size_t stat = ((size_t)mp.b * heads + mp.head) * N + qc;
  const float L = lse[stat];
  if (h5 == 0 && q < N) delta[stat] = dl;

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

  const int nt = (N + ST - 1) / ST;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(qkv + (size_t)mp.b * N * ld), 0, (unsigned)N * (unsigned)ld * 2u, 0x00020000);
  unsigned sk1[2], sk2[2];                                               // K chunk under swz_k / swz_v; V = the swz_k offset + 2*D bytes (soffset)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3), ch = lane & 7;
    sk1[i] = (unsigned)(row * ld + D + mp.head * HD + swz_k(row, ch) * 8) * 2u;
    sk2[i] = (unsigned)(row * ld + D + mp.head * HD + swz_v(row, ch) * 8) * 2u;
  }
  const unsigned tile_step = (unsigned)(ST * ld) * 2u;
  const int v_delta = D * 2;
  auto stage = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* dst = smem + BUF * STAGE + (i * 32 + wave * 8) * 128;
      LDS_DMA16(rs, dst, sk1[i], 0);
      LDS_DMA16(rs, dst + TILE, sk2[i], 0);
      LDS_DMA16(rs, dst + 2 * TILE, sk1[i], v_delta);
      sk1[i] += tile_step;
      sk2[i] += tile_step;
    }
  };
  int roff[4], toff[2];
#pragma unroll
  for (int sd = 0; sd < 4; ++sd) roff[sd] = l31 * 128 + swz_k(l31, 2 * sd + h5) * 16;       // +4096 per 32 rows keeps the swizzle
  {
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int row = 4 * h5 + (i16 >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int dst = dt * 32 + g1 * 16 + 4 * (i16 & 3);
      toff[dt] = row * 128 + swz_v(row, dst >> 3) * 16 + (dst & 7) * 2;                      // +8/16/32 rows keep the swizzle
    }
  }

  // TAIL (compile time): the partial last key tile.  As a run-time test inside the element loop the mask costs a compare and two
  // selects per score on EVERY tile (96 of ~250 vector instructions per tile); the last tile is peeled instead.
  auto tile = [&](int t, auto bufc, auto tailc) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr bool TAIL = decltype(tailc)::value != 0;
    dma_landed_barrier();                                  // this wave's LDS-DMA pieces of tile t, then everyone's
    if (t + 1 < nt) stage(IntC<BUF ^ 1>{});
    if (!live) return;
    const char* kb = smem + BUF * STAGE;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (TAIL && kt == 1 && t * ST + 32 >= N) break;      // (wave-uniform, last tile only) no real key in the second 32-key block: P = 0 there
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = -L; dp[i] = -dl; }
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb + kt * 4096 + roff[sd]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[sd], s, 0, 0, 0);
        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(kb + 2 * TILE + kt * 4096 + roff[sd]);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[sd], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(s[r]);
        if constexpr (TAIL) {                                            // zero-filled K rows give P = exp2(-L), not 0
          const int key = t * ST + kt * 32 + 4 * h5 + (r & 3) + 8 * (r >> 2);
          if (key >= N) p = 0.f;
        }
        s[r] = p * dp[r];                                            // dS^T
      }
      {                                                          // four K^T fragments requested together, taken in order
        TrFrag f00 = tr_issue(kb + TILE + (kt * 32) * 128, toff[0]), f01 = tr_issue(kb + TILE + (kt * 32) * 128, toff[1]);
        TrFrag f10 = tr_issue(kb + TILE + (kt * 32 + 16) * 128, toff[0]), f11 = tr_issue(kb + TILE + (kt * 32 + 16) * 128, toff[1]);
        const bf16x8 ds0 = pack_b(s, 0), ds1 = pack_b(s, 1);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_take<6>(f00), ds0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_take<4>(f01), ds0, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_take<2>(f10), ds1, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_take<0>(f11), ds1, acc[1], 0, 0, 0);
      }
    }
  };
  stage(IntC<0>{});
  {
    const int nfull = (N & (ST - 1)) != 0 ? nt - 1 : nt;      // tiles with all 64 keys in range
    int t = 0;
    for (; t + 1 < nfull; t += 2) {
      tile(t, IntC<0>{}, IntC<0>{});
      tile(t + 1, IntC<1>{}, IntC<0>{});
    }
    if (t < nfull) {
      tile(t, IntC<0>{}, IntC<0>{});
      if (t + 1 < nt) tile(t + 1, IntC<1>{}, IntC<1>{});
    } else if (t < nt) {
      tile(t, IntC<0>{}, IntC<1>{});
    }
  }
  if (!live) return;
  const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(dqkv + (size_t)mp.b * N * ldd, 0, (unsigned)N * (unsigned)ldd * 2u, 0x00020000);
  store_rows64(acc, qscale, rs_o, q < N ? ((unsigned)q * (unsigned)ldd + (unsigned)(mp.head * HD + 8 * h5)) * 2u : ROW_DROPPED);
}

// ---------------------------------------------------------------------------------------------------------------------
// L[q] and delta[q] vary along the accumulator ROWS here, so they come through LDS as raw copies (two 256-byte DMAs per tile) and
// the signs are moved to where they are free: the wave's K and V fragments are negated once, so the accumulators hold
// L - S and delta - dP; P = exp2(-(L - S)) is a source modifier of v_exp_f32, dS comes out negated and the final dK scale is
// -ln 2.  Out-of-range query rows need no mask: their Q', dO, L and delta all read as zero, so P = 1 meets dO = 0 and dS = 0.
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16_raw* __restrict__ qkv, const bf16_raw* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               bf16_raw* __restrict__ dqkv, int ldd, int N, int heads, int npairs) {
  // per stage: Q' (swz_k) | Q' (swz_v) | dO (swz_k) | dO (swz_v) | L [64] | delta [64]
  constexpr int STAGE = 4 * TILE + 2 * ST * 4;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5, l31 = lane & 31;
  const Map mp = decode(N, heads, npairs);
  if (!mp.ok) return;
  const int D = heads * HD, ld = 3 * D;
  const int k0 = mp.blk * WT + wave * 32;
  const bool live = k0 < N;                              // (wave-uniform) see the dQ kernel
  const bf16_raw* base = qkv + (size_t)mp.b * N * ld + mp.head * HD;
  const int key = k0 + l31, kc = key < N ? key : N - 1;

  bf16x8 kf[4], vf[4];                                                   // NEGATED K and V rows of this lane's key
  {
    const bf16_raw* kp = base + (size_t)kc * ld + D + 8 * h5;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4_t kk = *reinterpret_cast<const u32x4_t*>(kp + 16 * s), vv = *reinterpret_cast<const u32x4_t*>(kp + D + 16 * s);
#pragma unroll
      for (int e = 0; e < 4; ++e) { kk[e] ^= 0x80008000u; vv[e] ^= 0x80008000u; }
      kf[s] = __builtin_bit_cast(bf16x8, kk);
      vf[s] = __builtin_bit_cast(bf16x8, vv);
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }

  const int nt = (N + ST - 1) / ST;
  const auto rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(qkv + (size_t)mp.b * N * ld), 0, (unsigned)N * (unsigned)ld * 2u, 0x00020000);
  const auto rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(dout + (size_t)mp.b * N * D), 0, (unsigned)N * (unsigned)D * 2u, 0x00020000);
  const size_t stat0 = ((size_t)mp.b * heads + mp.head) * N;
  const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((wave & 1) ? delta + stat0 : lse + stat0), 0, (unsigned)N * 4u, 0x00020000);
  unsigned sq1[2], sq2[2], sd1[2], sd2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3), ch = lane & 7;
    sq1[i] = (unsigned)(row * ld + mp.head * HD + swz_k(row, ch) * 8) * 2u;
    sq2[i] = (unsigned)(row * ld + mp.head * HD + swz_v(row, ch) * 8) * 2u;
    sd1[i] = (unsigned)(row * D + mp.head * HD + swz_k(row, ch) * 8) * 2u;
    sd2[i] = (unsigned)(row * D + mp.head * HD + swz_v(row, ch) * 8) * 2u;
  }
  unsigned ss = (unsigned)lane * 4u;                                     // waves 0 / 1: the tile's 64 L / delta values
  const unsigned q_step = (unsigned)(ST * ld) * 2u, d_step = (unsigned)(ST * D) * 2u;
  auto stage = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* dst = smem + BUF * STAGE + (i * 32 + wave * 8) * 128;
      LDS_DMA16(rs_q, dst, sq1[i], 0);
      LDS_DMA16(rs_q, dst + TILE, sq2[i], 0);
      LDS_DMA16(rs_d, dst + 2 * TILE, sd1[i], 0);
      LDS_DMA16(rs_d, dst + 3 * TILE, sd2[i], 0);
      sq1[i] += q_step;
      sq2[i] += q_step;
      sd1[i] += d_step;
      sd2[i] += d_step;
    }
    if (wave < 2) {
      char* dst = smem + BUF * STAGE + 4 * TILE + wave * (ST * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (__attribute__((address_space(3))) void*)dst, 4, ss, 0, 0, 0);
      ss += ST * 4;
    }
  };
  int roff[4], toff[2];
#pragma unroll
  for (int sd = 0; sd < 4; ++sd) roff[sd] = l31 * 128 + swz_k(l31, 2 * sd + h5) * 16;
  {
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int row = 4 * h5 + (i16 >> 2);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const int dst = dt * 32 + g1 * 16 + 4 * (i16 & 3);
      toff[dt] = row * 128 + swz_v(row, dst >> 3) * 16 + (dst & 7) * 2;
    }
  }

  // LAST (compile time): the last query tile, whose second 32-query block may lie entirely past the last token -- its Q', dO, L and delta
  // are all zero there, so it contributes nothing and is skipped; the main-loop body carries no test for it.
  auto tile = [&](int t, auto bufc, auto lastc) {
    constexpr int BUF = decltype(bufc)::value;
    constexpr bool LAST = decltype(lastc)::value != 0;
    dma_landed_barrier();                                  // this wave's LDS-DMA pieces of tile t, then everyone's
    if (!LAST) stage(IntC<BUF ^ 1>{});
    if (!live) return;
    const char* sb = smem + BUF * STAGE;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if (LAST && qt == 1 && t * ST + 32 >= N) break;      // (wave-uniform)
      // accumulator rows r <-> query qt*32 + 4*h5 + (r&3) + 8*(r>>2): initial values L[q], delta[q] (four float4 each)
      f32x16 s, dp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(sb + 4 * TILE + (qt * 32 + 8 * g + 4 * h5) * 4);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(sb + 4 * TILE + ST * 4 + (qt * 32 + 8 * g + 4 * h5) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[4 * g + e] = l4[e]; dp[4 * g + e] = d4[e]; }
      }
#pragma unroll
      for (int sd = 0; sd < 4; ++sd) {
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(sb + qt * 4096 + roff[sd]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[sd], s, 0, 0, 0);                  // L - S
        const bf16x8 da = *reinterpret_cast<const bf16x8*>(sb + 2 * TILE + qt * 4096 + roff[sd]);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[sd], dp, 0, 0, 0);                // delta - dP
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __builtin_amdgcn_exp2f(-s[r]);                        // P
        dp[r] = s[r] * dp[r];                                        // -dS
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                           // four dO^T / Q^T fragments requested together, taken in order
        const char* rows = sb + (qt * 32 + ks * 16) * 128;
        TrFrag fv0 = tr_issue(rows + 3 * TILE, toff[0]), fk0 = tr_issue(rows + TILE, toff[0]);
        TrFrag fv1 = tr_issue(rows + 3 * TILE, toff[1]), fk1 = tr_issue(rows + TILE, toff[1]);
        const bf16x8 pb = pack_b(s, ks), dsb = pack_b(dp, ks);
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_take<6>(fv0), pb, dv[0], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_take<4>(fk0), dsb, dk[0], 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_take<2>(fv1), pb, dv[1], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_take<0>(fk1), dsb, dk[1], 0, 0, 0);
      }
    }
  };
  stage(IntC<0>{});
  {
    const int nmain = nt - 1;
    for (int t = 0; t < nmain; t += 2) {
      tile(t, IntC<0>{}, IntC<0>{});
      if (t + 1 < nmain) tile(t + 1, IntC<1>{}, IntC<0>{});
    }
    if (nmain & 1) tile(nmain, IntC<1>{}, IntC<1>{});
    else tile(nmain, IntC<0>{}, IntC<1>{});
  }
  if (!live) return;
  constexpr float NLN2 = -0.69314718055994531f;                          // dk holds -dS^T Q'
  const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(dqkv + (size_t)mp.b * N * ldd, 0, (unsigned)N * (unsigned)ldd * 2u, 0x00020000);
  const unsigned off = key < N ? ((unsigned)key * (unsigned)ldd + (unsigned)(D + mp.head * HD + 8 * h5)) * 2u : ROW_DROPPED;
  store_rows64(dk, NLN2, rs_o, off);
  store_rows64(dv, 1.f, rs_o, off + (unsigned)D * 2u);
}
#undef LDS_DMA16

}  // namespace ucod

extern "C" int ucod_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                                  int ld_dqkv, int B, int tok, int heads, void* stream) {
  UCOD_BF16_ONLY();
  using namespace ucod;
  if (!qkv || !out || !dout || !lse || !delta || !dqkv || B <= 0 || tok <= 0 || heads <= 0 || ld_dqkv < 3 * heads * HD || (ld_dqkv & 7))
    return UCOD_EINVAL;
  if ((size_t)tok * ld_dqkv * 2 >= (1ull << 31)) return UCOD_EINVAL;     // per-image output rows are addressed with 32-bit byte offsets (16-byte stores)
  UCOD_PROF(PROF_ATTN_BWD, stream);
  const int npairs = B * heads, nb = cdiv(tok, WT);
  dim3 grid(cdiv(npairs, 8) * 8 * nb), block(256);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, block, 0, s, (const bf16_raw*)qkv, (const bf16_raw*)out, (const bf16_raw*)dout, lse, delta,
                     (bf16_raw*)dqkv, ld_dqkv, tok, heads, npairs, 0.125f);
  UCOD_CHECK_LAUNCH();
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, block, 0, s, (const bf16_raw*)qkv, (const bf16_raw*)dout, lse, delta, (bf16_raw*)dqkv, ld_dqkv, tok,
                     heads, npairs);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
