// f32-equivalent GEMMs of the DBA decoder on the bf16 matrix pipe (models/modules/DBA.py:13,35 and its weight gradient): every f32
// operand is split into three bf16 terms  v = v1 + v2 + v3  (v1 = bf16(v), v2 = bf16(v - v1), v3 = bf16(v - v1 - v2): 24 significand
// bits, the subtractions are exact), and a product a*b is the sum of the six partial products whose weight is >= 2^-16,
//       a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1,
// each exact in the f32 accumulator of v_mfma_f32_32x32x16_bf16 (8 x 8 significand bits); what is dropped (a2 b3, a3 b2, a3 b3) is below
// 2^-24 of |a b|, the rounding an f32 FMA chain commits anyway.  Six bf16 MFMAs of 32 cycles replace eight v_mfma_f32_32x32x2_f32 of
// 64 cycles per 16-deep step: 2.7x less matrix-pipe time, and these two kernels (0.38 of the 1.3 ms decoder step) are what the
// pipelined training step actually pays for the decoder -- skipping them returns 0.35 ms of 10.0 (tools/ measurement in DESIGN.md).
// The exact-f32 kernels of gemm_f32.hip stay as the reference form (UCOD_DBA_EXACT_F32=1).
//
//   project: d[b][n][p] = sum_c W[n][c] x[b][c][p] + bias[n]      W pre-split once per call into bf16 planes (LDS-DMA), x split while staged
//   wgrad  : gW[n][c]  += sum_{b,p} gd[b][n][p] x[b][c][p]         both operands split while staged; split-K over (image, pixel chunk)
#include <type_traits>
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

typedef __attribute__((ext_vector_type(8))) __bf16 b16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 b16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 b16x2;
typedef __attribute__((ext_vector_type(2))) float f2_t;

__device__ __forceinline__ unsigned short bf16_rne(float v) {
  const f2_t t = {v, 0.f};
  return (unsigned short)(__builtin_bit_cast(unsigned, __builtin_convertvector(t, b16x2)) & 0xffffu);
}
__device__ __forceinline__ float bf16_up(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// v = t[0] + t[1] + t[2] up to 2^-25 |v|
__device__ __forceinline__ void split3(float v, unsigned short (&t)[3]) {
  t[0] = bf16_rne(v);
  const float r1 = v - bf16_up(t[0]);
  t[1] = bf16_rne(r1);
  const float r2 = r1 - bf16_up(t[1]);
  t[2] = bf16_rne(r2);
}

// position of reduction index k (0..15) inside a 16-element row of an MFMA-operand image: the transpose-read of the [k][column] image
// hands lane-half h the k set {4h..4h+3, 8+4h..8+4h+3}, so the K-contiguous operand keeps its 16 elements in the order
// [0-3, 8-11 | 4-7, 12-15] and each half is one 16-byte read.
__device__ __host__ __forceinline__ int kpos(int k) { return (k & 3) | ((k & 8) >> 1) | ((k & 4) << 1); }

// ------------------------------------------------------------------------------------------- W -> three bf16 planes [3][Nout][C]
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wp, int rows, int C) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * C) return;
  const int r = (int)(i / C), c = (int)(i - (long)r * C);
  unsigned short t[3];
  split3(W[i], t);
  const long o = (long)r * C + (c & ~15) + kpos(c & 15);
#pragma unroll
  for (int s = 0; s < 3; ++s) Wp[(long)s * rows * C + o] = t[s];
}

// LDS fragment reads issued from asm: hipcc's wait-count pass puts `s_waitcnt vmcnt(0)` in front of every LDS read it can see while an
// LDS-DMA (or, across the loop back-edge, any load) may be in flight, which would expose the latency of the operands requested for the
// NEXT tiles on every tile.  The reads below are invisible to it; lds_join() is the explicit wait, pin() ties a fragment to it.
__device__ __forceinline__ b16x8 lds_read_b128_asm(unsigned addr, int imm) {
  b16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(imm));
  return r;
}
__device__ __forceinline__ b16x4 lds_read_tr16_asm(unsigned addr, int imm) {
  b16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(imm));
  return r;
}
__device__ __forceinline__ void lds_join() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void lds_join_counted() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }   // all but the N youngest LDS operations
__device__ __forceinline__ void lds_write_b32_asm(unsigned addr, unsigned v, int imm) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(imm) : "memory");
}
template <typename T>
__device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }

constexpr int XROW = 128;                 // bytes per k-row of the x image: 64 pixels x bf16
__device__ __forceinline__ int swz_x(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }   // as the attention V tile

// ------------------------------------------------------------------------------------------- project
// block: 8 waves = NR*128 output channels x 192 pixels of one image (1369 = 7 x 192 + 25: 8 x B workgroups, ONE per CU at B = 32).
// Wave (wr, wc): channels 32 NR wr .. + 32 NR - 1, pixels 96 wc .. + 95 (three 32-pixel column blocks).  K-tile 16 input channels.
// A tile lasts ~1.4 us, less than a memory round trip under load, so every operand is requested TWO tiles ahead: the W planes by
// LDS-DMA into a ring of three stages, x into three rotating register sets (split + written to LDS one tile before use).  LDS per stage:
// W planes [3][NR*128 rows][32 B] (the two 16-byte halves of a row swapped on odd 8-row groups: conflict-free ds_read_b128), x planes
// [3][3 groups of 64 pixels][16 k][128 B] read back transposed by ds_read_b64_tr_b16 (the attention V-tile layout).  Every LDS access of
// the loop is issued from asm and waited for by hand: hipcc would put `s_waitcnt vmcnt(0)` in front of each one while a DMA is in flight.
// (Measured alternatives: 64- and 96-pixel tiles with two workgroups per CU -- the W planes re-streamed per workgroup sit at the
// L2 -> LDS limit, and two co-resident workgroups overlapped only 18 %: 157 us; this form: see DESIGN.md.)
constexpr int PXT = 192;
template <int NR>
__global__ __launch_bounds__(512, 2) void dba_project_b3_kernel(const float* __restrict__ x, const unsigned short* __restrict__ Wp,
                                                                const float* __restrict__ bias, float* __restrict__ d, int C, int HW, int Nout) {
  constexpr int ROWS = NR * 128;
  constexpr int WPL = ROWS * 32;           // bytes of one W plane per stage
  constexpr int XG = 16 * XROW;            // one 64-pixel group of one x plane
  constexpr int XPL = 3 * XG;              // bytes of one x plane per stage
  constexpr int STAGE = 3 * WPL + 3 * XPL;
  __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int h5 = lane >> 5, l31 = lane & 31;
  const int p0 = blockIdx.x * PXT, b = blockIdx.y;
  const float* xb = x + (long)b * C * HW;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  // W DMA: one instruction = 64 lanes x 16 B = 32 rows; lane l -> row l/2, slot l&1 holding k-half (l&1) ^ ((row >> 3) & 1).
  // 3 planes x 4 NR groups of 32 rows = 12 NR instructions per stage; NDMA per wave (waves past the count repeat the last one, so that
  // every wave has the same number of vector-memory operations in flight and one counted wait fits all)
  constexpr int NDMA = (12 * NR + 7) / 8;
  const int wrow = lane >> 1, whalf = (lane & 1) ^ ((wrow >> 3) & 1);
  auto stage_w = [&](int t, auto bufc) {
    constexpr int BUFW = decltype(bufc)::value;
    char* base = smem + BUFW * STAGE;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
      int q = wave + 8 * j;                                     // (plane, group)
      q = q < 12 * NR ? q : 12 * NR - 1;
      const int s = q / (4 * NR), g = q - s * (4 * NR), row = g * 32 + wrow;
      const unsigned short* src = Wp + ((long)s * Nout + (row < Nout ? row : Nout - 1)) * C + t * 16 + whalf * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + s * WPL + g * 1024), 16, 0, 0);
    }
  };
  // x: 16 k x 96 pixel PAIRS = 3 pairs per thread, pair index e = tid + 512 i -> (k, pair) = (e / 96, e % 96); the two pixels of a pair
  // go to LDS as one dword per plane.  Loads are clamped, not branched around.
  int xsrc[3][2], xdst[3];
  bool xin[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = tid + 512 * i, k = e / 96, col = 2 * (e - k * 96);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int px = p0 + col + j;
      xin[i][j] = px < HW;
      xsrc[i][j] = k * HW + (px < HW ? px : HW - 1);
    }
    xdst[i] = 3 * WPL + (col >> 6) * XG + k * XROW + swz_x(k, (col & 63) >> 3) * 16 + (col & 7) * 2;
  }
  float rx[3][6];                                                // x(j) travels in rx[j % 3]
  auto load_x = [&](int t, float (&r)[6]) {
    const float* xt = xb + (long)t * 16 * HW;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) r[2 * i + j] = xt[xsrc[i][j]];
  };
  auto store_x = [&](auto bufc, const float (&r)[6]) {
    constexpr int BUFX = decltype(bufc)::value;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      unsigned short t0[3], t1[3];
      split3(xin[i][0] ? r[2 * i] : 0.f, t0);
      split3(xin[i][1] ? r[2 * i + 1] : 0.f, t1);
#pragma unroll
      for (int s = 0; s < 3; ++s) lds_write_b32_asm(lds0 + (unsigned)(xdst[i] + BUFX * STAGE), (unsigned)t0[s] | ((unsigned)t1[s] << 16), s * XPL);
    }
  };
  // fragment addresses
  unsigned aaddr[NR], xaddr[3];
#pragma unroll
  for (int i = 0; i < NR; ++i) {                                 // W fragment of row-block i: row = 32 (NR wr + i) + l31, slot h5 ^ ((row>>3)&1)
    const int row = 32 * (NR * wr + i) + l31;
    aaddr[i] = lds0 + (unsigned)(row * 32 + ((h5 ^ ((row >> 3) & 1)) * 16));
  }
  {
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int krow = 4 * h5 + (i16 >> 2);
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {                             // x fragment of column block cb (pixels 96 wc + 32 cb ..) via ds_read_b64_tr_b16
      const int dst = 96 * wc + 32 * cb + g1 * 16 + 4 * (i16 & 3);
      xaddr[cb] = lds0 + (unsigned)(3 * WPL + (dst >> 6) * XG + krow * XROW + swz_x(krow, (dst & 63) >> 3) * 16 + (dst & 7) * 2);
    }
  }

  f32x16 acc[NR][3];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][cb][r] = 0.f;

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  const int nt = C / 16;
  // prologue.  Vector-memory issue order from here on: W(0) x(0) | W(1) x(1) x(2) | tile t: W(t+2) x(t+3)
  stage_w(0, I0{});
  load_x(0, rx[0]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_x(I0{}, rx[0]);
  if (nt > 1) {
    stage_w(1, I1{});
    load_x(1, rx[1]);
  }
  if (nt > 2) load_x(2, rx[2]);
  constexpr int IN_FLIGHT = NDMA + 12;                           // behind W(t): x(t+1) [may be needed: see below], W(t+1), x(t+2) -- or W(t+1) x(t+2) | W(t+2)..
  auto tile = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value, B1 = (BUF + 1) % 3, B2 = (BUF + 2) % 3;
    // this wave's W(t) DMAs have landed, its x(t) writes are done; then everyone's.  Younger than W(t) and allowed to fly on:
    // x(t+1)? no -- it was waited for by the split of the previous tile; W(t+1) [NDMA], x(t+2) [6], x(t+3)? not issued yet ... at most NDMA + 12
    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(IN_FLIGHT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    b16x8 af[NR][3], xf[3][3];
    b16x4 lo[3][3], hi[3][3];
    constexpr int RPP = NR + 6;                                  // LDS reads per plane
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < NR; ++i) af[i][s] = lds_read_b128_asm(aaddr[i] + BUF * STAGE, s * WPL);   // (stage offset in the register: 16-bit immediates)
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) {
        lo[cb][s] = lds_read_tr16_asm(xaddr[cb] + BUF * STAGE, s * XPL);
        hi[cb][s] = lds_read_tr16_asm(xaddr[cb] + BUF * STAGE, s * XPL + 8 * XROW);
      }
    }
    if (t + 2 < nt) stage_w(t + 2, std::integral_constant<int, B2>{});
    auto plane_ready = [&](auto sc) {
      constexpr int S = decltype(sc)::value;
      lds_join_counted<((2 - S) * RPP > 15 ? 15 : (2 - S) * RPP)>();     // (the counter saturates at 15: a little stricter for plane 0)
#pragma unroll
      for (int i = 0; i < NR; ++i) pin(af[i][S]);
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) {
        pin(lo[cb][S]);
        pin(hi[cb][S]);
        xf[cb][S] = (b16x8){lo[cb][S][0], lo[cb][S][1], lo[cb][S][2], lo[cb][S][3], hi[cb][S][0], hi[cb][S][1], hi[cb][S][2], hi[cb][S][3]};
      }
    };
    auto product = [&](auto sac, auto sxc) {                     // one partial product over all accumulators (consecutive MFMAs never share one)
      constexpr int SA = decltype(sac)::value, SX = decltype(sxc)::value;
#pragma unroll
      for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][SA], xf[cb][SX], acc[i][cb], 0, 0, 0);
    };
    // products in the order their planes arrive; the split of x(t+1) (VALU + asm LDS writes) is free to sit among the MFMAs
    plane_ready(I0{});
    product(I0{}, I0{});
    plane_ready(I1{});
    product(I0{}, I1{});
    product(I1{}, I0{});
    product(I1{}, I1{});
    plane_ready(I2{});
    product(I0{}, I2{});
    product(I2{}, I0{});
    if (t + 1 < nt) store_x(std::integral_constant<int, B1>{}, rx[B1]);       // x(t+1), requested two tiles ago
    if (t + 3 < nt) load_x(t + 3, rx[BUF]);                                    // x(t+3) takes the registers x(t) left
  };
  for (int t = 0; t < nt; t += 3) {
    tile(t, I0{});
    if (t + 1 < nt) tile(t + 1, I1{});
    if (t + 2 < nt) tile(t + 2, I2{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // C/D map of the 32x32 accumulator: col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel)
  // the lane's 16 NR bias values in ONE batch of loads (round 3: `acc + bias[n]` inside the store loop was a load, a vmcnt(0) and a store per
  // element, and the `n < Nout` guard a branch per store -- 96 serialised round trips per thread at the end of every workgroup)
  float* db = d + (long)b * Nout * HW;
  float bv[NR][16];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 32 * (NR * wr + i) + (r & 3) + 8 * (r >> 2) + 4 * h5;
      bv[i][r] = bias[n];                                         // (Nout == NR * 128: the entry point admits nothing else)
    }
#pragma unroll
  for (int cb = 0; cb < 3; ++cb) {
    const int p = p0 + 96 * wc + 32 * cb + l31;
    if (p >= HW) continue;
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * (NR * wr + i) + (r & 3) + 8 * (r >> 2) + 4 * h5;
        db[(long)n * HW + p] = acc[i][cb][r] + bv[i][r];           // no per-element guard: a branch per store is a vmcnt(0) per store
      }
  }
}

// ------------------------------------------------------------------------------------------- wgrad
// gW[n][c] += sum over one (image, pixel chunk) of gd[n][p] x[c][p]: both operands are f32 activations with the reduction index (pixels)
// contiguous, so both are split while staged and both fragments are plain ds_read_b128 from [row][16 k] images (32-byte rows, the two
// halves swapped on odd 8-row groups).  block: 8 waves = all 128 n x CT = 384 channels (C = 768: two channel tiles, x is read exactly
// once); wave (wr, wc): n 64 wr .., channels 96 wc ..; K-tile 16 pixels; ring of three stages, loads two K-tiles ahead in registers.
// Split-K over (image, pixel chunk) with f32 atomics into gW, as the f32 kernel.
constexpr int WCT = 384;
__global__ __launch_bounds__(512, 2) void dba_wgrad_b3_kernel(const float* __restrict__ gd, const float* __restrict__ x, float* __restrict__ gW,
                                                              int C, int HW, int chunk) {
  constexpr int ROWS = 128 + WCT;          // gd rows 0..127 | x rows 128..
  constexpr int PL = ROWS * 32;            // bytes of one plane per stage
  constexpr int STAGE = 3 * PL;
  __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int h5 = lane >> 5, l31 = lane & 31;
  const int c0 = blockIdx.x * WCT, b = blockIdx.z;
  const int ps = blockIdx.y * chunk, pe = min(ps + chunk, HW);
  const float* gdb = gd + (long)b * 128 * HW;
  const float* xb = x + (long)b * C * HW;
  const unsigned lds0 = (unsigned)(uintptr_t)smem;

  // staging: 512 rows x 8 pixel pairs = 4096 pairs, 8 per thread: pair e = tid + 512 i -> (row, pair) = (e / 8, e % 8); a load instruction
  // covers 8 rows x 64 contiguous bytes
  constexpr int NP = ROWS * 8 / 512;       // 8: row = tid / 8 + 64 i, so i < 2 are gd rows and i >= 2 are x rows; the pair index is tid % 8 for all
  const int kp = 2 * (tid & 7);
  int rbase[NP];                           // element offset of the row in its tensor (rows past C clamped, masked at the split)
  unsigned wdst[NP];
  bool rowok[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = (tid >> 3) + 64 * i;
    const int c = c0 + row - 128;
    rowok[i] = i < 2 || c < C;
    rbase[i] = (i < 2 ? row : (c < C ? c : C - 1)) * HW;
    wdst[i] = lds0 + (unsigned)(row * 32 + (((kp >> 3) ^ ((row >> 3) & 1)) * 16) + (kp & 7) * 2);
  }
  float rv[3][2 * NP];
  auto load_t = [&](int t, float (&r)[2 * NP]) {
    const int p = ps + t * 16 + kp;
    const int p0c = p < pe ? p : pe - 1, p1c = p + 1 < pe ? p + 1 : pe - 1;     // clamped (pe >= 1); masked to zero at the split
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const float* src = (i < 2 ? gdb : xb) + rbase[i];
      r[2 * i] = src[p0c];
      r[2 * i + 1] = src[p1c];
    }
  };
  auto store_t = [&](int t, auto bufc, const float (&r)[2 * NP]) {
    constexpr int BUFX = decltype(bufc)::value;
    const int p = ps + t * 16 + kp;
    const bool in0 = p < pe, in1 = p + 1 < pe;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      unsigned short t0[3], t1[3];
      split3((rowok[i] && in0) ? r[2 * i] : 0.f, t0);
      split3((rowok[i] && in1) ? r[2 * i + 1] : 0.f, t1);
#pragma unroll
      for (int s = 0; s < 3; ++s) lds_write_b32_asm(wdst[i] + BUFX * STAGE, (unsigned)t0[s] | ((unsigned)t1[s] << 16), s * PL);
    }
  };
  unsigned aaddr[2], baddr[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 64 * wr + 32 * i + l31;
    aaddr[i] = lds0 + (unsigned)(row * 32 + ((h5 ^ ((row >> 3) & 1)) * 16));
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int row = 128 + 96 * wc + 32 * j + l31;
    baddr[j] = lds0 + (unsigned)(row * 32 + ((h5 ^ ((row >> 3) & 1)) * 16));
  }
  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  const int nt = (pe - ps + 15) / 16;
  load_t(0, rv[0]);
  if (nt > 1) load_t(1, rv[1]);
  if (nt > 2) load_t(2, rv[2]);
  store_t(0, I0{}, rv[0]);                                       // (the compiler waits for rv[0] only: the loads retire in order)
  auto tile = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value, B1 = (BUF + 1) % 3;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // this wave's LDS writes of tile t are done; then everyone's
    b16x8 af[2][3], bf[3][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i][s] = lds_read_b128_asm(aaddr[i] + BUF * STAGE, s * PL);
#pragma unroll
      for (int j = 0; j < 3; ++j) bf[j][s] = lds_read_b128_asm(baddr[j] + BUF * STAGE, s * PL);
    }
    auto plane_ready = [&](auto sc) {
      constexpr int S = decltype(sc)::value;
      lds_join_counted<(2 - S) * 5>();
#pragma unroll
      for (int i = 0; i < 2; ++i) pin(af[i][S]);
#pragma unroll
      for (int j = 0; j < 3; ++j) pin(bf[j][S]);
    };
    auto product = [&](auto sac, auto sbc) {
      constexpr int SA = decltype(sac)::value, SB = decltype(sbc)::value;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][SA], bf[j][SB], acc[i][j], 0, 0, 0);
    };
    plane_ready(I0{});
    product(I0{}, I0{});
    plane_ready(I1{});
    product(I0{}, I1{});
    product(I1{}, I0{});
    product(I1{}, I1{});
    plane_ready(I2{});
    product(I0{}, I2{});
    product(I2{}, I0{});
    if (t + 1 < nt) store_t(t + 1, std::integral_constant<int, B1>{}, rv[B1]);   // requested two tiles ago
    if (t + 3 < nt) load_t(t + 3, rv[BUF]);
  };
  for (int t = 0; t < nt; t += 3) {
    tile(t, I0{});
    if (t + 1 < nt) tile(t + 1, I1{});
    if (t + 2 < nt) tile(t + 2, I2{});
  }
  // C/D map: col = lane&31 (channel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (n)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = c0 + 96 * wc + 32 * j + l31;
      if (c >= C) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 64 * wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h5;
        atomicAdd(&gW[(long)n * C + c], acc[i][j][r]);
      }
    }
}

}  // namespace ucod

extern "C" size_t ucod_dba_project_split_workspace_bytes(int C, int Nout) { return (size_t)3 * Nout * C * sizeof(unsigned short); }

extern "C" int ucod_dba_project_split(const float* x, const float* W, const float* bias, float* d, void* ws, size_t ws_bytes, int B, int C,
                                      int HW, int Nout, void* stream) {
  using namespace ucod;
  if (!x || !W || !bias || !d || !ws || B <= 0 || C <= 0 || HW <= 0 || (C % 16) != 0 || (Nout != 128 && Nout != 256) ||
      ws_bytes < ucod_dba_project_split_workspace_bytes(C, Nout) || (((uintptr_t)ws) % 16) != 0)
    return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  UCOD_PROF(PROF_DBA_PROJECT, s);
  hipLaunchKernelGGL(split3_rows_kernel, dim3(cdiv((long)Nout * C, 256)), dim3(256), 0, s, W, (unsigned short*)ws, Nout, C);
  dim3 grid(cdiv(HW, PXT), B), block(512);
  if (Nout == 256) hipLaunchKernelGGL(dba_project_b3_kernel<2>, grid, block, 0, s, x, (const unsigned short*)ws, bias, d, C, HW, Nout);
  else hipLaunchKernelGGL(dba_project_b3_kernel<1>, grid, block, 0, s, x, (const unsigned short*)ws, bias, d, C, HW, Nout);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_dba_wgrad_split(const float* gd, const float* x, float* gW, int B, int C, int HW, void* stream) {
  using namespace ucod;
  if (!gd || !x || !gW || B <= 0 || C <= 0 || HW <= 0) return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int ctiles = cdiv(C, WCT);
  // split-K granularity: about one workgroup per CU (chunk a multiple of the 16-pixel K-tile)
  int nchunk = max(1, 256 / (ctiles * B));
  int chunk = cdiv(cdiv(HW, nchunk), 16) * 16;
  nchunk = cdiv(HW, chunk);
  UCOD_PROF(PROF_DBA_WGRAD, s);
  if (!ucod::accumulators_prezeroed()) {
    hipError_t e = hipMemsetAsync(gW, 0, sizeof(float) * 128 * (size_t)C, s);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(dba_wgrad_b3_kernel, dim3(ctiles, nchunk, B), dim3(512), 0, s, gd, x, gW, C, HW, chunk);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
