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
template <typename T>
__device__ __forceinline__ void pin(T& v) { asm volatile("" : "+v"(v)); }

constexpr int XROW = 128;                 // bytes per k-row of the x image: 64 pixels x bf16
__device__ __forceinline__ int swz_x(int row, int chunk) { return chunk ^ (((row >> 1) & 1) << 2); }   // as the attention V tile

// ------------------------------------------------------------------------------------------- project
// block: 4 waves = NR*128 output channels x 96 pixels of one image, TWO workgroups per CU (72 KB of LDS each) so that one's matrix
// phase runs beside the other's staging / split / barrier phase (a single 8-wave workgroup per CU ran every phase in lockstep: 50 % matrix
// pipe).  1369 = 14 x 96 + 25: 15 x B = 480 workgroups on 512 slots at B = 32.  Wave w: channels 32 NR w .. + 32 NR - 1, all 96 pixels
// (three 32-pixel column blocks).  K-tile 16 input channels.  LDS per stage: W planes [3][NR*128 rows][32 B] by LDS-DMA (the two 16-byte
// halves of a row swapped on odd 8-row groups: conflict-free ds_read_b128), x planes [3][2 groups of 64 pixels][16 k][128 B] written
// after the split and read back transposed by ds_read_b64_tr_b16 (the attention V-tile layout).
constexpr int PXT = 96;
template <int NR>
__global__ __launch_bounds__(256, 2) void dba_project_b3_kernel(const float* __restrict__ x, const unsigned short* __restrict__ Wp,
                                                                const float* __restrict__ bias, float* __restrict__ d, int C, int HW, int Nout) {
  constexpr int ROWS = NR * 128;
  constexpr int WPL = ROWS * 32;           // bytes of one W plane per stage
  constexpr int XG = 16 * XROW;            // one 64-pixel group of one x plane
  constexpr int XPL = 2 * XG;              // bytes of one x plane per stage (pixels 64..95 use half of the second group)
  constexpr int STAGE = 3 * WPL + 3 * XPL;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5, l31 = lane & 31;
  const int p0 = blockIdx.x * PXT, b = blockIdx.y;
  const float* xb = x + (long)b * C * HW;

  // W DMA: one instruction = 64 lanes x 16 B = 32 rows; lane l -> row l/2, slot l&1 holding k-half (l&1) ^ ((row >> 3) & 1).
  // 3 planes x 4 NR groups of 32 rows = 12 NR instructions per stage, 3 NR per wave
  const int wrow = lane >> 1, whalf = (lane & 1) ^ ((wrow >> 3) & 1);
  auto stage_w = [&](int t, int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < 3 * NR; ++j) {
      const int q = wave + 4 * j;                               // (plane, group)
      const int s = q / (4 * NR), g = q - s * (4 * NR), row = g * 32 + wrow;
      const unsigned short* src = Wp + ((long)s * Nout + (row < Nout ? row : Nout - 1)) * C + t * 16 + whalf * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + s * WPL + g * 1024), 16, 0, 0);
    }
  };
  // x: 16 k x 48 pixel PAIRS = 3 pairs per thread, pair index e = tid + 256 i -> (k, pair) = (e / 48, e % 48); the two pixels of a pair
  // go to LDS as one dword per plane (a ds_write_b16 per pixel is a 2-way bank conflict).  Loads are clamped, not branched around.
  int xsrc[3][2], xdst[3];
  bool xin[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = tid + 256 * i, k = e / 48, col = 2 * (e - k * 48);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int px = p0 + col + j;
      xin[i][j] = px < HW;
      xsrc[i][j] = k * HW + (px < HW ? px : HW - 1);
    }
    xdst[i] = (col >> 6) * XG + k * XROW + swz_x(k, (col & 63) >> 3) * 16 + (col & 7) * 2;
  }
  float rxa[6], rxb[6];
  auto load_x = [&](int t, float (&rx)[6]) {
    const float* xt = xb + (long)t * 16 * HW;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) rx[2 * i + j] = xt[xsrc[i][j]];
  };
  auto store_x = [&](int buf, const float (&rx)[6]) {
    char* base = smem + buf * STAGE + 3 * WPL;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      unsigned short t0[3], t1[3];
      split3(xin[i][0] ? rx[2 * i] : 0.f, t0);
      split3(xin[i][1] ? rx[2 * i + 1] : 0.f, t1);
#pragma unroll
      for (int s = 0; s < 3; ++s) *reinterpret_cast<unsigned*>(base + s * XPL + xdst[i]) = (unsigned)t0[s] | ((unsigned)t1[s] << 16);
    }
  };
  // fragment addresses
  int aoff[NR];                                                  // W fragment of row-block i: row = 32 (NR wave + i) + l31, slot h5 ^ ((row>>3)&1)
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int row = 32 * (NR * wave + i) + l31;
    aoff[i] = row * 32 + ((h5 ^ ((row >> 3) & 1)) * 16);
  }
  int xoff[3];                                                   // x fragment of column block cb (pixels 32 cb ..) via ds_read_b64_tr_b16
  {
    const int i16 = lane & 15, g1 = (lane >> 4) & 1;
    const int krow = 4 * h5 + (i16 >> 2);
#pragma unroll
    for (int cb = 0; cb < 3; ++cb) {
      const int dst = 32 * cb + g1 * 16 + 4 * (i16 & 3);        // first pixel of this lane's 4-pixel piece
      xoff[cb] = 3 * WPL + (dst >> 6) * XG + krow * XROW + swz_x(krow, (dst & 63) >> 3) * 16 + (dst & 7) * 2;
    }
  }

  f32x16 acc[NR][3];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int cb = 0; cb < 3; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][cb][r] = 0.f;

  const int nt = C / 16;
  stage_w(0, 0);
  load_x(0, rxa);
  store_x(0, rxa);
  if (nt > 1) load_x(1, rxb);
  // x (HBM) travels two tiles ahead in registers, W (L2) one tile ahead by LDS-DMA.  Program order inside a tile: fragment reads,
  // DMA of W(t+1), products, split + LDS writes of x(t+1), loads of x(t+2) -- so that the `s_waitcnt vmcnt` hipcc puts in front of any LDS
  // access that follows an LDS-DMA in flight never covers a request younger than a tile.
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  unsigned aaddr[NR], xaddr[3];
#pragma unroll
  for (int i = 0; i < NR; ++i) aaddr[i] = lds0 + (unsigned)aoff[i];
#pragma unroll
  for (int cb = 0; cb < 3; ++cb) xaddr[cb] = lds0 + (unsigned)xoff[cb];
  auto tile = [&](int t, auto bufc, float (&r_next)[6], float (&r_far)[6]) {   // r_next holds x(t+1), r_far receives x(t+2)
    constexpr int BUF = decltype(bufc)::value;
    // this wave's W(t) DMAs have landed (only the six loads of x(t+1) may still fly), its x(t) writes are done; then everyone's
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    b16x8 af[NR][3], xf[3][3];
    b16x4 lo[3][3], hi[3][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < NR; ++i) af[i][s] = lds_read_b128_asm(aaddr[i], BUF * STAGE + s * WPL);
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) {
        lo[cb][s] = lds_read_tr16_asm(xaddr[cb], BUF * STAGE + s * XPL);       // (xoff already carries the 3 WPL offset)
        hi[cb][s] = lds_read_tr16_asm(xaddr[cb], BUF * STAGE + s * XPL + 8 * XROW);
      }
    }
    if (t + 1 < nt) stage_w(t + 1, BUF ^ 1);
    lds_join();
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < NR; ++i) pin(af[i][s]);
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) {
        pin(lo[cb][s]);
        pin(hi[cb][s]);
        xf[cb][s] = (b16x8){lo[cb][s][0], lo[cb][s][1], lo[cb][s][2], lo[cb][s][3], hi[cb][s][0], hi[cb][s][1], hi[cb][s][2], hi[cb][s][3]};
      }
    }
    // smallest partial products first; the accumulators in the inner loops: consecutive MFMAs never touch the same accumulator
    constexpr int SA[6] = {2, 0, 1, 1, 0, 0}, SX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int cb = 0; cb < 3; ++cb)
          acc[i][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][SA[q]], xf[cb][SX[q]], acc[i][cb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < nt) store_x(BUF ^ 1, r_next);                    // (at the END of the tile: x(t+1) was requested a whole tile ago)
    __builtin_amdgcn_sched_barrier(0);
    if (t + 2 < nt) load_x(t + 2, r_far);
  };
  for (int t = 0; t < nt; t += 2) {
    tile(t, std::integral_constant<int, 0>{}, rxb, rxa);
    if (t + 1 < nt) tile(t + 1, std::integral_constant<int, 1>{}, rxa, rxb);
  }
  // C/D map of the 32x32 accumulator: col = lane&31 (pixel), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (channel)
  float* db = d + (long)b * Nout * HW;
#pragma unroll
  for (int cb = 0; cb < 3; ++cb) {
    const int p = p0 + 32 * cb + l31;
    if (p >= HW) continue;
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * (NR * wave + i) + (r & 3) + 8 * (r >> 2) + 4 * h5;
        if (n < Nout) db[(long)n * HW + p] = acc[i][cb][r] + bias[n];
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
  dim3 grid(cdiv(HW, PXT), B), block(256);
  if (Nout == 256) hipLaunchKernelGGL(dba_project_b3_kernel<2>, grid, block, 0, s, x, (const unsigned short*)ws, bias, d, C, HW, Nout);
  else hipLaunchKernelGGL(dba_project_b3_kernel<1>, grid, block, 0, s, x, (const unsigned short*)ws, bias, d, C, HW, Nout);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
