// Probe for the K=768 GEMM drain: what does it cost to turn the 16x16 f32 accumulator tiles of one wave (128 x 48 wave
// tile = 24 tiles, lane l = 16 g + c holds rows 4g..4g+3 of column c) into 16-byte bf16 row chunks WITHOUT the LDS round trip?
// Network per PAIR of tiles (8 values per lane), inside every group of 8 lanes (an 8 x 8 transpose lane <-> value index):
//   stage 0: exchange with lane ^ 1 (quad_perm), 8 v_cndmask_b32_dpp;   pack adjacent columns: 4 v_cvt_pk_bf16_f32
//   stage 1: exchange with lane ^ 2 (quad_perm), 4 v_cndmask_b32_dpp;   stage 2: lane ^ 4 (row_shl:4 / row_shr:4), 4 v_cndmask_b32_dpp
// = 20 VALU + 6 s_mov of VCC per pair, 240 VALU per wave tile.  Afterwards lane (g, c3, L) holds columns 8 c3..8 c3 + 7 of row 4 g + (L & 3)
// of tile (L >> 2) of the pair: one 16-byte store.
// The probe (1) checks the result against the plain conversion, (2) times REPS passes of the 12-pair network with 1 and 2 waves per SIMD and
// a calibration loop of the same number of independent v_add_f32, so that the cost reads in VALU issue slots.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/acc_transpose_probe.hip -o /tmp/accT && /tmp/accT
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;

// one pair of tiles: x[0..3] = tile 0 rows 4g..4g+3, x[4..7] = tile 1; result p[0..3] = 8 packed bf16 (16 bytes)
__device__ __forceinline__ void pair_network(float (&x)[8], unsigned (&p)[4], u64 m0, u64 m0n, u64 m1, u64 m1n, u64 m2, u64 m2n) {
  float h1, h3, h5, h7, l0, l2, l4, l6;
  asm volatile(
      "s_mov_b64 vcc, %8\n"
      "v_cndmask_b32_dpp %0, %10, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_dpp %1, %12, %13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_dpp %2, %14, %15, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_dpp %3, %16, %17, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 vcc, %9\n"
      "v_cndmask_b32_dpp %4, %11, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_dpp %5, %13, %12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_dpp %6, %15, %14, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_dpp %7, %17, %16, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      : "=&v"(h1), "=&v"(h3), "=&v"(h5), "=&v"(h7), "=&v"(l0), "=&v"(l2), "=&v"(l4), "=&v"(l6)
      : "s"(m0), "s"(m0n), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7])
      : "vcc");
  unsigned q0, q1, q2, q3;
  asm volatile(
      "v_cvt_pk_bf16_f32 %0, %4, %5\n"
      "v_cvt_pk_bf16_f32 %1, %6, %7\n"
      "v_cvt_pk_bf16_f32 %2, %8, %9\n"
      "v_cvt_pk_bf16_f32 %3, %10, %11\n"
      : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
      : "v"(l0), "v"(h1), "v"(l2), "v"(h3), "v"(l4), "v"(h5), "v"(l6), "v"(h7));
  unsigned r0, r1, r2, r3;
  asm volatile(
      "s_mov_b64 vcc, %4\n"
      "s_nop 0\n"
      "v_cndmask_b32_dpp %1, %6, %7, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_dpp %3, %8, %9, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "s_mov_b64 vcc, %5\n"
      "v_cndmask_b32_dpp %0, %7, %6, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_dpp %2, %9, %8, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "s"(m1), "s"(m1n), "v"(q0), "v"(q1), "v"(q2), "v"(q3)
      : "vcc");
  asm volatile(
      "s_mov_b64 vcc, %4\n"
      "s_nop 0\n"
      "v_cndmask_b32_dpp %2, %6, %8, vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_cndmask_b32_dpp %3, %7, %9, vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "s_mov_b64 vcc, %5\n"
      "v_cndmask_b32_dpp %0, %8, %6, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      "v_cndmask_b32_dpp %1, %9, %7, vcc row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
      : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3])
      : "s"(m2), "s"(m2n), "v"(r0), "v"(r1), "v"(r2), "v"(r3)
      : "vcc");
}

__device__ __forceinline__ u64 lane_bit_mask(int b) {   // bit l set when lane l has bit b set
  return b == 0 ? 0xAAAAAAAAAAAAAAAAull : b == 1 ? 0xCCCCCCCCCCCCCCCCull : 0xF0F0F0F0F0F0F0F0ull;
}

// src: [128][48] f32 per wave; out: [128][48] bf16 per wave
__global__ void __launch_bounds__(256) transpose_check(const float* __restrict__ src, __hip_bfloat16* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const float* s = src + (size_t)wave * 128 * 48;
  uint4* o = reinterpret_cast<uint4*>(out + (size_t)wave * 128 * 48);
  const int g = lane >> 4, c = lane & 15, c3 = (lane >> 3) & 1, L = lane & 7;
  const u64 m0 = lane_bit_mask(0), m1 = lane_bit_mask(1), m2 = lane_bit_mask(2);
  for (int pr = 0; pr < 12; ++pr) {            // pair pr = tiles 2 pr, 2 pr + 1 in (row tile, column tile) order
    float x[8];
    int rt[2], ct[2];
    for (int t = 0; t < 2; ++t) {
      const int tile = 2 * pr + t;
      rt[t] = tile / 3; ct[t] = tile % 3;
      for (int r = 0; r < 4; ++r) x[4 * t + r] = s[(rt[t] * 16 + 4 * g + r) * 48 + ct[t] * 16 + c];
    }
    unsigned p[4];
    pair_network(x, p, m0, ~m0, m1, ~m1, m2, ~m2);
    const int t = L >> 2, row = rt[t] * 16 + 4 * g + (L & 3), col = ct[t] * 16 + 8 * c3;
    o[(row * 48 + col) >> 3] = make_uint4(p[0], p[1], p[2], p[3]);
  }
}

template <int MODE>   // 0: the network, 1: calibration (240 independent v_add_f32 per pass), 2: plain per-element convert (48 cvt_pk of column PAIRS is impossible: 96 v_cvt + nothing, the lower bound of a no-transpose drain)
__global__ void __launch_bounds__(512) transpose_time(float* __restrict__ sink, int reps, float seed) {
  const int lane = threadIdx.x & 63;
  float acc[96];
#pragma unroll
  for (int i = 0; i < 96; ++i) acc[i] = seed * (float)(i + 1) + (float)lane;
  const u64 m0 = lane_bit_mask(0), m1 = lane_bit_mask(1), m2 = lane_bit_mask(2);
  unsigned keep = 0;
  for (int it = 0; it < reps; ++it) {
#pragma unroll
    for (int i = 0; i < 96; ++i) asm volatile("" : "+v"(acc[i]));
    if (MODE == 0) {
#pragma unroll
      for (int pr = 0; pr < 12; ++pr) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = acc[8 * pr + k];
        unsigned p[4];
        pair_network(x, p, m0, ~m0, m1, ~m1, m2, ~m2);
        asm volatile("" :: "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));
        keep ^= p[0];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 240; ++k) {
        float d;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(acc[k % 96]), "v"(acc[(k + 7) % 96]));
        asm volatile("" :: "v"(d));
      }
    }
  }
  if (keep == 0x12345u) sink[threadIdx.x] = acc[0];
}

// rate table: 240 independent instructions of one kind per pass (the same harness as above)
#define RATE_KERNEL(NAME, TEXT)                                                                            \
  __global__ void __launch_bounds__(512) NAME(float* __restrict__ sink, int reps, float seed) {            \
    const int lane = threadIdx.x & 63;                                                                      \
    float acc[96];                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 96; ++i) acc[i] = seed * (float)(i + 1) + (float)lane;          \
    for (int it = 0; it < reps; ++it) {                                                                     \
      _Pragma("unroll") for (int i = 0; i < 96; ++i) asm volatile("" : "+v"(acc[i]));                     \
      _Pragma("unroll") for (int k = 0; k < 240; ++k) {                                                    \
        float d;                                                                                            \
        asm volatile(TEXT : "=v"(d) : "v"(acc[k % 96]), "v"(acc[(k + 7) % 96]), "s"(0xF0F0F0F0F0F0F0F0ull)); \
        asm volatile("" :: "v"(d));                                                                         \
      }                                                                                                     \
    }                                                                                                       \
    if (seed == 0.12345f) sink[threadIdx.x] = acc[0];                                                       \
  }
RATE_KERNEL(rate_add, "v_add_f32 %0, %1, %2")
RATE_KERNEL(rate_cndmask, "v_cndmask_b32 %0, %1, %2, vcc")
RATE_KERNEL(rate_mov_dpp_quad, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
RATE_KERNEL(rate_cnd_dpp_quad, "v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
RATE_KERNEL(rate_cnd_dpp_shl4, "v_cndmask_b32_dpp %0, %1, %2, vcc row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
RATE_KERNEL(rate_add_dpp_quad, "v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
RATE_KERNEL(rate_cvt_pk, "v_cvt_pk_bf16_f32 %0, %1, %2")
RATE_KERNEL(rate_cnd_e64, "v_cndmask_b32_e64 %0, %1, %2, %3")
RATE_KERNEL(rate_bfi, "v_bfi_b32 %0, %1, %2, %1")
RATE_KERNEL(rate_perm, "v_perm_b32 %0, %1, %2, %1")

// the same without the s_nop the compiler puts between single-instruction asm statements: 30 blocks of 8 independent instructions per pass
#define RATE8_KERNEL(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                  \
  __global__ void __launch_bounds__(512) NAME(float* __restrict__ sink, int reps, float seed) {            \
    const int lane = threadIdx.x & 63;                                                                      \
    float acc[96];                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 96; ++i) acc[i] = seed * (float)(i + 1) + (float)lane;          \
    for (int it = 0; it < reps; ++it) {                                                                     \
      _Pragma("unroll") for (int i = 0; i < 96; ++i) asm volatile("" : "+v"(acc[i]));                     \
      _Pragma("unroll") for (int k = 0; k < 30; ++k) {                                                     \
        float d0, d1, d2, d3, d4, d5, d6, d7;                                                               \
        u64 m;                                                                                              \
        asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7                       \
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7), "=&s"(m) \
                     : "v"(acc[(3 * k) % 96]), "v"(acc[(3 * k + 7) % 96]), "s"(0xF0F0F0F0F0F0F0F0ull) : "vcc");        \
        asm volatile("" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7), "s"(m)); \
      }                                                                                                     \
    }                                                                                                       \
    if (seed == 0.12345f) sink[threadIdx.x] = acc[0];                                                       \
  }
RATE8_KERNEL(rate8_add, "v_add_f32 %0, %9, %10", "v_add_f32 %1, %9, %10", "v_add_f32 %2, %9, %10", "v_add_f32 %3, %9, %10", "v_add_f32 %4, %9, %10", "v_add_f32 %5, %9, %10",
             "v_add_f32 %6, %9, %10", "v_add_f32 %7, %9, %10\n s_mov_b64 %8, 0")
RATE8_KERNEL(rate8_add_chain, "v_add_f32 %0, %9, %10", "v_add_f32 %0, %0, %10", "v_add_f32 %0, %0, %10", "v_add_f32 %0, %0, %10", "v_add_f32 %0, %0, %10", "v_add_f32 %0, %0, %10",
             "v_add_f32 %0, %0, %10", "v_add_f32 %0, %0, %10\n v_mov_b32 %1, %0\n v_mov_b32 %2, %0\n v_mov_b32 %3, %0\n v_mov_b32 %4, %0\n v_mov_b32 %5, %0\n v_mov_b32 %6, %0\n v_mov_b32 %7, %0\n s_mov_b64 %8, 0")
RATE8_KERNEL(rate8_exp_chain, "v_exp_f32 %0, %9", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0",
             "v_exp_f32 %0, %0", "v_exp_f32 %0, %0\n v_mov_b32 %1, %0\n v_mov_b32 %2, %0\n v_mov_b32 %3, %0\n v_mov_b32 %4, %0\n v_mov_b32 %5, %0\n v_mov_b32 %6, %0\n v_mov_b32 %7, %0\n s_mov_b64 %8, 0")
RATE8_KERNEL(rate8_exp, "v_exp_f32 %0, %9", "v_exp_f32 %1, %9", "v_exp_f32 %2, %9", "v_exp_f32 %3, %9", "v_exp_f32 %4, %10", "v_exp_f32 %5, %10",
             "v_exp_f32 %6, %10", "v_exp_f32 %7, %10\n s_mov_b64 %8, 0")
RATE8_KERNEL(rate8_cnd_vcc, "s_mov_b64 vcc, %11\n v_cndmask_b32 %0, %9, %10, vcc", "v_cndmask_b32 %1, %9, %10, vcc", "v_cndmask_b32 %2, %9, %10, vcc", "v_cndmask_b32 %3, %9, %10, vcc",
             "v_cndmask_b32 %4, %9, %10, vcc", "v_cndmask_b32 %5, %9, %10, vcc", "v_cndmask_b32 %6, %9, %10, vcc", "v_cndmask_b32 %7, %9, %10, vcc\n s_mov_b64 %8, 0")
RATE8_KERNEL(rate8_cnd_e64, "v_cndmask_b32_e64 %0, %9, %10, %11", "v_cndmask_b32_e64 %1, %9, %10, %11", "v_cndmask_b32_e64 %2, %9, %10, %11", "v_cndmask_b32_e64 %3, %9, %10, %11",
             "v_cndmask_b32_e64 %4, %9, %10, %11", "v_cndmask_b32_e64 %5, %9, %10, %11", "v_cndmask_b32_e64 %6, %9, %10, %11", "v_cndmask_b32_e64 %7, %9, %10, %11\n s_mov_b64 %8, 0")
RATE8_KERNEL(rate8_cmp_cnd_vcc, "v_cmp_gt_f32 vcc, %9, %10", "s_nop 1\n v_cndmask_b32 %0, %9, %10, vcc", "v_cmp_lt_f32 vcc, %9, %10", "s_nop 1\n v_cndmask_b32 %1, %9, %10, vcc",
             "v_cmp_gt_f32 vcc, %10, %9", "s_nop 1\n v_cndmask_b32 %2, %9, %10, vcc", "v_cmp_lt_f32 vcc, %10, %9", "s_nop 1\n v_cndmask_b32 %3, %9, %10, vcc\n v_mov_b32 %4, 0\n v_mov_b32 %5, 0\n v_mov_b32 %6, 0\n v_mov_b32 %7, 0\n s_mov_b64 %8, 0")
RATE8_KERNEL(rate8_cmp_cnd_e64, "v_cmp_gt_f32 %8, %9, %10", "s_nop 1\n v_cndmask_b32_e64 %0, %9, %10, %8", "v_cmp_lt_f32 %8, %9, %10", "s_nop 1\n v_cndmask_b32_e64 %1, %9, %10, %8",
             "v_cmp_gt_f32 %8, %10, %9", "s_nop 1\n v_cndmask_b32_e64 %2, %9, %10, %8", "v_cmp_lt_f32 %8, %10, %9", "s_nop 1\n v_cndmask_b32_e64 %3, %9, %10, %8\n v_mov_b32 %4, 0\n v_mov_b32 %5, 0\n v_mov_b32 %6, 0\n v_mov_b32 %7, 0")

// packed f32: 240 instructions per pass on 64-bit aligned register pairs (independent destinations)
#define RATEPK_KERNEL(NAME, TEXT)                                                                          \
  __global__ void __launch_bounds__(512) NAME(float* __restrict__ sink, int reps, float seed) {            \
    const int lane = threadIdx.x & 63;                                                                      \
    double acc[48];                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 48; ++i) acc[i] = (double)(seed * (float)(i + 1) + (float)lane); \
    for (int it = 0; it < reps; ++it) {                                                                     \
      _Pragma("unroll") for (int i = 0; i < 48; ++i) asm volatile("" : "+v"(acc[i]));                     \
      _Pragma("unroll") for (int k = 0; k < 30; ++k) {                                                     \
        double d0, d1, d2, d3, d4, d5, d6, d7;                                                              \
        asm volatile(TEXT : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7) \
                     : "v"(acc[(3 * k) % 48]), "v"(acc[(3 * k + 7) % 48]));                                 \
        asm volatile("" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));        \
      }                                                                                                     \
    }                                                                                                       \
    if (seed == 0.12345f) sink[threadIdx.x] = (float)acc[0];                                                \
  }
#define PK8(OP) OP " %0, %8, %9\n" OP " %1, %8, %9\n" OP " %2, %8, %9\n" OP " %3, %8, %9\n" OP " %4, %8, %9\n" OP " %5, %8, %9\n" OP " %6, %8, %9\n" OP " %7, %8, %9"
RATEPK_KERNEL(ratepk_add, PK8("v_pk_add_f32"))
RATEPK_KERNEL(ratepk_mul, PK8("v_pk_mul_f32"))
RATEPK_KERNEL(ratepk_fma, "v_pk_fma_f32 %0, %8, %9, %8\n v_pk_fma_f32 %1, %8, %9, %8\n v_pk_fma_f32 %2, %8, %9, %8\n v_pk_fma_f32 %3, %8, %9, %8\n v_pk_fma_f32 %4, %8, %9, %8\n v_pk_fma_f32 %5, %8, %9, %8\n v_pk_fma_f32 %6, %8, %9, %8\n v_pk_fma_f32 %7, %8, %9, %8")
RATEPK_KERNEL(ratepk_addf64, PK8("v_add_f64"))
// dependent chains: every instruction reads the previous one's result
RATEPK_KERNEL(ratepk_add_chain, "v_pk_add_f32 %0, %8, %9\n v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %0, %0, %9\n v_pk_add_f32 %0, %0, %9\n v_pk_mov_b32 %1, %0, %0\n v_pk_mov_b32 %2, %0, %0\n v_pk_mov_b32 %3, %0, %0\n v_pk_mov_b32 %4, %0, %0\n v_pk_mov_b32 %5, %0, %0\n v_pk_mov_b32 %6, %0, %0\n v_pk_mov_b32 %7, %0, %0")
RATEPK_KERNEL(ratepk_addf64_chain, "v_add_f64 %0, %8, %9\n v_add_f64 %0, %0, %9\n v_add_f64 %0, %0, %9\n v_add_f64 %0, %0, %9\n v_add_f64 %0, %0, %9\n v_add_f64 %0, %0, %9\n v_add_f64 %0, %0, %9\n v_add_f64 %0, %0, %9\n v_pk_mov_b32 %1, %0, %0\n v_pk_mov_b32 %2, %0, %0\n v_pk_mov_b32 %3, %0, %0\n v_pk_mov_b32 %4, %0, %0\n v_pk_mov_b32 %5, %0, %0\n v_pk_mov_b32 %6, %0, %0\n v_pk_mov_b32 %7, %0, %0")

// does a vector instruction overlap with an MFMA in flight on the same SIMD?  One wave per SIMD, per pass 32 x { one 32x32x16 MFMA (8 passes = 32
// cycles, four independent accumulators in rotation), FILL }.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
#define MFMA_MIX_KERNEL(NAME, FILL)                                                                        \
  __global__ void __launch_bounds__(256) NAME(float* __restrict__ sink, int reps, float seed) {            \
    const int lane = threadIdx.x & 63;                                                                      \
    f32x16_t c0, c1, c2, c3;                                                                                \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) { c0[i] = seed; c1[i] = seed; c2[i] = seed; c3[i] = seed; } \
    f32x4_t a = {seed, seed, seed, seed}, b = {seed + (float)lane, seed, seed, seed};                       \
    double x = (double)seed, y = (double)lane;                                                              \
    float xf = seed, yf = (float)lane;                                                                      \
    for (int it = 0; it < reps; ++it) {                                                                     \
      _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                      \
        double d0, d1, d2, d3;                                                                              \
        float f0, f1, f2, f3;                                                                               \
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %12, %13, %0\n" FILL "v_mfma_f32_32x32x16_bf16 %1, %12, %13, %1\n" FILL   \
                     "v_mfma_f32_32x32x16_bf16 %2, %12, %13, %2\n" FILL "v_mfma_f32_32x32x16_bf16 %3, %12, %13, %3\n" FILL   \
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3) \
                     : "v"(a), "v"(b), "v"(x), "v"(y), "v"(xf), "v"(yf));                                   \
        asm volatile("" :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(f0), "v"(f1), "v"(f2), "v"(f3));        \
      }                                                                                                     \
    }                                                                                                       \
    if (seed == 0.12345f) sink[threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0];                               \
  }
MFMA_MIX_KERNEL(mix_none, "")
MFMA_MIX_KERNEL(mix_add4, "v_add_f32 %8, %16, %17\n v_add_f32 %9, %16, %17\n v_add_f32 %10, %16, %17\n v_add_f32 %11, %16, %17\n")
MFMA_MIX_KERNEL(mix_pk2, "v_pk_add_f32 %4, %14, %15\n v_pk_add_f32 %5, %14, %15\n")
MFMA_MIX_KERNEL(mix_pk1_add2, "v_pk_add_f32 %4, %14, %15\n v_add_f32 %8, %16, %17\n v_add_f32 %9, %16, %17\n")
MFMA_MIX_KERNEL(mix_exp2, "v_exp_f32 %8, %16\n v_exp_f32 %9, %17\n")
MFMA_MIX_KERNEL(mix_f64, "v_add_f64 %4, %14, %15\n")

static float bf16_to_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f_to_bf16_rne(float f) {
  unsigned u; memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

int main() {
  const int waves = 8, n = waves * 128 * 48;
  std::vector<float> h(n);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = (float)(rand() % 20001 - 10000) / 37.0f;
  float* d_src; __hip_bfloat16* d_out; float* d_sink;
  CHECK(hipMalloc(&d_src, n * 4)); CHECK(hipMalloc(&d_out, n * 2)); CHECK(hipMalloc(&d_sink, 4096));
  CHECK(hipMemcpy(d_src, h.data(), n * 4, hipMemcpyHostToDevice));
  CHECK(hipMemset(d_out, 0xFF, n * 2));
  hipLaunchKernelGGL(transpose_check, dim3(waves / 4), dim3(256), 0, 0, d_src, d_out);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned short> o(n);
  CHECK(hipMemcpy(o.data(), d_out, n * 2, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i)
    if (o[i] != f_to_bf16_rne(h[i])) { if (bad < 8) printf("mismatch at wave %d row %d col %d: got %g want %g\n", i / 6144, (i % 6144) / 48, i % 48, bf16_to_f(o[i]), h[i]); ++bad; }
  printf("correctness: %d of %d elements differ from the plain f32 -> bf16 (rne) conversion in row-major order\n", bad, n);

  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int reps = 2000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int threads = 256 * wps;          // 4 SIMDs x wps waves
    float ms[2];
    for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(transpose_time<0>, dim3(256), dim3(threads), 0, 0, d_sink, reps, 1.0f);
        else hipLaunchKernelGGL(transpose_time<1>, dim3(256), dim3(threads), 0, 0, d_sink, reps, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
      }
    }
    const double per_pass_net = ms[0] * 1e6 / reps, per_pass_cal = ms[1] * 1e6 / reps;   // ns per pass of one SIMD's wps waves
    printf("%d wave(s)/SIMD: network %.1f ns per wave-tile pass (all %d waves of a SIMD together), 240 x v_add_f32 %.1f ns -> the network costs %.2f x 240 = %.0f VALU issue slots per wave tile; "
           "at 4 cycles per slot = %.0f cycles\n", wps, per_pass_net, wps, per_pass_cal, per_pass_net / per_pass_cal, 240.0 * per_pass_net / per_pass_cal, 960.0 * per_pass_net / per_pass_cal);
  }
  typedef void (*kern_t)(float*, int, float);
  struct { const char* name; kern_t k; } table[] = {
      {"v_add_f32", rate_add}, {"v_cndmask_b32 (vcc)", rate_cndmask}, {"v_mov_b32_dpp quad_perm", rate_mov_dpp_quad},
      {"v_cndmask_b32_dpp quad_perm", rate_cnd_dpp_quad}, {"v_cndmask_b32_dpp row_shl:4", rate_cnd_dpp_shl4}, {"v_add_f32_dpp quad_perm", rate_add_dpp_quad},
      {"v_cvt_pk_bf16_f32", rate_cvt_pk}, {"v_cndmask_b32_e64 (sgpr pair)", rate_cnd_e64}, {"v_bfi_b32", rate_bfi}, {"v_perm_b32", rate_perm}};
  printf("rate table: ns per 240 instructions of one kind, one SIMD, 1 and 2 waves (the 2-wave figure is for BOTH waves' 480 instructions)\n");
  for (auto& e : table) {
    float ms[3] = {0, 0, 0};
    for (int wps = 1; wps <= 2; ++wps)
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(e.k, dim3(256), dim3(256 * wps), 0, 0, d_sink, reps, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[wps], e0, e1));
      }
    printf("  %-40s 1 wave %7.1f ns   2 waves %7.1f ns\n", e.name, ms[1] * 1e6 / reps, ms[2] * 1e6 / reps);
  }
  struct { const char* name; kern_t k; } table8[] = {
      {"8 x v_add_f32 per block", rate8_add}, {"8 x v_add_f32, each reading the previous result (+ 7 v_mov)", rate8_add_chain}, {"8 x v_exp_f32 independent", rate8_exp}, {"8 x v_exp_f32, each reading the previous result (+ 7 v_mov)", rate8_exp_chain},
      {"8 x v_cndmask_b32 (VOP2, vcc) per block", rate8_cnd_vcc}, {"8 x v_cndmask_b32_e64 (sgpr pair) per block", rate8_cnd_e64},
      {"4 x (v_cmp -> vcc, s_nop 1, v_cndmask VOP2) + 4 v_mov", rate8_cmp_cnd_vcc}, {"4 x (v_cmp -> sgpr pair, s_nop 1, v_cndmask_e64) + 4 v_mov", rate8_cmp_cnd_e64}};
  printf("blocks of 8 (no compiler s_nop inside a block): ns per pass of 30 blocks\n");
  for (auto& e : table8) {
    float ms[3] = {0, 0, 0};
    for (int wps = 1; wps <= 2; ++wps)
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(e.k, dim3(256), dim3(256 * wps), 0, 0, d_sink, reps, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[wps], e0, e1));
      }
    printf("  %-62s 1 wave %7.1f ns   2 waves %7.1f ns\n", e.name, ms[1] * 1e6 / reps, ms[2] * 1e6 / reps);
  }
  struct { const char* name; kern_t k; } tablepk[] = {
      {"8 x v_pk_add_f32 per block", ratepk_add}, {"8 x v_pk_mul_f32 per block", ratepk_mul}, {"8 x v_pk_fma_f32 per block", ratepk_fma},
      {"8 x v_add_f64 per block", ratepk_addf64},
      {"8 x v_pk_add_f32, each reading the previous result (+ 7 v_pk_mov)", ratepk_add_chain}, {"8 x v_add_f64, each reading the previous result (+ 7 v_pk_mov)", ratepk_addf64_chain}};
  printf("packed f32 (64-bit register pairs): ns per pass of 30 blocks of 8\n");
  for (auto& e : tablepk) {
    float ms[3] = {0, 0, 0};
    for (int wps = 1; wps <= 2; ++wps)
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(e.k, dim3(256), dim3(256 * wps), 0, 0, d_sink, reps, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[wps], e0, e1));
      }
    printf("  %-62s 1 wave %7.1f ns   2 waves %7.1f ns\n", e.name, ms[1] * 1e6 / reps, ms[2] * 1e6 / reps);
  }
  struct { const char* name; kern_t k; } tablemix[] = {
      {"MFMA alone", mix_none}, {"MFMA + 4 v_add_f32", mix_add4}, {"MFMA + 2 v_pk_add_f32", mix_pk2}, {"MFMA + 1 v_pk_add_f32 + 2 v_add_f32", mix_pk1_add2},
      {"MFMA + 2 v_exp_f32", mix_exp2}, {"MFMA + 1 v_add_f64", mix_f64}};
  printf("32x32x16 bf16 MFMA (32 cycles) with vector fillers behind each, one wave per SIMD: ns per 32 MFMAs (zero data: the clock is not power-limited)\n");
  for (auto& e : tablemix) {
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(256), dim3(256), 0, 0, d_sink, reps, 0.0f);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("  %-50s %7.1f ns\n", e.name, ms * 1e6 / reps);
  }
  return bad != 0;
}
