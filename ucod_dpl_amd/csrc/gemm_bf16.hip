// bf16 MFMA GEMM with fused epilogues for the ViT backbone (SURVEY.md 8a rows B1,B4,B5,B7,B8).
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        A:[M,K]  B:[N,K]  both row-major, K contiguous (bf16)
//
// which is exactly nn.Linear (y = x W^T) with A = activations, B = weight -- and, with the operands
// swapped (A = W_key, B = tokens), the last layer's key projection written straight into the
// [B,C,h,w] map the reference's hook produces (data/utils/feature_extractor.py:46-47,55-58).
//
// gfx950 design: 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  Operand tiles are staged HBM->LDS with 16-byte LDS-DMA
// (global_load_lds_dwordx4; the LDS image is lane-linear, so the bank swizzle is applied to the
// per-lane SOURCE address and again on the ds_read_b128 side), double buffered, one barrier per
// K-tile.  The workgroup->tile map is XCD-aware (bijective remap: blocks b and b+8 share an XCD/L2).
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

struct GemmArgs {
  const bf16_raw* A;
  const bf16_raw* B;
  void* out;
  const float* bias;
  const float* scale;
  const float* resid;
  const float* pos;
  int M, N, K;
  int tok;   // tokens per image incl. CLS (PATCH / KEY epilogues)
  int tiles_m, tiles_n;
};

// 16-byte chunk swizzle inside a 128-byte (64 x bf16) tile row: conflict-free ds_read_b128 for the
// 16x16x32 fragment pattern (rows l&15, chunk l>>4) under the 64-bank / 16-lane-group rule.
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <bool GLDS>
__device__ __forceinline__ void stage_tile(const bf16_raw* __restrict__ G, int rows_total, int row0, int K, int k0,
                                           char* lds_tile, int wave, int lane, u32x4 (&regs)[4]) {
  // 128 rows x 8 chunks; wave-instruction i covers rows (i*4+wave)*8 .. +7, lane -> (row l>>3, phys chunk l&7)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
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

__device__ __forceinline__ void write_tile(char* lds_tile, int wave, int lane, const u32x4 (&regs)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<u32x4*>(lds_tile + (i * 4 + wave) * 1024 + lane * 16) = regs[i];
  }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& a, int m, int n, float v) {
  if (m >= a.M || n >= a.N) return;
  if constexpr (EPI == UCOD_EPI_BIAS_BF16) {
    reinterpret_cast<bf16_raw*>(a.out)[(size_t)m * a.N + n] = f32_to_bf16(v + a.bias[n]);
  } else if constexpr (EPI == UCOD_EPI_BIAS_GELU_BF16) {
    reinterpret_cast<bf16_raw*>(a.out)[(size_t)m * a.N + n] = f32_to_bf16(gelu_erf(v + a.bias[n]));
  } else if constexpr (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32) {
    const size_t i = (size_t)m * a.N + n;
    reinterpret_cast<float*>(a.out)[i] = a.resid[i] + a.scale[n] * (v + a.bias[n]);
  } else if constexpr (EPI == UCOD_EPI_PATCH_TOKENS_F32) {
    // row m = b*(tok-1)+p  ->  token row b*tok + 1 + p ; + bias + position embedding of token 1+p
    const int np = a.tok - 1;
    const int b = m / np, p = m - b * np;
    reinterpret_cast<float*>(a.out)[((size_t)b * a.tok + 1 + p) * a.N + n] = v + a.bias[n] + a.pos[(size_t)(1 + p) * a.N + n];
  } else if constexpr (EPI == UCOD_EPI_KEY_NCHW_F32) {
    // m = channel, n = global token index; drop CLS, write [B, C, tok-1]
    const int b = n / a.tok, t = n - b * a.tok;
    if (t == 0) return;
    reinterpret_cast<float*>(a.out)[((size_t)b * a.M + m) * (a.tok - 1) + (t - 1)] = v + a.bias[m];
  } else if constexpr (EPI == UCOD_EPI_BIAS_F32) {
    reinterpret_cast<float*>(a.out)[(size_t)m * a.N + n] = v + a.bias[n];
  }
}

template <int EPI, bool GLDS>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const GemmArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [stage][A|B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware bijective remap of the 1-D grid, then tn fastest (neighbours share the A row panel)
  const int nwg = a.tiles_m * a.tiles_n;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
  const int tm = wg / a.tiles_n, tn = wg - tm * a.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = a.K / BK;
  u32x4 ra[4], rb[4];
  stage_tile<GLDS>(a.A, a.M, m0, a.K, 0, smem, wave, lane, ra);
  stage_tile<GLDS>(a.B, a.N, n0, a.K, 0, smem + TILE_BYTES, wave, lane, rb);
  if constexpr (!GLDS) {
    write_tile(smem, wave, lane, ra);
    write_tile(smem + TILE_BYTES, wave, lane, rb);
  }

  for (int t = 0; t < nt; ++t) {
    __syncthreads();  // tile t visible (the fence drains the LDS-DMA); everyone is done with the other stage
    char* curA = smem + (t & 1) * 2 * TILE_BYTES;
    char* curB = curA + TILE_BYTES;
    char* nxtA = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
    const bool more = (t + 1 < nt);
    if (more) {
      stage_tile<GLDS>(a.A, a.M, m0, a.K, (t + 1) * BK, nxtA, wave, lane, ra);
      stage_tile<GLDS>(a.B, a.N, n0, a.K, (t + 1) * BK, nxtA + TILE_BYTES, wave, lane, rb);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wr * 64 + i * 16 + (lane & 15);
        fa[i] = *reinterpret_cast<const bf16x8*>(curA + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wc * 64 + j * 16 + (lane & 15);
        fb[j] = *reinterpret_cast<const bf16x8*>(curB + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if constexpr (!GLDS) {
      if (more) {
        write_tile(nxtA, wave, lane, ra);
        write_tile(nxtA + TILE_BYTES, wave, lane, rb);
      }
    }
  }

  // C/D map of v_mfma_f32_16x16x32: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int m = m0 + wr * 64 + i * 16 + (lane >> 4) * 4 + rg;
        const int n = n0 + wc * 64 + j * 16 + (lane & 15);
        epilogue_store<EPI>(a, m, n, acc[i][j][rg]);
      }
}

template <int EPI>
static int launch(const GemmArgs& a, int variant, hipStream_t s) {
  dim3 grid(a.tiles_m * a.tiles_n), block(256);
  if (variant == 1)
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, false>), grid, block, 0, s, a);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<EPI, true>), grid, block, 0, s, a);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

}  // namespace ucod

extern "C" int ucod_gemm_bf16(int epilogue, const void* A, const void* B, void* out, int M, int N, int K, const float* bias,
                              const float* scale, const float* resid, const float* pos, int tokens_per_image, int variant,
                              void* stream) {
  using namespace ucod;
  if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0) return UCOD_EINVAL;
  GemmArgs a;
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
  hipStream_t s = (hipStream_t)stream;
  UCOD_PROF(epilogue >= 0 && epilogue <= 5 ? epilogue : 5, s);
  switch (epilogue) {
    case UCOD_EPI_BIAS_BF16: if (!bias) return UCOD_EINVAL; return launch<UCOD_EPI_BIAS_BF16>(a, variant, s);
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
    case UCOD_EPI_BIAS_F32: if (!bias) return UCOD_EINVAL; return launch<UCOD_EPI_BIAS_F32>(a, variant, s);
    default: return UCOD_EINVAL;
  }
}
