// bf16 MFMA GEMM with fused epilogues for the ViT backbone (SURVEY.md 8a rows B1,B4,B5,B7,B8; training epilogues: row B9).
//
//   C[m][n] = sum_k A[m][k] * B[n][k]        A:[M,K]  B:[N,K]  both row-major, K contiguous (bf16)
//
// which is exactly nn.Linear (y = x W^T) with A = activations, B = weight -- and, with the operands
// swapped (A = W_key, B = tokens), the last layer's key projection written straight into the
// [B,C,h,w] map the reference's hook produces (data/utils/feature_extractor.py:46-47,55-58).
//
// PRODUCT kernels (variant numbers of ucod_gemm_bf16 in brackets; 0 = auto picks among them by shape):
//   * gemm_bf16_kernel        [1,2,12]  128x128x64 (or 64x64x64) tile, 4 waves (2x2), 2 workgroups per CU: small shapes.
//   * gemm_bf16_big_kernel    [9,10]    256 x 256|192 x 64 tile, 8 waves (2x4), ONE workgroup per CU; LDS-DMA operands that stay in flight
//                                       across raw s_barriers, staggered wave groups, two barrier intervals per K-tile; leftover tiles of
//                                       one- and two-round launches as patches (gemm_bf16_tiles.h).
//   * gemm_bf16_mixed_kernel  [13,14]   the same loop with a few 288-row tiles among the 256-row ones so that launches of three or more
//                                       rounds are whole rounds; also the home of the fp16-residual and e4m3 epilogues.
// All share the XCD-aware tile order (blocks b and b+8 share an XCD/L2) and, for the hot epilogues, `big_epilogue`
// (gemm_bf16_epilogue.h).  The four-interval, un-staggered and persistent forms measured on the way (variants 3-8) are laboratory code:
// variants/gemm_bf16_lab.hip, built by `make variants`, never loaded by the product path.
#include "gemm_bf16_tiles.h"
#include "gemm_bf16_plan.h"

namespace ucod {

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

#ifndef UCOD_GEMM_EARLY_A
#define UCOD_GEMM_EARLY_A 1
#endif
template <int N>
__device__ __forceinline__ void wait_vmcnt_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// (ablation build only: -DUCOD_GEMM_NOBAR removes the K loop's barriers of the mixed-height kernel -- wrong results, timing only)
#ifdef UCOD_GEMM_NOBAR
#define UCOD_MIXED_BARRIER() do {} while (0)
#else
#define UCOD_MIXED_BARRIER() __builtin_amdgcn_s_barrier()
#endif
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
  // LayerNorm-folded epilogues: the column sums beside the bias, and this thread's row statistics (or partial sums) for the tile's LDS table --
  // all requested ahead of the operand DMAs (see fold_request)
  FoldCtx<NT> fold;
  FoldReq freq;
  char* const fold_tab = smem + 2 * MixCfg<NT, 1>::BUF;
  if constexpr (kFold<EPI>) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      int n = n0 + wn * 16 * NT + (lane & 15) + j * 16;
      n = n < a.N ? n : a.N - 1;
      fold.cc[j] = a.colsum[n];
    }
    fold_request(a, m0, 2 * Cfg::RG, wave, lane, freq);
  }
  stageA(0, 0);
  stageA(0, 1);
  stageB(0);
  // EARLY_A: the A tiles of K-tile 1 are requested here too (buffer 1 is idle until K-tile 1), not during K-tile 0.  At K = 768 a tile is
  // only 12 K-tiles long and every workgroup of a round starts at the same moment: requested one K-tile (~1.2 us) ahead, the first A
  // panels arrive late behind that synchronised burst and the loop stalls at K-tile 1; requested here they have the whole prologue wait
  // and K-tile 0 to land.  The counted wait below leaves all of K-tile 1's requests in flight.
  constexpr bool EARLY_A = UCOD_GEMM_EARLY_A != 0;
  if (nt > 1) {
    if constexpr (EARLY_A) { stageA(1, 0); stageA(1, 1); }
    stageB(1);
    if constexpr (EARLY_A) {
      if (XT == 1 && wave < 2) wait_vmcnt_n<Cfg::NB + 6>(); else wait_vmcnt_n<Cfg::NB + 4>();   // (waves 0 / 1 of a tall tile issue one more piece per A slot)
    } else {
      wait_vmcnt<Cfg::NB>();
    }
  } else {
    wait_vmcnt<0>();
  }
  finish_col_consts<EPI, NT>(a, cb, cs);
  float craw[NT];                                // the folded epilogues' column sums as loaded (the rank-one MFMA runs before the column scale)
  if constexpr (kFold<EPI>) {                   // table written here, read in the epilogue: every barrier of the K loop lies in between
    fold_finish(a, fold_tab, 2 * Cfg::RG, wave, lane, freq);
    fold.tab = reinterpret_cast<const float*>(fold_tab) + wm * Cfg::RG;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      craw[j] = fold.cc[j];
      fold.cc[j] *= cs[j];
      fold.cb[j] = cb[j] * cs[j];
      cb[j] = 0.f;                              // accumulators start at zero: the folded bias is added behind the per-row scaling
    }
  }
  f32x4 acc[NI][NT];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){cb[j], cb[j], cb[j], cb[j]};
  UCOD_MIXED_BARRIER();
  if (wm == 1) UCOD_MIXED_BARRIER();            // staggered wave groups (see gemm_bf16_big_kernel)

  for (int t = 0; t < nt; ++t) {
    const char* bufA = smem + (t & 1) * Cfg::BUF + wm * Cfg::SLOT;
    const char* bufB = smem + (t & 1) * Cfg::BUF + 2 * Cfg::SLOT;
    const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
    const bool stage_a = more1 && (!EARLY_A || t > 0);   // K-tile 1's A tiles were requested in the prologue
    hx8 fb[NT][2];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      if constexpr (NPH == 2) {
        if (ph == 0 && stage_a) { stageA(t + 1, 0); stageA(t + 1, 1); }
        if (ph == 1 && more2) stageB(t + 2);
      } else {
        if (ph == 0 && stage_a) stageA(t + 1, 0);
        if (ph == 1 && stage_a) stageA(t + 1, 1);
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
      UCOD_MIXED_BARRIER();
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
      UCOD_MIXED_BARRIER();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (wm == 0) UCOD_MIXED_BARRIER();
  if constexpr (kFold<EPI> && UCOD_FOLD_RANK1) fold_rank_one<NT, NI>(acc, fold.tab, craw, lane);
  big_epilogue<EPI, NT, NI, AUX>(a, acc, cs, smem + wave * (32 * EPI_PITCH(16 * NT)), m0 + wm * Cfg::RG, n0 + wn * 16 * NT, lane, &fold);
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
  __shared__ __attribute__((aligned(16))) char smem[2 * MixCfg<NT, 1>::BUF + (kFold<EPI> ? FOLD_TAB_BYTES : 0)];   // (+ the folded epilogues' row table)
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

// tile-order knobs of the large-tile launches (tuning(): read once per process)
static void apply_order_tuning(GemmArgs& a) {
  const GemmTuning& t = tuning();
  if (t.group_m >= 0) a.group_m = t.group_m > 0 ? t.group_m : a.tiles_m;
  if (t.col_fast >= 0) a.col_fast = t.col_fast;
}

// variant: 0 auto | 1 128^2 register staging | 2 128^2 LDS-DMA | 12 64^2 | 9 / 10 large tile 256 / 192 wide | 13 / 14 mixed-height 256 / 192 wide
template <int EPI>
static int launch(GemmArgs a, int variant, hipStream_t s) {
  constexpr bool kTrainEpi = (EPI == UCOD_EPI_GELU_BWD_BF16 || EPI == UCOD_EPI_BIAS_GELU_SAVE_BF16);
  constexpr bool kPatchEpi = !kTrainEpi && !kFold<EPI>;        // (leftover-as-patches: not for the LayerNorm-folded epilogues, whose row scalars live in the tile's LDS table)
  const bool auto_small = variant == 0;                       // only `auto` may pick the 64 x 64 tile by itself
  if (variant == 0) {
    variant = 2;
    // large tiles when either dimension is long enough to fill the chip with 256-row tiles (the key hook has M = channels = 768
    // but N = all tokens: 3 x 172 tiles)
    const bool big_enough = a.M >= 2048 || (a.M >= 512 && (long)a.M * a.N >= (1L << 24));
    if (big_enough && a.K >= 128 && (a.N & 3) == 0 && (!(kBiasLike<EPI> || kGeluLike<EPI>) || (a.N & 7) == 0)) {
      // two 32-MFMA barrier intervals per K-tile (variants 9/10) beat four 16-MFMA ones (5/6) by 2-4 % and the persistent
      // form (7/8) by 1-5 % on every backbone shape (tools/gemm_bench.py); the width with the shorter modelled makespan
      variant = (!kFold<EPI> && big_plan(a.M, a.N, a.K, 192, kPatchEpi).cost < big_plan(a.M, a.N, a.K, 256, kPatchEpi).cost) ? 10 : 9;
      // three or more rounds with a nearly empty last one (QKV 6.05, fc1 8.06): mixed-height tiles make it whole rounds (-7 % / -8 %,
      // tools/gemm_order_sweep.py); at one or two rounds the patches above already do that at the same cost
      const MixedPlan mp = mixed_plan(a.M, a.N, 256);
      if (kColFused<EPI> && mp.feasible && mp.rounds >= 3 && !tuning().no_mixed) variant = 13;
    }
  }
  if (kTrainEpi || ((EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_F32) && !a.bias)) {   // large-tile kernels only
    if (kTrainEpi && ((a.N & 7) != 0 || a.K < 128)) return UCOD_EINVAL;
    if (variant < 3) variant = (big_plan(a.M, a.N, a.K, 192, kPatchEpi).cost < big_plan(a.M, a.N, a.K, 256, kPatchEpi).cost) ? 10 : 9;
  }
  constexpr bool kBf16Out = (kBiasLike<EPI> || kGeluLike<EPI> || kTrainEpi);
  if ((variant >= 3 && variant <= 8) || variant == 11 || variant > 14 || variant < 0) return UCOD_EINVAL;   // 3-8: laboratory variants (variants/gemm_bf16_lab.hip)
  if constexpr (kFold<EPI>) {                                    // the LayerNorm-folded epilogues exist for 64-column waves only: the 256-wide forms
    if (variant == 10) variant = 9;
    if (variant == 14) variant = 13;
    if (a.part_in && variant < 9) variant = 9;                    // row partials are summed by the large-tile kernels' prologue only
  }
  if (variant >= 9 && variant <= 10 && ((a.N & 3) != 0 || (kBf16Out && (a.N & 7) != 0))) return UCOD_EINVAL;   // 16-byte row stores
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
        apply_order_tuning(a);
        // bf16 outputs of the two forward epilogues (qkv, MLP hidden) leave with the non-temporal policy: they are read once, by the
        // next kernel, and displace less of what the running launch re-reads (in the step: QKV 153.5 -> 145.8 us, fc1 223.7 -> 216.7)
        const int aux = tuning().st_aux >= 0 ? tuning().st_aux : 2;
        dim3 grid(a.tiles_m * a.tiles_n), block(512);
        if constexpr (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16) {   // store-policy experiment builds exist for the two bf16 forward epilogues
          if (aux == 2 && variant == 13) { hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 4, 2>), grid, block, 0, s, a); UCOD_CHECK_LAUNCH(); return UCOD_OK; }
          if (aux == 16 && variant == 13) { hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 4, 16>), grid, block, 0, s, a); UCOD_CHECK_LAUNCH(); return UCOD_OK; }
        }
        if constexpr (kFold<EPI>) {                               // (always 256 wide, non-temporal output stores like the unfolded forward epilogues)
          hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 4, 2>), grid, block, 0, s, a);
          UCOD_CHECK_LAUNCH();
          return UCOD_OK;
        } else {
          if (variant == 13) hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 4>), grid, block, 0, s, a);
          else hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 3>), grid, block, 0, s, a);
        }
        UCOD_CHECK_LAUNCH();
        return UCOD_OK;
      }
    }
  } else {
    if (variant == 13 || variant == 14) variant -= 4;
  }
  if constexpr (kRowMapped<EPI>) {                               // the 256-wide kernel drains these with 32-bit buffer offsets and 16-byte stores
    if (variant == 9) {
      const int tok = a.tok > 1 ? a.tok : 2;
      const unsigned long out_bytes = EPI == UCOD_EPI_KEY_NCHW_F32 ? (unsigned long)(a.N / tok) * a.M * (tok - 1) * 4ul
                                                                   : (unsigned long)(a.M / (tok - 1)) * tok * a.N * (EPI == UCOD_EPI_PATCH_TOKENS_H16 ? 2ul : 4ul);
      const bool whole = EPI == UCOD_EPI_KEY_NCHW_F32 ? (a.N % tok) == 0 : (a.M % (tok - 1)) == 0;   // whole images (the drains size the output from them)
      const bool ok = a.tok > 1 && whole && out_bytes < 0x7FFFFFF0ul && (unsigned long)a.tok * a.N * 4ul < 0x7FFFFFF0ul &&
                      (EPI != UCOD_EPI_PATCH_TOKENS_H16 || (a.N & 7) == 0);
      if (!ok) variant = 10;                                      // (192-wide tiles keep the chunk-by-chunk drain)
    }
  }
  if (variant == 9 || variant == 10) {
    const bool wide = variant == 9;
    a.tiles_m = cdiv(a.M, 256);
    a.tiles_n = cdiv(a.N, wide ? 256 : 192);
    // few column tiles (proj / fc2: N = 768): sweep one A panel's columns back to back (proj 75 -> 70 us, L2 fetch 263 -> 230 MB);
    // many (QKV 9, fc1 12): row-tile fastest in groups of 8 (fc1 is 3-6 % slower column-fastest: its weight matrix alone exceeds the L2)
    a.col_fast = a.tiles_n <= 4;
    apply_order_tuning(a);
    dim3 grid(a.tiles_m * a.tiles_n), block(512);
    const BigPlan pl = big_plan(a.M, a.N, a.K, wide ? 256 : 192, kPatchEpi);
    if (pl.patches) {                                          // leftover-as-patches: exactly rounds x n_cu workgroups
      a.main_tiles = pl.rounds * device_cus();
      a.patches_per_wg = cdiv((long)pl.left * pl.ppt, a.main_tiles);
      grid.x = a.main_tiles;
    }
    if constexpr (kFold<EPI>) {
      hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, 4, true, 2>), grid, block, 0, s, a);
    } else {
      if (wide) hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, 4, true, 2>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((gemm_bf16_big_kernel<EPI, 3, true, 2>), grid, block, 0, s, a);
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
template <int EPI = UCOD_EPI_BIAS_SCALE_RESID_H16>
static int launch_resid_h16(ucod::GemmArgs a, hipStream_t s) {
  using namespace ucod;
  if ((a.N & 7) != 0 || a.K < 128 || !a.bias || !a.scale || !a.resid) return UCOD_EINVAL;
  if constexpr (kStats<EPI>) {                                    // whole 64-column slots; the table's byte offsets fit the descriptors
    if ((a.N & 63) != 0 || !a.part_out || a.nslot != a.N / 64 || (long)a.M * a.nslot * 8 >= (1L << 31)) return UCOD_EINVAL;
  }
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
  apply_order_tuning(a);
  // store policy of the new stream: default (kept in L2: the next launch reads it) or, UCOD_RESID16_NT=1, non-temporal (measurement switch, round 5)
  static const bool nt = [] { const char* e = ucod::lab_env("UCOD_RESID16_NT"); return e && e[0] == '1'; }();
  if (nt) hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 4, 2>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((gemm_bf16_mixed_kernel<EPI, 4>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, s, a);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

// Patch embedding onto the f16 residual stream WITH row partials (UCOD_EPI_PATCH_TOKENS_H16_STATS): the 256-wide one-shot large-tile kernel and its
// offset-scheme drain only (no leftover patches: they would bypass the partial sums); anything it cannot take is refused, the caller then uses the
// plain epilogue and ucod_row_stats_h16.
static int launch_patch_h16_stats(ucod::GemmArgs a, hipStream_t s) {
  using namespace ucod;
  if (!a.bias || !a.pos || a.tok < 2 || !a.part_out || (a.N & 63) != 0 || a.nslot != a.N / 64 || a.K < 128 || a.M < 2048) return UCOD_EINVAL;
  const int np = a.tok - 1;
  const unsigned long out_bytes = (unsigned long)(a.M / np) * a.tok * a.N * 2ul;
  if ((a.M % np) != 0 || out_bytes >= 0x7FFFFFF0ul || (unsigned long)a.tok * a.N * 4ul >= 0x7FFFFFF0ul ||
      (unsigned long)(a.M / np) * a.tok * a.nslot * 8ul >= 0x7FFFFFF0ul) return UCOD_EINVAL;
  a.tiles_m = cdiv(a.M, 256);
  a.tiles_n = cdiv(a.N, 256);
  a.col_fast = a.tiles_n <= 4;
  apply_order_tuning(a);
  a.main_tiles = 0;
  a.patches_per_wg = 0;
  hipLaunchKernelGGL((gemm_bf16_big_kernel<UCOD_EPI_PATCH_TOKENS_H16_STATS, 4, true, 2>), dim3(a.tiles_m * a.tiles_n), dim3(512), 0, s, a);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

static int gemm_entry(int epilogue, const void* A, const void* B, void* out, int M, int N, int K, const float* bias,
                      const float* scale, const float* resid, const float* pos, int tokens_per_image, int variant,
                      void* stream, const void* aux, void* out2, const float* stats = nullptr, const float* colsum = nullptr,
                      const float* part_in = nullptr, float* part_out = nullptr, int nslot = 0, float eps = 0.f) {
  using namespace ucod;
  if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || (K % BK) != 0) return UCOD_EINVAL;
  GemmArgs a;
  a.aux = aux;
  a.out2 = out2;
  a.stats = stats;
  a.colsum = colsum;
  a.part_in = part_in;
  a.part_out = part_out;
  a.nslot = nslot;
  a.eps = eps;
  a.ovf = (epilogue == UCOD_EPI_BIAS_SCALE_RESID_H16 || epilogue == UCOD_EPI_PATCH_TOKENS_H16 || epilogue == UCOD_EPI_BIAS_SCALE_RESID_H16_STATS ||
           epilogue == UCOD_EPI_PATCH_TOKENS_H16_STATS || epilogue == UCOD_EPI_LNFOLD_BIAS_BF16 || epilogue == UCOD_EPI_LNFOLD_GELU_BF16)
              ? resid16_overflow_counter() : nullptr;           // (the folded consumers count rows outside the fold's range into the same word: fold_finish)
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
  UCOD_PROF(epilogue == UCOD_EPI_QKV_FP8 || epilogue == UCOD_EPI_LNFOLD_BIAS_BF16 ? 0 : (epilogue == UCOD_EPI_LNFOLD_GELU_BF16 || epilogue == UCOD_EPI_BIAS_GELU_SPLIT2) ? 1 : (epilogue == UCOD_EPI_BIAS_SCALE_RESID_H16 || epilogue == UCOD_EPI_BIAS_SCALE_RESID_H16_STATS) ? 2 : (epilogue == UCOD_EPI_PATCH_TOKENS_H16 || epilogue == UCOD_EPI_PATCH_TOKENS_H16_STATS) ? 3 : (epilogue >= 0 && epilogue <= 5 ? epilogue : (epilogue == UCOD_EPI_GELU_BWD_BF16 ? PROF_GEMM_EPI6 : PROF_GEMM_EPI7)), s);
  switch (epilogue) {
    case UCOD_EPI_BIAS_BF16:                                   // NULL bias (plain product) only in the large-tile kernels
      if (!bias && (variant == 1 || variant == 2 || K < 128 || (N & 3))) return UCOD_EINVAL;
      return launch<UCOD_EPI_BIAS_BF16>(a, variant, s);
    case UCOD_EPI_GELU_BWD_BF16: if (!aux) return UCOD_EINVAL; return launch<UCOD_EPI_GELU_BWD_BF16>(a, variant, s);
    case UCOD_EPI_BIAS_GELU_SAVE_BF16: if (!bias || !out2) return UCOD_EINVAL; return launch<UCOD_EPI_BIAS_GELU_SAVE_BF16>(a, variant, s);
    case UCOD_EPI_BIAS_GELU_BF16: if (!bias) return UCOD_EINVAL; return launch<UCOD_EPI_BIAS_GELU_BF16>(a, variant, s);
    case UCOD_EPI_BIAS_GELU_SPLIT2:                                 // rows of 3 N bf16: the drains address them with 31-bit byte offsets
      if (!bias || (N & 7) != 0 || (long)M * 3 * N * 2 >= (1L << 31) - 16) return UCOD_EINVAL;
      return launch<UCOD_EPI_BIAS_GELU_SPLIT2>(a, variant, s);
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
    case UCOD_EPI_LNFOLD_BIAS_BF16:
    case UCOD_EPI_LNFOLD_GELU_BF16:
      if (!bias || (!stats && !part_in) || !colsum || (N & 7) != 0) return UCOD_EINVAL;
      if (part_in && (nslot < 2 || nslot > 2 * FOLD_MAX_SLOT_PAIRS || (nslot & 1) || K < 128 || (long)M * nslot * 8 >= (1L << 32) - 16)) return UCOD_EINVAL;
      if (!part_in && (long)M * 8 >= (1L << 32) - 16) return UCOD_EINVAL;
      return epilogue == UCOD_EPI_LNFOLD_BIAS_BF16 ? launch<UCOD_EPI_LNFOLD_BIAS_BF16>(a, variant, s) : launch<UCOD_EPI_LNFOLD_GELU_BF16>(a, variant, s);
    case UCOD_EPI_BIAS_SCALE_RESID_H16_STATS:
      if (!bias || !scale || !resid || !a.ovf || !part_out) return UCOD_EINVAL;
      if (M >= 2048 && K >= 128 && (N & 7) == 0 && (long)M * K * 2 < (1L << 32) && (long)N * K * 2 < (1L << 32)) return launch_resid_h16<UCOD_EPI_BIAS_SCALE_RESID_H16_STATS>(a, s);
      return UCOD_EINVAL;                                          // (small passes: the caller takes the plain epilogue + ucod_row_stats_h16)
    case UCOD_EPI_PATCH_TOKENS_H16_STATS:
      if (!a.ovf) return UCOD_EINVAL;
      return launch_patch_h16_stats(a, s);
    case UCOD_EPI_BIAS_SCALE_RESID_H16:
      if (!bias || !scale || !resid || !a.ovf) return UCOD_EINVAL;
      // large passes: the mixed-height large-tile kernel; small ones (a batch-1 Look-Twice pass) the 128 x 128 / 64 x 64 kernel, so that the
      // stream type is a property of the engine and not of the batch size
      if (M >= 2048 && K >= 128 && (N & 7) == 0 && (long)M * K * 2 < (1L << 32) && (long)N * K * 2 < (1L << 32)) return launch_resid_h16(a, s);
      return launch<UCOD_EPI_BIAS_SCALE_RESID_H16>(a, variant == 0 || variant == 1 || variant == 2 || variant == 12 ? variant : 0, s);
    case UCOD_EPI_PATCH_TOKENS_H16:
      if (!bias || !pos || tokens_per_image < 2 || !a.ovf) return UCOD_EINVAL;
      return launch<UCOD_EPI_PATCH_TOKENS_H16>(a, variant, s);
    default: return UCOD_EINVAL;
  }
}

extern "C" int ucod_gemm_bf16(int epilogue, const void* A, const void* B, void* out, int M, int N, int K, const float* bias,
                              const float* scale, const float* resid, const float* pos, int tokens_per_image, int variant,
                              void* stream) {
  if (epilogue == UCOD_EPI_GELU_BWD_BF16 || epilogue == UCOD_EPI_BIAS_GELU_SAVE_BF16) return UCOD_EINVAL;   // need ucod_gemm_bf16_train
  if (epilogue == UCOD_EPI_LNFOLD_BIAS_BF16 || epilogue == UCOD_EPI_LNFOLD_GELU_BF16) return UCOD_EINVAL;   // need ucod_gemm_lnfold
  if (epilogue == UCOD_EPI_BIAS_SCALE_RESID_H16_STATS || epilogue == UCOD_EPI_PATCH_TOKENS_H16_STATS) return UCOD_EINVAL;   // need ucod_gemm_bf16_stats
  return gemm_entry(epilogue, A, B, out, M, N, K, bias, scale, resid, pos, tokens_per_image, variant, stream, nullptr, nullptr);
}

extern "C" int ucod_gemm_bf16_train(int epilogue, const void* A, const void* B, void* out, int M, int N, int K, const float* bias,
                                    const void* aux_bf16, void* out2_bf16, int variant, void* stream) {
  UCOD_BF16_ONLY();
  return gemm_entry(epilogue, A, B, out, M, N, K, bias, nullptr, nullptr, nullptr, 0, variant, stream, aux_bf16, out2_bf16);
}


// LayerNorm folded into the consumer GEMM: A = the fp16 residual stream itself (so this entry exists in the fp16-operand build only: an MFMA
// takes both operands in one type), B = fp16(gamma (.) W), out = 16-bit [M,N].  See kFold in gemm_bf16_epilogue.h.
extern "C" int ucod_gemm_lnfold(int epilogue, const void* x_f16, const void* w_folded, void* out, int M, int N, int K, const float* bias_folded,
                                const float* colsum, const float* stats, const float* row_partials, int nslot, float eps, const float* scale,
                                int variant, void* stream) {
#ifndef UCOD_HALF_F16
  return UCOD_EINVAL;
#else
  if (epilogue != UCOD_EPI_LNFOLD_BIAS_BF16 && epilogue != UCOD_EPI_LNFOLD_GELU_BF16) return UCOD_EINVAL;
  if (epilogue == UCOD_EPI_LNFOLD_GELU_BF16 && scale) return UCOD_EINVAL;
  return gemm_entry(epilogue, x_f16, w_folded, out, M, N, K, bias_folded, scale, nullptr, nullptr, 0, variant, stream, nullptr, nullptr, stats, colsum,
                    row_partials, nullptr, nslot, eps);
#endif
}

// The two residual-stream producers with row partials for the next LayerNorm-folded consumer (large passes only; UCOD_EINVAL otherwise, nothing launched).
extern "C" int ucod_gemm_bf16_stats(int epilogue, const void* A, const void* B, void* out, int M, int N, int K, const float* bias, const float* scale,
                                    const void* resid, const float* pos, int tokens_per_image, float* row_partials, int nslot, void* stream) {
  if (epilogue != UCOD_EPI_BIAS_SCALE_RESID_H16_STATS && epilogue != UCOD_EPI_PATCH_TOKENS_H16_STATS) return UCOD_EINVAL;
  if (!row_partials || nslot <= 0) return UCOD_EINVAL;
  return gemm_entry(epilogue, A, B, out, M, N, K, bias, scale, (const float*)resid, pos, tokens_per_image, 0, stream, nullptr, nullptr, nullptr, nullptr,
                    nullptr, row_partials, nslot, 0.f);
}

// Re-read the UCOD_GEMM_* tuning variables (gemm_bf16_plan.h): they are read once per process, not per launch.
extern "C" void ucod_gemm_reload_tuning(void) { ucod::tuning() = ucod::read_gemm_tuning(); }
