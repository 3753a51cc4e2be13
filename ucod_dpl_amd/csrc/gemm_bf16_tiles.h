// The large-tile kernel of the bf16 MFMA GEMM (256 x 256|192 x 64, 8 waves, one workgroup per CU) and its leftover-as-patches
// mode.  A header because the product (gemm_bf16.hip: two barrier intervals per K-tile, staggered wave groups) and the laboratory
// (variants/gemm_bf16_lab.hip: four intervals, no stagger, persistent form) instantiate the same template.
#pragma once
#include "gemm_bf16_epilogue.h"

namespace ucod {

// ---- leftover tiles as patches ------------------------------------------------------------------------------------
// One large-tile workgroup fills a CU, so a launch runs in rounds of n_cu tiles and the backbone's shapes all land just past
// a whole number of rounds (32 x 1370 rows: 516 = 2 x 256 + 4 tiles for proj / fc2): the last 4 tiles ran alone on 4 CUs
// while 252 idled -- 18 % of fc2, 11 % of proj (tools/gemm_tail_probe.py: M = 43520 vs 43840).  In this mode the launch has
// exactly rounds x n_cu workgroups, and the outputs of the remaining L tiles are cut into 16 x 32 patches that the
// workgroups compute on the side, one or two each, BEFORE their own tile: the patch's operand loads are in flight together
// with the tile's first K-tile DMAs (a latency every workgroup pays anyway), the K range is dealt round-robin to the 8 waves
// (v_mfma_f32_16x16x32_bf16 straight from global registers), partial sums meet in the LDS slot the main loop touches last.
// Deterministic: a patch is summed by one workgroup in a fixed order.  Result bits differ from the tile path only by the
// order of the f32 adds over K.
template <int C> struct PatchC { static constexpr int value = C; };

template <int EPI>
constexpr bool kPatchPrefetch = (EPI == UCOD_EPI_BIAS_BF16 || EPI == UCOD_EPI_BIAS_GELU_BF16 || EPI == UCOD_EPI_BIAS_SCALE_RESID_F32 ||
                                 EPI == UCOD_EPI_BIAS_F32);

template <int EPI, int BN_>
__device__ __forceinline__ void patch_phase(const GemmArgs& a, char* scratch /* 16 KiB */, int orig, int wave, int lane) {
  constexpr int PC = BN_ / 32, PPT = 16 * PC;                   // patches per leftover tile
  const int total = a.tiles_m * a.tiles_n;
  const int npatch = (total - a.main_tiles) * PPT;
  const int K = a.K, steps = K >> 5;
  const int l15 = lane & 15, q = lane >> 4;
  for (int pi = 0; pi < a.patches_per_wg; ++pi) {
    const int p = orig * a.patches_per_wg + pi;
    if (p >= npatch) break;
    const int wg = a.main_tiles + p / PPT, rem = p % PPT;
    int ptm, ptn;
    tile_of(a, wg, ptm, ptn);
    const int r0 = ptm * 256 + (rem / PC) * 16;
    const int c0 = ptn * BN_ + (rem % PC) * 32;
    if (r0 >= a.M || c0 >= a.N) continue;                       // ragged last row / column tile: nothing there
    // this thread's output of the patch (one of 16 x 32) and its epilogue operands, requested before the operand loads so that
    // nothing is left to fetch once the partial sums meet
    const int idx = wave * 64 + lane, om = r0 + (idx >> 5), on = c0 + (idx & 31);
    const bool live = om < a.M && on < a.N;
    const int cm = om < a.M ? om : a.M - 1, cn = on < a.N ? on : a.N - 1;
    float e_bias = 0.f, e_scale = 1.f, e_resid = 0.f;
    if constexpr (kPatchPrefetch<EPI>) {
      if (a.bias) e_bias = a.bias[cn];
      if constexpr (EPI == UCOD_EPI_BIAS_BF16) { if (a.scale) e_scale = a.scale[cn]; }
      if constexpr (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32) {
        e_scale = a.scale[cn];
        e_resid = a.resid[(size_t)cm * a.N + cn];
      }
    }
    int ar = r0 + l15, br0 = c0 + l15, br1 = c0 + 16 + l15;
    ar = ar < a.M ? ar : a.M - 1;
    br0 = br0 < a.N ? br0 : a.N - 1;
    br1 = br1 < a.N ? br1 : a.N - 1;
    const bf16_raw* pa = a.A + (size_t)ar * K + q * 8;
    const bf16_raw* pb0 = a.B + (size_t)br0 * K + q * 8;
    const bf16_raw* pb1 = a.B + (size_t)br1 * K + q * 8;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // this wave's k-steps: wave, wave + 8, ...; loaded in the largest chunks that fit (every load is a real one: the patch is
    // bound by the 64 B/clk/CU of the vector-memory path, 48 rows x K x 2 bytes per patch)
    auto chunk = [&](int s0, auto cnt) {
      constexpr int C = decltype(cnt)::value;
      hx8 fa[C], f0[C], f1[C];
#pragma unroll
      for (int i = 0; i < C; ++i) {
        const int st = s0 + 8 * i;
        fa[i] = *reinterpret_cast<const hx8*>(pa + st * 32);
        f0[i] = *reinterpret_cast<const hx8*>(pb0 + st * 32);
        f1[i] = *reinterpret_cast<const hx8*>(pb1 + st * 32);
      }
#pragma unroll
      for (int i = 0; i < C; ++i) {
        acc0 = UCOD_MFMA16(fa[i], f0[i], acc0);
        acc1 = UCOD_MFMA16(fa[i], f1[i], acc1);
      }
    };
    {
      int s0 = wave, left = (steps - wave + 7) >> 3;            // wave-uniform
      for (; left >= 12; left -= 12, s0 += 96) chunk(s0, PatchC<12>{});
      if (left >= 6) { chunk(s0, PatchC<6>{}); left -= 6; s0 += 48; }
      if (left >= 3) { chunk(s0, PatchC<3>{}); left -= 3; s0 += 24; }
      for (; left > 0; --left, s0 += 8) chunk(s0, PatchC<1>{});
    }
    // partial sums [wave][16 rows][32 cols]; C layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    float* sc = reinterpret_cast<float*>(scratch) + wave * 512;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      sc[(4 * q + rg) * 32 + l15] = acc0[rg];
      sc[(4 * q + rg) * 32 + 16 + l15] = acc1[rg];
    }
    __syncthreads();
    {
      const float* rd = reinterpret_cast<const float*>(scratch) + idx;
      float v = rd[0];
#pragma unroll
      for (int w = 1; w < 8; ++w) v += rd[w * 512];
      if constexpr (kPatchPrefetch<EPI>) {
        if (live) {
          const size_t o = (size_t)om * a.N + on;
          if constexpr (EPI == UCOD_EPI_BIAS_BF16) reinterpret_cast<bf16_raw*>(a.out)[o] = f32_to_h((v + e_bias) * e_scale);
          else if constexpr (EPI == UCOD_EPI_BIAS_GELU_BF16) reinterpret_cast<bf16_raw*>(a.out)[o] = f32_to_h(gelu_erf(v + e_bias));
          else if constexpr (EPI == UCOD_EPI_BIAS_SCALE_RESID_F32) reinterpret_cast<float*>(a.out)[o] = e_resid + e_scale * (v + e_bias);
          else reinterpret_cast<float*>(a.out)[o] = v + e_bias;
        }
      } else {
        epilogue_store<EPI>(a, om, on, v);
      }
    }
    if (pi + 1 < a.patches_per_wg) __syncthreads();            // scratch is reused by the next patch
  }
}

// =====================================================================================================
// Large-tile kernel: 256 x (64*NT) x 64 block tile, 8 waves (2 in M x 4 in N), one workgroup per CU.
//   * per wave 128 x 16*NT outputs; a K-tile is consumed in FOUR phases of 32 rows each (2 x NT tiles x 2 k-steps
//     = 4*NT MFMAs per phase); the wave's B fragments are read once per K-tile (phase 1) and stay in registers;
//   * LDS = two K-tile buffers {A0 | A1 | B}; operands arrive by 16-byte LDS-DMA that stays IN FLIGHT across the
//     phase barriers: phase 1/2 stage A0/A1 of tile t+1 into the other buffer, phase 3/4 stage B of tile t+2 into
//     THIS buffer (its B slot is dead after phase 1), and the only wait is a counted `s_waitcnt vmcnt(BN/64)` at
//     phase 4 that leaves exactly the B(t+2) DMAs outstanding; raw s_barrier (a __syncthreads would drain vmcnt);
//   * 256-row tiles halve the L2->LDS traffic per FLOP of the 128x128 kernel, which is L2-bandwidth bound
//     (2 WGs/CU x 32 KB per 1024 MFMA cycles ~ 39 TB/s chip-wide, above the ~34.5 TB/s L2 ceiling).
// Hazards: RAW -- every wave waits for its own DMAs (vmcnt) BEFORE the phase-4 barrier, reads happen after it;
//          WAR -- B slot of buffer b: last ds_read in phase 1 (retired before its MFMAs), first restaged in phase 3;
//                 A slots of buffer b^1: last read in phase 4 of tile t-1, first restaged in phase 1 of tile t,
//                 with the phase-4 barrier in between.
// =====================================================================================================
constexpr int SLOT_A = 128 * 128;  // bytes: 128 rows x 64 bf16

template <int NT>
struct BigCfg {
  static constexpr int BN_ = 64 * NT;
  static constexpr int NB = BN_ / 64;                 // LDS-DMA instructions per thread for the B tile
  static constexpr int BUF = 2 * SLOT_A + BN_ * 128;  // bytes per K-tile buffer
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

template <int EPI, int NT, bool STAGGER, int NPH = 4>
__global__ __launch_bounds__(512) void gemm_bf16_big_kernel(const GemmArgs a) {
  // NPH phases of 128/NPH rows per K-tile and wave group.  NPH = 2 halves the number of barrier intervals per MFMA (two
  // 32-MFMA intervals instead of four 16-MFMA ones per K-tile and group) at the price of 16 more fragment registers.
  constexpr int IT = 8 / NPH;                                     // 16-row i-tiles per phase
  using Cfg = BigCfg<NT>;
  __shared__ __attribute__((aligned(16))) char smem[2 * Cfg::BUF + (kFold<EPI> ? FOLD_TAB_BYTES : 0)];   // (+ the LayerNorm-folded epilogues' row table)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  const int nwg = a.main_tiles > 0 ? a.main_tiles : a.tiles_m * a.tiles_n;   // leftover-as-patches mode: the first main_tiles tiles of the order
  const int orig = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
  // grouped order inside each XCD's chunk: GROUP_M row-tiles x all column-tiles, row-tile fastest -- the workgroups that are
  // resident together on an XCD then share a few B (weight) panels and GROUP_M A panels that fit its 4 MiB L2, instead of
  // every row-tile streaming the whole weight matrix through L2 (FETCH_SIZE was 5x the algorithmic bytes on fc1).
  int tm, tn;
  tile_of(a, wg, tm, tn);
  const int m0 = tm * 256, n0 = tn * Cfg::BN_;
  const int K = a.K, nt = K / BK;

  // per-thread LDS-DMA source rows (fixed for the whole K loop): A0,A1 -> 2 instructions each; B -> NB instructions
  const bf16_raw* srcA[2][2];
  const bf16_raw* srcB[Cfg::NB];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (i * 8 + wave) * 8 + (lane >> 3);
      int gr = m0 + h * 128 + r;
      gr = gr < a.M ? gr : a.M - 1;
      srcA[h][i] = a.A + (size_t)gr * K + swz(r, lane & 7) * 8;
    }
#pragma unroll
  for (int i = 0; i < Cfg::NB; ++i) {
    const int r = (i * 8 + wave) * 8 + (lane >> 3);
    int gr = n0 + r;
    gr = gr < a.N ? gr : a.N - 1;
    srcB[i] = a.B + (size_t)gr * K + swz(r, lane & 7) * 8;
  }
  auto dmaA = [&](const bf16_raw* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, UCOD_LD_AUX_A);
  };
  auto dmaB = [&](const bf16_raw* src, char* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, UCOD_LD_AUX_B);
  };
  auto stageA = [&](int t, int h) {
    char* slot = smem + (t & 1) * Cfg::BUF + h * SLOT_A;
#pragma unroll
    for (int i = 0; i < 2; ++i) dmaA(srcA[h][i] + t * BK, slot + (i * 8 + wave) * 1024);
  };
  auto stageB = [&](int t, int i0, int i1) {
    char* slot = smem + (t & 1) * Cfg::BUF + 2 * SLOT_A;
#pragma unroll
    for (int i = 0; i < Cfg::NB; ++i)
      if (i >= i0 && i < i1) dmaB(srcB[i] + t * BK, slot + (i * 8 + wave) * 1024);
  };
  constexpr int B_SPLIT = Cfg::NB >= 2 ? 2 : 1;   // phase 3 issues [0,B_SPLIT), phase 4 the rest

  // per-column epilogue constants first (oldest in the vmcnt queue: landed long before the accumulators are initialised)
  float cb[NT], cs[NT];
  load_col_consts<EPI, NT>(a, n0 + wn * 16 * NT + (lane & 15), cb, cs);
  constexpr bool FASTRM = kFastRowMapped<EPI, NT>;                 // key hook / patch embedding with 64-column waves: see big_epilogue
  FoldCtx<NT> fold;                                                // LayerNorm-folded epilogues: see gemm_bf16_epilogue.h
  FoldReq freq;
  char* const fold_tab = smem + 2 * Cfg::BUF;
  if constexpr (kFold<EPI>) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      int n = n0 + wn * 16 * NT + (lane & 15) + j * 16;
      n = n < a.N ? n : a.N - 1;
      fold.cc[j] = a.colsum[n];
    }
    fold_request(a, m0, 256, wave, lane, freq);
  }

  // prologue: tile 0 complete, B of tile 1 in flight
  stageA(0, 0);
  stageA(0, 1);
  stageB(0, 0, Cfg::NB);
  if (nt > 1) stageB(1, 0, Cfg::NB);
  if constexpr (EPI != UCOD_EPI_GELU_BWD_BF16 && EPI != UCOD_EPI_BIAS_GELU_SAVE_BF16) {
    // scratch: the A0 slot of buffer 1, first written by the DMAs of K-tile 1 after the barrier below.  vmcnt retires in order,
    // so the patch's stores (older than every later DMA) never disturb the counted waits of the main loop.
    if (a.patches_per_wg > 0) patch_phase<EPI, Cfg::BN_>(a, smem + Cfg::BUF, orig, wave, lane);
  }
  if (nt > 1) wait_vmcnt<Cfg::NB>(); else wait_vmcnt<0>();
  finish_col_consts<EPI, NT>(a, cb, cs);
  float craw[NT];
  if constexpr (kFold<EPI>) {
    fold_finish(a, fold_tab, 256, wave, lane, freq);
    fold.tab = reinterpret_cast<const float*>(fold_tab) + wm * 128;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      craw[j] = fold.cc[j];
      fold.cc[j] *= cs[j];
      fold.cb[j] = cb[j] * cs[j];
      cb[j] = 0.f;
    }
  }
  f32x4 acc[8][NT];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){cb[j], cb[j], cb[j], cb[j]};
  __builtin_amdgcn_s_barrier();
  // STAGGER: the wm==1 waves run one barrier interval behind the wm==0 waves, so on every SIMD one wave is in its
  // MFMA interval while its partner is in its LDS-read / DMA-issue interval (two barriers per phase: R | M).
  // All waves execute the same number of barriers (extra one here for wm==1, extra one after the loop for wm==0).
  if (STAGGER && wm == 1) __builtin_amdgcn_s_barrier();

  for (int t = 0; t < nt; ++t) {
    const char* bufA = smem + (t & 1) * Cfg::BUF + wm * SLOT_A;
    const char* bufB = smem + (t & 1) * Cfg::BUF + 2 * SLOT_A;
    const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
    hx8 fb[NT][2];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      if constexpr (NPH == 4) {
        if (ph == 0 && more1) stageA(t + 1, 0);
        if (ph == 1 && more1) stageA(t + 1, 1);
        if (ph == 2 && more2) stageB(t + 2, 0, B_SPLIT);
        if (ph == 3 && more2) stageB(t + 2, B_SPLIT, Cfg::NB);
      } else {
        if (ph == 0 && more1) { stageA(t + 1, 0); stageA(t + 1, 1); }
        if (ph == 1 && more2) stageB(t + 2, 0, Cfg::NB);
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
          const int r = ph * (128 / NPH) + i * 16 + (lane & 15);
          fa[i][ks] = *reinterpret_cast<const hx8*>(bufA + r * 128 + swz(r, ks * 4 + (lane >> 4)) * 16);
        }
      if constexpr (STAGGER) {
        // RAW: every wave retires its tile-(t+1) DMAs BEFORE the barrier that precedes the leading group's first read
        if (ph == NPH - 1) {
          if (more2) wait_vmcnt<Cfg::NB>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[ph * IT + i][j] = UCOD_MFMA16(fa[i][ks], fb[j][ks], acc[ph * IT + i][j]);
      __builtin_amdgcn_s_setprio(0);
      if constexpr (!STAGGER) {
        if (ph == NPH - 1) {
          if (more2) wait_vmcnt<Cfg::NB>(); else wait_vmcnt<0>();
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (STAGGER && wm == 0) __builtin_amdgcn_s_barrier();

  if constexpr (kFold<EPI> && UCOD_FOLD_RANK1) fold_rank_one<NT, 8>(acc, fold.tab, craw, lane);
  // epilogue through a wave-private LDS region (operand tiles are dead: last barrier passed)
  big_epilogue<EPI, NT, 8, UCOD_ST_AUX, FASTRM>(a, acc, cs, smem + wave * (32 * EPI_PITCH(16 * NT)), m0 + wm * 128, n0 + wn * 16 * NT, lane, &fold);
}

}  // namespace ucod
