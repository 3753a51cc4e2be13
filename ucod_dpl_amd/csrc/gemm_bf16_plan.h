// Host-side launch planning of the bf16 GEMM: tuning knobs, the large-tile plan (whole rounds, leftover tiles as patches) and the
// mixed-height plan.  Shared by gemm_bf16.hip (product) and variants/gemm_bf16_lab.hip.
#pragma once
#include <cstdlib>
#include "common.h"

namespace ucod {

// Tuning knobs, read from the environment ONCE (first use, or again on ucod_gemm_reload_tuning(): tests and tools/gemm_order_sweep.py
// change them inside one process) -- never on the launch path.
//   UCOD_GEMM_NO_PATCH=1      leftover tiles as a partly filled round instead of patches (bitwise batch-position independence)
//   UCOD_GEMM_PATCH_ROUNDS=n  patches only for launches of at most n whole rounds (default 2)
//   UCOD_GEMM_NO_MIXED=1      no mixed-height launches
//   UCOD_GEMM_GROUP_M / UCOD_GEMM_COL_FAST / UCOD_GEMM_ST_AUX   tile order and output store policy (tools/gemm_order_sweep.py): measurement knobs,
//                             honoured by -DUCOD_LAB_KNOBS builds only (common.h: lab_env)
struct GemmTuning {
  bool no_patch = false, no_mixed = false;
  int patch_rounds = 2, group_m = -1, col_fast = -1, st_aux = -1;      // -1: the launch's own default
};
inline GemmTuning read_gemm_tuning() {
  GemmTuning t;
  auto flag = [](const char* n) { const char* e = getenv(n); return e && e[0] && e[0] != '0'; };
  auto num = [](const char* n, int dflt) { const char* e = getenv(n); return e && e[0] ? atoi(e) : dflt; };
  auto lab_num = [](const char* n, int dflt) { const char* e = lab_env(n); return e && e[0] ? atoi(e) : dflt; };       // measurement knobs: -DUCOD_LAB_KNOBS builds only
  t.no_patch = flag("UCOD_GEMM_NO_PATCH");
  t.no_mixed = flag("UCOD_GEMM_NO_MIXED");
  t.patch_rounds = num("UCOD_GEMM_PATCH_ROUNDS", 2);
  t.group_m = lab_num("UCOD_GEMM_GROUP_M", -1);
  t.col_fast = lab_num("UCOD_GEMM_COL_FAST", -1);
  t.st_aux = lab_num("UCOD_GEMM_ST_AUX", -1);
  return t;
}
inline GemmTuning& tuning() {
  static GemmTuning t = read_gemm_tuning();
  return t;
}

// Large-tile launch plan for a tile width: whole rounds of n_cu tiles, and whether the tiles past the last whole round are few
// enough to be computed as patches on the side (patch_phase) instead of as a nearly empty extra round.
struct BigPlan {
  int total, rounds, left, ppt;
  bool patches;
  double cost;        // makespan model, fitted to tools/gemm_bench.py on MI355X: a tile costs a fixed part (A-panel DMA, prologue,
};                    // epilogue set-up) plus a part proportional to its width; a patch ~2 % of a tile per round
inline int device_cus() {
  static const int n_cu = [] { hipDeviceProp_t p; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&p, d) == hipSuccess ? p.multiProcessorCount : 256; }();
  return n_cu;
}
inline BigPlan big_plan(int M, int N, int K, int bn, bool patch_epi) {
  // tuning().no_patch (UCOD_GEMM_NO_PATCH=1): every output through the tile path, whose f32 sum over K has one fixed order -- results
  // are then bitwise independent of where a row sits in the batch; a patch sums K in 8 interleaved partials
  const bool off = tuning().no_patch;
  const int max_rounds = tuning().patch_rounds;         // 3-4 rounds measured: no gain alone (ViT-L QKV), -2 % in the two-stream step (the other stream fills those tails)
  const int n_cu = device_cus();
  BigPlan p;
  p.total = cdiv(M, 256) * cdiv(N, bn);
  p.rounds = p.total / n_cu;
  p.left = p.total - p.rounds * n_cu;
  p.ppt = 16 * (bn / 32);
  // (more rounds dilute the tail below what a patch costs every workgroup -- QKV / fc1 of ViT-B: 6 and 8 rounds -- the cost model decides)
  p.patches = patch_epi && !off && p.rounds >= 1 && p.rounds <= max_rounds && p.left > 0 && (long)p.left * p.ppt <= 2L * p.rounds * n_cu && (K & 31) == 0;
  // Makespan in tile units.  A last, partly filled round is cheaper than a full one (its tiles run on an otherwise idle chip: measured
  // 0.42 of a round at 1.6 % fill, fc2 2 rounds 183 us -> 2.016 rounds 221 us): 0.4 + 0.6 * fill.  A patch costs every workgroup ~8 % of
  // its tile (3.4 us of 41 at K = 768, 7.4 of 91 at K = 3072).  With these two numbers the model reproduces the measured choices: patches
  // for ViT-B proj / fc2 (553 vs 617 units = the measured 198 vs 221 us), plain 256-wide tiles for ViT-L's N = 1024 (1.34 rounds).
  const double tile = 0.45 * 256 + 0.55 * bn;
  const double fill = (double)p.left / n_cu;
  const double plain = (p.rounds + (p.left ? 0.4 + 0.6 * fill : 0.0)) * tile;
  const double patched = p.rounds * tile * 1.08;
  if (p.patches && patched >= plain) p.patches = false;
  p.cost = p.patches ? patched : plain;
  return p;
}

// Mixed-height plan (see gemm_bf16_mixed_kernel): row-tiles, how many of them tall, and their spacing; feasible = false when the shape
// already fills whole rounds or when 32 extra rows on every row-tile would not be enough.
struct MixedPlan { bool feasible; int tiles_m, n_tall, stride, rounds; };
inline MixedPlan mixed_plan(int M, int N, int bn) {
  MixedPlan p{false, 0, 0, 1, 0};
  const int n_cu = device_cus(), tiles_n = cdiv(N, bn), t0 = cdiv(M, 256) * tiles_n;
  const int rounds = t0 / n_cu;
  if (rounds < 1 || t0 == rounds * n_cu) return p;
  const int tm = (rounds * n_cu) / tiles_n;                     // row-tiles that fit `rounds` whole rounds
  const long short_rows = (long)M - 256L * tm;
  if (tm < 1 || short_rows <= 0) return p;
  const int n_tall = (int)cdiv(short_rows, 32L);
  if (n_tall > tm) return p;
  p.feasible = true;
  p.tiles_m = tm;
  p.n_tall = n_tall;
  p.stride = tm / n_tall;
  p.rounds = rounds;
  return p;
}

}  // namespace ucod
