// DBA decoder tail (models/modules/DBA.py:36-52): per-channel scale, L2 normalisation over the PIXEL axis,
// sigmoid gate + residual, the two 64->1 heads; the orthogonality loss (DBA.py:25-29) in its exact Gram form
// (no [B,HW,HW] tensor: SURVEY.md 8a row A3); and the closed-form backward of all of it w.r.t. the
// decoupled features d (SURVEY.md section 7 "hard parts").
//
// d is addressed as a [B, ld_c, HW] f32 buffer of which channels c0..c0+127 belong to this decoder
// (branch 1 = c0..c0+63, branch 2 = c0+64..c0+127), so the student and the EMA teacher can share one
// 256-row projection.  All kernels stream d with pixel-contiguous (coalesced) accesses.
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {

constexpr int E = 64;
constexpr float NORM_EPS = 1e-12f;

// ---------------------------------------------------------------------------------- column norms
__global__ __launch_bounds__(256) void colnorm_kernel(const float* __restrict__ d, int ld_c, int c0, const float* __restrict__ emb,
                                                      float* __restrict__ norm, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x, b = blockIdx.y;
  const float e = emb[c];  // emb [2,64] flattened == channel index within this decoder
  const float* row = d + ((long)b * ld_c + c0 + c) * HW;
  float s = 0.f;
#pragma unroll 8
  for (int p = threadIdx.x; p < HW; p += 256) {
    const float u = row[p] * e;
    s = fmaf(u, u, s);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) norm[b * 128 + c] = fmaxf(sqrtf(s), NORM_EPS);
}

// ---------------------------------------------------------------------------------- gate + heads (+ sdiag)
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ d, int ld_c, int c0, const float* __restrict__ emb,
                                                        const float* __restrict__ norm, const float* __restrict__ head_w,
                                                        const float* __restrict__ head_b, float* __restrict__ fg,
                                                        float* __restrict__ bg, float* __restrict__ sdiag, int HW) {
  __shared__ float kf[128], hw[128];
  __shared__ float red[16];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < 128) {
    kf[tid] = emb[tid] / norm[b * 128 + tid];
    hw[tid] = head_w[tid];
  }
  __syncthreads();
  const int p = blockIdx.x * 256 + tid;
  float s2 = 0.f;
  if (p < HW) {
    const float* dp = d + ((long)b * ld_c + c0) * HW + p;
    float accf = head_b[0], accb = head_b[1], s = 0.f;
    for (int c0 = 0; c0 < E; c0 += 16) {                          // 32 loads out before the first use (hipcc's own schedule of the unrolled
      float v1[16], v2[16];                                       // loop waited after every pair: two loads in flight per thread)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        v1[j] = dp[(long)(c0 + j) * HW];
        v2[j] = dp[(long)(c0 + j + E) * HW];
      }
      __builtin_amdgcn_sched_barrier(0);                          // (the scheduler would sink the loads back next to their uses)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int c = c0 + j;
        const float d1 = v1[j], d2 = v2[j];
        const float f1 = d1 * kf[c], f2 = d2 * kf[c + E];
        accf = fmaf(hw[c], sigmoid_gate(f1 * d1) + d1, accf);
        accb = fmaf(hw[c + E], sigmoid_gate(f2 * d2) + d2, accb);
        s = fmaf(f1, f2, s);
      }
    }
    fg[(long)b * HW + p] = accf;
    if (bg) bg[(long)b * HW + p] = accb;
    s2 = s * s;
  }
  if (sdiag) {  // uniform branch
    s2 = block_sum(s2, red);
    if (tid == 0) atomicAdd(&sdiag[b], s2);
  }
}

// ---------------------------------------------------------------------------------- Gram matrices (f32 MFMA)
// grid (S pixel chunks, B); partial[b][s][branch][64][64] = sum over the chunk of f f^T.
constexpr int GCH = 512;   // pixels per workgroup
constexpr int GK = 64;     // pixels per K-tile: a wave instruction of the staging loads reads 256 contiguous bytes of ONE channel row
constexpr int GLD = 129;   // (round 3; was 16 pixels = four 64-byte pieces of four rows per instruction: 78 us for 76 MB.  The MFMAs consume
                           //  the same (k, k+1) pixel pairs in the same order, so the partial sums are bitwise what they were.)
__global__ __launch_bounds__(256, 2) void gram_partial_kernel(const float* __restrict__ d, int ld_c, int c0,
                                                           const float* __restrict__ emb, const float* __restrict__ norm,
                                                           float* __restrict__ partial, int HW, int S) {
  __shared__ float Fs[2][GK * GLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, ps = blockIdx.x * GCH, pe = min(ps + GCH, HW);
  const float* dbase = d + ((long)b * ld_c + c0) * HW;
  const int sk = tid & 63, srow = tid >> 6;  // staging: a wave = 64 consecutive pixels of one channel row, 4 rows per pass
  const int br = wave >> 1, rt = wave & 1;   // wave -> (branch, row tile); both column tiles
  float kf[32];                              // emb / norm of this thread's 32 rows
#pragma unroll
  for (int i = 0; i < 32; ++i) kf[i] = emb[srow + 4 * i] / norm[b * 128 + srow + 4 * i];
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

  float r[32];
  auto gload = [&](int k0) {                 // clamped address + select: a predicated load would be a branch with its own vmcnt(0) per pair
    const int k = k0 + sk;
    const float* src = dbase + (k < pe ? k : pe - 1);
    const float live = k < pe ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) r[i] = src[(long)(srow + 4 * i) * HW];
#pragma unroll
    for (int i = 0; i < 32; ++i) r[i] *= kf[i] * live;
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 32; ++i) Fs[buf][sk * GLD + srow + 4 * i] = r[i];
  };
  const int nt = (pe - ps + GK - 1) / GK;
  gload(ps);
  lstore(0);
  for (int t = 0; t < nt; ++t) {
    __syncthreads();
    const bool more = t + 1 < nt;
    if (more) gload(ps + (t + 1) * GK);
    const float* fs = Fs[t & 1];
#pragma unroll 8
    for (int kk = 0; kk < GK; kk += 2) {
      const int k = kk + (lane >> 5);
      const float a = fs[k * GLD + br * 64 + rt * 32 + (lane & 31)];
      const float b0 = fs[k * GLD + br * 64 + (lane & 31)];
      const float b1 = fs[k * GLD + br * 64 + 32 + (lane & 31)];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
    }
    if (more) lstore((t + 1) & 1);
  }
  float* out = partial + (((long)b * S + blockIdx.x) * 2 + br) * (E * E);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
      const int i = rt * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * (lane >> 5);
      const int j = ct * 32 + (lane & 31);
      out[i * E + j] = acc[ct][rg];
    }
}

// grid (B, FIN_Y): gram[b] = sum_s partial[b][s]; trace[b][y] = this workgroup's share of sum_ij G1_ij G2_ij (= tr(G1 G2), both symmetric)
constexpr int FIN_Y = E * E / 256;
__global__ __launch_bounds__(256) void gram_finalize_kernel(const float* __restrict__ partial, float* __restrict__ gram,
                                                            float* __restrict__ trace, int S) {
  __shared__ float red[16];
  const int b = blockIdx.x, i = blockIdx.y * 256 + threadIdx.x;
  float g1 = 0.f, g2 = 0.f;
  for (int s = 0; s < S; ++s) {
    const float* p = partial + (((long)b * S + s) * 2) * (E * E);
    g1 += p[i];
    g2 += p[E * E + i];
  }
  gram[((long)b * 2 + 0) * (E * E) + i] = g1;
  gram[((long)b * 2 + 1) * (E * E) + i] = g2;
  const float tr = block_sum(g1 * g2, red);
  if (threadIdx.x == 0) trace[b * FIN_Y + blockIdx.y] = tr;
}

__global__ void orth_loss_kernel(const float* __restrict__ trace, const float* __restrict__ sdiag, float* __restrict__ loss, int B,
                                 double inv_z) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = 0.0;
    for (int b = 0; b < B; ++b) {
      double tb = 0.0;
      for (int y = 0; y < FIN_Y; ++y) tb += (double)trace[b * FIN_Y + y];
      t += tb - (double)sdiag[b];
    }
    loss[0] = (float)(t * inv_z);
  }
}

// ---------------------------------------------------------------------------------- backward, pass A
// gfeat[b][c][p] = dL/df for both branches = orthogonality term + gate term:
//   gfeat1 = coef*(G2 f1 - s f2) + g_fg*w_fg*sig'(f1 d1)*d1 ,  gfeat2 = coef*(G1 f2 - s f1) + g_bg*w_bg*sig'(f2 d2)*d2 ,
//   s_p = f1_p . f2_p.  The two 64x64 mat-vecs per pixel are run as exact-f32 MFMA products G[64x64] x F[64 x 256 px]:
// A operand = Gram matrix from LDS (symmetric, so row-major IS the [k][row] image), B operand = normalised features read
// straight from global memory (lane = pixel: coalesced; each element is used by exactly one MFMA, so no LDS staging).
// Workgroup = 256 pixels of one image, wave = 64 pixels, both branches in one k loop (s_p falls out of the same loads).
__global__ __launch_bounds__(256, 2) void dba_bwd_a_kernel(const float* __restrict__ d, int ld_c, int c0, const float* __restrict__ emb,
                                                           const float* __restrict__ norm, const float* __restrict__ head_w,
                                                           const float* __restrict__ gram, const float* __restrict__ gfg,
                                                           const float* __restrict__ gbg, float coef /* 2*gextra/Z */,
                                                           float* __restrict__ gfeat, int HW) {
  __shared__ float G[2][E * E];
  __shared__ float kf[128], hw[128];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h5 = lane >> 5;
  for (int i = tid; i < 2 * E * E; i += 256) (&G[0][0])[i] = gram[(long)b * 2 * E * E + i];
  if (tid < 128) {
    kf[tid] = emb[tid] / norm[b * 128 + tid];
    hw[tid] = head_w[tid];
  }
  __syncthreads();
  const int pbase = blockIdx.x * 256 + wave * 64;
  int px[2];
  bool ok[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int p = pbase + ct * 32 + l31;
    ok[ct] = p < HW;
    px[ct] = ok[ct] ? p : HW - 1;
  }
  const float* dp = d + ((long)b * ld_c + c0) * HW;

  f32x16 a1[2][2], a2[2][2];   // [row tile][pixel tile]: G2 f1 and G1 f2
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a1[0][0][i] = a1[0][1][i] = a1[1][0][i] = a1[1][1][i] = 0.f;
    a2[0][0][i] = a2[0][1][i] = a2[1][0][i] = a2[1][1][i] = 0.f;
  }
  float spart[2] = {0.f, 0.f};
  for (int kk0 = 0; kk0 < E; kk0 += 16) {                          // the features of eight k-pairs (32 loads) out before the first MFMA
    float r1[8][2], r2[8][2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kk0 + 2 * j + h5;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        r1[j][ct] = dp[(long)k * HW + px[ct]];
        r2[j][ct] = dp[(long)(k + E) * HW + px[ct]];
      }
    }
    __builtin_amdgcn_sched_barrier(0);                            // (the scheduler would sink the loads back next to their uses)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = kk0 + 2 * j + h5;
      const float k1 = kf[k], k2 = kf[E + k];
      float f1[2], f2[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f1[ct] = r1[j][ct] * k1;
        f2[ct] = r2[j][ct] * k2;
        spart[ct] = fmaf(f1[ct], f2[ct], spart[ct]);
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float g2 = G[1][k * E + rt * 32 + l31];
        const float g1 = G[0][k * E + rt * 32 + l31];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          a1[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(g2, f1[ct], a1[rt][ct], 0, 0, 0);
          a2[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, f2[ct], a2[rt][ct], 0, 0, 0);
        }
      }
    }
  }
  float* gp = gfeat + (long)b * 128 * HW;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const float s = spart[ct] + __shfl_xor(spart[ct], 32, 64);
    const long gi = (long)b * HW + px[ct];
    const float g1 = gfg[gi], g2 = gbg[gi];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h5;
        const float d1 = dp[(long)c * HW + px[ct]], d2 = dp[(long)(c + E) * HW + px[ct]];
        const float f1 = d1 * kf[c], f2 = d2 * kf[c + E];
        const float s1 = sigmoid_gate(f1 * d1), s2 = sigmoid_gate(f2 * d2);
        if (ok[ct]) {
          gp[(long)c * HW + px[ct]] = coef * (a1[rt][ct][r] - s * f2) + g1 * hw[c] * s1 * (1.f - s1) * d1;
          gp[(long)(c + E) * HW + px[ct]] = coef * (a2[rt][ct][r] - s * f1) + g2 * hw[c + E] * s2 * (1.f - s2) * d2;
        }
      }
  }
}

// ---------------------------------------------------------------------------------- backward, pass B
// workgroup = one (image, channel) row: projection backward of f = u/max(||u||,eps) over the pixel axis,
// then gd; plus the per-channel parameter-gradient reductions (atomics into `small`).
__global__ __launch_bounds__(256) void dba_bwd_b_kernel(const float* __restrict__ d, int ld_c, int c0, const float* __restrict__ emb,
                                                        const float* __restrict__ norm, const float* __restrict__ head_w,
                                                        const float* __restrict__ gfg, const float* __restrict__ gbg,
                                                        const float* __restrict__ gfeat, float* __restrict__ gd,
                                                        float* __restrict__ g_head_w, float* __restrict__ g_head_b,
                                                        float* __restrict__ g_dec_bias, int HW) {
  __shared__ float red[16];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float e = emb[c], n = norm[b * 128 + c], k = e / n, w = head_w[c];
  const float* drow = d + ((long)b * ld_c + c0 + c) * HW;
  const float* grow = gfeat + ((long)b * 128 + c) * HW;
  const float* gup = (c < E ? gfg : gbg) + (long)b * HW;
  float* orow = gd + ((long)b * 128 + c) * HW;
  float r = 0.f;
#pragma unroll 8
  for (int p = tid; p < HW; p += 256) r = fmaf(drow[p] * k, grow[p], r);
  r = block_sum(r, red);
  const bool clamped = (n <= NORM_EPS);
  float sgd = 0.f, sga = 0.f, sg_up = 0.f;
  for (int p0 = tid; p0 < HW; p0 += 256 * 6) {                     // six pixels per thread and trip: 18 loads out before the first use
    float dv6[6], gf6[6], g6[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int p = p0 + 256 * j, pc = p < HW ? p : HW - 1;
      dv6[j] = drow[pc];
      gf6[j] = grow[pc];
      g6[j] = gup[pc];
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int p = p0 + 256 * j;
      if (p >= HW) break;
      const float dv = dv6[j], f = dv * k, gf = gf6[j], g = g6[j];
      const float sg = sigmoid_gate(f * dv);
      const float gu = clamped ? gf / NORM_EPS : (gf - f * r) / n;
      const float o = fmaf(gu, e, g * w * fmaf(sg * (1.f - sg), f, 1.f));
      orow[p] = o;
      sgd += o;
      sga = fmaf(g, sg + dv, sga);
      sg_up += g;
    }
  }
  sgd = block_sum(sgd, red);
  sga = block_sum(sga, red);
  if (tid == 0) {
    atomicAdd(&g_dec_bias[c], sgd);
    atomicAdd(&g_head_w[c], sga);
  }
  if (c == 0 || c == E) {             // uniform per block
    sg_up = block_sum(sg_up, red);
    if (tid == 0) atomicAdd(&g_head_b[c == E ? 1 : 0], sg_up);
  }
}

}  // namespace ucod

using namespace ucod;

extern "C" int ucod_dba_colnorm(const float* d, int ld_c, int c0, const float* emb, float* norm, int B, int HW, void* stream) {
  if (!d || !emb || !norm || B <= 0 || HW <= 0 || c0 < 0 || c0 + 128 > ld_c) return UCOD_EINVAL;
  UCOD_PROF(PROF_DBA_COLNORM, stream);
  hipLaunchKernelGGL(colnorm_kernel, dim3(128, B), dim3(256), 0, (hipStream_t)stream, d, ld_c, c0, emb, norm, HW);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_dba_heads_fwd(const float* d, int ld_c, int c0, const float* emb, const float* norm, const float* head_w,
                                  const float* head_b, float* fg, float* bg, float* sdiag, int B, int HW, void* stream) {
  if (!d || !emb || !norm || !head_w || !head_b || !fg || B <= 0 || HW <= 0 || c0 < 0 || c0 + 128 > ld_c) return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  UCOD_PROF(PROF_DBA_HEADS, s);
  if (sdiag && !ucod::accumulators_prezeroed()) {
    hipError_t e = hipMemsetAsync(sdiag, 0, sizeof(float) * B, s);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(heads_fwd_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, s, d, ld_c, c0, emb, norm, head_w, head_b, fg, bg, sdiag, HW);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" size_t ucod_orth_workspace_bytes(int B, int HW) {
  const size_t S = (size_t)cdiv(HW, GCH);
  return ((size_t)B * S * 2 * E * E + (size_t)B * FIN_Y) * sizeof(float);
}

extern "C" int ucod_orth_gram_fwd(const float* d, int ld_c, int c0, const float* emb, const float* norm, const float* sdiag,
                                  float* gram, float* loss, void* ws, int B, int HW, void* stream) {
  if (!d || !emb || !norm || !sdiag || !gram || !loss || !ws || B <= 0 || HW <= 0 || c0 < 0 || c0 + 128 > ld_c) return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  UCOD_PROF(PROF_ORTH, s);
  const int S = cdiv(HW, GCH);
  float* partial = (float*)ws;
  float* trace = partial + (size_t)B * S * 2 * E * E;
  hipLaunchKernelGGL(gram_partial_kernel, dim3(S, B), dim3(256), 0, s, d, ld_c, c0, emb, norm, partial, HW, S);
  hipLaunchKernelGGL(gram_finalize_kernel, dim3(B, FIN_Y), dim3(256), 0, s, partial, gram, trace, S);
  const double inv_z = 1.0 / ((double)B * (double)HW * (double)HW);
  hipLaunchKernelGGL(orth_loss_kernel, dim3(1), dim3(64), 0, s, trace, sdiag, loss, B, inv_z);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" size_t ucod_dba_bwd_workspace_bytes(int B, int HW) { return (size_t)B * 128 * HW * sizeof(float); }

extern "C" int ucod_dba_bwd(const float* d, int ld_c, int c0, const float* emb, const float* norm, const float* head_w,
                            const float* gram, const float* gfg, const float* gbg, float gextra, float* gd, float* g_head_w,
                            float* g_head_b, float* g_dec_bias, void* ws, int B, int HW, void* stream) {
  if (!d || !emb || !norm || !head_w || !gram || !gfg || !gbg || !gd || !g_head_w || !g_head_b || !g_dec_bias || !ws || B <= 0 ||
      HW <= 0 || c0 < 0 || c0 + 128 > ld_c)
    return UCOD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* gfeat = (float*)ws;
  UCOD_PROF(PROF_DBA_BWD, s);
  hipError_t e = hipSuccess;
  if (ucod::accumulators_prezeroed()) {                                  // (ucod_accumulators_prezeroed: the caller's one zero launch covered them)
  } else if (g_head_w == g_dec_bias + 128 && g_head_b == g_head_w + 128) {      // the flat gradient arena (bias | head_w | head_b): one fill
    e = hipMemsetAsync(g_dec_bias, 0, sizeof(float) * 258, s);
  } else {
    e = hipMemsetAsync(g_head_w, 0, sizeof(float) * 128, s);
    if (e == hipSuccess) e = hipMemsetAsync(g_head_b, 0, sizeof(float) * 2, s);
    if (e == hipSuccess) e = hipMemsetAsync(g_dec_bias, 0, sizeof(float) * 128, s);
  }
  if (e != hipSuccess) return (int)e;
  const float coef = (float)(2.0 * (double)gextra / ((double)B * (double)HW * (double)HW));
  hipLaunchKernelGGL(dba_bwd_a_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, s, d, ld_c, c0, emb, norm, head_w, gram, gfg, gbg, coef, gfeat, HW);
  hipLaunchKernelGGL(dba_bwd_b_kernel, dim3(128, B), dim3(256), 0, s, d, ld_c, c0, emb, norm, head_w, gfg, gbg, gfeat, gd, g_head_w, g_head_b, g_dec_bias, HW);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
