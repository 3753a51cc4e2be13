// fp8 (OCP e4m3) attention path on the block-scaled CDNA4 matrix instruction -- BASELINE.json configs[4]
// ("DINOv2 ViT-B/14 fp8 (CDNA4 fp8 MFMA) attention path, 518x518, batch 64, 1 GPU throughput-only").
// Replaces, like attention.hip, the softmax(Q K^T / sqrt d) V of transformers' eager_attention_forward as the reference runs it
// (data/utils/feature_extractor.py:51-54 -> modeling_dinov2.py:171-196), with Q, K, V and P quantised to e4m3.
//
// Two kernels:
//   * qkv_to_fp8_kernel: the QKV GEMM's 16-bit output [B*N, 3D] (Q pre-scaled by hd^-1/2 * log2 e) -> per (image, head):
//       Q8, K8  [Npad][64] bytes, one 64-byte row per token (Npad = whole 64-key tiles, rows >= N zero);
//       Vt8     [tiles][64 d][64] bytes: V TRANSPOSED per 64-key tile, the 64 keys of a row stored in the order the P operand of the
//               second product holds them (below), so that both matrix operands are plain 32-byte row reads.
//     Each tensor is multiplied by a power of two before rounding (2^q_exp, 2^k_exp, 2^v_exp) to sit in e4m3's normal range
//     (2^-6 .. 448); the inverse goes into the matrix instruction's E8M0 block scales, i.e. costs nothing.  Values are clamped to
//     +-448 (e4m3fn has no infinity).
//   * attn_fwd_fp8_kernel: the v5 kernel's structure (attention.hip: 128 query rows per workgroup, 4 waves, 64-key tiles, swapped
//     Q K^T with the query on the lane, accumulator initialised with -m, deferred rescale, exp2) on
//     v_mfma_scale_f32_32x32x64_f8f6f4: ONE instruction per 32-key block for Q K^T (K = head_dim = 64) and ONE per 32-wide d half
//     for P V (K = the tile's 64 keys): 4 matrix instructions of 64 cycles per tile instead of 16 of 32.
//
// Fused form (ucod_attention_fwd_fp8_fused): the QKV GEMM's UCOD_EPI_QKV_FP8 epilogue writes Q8 / K8 / V8 itself, all three row-major
// [Npad][64]; the kernel (VROW = true) then fetches the V operand with ds_read_b64_tr_b8.  Measured semantics of that instruction on
// gfx950 (tools/probes/tr8_probe.hip): in each group of 16 lanes the EVEN lanes 2q supply the addresses of the 8 rows (8 bytes each)
// of one 8x8 byte block and the ODD lanes 2q+1 those of a second block; lane i < 8 of the group receives column i of the first
// block, lane 8 + i column i of the second -- rows in the order of the supplying lanes.  With lane = 16 G + u: rows = the 8 keys of
// slots 8 i .. 8 i + 7 of lane half G >> 1, columns = d 16 (G & 1) .. + 7 (even block) and + 8 .. + 15 (odd block) of the 32-wide d half.
//
// Operand slots.  For this instruction a lane holds 32 bytes of A (row = lane & 31) and 32 bytes of B (column = lane & 31); the
// two lanes l and l + 32 of a row / column hold the two halves of K.  Byte j of lane half h of A multiplies byte j of lane half h
// of B, so any assignment of k to (h, j) is valid as long as both operands use the same one:
//   Q K^T: k = d, slot (h, j) = d = 32 h + j for both K8 and Q8 rows (natural order).
//   P V  : k = key of the tile.  After the first product lane (q, h) holds, for block kt and register r, the score of key
//          32 kt + (r & 3) + 8 (r >> 2) + 4 h (the 32x32 C layout); packed in register order that is slot (h, j = 16 kt + r).
//          Vt8 stores key(h, j) at byte 32 h + j of its row.
#include "common.h"
#include "../../include/ucod_dpl.h"

namespace ucod {
namespace f8 {

constexpr int HD = 64, QT = 128, KT = 64;
constexpr int TILE_BYTES = KT * 64;              // one K8 tile (64 tokens x 64 B) = one Vt8 tile (64 d x 64 B) = 4 KiB
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
constexpr float E4M3_MAX = 448.f;
constexpr float DEFER_THR = 8.0f;                // as attention.hip: P <= 2^8 = 256 < 448

__device__ __forceinline__ int key_of_slot(int h, int j) { return 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h; }

__device__ __forceinline__ float clamp8(float v) { return fminf(fmaxf(v, -E4M3_MAX), E4M3_MAX); }
// four f32 -> four e4m3 bytes (byte i = x_i), round-to-nearest-even (v_cvt_pk_fp8_f32; OCP e4m3fn on gfx950)
__device__ __forceinline__ unsigned pack4_fp8(float x0, float x1, float x2, float x3) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, w, true);
  return (unsigned)w;
}

// One workgroup = one 64-token tile of one (image, head).  Thread -> (token, 16-wide d chunk) for Q and K; V goes through LDS and
// leaves as (d, 16-byte chunk of the permuted key order).
__global__ __launch_bounds__(256) void qkv_to_fp8_kernel(const h_raw* __restrict__ qkv, char* __restrict__ q8, char* __restrict__ k8,
                                                         char* __restrict__ vt8, int N, int heads, int nt, float sq, float sk, float sv) {
  __shared__ float vs[KT][HD + 1];
  const int t = blockIdx.x, pair = blockIdx.y, tid = threadIdx.x;
  const int b = pair / heads, head = pair - b * heads, D = heads * HD, ld = 3 * D;
  const int tok = tid >> 2, c = tid & 3;
  const int token = t * KT + tok;
  const size_t npad = (size_t)nt * KT;
  u32x4 w[3][2];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int i = 0; i < 2; ++i) w[m][i] = (u32x4){0u, 0u, 0u, 0u};
  if (token < N) {
    const h_raw* src = qkv + ((size_t)b * N + token) * ld + head * HD + c * 16;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int i = 0; i < 2; ++i) w[m][i] = *reinterpret_cast<const u32x4*>(src + m * D + i * 8);
  }
  const float sc[3] = {sq, sk, sv};
  u32x4 o8[2];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo, hi;
        unpack_h2(w[m][i][e], lo, hi);
        x[i * 8 + 2 * e] = clamp8(lo * sc[m]);
        x[i * 8 + 2 * e + 1] = clamp8(hi * sc[m]);
      }
    if (m < 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o8[m][e] = pack4_fp8(x[4 * e], x[4 * e + 1], x[4 * e + 2], x[4 * e + 3]);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) vs[tok][c * 16 + e] = x[e];
    }
  }
  const size_t row = ((size_t)pair * npad + token) * 64 + c * 16;
  *reinterpret_cast<u32x4*>(q8 + row) = o8[0];
  *reinterpret_cast<u32x4*>(k8 + row) = o8[1];
  __syncthreads();
  {
    const int d = tid >> 2, h = c >> 1, kt = c & 1;       // 16-byte chunk c of row d holds slots (h, 16 kt .. 16 kt + 15)
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = vs[key_of_slot(h, 16 * kt + 4 * e + u)][d];
      o[e] = pack4_fp8(x[0], x[1], x[2], x[3]);
    }
    *reinterpret_cast<u32x4*>(vt8 + (((size_t)pair * nt + t) * HD + d) * 64 + c * 16) = o;
  }
}

// 16-byte chunk swizzle inside a 64-byte row: four rows cover the 256-byte bank row, so the chunk is XORed with (row >> 2) & 3 --
// conflict-free for ds_read_b128 with lane -> row (16-lane groups {0-3,12-15,20-27}: (row & 3, (row >> 2) & 3) all distinct).
__device__ __forceinline__ int swz64(int row, int chunk) { return chunk ^ ((row >> 2) & 3); }

template <int V> struct IntC { static constexpr int value = V; };

__device__ __forceinline__ float xhalf_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <bool VROW>
__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(const char* __restrict__ q8, const char* __restrict__ k8, const char* __restrict__ vt8,
                                                               h_raw* __restrict__ out, int N, int heads, int npairs, int nt, int qk_scale_bytes,
                                                               int v_scale_bytes) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];      // [buffer][K8 tile | Vt8 tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h5 = lane >> 5, l31 = lane & 31;
  const int nq = (N + QT - 1) / QT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pair = (slot / nq) * 8 + xcd, qt = slot - (slot / nq) * nq;
  if (pair >= npairs) return;
  const int head = pair % heads, b = pair / heads;
  const int D = heads * HD;
  const int q0 = qt * QT + wave * 32;
  const size_t npad = (size_t)nt * KT;

  // this lane's half of its query row: 32 bytes = slots (h5, 0..31) = d 32 h5 .. 32 h5 + 31 (rows >= N are zero in Q8: harmless)
  v8i qf;
  {
    int qr = q0 + l31;
    qr = qr < (int)npad ? qr : (int)npad - 1;
    const char* qp = q8 + ((size_t)pair * npad + qr) * 64 + 32 * h5;
    const u32x4 a = *reinterpret_cast<const u32x4*>(qp), c = *reinterpret_cast<const u32x4*>(qp + 16);
    qf = (v8i){(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)c[0], (int)c[1], (int)c[2], (int)c[3]};
  }

  // K8 / Vt8 tiles of this pair as buffers; one 16-byte DMA per thread and tile for each: thread -> (row = tid >> 2, chunk = tid & 3),
  // LDS image linear in thread order, swizzle applied to the SOURCE chunk
  const auto rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(k8 + (size_t)pair * npad * 64), 0, (unsigned)(npad * 64), 0x00020000);
  const auto rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vt8 + (size_t)pair * npad * 64), 0, (unsigned)(npad * 64), 0x00020000);
  unsigned src_off;
  {
    const int row = wave * 16 + (lane >> 2), ch = lane & 3;
    src_off = (unsigned)(row * 64 + swz64(row, ch) * 16);
  }
  auto stage = [&](auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    char* dst = smem + BUF * (2 * TILE_BYTES) + wave * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)dst, 16, src_off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(dst + TILE_BYTES), 16, src_off, 0, 0, 0);
    src_off += TILE_BYTES;
  };
  // loop-invariant LDS offsets of this lane's two 16-byte chunks (logical chunks 2 h5, 2 h5 + 1 of row l31; + 2048 for rows 32..63)
  const int off0 = l31 * 64 + swz64(l31, 2 * h5) * 16, off1 = l31 * 64 + swz64(l31, 2 * h5 + 1) * 16;
  auto frag = [&](const char* base) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(base + off0), c = *reinterpret_cast<const u32x4*>(base + off1);
    return (v8i){(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)c[0], (int)c[1], (int)c[2], (int)c[3]};
  };
  // VROW: this lane's row address in the transposed 8x8-block reads of a row-major V8 tile (header comment): key of row q = u >> 1 of
  // slot group i = 32 (i >> 1) + 16 (i & 1) + 4 h5 + {0,1,2,3,8,9,10,11}[q]; the chunk swizzle (key >> 2) & 3 = (h5 + 2 (q >> 2)) & 3 does
  // not depend on i, so one offset per d half and an immediate per i
  int voff[2] = {0, 0};
  if constexpr (VROW) {
    const int G = lane >> 4, u = lane & 15, qrow = u >> 1, odd = u & 1;
    const int key0 = 4 * h5 + (qrow & 3) + 8 * (qrow >> 2);
    const int sw = (h5 + 2 * (qrow >> 2)) & 3;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) voff[dt] = key0 * 64 + (((2 * dt + (G & 1)) ^ sw) * 16) + 8 * odd;
  }
  typedef int v2i __attribute__((ext_vector_type(2)));
  auto vfrag = [&](const char* vb, int dt) {
    if constexpr (VROW) {
      v8i r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const v2i w = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(vb + voff[dt] + (i >> 1) * 2048 + (i & 1) * 1024));
        r[2 * i] = w[0];
        r[2 * i + 1] = w[1];
      }
      return r;
    } else {
      return frag(vb + dt * 2048);
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = 0.f;
  f32x2_t lsum = {0.f, 0.f};

  auto tile = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    dma_landed_barrier();                                // this wave's DMAs of tile t have landed; everyone is done with tile t-1
    if (t + 1 < nt) stage(IntC<BUF ^ 1>{});
    const char* kb = smem + BUF * (2 * TILE_BYTES);
    const char* vb = kb + TILE_BYTES;

    f32x16 s[2];
    const float neg_m = -m_run;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kt][i] = neg_m;
      // S^T[key][q] = sum_d K8[key][d] Q8[q][d] * 2^-(q_exp + k_exp): A = 32 keys of the tile, B = this wave's 32 queries
      s[kt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(kb + kt * 2048), qf, s[kt], 0, 0, 0, qk_scale_bytes, 0, 0x7F7F7F7F);
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
    // one rescale decision per tile (both 32-key blocks), then all 32 exponentials; probabilities leave as e4m3 in register order
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
    v8i pf;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x2_t e0 = {__builtin_amdgcn_exp2f(s[kt][4 * g]), __builtin_amdgcn_exp2f(s[kt][4 * g + 1])};
        const f32x2_t e1 = {__builtin_amdgcn_exp2f(s[kt][4 * g + 2]), __builtin_amdgcn_exp2f(s[kt][4 * g + 3])};
        lsum[0] += e0[0];                                 // plain v_add_f32 (file built with -fno-slp-vectorize): packed f32 adds cost more beside MFMAs
        lsum[1] += e0[1];
        lsum[0] += e1[0];
        lsum[1] += e1[1];
        pf[kt * 4 + g] = (int)pack4_fp8(e0[0], e0[1], e1[0], e1[1]);
      }
    // O^T[d][q] += sum_key Vt8[d][key] P[key][q] * 2^-v_exp: A = 32 d rows of the tile's Vt8 block, B = P (64 keys x 32 queries)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
      o[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vfrag(vb, dt), pf, o[dt], 0, 0, 0, v_scale_bytes, 0, 0x7F7F7F7F);
  };

  stage(IntC<0>{});
  for (int t = 0; t < nt; t += 2) {
    tile(t, IntC<0>{});
    if (t + 1 < nt) tile(t + 1, IntC<1>{});
  }

  const float lane_sum = lsum[0] + lsum[1];
  const float denom = lane_sum + __shfl_xor(lane_sum, 32, 64);
  const float inv = 1.0f / denom;
  const int q = q0 + l31;
  if (q < N) {
    h_raw* op = out + ((size_t)b * N + q) * D + head * HD + 4 * h5;
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

}  // namespace f8
}  // namespace ucod

extern "C" size_t ucod_attention_fp8_workspace_bytes(int B, int tok, int heads) {
  if (B <= 0 || tok <= 0 || heads <= 0) return 0;
  const size_t npad = (size_t)ucod::cdiv(tok, ucod::f8::KT) * ucod::f8::KT;
  return 3 * (size_t)B * heads * npad * 64;
}

extern "C" int ucod_attention_fwd_fp8(const void* qkv, void* out, void* workspace, size_t workspace_bytes, int B, int tok, int heads, int q_exp,
                                      int k_exp, int v_exp, void* stream) {
  using namespace ucod;
  using namespace ucod::f8;
  if (!qkv || !out || !workspace || B <= 0 || tok <= 0 || heads <= 0) return UCOD_EINVAL;
  if (q_exp + k_exp < -60 || q_exp + k_exp > 60 || v_exp < -60 || v_exp > 60) return UCOD_EINVAL;
  const size_t need = ucod_attention_fp8_workspace_bytes(B, tok, heads);
  if (workspace_bytes < need) return UCOD_ENOMEM;
  const int nt = cdiv(tok, KT), npairs = B * heads, nq = cdiv(tok, QT);
  char* q8 = (char*)workspace;
  char* k8 = q8 + need / 3;
  char* vt8 = k8 + need / 3;
  UCOD_PROF(PROF_ATTN, stream);
  hipLaunchKernelGGL(qkv_to_fp8_kernel, dim3(nt, npairs), dim3(256), 0, (hipStream_t)stream, (const h_raw*)qkv, q8, k8, vt8, tok, heads, nt,
                     __builtin_ldexpf(1.f, q_exp), __builtin_ldexpf(1.f, k_exp), __builtin_ldexpf(1.f, v_exp));
  // E8M0 block scales (one byte per 32-element block, replicated): value 127 + e means 2^e; the products are scaled back by
  // 2^-(q_exp + k_exp) on the K8 operand and 2^-v_exp on the Vt8 operand, the other operand keeps 2^0
  const int qk = 127 - (q_exp + k_exp), vs = 127 - v_exp;
  const int qk_bytes = qk | (qk << 8) | (qk << 16) | (qk << 24), v_bytes = vs | (vs << 8) | (vs << 16) | (vs << 24);
  hipLaunchKernelGGL(attn_fwd_fp8_kernel<false>, dim3(cdiv(npairs, 8) * 8 * nq), dim3(256), 0, (hipStream_t)stream, q8, k8, vt8, (h_raw*)out, tok, heads,
                     npairs, nt, qk_bytes, v_bytes);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

namespace ucod {
namespace f8 {
// rows tok .. npad-1 of every (tensor, image, head): 64 bytes each
__global__ __launch_bounds__(256) void zero_pad_kernel(char* __restrict__ ws, int tok, int npad, int total_pairs3) {
  const int pad = npad - tok;
  const long n = (long)total_pairs3 * pad * 4;                // 16-byte chunks
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long row = i >> 2;
    const long p = row / pad, r = row - p * pad;
    *reinterpret_cast<u32x4*>(ws + ((size_t)p * npad + tok + r) * 64 + (i & 3) * 16) = (u32x4){0u, 0u, 0u, 0u};
  }
}
}  // namespace f8
}  // namespace ucod

extern "C" int ucod_attention_fp8_zero_pad(void* q8k8v8, int B, int tok, int heads, void* stream) {
  using namespace ucod;
  using namespace ucod::f8;
  if (!q8k8v8 || B <= 0 || tok <= 0 || heads <= 0) return UCOD_EINVAL;
  const int npad = cdiv(tok, KT) * KT;
  if (npad == tok) return UCOD_OK;
  const int total = 3 * B * heads;
  hipLaunchKernelGGL(zero_pad_kernel, dim3(cdiv((long)total * (npad - tok) * 4, 256)), dim3(256), 0, (hipStream_t)stream, (char*)q8k8v8, tok, npad, total);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_attention_fwd_fp8_fused(const void* q8k8v8, void* out, int B, int tok, int heads, int q_exp, int k_exp, int v_exp, void* stream) {
  using namespace ucod;
  using namespace ucod::f8;
  if (!q8k8v8 || !out || B <= 0 || tok <= 0 || heads <= 0) return UCOD_EINVAL;
  if (q_exp + k_exp < -60 || q_exp + k_exp > 60 || v_exp < -60 || v_exp > 60) return UCOD_EINVAL;
  const size_t third = ucod_attention_fp8_workspace_bytes(B, tok, heads) / 3;
  const int nt = cdiv(tok, KT), npairs = B * heads, nq = cdiv(tok, QT);
  const char* q8 = (const char*)q8k8v8;
  UCOD_PROF(PROF_ATTN, stream);
  const int qk = 127 - (q_exp + k_exp), vs = 127 - v_exp;
  const int qk_bytes = qk | (qk << 8) | (qk << 16) | (qk << 24), v_bytes = vs | (vs << 8) | (vs << 16) | (vs << 24);
  hipLaunchKernelGGL(attn_fwd_fp8_kernel<true>, dim3(cdiv(npairs, 8) * 8 * nq), dim3(256), 0, (hipStream_t)stream, q8, q8 + third, q8 + 2 * third,
                     (h_raw*)out, tok, heads, npairs, nt, qk_bytes, v_bytes);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}
