// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of the UCOD-DPL hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

// Measurement knobs (alternative kernel forms, grid sizes, store policies: what tools/ sweeps) are read from the environment ONLY in builds made with
// -DUCOD_LAB_KNOBS (`make knobs` -> ../_native/libucod_dpl*_knobs.so, selected by tools/ through UCOD_DPL_EXPERIMENT_LIB); in the product libraries
// ucod::lab_env() is a constant nullptr and every knob takes its default.
#include <cstdlib>
namespace ucod {
#ifdef UCOD_LAB_KNOBS
inline const char* lab_env(const char* name) { return getenv(name); }
#else
inline const char* lab_env(const char*) { return nullptr; }
#endif
}  // namespace ucod

#define UCOD_OK 0
#define UCOD_EINVAL (-1)
#define UCOD_ENOMEM (-2)

#define UCOD_CHECK_LAUNCH()                          \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef unsigned short bf16_raw;

// ---- 16-bit operand type of the FORWARD path (GEMM operands, LayerNorm output, attention) ----------------------------------
// Default build: bf16 (BASELINE configs[1]).  -DUCOD_HALF_F16 builds the same kernels on IEEE fp16 operands (11 significand bits
// instead of 8; the arithmetic type of the reference's own fp16-autocast launcher): libucod_dpl_f16.so, selected per engine with
// ViTEngine(half="f16").  Same MFMA rate, same bytes; the backbone-backward entry points are bf16-only and refuse to run there.
#ifdef UCOD_HALF_F16
typedef _Float16 half_t;
#define UCOD_HALF_NAME "f16"
#define UCOD_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define UCOD_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define UCOD_MFMA32_ASM "v_mfma_f32_32x32x16_f16"
typedef __fp16 ucod_fp16x4_b __attribute__((__vector_size__(4 * sizeof(__fp16))));    // the builtin's own vector type
#define UCOD_TR16(p) __builtin_bit_cast(hx4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) ucod_fp16x4_b*)(p)))
#else
typedef __bf16 half_t;
#define UCOD_HALF_NAME "bf16"
#define UCOD_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define UCOD_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define UCOD_MFMA32_ASM "v_mfma_f32_32x32x16_bf16"
#define UCOD_TR16(p) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) hx4*)(p))
#endif
typedef __attribute__((ext_vector_type(8))) half_t hx8;
typedef __attribute__((ext_vector_type(4))) half_t hx4;
typedef __attribute__((ext_vector_type(2))) half_t hx2;
typedef unsigned short h_raw;
// backbone-backward entry points: bf16 only (their kernels unpack operands by bit tricks that are bf16-specific)
#ifdef UCOD_HALF_F16
#define UCOD_BF16_ONLY() return UCOD_EINVAL
#else
#define UCOD_BF16_ONLY() do { } while (0)
#endif

namespace ucod {

constexpr int WAVE = 64;

__device__ __forceinline__ float h_to_f32(h_raw v) { return (float)__builtin_bit_cast(half_t, v); }
__device__ __forceinline__ h_raw f32_to_h(float f) { return __builtin_bit_cast(h_raw, (half_t)f); }      // round-to-nearest-even
// two conversions in one instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned pack_h2(float lo, float hi) {
  typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((pk_f32x2){lo, hi}, hx2));
}
// IEEE fp16 pairs regardless of the build's operand type (the f16 residual stream, ucod_vit_desc.resid16)
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
  typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((pk_f32x2){lo, hi}, f16x2_t));
}
__device__ __forceinline__ void unpack_f16x2(unsigned w, float& lo, float& hi) {
  const f16x2_t v = __builtin_bit_cast(f16x2_t, w);
  lo = (float)v[0];
  hi = (float)v[1];
}
// The f16 residual stream saturates instead of overflowing (an inf in x turns every later LayerNorm of the row into NaN with no error):
// values are clamped to +-65504 before the conversion, and the number of saturated (or NaN) elements is added to a device counter the
// host polls (ucod_resid16_overflow_*, include/ucod_dpl.h).
constexpr float F16_MAX = 65504.f;
__device__ __forceinline__ float clamp_f16(float x) { return __builtin_amdgcn_fmed3f(x, -F16_MAX, F16_MAX); }
__device__ __forceinline__ bool beyond_f16(float x) { return !(__builtin_fabsf(x) <= F16_MAX); }     // true for NaN too
// the two values of a packed register as f32
__device__ __forceinline__ void unpack_h2(unsigned w, float& lo, float& hi) {
#ifdef UCOD_HALF_F16
  const hx2 v = __builtin_bit_cast(hx2, w);
  lo = (float)v[0];
  hi = (float)v[1];
#else
  lo = __uint_as_float(w << 16);
  hi = __uint_as_float(w & 0xFFFF0000u);
#endif
}

__device__ __forceinline__ float bf16_to_f32(bf16_raw v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_raw, b);
}

// two round-to-nearest-even conversions in ONE v_cvt_pk_bf16_f32 (the scalar form costs cvt, cvt, shift, or)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((pk_f32x2){lo, hi}, pk_bf16x2));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x a multiple of 64 (<= 1024); result valid in every thread
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem /* >= 16 entries */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  T r = 0;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
// the DBA gate's sigmoid (128 per pixel in heads_fwd / dba_bwd, which were bound by expf + the IEEE division of sigmoid_acc): v_exp_f32 of
// the scaled argument and v_rcp_f32 -- 1 ulp each plus |x| * 6e-8 from the argument's rounding, against the 5e-5 the step is pinned to
__device__ __forceinline__ float sigmoid_gate(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// device address of the f16-residual-stream saturation counter of the current device (vit_misc.hip)
unsigned* resid16_overflow_counter();
// ucod_accumulators_prezeroed (elementwise.hip): the caller has zeroed the accumulating outputs of the calls it issues next from this host thread
// (one ucod_zero_segments launch per step instead of a memset in front of every producer)
bool accumulators_prezeroed();

// op classes for the optional event profiler (prof.hip); the first six follow the UCOD_EPI_* numbering
enum ProfClass {
  PROF_GEMM_EPI0 = 0, PROF_GEMM_EPI1, PROF_GEMM_EPI2, PROF_GEMM_EPI3, PROF_GEMM_EPI4, PROF_GEMM_EPI5, PROF_ATTN, PROF_LN,
  PROF_IM2COL, PROF_CLS, PROF_BILINEAR, PROF_DBA_PROJECT, PROF_DBA_COLNORM, PROF_DBA_HEADS, PROF_ORTH, PROF_DBA_BWD,
  PROF_DBA_WGRAD, PROF_DISC_FWD, PROF_DISC_BWD, PROF_APM, PROF_BINARIZE, PROF_ADAMW, PROF_CROP, PROF_CAST, PROF_LN_BWD, PROF_LORA,
  PROF_ATTN_BWD, PROF_GEMM_EPI6, PROF_GEMM_EPI7, PROF_ROW_STATS, PROF_SPLIT, PROF_LN_SPLIT, PROF_ATTN_SPLIT, PROF_NUM
};
struct ProfScope {
  int idx;
  hipStream_t stream;
  ProfScope(int cls, hipStream_t s);
  ~ProfScope();
};
#define UCOD_PROF(cls, stream) ucod::ProfScope prof_scope__((cls), (hipStream_t)(stream))


// Workgroup barrier of a kernel that stages tiles by LDS-DMA (buffer_load / global_load ... lds): every wave first waits for ITS OWN
// DMAs (vmcnt(0)), then the barrier makes all waves' pieces visible.  __syncthreads() alone does not imply the vmcnt wait on gfx950 (a
// workgroup-scope fence needs none for global memory); hipcc usually adds one in front of the first LDS read that may alias a DMA in
// flight, but that is a property of its alias analysis, not a guarantee -- and reads issued from asm statements get none.
#if defined(__HIPCC__)
__device__ __forceinline__ void dma_landed_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
#endif
}  // namespace ucod
