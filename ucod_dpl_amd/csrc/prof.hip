// Per-op HIP-event timing on the launch stream (used by bench.py for the roofline figures: the duration of each
// kernel class is measured live, inside the timed region, on the stream the kernels are launched on).
#include "common.h"
#include "../../include/ucod_dpl.h"
#include <vector>
#include <mutex>

namespace ucod {

static const char* kNames[PROF_NUM] = {
    "gemm_bf16_qkv_bias", "gemm_bf16_fc1_gelu", "gemm_bf16_proj_fc2_scale_resid", "gemm_bf16_patch_embed", "gemm_bf16_key_nchw",
    "gemm_bf16_bias_f32", "attention_fwd", "layernorm", "patch_im2col", "cls_rows", "bilinear_resize", "dba_project_f32",
    "dba_colnorm", "dba_heads_fwd", "orth_gram_fwd", "dba_bwd", "dba_wgrad_f32", "disc_fwd", "disc_bwd", "apm_bce", "binarize",
    "adamw_ema", "crop_resize_norm", "cast", "layernorm_bwd", "lora_rowwise", "attention_bwd", "gemm_bf16_gelu_bwd",
    "gemm_bf16_fc1_gelu_save", "row_stats", "split_operands", "layernorm_split", "attention_split_fwd"};

struct Rec { int cls; hipEvent_t a, b; };
static std::mutex g_mu;
static bool g_on = false;
static std::vector<Rec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

ProfScope::ProfScope(int cls, hipStream_t s) : idx(-1), stream(s) {
  if (!g_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r{cls, get_event(), get_event()};
  if (!r.a || !r.b) return;
  (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  idx = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[idx].b, stream);
}

}  // namespace ucod

using namespace ucod;

// (s_memtime, s_memrealtime) of the CU workgroup 0 lands on: see include/ucod_dpl.h
__global__ void clock_probe_kernel(unsigned long long* __restrict__ out) {
  if (threadIdx.x == 0) {
    unsigned long long t, rt;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(rt)::"memory");
    out[0] = t;
    out[1] = rt;
  }
}
extern "C" int ucod_clock_probe(unsigned long long* out_dev, void* stream) {
  if (!out_dev) return UCOD_EINVAL;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev);
  UCOD_CHECK_LAUNCH();
  return UCOD_OK;
}

extern "C" int ucod_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int prev = g_on ? 1 : 0;
  g_on = on != 0;
  return prev;
}
extern "C" int ucod_prof_num_classes(void) { return PROF_NUM; }
extern "C" const char* ucod_prof_class_name(int cls) { return (cls >= 0 && cls < PROF_NUM) ? kNames[cls] : ""; }
extern "C" int ucod_prof_collect(double* total_ms, long long* count) {
  if (!total_ms || !count) return UCOD_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < PROF_NUM; ++i) { total_ms[i] = 0.0; count[i] = 0; }
  for (auto& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { total_ms[r.cls] += ms; count[r.cls] += 1; }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return UCOD_OK;
}
