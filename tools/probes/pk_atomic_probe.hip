// Probe: what does a fused attention backward pay for its dQ accumulation if dQ is summed across key-workgroups with PACKED bf16 atomics
// (global_atomic_pk_add_bf16: 2 bf16 per lane, 256 B per wave instruction) instead of f32 atomics?  The access pattern of the 256-key form priced in
// docs/lab/r05.md #5: one workgroup per (image, head, 256-key slab) = 32 x 12 x 6 workgroups of 4 waves, each walking the 43 blocks of 32 queries and adding its
// 32 x 64 partial dQ tile into dQ[b, q, h, 0..63] (row pitch D = 768 elements): 6 adders per address, 0.40 GB of bf16 adds per layer (0.81 GB as f32).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/pk_atomic_probe.hip -o /tmp/pk_probe && /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>   // 0: global_atomic_pk_add_bf16 on bf16 dQ, 1: global_atomic_add_f32 on f32 dQ, 2: plain stores of the bf16 tile (the traffic without the adds)
__global__ __launch_bounds__(256) void dq_probe(void* dq, int tok, int heads, int D, int slabs) {
  const int wg = blockIdx.x;
  const int slab = wg % slabs, bh = wg / slabs;
  const int b = bh / heads, h = bh - b * heads;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nqb = (tok + 31) / 32;
  for (int qb0 = 0; qb0 < nqb; ++qb0) {
    const int qb = (qb0 + slab * 7) % nqb;                       // the slabs of one head walk the query blocks out of phase (as staggered workgroups would)
    // the 32 x 64 tile as 4 waves x 4 instructions; one wave instruction = 2 query rows x 32 packed dwords (MODE 0 / 2) or 1 row x 64 floats (MODE 1)
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = qb * 32 + wave * 8 + i;
        if (q >= tok) continue;
        float* p = (float*)dq + ((size_t)(b * tok + q) * D + h * 64 + lane);
        atomicAdd(p, 1.0f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = qb * 32 + wave * 8 + i * 2 + (lane >> 5);
        if (q >= tok) continue;
        unsigned* p = (unsigned*)dq + ((size_t)(b * tok + q) * D + h * 64) / 2 + (lane & 31);
        const unsigned one2 = 0x3F803F80u;      // (1.0bf16, 1.0bf16)
        if (MODE == 0) asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"(p), "v"(one2) : "memory");
        else *p = one2;
      }
    }
  }
}

int main() {
  const int B = 32, tok = 1370, heads = 12, D = 768, slabs = 6;
  void* dq;
  hipMalloc(&dq, (size_t)B * tok * D * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double elems = (double)B * tok * D * slabs;              // added elements per launch
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(dq, 0, (size_t)B * tok * D * 4);
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(dq_probe<0>, dim3(B * heads * slabs), dim3(256), 0, 0, dq, tok, heads, D, slabs);
      if (mode == 1) hipLaunchKernelGGL(dq_probe<1>, dim3(B * heads * slabs), dim3(256), 0, 0, dq, tok, heads, D, slabs);
      if (mode == 2) hipLaunchKernelGGL(dq_probe<2>, dim3(B * heads * slabs), dim3(256), 0, 0, dq, tok, heads, D, slabs);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const int bytes_per = mode == 1 ? 4 : 2;
    printf("%-46s %8.1f us   %6.2f TB/s of added bytes   %6.2f TB/s in f32-equivalent bytes (4 B per element)\n",
           mode == 0 ? "global_atomic_pk_add_bf16 (bf16 dQ, 6 adders)" : mode == 1 ? "global_atomic_add_f32     (f32 dQ, 6 adders)" : "plain 4-byte stores of the same tiles",
           best * 1e3, elems * bytes_per / (best * 1e-3) / 1e12, elems * 4 / (best * 1e-3) / 1e12);
    if (mode == 0) {                                              // the sums are exact in bf16 while they stay below 256: 6 adders x 6 launches = 36
      unsigned short hsum;
      hipMemcpy(&hsum, dq, 2, hipMemcpyDeviceToHost);
      printf("  (first element after 6 launches: bf16 bits 0x%04x = %g; expected 36)\n", hsum, (double)(*(float*)(unsigned[]){(unsigned)hsum << 16}));
    }
  }
  return 0;
}
