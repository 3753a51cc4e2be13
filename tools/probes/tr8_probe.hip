// Probe of ds_read_b64_tr_b8 on gfx950: which LDS bytes does lane l receive, given the per-lane addresses?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/tr8_probe.hip -o /tmp/tr8_probe && /tmp/tr8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));

__global__ void probe(const int* lane_addr, unsigned* out_lo, unsigned* out_hi) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned char)(i & 0xFF);
  __syncthreads();
  v2i a = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds + lane_addr[threadIdx.x]));
  __syncthreads();
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned char)((i >> 8) & 0xFF);
  __syncthreads();
  v2i b = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)(lds + lane_addr[threadIdx.x]));
  // byte k of the result came from LDS address (b_k << 8) | a_k
  for (int w = 0; w < 2; ++w) {
    out_lo[threadIdx.x * 2 + w] = (unsigned)a[w];
    out_hi[threadIdx.x * 2 + w] = (unsigned)b[w];
  }
}

int main() {
  int h_addr[64];
  unsigned h_lo[128], h_hi[128];
  int *d_addr; unsigned *d_lo, *d_hi;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_lo, sizeof(h_lo)); hipMalloc(&d_hi, sizeof(h_hi));
  for (int cfg = 0; cfg < 2; ++cfg) {
    // cfg 0: lane l -> row l of a [64 rows][64 B] image, byte 0 ; cfg 1: lane l -> row (l & 15) of 128-B rows, byte 8 * (l >> 4)
    for (int l = 0; l < 64; ++l) h_addr[l] = cfg == 0 ? l * 64 : (l & 15) * 128 + 8 * (l >> 4);
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_lo, d_hi);
    hipMemcpy(h_lo, d_lo, sizeof(h_lo), hipMemcpyDeviceToHost);
    hipMemcpy(h_hi, d_hi, sizeof(h_hi), hipMemcpyDeviceToHost);
    printf("== cfg %d: lane: supplied address -> the 8 source addresses of its result bytes (as row*stride+col)\n", cfg);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d addr %5d :", l, h_addr[l]);
      for (int k = 0; k < 8; ++k) {
        const unsigned lo = (h_lo[l * 2 + k / 4] >> (8 * (k % 4))) & 0xFF, hi = (h_hi[l * 2 + k / 4] >> (8 * (k % 4))) & 0xFF;
        printf(" %5u", (hi << 8) | lo);
      }
      printf("\n");
    }
  }
  return 0;
}
