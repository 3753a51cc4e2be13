// Probe (first gate): how fast does the chip DELIVER the operand K-tiles of a large-tile GEMM into LDS when nothing is multiplied?
// The 256 x 256 x 64 tiling of the product kernels moves (256 + 256) x K x 2 bytes per 256 x 256 outputs; a 256 x 384 x 32 tiling (one wave per SIMD, 384
// accumulators per wave) would move (256 + 384) x K x 2 per 256 x 384 outputs: 17 % fewer L2 -> LDS bytes per flop.  The round-5 hand-placed kernel's DMA-only
// ablation ran the QKV shape in 72.7 us; the gate for building the new tiling is <= 62 us for ITS DMA-only form.  This probe issues exactly the LDS-DMA stream of
// either tiling (16-byte buffer_load ... lds, NST stages, one counted wait + one barrier per K-tile, one workgroup per output tile, XCD-chunked tile order,
// column-tile fastest) and nothing else.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -w tools/probes/dma_tile_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CNT>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory"); }

template <int BM, int BN, int BK, int NST, int NW>
__device__ __forceinline__ void dma_body(char* smem, const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, int M, int N, int K, int tiles_m, int tiles_n,
                                         unsigned* __restrict__ sink) {
  constexpr int ROWB = BK * 2;                       // bytes of a tile row
  constexpr int RPI = 1024 / ROWB;                   // rows per wave instruction (1 KiB)
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int PA = BM / RPI, PB = BN / RPI;        // 1-KiB pieces per K-tile
  constexpr int PPW = (PA + PB + NW - 1) / NW;       // pieces per wave and K-tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = tiles_m * tiles_n, orig = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int wg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const unsigned long bytesA = (unsigned long)M * K * 2ul, bytesB = (unsigned long)N * K * 2ul;
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(A), 0, (unsigned)bytesA, 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(B), 0, (unsigned)bytesB, 0x00020000);
  unsigned off[PPW];
  bool isA[PPW], live[PPW];
  int dst[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int piece = i * NW + wave;
    live[i] = piece < PA + PB;
    isA[i] = piece < PA;
    const int pr = isA[i] ? piece : piece - PA;
    const int row = pr * RPI + lane / (ROWB / 16), chunk = lane % (ROWB / 16);
    const int grow = (isA[i] ? m0 : n0) + row;
    off[i] = ((unsigned)grow * (unsigned)K) * 2u + (unsigned)chunk * 16u;     // rows past the end fail the descriptor's range check: zeros
    dst[i] = (isA[i] ? 0 : BM * ROWB) + pr * 1024;
  }
  const int nt = K / BK;
  auto stage = [&](int t) {
    char* base = smem + (t % NST) * STAGE;
    const unsigned kt = (unsigned)t * ROWB;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      if (!live[i]) continue;
      if (isA[i]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + dst[i]), 16, off[i], kt, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base + dst[i]), 16, off[i], kt, 0, 0);
    }
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nt) stage(t);
  unsigned acc = 0;
  for (int t = 0; t < nt; ++t) {
    if (t + NST - 1 < nt) stage(t + NST - 1);
    // K-tile t has landed when at most the pieces of the NST - 1 younger K-tiles are still in flight (pieces per wave are equal for the live waves)
    if (t + NST - 1 < nt) wait_vm<(NST - 1) * PPW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    acc += *reinterpret_cast<const unsigned*>(smem + (t % NST) * STAGE + ((threadIdx.x * 16) % STAGE));   // one LDS read per K-tile keeps the stream honest
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int BM, int BN, int BK, int NST, int NW>
__global__ __launch_bounds__(NW * 64) void dma_only(const unsigned short* A, const unsigned short* B, int M, int N, int K, int tiles_m, int tiles_n, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem_dyn[];
  dma_body<BM, BN, BK, NST, NW>(smem_dyn, A, B, M, N, K, tiles_m, tiles_n, sink);
}

template <int BM, int BN, int BK, int NST, int NW>
float run(const unsigned short* A, const unsigned short* B, int M, int N, int K, unsigned* sink) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  const size_t lds = (size_t)NST * (BM + BN) * BK * 2;
  hipFuncSetAttribute((const void*)dma_only<BM, BN, BK, NST, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 12; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((dma_only<BM, BN, BK, NST, NW>), dim3(tiles_m * tiles_n), dim3(NW * 64), lds, 0, A, B, M, N, K, tiles_m, tiles_n, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2 && ms < best) best = ms;
  }
  const double bytes = (double)tiles_m * tiles_n * K * (BM + BN) * 2.0;
  printf("  %3d x %3d x %2d, %d stages, %d waves: %4d tiles, %6.1f MB L2->LDS, %7.1f us  = %5.2f TB/s into LDS\n", BM, BN, BK, NST, NW, tiles_m * tiles_n, bytes / 1e6, best * 1e3,
         bytes / (best * 1e-3) / 1e12);
  return best;
}

int main() {
  const int M = 43840, K = 768;
  unsigned short *A, *B;
  unsigned* sink;
  hipMalloc(&A, (size_t)M * K * 2);
  hipMalloc(&B, (size_t)3072 * K * 2);
  hipMalloc(&sink, 64);
  hipMemset(A, 0x11, (size_t)M * K * 2);
  hipMemset(B, 0x22, (size_t)3072 * K * 2);
  for (int N : {2304, 3072}) {
    printf("DMA-only operand stream, M = %d, N = %d, K = %d (%s)\n", M, N, K, N == 2304 ? "QKV" : "fc1");
    run<256, 256, 64, 2, 8>(A, B, M, N, K, sink);      // the product's tiling
    run<256, 256, 32, 3, 8>(A, B, M, N, K, sink);
    run<256, 384, 32, 3, 4>(A, B, M, N, K, sink);      // the proposed one: one wave per SIMD
    run<256, 384, 32, 3, 8>(A, B, M, N, K, sink);
    run<256, 384, 64, 2, 8>(A, B, M, N, K, sink);      // 160 KiB of LDS: does not fit beside anything else
  }
  return 0;
}
