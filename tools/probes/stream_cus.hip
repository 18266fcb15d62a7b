// Probe: HBM rate of the batched GEMV's weight stream (LDS-DMA tiles [16 rows][256 B] into wave-private rings, no arithmetic) by number of
// streaming workgroups (one per CU) and ring depth: is ~25 GB/s per CU the chip's rate / 256, or a per-CU limit that holds when fewer CUs
// stream?   hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_cus tools/probes/stream_cus.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

template <int AUX, int RING, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream_kernel(const char* __restrict__ W, size_t bytes_per_wave, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = smem + wave * RING * 4096;
  const char* base = W + ((size_t)blockIdx.x * WAVES + wave) * bytes_per_wave + lane * 16;
  const int ntiles = (int)(bytes_per_wave / 4096);
  auto issue = [&](int t) __attribute__((always_inline)) {
    const int tt = t < ntiles ? t : ntiles - 1;
    const char* p = base + (size_t)tt * 4096;
    char* dst = ring + (t % RING) * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(p + q * 1024), (lds_ptr_t)(dst + q * 1024), 16, 0, AUX);
  };
#pragma unroll
  for (int i = 0; i < RING; ++i) issue(i);
  unsigned acc = 0;
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * 4) : "memory");
    acc ^= *reinterpret_cast<const unsigned*>(ring + (t % RING) * 4096 + lane * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    issue(t + RING);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

template <int AUX, int RING, int WAVES>
void run(const char* W, size_t total, int grid, unsigned* sink) {
  const size_t smem = (size_t)WAVES * RING * 4096;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<AUX, RING, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const size_t per_wave = total / ((size_t)grid * WAVES) / 4096 * 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<AUX, RING, WAVES>), dim3(grid), dim3(WAVES * 64), smem, 0, W, per_wave, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    if (rep >= 1 && t < best) best = t;
  }
  const double bytes = (double)per_wave * grid * WAVES;
  printf("grid %3d x %d waves, ring %2d tiles (%3zu KB in flight per CU), aux %d: %7.1f us  %5.2f TB/s  %5.1f GB/s per CU\n", grid, WAVES, RING, smem / 1024, AUX,
         best * 1e3, bytes / 1e9 / best, bytes / 1e6 / best / grid);
}

int main() {
  const size_t bytes = (size_t)448 << 20;
  char* W; hipMalloc(&W, bytes); hipMemset(W, 1, bytes);
  unsigned* sink; hipMalloc(&sink, 1 << 20);
  for (int grid : {256, 224, 192, 160, 128, 64}) {
    run<2, 7, 4>(W, bytes / 2, grid, sink);
    run<2, 9, 4>(W, bytes / 2, grid, sink);
    run<2, 4, 8>(W, bytes / 2, grid, sink);
    run<0, 7, 4>(W, bytes / 2, grid, sink);
  }
  return 0;
}
