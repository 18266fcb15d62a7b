// Probe: does touching a weight matrix ahead of time (one dword per 128-byte line, the data thrown away) make the later LDS-DMA stream of
// the same bytes faster -- i.e. do the lines wait in the Infinity Cache (256 MB, memory side) / L2 for the stream?
//   stream kernel: 256 workgroups x 4 waves, each wave copies its rows as [16 rows][256 B] tiles into a private LDS ring (the batched GEMV's
//   weight stream, no arithmetic), nt or default policy.
//   touch kernel : every lane loads one dword of a different 128-byte line (grid-stride), default or nt policy.
// Sequence per measurement: [flush: stream a 512 MB scratch buffer] -> [touch W | touch other | nothing] -> stream W (timed with events).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mall_prefetch tools/probes/mall_prefetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

template <int AUX, int RING>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ W, size_t bytes_per_wave, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = smem + wave * RING * 4096;
  const char* base = W + ((size_t)blockIdx.x * 4 + wave) * bytes_per_wave + lane * 16;
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

template <int NT>
__global__ __launch_bounds__(256) void touch_kernel(const char* __restrict__ W, size_t bytes, unsigned* sink) {
  unsigned acc = 0;
  const size_t lines = bytes / 128;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < lines; i += (size_t)gridDim.x * 256) {
    const unsigned* p = reinterpret_cast<const unsigned*>(W + i * 128);
    acc ^= NT ? __builtin_nontemporal_load(p) : *p;
  }
  if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
  const size_t MB = argc > 1 ? atoi(argv[1]) : 224;          // bytes streamed = MB * 2^20 (multiple of 1 MB: 1024 waves x 1 KB...)
  const size_t bytes = MB << 20, per_wave = bytes / 1024;    // 256 workgroups x 4 waves
  char *W, *other, *flush;
  hipMalloc(&W, bytes); hipMalloc(&other, bytes); hipMalloc(&flush, (size_t)768 << 20);
  hipMemset(W, 1, bytes); hipMemset(other, 2, bytes); hipMemset(flush, 3, (size_t)768 << 20);
  unsigned* sink; hipMalloc(&sink, 1 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  constexpr int RING = 7;
  const size_t smem = 4 * RING * 4096;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<2, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<0, RING>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  auto run = [&](const char* name, int touch_mode /*0 none, 1 W default, 2 W nt, 3 other*/, int stream_nt, size_t touch_bytes) {
    std::vector<float> ms;
    for (int rep = 0; rep < 6; ++rep) {
      hipLaunchKernelGGL((stream_kernel<2, RING>), dim3(256), dim3(256), smem, 0, flush, ((size_t)768 << 20) / 1024, sink);   // evict
      if (touch_mode == 1) hipLaunchKernelGGL(touch_kernel<0>, dim3(1024), dim3(256), 0, 0, W, touch_bytes, sink);
      if (touch_mode == 2) hipLaunchKernelGGL(touch_kernel<1>, dim3(1024), dim3(256), 0, 0, W, touch_bytes, sink);
      if (touch_mode == 3) hipLaunchKernelGGL(touch_kernel<0>, dim3(1024), dim3(256), 0, 0, other, touch_bytes, sink);
      hipEventRecord(e0);
      if (stream_nt) hipLaunchKernelGGL((stream_kernel<2, RING>), dim3(256), dim3(256), smem, 0, W, per_wave, sink);
      else hipLaunchKernelGGL((stream_kernel<0, RING>), dim3(256), dim3(256), smem, 0, W, per_wave, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float t; hipEventElapsedTime(&t, e0, e1);
      if (rep >= 2) ms.push_back(t);
    }
    float best = 1e9, sum = 0; for (float t : ms) { best = t < best ? t : best; sum += t; }
    printf("%-44s stream %s: best %7.1f us  mean %7.1f us  = %5.2f TB/s\n", name, stream_nt ? "nt " : "def", best * 1e3, sum / ms.size() * 1e3, bytes / 1e9 / best);
  };
  // time of the touch itself
  {
    hipLaunchKernelGGL((stream_kernel<2, RING>), dim3(256), dim3(256), smem, 0, flush, ((size_t)768 << 20) / 1024, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(touch_kernel<0>, dim3(1024), dim3(256), 0, 0, W, bytes, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    printf("touch of %zu MB (one dword per 128 B): %.1f us = %.2f TB/s of lines\n", MB, t * 1e3, bytes / 1e9 / t);
  }
  for (int snt = 1; snt >= 0; --snt) {
    run("no touch", 0, snt, 0);
    run("touch OTHER buffer (control)", 3, snt, bytes);
    run("touch W (default policy)", 1, snt, bytes);
    run("touch W (nt)", 2, snt, bytes);
    run("touch first 64 MB of W (default)", 1, snt, (size_t)64 << 20);
    run("touch first 128 MB of W (default)", 1, snt, (size_t)128 << 20);
  }
  return 0;
}
