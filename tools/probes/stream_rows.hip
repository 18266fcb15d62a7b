// Probe: the batched GEMV's weight stream with its REAL address pattern -- a wave owns R consecutive rows of a row-major [N][K] bf16 matrix
// and copies, per 128-k step, one tile [16 rows][256 B] per 16 rows (LDS-DMA, wave-private ring) -- for row strides of exactly K (a power of
// two for K = 4096: at any moment every wave of the chip reads the same 256-byte window of ITS rows) and K + pad.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_rows tools/probes/stream_rows.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

template <int AUX, int RING, int RT>
__global__ __launch_bounds__(256) void rows_kernel(const char* __restrict__ W, size_t stride, int nsteps, int kbeg_bytes, int rot, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* ring = smem + wave * RING * 4096;
  const size_t r0 = ((size_t)blockIdx.x * 4 + wave) * 16 * RT;
  const int ntiles = nsteps * RT;
  const size_t r0wg = (size_t)blockIdx.x * 4 * 16 * RT;
  // rot 1: rotated K order per row tile (128-k units); 2: per 192-row group of the WORKGROUP's first row, 128-k units; 3: the same in 512-k units
  const int shift = rot == 1 ? (int)((r0 / 16 * 5) % nsteps) : rot == 2 ? (int)((r0wg / 192 * 5) % nsteps) : rot == 3 ? (int)((r0wg / 192 * 5) % (nsteps / 4)) * 4 : 0;
  auto issue = [&](int t) __attribute__((always_inline)) {
    const int tt = t < ntiles ? t : ntiles - 1;
    const int ss = ((tt / RT) + shift) % nsteps, rt = tt % RT;
    char* dst = ring + (t % RING) * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = q * 4 + (lane >> 4);
      const char* p = W + (r0 + rt * 16 + row) * stride + kbeg_bytes + ss * 256 + (((lane & 15) ^ row) << 4);
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lds_ptr_t)(dst + q * 1024), 16, 0, AUX);
    }
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

template <int AUX, int RT>
void run(const char* name, const char* W, size_t stride, int grid, int nsteps, int kbeg, int rot, unsigned* sink) {
  constexpr int RING = 7;
  const size_t smem = 4 * RING * 4096;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_kernel<AUX, RING, RT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rows_kernel<AUX, RING, RT>), dim3(grid), dim3(256), smem, 0, W, stride, nsteps, kbeg, rot, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    if (rep >= 1 && t < best) best = t;
  }
  const double bytes = (double)grid * 4 * 16 * RT * nsteps * 256;
  printf("%-34s grid %3d RT %d steps %2d stride %5zu rot %d aux %d: %6.1f us  %5.2f TB/s\n", name, grid, RT, nsteps, stride, rot, AUX, best * 1e3, bytes / 1e9 / best);
}

int main() {
  char* W; hipMalloc(&W, (size_t)1 << 30); hipMemset(W, 1, (size_t)1 << 30);
  unsigned* sink; hipMalloc(&sink, 1 << 20);
  for (size_t stride : {(size_t)8192, (size_t)8448, (size_t)8704, (size_t)9216}) {
    run<2, 2>("gate/up (28672 x 4096)", W, stride, 224, 32, 0, 0, sink);
    run<0, 2>("gate/up (28672 x 4096)", W, stride, 224, 32, 0, 0, sink);
    run<2, 1>("qkv K half (3 x 16 rows / WG)", W, stride, 256, 16, 0, 0, sink);
    run<2, 1>("o K quarter", W, stride, 256, 8, 0, 0, sink);
  }
  for (int rot = 1; rot <= 3; ++rot) {
    run<2, 2>("gate/up rotated K order", W, 8192, 224, 32, 0, rot, sink);
    run<2, 1>("qkv K half rotated", W, 8192, 256, 16, 0, rot, sink);
    run<2, 1>("o K quarter rotated", W, 8192, 256, 8, 0, rot, sink);
    run<2, 1>("down K quarter rotated", W, 28672, 256, 28, 0, rot, sink);
    run<2, 1>("lm_head-like (64-row WGs, K 4096)", W, 8192, 1024, 32, 0, rot, sink);
  }
  run<2, 1>("lm_head-like (64-row WGs, K 4096)", W, 8192, 1024, 32, 0, 0, sink);
  run<2, 1>("down K quarter (4096 x 14336)", W, 28672, 256, 28, 0, 0, sink);
  run<2, 1>("down K quarter stride + 256", W, 28928, 256, 28, 0, 0, sink);
  return 0;
}
