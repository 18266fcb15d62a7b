// Probe: does the SHAPE of a wave's load instruction matter for HBM streaming of a row-major [rows][K] bf16 matrix when a wave
// owns 32 rows (the batched decode GEMV's situation)?  Every variant moves the same bytes with the same number of 16-byte loads in
// flight (32 per lane, two sets); only the bytes per row per instruction differ: PW = 64 (MFMA fragment order), 256 (what
// gemv_mfma3/4 copy), 512, 1024 (one row per instruction, what the single-row stream kernel does).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/stream_pattern tools/probes/stream_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PW, int NT, int NL = 32>
__global__ __launch_bounds__(256) void stream_kernel(const char* __restrict__ W, int K2 /* bytes per row */, int rows_per_wave, u32x4* out) {
  constexpr int LPR = PW / 16;          // lanes per row
  constexpr int RPI = 64 / LPR;         // rows per instruction
  constexpr int GROUPS = 32 / RPI;      // instructions per 32 rows of one k-block
  constexpr int KB = NL / GROUPS;       // k-blocks per step  (step = 32 rows x KB*PW bytes = NL KB per wave)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t r0 = ((size_t)blockIdx.x * 4 + wave) * 32;
  const char* base = W + (r0 + lane / LPR) * (size_t)K2 + (lane % LPR) * 16;
  u32x4 acc = {0, 0, 0, 0};
  const int nstep = K2 / (KB * PW);
  u32x4 v[2][NL];
  auto issue = [&](int s, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int g = i % GROUPS, kb = i / GROUPS;
      const char* p = base + (size_t)g * RPI * K2 + (size_t)(s * KB + kb) * PW;
      if (NT) v[set][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
      else v[set][i] = *reinterpret_cast<const u32x4*>(p);
    }
  };
  issue(0, 0);
  for (int s = 0; s < nstep; s += 2) {
    if (s + 1 < nstep) issue(s + 1, 1);
#pragma unroll
    for (int i = 0; i < NL; ++i) acc ^= v[0][i];
    if (s + 2 < nstep) issue(s + 2, 0);
    if (s + 1 < nstep) {
#pragma unroll
      for (int i = 0; i < NL; ++i) acc ^= v[1][i];
    }
  }
  if (acc.x == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int PW, int NT, int NL = 32>
float run(const std::vector<char*>& copies, int rows, int K2, u32x4* out, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = rows / 128;
  for (auto c : copies) hipLaunchKernelGGL((stream_kernel<PW, NT, NL>), dim3(grid), dim3(256), 0, 0, c, K2, 32, out);
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r)
    for (auto c : copies) hipLaunchKernelGGL((stream_kernel<PW, NT, NL>), dim3(grid), dim3(256), 0, 0, c, K2, 32, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / (reps * copies.size());
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 28672, K = argc > 2 ? atoi(argv[2]) : 4096;
  const int K2 = K * 2;
  const size_t bytes = (size_t)rows * K2;
  std::vector<char*> copies(6);
  for (auto& c : copies) { hipMalloc(&c, bytes); hipMemset(c, 1, bytes); }
  u32x4* out; hipMalloc(&out, 1 << 24);
  printf("rows %d K %d: %.1f MB per launch, grid %d x 256 threads\n", rows, K, bytes / 1e6, rows / 128);
#define RUN(PW, NT) { float ms = run<PW, NT>(copies, rows, K2, out, 5); printf("PW=%4d nt=%d: %7.2f us  %7.1f GB/s\n", PW, NT, ms * 1e3, bytes / 1e6 / ms); }
#define RUNL(PW, NL) { float ms = run<PW, 1, NL>(copies, rows, K2, out, 5); printf("PW=%4d nt=1 loads/set=%2d (%3d KB in flight per CU, two sets): %7.2f us  %7.1f GB/s\n", PW, NL, NL * 4 * 2, ms * 1e3, bytes / 1e6 / ms); }
  RUNL(256, 8) RUNL(256, 16) RUNL(256, 32)
  RUN(64, 0) RUN(256, 0) RUN(512, 0) RUN(1024, 0)
  RUN(64, 1) RUN(256, 1) RUN(512, 1) RUN(1024, 1)
  return 0;
}
