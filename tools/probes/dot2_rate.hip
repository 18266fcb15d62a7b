// VALU issue rate of v_dot2c_f32_bf16 against v_fma_f32 / v_pk_fma_f32 on gfx950 (what bounds the small-batch decode step's arithmetic):
// 256 workgroups x 512 threads (two waves per SIMD, as the decode step), 16 independent accumulators per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dot2_rate.hip -o /tmp/dot2_rate && /tmp/dot2_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16pair;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, unsigned seed) {
  float acc[16];
  f32x2 acc2[16];
  unsigned w = seed + threadIdx.x, x = seed * 3 + threadIdx.x;
  for (int i = 0; i < 16; ++i) { acc[i] = (float)i; acc2[i] = (f32x2){(float)i, 1.0f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair, w), __builtin_bit_cast(bf16pair, x), acc[i], false);
        if (MODE == 1) acc[i] = __builtin_fmaf(__uint_as_float(w), __uint_as_float(x), acc[i]);
        if (MODE == 2) { f32x2 a = {__uint_as_float(w), __uint_as_float(x)}; acc2[i] = __builtin_elementwise_fma(a, a, acc2[i]); }
        asm volatile("" : "+v"(w), "+v"(x));
      }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i][0] + acc2[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, float* out) {
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate_kernel<MODE><<<256, 512>>>(out, 16, 1);
  hipEventRecord(e0);
  rate_kernel<MODE><<<256, 512>>>(out, iters, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * 64 * 2;          // 64 per iteration and wave, two waves per SIMD
  printf("%-16s %8.3f ms  %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0>("v_dot2_f32_bf16", out);
  run<1>("v_fma_f32", out);
  run<2>("v_pk_fma_f32", out);
  return 0;
}
