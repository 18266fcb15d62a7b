// How fast can all 256 workgroups read the SAME freshly written hand-over vector (the act / x vectors of the fused decode step)?
// A writer launch stores `words` tagged words with sc1 dword stores from all CUs; a reader launch has every workgroup (448 threads) read all
// of them with one of several load flavours.  In-kernel stamps: first start -> last end.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/bcast_read.hip -o tools/probes/bcast_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__global__ void write_k(uint32_t* x, int words, uint32_t tag) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x)
    __hip_atomic_store(x + i, (tag << 16) | (i & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE>
__global__ __launch_bounds__(448) void read_k(const uint32_t* x, int words, unsigned long long* stamps, uint32_t* sink, uint32_t tag) {
  const unsigned long long t0 = wall_clock64();
  uint32_t acc = 0, bad = 0;
  for (int i0 = threadIdx.x * 4; i0 < words; i0 += 448 * 4 * 4) {
    u32x4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 448 * 4;
      const uint32_t* p = x + (i < words ? i : 0);
      if (MODE == 0) {
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
        const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[u] = (u32x4_t){(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
      } else if (MODE == 1) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v[u]) : "v"(p) : "memory");
      } else if (MODE == 2) {
        v[u] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_t*>((const __attribute__((address_space(1))) uint32_t*)p);
      } else if (MODE == 3) {
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v[u]) : "v"(p) : "memory");
      } else {
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(v[u]) : "v"(p) : "memory");
      }
    }
    if (MODE == 1 || MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 1 || MODE >= 3) asm volatile("" : "+v"(v[u]));
      if (i0 + u * 448 * 4 < words) {
        acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
        bad += ((v[u][0] >> 16) != tag) + ((v[u][3] >> 16) != tag);
      }
    }
  }
  sink[blockIdx.x * 448 + threadIdx.x] = acc + (bad << 20);
  __syncthreads();
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t0; stamps[blockIdx.x * 2 + 1] = wall_clock64(); }
  if (bad) atomicAdd(sink + 256 * 448, 1u);
}
template <int MODE> void run(const char* name, uint32_t* x, int words, unsigned long long* st, uint32_t* sink) {
  std::vector<double> us;
  unsigned stale = 0;
  for (int rep = 0; rep < 12; ++rep) {
    hipMemset(sink + 256 * 448, 0, 4);
    write_k<<<256, 256>>>(x, words, rep + 1);
    read_k<MODE><<<256, 448>>>(x, words, st, sink, rep + 1);
    hipDeviceSynchronize();
    unsigned long long h[512]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    unsigned s; hipMemcpy(&s, sink + 256 * 448, 4, hipMemcpyDeviceToHost); stale += s;
    unsigned long long a = ~0ull, b = 0; for (int i = 0; i < 256; ++i) { a = std::min(a, h[2 * i]); b = std::max(b, h[2 * i + 1]); }
    if (rep >= 2) us.push_back((b - a) / 100.0);
  }
  std::sort(us.begin(), us.end());
  const double med = us[us.size() / 2];
  printf("  %-28s %7.2f us (min %6.2f)  = %6.2f TB/s delivered to the CUs, stale threads %u\n", name, med, us[0], 256.0 * words * 4 / med / 1e6, stale);
}
int main() {
  uint32_t *x, *sink; unsigned long long* st;
  hipMalloc(&x, 8 * 14336 * 4); hipMalloc(&sink, (256 * 448 + 16) * 4); hipMalloc(&st, 512 * 8);
  for (int words : {4 * 3584, 4 * 14336, 8 * 14336}) {
    printf("%d words (%d KB) read by every workgroup:\n", words, words * 4 / 1024);
    run<0>("2 x 8-B atomic sc1 (now)", x, words, st, sink);
    run<1>("16-B sc1", x, words, st, sink);
    run<3>("16-B sc0 sc1", x, words, st, sink);
    run<2>("16-B plain", x, words, st, sink);
    run<4>("16-B nt", x, words, st, sink);
  }
  return 0;
}
