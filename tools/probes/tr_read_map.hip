// Lane map of ds_read_b64_tr_b16 (gfx950): each lane supplies the LDS address of 4 contiguous 16-bit elements; what does lane i get?
// Hypothesis (used by attn_fast64_kernel<VROW>): within a 16-lane group, out[lane i][j] = element (i & 3) of the 4 elements fetched
// by lane 4 j + (i >> 2) -- i.e. the group reads a [4][16] row-major block (lane L: row L >> 2, columns 4 (L & 3) ..) and lane i
// receives column i.   hipcc --offload-arch=gfx950 -O2 tr_read_map.hip -o tr_read_map && ./tr_read_map
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(const int* lane_addr, short* out) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + lane_addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  int h_addr[64]; short h_out[256];
  int* d_addr; short* d_out;
  hipMalloc(&d_addr, sizeof h_addr); hipMalloc(&d_out, sizeof h_out);
  int bad_total = 0;
  for (int variant = 0; variant < 2; ++variant) {
    // variant 0: dense [4][16] blocks, one per group; variant 1: rows of 64 elements (128-byte row stride), group g at columns 16 g, rows 8 (g >> 1) + ..
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, L = l & 15;
      h_addr[l] = variant == 0 ? g * 64 + L * 4 : ((g >> 1) * 8 + (L >> 2)) * 64 + 16 * (g & 1) + 4 * (L & 3) + 1024;
    }
    hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, i = l & 15;
      for (int j = 0; j < 4; ++j) {
        const int src_lane = g * 16 + 4 * j + (i >> 2);
        const int expect = h_addr[src_lane] + (i & 3);
        if (h_out[l * 4 + j] != (short)expect) ++bad;
      }
    }
    printf("variant %d: %d mismatches vs the hypothesis\n", variant, bad);
    bad_total += bad;
    if (bad) for (int l = 0; l < 64; ++l) printf("lane %2d addr %4d -> %5d %5d %5d %5d\n", l, h_addr[l], h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
  }
  return bad_total != 0;
}
