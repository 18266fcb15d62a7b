// In-launch hand-over of bf16 vectors between workgroups as {tag : bf16} words, and the RMSNorm prologue that reproduces the
// summation order of the stand-alone GEMV launch.  Shared by the decode MLP chain (pcy_gemv.hip) and the decode attention block
// (pcy_attn.hip).
//
// A producer stores every element as ONE 32-bit word {tag : bf16 value}, tag = low half of a device counter that advances once
// per decode step, written through to memory (agent scope) and never waited for.  A consumer loads the words it wants with
// L1-bypassing loads and accepts them when every tag is the current one, else asks again: one memory round trip when the data is
// there, no store drain, no second trip for a flag.  Each vector lives in a per-layer slot that only these launches write, so
// a word with the current tag can only have been written in THIS decode step.
#pragma once
#include "pcy_common.h"

namespace {

__device__ __forceinline__ void st8_agent(void* p, uint32_t lo, uint32_t hi) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)lo | ((unsigned long long)hi << 32), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// This wave's share of a tagged vector -> LDS as plain bf16: words [w0 + (j*64 + lane)*4, +4), j < NV.  `pre` holds loads issued
// earlier by mc_fetch_issue (checked first); asks again until every tag matches.
template <int NV>
__device__ __forceinline__ void mc_fetch_issue(const uint32_t* src, int w0, int lane, uint4 (&pre)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) pre[j] = ld16_agent(src + w0 + (j * 64 + lane) * 4);
}
template <int NV>
__device__ __forceinline__ void mc_fetch_finish(const uint32_t* src, int w0, int lane, uint32_t tag, bf16_t* dst, uint4 (&pre)[NV], unsigned* err,
                                                unsigned code) {
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      ok = ok && (pre[j].x >> 16) == tag && (pre[j].y >> 16) == tag && (pre[j].z >> 16) == tag && (pre[j].w >> 16) == tag;
    if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
    if (pcy_wait_give_up(spins, 1u << 19, err, code, lane)) break;
    __builtin_amdgcn_s_sleep(16);
    mc_fetch_issue<NV>(src, w0, lane, pre);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j)
    *reinterpret_cast<uint2*>(dst + w0 + (j * 64 + lane) * 4) =
        make_uint2((pre[j].x & 0xffffu) | (pre[j].y << 16), (pre[j].z & 0xffffu) | (pre[j].w << 16));
}

// A whole tagged vector of 8 x 512 words -> LDS: one wave watches a 1 KB sample until it is current (a workgroup then asks for
// 1 KB per round trip instead of 16 KB while it waits), then every wave takes its 512 words.  All 512 threads; ends with a barrier.
struct McNoHook { __device__ __forceinline__ void operator()() const {} };
// after_issue: runs once per wave right behind the loads that fetch the vector (weight prefetches belong THERE: a CU's loads return in
// order, and a hand-over load queued behind 16-32 KB of weights per wave waits for all of them)
template <typename AfterIssue = McNoHook>
__device__ __forceinline__ void mc_fetch_vector(const uint32_t* src, int n, int watch_wave, uint32_t tag, bf16_t* dst, unsigned* err, unsigned code,
                                                AfterIssue after_issue = AfterIssue()) {
  const int tid_ = pcy_tid(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  if (wave == watch_wave) {
    unsigned spins = 0;
    for (;;) {
      const uint4 v = ld16_agent(src + n - 256 + lane * 4);
      const bool ok = (v.x >> 16) == tag && (v.y >> 16) == tag && (v.z >> 16) == tag && (v.w >> 16) == tag;
      if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
      if (pcy_wait_give_up(spins, 1u << 19, err, code, lane)) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  uint4 t[2];
  mc_fetch_issue<2>(src, wave * 512, lane, t);
  after_issue();
  mc_fetch_finish<2>(src, wave * 512, lane, tag, dst, t, err, code);
  __syncthreads();
}

// The same for a vector stored SPARSELY: producer workgroup b owns the 128-byte line b of `src` and fills its first 16 words
// (element 16 b + i at word 32 b + i), so that no line has two writers.  n = 16 x (number of producing workgroups) elements,
// n % 512 == 0 and n <= 4096; dst [n] bf16.
template <typename AfterIssue = McNoHook>
__device__ __forceinline__ void mc_fetch_vector_lines(const uint32_t* src, int n, int watch_wave, uint32_t tag, bf16_t* dst, unsigned* err, unsigned code,
                                                      AfterIssue after_issue = AfterIssue()) {
  const int tid_ = pcy_tid(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int nlines = n >> 4;
  auto word_of = [&](int line, int piece) __attribute__((always_inline)) { return line * 32 + piece * 4; };
  if (wave == watch_wave) {   // the last 16 lines
    unsigned spins = 0;
    for (;;) {
      const uint4 v = ld16_agent(src + word_of(nlines - 16 + (lane >> 2), lane & 3));
      const bool ok = (v.x >> 16) == tag && (v.y >> 16) == tag && (v.z >> 16) == tag && (v.w >> 16) == tag;
      if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
      if (pcy_wait_give_up(spins, 1u << 19, err, code, lane)) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  if (!(wave * 512 < n)) after_issue();
  if (wave * 512 < n) {   // this wave's 512 elements = 32 lines: two loads of 16 lines x 4 pieces
    uint4 t[2];
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        t[j] = ld16_agent(src + word_of(wave * 32 + j * 16 + (lane >> 2), lane & 3));
        ok = ok && (t[j].x >> 16) == tag && (t[j].y >> 16) == tag && (t[j].z >> 16) == tag && (t[j].w >> 16) == tag;
      }
      if (spins == 0) after_issue();
      if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
      if (pcy_wait_give_up(spins, 1u << 19, err, code, lane)) break;
      __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *reinterpret_cast<uint2*>(dst + (wave * 32 + j * 16 + (lane >> 2)) * 16 + (lane & 3) * 4) =
          make_uint2((t[j].x & 0xffffu) | (t[j].y << 16), (t[j].z & 0xffffu) | (t[j].w << 16));
  }
  __syncthreads();
}

// xs[0..K) = bf16( RMSNorm(x) * w ) with the statistic summed like gemv_stream_kernel launched with `vthr` threads.  x: global
// (written before this launch) or LDS.  K <= 8 * MC_NT.  All threads; ends with a barrier.
template <typename AfterLoads>
__device__ __forceinline__ void mc_rms_stage(const bf16_t* x, const bf16_t* __restrict__ w, int K, int vthr, float eps, int cast,
                                             bf16_t* xs, float* red, AfterLoads after_loads) {
  const int tid = pcy_tid();
  auto ldx = [&](int k) __attribute__((always_inline)) { return *reinterpret_cast<const uint4*>(x + k); };
  // vector loads return in order: x (a few KiB, the head of the dependent chain) is requested BEFORE any weight batch
  uint4 xr[4], xv = make_uint4(0, 0, 0, 0), g = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = (tid + i * vthr) * 8;
    xr[i] = (tid < vthr && k < K) ? ldx(k) : make_uint4(0, 0, 0, 0);
  }
  const int ks = tid * 8;
  if (ks < K) { xv = ldx(ks); g = *reinterpret_cast<const uint4*>(w + ks); }
  after_loads();
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = (tid + i * vthr) * 8;
    if (tid < vthr && k < K) {
      const uint32_t w4[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float f0 = lo_bf(w4[j]), f1 = hi_bf(w4[j]); ss += f0 * f0 + f1 * f1; }
    }
  }
  ss = block_sum_rt(ss, red, vthr >> 6);
  const float rs = rsqrtf(ss / (float)K + eps);
  if (ks < K) {
    const uint32_t xin[4] = {xv.x, xv.y, xv.z, xv.w}, gin[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x0 = lo_bf(xin[j]) * rs, x1 = hi_bf(xin[j]) * rs;
      if (cast == 0) { x0 = rbf(x0); x1 = rbf(x1); }
      o[j] = pack_bf(lo_bf(gin[j]) * x0, hi_bf(gin[j]) * x1);
    }
    *reinterpret_cast<uint4*>(xs + ks) = make_uint4(o[0], o[1], o[2], o[3]);
  }
  __syncthreads();
}


}  // namespace
