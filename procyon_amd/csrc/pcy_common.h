// Shared device helpers for the ProCyon gfx950 engine (CDNA4, wave64).
//
// Numerics contract (DESIGN.md "Rounding points"): every kernel accumulates in fp32 and
// rounds to bf16 (round-to-nearest-even, as torch does) exactly where the reference's
// bf16 torch ops materialise a tensor.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16pair;

#define PCY_WAVE 64
#define PCY_BF16_MIN (-3.3895313892515355e38f)  // torch.finfo(torch.bfloat16).min

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round to nearest even; NaN stays NaN (c10::BFloat16 semantics).  gfx950 has the conversion in
// hardware (v_cvt_pk_bf16_f32): one instruction instead of the 6-op integer emulation.
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// round-trip: the value a bf16 tensor would hold
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pack_bf(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16pair));
}

// threadIdx.x through an optimisation barrier: in a kernel that runs the same body once per layer (decode_step_kernel) the
// compiler otherwise hoists every lane-derived address out of the layer loop and keeps it live (256 VGPRs + 86 spilled)
__device__ __forceinline__ int pcy_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == NT (multiple of 64); red = NT/64 floats of LDS
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}
// block-wide sum with a run-time wave count (<= 8)
__device__ __forceinline__ float block_sum_rt(float v, float* red, int nwaves) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nwaves; ++i) t += red[i];
  return t;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, red[i]);
  return t;
}

// torch.nn.functional.silu on a bf16 tensor: fp32 x / (1 + exp(-x)), one rounding
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// erf for the ESM GELU chain (bf16 in, bf16 out): odd rational minimax x.P(x^2)/Q(x^2) on [-4, 4] (the classic single-precision form used by
// vectorised math libraries), relative error 2.8e-7.  Rounded to bf16 it equals bf16(erf(x)) for every finite bf16 input
// (checked exhaustively on the host), at a third of libm erff's instruction count -- the ESM fc1 epilogue was 25 % of
// that GEMM's time.
__device__ __forceinline__ float erf_fast(float x) {
  x = fminf(fmaxf(x, -4.0f), 4.0f);
  const float x2 = x * x;
  float p = -2.72614225801306e-10f;
  p = fmaf(x2, p, 2.77068142495902e-08f);
  p = fmaf(x2, p, -2.10102402082508e-06f);
  p = fmaf(x2, p, -5.69250639462346e-05f);
  p = fmaf(x2, p, -7.34990630326855e-04f);
  p = fmaf(x2, p, -2.95459980854025e-03f);
  p = fmaf(x2, p, -1.60960333262415e-02f);
  p *= x;
  float q = -1.45660718464996e-05f;
  q = fmaf(x2, q, -2.13374055278905e-04f);
  q = fmaf(x2, q, -1.68282697438203e-03f);
  q = fmaf(x2, q, -7.37332916720468e-03f);
  q = fmaf(x2, q, -1.42647390514189e-02f);
  return p * __builtin_amdgcn_rcpf(q);
}
// nn.GELU() (erf form) in fp32
// (libm erff here: 1 + erf(x) cancels in the negative tail, where the approximation's 2.8e-7 would show; this epilogue
// only runs in the projector MLPs)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// fair-esm / HF-ESM gelu evaluated op by op on a bf16 tensor:
//   x * 0.5 * (1.0 + erf(x / sqrt(2)))  -> five bf16 tensors
__device__ __forceinline__ float gelu_esm_chain(float x /* already bf16-valued */) {
  float t1 = rbf(x * 0.5f);
  float t2 = rbf(x / 1.4142135623730951f);
  float t3 = rbf(erf_fast(t2));
  float t4 = rbf(1.0f + t3);
  return rbf(t1 * t4);
}

// 8 bf16 x 8 bf16 -> fp32 (v_dot2c_f32_bf16 x 4) and the 16-byte non-temporal (streaming) weight load
__device__ __forceinline__ float dot8(const uint4& w, const uint4& x, float acc) {
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair, w.x), __builtin_bit_cast(bf16pair, x.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair, w.y), __builtin_bit_cast(bf16pair, x.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair, w.z), __builtin_bit_cast(bf16pair, x.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16pair, w.w), __builtin_bit_cast(bf16pair, x.w), acc, false);
  return acc;
}

// Workgroup barrier for LDS traffic only.  __syncthreads() also drains vmcnt (its release fence covers global stores, and
// loads share the counter): a barrier inside a prefetching loop then waits for every load just issued (decode attention: 6.8 us
// from the position load to the first barrier; prefill attention: one memory round trip per key block) and the prefetch
// overlaps nothing.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// (The pointer is cast to the GLOBAL address space: a weight pointer that the kernel has itself loaded from memory -- the per-layer table of
// decode_step_kernel -- is otherwise a generic pointer, and hipcc emits flat_load for it.  A flat load counts in lgkmcnt as well as in vmcnt,
// so every wait for an LDS read also waited for all weight loads in flight.)
typedef const __attribute__((address_space(1))) u32x4_t* pcy_gptr_u32x4;
__device__ __forceinline__ uint4 ldg_nt(const bf16_t* p) {
  const u32x4_t v = __builtin_nontemporal_load((pcy_gptr_u32x4)(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
// plain 16-byte load from global memory (same reason)
__device__ __forceinline__ uint4 ldg16(const void* p) {
  const u32x4_t v = *(pcy_gptr_u32x4)(p);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// Vectors handed from one workgroup to another INSIDE a launch (CDNA4 guide, Guideline 16 R1): written through (sc1 stores), the
// writing waves drain (s_waitcnt vmcnt(0)), ONE flag store per workgroup; the reader polls relaxed and then loads with
// L1-bypassing (agent-scope) loads.  Placement-independent.
__device__ __forceinline__ uint4 ld16_agent(const void* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}
__device__ __forceinline__ void st_bf16_agent(bf16_t* p, bf16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Launch-time caches are PER DEVICE (a process may drive several: hipFuncSetAttribute and the occupancy answers belong to the device that
// is current when they are made).  Host code only.
constexpr int PCY_MAX_DEV = 16;
inline int pcy_cur_dev() {
  int d = 0;
  (void)hipGetDevice(&d);
  return (d >= 0 && d < PCY_MAX_DEV) ? d : 0;
}
struct PcyLdsAttr {   // the largest dynamic-LDS size a kernel has been configured for on each device
  size_t configured[PCY_MAX_DEV] = {};
  template <typename K>
  bool ensure(K kernel, size_t smem, size_t above = 65536) {   // false: the device refused the size
    const int d = pcy_cur_dev();
    if (smem > above && smem > configured[d]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
        (void)hipGetLastError();
        return false;
      }
      configured[d] = smem;
    }
    return true;
  }
};
struct PcyResidentCache {   // "all workgroups of this launch are resident at once" per device and LDS size
  int ok[PCY_MAX_DEV];
  size_t smem[PCY_MAX_DEV];
  PcyResidentCache() { for (int i = 0; i < PCY_MAX_DEV; ++i) { ok[i] = -1; smem[i] = 0; } }
  template <typename Q>
  bool check(size_t s, Q query) {
    const int d = pcy_cur_dev();
    if (ok[d] < 0 || smem[d] != s) { ok[d] = query() ? 1 : 0; smem[d] = s; }
    return ok[d] == 1;
  }
};

// Watchdog of the in-launch cross-workgroup waits (hand-overs, flags): give up after `limit` polls and record `code` in the
// context's sticky error word -- and give up at once when ANY wait of the launch has already done so (otherwise a launch whose
// workgroups are not all resident, e.g. beside a competing kernel, would sit out the full limit at every one of its ~200
// waits).  The host reads the word at the next ABI call (pinned, host-visible memory) and fails that call.
__device__ __forceinline__ bool pcy_wait_give_up(unsigned& spins, unsigned limit, unsigned* err, unsigned code, int lane) {
  ++spins;
  if (err && (spins & 255u) == 0u &&
      __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != 0) return true;
  if (spins > limit) {
    if (lane == 0 && err) __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return true;
  }
  return false;
}

// Epilogue selectors shared by the GEMM (prefill / encoder) and GEMV (decode) kernels.
enum PcyEpi : int {
  EPI_STORE = 0,      // y = bf16(acc [+ bias])
  EPI_RESID = 1,      // y = bf16( bf16(acc [+ bias]) + residual )
  EPI_GELU_ERF = 2,   // y = bf16( gelu_erf( bf16(acc + bias) ) )            (create_mlp, nn.GELU)
  EPI_GELU_ESM = 3,   // y = esm op-by-op gelu chain of bf16(acc + bias)     (ESM fc1)
  EPI_SWIGLU = 4,     // y = bf16( bf16(silu(bf16(acc_gate))) * bf16(acc_up) ) (Llama MLP)
};
