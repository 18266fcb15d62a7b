// Decode MLP chain (batch 1): the streaming helpers and the gate/up -> down body shared by mlp_chain_kernel (pcy_gemv.hip) and the
// decode layer kernel (pcy_attn.hip).  See the comment above mlp_chain_kernel for the hand-over scheme.
#pragma once
#include "pcy_common.h"
#include "pcy_internal.h"
#include <type_traits>
#include "pcy_handover.h"

namespace {

constexpr int MC_NT = 512, MC_WV = 8, MC_UNB_D = 7;
constexpr int MC_LDS_IT0 = 4;      // the gate/up batch (k-iterations 4, 5 of a wave's first unit) the decode layer's projection workgroups keep in LDS

// One wave streams its units (RW weight rows each, row i of unit u at W + row_off(u, i)) against xs, UNB k-iterations of 512
// elements per batch, two batches in flight (wa / wb).  primed: the first two batches of (u0, it 0) are already in wa / wb.
// before_batch(it0) runs ahead of the arithmetic of every batch (the down stage waits there for the second half of its input).
// LDS_IT0 >= 0 and lds_batch != nullptr: the batch (u0, LDS_IT0) waits in LDS (this wave's own 16 KB, image of mc_lds_prefetch: register j of lane l at
// j * 1024 + l * 16) and is taken from there instead of from memory.
// KTAIL: K % 512 may be 256 (ffn 11008 = 21.5 x 512, Llama-2-7B): the last k-iteration is cut per lane, as gemv_stream_kernel cuts it.
// shift(u) (McNoShift: none): unit u walks its k-iterations rotated, it -> (it + shift(u)) % nit (K % 512 == 0) -- see PcyGemvArgs::krot.
struct McNoShift { __device__ __forceinline__ int operator()(int) const { return 0; } };
template <int RW, int UNB, int LDS_IT0 = -1, bool KTAIL = false, typename RowOff, typename Finish, typename Before, typename Shift = McNoShift>
__device__ __forceinline__ void mc_stream(const bf16_t* __restrict__ W, int K, const bf16_t* xs, int lane, int u0, int ustride, int uend,
                                          uint4 (&wa)[16], uint4 (&wb)[16], bool primed, RowOff row_off, Finish finish, Before before_batch,
                                          const char* lds_batch = nullptr, Shift shift = Shift()) {
  static_assert(RW * UNB <= 16, "batch size");
  constexpr bool ROT = !std::is_same<Shift, McNoShift>::value;
  static_assert(!(ROT && KTAIL), "rotated order: whole k-iterations only");
  const int nit = KTAIL ? (K + 511) >> 9 : K >> 9;
  auto rot = [&](int u, int it) __attribute__((always_inline)) { if (!ROT) return it; const int r = it + shift(u); return r >= nit ? r - nit : r; };
  auto issue = [&](int u, int it0, uint4 (&w)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int un = 0; un < UNB; ++un) {
      const int k = (rot(u, it0 + un) * 64 + lane) * 8;
      const bool ok = KTAIL ? k < K : (it0 + un) < nit;
#pragma unroll
      for (int i = 0; i < RW; ++i) w[un * RW + i] = ok ? ldg_nt(W + row_off(u, i) + k) : make_uint4(0, 0, 0, 0);
    }
  };
  float acc[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) acc[i] = 0.f;
  auto compute = [&](int u, int it0, const uint4 (&w)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int un = 0; un < UNB; ++un) {
      if (KTAIL ? ((it0 + un) * 64 + lane) * 8 < K : (it0 + un) < nit) {
        const uint4 xv = *reinterpret_cast<const uint4*>(xs + (rot(u, it0 + un) * 64 + lane) * 8);
#pragma unroll
        for (int i = 0; i < RW; ++i) acc[i] = dot8(w[un * RW + i], xv, acc[i]);
      }
    }
  };
  auto next_pos = [&](int cu, int cit, int& nu, int& nit_) __attribute__((always_inline)) { nit_ = cit + UNB; nu = cu; if (nit_ >= nit) { nit_ = 0; nu = cu + ustride; } };
  int u = u0, it0 = 0, u1, it1;
  bool have = u < uend;
  next_pos(u, 0, u1, it1);
  bool have1 = have && u1 < uend;
  if (!primed) {
    if (have) issue(u, 0, wa);
    if (have1) issue(u1, it1, wb);
  }
  if constexpr (LDS_IT0 >= 0) {
    // The first unit's third batch waits in LDS: a straight-line prologue (nothing of this in the loop below: a test there cost 5 spilled
    // VGPRs, and a spill reload is a VMEM operation whose wait drains every weight load in flight -- decode step 2.57 -> 3.15 ms).
    //   b0 (wa) -> request b3 into wa | b1 (wb) | b2: LDS -> wb, compute -> request the next unit's b0 into wb | loop from b3 (wa), next (wb)
    static_assert(LDS_IT0 < 0 || (UNB == 2 && LDS_IT0 == 4), "prologue written for 4 batches of 2 k-iterations per unit");
    if (lds_batch != nullptr && nit == 8 && have && u0 + ustride < uend) {
      before_batch(0);
      compute(u0, 0, wa);
      issue(u0, 6, wa);
      before_batch(2);
      compute(u0, 2, wb);
      // LDS-DMA copies are not covered by the compiler's wait insertion.  They are older than b3's 16 loads (and b1's before them); loads
      // return in order, so "at most 16 outstanding" means they have landed -- and b3 stays in flight.
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 16; ++j) wb[j] = *reinterpret_cast<const uint4*>(lds_batch + j * 1024 + lane * 16);
      before_batch(4);
      compute(u0, 4, wb);
      issue(u0 + ustride, 0, wb);
      u = u0; it0 = 6; u1 = u0 + ustride; it1 = 0; have = true; have1 = true;
    }
  }
#define PCY_MC_STEP(CUR)                                   \
  {                                                        \
    before_batch(it0);                                     \
    compute(u, it0, CUR);                                  \
    if (it0 + UNB >= nit) {                                \
      _Pragma("unroll") for (int i = 0; i < RW; ++i) acc[i] = wave_sum(acc[i]); \
      finish(u, acc);                                      \
      _Pragma("unroll") for (int i = 0; i < RW; ++i) acc[i] = 0.f; \
    }                                                      \
    int u2, it2;                                           \
    next_pos(u1, it1, u2, it2);                            \
    const bool have2 = have1 && u2 < uend;                 \
    if (have2) issue(u2, it2, CUR);                        \
    u = u1; it0 = it1; have = have1;                       \
    u1 = u2; it1 = it2; have1 = have2;                     \
  }
  while (have) {
    PCY_MC_STEP(wa)
    if (!have) break;
    PCY_MC_STEP(wb)
  }
#undef PCY_MC_STEP
}
// request the first two batches of unit u0 (what mc_stream(primed = true) expects to find)
// WHICH: 1 = the first batch only, 2 = the second only, 3 = both
template <int RW, int UNB, int WHICH = 3, typename RowOff, typename Shift = McNoShift>
__device__ __forceinline__ void mc_prime(const bf16_t* __restrict__ W, int K, int lane, int u0, int ustride, int uend, uint4 (&wa)[16], uint4 (&wb)[16],
                                         RowOff row_off, Shift shift = Shift()) {
  const int nit = K >> 9;
  constexpr bool ROT = !std::is_same<Shift, McNoShift>::value;
  auto rot = [&](int u, int it) __attribute__((always_inline)) { if (!ROT) return it; const int r = it + shift(u); return r >= nit ? r - nit : r; };
  auto issue = [&](int u, int it0, uint4 (&w)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int un = 0; un < UNB; ++un) {
      const int k = (rot(u, it0 + un) * 64 + lane) * 8;
      const bool ok = (it0 + un) < nit;
#pragma unroll
      for (int i = 0; i < RW; ++i) w[un * RW + i] = ok ? ldg_nt(W + row_off(u, i) + k) : make_uint4(0, 0, 0, 0);
    }
  };
  if (u0 < uend) {
    if (WHICH & 1) issue(u0, 0, wa);
    int u1 = u0, it1 = UNB;
    if (it1 >= nit) { it1 = 0; u1 = u0 + ustride; }
    if ((WHICH & 2) && u1 < uend) issue(u1, it1, wb);
  }
}


// gate/up row of feature 4u + (i & 3): gate rows i < 4, up rows i >= 4 (16-row gate/up interleave of the packed matrix)
struct McRowG {
  int F, d;
  __device__ __forceinline__ size_t operator()(int u, int i) const {
    const int f = u * 4 + (i & 3);
    const int fc = f < F ? f : F - 1;
    return (size_t)((fc >> 4) * 32 + (fc & 15) + (i >= 4 ? 16 : 0)) * d;
  }
};
// rotated k order of the gate/up unit u (4 features = output rows 4u ..: (r / 4) % nit, PcyGemvArgs::krot)
struct McShiftG {
  int nit;
  __device__ __forceinline__ int operator()(int u) const { return u % nit; }
};
__device__ __forceinline__ int mc_rot(int it, int sh, int nit) { const int r = it + sh; return r >= nit ? r - nit : r; }
// LDS-DMA of the batch (k-iterations it0, it0 + 1) of this wave's first gate/up unit into the wave's own 16 KB at dst (register j of lane l
// at j * 1024 + l * 16): what mc_stream<.., LDS_IT0 = it0>(lds_batch = dst) expects.  Issued by the projection workgroups of the decode layer
// while the attention runs: their registers are full (Wo rows + the first batch), their LDS is idle.
__device__ __forceinline__ void mc_lds_prefetch(const PcyMlpChainArgs& a, int lane, int gidx, int it0, char* dst) {
  typedef __attribute__((address_space(3))) void* mc_lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* mc_gptr_t;
  const McRowG row_g{a.F, a.d};
#pragma unroll
  for (int un = 0; un < 2; ++un)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((mc_gptr_t)(a.wgu + row_g(gidx, i) + (mc_rot(it0 + un, gidx % (a.d >> 9), a.d >> 9) * 64 + lane) * 8),
                                       (mc_lds_ptr_t)(dst + (un * 8 + i) * 1024), 16, 0, 2);
}
// first batch (k-iterations 0, 1) of this wave's first gate/up unit -> wa, and with `both` the second (2, 3) -> wb: what
// mc_mlp_body(primed = 1 / 3) expects to find
__device__ __forceinline__ void mc_prime_gate_up(const PcyMlpChainArgs& a, int lane, int gidx, uint4 (&wa)[16], uint4 (&wb)[16], bool both) {
  const McRowG row_g{a.F, a.d};
  const int units_g = (a.F + 3) / 4;
  if (gidx < units_g) {
#pragma unroll
    for (int un = 0; un < 2; ++un)
#pragma unroll
      for (int i = 0; i < 8; ++i) wa[un * 8 + i] = ldg_nt(a.wgu + row_g(gidx, i) + (mc_rot(un, gidx % (a.d >> 9), a.d >> 9) * 64 + lane) * 8);
    if (both) {
#pragma unroll
      for (int un = 0; un < 2; ++un)
#pragma unroll
        for (int i = 0; i < 8; ++i) wb[un * 8 + i] = ldg_nt(a.wgu + row_g(gidx, i) + (mc_rot(2 + un, gidx % (a.d >> 9), a.d >> 9) * 64 + lane) * 8);
    }
  }
}

// Stages 1 and 2 of the MLP chain for one of G workgroups (index wg) of MC_NT threads: act = SwiGLU(RMSNorm(x) * ln2 . Wgu^T) handed over
// as tagged words, x_out = x + act . Wdown^T.  LDS: [d] normalised x | [F] act | [d] x (XLDS: already there) | red.
// XLDS: x is in LDS (fetched from a tagged vector by the caller) instead of global memory written before the launch.
// primed (workgroup-uniform): bit 0 / 1 = the first / second batch of the wave's first gate/up unit is already in wa / wb
// (mc_prime_gate_up).
// x_out_lines != nullptr: the result is handed over as tagged words, workgroup wg's 16 rows in the first half of line wg
// (mc_fetch_vector_lines), instead of being stored to a.x_out.
// pf_off > 0: the batch of k-iterations MC_LDS_IT0.. of each wave's first gate/up unit waits in LDS at smem + pf_off + wave * 16 KB
// (mc_lds_prefetch, issued by the caller), see mc_stream.
template <bool XLDS>
__device__ __forceinline__ void mc_mlp_body(const PcyMlpChainArgs& a, char* smem, int vthr_gu, uint32_t tag, int G, int wg, int primed,
                                            uint4 (&wa)[16], uint4 (&wb)[16], unsigned long long* tr, uint32_t* x_out_lines = nullptr,
                                            int pf_off = 0) {
  const int d = a.d, F = a.F;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* xa = xs + d;
  bf16_t* xr = xa + F;
  float* red = reinterpret_cast<float*>(xr + d);
  const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NW = G * MC_WV, gw = wg * MC_WV + wave;
  uint4 tq[4];
#define MC_T(i) if (tr && tid == 0) tr[i] = wall_clock64();
  const bf16_t* xin = XLDS ? xr : a.x;
  // down rows: one unit of two rows per wave (d == 2 * NW), MC_UNB_D k-iterations per batch
  const int units_d = d / 2, r0 = gw * 2;
  auto row_d = [&](int u, int i) __attribute__((always_inline)) -> size_t { return (size_t)(u * 2 + i) * F; };
  const int half = F >> 1;                                           // act words per half; waves 0..6 fetch 1024 words of it each
  // ---- stage 1: units of 4 features (8 weight rows), 7 waves per workgroup ----
  const int units_g = (F + 3) / 4, NWG7 = G * 7, gidx = wg * 7 + wave;
  const McRowG row_g{F, d};
  mc_rms_stage(xin, a.ln2, d, vthr_gu, a.rms_eps, a.rms_cast, xs, red, [&]() __attribute__((always_inline)) {
    if (wave < 7) {
      if (primed == 0) mc_prime<8, 2, 3>(a.wgu, d, lane, gidx, NWG7, units_g, wa, wb, row_g, McShiftG{d >> 9});
      else if (primed == 1) mc_prime<8, 2, 2>(a.wgu, d, lane, gidx, NWG7, units_g, wa, wb, row_g, McShiftG{d >> 9});
    }
  });
  if (wave < 7) {
    const char* lds_batch = (pf_off > 0 && gidx < units_g) ? smem + pf_off + wave * 16384 : nullptr;
    mc_stream<8, 2, MC_LDS_IT0>(a.wgu, d, xs, lane, gidx, NWG7, units_g, wa, wb, true, row_g, [&](int u, const float (&acc)[8]) __attribute__((always_inline)) {
      if (lane == 0) {
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float g = rbf(acc[i]), up = rbf(acc[i + 4]);
          o[i] = (tag << 16) | f2bf(rbf(silu_f(g)) * up);
        }
        st8_agent(a.act_tag + u * 4, o[0], o[1]);
        st8_agent(a.act_tag + u * 4 + 2, o[2], o[3]);
      }
    }, [](int) __attribute__((always_inline)) {}, lds_batch, McShiftG{d >> 9});
    MC_T(1)
    // ---- stage 2 begins for this wave: first half of act, then (once it is there) the first two batches of its down rows ----
    mc_fetch_issue<4>(a.act_tag, wave * 1024, lane, tq);
  }
  // (the down rows are primed BEHIND the arrival of the first act half: requested in front, their 32 KB per wave -- and every other
  // workgroup's -- were in the queues the hand-over loads and the producers' stores go through: 2.490 -> 2.478 ms per token)
  if (wave < 7) mc_fetch_finish<4>(a.act_tag, wave * 1024, lane, tag, xa, tq, a.err, 6u);
  mc_prime<2, MC_UNB_D>(a.wdown, F, lane, gw, NW, units_d, wa, wb, row_d);
  if (wave < 7) mc_fetch_issue<4>(a.act_tag, half + wave * 1024, lane, tq);   // second half: checked nb/2 batches from now
  __syncthreads();
  MC_T(2)
  float acc[2] = {0.f, 0.f};
  const int it_half = (F >> 9) / 2;
  mc_stream<2, MC_UNB_D>(a.wdown, F, xa, lane, gw, NW, units_d, wa, wb, true, row_d,
                         [&](int u, const float (&acc2)[2]) __attribute__((always_inline)) { acc[0] = acc2[0]; acc[1] = acc2[1]; },
                         [&](int it0) __attribute__((always_inline)) {
                           if (it0 == it_half) {   // (workgroup-uniform: every wave walks the same batches of its one unit)
                             if (wave < 7) mc_fetch_finish<4>(a.act_tag, half + wave * 1024, lane, tag, xa, tq, a.err, 7u);
                             __syncthreads();
                             MC_T(3)
                           }
                         });
  // ---- x_out = x + act . Wdown^T ----
  if (lane == 0) {
    uint32_t o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v = rbf(acc[i]);
      v = rbf(v + bf2f(xin[r0 + i]));
      o[i] = f2bf(v);
    }
    if (x_out_lines) { red[2 * wave] = __uint_as_float(o[0]); red[2 * wave + 1] = __uint_as_float(o[1]); }
    else *reinterpret_cast<uint32_t*>(a.x_out + r0) = o[0] | (o[1] << 16);
  }
  if (x_out_lines) {   // the workgroup's 16 rows as ONE 64-byte store into its own line
    __syncthreads();
    if (wave == 0 && lane < 16)
      __hip_atomic_store(x_out_lines + wg * 32 + lane, (tag << 16) | __float_as_uint(red[lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#undef MC_T
}

}  // namespace
