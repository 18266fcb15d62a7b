// Batch-1 decode step of the multi-head geometry (ProCyon-Split = Llama-2-7B: 32 query heads = 32 kv heads of 128, d 4096, ffn 11008):
// every decoder layer of a step in ONE launch, as decode_step_kernel (pcy_attn.hip) is for Llama-3-8B.  Reference: the decoder loop of
// /root/reference/procyon/model/pmc_llama.py:571-588 under generate() (model_unified.py:859-915) with the released ProCyon-Split
// checkpoint (/root/reference/README.md:50-51).
//
// What differs from the grouped-query geometry and why this is its own body:
//   * Wqkv has 12288 rows (q, k and v all 4096 wide): 8 rows per projection wave instead of 4 -- two units of 4 rows streamed through two
//     register batches (mc_stream), the first unit requested around the fetch of x;
//   * 32 kv heads, one query head each: an attention unit = (kv head, 64 output columns), 2 x 32 = 64 units on workgroups [0, 64) -- the
//     head's keys split between its two slices above PCY_AO_XMIN keys (the score exchange of attn_dec_body);
//   * ffn = 11008 = 21.5 x 512: the gate/up stage deals feature PAIRS (4 weight rows) to all 2048 waves, 5504 pairs in three rounds, the
//     1408 pairs of the last one wave-major so that every CU keeps 5-6 waves streaming; the down projection starts on the first 7168 act
//     words (14 k-iterations, as on Llama-3) and its last k-iteration is half a one (KTAIL);
//   * every projection walks its k-iterations in the rotated order of PcyGemvArgs::krot (K = 4096: 8 KB row stride, see there);
//   * barriers inside the layer are lds_barrier (no vmcnt drain): a __syncthreads() behind a weight request waits for the weights.
// Per-row arithmetic, accumulation order and rounding points are those of gemv_stream_kernel with krot (RMSNorm statistics summed with the
// stand-alone launches' thread counts) and of attn_dec_kernel<128, 1, 64>: bit-identical to the launch-per-stage step with 64-column
// attention workgroups (PCY_DISABLE=decode_step,decode_layer; tests/test_gpu_round6.py).  In-kernel stamps (tools/bench_decode_mha.py,
// PCY_MC_TRACE=1), one layer at t ~ 600: x arrives at 11 us (48 MB of Wqkv requested at 3), qkv rows published at 20, attention 22 -> 30,
// o rows stored at 34, gate/up from 39, down from ~63, end at 76-78 us: 417 MB at 5.4 TB/s.
#include <stdlib.h>
#include "pcy_internal.h"
#include "pcy_handover.h"
#include "pcy_mlp_chain.h"
#include "pcy_attn_dec.h"

#ifndef MH_KROT_MASK
#define MH_KROT_MASK 7   // projections that walk k rotated (PcyGemvArgs::krot): 1 qkv, 2 o, 4 gate/up; anything but 7 is a measurement build (no twin)
#endif

namespace {

constexpr int MH_DH = 128, MH_DS = 64, MH_HKV = 32, MH_NATTN = (MH_DH / MH_DS) * MH_HKV;   // 64 attention workgroups
constexpr int MH_NW = 256 * 8;       // waves of the launch: the gate/up stage deals PAIRS of features (4 weight rows) to all of them, three rounds
constexpr int MH_D = 4096;           // hidden size; LDS of every workgroup: [d] normalised x / attention output | [d] x | red [128] | act [F]
constexpr int MH_OFF_X = MH_D * 2, MH_OFF_RED = 2 * MH_D * 2, MH_OFF_ACT = MH_OFF_RED + 512;
constexpr int MH_HALF = 7168;        // act words the down rows start on = 14 k-iterations (pairs [0, 3584): rounds one and two, mostly)

// weight row i (0, 1: gate, 2, 3: up) of feature pair h (16-row gate/up interleave of the packed matrix)
struct MhRowPair {
  int F, d;
  __device__ __forceinline__ size_t operator()(int h, int i) const {
    const int f = h * 2 + (i & 1);
    const int fc = f < F ? f : F - 1;
    return (size_t)((fc >> 4) * 32 + (fc & 15) + (i >= 2 ? 16 : 0)) * d;
  }
};
// this wave's share of a tagged vector, `nj` (wave-uniform, <= NV) loads of 256 words: mc_fetch_issue / mc_fetch_finish with a ragged end
template <int NV>
__device__ __forceinline__ void mh_fetch_issue(const uint32_t* src, int w0, int lane, uint4 (&pre)[NV], int nj) {
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (j < nj) pre[j] = ld16_agent(src + w0 + (j * 64 + lane) * 4);
}
template <int NV>
__device__ __forceinline__ void mh_fetch_finish(const uint32_t* src, int w0, int lane, uint32_t tag, bf16_t* dst, uint4 (&pre)[NV], int nj, unsigned* err,
                                                unsigned code) {
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (j < nj) ok = ok && (pre[j].x >> 16) == tag && (pre[j].y >> 16) == tag && (pre[j].z >> 16) == tag && (pre[j].w >> 16) == tag;
    if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
    if (pcy_wait_give_up(spins, 1u << 19, err, code, lane)) break;
    __builtin_amdgcn_s_sleep(16);
    mh_fetch_issue<NV>(src, w0, lane, pre, nj);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (j < nj)
      *reinterpret_cast<uint2*>(dst + w0 + (j * 64 + lane) * 4) =
          make_uint2((pre[j].x & 0xffffu) | (pre[j].y << 16), (pre[j].z & 0xffffu) | (pre[j].w << 16));
}

// mc_fetch_vector (pcy_handover.h) with barriers that leave the weight requests in flight alone (lds_barrier: no vmcnt drain)
__device__ __forceinline__ void mh_fetch_vector(const uint32_t* src, int n, int watch_wave, uint32_t tag, bf16_t* dst, unsigned* err, unsigned code) {
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
  lds_barrier();
  uint4 t[2];
  mc_fetch_issue<2>(src, wave * 512, lane, t);
  mc_fetch_finish<2>(src, wave * 512, lane, tag, dst, t, err, code);
  lds_barrier();
}

// first (WHICH & 1) / second (WHICH & 2) half (k-iterations 0-3 / 4-7 in the rotated order) of the 4 rows of feature pair gw (round one)
// -> wa / wb: what mh_mlp_body(primed) expects
template <int WHICH>
__device__ __forceinline__ void mh_prime_gate_up(const PcyMlpChainArgs& a, int lane, int gw, uint4 (&wa)[16], uint4 (&wb)[16]) {
  const int nit_g = a.d >> 9;
  mc_prime<4, 4, WHICH>(a.wgu, a.d, lane, gw, 1, a.F / 2, wa, wb, MhRowPair{a.F, a.d}, [&](int h) __attribute__((always_inline)) { return (MH_KROT_MASK & 4) ? (h >> 1) % nit_g : 0; });
}

// mc_rms_stage (pcy_handover.h) for x in LDS, with barriers that leave the weight requests in flight alone: xs[0..K) = bf16(RMSNorm(x) * w),
// the statistic summed like gemv_stream_kernel launched with `vthr` threads.  K <= 8 * 512.  All threads; ends with a barrier.
__device__ __forceinline__ void mh_rms_stage(const bf16_t* x, const bf16_t* __restrict__ w, int K, int vthr, float eps, int cast, bf16_t* xs, float* red) {
  const int tid = pcy_tid();
  const int ks = tid * 8;
  uint4 g = make_uint4(0, 0, 0, 0);
  if (ks < K) g = ldg16(w + ks);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = (tid + i * vthr) * 8;
    if (tid < vthr && k < K) {
      const uint4 xv = *reinterpret_cast<const uint4*>(x + k);
      const uint32_t w4[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float f0 = lo_bf(w4[j]), f1 = hi_bf(w4[j]); ss += f0 * f0 + f1 * f1; }
    }
  }
  {   // block_sum_rt(ss, red, vthr >> 6)
    ss = wave_sum(ss);
    lds_barrier();
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    lds_barrier();
    float t = 0.f;
    for (int i = 0; i < (vthr >> 6); ++i) t += red[i];
    ss = t;
  }
  const float rs = rsqrtf(ss / (float)K + eps);
  if (ks < K) {
    const uint4 xv = *reinterpret_cast<const uint4*>(x + ks);
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
  lds_barrier();
}

// Stages 1 and 2 of the MLP for one of 256 workgroups (mc_mlp_body of pcy_mlp_chain.h for 7168 < F <= 14336, F % 256 == 0, x in LDS):
//   act = SwiGLU(RMSNorm(x) * ln2 . Wgu^T) handed over as tagged words, x_out = x + act . Wdown^T.
// LDS: [d] normalised x | [d] x (already there) | red | [F] act.  primed: bit 0 / 1 = the first / second batch of the wave's first gate/up
// unit is already in wa / wb.  x_out_lines as mc_mlp_body.
__device__ __forceinline__ void mh_mlp_body(const PcyMlpChainArgs& a, char* smem, int vthr_gu, uint32_t tag, int wg, int primed, uint4 (&wa)[16],
                                            uint4 (&wb)[16], unsigned long long* tr, uint32_t* x_out_lines) {
  constexpr int G = 256;
  const int d = a.d, F = a.F;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* xr = reinterpret_cast<bf16_t*>(smem + MH_OFF_X);
  float* red = reinterpret_cast<float*>(smem + MH_OFF_RED);
  bf16_t* xa = reinterpret_cast<bf16_t*>(smem + MH_OFF_ACT);
  const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NW = G * MC_WV, gw = wg * MC_WV + wave;
  uint4 tq[4];
#define MH_T(i) if (tr && tid == 0) tr[i] = wall_clock64();
  const bf16_t* xin = xr;
  const int units_d = d / 2, r0 = gw * 2;
  auto row_d = [&](int u, int i) __attribute__((always_inline)) -> size_t { return (size_t)(u * 2 + i) * F; };
  // ---- stage 1: feature pairs (4 weight rows, two batches of 4 k-iterations), every wave: pair gw in round one, 2048 + gw in round two, and
  // the 1408 pairs of round three dealt wave-major (5-6 waves of every CU).  In units of 4 features on 7 waves (the Llama-3 body) 960 units
  // were left for a second round that ran on 3-4 waves per CU with 32 KB in flight each: 13.6 us for 63 MB (in-kernel stamps).
  const int n_pairs = F >> 1;
  const int h3 = 2 * MH_NW + wave * G + wg;
  const int nloc = h3 < n_pairs ? 3 : 2;
  const MhRowPair row_p{F, d};
  auto pair_of = [&](int u) __attribute__((always_inline)) { return u == 0 ? gw : (u == 1 ? MH_NW + gw : h3); };
  auto row_gl = [&](int u, int i) __attribute__((always_inline)) -> size_t { return row_p(pair_of(u), i); };
  const int nit_g = d >> 9;
  auto shift_g = [&](int u) __attribute__((always_inline)) { return (MH_KROT_MASK & 4) ? (pair_of(u) >> 1) % nit_g : 0; };   // rotated k order (PcyGemvArgs::krot)
  mh_rms_stage(xin, a.ln2, d, vthr_gu, a.rms_eps, a.rms_cast, xs, red);
  if (primed == 0) mc_prime<4, 4, 3>(a.wgu, d, lane, 0, 1, nloc, wa, wb, row_gl, shift_g);
  else if (primed == 1) mc_prime<4, 4, 2>(a.wgu, d, lane, 0, 1, nloc, wa, wb, row_gl, shift_g);
  // (Three register buffers here -- the qkv / Wo registers are idle in the MLP -- put 48 KB per wave in flight instead of 32: wave 0 was
  // through its pairs 10 us earlier and the act hand-over took 20 us: 100 MB in the chip's queues is 19 us of latency for every poll behind
  // them.  2.40 -> 2.54 ms per token; reverted.)
  MH_T(5)
  mc_stream<4, 4>(a.wgu, d, xs, lane, 0, 1, nloc, wa, wb, true, row_gl, [&](int u, const float (&acc)[4]) __attribute__((always_inline)) {
    if (u == 0) { MH_T(4) }
    if (lane == 0) {
      uint32_t o[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float g = rbf(acc[i]), up = rbf(acc[i + 2]);
        o[i] = (tag << 16) | f2bf(rbf(silu_f(g)) * up);
      }
      st8_agent(a.act_tag + pair_of(u) * 2, o[0], o[1]);
    }
  }, [](int) __attribute__((always_inline)) {}, nullptr, shift_g);
  MH_T(1)
  // ---- stage 2 begins for this wave: the first part of act, then (once it is there) the first two batches of its down rows ----
  if (wave < 7) mc_fetch_issue<4>(a.act_tag, wave * 1024, lane, tq);
  if (wave < 7) mc_fetch_finish<4>(a.act_tag, wave * 1024, lane, tag, xa, tq, a.err, 6u);
  mc_prime<2, MC_UNB_D>(a.wdown, F, lane, gw, NW, units_d, wa, wb, row_d);
  // the second part: words [MH_HALF, F), 1024 per wave in loads of 256
  int nj2 = (F - MH_HALF - wave * 1024) / 256;
  nj2 = wave < 7 ? (nj2 < 0 ? 0 : (nj2 > 4 ? 4 : nj2)) : 0;
  mh_fetch_issue<4>(a.act_tag, MH_HALF + wave * 1024, lane, tq, nj2);
  lds_barrier();   // (every wave's part of act is in LDS; the down rows just requested stay in flight)
  MH_T(2)
  float acc[2] = {0.f, 0.f};
  constexpr int it_half = MH_HALF / 512;   // 14 = two batches of MC_UNB_D
  static_assert(it_half % MC_UNB_D == 0, "the second part of act is checked in front of a batch");
  mc_stream<2, MC_UNB_D, -1, true>(a.wdown, F, xa, lane, gw, NW, units_d, wa, wb, true, row_d,
                                   [&](int u, const float (&acc2)[2]) __attribute__((always_inline)) { acc[0] = acc2[0]; acc[1] = acc2[1]; },
                                   [&](int it0) __attribute__((always_inline)) {
                                     if (it0 == it_half) {   // (workgroup-uniform: every wave walks the same batches of its one unit)
                                       mh_fetch_finish<4>(a.act_tag, MH_HALF + wave * 1024, lane, tag, xa, tq, nj2, a.err, 7u);
                                       lds_barrier();
                                       MH_T(3)
                                     }
                                   });
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
#undef MH_T
}

// One decoder layer for workgroup blockIdx.x of 256 (see the file comment; structure and hand-overs of decode_layer_body, pcy_attn.hip).
__device__ __forceinline__ void mh_layer_body(PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, unsigned xepoch, int vthr_qkv,
                                              size_t stage_off, int vthr_gu, char* smem, const uint32_t* x_in_lines, uint32_t* x_out_lines,
                                              unsigned long long* tr_base) {
  constexpr int DH = MH_DH, G = 1;
  const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t tag = *p.epoch & 0xffffu;
  unsigned long long* tr = tr_base ? tr_base + (size_t)blockIdx.x * 16 : nullptr;
#define MH_T(i) if (tr && tid == 0) tr[i] = wall_clock64();
  MH_T(0)
  if ((int)blockIdx.x < MH_NATTN) {
    constexpr int slices = DH / MH_DS;
    const int unit = blockIdx.x;
    const int bx = (unit / MH_HKV) % slices, kvh = unit % MH_HKV;
    bf16_t* stage = reinterpret_cast<bf16_t*>(smem + stage_off);   // [G + 2][DH]
    a.xepoch = xepoch;
    a.xerr = p.err;
    a.staged = stage; a.o_tag = p.ao_tag; a.tag = tag;
    const uint32_t* qt = p.qkv_tag;
    const int H = a.H, Hkv = a.Hkv;
    unsigned* err = p.err;
    auto hook = [=]() __attribute__((always_inline)) {
      constexpr int NV4 = (G + 2) * DH / 4;          // one uint4 of tagged words per thread
      const int seg = tid / (DH / 4), e4 = tid % (DH / 4);
      const int w0 = (seg < G ? (kvh * G + seg) : (seg == G ? H + kvh : H + Hkv + kvh)) * DH + e4 * 4;
      const bool mine = tid < NV4;
      if (wave * 64 < NV4) {
        uint4 v = make_uint4(0, 0, 0, 0);
        unsigned spins = 0;
        for (;;) {
          if (mine) v = ld16_agent(qt + w0);
          const bool ok = !mine || ((v.x >> 16) == tag && (v.y >> 16) == tag && (v.z >> 16) == tag && (v.w >> 16) == tag);
          if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
          if (pcy_wait_give_up(spins, 1u << 19, err, 9u, lane)) break;
          __builtin_amdgcn_s_sleep(8);
        }
        if (mine) *reinterpret_cast<uint2*>(stage + seg * DH + e4 * 4) = make_uint2((v.x & 0xffffu) | (v.y << 16), (v.z & 0xffffu) | (v.w << 16));
      }
      lds_barrier();
      if (tr && tid == 0) tr[1] = wall_clock64();
    };
    attn_dec_body<DH, G, MH_DS>(a, smem, bx, kvh, 0, hook);
    MH_T(2)
    {
      uint4 wa[16], wb[16];
      mh_prime_gate_up<3>(mc, lane, (int)blockIdx.x * 8 + wave, wa, wb);   // 32 KB per wave while x is on its way
      lds_barrier();                                     // the attention's LDS is dead
      bf16_t* xr = reinterpret_cast<bf16_t*>(smem + MH_OFF_X);
      mh_fetch_vector(p.xo_tag, mc.d, 7, tag, xr, p.err, 12u);
      MH_T(3)
      mh_mlp_body(mc, smem, vthr_gu, tag, blockIdx.x, 3, wa, wb, tr ? tr + 8 : nullptr, x_out_lines);
      MH_T(5)
    }
    return;
  }
  // ---- projection workgroups: 64 qkv rows each (two units of 4 per wave), the first 128 of them 32 o rows each ----
  const int d = p.d, K = a.H * DH;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                   // [d]  RMSNorm(x) * ln1, then (the qkv rows done) ...
  bf16_t* xa = xs;                                                // [K]  ... the attention output
  bf16_t* xin = reinterpret_cast<bf16_t*>(smem + MH_OFF_X);       // [d]  the layer's input when it arrives as a tagged vector, then x after o
  float* red = reinterpret_cast<float*>(smem + MH_OFF_RED);       // [64] + a 64-word line
  const int pw = (int)blockIdx.x - MH_NATTN;
  uint4 wa[16], wb[16], ga[16], gb[16];
  const int rq0 = pw * 64 + wave * 4;
  auto row_q = [&](int u, int i) __attribute__((always_inline)) -> size_t { return (size_t)(rq0 + u * 32 + i) * d; };
  const int nit_q = d >> 9;
  auto shift_q = [&](int u) __attribute__((always_inline)) { return (MH_KROT_MASK & 1) ? ((rq0 + u * 32) >> 2) % nit_q : 0; };   // rotated k order (PcyGemvArgs::krot)
  // The wave's 8 rows = four batches of 4 rows x 4 k-iterations; THREE of them are requested before x is there (wa, wb, and ga, which is idle
  // until the attention): 72 MB of the 100 MB of Wqkv in flight from the first cycle of the layer (two batches: 48 MB had landed 2.5 us before x
  // arrived and the fourth was one more round trip).  The first in front of the loads that fetch x, the others behind them (a CU's loads return
  // in order).  Same k order per row as mc_stream<4, 4> with shift_q.
  auto q_it = [&](int u, int it) __attribute__((always_inline)) { const int r = it + shift_q(u); return r >= nit_q ? r - nit_q : r; };
  auto q_issue = [&](int u, int half, uint4 (&w)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int un = 0; un < 4; ++un) {
      const int k = (q_it(u, half * 4 + un) * 64 + lane) * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) w[un * 4 + i] = ldg_nt(p.wqkv + row_q(u, i) + k);
    }
  };
  if (x_in_lines) {
    q_issue(0, 0, wa);
    mc_fetch_vector_lines(x_in_lines, d, 7, tag, xin, p.err, 14u, [&]() __attribute__((always_inline)) { q_issue(0, 1, wb); q_issue(1, 0, ga); });
    mc_rms_stage(xin, p.ln1, d, vthr_qkv, p.rms_eps, p.rms_cast, xs, red, []() __attribute__((always_inline)) {});
  } else {
    mc_rms_stage(p.x, p.ln1, d, vthr_qkv, p.rms_eps, p.rms_cast, xs, red, [&]() __attribute__((always_inline)) { q_issue(0, 0, wa); q_issue(0, 1, wb); q_issue(1, 0, ga); });
  }
  MH_T(14)
  uint32_t* line = reinterpret_cast<uint32_t*>(red) + 64;
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto q_compute = [&](int u, int half, const uint4 (&w)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int un = 0; un < 4; ++un) {
        const uint4 xv = *reinterpret_cast<const uint4*>(xs + (q_it(u, half * 4 + un) * 64 + lane) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = dot8(w[un * 4 + i], xv, acc[i]);
      }
    };
    auto q_finish = [&](int u) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) line[u * 32 + wave * 4 + i] = (tag << 16) | f2bf(rbf(acc[i]));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = 0.f;
    };
    q_compute(0, 0, wa);
    q_issue(1, 1, wa);
    q_compute(0, 1, wb);
    q_finish(0);
    q_compute(1, 0, ga);
    q_compute(1, 1, wa);
    q_finish(1);
  }
  // the workgroup's 64 qkv rows = two 128-byte lines of the tagged vector, stored by one instruction -- in FRONT of the weight requests below
  // and behind a barrier that does not drain them (__syncthreads() waits for vmcnt(0): with the Wo rows requested first the rows were
  // published 7 us later, in-kernel stamps)
  lds_barrier();
  if (wave == 0) __hip_atomic_store(p.qkv_tag + pw * 64 + lane, line[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  MH_T(1)
  // o rows [r0, r0 + 4) (the first d / 32 projection workgroups): into the same registers while the attention runs; the first batch of the
  // wave's gate/up rows beside them.  (The second batch by LDS-DMA as well, 16 KB per wave into the idle act region: 100 MB instead of 67 in
  // flight during the ~9 us of the attention, the polls for its output queue behind them -- 2.47 -> 2.62 ms per token.)
  const int r0 = pw * 32 + wave * 4;
  const bool active = r0 < d;                 // (workgroup-uniform: d % 32 == 0)
  auto row_o = [&](int u, int i) __attribute__((always_inline)) -> size_t { return (size_t)(r0 + i) * K; };
  const int nit_o = K >> 9;
  auto shift_o = [&](int) __attribute__((always_inline)) { return (MH_KROT_MASK & 2) ? (r0 >> 2) % nit_o : 0; };
  float res[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    mc_prime<4, 4, 3>(p.wo, K, lane, 0, 1, 1, wa, wb, row_o, shift_o);
    // (two address spaces, two branches: through one pointer these would be flat loads, which count in vmcnt AND lgkmcnt)
    if (x_in_lines) {
#pragma unroll
      for (int i = 0; i < 4; ++i) res[i] = bf2f(xin[r0 + i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) res[i] = bf2f(p.x[r0 + i]);
    }
  }
  mh_prime_gate_up<1>(mc, lane, (int)blockIdx.x * 8 + wave, ga, gb);
  // the attention output: one wave watches a 1 KB sample, then every wave takes its share
  if (wave == 0) {
    unsigned spins = 0;
    for (;;) {
      const uint4 v = ld16_agent(p.ao_tag + lane * 4);
      const bool ok = (v.x >> 16) == tag && (v.y >> 16) == tag && (v.z >> 16) == tag && (v.w >> 16) == tag;
      if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
      if (pcy_wait_give_up(spins, 1u << 19, p.err, 10u, lane)) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  lds_barrier();
  MH_T(2)
  {
    uint4 tq[2];
    mc_fetch_issue<2>(p.ao_tag, wave * 512, lane, tq);
    mc_fetch_finish<2>(p.ao_tag, wave * 512, lane, tag, xa, tq, p.err, 11u);
  }
  lds_barrier();
  MH_T(3)
  if (active) {
    uint32_t* oline = reinterpret_cast<uint32_t*>(red);   // the workgroup's 32 results, stored as one line by one wave
    mc_stream<4, 4>(p.wo, K, xa, lane, 0, 1, 1, wa, wb, true, row_o, [&](int u, const float (&acc)[4]) __attribute__((always_inline)) {
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = rbf(acc[i]);
          v = rbf(v + res[i]);
          oline[wave * 4 + i] = f2bf(v);
        }
      }
    }, [](int) __attribute__((always_inline)) {}, nullptr, shift_o);
    lds_barrier();
    if (wave == 0 && lane < 32) __hip_atomic_store(p.xo_tag + pw * 32 + lane, (tag << 16) | oline[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  MH_T(4)
  // the second batch of this wave's gate/up rows while the residual stream is on its way.  (BEHIND the loads that fetch it -- they wait for
  // whatever was requested in front of them -- the first fetch often finds a line of another workgroup not yet current, and the retry then
  // queues behind the batch: 2.47 -> 2.59 ms per token.)
  mh_prime_gate_up<2>(mc, lane, (int)blockIdx.x * 8 + wave, ga, gb);
  lds_barrier();                                       // every wave is done with the attention output in LDS
  mh_fetch_vector(p.xo_tag, mc.d, 7, tag, xin, p.err, 13u);
  mh_mlp_body(mc, smem, vthr_gu, tag, blockIdx.x, 3, ga, gb, tr ? tr + 8 : nullptr, x_out_lines);
  MH_T(5)
#undef MH_T
}

__global__ __launch_bounds__(512) void decode_layer_mha_kernel(PcyDecAttnArgs a, PcyAttnBlockArgs p, PcyMlpChainArgs mc, const unsigned* step_epoch, int vthr_qkv,
                                                               size_t stage_off, int vthr_gu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  mh_layer_body(a, p, mc, *step_epoch, vthr_qkv, stage_off, vthr_gu, smem, nullptr, nullptr, p.trace);
}

// All decoder layers in ONE launch: the residual stream crosses the layer boundary as a tagged vector (one line per workgroup).
__global__ __launch_bounds__(512) void decode_step_mha_kernel(PcyDecAttnArgs a, PcyAttnBlockArgs p, PcyMlpChainArgs mc, PcyDecodeStepArgs st, const unsigned* step_epoch,
                                                              int vthr_qkv, size_t stage_off, int vthr_gu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned xepoch = *step_epoch;
  for (int l = 0; l < st.n_layers; ++l) {
    const PcyLayerWeightsDev lw = st.layers[l];
    p.ln1 = lw.ln1; p.wqkv = lw.wqkv; p.wo = lw.wo;
    mc.ln2 = lw.ln2; mc.wgu = lw.wgu; mc.wdown = lw.wdown;
    PcyDecAttnArgs al = a;
    al.kcache = a.kcache + (size_t)l * st.kv_layer_stride; al.vcache = a.vcache + (size_t)l * st.kv_layer_stride;
    al.xflags = a.xflags ? a.xflags + (size_t)l * st.xflags_stride : nullptr;
    uint32_t* tags = st.tags + (size_t)l * st.tag_stride;
    mc.act_tag = tags; p.qkv_tag = tags + mc.F; p.ao_tag = p.qkv_tag + p.Nq; p.xo_tag = p.ao_tag + a.H * MH_DH;
    const uint32_t* xin = l > 0 ? st.x_lines + (size_t)(l - 1) * st.x_lines_stride : nullptr;
    uint32_t* xout = l + 1 < st.n_layers ? st.x_lines + (size_t)l * st.x_lines_stride : nullptr;
    if (l > 0) __syncthreads();   // the previous layer's LDS is dead
    mh_layer_body(al, p, mc, xepoch, vthr_qkv, stage_off, vthr_gu, smem, xin, xout, p.trace ? p.trace + (size_t)l * 256 * 16 : nullptr);
  }
}

}  // namespace

// Geometry the multi-head step covers (what pcy_engine.hip asks before it forces the launch-per-stage twin's attention to 64 columns).
bool pcy_decode_mha_covers(int d, int H, int Hkv, int dh, int F, int n_cu) {
  return dh == MH_DH && H == MH_HKV && Hkv == MH_HKV && d == MH_D && F / 2 > 2 * 2048 && F / 2 <= 3 * 2048 && F % 256 == 0 && n_cu >= 256;
}
int pcy_decode_mha_ds() { return MH_DS; }

// One layer (st == nullptr) or all layers of a batch-1 decode step; false = not covered, nothing launched.  Arguments as
// pcy_launch_decode_layer / pcy_launch_decode_step (pcy_attn.hip).
bool pcy_launch_decode_mha(hipStream_t s, PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, const PcyDecodeStepArgs* st, int n_cu,
                           const unsigned* step_epoch, unsigned* xflags) {
  if (a.B != 1 || a.dbg || !pcy_decode_mha_covers(p.d, a.H, a.Hkv, a.dh, mc.F, n_cu) || mc.d != p.d || p.Nq != 3 * 4096) return false;
  if (st && st->n_layers < 1) return false;
  a.o_sc1 = 0;
  const char* xe = getenv("PCY_AO_XMIN");   // key split between the two slice workgroups of a head (see launch_attn_o_rw)
  const int xmin = xe ? atoi(xe) : 768;
  a.xflags = (xmin > 0 && a.scratch) ? xflags : nullptr;
  a.xmin = xmin;
  a.unit_map = 1;
  const size_t stage_off = (attn_dec_smem_bytes(1, MH_DS, MH_DH, a.Tmax) + 15) & ~(size_t)15;
  const size_t smem_attn = stage_off + (size_t)3 * MH_DH * 2, smem_o = (size_t)MH_OFF_ACT;
  const size_t smem_mlp = (size_t)MH_OFF_ACT + (size_t)mc.F * 2 + 128;
  size_t smem = smem_attn > smem_o ? smem_attn : smem_o;
  smem = smem > smem_mlp ? smem : smem_mlp;
  if (smem > 160 * 1024) return false;
  static PcyLdsAttr lds[2];
  if (!(st ? lds[1].ensure(&decode_step_mha_kernel, smem) : lds[0].ensure(&decode_layer_mha_kernel, smem))) return false;
  static PcyResidentCache res[2];
  if (!res[st ? 1 : 0].check(smem, [&] {
        return st ? pcy_all_resident(decode_step_mha_kernel, 512, smem, 256, n_cu) : pcy_all_resident(decode_layer_mha_kernel, 512, smem, 256, n_cu);
      }))
    return false;
  if (st)
    hipLaunchKernelGGL(decode_step_mha_kernel, dim3(256), dim3(512), smem, s, a, p, mc, *st, step_epoch, pcy_gemv_rms_threads(p.Nq), stage_off,
                       pcy_gemv_rms_threads(mc.F));
  else
    hipLaunchKernelGGL(decode_layer_mha_kernel, dim3(256), dim3(512), smem, s, a, p, mc, step_epoch, pcy_gemv_rms_threads(p.Nq), stage_off,
                       pcy_gemv_rms_threads(mc.F));
  return true;
}
