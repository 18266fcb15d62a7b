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
//   * ffn = 11008 = 21.5 x 512: the gate/up stage has 2752 units of 4 features for 1792 waves -- round one is exactly the first 7168 features
//     (the first 14 k-iterations of the down projection, as on Llama-3), the 960 units of round two are dealt one per (wave, workgroup) with
//     the wave index major, so that every CU keeps 3-4 waves streaming; the down projection's last k-iteration is half a one (KTAIL).
// Per-row arithmetic, accumulation order and rounding points are those of gemv_stream_kernel (RMSNorm statistics summed with the
// stand-alone launches' thread counts) and of attn_dec_kernel<128, 1, 64>: bit-identical to the launch-per-stage step with 64-column
// attention workgroups (PCY_DISABLE=decode_step,decode_layer; tests/test_gpu_round6.py).
#include <stdlib.h>
#include "pcy_internal.h"
#include "pcy_handover.h"
#include "pcy_mlp_chain.h"
#include "pcy_attn_dec.h"

namespace {

constexpr int MH_DH = 128, MH_DS = 64, MH_HKV = 32, MH_NATTN = (MH_DH / MH_DS) * MH_HKV;   // 64 attention workgroups
constexpr int MH_R1 = 256 * 7;                                                             // gate/up units of round one (one per wave)
constexpr int MH_HALF = MH_R1 * 4;                                                         // act words of round one = 14 k-iterations of the down rows

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

// Stages 1 and 2 of the MLP for one of 256 workgroups (mc_mlp_body of pcy_mlp_chain.h for 7168 < F <= 14336, F % 256 == 0, x in LDS):
//   act = SwiGLU(RMSNorm(x) * ln2 . Wgu^T) handed over as tagged words, x_out = x + act . Wdown^T.
// LDS: [d] normalised x | [F] act | [d] x (already there) | red.  primed: bit 0 / 1 = the first / second batch of the wave's first gate/up
// unit is already in wa / wb.  x_out_lines as mc_mlp_body.
__device__ __forceinline__ void mh_mlp_body(const PcyMlpChainArgs& a, char* smem, int vthr_gu, uint32_t tag, int wg, int primed, uint4 (&wa)[16],
                                            uint4 (&wb)[16], unsigned long long* tr, uint32_t* x_out_lines) {
  constexpr int G = 256;
  const int d = a.d, F = a.F;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* xa = xs + d;
  bf16_t* xr = xa + F;
  float* red = reinterpret_cast<float*>(xr + d);
  const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NW = G * MC_WV, gw = wg * MC_WV + wave;
  uint4 tq[4];
#define MH_T(i) if (tr && tid == 0) tr[i] = wall_clock64();
  const bf16_t* xin = xr;
  const int units_d = d / 2, r0 = gw * 2;
  auto row_d = [&](int u, int i) __attribute__((always_inline)) -> size_t { return (size_t)(u * 2 + i) * F; };
  // ---- stage 1: units of 4 features (8 weight rows), 7 waves per workgroup; local unit 0 = round one, 1 = round two ----
  const int units_g = (F + 3) / 4, gidx = wg * 7 + wave;
  const int u2 = MH_R1 + wave * G + wg;
  const int nloc = u2 < units_g ? 2 : 1;
  const McRowG row_g{F, d};
  auto row_gl = [&](int u, int i) __attribute__((always_inline)) -> size_t { return row_g(u == 0 ? gidx : u2, i); };
  mc_rms_stage(xin, a.ln2, d, vthr_gu, a.rms_eps, a.rms_cast, xs, red, [&]() __attribute__((always_inline)) {
    if (wave < 7) {
      if (primed == 0) mc_prime<8, 2, 3>(a.wgu, d, lane, 0, 1, nloc, wa, wb, row_gl);
      else if (primed == 1) mc_prime<8, 2, 2>(a.wgu, d, lane, 0, 1, nloc, wa, wb, row_gl);
    }
  });
  if (wave < 7) {
    mc_stream<8, 2>(a.wgu, d, xs, lane, 0, 1, nloc, wa, wb, true, row_gl, [&](int u, const float (&acc)[8]) __attribute__((always_inline)) {
      if (lane == 0) {
        const int unit = u == 0 ? gidx : u2;
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float g = rbf(acc[i]), up = rbf(acc[i + 4]);
          o[i] = (tag << 16) | f2bf(rbf(silu_f(g)) * up);
        }
        st8_agent(a.act_tag + unit * 4, o[0], o[1]);
        st8_agent(a.act_tag + unit * 4 + 2, o[2], o[3]);
      }
    }, [](int) __attribute__((always_inline)) {});
    MH_T(1)
    // ---- stage 2 begins for this wave: round one's part of act, then (once it is there) the first two batches of its down rows ----
    mc_fetch_issue<4>(a.act_tag, wave * 1024, lane, tq);
  }
  if (wave < 7) mc_fetch_finish<4>(a.act_tag, wave * 1024, lane, tag, xa, tq, a.err, 6u);
  mc_prime<2, MC_UNB_D>(a.wdown, F, lane, gw, NW, units_d, wa, wb, row_d);
  // round two's part: words [MH_HALF, F), 1024 per wave in loads of 256
  int nj2 = (F - MH_HALF - wave * 1024) / 256;
  nj2 = wave < 7 ? (nj2 < 0 ? 0 : (nj2 > 4 ? 4 : nj2)) : 0;
  mh_fetch_issue<4>(a.act_tag, MH_HALF + wave * 1024, lane, tq, nj2);
  __syncthreads();
  MH_T(2)
  float acc[2] = {0.f, 0.f};
  constexpr int it_half = MH_HALF / 512;   // 14 = two batches of MC_UNB_D
  static_assert(it_half % MC_UNB_D == 0, "the second part of act is checked in front of a batch");
  mc_stream<2, MC_UNB_D, -1, true>(a.wdown, F, xa, lane, gw, NW, units_d, wa, wb, true, row_d,
                                   [&](int u, const float (&acc2)[2]) __attribute__((always_inline)) { acc[0] = acc2[0]; acc[1] = acc2[1]; },
                                   [&](int it0) __attribute__((always_inline)) {
                                     if (it0 == it_half) {   // (workgroup-uniform: every wave walks the same batches of its one unit)
                                       mh_fetch_finish<4>(a.act_tag, MH_HALF + wave * 1024, lane, tag, xa, tq, nj2, a.err, 7u);
                                       __syncthreads();
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
      if (wave < 7) mc_prime_gate_up(mc, lane, (int)blockIdx.x * 7 + wave, wa, wb, true);   // 32 KB per wave while x is on its way
      lds_barrier();                                     // the attention's LDS is dead
      bf16_t* xr = reinterpret_cast<bf16_t*>(smem) + mc.d + mc.F;
      mh_fetch_vector(p.xo_tag, mc.d, 7, tag, xr, p.err, 12u);
      MH_T(3)
      mh_mlp_body(mc, smem, vthr_gu, tag, blockIdx.x, 3, wa, wb, tr ? tr + 8 : nullptr, x_out_lines);
      MH_T(5)
    }
    return;
  }
  // ---- projection workgroups: 64 qkv rows each (two units of 4 per wave), the first 128 of them 32 o rows each ----
  const int d = p.d, K = a.H * DH;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);             // [d]  RMSNorm(x) * ln1
  bf16_t* xa = xs + d;                                      // [K]  attention output
  float* red = reinterpret_cast<float*>(xa + K);            // [64] + a 64-word line
  bf16_t* xin = reinterpret_cast<bf16_t*>(red + 128);       // [d]  the layer's input when it arrives as a tagged vector
  const int pw = (int)blockIdx.x - MH_NATTN;
  uint4 wa[16], wb[16], ga[16], gb[16];
  const int rq0 = pw * 64 + wave * 4;
  auto row_q = [&](int u, int i) __attribute__((always_inline)) -> size_t { return (size_t)(rq0 + u * 32 + i) * d; };
  if (x_in_lines) {
    // the first batch in front of the loads that fetch x, the second behind them (a CU's loads return in order)
    mc_prime<4, 4, 1>(p.wqkv, d, lane, 0, 1, 2, wa, wb, row_q);
    mc_fetch_vector_lines(x_in_lines, d, 7, tag, xin, p.err, 14u, [&]() __attribute__((always_inline)) { mc_prime<4, 4, 2>(p.wqkv, d, lane, 0, 1, 2, wa, wb, row_q); });
    mc_rms_stage(xin, p.ln1, d, vthr_qkv, p.rms_eps, p.rms_cast, xs, red, []() __attribute__((always_inline)) {});
  } else {
    mc_rms_stage(p.x, p.ln1, d, vthr_qkv, p.rms_eps, p.rms_cast, xs, red, [&]() __attribute__((always_inline)) { mc_prime<4, 4, 3>(p.wqkv, d, lane, 0, 1, 2, wa, wb, row_q); });
  }
  MH_T(14)
  uint32_t* line = reinterpret_cast<uint32_t*>(red) + 64;
  mc_stream<4, 4>(p.wqkv, d, xs, lane, 0, 1, 2, wa, wb, true, row_q, [&](int u, const float (&acc)[4]) __attribute__((always_inline)) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) line[u * 32 + wave * 4 + i] = (tag << 16) | f2bf(rbf(acc[i]));
    }
  }, [](int) __attribute__((always_inline)) {});
  // the workgroup's 64 qkv rows = two 128-byte lines of the tagged vector, stored by one instruction -- in FRONT of the weight requests below
  // and behind a barrier that does not drain them (__syncthreads() waits for vmcnt(0): with the Wo rows requested first the rows were
  // published 7 us later, in-kernel stamps)
  lds_barrier();
  if (wave == 0) __hip_atomic_store(p.qkv_tag + pw * 64 + lane, line[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  MH_T(1)
  // o rows [r0, r0 + 4) (the first d / 32 projection workgroups): into the same registers while the attention runs; the first batch of the
  // wave's gate/up rows beside them
  const int r0 = pw * 32 + wave * 4;
  const bool active = r0 < d;                 // (workgroup-uniform: d % 32 == 0)
  auto row_o = [&](int u, int i) __attribute__((always_inline)) -> size_t { return (size_t)(r0 + i) * K; };
  float res[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    mc_prime<4, 4, 3>(p.wo, K, lane, 0, 1, 1, wa, wb, row_o);
    // (two address spaces, two branches: through one pointer these would be flat loads, which count in vmcnt AND lgkmcnt)
    if (x_in_lines) {
#pragma unroll
      for (int i = 0; i < 4; ++i) res[i] = bf2f(xin[r0 + i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) res[i] = bf2f(p.x[r0 + i]);
    }
  }
  if (wave < 7) mc_prime_gate_up(mc, lane, (int)blockIdx.x * 7 + wave, ga, gb, false);   // (both batches for the workgroups without o rows: 210 spilled VGPRs)
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
    }, [](int) __attribute__((always_inline)) {});
    lds_barrier();
    if (wave == 0 && lane < 32) __hip_atomic_store(p.xo_tag + pw * 32 + lane, (tag << 16) | oline[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  MH_T(4)
  // the second batch of this wave's gate/up rows while the residual stream is on its way
  if (wave < 7) mc_prime<8, 2, 2>(mc.wgu, mc.d, lane, (int)blockIdx.x * 7 + wave, 1, (mc.F + 3) / 4, ga, gb, McRowG{mc.F, mc.d});
  lds_barrier();                                       // every wave is done with the attention output in LDS
  bf16_t* xr = reinterpret_cast<bf16_t*>(smem) + mc.d + mc.F;
  mh_fetch_vector(p.xo_tag, mc.d, 7, tag, xr, p.err, 13u);
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
  return dh == MH_DH && H == MH_HKV && Hkv == MH_HKV && d == 4096 && F > MH_HALF && F <= 2 * MH_HALF && F % 256 == 0 && n_cu >= 256;
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
  const size_t smem_attn = stage_off + (size_t)3 * MH_DH * 2, smem_o = (size_t)(2 * p.d + a.H * MH_DH) * 2 + 512;
  const size_t smem_mlp = (size_t)(2 * mc.d + mc.F) * 2 + 128;
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
