// Small-batch decode step (2..8 rows: beam search, the N > 1 points of the row split) as ONE launch: every decoder layer of a
// step for NB rows, the weights streamed ONCE.  Reference: the decode step of model_unified.py:769,887 at batch = beam_size
// (scripts/caption_bulk.py:123-132 generates with beam 5).
//
// Structure = decode_step_kernel's (pcy_attn.hip), generalised to NB accumulators per weight row:
//   workgroups [0, n_attn)   decode attention of one (row, kv head, DS-column slice): attn_dec_body<DH, G, DS>, the new token's q / k / v
//                            taken from the tagged qkv vector of ITS row.  DS grows with NB (32 / 64 / 128 columns) so that the attention never
//                            takes more than 64 workgroups.
//   workgroups [n_attn, 256) qkv projection (4 rows of Wqkv per wave in registers, NB dot products per row), the o projection behind
//                            the attention, first gate/up batches requested while the attention runs.
//   every workgroup          the MLP: gate/up (7 waves x 2 units of 4 features) -> act -> down (2 rows per wave).
// Hand-over between workgroups: {tag : bf16} words (pcy_handover.h), one vector per row.
//
// Per-row arithmetic is gemv_stream_kernel<NB, ...>'s (pcy_gemv.hip): the same k order per lane, the same wave reduction tree, the same
// rounding points, RMSNorm statistics summed with the stand-alone launches' thread counts -- the launch is bit-identical to the
// launch-per-stage step built from gemv_stream_kernel + attn_dec_kernel<DH, G, DS> (PCY_DISABLE=decode_nb_step; tests compare the two).
//
// LDS (16 KB x NB + 4 KB; 132 KB at 8 rows): two regions of [NB][4096] bf16.  `act` ([NB][14336]) never sits in LDS as a whole: the down
// projection walks K in four windows of 3584 (one batch of 7 k-iterations per window); window q is fetched from the tagged vector into
// registers while window q - 1 is consumed and lands in the region that window q - 2 has left (the two regions alternate).
#include <stdlib.h>
#include "pcy_internal.h"
#include "pcy_handover.h"
#include "pcy_mlp_chain.h"
#include "pcy_attn_dec.h"

namespace {

constexpr int NBD = 4096, NBF = 14336, NBWIN = NBF / 4, NBNQ = 6144;

template <int NB> struct NbGeom {
  static constexpr int DS = NB <= 1 ? 16 : NB == 2 ? 32 : NB <= 4 ? 64 : 128;
  static constexpr int SLICES = 128 / DS;
  static constexpr int XS = ((NB * 16 + 31) / 32) * 32;   // words of one workgroup's share of the residual stream between two layers
  static constexpr int P2 = NB <= 1 ? 1 : NB <= 2 ? 2 : NB <= 4 ? 4 : 8;
};

// Wave totals of N values in N - 1 exchanges (+ the steps that are left when N < 64) instead of 6 N: at offset 32 a lane of the lower
// half keeps the even value of every pair and receives the partner's copy of it, the upper half the odd one; and so on.  Lane l ends with
// the total of value nb_red_index<N>(l).  The additions of one value form the xor butterfly of wave_sum (own + partner at offsets
// 32, 16, .. 1; fp addition commutes), so every total has wave_sum's bits.
template <int N>
__device__ __forceinline__ float nb_wave_reduce(float (&v)[N], int lane) {
  static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "power of two");
  int off = 32;
#pragma unroll
  for (int n = N; n > 1; n >>= 1) {
    const bool up = (lane & off) != 0;
#pragma unroll
    for (int j = 0; j < n / 2; ++j) {
      const float keep = up ? v[2 * j + 1] : v[2 * j];
      const float send = up ? v[2 * j] : v[2 * j + 1];
      v[j] = keep + __shfl_xor(send, off, 64);
    }
    off >>= 1;
  }
  float r = v[0];
#pragma unroll
  for (; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
  return r;
}
// index of the value whose total lane `lane` holds after nb_wave_reduce<N>; the lanes that differ only in the low 6 - log2(N) bits hold
// the same total
template <int N>
__device__ __forceinline__ int nb_red_index(int lane) {
  int idx = 0, bit = 0;
#pragma unroll
  for (int n = N, off = 32; n > 1; n >>= 1, off >>= 1, ++bit) idx |= ((lane & off) ? 1 : 0) << bit;
  return idx;
}
template <int N>
__device__ __forceinline__ bool nb_red_owner(int lane) {
  int low = 63;
#pragma unroll
  for (int n = N, off = 32; n > 1; n >>= 1, off >>= 1) low &= ~off;
  return (lane & low) == 0;
}

// xs[b][0..K) = bf16(RMSNorm(x[b]) * w) for NB rows held in LDS (x, row stride K), the statistic of every row summed like
// gemv_stream_kernel launched with `vthr` threads (mc_rms_stage).  K == 4096, 512 threads.  Ends with a barrier.
template <int NB>
__device__ __forceinline__ void nb_rms_stage(const bf16_t* x, const bf16_t* __restrict__ w, int vthr, float eps, int cast, bf16_t* xs, float* red) {
  constexpr int K = NBD;
  const int tid = pcy_tid(), lane = tid & 63, wave = tid >> 6;
  const uint4 g = ldg16(w + tid * 8);
  float ss[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (tid + i * vthr) * 8;
      if (tid < vthr && k < K) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + b * K + k);
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float f0 = lo_bf(w4[j]), f1 = hi_bf(w4[j]); s += f0 * f0 + f1 * f1; }
      }
    }
    ss[b] = wave_sum(s);
  }
  lds_barrier();
  if (lane == 0) {
#pragma unroll
    for (int b = 0; b < NB; ++b) red[b * 8 + wave] = ss[b];
  }
  lds_barrier();
  const int nw = vthr >> 6;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[b * 8 + i];
    const float rs = rsqrtf(t / (float)K + eps);
    const uint4 xv = *reinterpret_cast<const uint4*>(x + b * K + tid * 8);
    const uint32_t xin[4] = {xv.x, xv.y, xv.z, xv.w}, gin[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x0 = lo_bf(xin[j]) * rs, x1 = hi_bf(xin[j]) * rs;
      if (cast == 0) { x0 = rbf(x0); x1 = rbf(x1); }
      o[j] = pack_bf(lo_bf(gin[j]) * x0, hi_bf(gin[j]) * x1);
    }
    *reinterpret_cast<uint4*>(xs + b * K + tid * 8) = make_uint4(o[0], o[1], o[2], o[3]);
  }
  lds_barrier();
}

// NB tagged vectors of 4096 words (row stride `sstride` words) -> LDS dst [NB][4096] bf16: one wave watches a sample of the last row,
// then every wave takes its 512 words of every row, four rows per round trip.  All 512 threads; ends with a barrier.
template <int NB>
__device__ __forceinline__ void nb_fetch_vectors(const uint32_t* src, size_t sstride, int watch_wave, uint32_t tag, bf16_t* dst, unsigned* err, unsigned code) {
  const int tid_ = pcy_tid(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  if (wave == watch_wave) {
    unsigned spins = 0;
    for (;;) {
      const uint4 v = ld16_agent(src + (size_t)(NB - 1) * sstride + NBD - 256 + lane * 4);
      const bool ok = (v.x >> 16) == tag && (v.y >> 16) == tag && (v.z >> 16) == tag && (v.w >> 16) == tag;
      if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
      if (pcy_wait_give_up(spins, 1u << 19, err, code, lane)) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
#pragma unroll
  for (int b0 = 0; b0 < NB; b0 += 4) {
    constexpr int CH = 4;
    uint4 t[CH][2];
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (b0 + c < NB) mc_fetch_issue<2>(src + (size_t)(b0 + c) * sstride, wave * 512, lane, t[c]);
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (b0 + c < NB) mc_fetch_finish<2>(src + (size_t)(b0 + c) * sstride, wave * 512, lane, tag, dst + (b0 + c) * NBD, t[c], err, code);
  }
  __syncthreads();
}

// The residual stream between two layers: workgroup g of the producing layer owns words [g * XS, g * XS + 16 NB) of `src`, element
// 16 g + i of row b at word g * XS + 16 b + i (a line has one writer for even NB).  -> dst [NB][4096] bf16.  Ends with a barrier.
template <int NB>
__device__ __forceinline__ void nb_fetch_lines(const uint32_t* src, int watch_wave, uint32_t tag, bf16_t* dst, unsigned* err, unsigned code) {
  constexpr int XS = NbGeom<NB>::XS;
  const int tid_ = pcy_tid(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  auto word_of = [&](int g, int b, int piece) __attribute__((always_inline)) { return g * XS + b * 16 + piece * 4; };
  if (wave == watch_wave) {   // the last 16 workgroups' words of the last row
    unsigned spins = 0;
    for (;;) {
      const uint4 v = ld16_agent(src + word_of(256 - 16 + (lane >> 2), NB - 1, lane & 3));
      const bool ok = (v.x >> 16) == tag && (v.y >> 16) == tag && (v.z >> 16) == tag && (v.w >> 16) == tag;
      if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
      if (pcy_wait_give_up(spins, 1u << 19, err, code, lane)) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  __syncthreads();
  // this wave: elements [512 wave, 512 wave + 512) of every row = workgroups [32 wave, 32 wave + 32): two loads of 16 workgroups x 4 pieces
#pragma unroll
  for (int b0 = 0; b0 < NB; b0 += 4) {
    uint4 t[4][2];
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (b0 + c < NB) {
            t[c][j] = ld16_agent(src + word_of(wave * 32 + j * 16 + (lane >> 2), b0 + c, lane & 3));
            ok = ok && (t[c][j].x >> 16) == tag && (t[c][j].y >> 16) == tag && (t[c][j].z >> 16) == tag && (t[c][j].w >> 16) == tag;
          }
      if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
      if (pcy_wait_give_up(spins, 1u << 19, err, code, lane)) break;
      __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (b0 + c < NB)
          *reinterpret_cast<uint2*>(dst + (b0 + c) * NBD + (wave * 32 + j * 16 + (lane >> 2)) * 16 + (lane & 3) * 4) =
              make_uint2((t[c][j].x & 0xffffu) | (t[c][j].y << 16), (t[c][j].z & 0xffffu) | (t[c][j].w << 16));
  }
  __syncthreads();
}

// acc[r][b] (r < RW rows, b < NB) -> one array in the index order 4 b + r (RW <= 4), padded to a power of two
template <int RW, int NB, int NP>
__device__ __forceinline__ void nb_gather(const float (&acc)[RW][NB], float (&v)[NP]) {
#pragma unroll
  for (int idx = 0; idx < NP; ++idx) v[idx] = (idx / RW < NB) ? acc[idx % RW][(idx / RW) < NB ? idx / RW : 0] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// The MLP of one layer for NB rows (every workgroup; `wg` = its index, 256 of them).  Entry: x after the o projection is on its way as
// the tagged vectors xo_tag [NB][4096]; wa / wb hold the first two batches of this wave's first gate/up unit (waves 0..6).
// ra / rb: the two LDS regions [NB][4096]; misc: >= 2 KB.
template <int NB>
__device__ __forceinline__ void nb_mlp_body(const PcyMlpChainArgs& a, const uint32_t* xo_tag, char* smem, int vthr_gu, uint32_t tag, int wg,
                                            uint4 (&wa)[16], uint4 (&wb)[16], unsigned long long* tr, uint32_t* x_out_lines) {
  constexpr int d = NBD, F = NBF, XS = NbGeom<NB>::XS, P2 = NbGeom<NB>::P2;
  bf16_t* ra = reinterpret_cast<bf16_t*>(smem);
  bf16_t* rb = ra + NB * d;
  float* red = reinterpret_cast<float*>(rb + NB * d);                 // [NB][8]
  uint32_t* line = reinterpret_cast<uint32_t*>(red + NB * 8);         // [NB][32]
  bf16_t* xres = reinterpret_cast<bf16_t*>(line + NB * 32);           // [NB][16] this workgroup's rows of x (the down epilogue's residual)
  const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#define NB_T(i) if (tr && tid == 0) tr[i] = wall_clock64();
  nb_fetch_vectors<NB>(xo_tag, d, 7, tag, rb, a.err, 13u);
  NB_T(8)
  if (tid < NB * 16) xres[tid] = rb[(tid >> 4) * d + wg * 16 + (tid & 15)];
  nb_rms_stage<NB>(rb, a.ln2, vthr_gu, a.rms_eps, a.rms_cast, ra, red);
  // ---- gate/up: units gidx and gidx + 1792 of 4 features (8 weight rows), 4 batches of 2 k-iterations each ----
  const int gw = wg * MC_WV + wave;
  auto issue_down = [&](int q, uint4 (&w)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int un = 0; un < MC_UNB_D; ++un)
#pragma unroll
      for (int i = 0; i < 2; ++i) w[un * 2 + i] = ldg_nt(a.wdown + (size_t)(gw * 2 + i) * F + ((q * MC_UNB_D + un) * 64 + lane) * 8);
  };
  uint4 tq[NB][2];
  auto win_issue = [&](int q) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < NB; ++b) mc_fetch_issue<2>(a.act_tag + (size_t)b * F + q * NBWIN, wave * 512, lane, tq[b]);
  };
  auto win_finish = [&](int q, bf16_t* slot) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      mc_fetch_finish<2>(a.act_tag + (size_t)b * F + q * NBWIN, wave * 512, lane, tag, slot + b * NBWIN, tq[b], a.err, 20u + (unsigned)q);
    }
  };
  if (wave < 7) {
    const McRowG row_g{F, d};
    const int gidx = wg * 7 + wave;
    float acc[8][NB];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[i][b] = 0.f;
    auto finish = [&](int u) __attribute__((always_inline)) {
      constexpr int NP = 4 * P2;
      float vg[NP], vu[NP];
#pragma unroll
      for (int idx = 0; idx < NP; ++idx) {
        const int b = idx >> 2, i = idx & 3;
        vg[idx] = b < NB ? acc[i][b < NB ? b : 0] : 0.f;
        vu[idx] = b < NB ? acc[i + 4][b < NB ? b : 0] : 0.f;
      }
      const float g = rbf(nb_wave_reduce<NP>(vg, lane)), up = rbf(nb_wave_reduce<NP>(vu, lane));
      const int idx = nb_red_index<NP>(lane), b = idx >> 2, i = idx & 3;
      if (nb_red_owner<NP>(lane) && b < NB)
        __hip_atomic_store(a.act_tag + (size_t)b * F + u * 4 + i, (tag << 16) | f2bf(rbf(silu_f(g)) * up), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) acc[r][bb] = 0.f;
    };
    const int u1 = gidx, u2 = gidx + 7 * 256;
    // (the lane index goes through an optimisation barrier in every iteration: otherwise all lane-derived addresses of the loop -- and of
    // the down rows requested at its end -- are hoisted in front of it and kept live: 28 + VGPRs, spills)
    int ln = lane;
    auto issue = [&](int u, int it0, uint4 (&w)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int un = 0; un < 2; ++un)
#pragma unroll
        for (int i = 0; i < 8; ++i) w[un * 8 + i] = ldg_nt(a.wgu + row_g(u, i) + ((it0 + un) * 64 + ln) * 8);
    };
    auto compute = [&](int it0, const uint4 (&w)[16]) __attribute__((always_inline)) {
#pragma unroll
      for (int un = 0; un < 2; ++un)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const uint4 xv = *reinterpret_cast<const uint4*>(ra + b * d + ((it0 + un) * 64 + ln) * 8);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i][b] = dot8(w[un * 8 + i], xv, acc[i][b]);
        }
    };
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
      asm volatile("" : "+v"(ln));
      const int u = s < 2 ? u1 : u2, it0 = (s & 1) * 4;
      const int un_ = s + 1 < 2 ? u1 : u2, itn = ((s + 1) & 1) * 4;
      compute(it0, wa);
      issue(un_, itn, wa);
      compute(it0 + 2, wb);
      if (s & 1) finish(u);
      issue(un_, itn + 2, wb);
    }
    asm volatile("" : "+v"(ln));
    compute(4, wa);
    compute(6, wb);
    finish(u2);
    NB_T(9)
  }
  // (one piece of code for all eight waves: with the requests inside the branch above and in an else-branch for wave 7 the two
  // definitions of wa / wb met in 32 four-register copies and the allocator spilled both batches)
  if (wave < 7) win_issue(0);
  issue_down(0, wa);
  issue_down(1, wb);
  if (wave < 7) { win_finish(0, rb); win_issue(1); }
  lds_barrier();   // window 0 is in rb; nobody reads ra (normalised x) any more
  NB_T(10)
  // ---- down: rows 2 gw, 2 gw + 1; batch q = k-iterations [7 q, 7 q + 7) against window q ----
  float dacc[2][NB];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b) dacc[i][b] = 0.f;
  auto dcompute = [&](const bf16_t* slot, const uint4 (&w)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int un = 0; un < MC_UNB_D; ++un)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint4 xv = *reinterpret_cast<const uint4*>(slot + b * NBWIN + (un * 64 + lane) * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i) dacc[i][b] = dot8(w[un * 2 + i], xv, dacc[i][b]);
      }
    // the sums are pinned here: otherwise the dot products of ALL four batches sink to the end of the function (their results are only used
    // there), behind the window barriers, and four batches of weights stay live -- 224 VGPRs, 80 + spilled
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(dacc[i][b]));
  };
  dcompute(rb, wa);
  issue_down(2, wa);
  if (wave < 7) { win_finish(1, ra); win_issue(2); }
  lds_barrier();   // window 1 in ra; everyone is done with window 0
  dcompute(ra, wb);
  issue_down(3, wb);
  if (wave < 7) { win_finish(2, rb); win_issue(3); }
  lds_barrier();
  NB_T(11)
  dcompute(rb, wa);
  if (wave < 7) win_finish(3, ra);
  lds_barrier();
  dcompute(ra, wb);
  // ---- x_out = x + act . Wdown^T ----
  {
    constexpr int NP = 2 * P2;
    float v[NP];
#pragma unroll
    for (int idx = 0; idx < NP; ++idx) v[idx] = (idx >> 1) < NB ? dacc[idx & 1][(idx >> 1) < NB ? (idx >> 1) : 0] : 0.f;
    const float tot = nb_wave_reduce<NP>(v, lane);
    const int idx = nb_red_index<NP>(lane), b = idx >> 1, i = idx & 1;
    if (nb_red_owner<NP>(lane) && b < NB) {
      float r = rbf(tot);
      r = rbf(r + bf2f(xres[b * 16 + wave * 2 + i]));
      line[b * 16 + wave * 2 + i] = f2bf(r);
    }
  }
  lds_barrier();
  if (tid < NB * 16) {
    const int b = tid >> 4, e = tid & 15;
    if (x_out_lines) __hip_atomic_store(x_out_lines + wg * XS + tid, (tag << 16) | line[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else a.x_out[(size_t)b * d + wg * 16 + e] = (bf16_t)line[tid];
  }
  NB_T(12)
#undef NB_T
}

// ------------------------------------------------------------------------------------------------
// One decoder layer for NB rows (see the head of the file).  x_in_lines == nullptr: the layer's input is p.x [NB][4096] in global memory
// (written before the launch); x_out_lines == nullptr: the result goes to mc.x_out [NB][4096].
template <int DH, int G, int NB>
__device__ __forceinline__ void nb_layer_body(PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, int n_attn, unsigned xepoch,
                                              int vthr_qkv, size_t stage_off, int vthr_gu, char* smem, const uint32_t* x_in_lines,
                                              uint32_t* x_out_lines, unsigned long long* tr_base) {
  constexpr int DS = NbGeom<NB>::DS, SLICES = NbGeom<NB>::SLICES, P2 = NbGeom<NB>::P2;
  constexpr int d = NBD, K = NBD;
  const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t tag = *p.epoch & 0xffffu;
  unsigned long long* tr = tr_base ? tr_base + (size_t)blockIdx.x * 16 : nullptr;
#define NB_T(i) if (tr && tid == 0) tr[i] = wall_clock64();
  NB_T(0)
  const int wg = (int)blockIdx.x;
  if (wg < n_attn) {
    // ---- attention of one (row, kv head, column slice); units beyond SLICES x Hkv x NB (NB = 3, 5, 6, 7) have none ----
    uint4 wa[16], wb[16];
    const int unit = wg;
    const int kvh = unit % a.Hkv, bx = (unit / a.Hkv) % SLICES, b = unit / (a.Hkv * SLICES);   // kv head in the low digits: the slices of a head share an XCD's L2
    if (b < NB) {
      bf16_t* stage = reinterpret_cast<bf16_t*>(smem + stage_off);   // [G + 2][DH]
      a.xepoch = xepoch;
      a.xerr = p.err;
      if constexpr (SLICES == 1) a.xflags = nullptr;   // one workgroup per (row, kv head): no key split, and its code is not compiled
      a.staged = stage; a.o_tag = p.ao_tag; a.tag = tag;
      const uint32_t* qt = p.qkv_tag + (size_t)b * NBNQ;
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
      attn_dec_body<DH, G, DS>(a, smem, bx, kvh, b, hook);
    }
    NB_T(2)
    if (wave < 7) mc_prime_gate_up(mc, lane, wg * 7 + wave, wa, wb, true);   // 32 KB per wave while x is on its way
    __syncthreads();                                   // the attention's LDS is dead
    nb_mlp_body<NB>(mc, p.xo_tag, smem, vthr_gu, tag, wg, wa, wb, tr, x_out_lines);
    return;
  }
  // ---- projection workgroups (192): 4 qkv rows per wave; the first d / 32 of them also 4 o rows per wave ----
  bf16_t* ra = reinterpret_cast<bf16_t*>(smem);             // [NB][d]  RMSNorm(x) * ln1, then the attention output
  bf16_t* rb = ra + NB * d;                                 // [NB][d]  the layer's input
  float* red = reinterpret_cast<float*>(rb + NB * d);       // [NB][8]
  uint32_t* line = reinterpret_cast<uint32_t*>(red + NB * 8);   // [NB][32]
  constexpr int NP = 4 * P2;
  const int gwo = (wg - n_attn) * 8 + wave;
  const int r0 = gwo * 4;
  const bool active = r0 < d;                  // workgroup-uniform (d % 32 == 0)
  {
    uint4 w[32];   // qkv rows [r0, r0 + 4): all 32 KB of this wave requested in front of the wait for x
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int it = 0; it < 8; ++it) w[i * 8 + it] = ldg_nt(p.wqkv + (size_t)(r0 + i) * d + (it * 64 + lane) * 8);
    if (x_in_lines) {
      nb_fetch_lines<NB>(x_in_lines, 7, tag, rb, p.err, 14u);
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b) *reinterpret_cast<uint4*>(rb + b * d + tid * 8) = ldg16(p.x + (size_t)b * d + tid * 8);
      __syncthreads();
    }
    nb_rms_stage<NB>(rb, p.ln1, vthr_qkv, p.rms_eps, p.rms_cast, ra, red);
    float acc[4][NB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[i][b] = 0.f;
#pragma unroll
    for (int it = 0; it < 8; ++it)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint4 xv = *reinterpret_cast<const uint4*>(ra + b * d + (it * 64 + lane) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][b] = dot8(w[i * 8 + it], xv, acc[i][b]);
      }
    float v[NP];
    nb_gather<4, NB, NP>(acc, v);
    const float tot = nb_wave_reduce<NP>(v, lane);
    const int idx = nb_red_index<NP>(lane), b = idx >> 2, i = idx & 3;
    // the workgroup's 32 rows of a row b = one 128-byte line of its tagged vector, stored by one instruction
    if (nb_red_owner<NP>(lane) && b < NB) line[b * 32 + wave * 4 + i] = (tag << 16) | f2bf(rbf(tot));
  }
  __syncthreads();   // (also: every wave is done with RMSNorm(x) in ra)
  if (wave < NB && lane < 32)
    __hip_atomic_store(p.qkv_tag + (size_t)wave * NBNQ + (r0 & ~31) + lane, line[wave * 32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  NB_T(1)
  uint4 wa[16], wb[16];
  // the first gate/up batch of the MLP while the attention runs (16 KB per wave)
  if (wave < 7) mc_prime_gate_up(mc, lane, wg * 7 + wave, wa, wb, false);
  if (active) {
    uint4 w[32];   // o rows [r0, r0 + 4) wait in registers while the attention runs
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 8; ++c) w[i * 8 + c] = ldg_nt(p.wo + (size_t)(r0 + i) * K + (c * 64 + lane) * 8);
    // the attention output [NB][K] -> ra: one wave watches a sample of every row, then every wave takes its share
    if (wave == 0) {
      const int bw = lane % NB;
      unsigned spins = 0;
      for (;;) {
        const uint4 v = ld16_agent(p.ao_tag + (size_t)bw * K + (lane / NB) * 4);
        const bool ok = (v.x >> 16) == tag && (v.y >> 16) == tag && (v.z >> 16) == tag && (v.w >> 16) == tag;
        if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
        if (pcy_wait_give_up(spins, 1u << 19, p.err, 10u, lane)) break;
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __syncthreads();
    NB_T(2)
    // (two rows per round trip: the Wo rows and the first gate/up batch fill 192 registers here)
#pragma unroll
    for (int b0 = 0; b0 < NB; b0 += 2) {
      uint4 t[2][2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (b0 + c < NB) mc_fetch_issue<2>(p.ao_tag + (size_t)(b0 + c) * K, wave * 512, lane, t[c]);
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (b0 + c < NB) mc_fetch_finish<2>(p.ao_tag + (size_t)(b0 + c) * K, wave * 512, lane, tag, ra + (b0 + c) * K, t[c], p.err, 11u);
    }
    __syncthreads();
    NB_T(3)
    float acc[4][NB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[i][b] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint4 xv = *reinterpret_cast<const uint4*>(ra + b * K + (c * 64 + lane) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][b] = dot8(w[i * 8 + c], xv, acc[i][b]);
      }
    float v[NP];
    nb_gather<4, NB, NP>(acc, v);
    const float tot = nb_wave_reduce<NP>(v, lane);
    const int idx = nb_red_index<NP>(lane), b = idx >> 2, i = idx & 3;
    if (nb_red_owner<NP>(lane) && b < NB) {
      float r = rbf(tot);
      r = rbf(r + bf2f(rb[b * d + r0 + i]));
      line[b * 32 + wave * 4 + i] = (tag << 16) | f2bf(r);
    }
    __syncthreads();
    if (wave < NB && lane < 32)
      __hip_atomic_store(p.xo_tag + (size_t)wave * d + (r0 & ~31) + lane, line[wave * 32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    NB_T(4)
  }
  // the second 16 KB of this wave's gate/up rows while the residual stream is on its way
  if (wave < 7) mc_prime<8, 2, 2>(mc.wgu, mc.d, lane, wg * 7 + wave, 256 * 7, (mc.F + 3) / 4, wa, wb, McRowG{mc.F, mc.d});
  __syncthreads();                                     // every wave is done with this phase's LDS
  nb_mlp_body<NB>(mc, p.xo_tag, smem, vthr_gu, tag, wg, wa, wb, tr, x_out_lines);
#undef NB_T
}

template <int DH, int G, int NB>
__global__ __launch_bounds__(512) void decode_step_nb_kernel(PcyDecAttnArgs a, PcyAttnBlockArgs p, PcyMlpChainArgs mc, PcyDecodeStepArgs st, int n_attn,
                                                             const unsigned* step_epoch, int vthr_qkv, size_t stage_off, int vthr_gu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned xepoch = *step_epoch;
  for (int l = 0; l < st.n_layers; ++l) {
    const PcyLayerWeightsDev lw = st.layers[l];
    p.ln1 = lw.ln1; p.wqkv = lw.wqkv; p.wo = lw.wo;
    mc.ln2 = lw.ln2; mc.wgu = lw.wgu; mc.wdown = lw.wdown;
    PcyDecAttnArgs al = a;
    al.kcache = a.kcache + (size_t)l * st.kv_layer_stride; al.vcache = a.vcache + (size_t)l * st.kv_layer_stride;
    al.xflags = a.xflags ? a.xflags + (size_t)l * st.xflags_stride : nullptr;
    uint32_t* tags = st.tags + (size_t)l * st.tag_stride;   // act [NB][F] | qkv [NB][Nq] | attention output [NB][H dh] | x after o [NB][d]
    mc.act_tag = tags; p.qkv_tag = tags + (size_t)NB * NBF; p.ao_tag = p.qkv_tag + (size_t)NB * NBNQ; p.xo_tag = p.ao_tag + (size_t)NB * NBD;
    const uint32_t* xin = l > 0 ? st.x_lines + (size_t)(l - 1) * st.x_lines_stride : nullptr;
    uint32_t* xout = l + 1 < st.n_layers ? st.x_lines + (size_t)l * st.x_lines_stride : nullptr;
    if (l > 0) __syncthreads();   // the previous layer's LDS is dead
    nb_layer_body<DH, G, NB>(al, p, mc, n_attn, xepoch, vthr_qkv, stage_off, vthr_gu, smem, xin, xout,
                             p.trace ? p.trace + (size_t)l * 256 * 16 : nullptr);
  }
}

struct NbLaunchCache { size_t configured = 0; int resident = -1; size_t resident_smem = 0; };
NbLaunchCache g_nb_cache[16][9];   // [device][NB]

template <int NB>
bool launch_nb(hipStream_t s, int device, PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, const PcyDecodeStepArgs& st,
               const unsigned* step_epoch, int n_cu, int xmin) {
  constexpr int DH = 128, G = 4, DS = NbGeom<NB>::DS, SLICES = NbGeom<NB>::SLICES;
  constexpr int n_attn = 64;   // (units beyond SLICES x Hkv x NB idle through the attention phase)
  static_assert(SLICES * 8 * NB <= 64, "attention units");
  a.o_sc1 = 0;
  a.xflags = (xmin > 0 && a.scratch && SLICES > 1) ? a.xflags : nullptr;
  a.xmin = xmin;
  a.unit_map = 1;
  const size_t stage_off = (attn_dec_smem_bytes(G, DS, DH, a.Tmax) + 15) & ~(size_t)15;
  const size_t smem_attn = stage_off + (size_t)(G + 2) * DH * 2, smem_body = (size_t)NB * 16384 + 4096;
  const size_t smem = smem_attn > smem_body ? smem_attn : smem_body;
  if (smem > 160 * 1024 || device < 0 || device >= 16) return false;
  NbLaunchCache& lc = g_nb_cache[device][NB];
  if (smem > 65536 && smem > lc.configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_step_nb_kernel<DH, G, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    lc.configured = smem;
  }
  // every workgroup waits for words the others write: all 256 must be resident at once
  if (lc.resident < 0 || lc.resident_smem != smem) {
    lc.resident = pcy_all_resident(decode_step_nb_kernel<DH, G, NB>, 512, smem, 256, n_cu) ? 1 : 0;
    lc.resident_smem = smem;
  }
  if (!lc.resident) return false;
  hipLaunchKernelGGL((decode_step_nb_kernel<DH, G, NB>), dim3(256), dim3(512), smem, s, a, p, mc, st, n_attn, step_epoch, pcy_gemv_rms_threads(p.Nq),
                     stage_off, pcy_gemv_rms_threads(mc.F));
  return true;
}

}  // namespace

// Words of tagged hand-over slots per layer / of the residual stream between two layers for an NB-row step
size_t pcy_decode_nb_tag_words(int NB) { return (size_t)NB * (NBF + NBNQ + NBD + NBD); }
size_t pcy_decode_nb_line_words(int NB) { return (size_t)256 * (((size_t)NB * 16 + 31) / 32 * 32); }
int pcy_decode_nb_ds(int B) { return B <= 1 ? 16 : B == 2 ? 32 : B <= 4 ? 64 : 128; }

// All decoder layers of a decode step for 2 <= B <= 8 rows in one launch.  Geometry: Llama-3-8B (d = 4096, ffn = 14336, 32 / 8 heads of
// 128), 256 CUs.  false = not covered, nothing launched.  st.tags / st.tag_stride / st.x_lines follow pcy_decode_nb_tag_words /
// pcy_decode_nb_line_words of B rows; p.epoch = the tag counter of THIS batch size's slots.
bool pcy_launch_decode_step_nb(hipStream_t s, int device, const PcyDecAttnArgs& a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc,
                               const PcyDecodeStepArgs& st, int n_cu, const unsigned* step_epoch, int B, int xmin) {
  if (B < 1 || B > 8 || a.B != B || a.dh != 128 || a.H != 32 || a.Hkv != 8 || a.dbg || n_cu < 256 || st.n_layers < 1) return false;
  if (p.d != NBD || p.Nq != NBNQ || mc.d != NBD || mc.F != NBF) return false;
#ifdef PCY_NB_ONLY   // (experiments: one instantiation)
  return B == PCY_NB_ONLY ? launch_nb<PCY_NB_ONLY>(s, device, a, p, mc, st, step_epoch, n_cu, xmin) : false;
#else
  switch (B) {
    case 1: return launch_nb<1>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
    case 2: return launch_nb<2>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
    case 3: return launch_nb<3>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
    case 4: return launch_nb<4>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
    case 5: return launch_nb<5>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
    case 6: return launch_nb<6>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
    case 7: return launch_nb<7>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
    default: return launch_nb<8>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
  }
#endif
}
