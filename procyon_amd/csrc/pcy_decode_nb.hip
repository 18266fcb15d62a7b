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
// one workgroup per CU here: the attention body requests the key tiles of four passes up front and the V rows of pass p + 1 before pass p is
// consumed (stand-alone launches with several workgroups per CU measured no gain from either; same rows, same order of the sums: same bits).
// 3 / 4 rows: 3.135 / 3.343 -> 3.086 / 3.272 ms per step
#define PCY_ATTN_DEC_VPF 1
#define PCY_ATTN_DEC_NP(DS) 4
#include "pcy_attn_dec.h"


namespace {

constexpr int NBD = 4096, NBF = 14336, NBWIN = NBF / 4, NBNQ = 6144;

template <int NB> struct NbGeom {
  static constexpr int DS = NB <= 1 ? 16 : NB == 2 ? 32 : 64;
  static constexpr int SLICES = 128 / DS;
  static constexpr int UNITS = SLICES * 8 * NB;   // attention units (row, kv head, column slice): <= 64 on the attention workgroups, the rest on the qkv-only ones
  static constexpr int XS = ((NB * 16 + 31) / 32) * 32;   // words of one workgroup's share of the residual stream between two layers
  static constexpr int P2 = NB <= 1 ? 1 : NB <= 2 ? 2 : NB <= 4 ? 4 : 8;
};

// Hand-over loads of this launch: ONE 16-byte sc1 buffer load per lane (the compiler counts it in vmcnt like any load) instead of ld16_agent's
// two 8-byte atomic loads -- half the instructions in the CU's memory queue per vector.  Every tagged vector of a step lives in one allocation
// (base); a word is validated by its own tag, so a load torn between two stores of a line is harmless.
struct NbBuf { __amdgpu_buffer_rsrc_t rs; const uint32_t* base; };
__device__ __forceinline__ uint4 nb_ld16(const NbBuf& nb, const void* p) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(nb.rs, (int)(reinterpret_cast<const char*>(p) - reinterpret_cast<const char*>(nb.base)), 0, 16 /* sc1 */);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
// mc_fetch_issue / mc_fetch_finish (pcy_handover.h) on these loads
template <int NV>
__device__ __forceinline__ void nb_fetch_issue(const NbBuf& nb, const uint32_t* src, int w0, int lane, uint4 (&pre)[NV]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) pre[j] = nb_ld16(nb, src + w0 + (j * 64 + lane) * 4);
}
template <int NV>
__device__ __forceinline__ void nb_fetch_finish(const NbBuf& nb, const uint32_t* src, int w0, int lane, uint32_t tag, bf16_t* dst, uint4 (&pre)[NV], unsigned* err,
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
    nb_fetch_issue<NV>(nb, src, w0, lane, pre);
  }
#pragma unroll
  for (int j = 0; j < NV; ++j)
    *reinterpret_cast<uint2*>(dst + w0 + (j * 64 + lane) * 4) =
        make_uint2((pre[j].x & 0xffffu) | (pre[j].y << 16), (pre[j].z & 0xffffu) | (pre[j].w << 16));
}

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
  constexpr int K = NBD, P2 = NbGeom<NB>::P2;
  const int tid = pcy_tid(), lane = tid & 63, wave = tid >> 6;
  const uint4 g = ldg16(w + tid * 8);
  // every row's values of this thread in registers first (independent LDS reads), then ONE transposing wave reduction for all rows: row by
  // row (six dependent exchanges each, a dependent LDS read per partial sum) the stage took 3.5 us at 4 rows and 6.5 us at 8 -- twice per layer
  uint4 xv[NB], xw[NB];
  const bool second = tid < vthr && (tid + vthr) * 8 < K;   // (K = 4096 <= 2 x vthr x 8 for every launch shape: at most two chunks per thread)
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    xv[b] = *reinterpret_cast<const uint4*>(x + b * K + tid * 8);
    xw[b] = second ? *reinterpret_cast<const uint4*>(x + b * K + (tid + vthr) * 8) : make_uint4(0, 0, 0, 0);
  }
  float ss[P2];
#pragma unroll
  for (int b = 0; b < P2; ++b) {
    float s = 0.f;
    if (b < NB) {
      const uint32_t w4[4] = {xv[b < NB ? b : 0].x, xv[b < NB ? b : 0].y, xv[b < NB ? b : 0].z, xv[b < NB ? b : 0].w};
      const uint32_t v4[4] = {xw[b < NB ? b : 0].x, xw[b < NB ? b : 0].y, xw[b < NB ? b : 0].z, xw[b < NB ? b : 0].w};
      if (tid < vthr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float f0 = lo_bf(w4[j]), f1 = hi_bf(w4[j]); s += f0 * f0 + f1 * f1; }
      }
      if (second) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float f0 = lo_bf(v4[j]), f1 = hi_bf(v4[j]); s += f0 * f0 + f1 * f1; }
      }
    }
    ss[b] = s;
  }
  const float tot = nb_wave_reduce<P2>(ss, lane);
  const int rb_ = nb_red_index<P2>(lane);
  lds_barrier();
  if (nb_red_owner<P2>(lane) && rb_ < NB) red[rb_ * 8 + wave] = tot;
  lds_barrier();
  const int nw = vthr >> 6;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(red + b * 8), r1 = *reinterpret_cast<const f32x4*>(red + b * 8 + 4);
    const float rr[8] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += i < nw ? rr[i] : 0.f;   // the first nw partial sums in order (+ 0 changes nothing)
    const float rs = rsqrtf(t / (float)K + eps);
    const uint32_t xin[4] = {xv[b].x, xv[b].y, xv[b].z, xv[b].w}, gin[4] = {g.x, g.y, g.z, g.w};
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
template <int NB, typename AfterIssue>
__device__ __forceinline__ void nb_fetch_vectors(const NbBuf& nbuf, const uint32_t* src, size_t sstride, int watch_wave, uint32_t tag, bf16_t* dst, unsigned* err, unsigned code,
                                                 AfterIssue after_issue) {
  const int tid_ = pcy_tid(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  // watch_wave < 0: no watching wave -- the caller knows the vector is about to be complete (the workgroups that have just stored their own part
  // of it), and a watch poll would only add a round trip in front of the fetch
  if (wave == watch_wave) {
    unsigned spins = 0;
    for (;;) {
      const uint4 v = nb_ld16(nbuf, src + (size_t)(NB - 1) * sstride + NBD - 256 + lane * 4);
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
      if (b0 + c < NB) nb_fetch_issue<2>(nbuf, src + (size_t)(b0 + c) * sstride, wave * 512, lane, t[c]);
    if (b0 + CH >= NB) after_issue();   // (weight prefetches go out BEHIND the last hand-over loads: a CU's loads return in order)
#pragma unroll
    for (int c = 0; c < CH; ++c)
      if (b0 + c < NB) nb_fetch_finish<2>(nbuf, src + (size_t)(b0 + c) * sstride, wave * 512, lane, tag, dst + (b0 + c) * NBD, t[c], err, code);
  }
  __syncthreads();
}

// The residual stream between two layers: workgroup g of the producing layer owns words [g * XS, g * XS + 16 NB) of `src`, element
// 16 g + i of row b at word g * XS + 16 b + i (a line has one writer for even NB).  -> dst [NB][4096] bf16.  Ends with a barrier.
template <int NB, typename AfterIssue>
__device__ __forceinline__ void nb_fetch_lines(const NbBuf& nbuf, const uint32_t* src, int watch_wave, uint32_t tag, bf16_t* dst, unsigned* err, unsigned code,
                                               AfterIssue after_issue) {
  constexpr int XS = NbGeom<NB>::XS;
  const int tid_ = pcy_tid(), lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  auto word_of = [&](int g, int b, int piece) __attribute__((always_inline)) { return g * XS + b * 16 + piece * 4; };
  if (wave == watch_wave) {   // the last 16 workgroups' words of the last row
    unsigned spins = 0;
    for (;;) {
      const uint4 v = nb_ld16(nbuf, src + word_of(256 - 16 + (lane >> 2), NB - 1, lane & 3));
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
            t[c][j] = nb_ld16(nbuf, src + word_of(wave * 32 + j * 16 + (lane >> 2), b0 + c, lane & 3));
            ok = ok && (t[c][j].x >> 16) == tag && (t[c][j].y >> 16) == tag && (t[c][j].z >> 16) == tag && (t[c][j].w >> 16) == tag;
          }
      if (b0 + 4 >= NB && spins == 0) after_issue();   // (weight rows go out BEHIND the last hand-over loads: a CU's loads return in order)
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

// first (which & 1) / second (which & 2) batch of UB k-iterations of this wave's first gate/up unit -> wa / wb: what nb_mlp_body expects to find
template <int UB>
__device__ __forceinline__ void nb_prime_gate_up(const PcyMlpChainArgs& a, int lane, int gidx, uint4 (&wa)[8 * UB], uint4 (&wb)[8 * UB], int which) {
  const McRowG row_g{a.F, a.d};
  if (which & 1) {
#pragma unroll
    for (int un = 0; un < UB; ++un)
#pragma unroll
      for (int i = 0; i < 8; ++i) wa[un * 8 + i] = ldg_nt(a.wgu + row_g(gidx, i) + (un * 64 + lane) * 8);
  }
  if (which & 2) {
#pragma unroll
    for (int un = 0; un < UB; ++un)
#pragma unroll
      for (int i = 0; i < 8; ++i) wb[un * 8 + i] = ldg_nt(a.wgu + row_g(gidx, i) + ((UB + un) * 64 + lane) * 8);
  }
}

// ------------------------------------------------------------------------------------------------
// The MLP of one layer for NB rows (every workgroup; `wg` = its index, 256 of them).  Entry: x after the o projection is on its way as
// the tagged vectors xo_tag [NB][4096]; wa / wb hold the first two batches (UB k-iterations of 8 rows each) of this wave's first gate/up
// unit (waves 0..6).  ra / rb: the two LDS regions [NB][4096]; misc: >= 2 KB behind them.
//
// gate/up: 7 waves x 2 units of 4 features, as at batch 1.
// down: K is split four ways.  A hand-over load returns behind everything its CU has in flight (~10 us under a saturated memory system),
// and with the whole of act needed by every workgroup the down stage was a chain of four such fetches (35 us for 117 MB at 4 rows).  Now
// workgroup group g = (wg % 8) / 2 -- two XCDs -- owns the 512-blocks {g + 4 j, j < 7} of K for ALL 4096 rows: 8 rows per wave, ONE fetch of
// NB x 3584 act values per workgroup (wave j takes block g + 4 j), partial sums handed to the workgroup that owns the row as {tag, fp32}
// words, which adds the four in group order, the residual, and publishes the row as before.  Per row:
//   p_g = wave_sum( sum over j of dot8 over block g + 4 j, in j order ) ;  x_out = bf16( bf16(((p0 + p1) + p2) + p3) + x )
// -- the arithmetic of gemv_kwin4_kernel (pcy_gemv.hip), the launch-per-stage twin.
template <int NB, int UB>
__device__ __forceinline__ void nb_mlp_body(const NbBuf& nbuf, const PcyMlpChainArgs& a, const uint32_t* xo_tag, unsigned long long* part, char* smem, int vthr_gu, uint32_t tag,
                                            int wg, uint4 (&wa)[8 * UB], uint4 (&wb)[8 * UB], unsigned long long* tr, uint32_t* x_out_lines, bool prime_second) {
  constexpr int d = NBD, F = NBF, XS = NbGeom<NB>::XS, P2 = NbGeom<NB>::P2;
  constexpr int GB = 8 / UB;                    // batches per gate/up unit
  constexpr int DB = (7 + UB - 1) / UB;         // batches of the down stage (7 k-iterations)
  bf16_t* ra = reinterpret_cast<bf16_t*>(smem);
  bf16_t* rb = ra + NB * d;
  float* red = reinterpret_cast<float*>(rb + NB * d);                 // [NB][8]
  uint32_t* line = reinterpret_cast<uint32_t*>(red + NB * 8);         // [NB][32]
  bf16_t* xres = reinterpret_cast<bf16_t*>(line + NB * 32);           // [NB][16] this workgroup's rows of x (the epilogue's residual)
  const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#define NB_T(i) if (tr && tid == 0) tr[i] = wall_clock64();
  // prime_second (workgroup-uniform): the second gate/up batch of the wave's first unit is requested here, behind the loads that fetch x
  nb_fetch_vectors<NB>(nbuf, xo_tag, d, (prime_second && NB <= 3) ? -1 : 7, tag, rb, a.err, 13u, [&]() __attribute__((always_inline)) {
    if (prime_second && wave < 7) nb_prime_gate_up<UB>(a, lane, wg * 7 + wave, wa, wb, 2);
  });
  NB_T(8)
  if (tid < NB * 16) xres[tid] = rb[(tid >> 4) * d + wg * 16 + (tid & 15)];
  const int kg = (wg & 7) >> 1, kq = (wg >> 3) * 2 + (wg & 1);       // group (an XCD pair), index in the group
  nb_rms_stage<NB>(rb, a.ln2, vthr_gu, a.rms_eps, a.rms_cast, ra, red);
  // down: this workgroup's K blocks and this wave's 8 rows
  const int dr0 = (kq * 8 + wave) * 8;
  auto issue_down = [&](int t, uint4 (&w)[8 * UB]) __attribute__((always_inline)) {   // batch t = k-iterations [t UB, t UB + UB) of the 7
#pragma unroll
    for (int un = 0; un < UB; ++un)
      if (t * UB + un < 7) {
#pragma unroll
        for (int i = 0; i < 8; ++i) w[un * 8 + i] = ldg_nt(a.wdown + (size_t)(dr0 + i) * F + ((kg + 4 * (t * UB + un)) * 64 + lane) * 8);
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
    auto issue = [&](int t, uint4 (&w)[8 * UB]) __attribute__((always_inline)) {      // batch t of the 2 GB batches of the two units
      const int u = t < GB ? u1 : u2, it0 = (t < GB ? t : t - GB) * UB;
#pragma unroll
      for (int un = 0; un < UB; ++un)
#pragma unroll
        for (int i = 0; i < 8; ++i) w[un * 8 + i] = ldg_nt(a.wgu + row_g(u, i) + ((it0 + un) * 64 + ln) * 8);
    };
    auto compute = [&](int t, const uint4 (&w)[8 * UB]) __attribute__((always_inline)) {
      const int it0 = (t < GB ? t : t - GB) * UB;
#pragma unroll
      for (int un = 0; un < UB; ++un)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const uint4 xv = *reinterpret_cast<const uint4*>(ra + b * d + ((it0 + un) * 64 + ln) * 8);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i][b] = dot8(w[un * 8 + i], xv, acc[i][b]);
        }
    };
#pragma unroll 1
    for (int pr = 0; pr < GB - 1; ++pr) {       // pairs of batches; wa / wb hold batches 2 pr, 2 pr + 1 on entry
      asm volatile("" : "+v"(ln));
      compute(2 * pr, wa);
      issue(2 * pr + 2, wa);
      compute(2 * pr + 1, wb);
      if (2 * pr + 1 == GB - 1) finish(u1);
      issue(2 * pr + 3, wb);
    }
    asm volatile("" : "+v"(ln));
    compute(2 * GB - 2, wa);
    compute(2 * GB - 1, wb);
    finish(u2);
    NB_T(9)
  }
  // ---- down ----
  // (one piece of code for all eight waves: requests inside the branch above and in an else-branch for wave 7 met in 32 four-register
  // copies and the allocator spilled both batches)
  // Wave j takes block kg + 4 j of act, all rows -- and NO down row is requested before the window has arrived.  A hand-over load is
  // served behind whatever the chip has in flight (in-kernel stamps at 4 rows: a block that had been complete for 10 us took 11 us to
  // fetch while the slower workgroups' last gate/up batches and the faster ones' first down batches were in the queues, and the
  // producers' tagged stores wait in the same queues), so the 32 KB per wave that used to be requested first cost more in the hop than
  // they saved behind it: both batches in front 3.24 / 4.10 ms per step at 4 / 8 rows, one 3.19 / 4.02, none 3.18 / 3.99.  (Where the
  // second-half blocks were requested -- in front of, between or behind the weight batches -- made no difference; longer sleeps between
  // the polls cost 0.5 ... 15 %; touching the blocks' pages ahead changed nothing.)
  uint4 tq[NB][2];
  if (wave < 7) {
#pragma unroll
    for (int b = 0; b < NB; ++b) nb_fetch_issue<2>(nbuf, a.act_tag + (size_t)b * F + (kg + 4 * wave) * 512, 0, lane, tq[b]);
  }
  NB_T(5)
  if (wave < 7) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
      nb_fetch_finish<2>(nbuf, a.act_tag + (size_t)b * F + (kg + 4 * wave) * 512, 0, lane, tag, rb + b * NBWIN + wave * 512, tq[b], a.err, 20u);
    NB_T(6)
    if (tr && wave == 6 && lane == 0) tr[7] = wall_clock64();
  }
  issue_down(0, wa);
  issue_down(1, wb);
  lds_barrier();   // this group's act values are in rb
  NB_T(10)
  {
    float dacc[8][NB];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) dacc[i][b] = 0.f;
    auto dcompute = [&](int t, const uint4 (&w)[8 * UB]) __attribute__((always_inline)) {
#pragma unroll
      for (int un = 0; un < UB; ++un)
        if (t * UB + un < 7) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const uint4 xv = *reinterpret_cast<const uint4*>(rb + b * NBWIN + ((t * UB + un) * 64 + lane) * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) dacc[i][b] = dot8(w[un * 8 + i], xv, dacc[i][b]);
          }
        }
      // the sums are pinned here: otherwise the dot products of ALL batches sink to the end of the function (their results are only used
      // there) and every batch of weights stays live -- spills
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(dacc[i][b]));
    };
#pragma unroll
    for (int t = 0; t < DB; t += 2) {
      dcompute(t, wa);
      if (t + 2 < DB) issue_down(t + 2, wa);
      if (t + 1 < DB) {
        dcompute(t + 1, wb);
        if (t + 3 < DB) issue_down(t + 3, wb);
      }
    }
    NB_T(11)
    // partial sums of this wave's 8 rows x NB -> the rows' owners: word (row, kg, b) = {fp32 : tag}
    constexpr int NP = 8 * P2;
    float v[NP];
#pragma unroll
    for (int idx = 0; idx < NP; ++idx) v[idx] = (idx >> 3) < NB ? dacc[idx & 7][(idx >> 3) < NB ? (idx >> 3) : 0] : 0.f;
    const float tot = nb_wave_reduce<NP>(v, lane);
    const int idx = nb_red_index<NP>(lane), b = idx >> 3, i = idx & 7;
    if (nb_red_owner<NP>(lane) && b < NB)
      __hip_atomic_store(part + ((size_t)(dr0 + i) * 4 + kg) * NB + b, ((unsigned long long)__float_as_uint(tot) << 32) | tag, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- x_out = x + act . Wdown^T: rows [16 wg, 16 wg + 16) of every batch row, thread (b, e) ----
  if (tid < NB * 16) {
    const int b = tid >> 4, e = tid & 15;
    const unsigned long long* src = part + ((size_t)(wg * 16 + e) * 4) * NB + b;
    unsigned long long pw[4];
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        pw[g] = __hip_atomic_load(src + (size_t)g * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = ok && (uint32_t)pw[g] == tag;
      }
      if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
      if (pcy_wait_give_up(spins, 1u << 19, a.err, 24u, lane)) break;
      __builtin_amdgcn_s_sleep(2);
    }
    float r = __uint_as_float((uint32_t)(pw[0] >> 32)) + __uint_as_float((uint32_t)(pw[1] >> 32));
    r += __uint_as_float((uint32_t)(pw[2] >> 32));
    r += __uint_as_float((uint32_t)(pw[3] >> 32));
    r = rbf(r);
    r = rbf(r + bf2f(xres[tid]));
    if (x_out_lines) __hip_atomic_store(x_out_lines + wg * XS + tid, (tag << 16) | f2bf(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else a.x_out[(size_t)b * d + wg * 16 + e] = f2bf(r);
  }
  NB_T(12)
#undef NB_T
}

// ------------------------------------------------------------------------------------------------
// One decoder layer for NB rows (see the head of the file).  x_in_lines == nullptr: the layer's input is p.x [NB][4096] in global memory
// (written before the launch); x_out_lines == nullptr: the result goes to mc.x_out [NB][4096].
template <int DH, int G, int NB, int UB>
__device__ __forceinline__ void nb_layer_body(const NbBuf& nbuf, PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, unsigned long long* part, int n_attn,
                                              unsigned xepoch, int vthr_qkv, size_t stage_off, int vthr_gu, char* smem, const uint32_t* x_in_lines,
                                              uint32_t* x_out_lines, unsigned long long* tr_base) {
  constexpr int DS = NbGeom<NB>::DS, SLICES = NbGeom<NB>::SLICES, P2 = NbGeom<NB>::P2, UNITS = NbGeom<NB>::UNITS;
  constexpr int d = NBD, K = NBD;
  const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t tag = *p.epoch & 0xffffu;
  unsigned long long* tr = tr_base ? tr_base + (size_t)blockIdx.x * 16 : nullptr;
#define NB_T(i) if (tr && tid == 0) tr[i] = wall_clock64();
  NB_T(0)
  const int wg = (int)blockIdx.x;
  int unit = wg;
  if (wg >= n_attn) {
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
    uint4 w[32];   // qkv rows [r0, r0 + 4): 16 KB of this wave requested in front of the wait for x, 16 KB behind the loads that fetch it
    auto load_rows = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
      for (int i = i0; i < i0 + 2; ++i)
#pragma unroll
        for (int it = 0; it < 8; ++it) w[i * 8 + it] = ldg_nt(p.wqkv + (size_t)(r0 + i) * d + (it * 64 + lane) * 8);
    };
    load_rows(0);
    if (x_in_lines) {
      // (all 32 KB in front: B = 4 / 8 at 3.41 / 4.18 ms per step; all behind: 3.36 / 4.11; half and half: 3.32 / 4.16 -- and again
      // after the down stage's change: half and half 3.18 / 3.99, all behind 3.23 / 4.07)
      nb_fetch_lines<NB>(nbuf, x_in_lines, 7, tag, rb, p.err, 14u, [&]() __attribute__((always_inline)) { load_rows(2); });
    } else {
      load_rows(2);
#pragma unroll
      for (int b = 0; b < NB; ++b) *reinterpret_cast<uint4*>(rb + b * d + tid * 8) = ldg16(p.x + (size_t)b * d + tid * 8);
      __syncthreads();
    }
    NB_T(13)
    nb_rms_stage<NB>(rb, p.ln1, vthr_qkv, p.rms_eps, p.rms_cast, ra, red);
    NB_T(14)
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
    NB_T(15)
  }
  __syncthreads();   // (also: every wave is done with RMSNorm(x) in ra)
  if (wave < NB && lane < 32)
    __hip_atomic_store(p.qkv_tag + (size_t)wave * NBNQ + (r0 & ~31) + lane, line[wave * 32 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  NB_T(1)
  if (active) {
    uint4 wa[8 * UB], wb[8 * UB];
    // the first gate/up batch of the MLP while the attention runs
    if (wave < 7) nb_prime_gate_up<UB>(mc, lane, wg * 7 + wave, wa, wb, 1);
    {
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
        const uint4 v = nb_ld16(nbuf, p.ao_tag + (size_t)bw * K + (lane / NB) * 4);
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
        if (b0 + c < NB) nb_fetch_issue<2>(nbuf, p.ao_tag + (size_t)(b0 + c) * K, wave * 512, lane, t[c]);
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (b0 + c < NB) nb_fetch_finish<2>(nbuf, p.ao_tag + (size_t)(b0 + c) * K, wave * 512, lane, tag, ra + (b0 + c) * K, t[c], p.err, 11u);
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
    __syncthreads();                                     // every wave is done with this phase's LDS
    // (the second batch of this wave's gate/up rows is requested inside, behind the fetch of the residual stream)
    nb_mlp_body<NB, UB>(nbuf, mc, p.xo_tag, part, smem, vthr_gu, tag, wg, wa, wb, tr, x_out_lines, true);
    return;
  }
    // qkv rows only: an attention unit of the second set (NB >= 5), or none
    unit = UNITS > 64 ? 64 + (wg - n_attn - d / 32) : -1;
    __syncthreads();                                   // the stores above have read `line`; the attention's LDS starts at 0
  }
  {
    // ---- attention of one (row, kv head, column slice): units [0, 64) on the attention workgroups, units [64, 128) (NB >= 5) on the
    // projection workgroups that have qkv rows only, behind their rows; a workgroup without a unit goes straight to the MLP ----
    uint4 wa[8 * UB], wb[8 * UB];
    const int kvh = unit % a.Hkv, bx = (unit / a.Hkv) % SLICES, b = unit / (a.Hkv * SLICES);   // kv head in the low digits: the slices of a head share an XCD's L2
    if (unit >= 0 && b < NB) {
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
            if (mine) v = nb_ld16(nbuf, qt + w0);
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
    if (wave < 7) nb_prime_gate_up<UB>(mc, lane, wg * 7 + wave, wa, wb, 3);   // two batches per wave while x is on its way
    __syncthreads();                                   // the attention's LDS is dead
    nb_mlp_body<NB, UB>(nbuf, mc, p.xo_tag, part, smem, vthr_gu, tag, wg, wa, wb, tr, x_out_lines, false);
    return;
  }
#undef NB_T
}

template <int DH, int G, int NB, int UB>
__global__ __launch_bounds__(512) void decode_step_nb_kernel(PcyDecAttnArgs a, PcyAttnBlockArgs p, PcyMlpChainArgs mc, PcyDecodeStepArgs st, int n_attn,
                                                             const unsigned* step_epoch, int vthr_qkv, size_t stage_off, int vthr_gu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned xepoch = *step_epoch;
  const NbBuf nbuf{__builtin_amdgcn_make_buffer_rsrc((void*)st.tags, 0, 0x7fffffff, 0x00020000), st.tags};
  for (int l = 0; l < st.n_layers; ++l) {
    const PcyLayerWeightsDev lw = st.layers[l];
    p.ln1 = lw.ln1; p.wqkv = lw.wqkv; p.wo = lw.wo;
    mc.ln2 = lw.ln2; mc.wgu = lw.wgu; mc.wdown = lw.wdown;
    PcyDecAttnArgs al = a;
    al.kcache = a.kcache + (size_t)l * st.kv_layer_stride; al.vcache = a.vcache + (size_t)l * st.kv_layer_stride;
    al.xflags = a.xflags ? a.xflags + (size_t)l * st.xflags_stride : nullptr;
    uint32_t* tags = st.tags + (size_t)l * st.tag_stride;   // act [NB][F] | qkv [NB][Nq] | attention output [NB][H dh] | x after o [NB][d] | down partials [d][4][NB] x 8 B
    mc.act_tag = tags; p.qkv_tag = tags + (size_t)NB * NBF; p.ao_tag = p.qkv_tag + (size_t)NB * NBNQ; p.xo_tag = p.ao_tag + (size_t)NB * NBD;
    unsigned long long* part = reinterpret_cast<unsigned long long*>(p.xo_tag + (size_t)NB * NBD);
    const uint32_t* xin = l > 0 ? st.x_lines + (size_t)(l - 1) * st.x_lines_stride : nullptr;
    uint32_t* xout = l + 1 < st.n_layers ? st.x_lines + (size_t)l * st.x_lines_stride : nullptr;
    if (l > 0) __syncthreads();   // the previous layer's LDS is dead
    nb_layer_body<DH, G, NB, UB>(nbuf, al, p, mc, part, n_attn, xepoch, vthr_qkv, stage_off, vthr_gu, smem, xin, xout,
                             p.trace ? p.trace + (size_t)l * 256 * 16 : nullptr);
  }
}

struct NbLaunchCache { size_t configured = 0; int resident = -1; size_t resident_smem = 0; };
NbLaunchCache g_nb_cache[16][9][2];   // [device][NB][UB - 1]

template <int NB, int UB>
bool launch_nb_ub(hipStream_t s, int device, PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, const PcyDecodeStepArgs& st,
               const unsigned* step_epoch, int n_cu, int xmin) {
  constexpr int DH = 128, G = 4, DS = NbGeom<NB>::DS, SLICES = NbGeom<NB>::SLICES;
  constexpr int n_attn = 64;   // (units beyond SLICES x Hkv x NB idle through the attention phase)
  static_assert(SLICES * 8 * NB <= 128, "attention units");
  a.o_sc1 = 0;
  a.xflags = (xmin > 0 && a.scratch && SLICES > 1) ? a.xflags : nullptr;
  a.xmin = xmin;
  a.unit_map = 1;
  const size_t stage_off = (attn_dec_smem_bytes(G, DS, DH, a.Tmax) + 15) & ~(size_t)15;
  const size_t smem_attn = stage_off + (size_t)(G + 2) * DH * 2, smem_body = (size_t)NB * 16384 + 4096;
  const size_t smem = smem_attn > smem_body ? smem_attn : smem_body;
  if (smem > 160 * 1024 || device < 0 || device >= 16) return false;
  NbLaunchCache& lc = g_nb_cache[device][NB][UB - 1];
  if (smem > 65536 && smem > lc.configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_step_nb_kernel<DH, G, NB, UB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    lc.configured = smem;
  }
  // every workgroup waits for words the others write: all 256 must be resident at once
  if (lc.resident < 0 || lc.resident_smem != smem) {
    lc.resident = pcy_all_resident(decode_step_nb_kernel<DH, G, NB, UB>, 512, smem, 256, n_cu) ? 1 : 0;
    lc.resident_smem = smem;
  }
  if (!lc.resident) return false;
  if (st.n_layers == 0) return true;   // (launchability query: pcy_decode_nb_launchable)
  hipLaunchKernelGGL((decode_step_nb_kernel<DH, G, NB, UB>), dim3(256), dim3(512), smem, s, a, p, mc, st, n_attn, step_epoch, pcy_gemv_rms_threads(p.Nq),
                     stage_off, pcy_gemv_rms_threads(mc.F));
  return true;
}

// UB = k-iterations per weight batch of the MLP streams (two batches in flight per wave: 16 or 32 KB); PCY_NB_UB=1|2 overrides (measurement)
template <int NB>
bool launch_nb(hipStream_t s, int device, PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, const PcyDecodeStepArgs& st,
               const unsigned* step_epoch, int n_cu, int xmin) {
  const char* e = getenv("PCY_NB_UB");
  const int ub = e ? atoi(e) : 2;
  if (ub == 1) return launch_nb_ub<NB, 1>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
  return launch_nb_ub<NB, 2>(s, device, a, p, mc, st, step_epoch, n_cu, xmin);
}

}  // namespace

// Words of tagged hand-over slots per layer / of the residual stream between two layers for an NB-row step
size_t pcy_decode_nb_tag_words(int NB) { return (size_t)NB * (NBF + NBNQ + NBD + NBD) + (size_t)NBD * 4 * NB * 2; }
size_t pcy_decode_nb_line_words(int NB) { return (size_t)256 * (((size_t)NB * 16 + 31) / 32 * 32); }
int pcy_decode_nb_ds(int B) { return B <= 1 ? 16 : B == 2 ? 32 : 64; }

// All decoder layers of a decode step for 2 <= B <= 8 rows in one launch.  Geometry: Llama-3-8B (d = 4096, ffn = 14336, 32 / 8 heads of
// 128), 256 CUs.  false = not covered, nothing launched.  st.tags / st.tag_stride / st.x_lines follow pcy_decode_nb_tag_words /
// pcy_decode_nb_line_words of B rows; p.epoch = the tag counter of THIS batch size's slots.
bool pcy_launch_decode_step_nb(hipStream_t s, int device, const PcyDecAttnArgs& a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc,
                               const PcyDecodeStepArgs& st, int n_cu, const unsigned* step_epoch, int B, int xmin) {
  if (B < 1 || B > 8 || a.B != B || a.dh != 128 || a.H != 32 || a.Hkv != 8 || a.dbg || n_cu < 256 || st.n_layers < 0) return false;
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

// Would pcy_launch_decode_step_nb launch for this batch size and cache capacity on this device?  (LDS of the attention phase grows with Tmax;
// every workgroup must be resident.)  The engine asks BEFORE it commits a step to the small-batch arithmetic: an uncovered shape takes the
// round-4 launches (MFMA GEMVs), not the twin's forced streaming kernels, and no hand-over counter is advanced for a step that never runs.
bool pcy_decode_nb_launchable(int device, int B, int Tmax, int n_cu) {
  PcyDecAttnArgs a{};
  a.B = B; a.dh = 128; a.H = 32; a.Hkv = 8; a.Tmax = Tmax;
  PcyAttnBlockArgs p{};
  p.d = NBD; p.Nq = NBNQ;
  PcyMlpChainArgs mc{};
  mc.d = NBD; mc.F = NBF;
  PcyDecodeStepArgs st{};
  st.n_layers = 0;
  return pcy_launch_decode_step_nb(nullptr, device, a, p, mc, st, n_cu, nullptr, B, 0);
}
