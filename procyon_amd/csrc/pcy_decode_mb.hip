// Mid-batch decode step (9..32 rows: beam 10 / 20 of the reference's callers, the 32-row batch of BASELINE configs[3]) as ONE launch: every
// decoder layer of a step, the weights streamed ONCE.  Reference: the decode step of model_unified.py:769,887 at batch = beam_size
// (scripts/caption_bulk.py:193-194 beam 10, evaluate/framework/procyon.py:72-76 beam 10 per input, the notebook's beam 20).
//
// The launch-per-stage step of these batch sizes is seven launches per layer (qkv GEMV, attention, o GEMV, finish + norm, gate/up GEMV, down
// GEMV, finish + norm): at every boundary the weight stream stops (~1.7 us + the ramp of the next launch), and the small kernels (o: 5 us of
// bytes) never reach the streaming rate at all.  Here the SAME work items run as phases of one persistent kernel:
//
//   256 workgroups x 512 threads, one per CU, all resident.  Per layer
//     Q  every workgroup: item (48 rows of Wqkv, K half)          -> fp32 partial sums                 (gemv_mfma4_kernel, ksplit 2)
//     A  workgroups [0, units): decode attention of one (row, kv head, DS columns) unit, q / k / v added up from the partial sums
//                                                                                                        (attn_dec_splitk_kernel)
//     O  every workgroup: item (64 rows of Wo, K quarter)         -> fp32 partial sums                 (gemv_mfma4_kernel, ksplit 4)
//     F1 workgroups [224, 224 + B): row b: x += sum of the quarters, xn = RMSNorm(x) * ln2             (gemv_splitk_finish_norm_kernel)
//     G  workgroups [0, 224): item (128 interleaved gate/up rows) -> act = SwiGLU                      (gemv_mfma4_kernel<EPI_SWIGLU>)
//     D  every workgroup: item (64 rows of Wdown, K quarter)      -> fp32 partial sums                 (gemv_mfma4_kernel, ksplit 4)
//     F2 workgroups [192, 192 + B): row b: x += sum, xn = RMSNorm(x) * (next layer's ln1 | final norm)
//   Every item is computed with the arithmetic of the stand-alone launch named on the right (same K ranges, same MFMA order, same order of
//   the split sums, same rounding points): the launch is BIT-IDENTICAL to the launch-per-stage step (PCY_DISABLE=decode_mb_step; tests).
//
//   Waves 0..3 of a workgroup are the GEMV waves: each streams ITS 16 (gate/up: 2 x 16) weight rows as 4 KB tiles [16 rows][256 B] by LDS-DMA
//   into a private ring of 7 tiles that is kept FULL at all times (a step refills the slots it has just read) -- the tile sequence of a wave runs through the phases and the
//   layers (Q, O, G, D, next layer's Q, ...) and does not stop at a phase boundary: while a workgroup waits for its inputs, publishes its
//   results or runs the attention, the rings fill with the next items' weights (the finishers F1 / F2 are workgroups WITHOUT an item in the
//   phase that follows, so nothing of theirs queues in front of the partial sums they fetch).
//   Wave 4 is the x loader: it watches the arrival flags of the phase's inputs, then copies the activations tile by tile (LDS-DMA, sc1: from
//   L2, never from a stale L1 line) into a ring shared by the four GEMV waves, SX - 1 steps ahead.  One barrier per 128-k step.
//   Waves 4..7 also relay the epilogues: the GEMV waves put their results into LDS, waves 4..7 store them (written through, sc1), drain and
//   raise the item's flag -- a GEMV wave never waits for its own stores, so it never drains the tiles it has in flight.
//
//   Hand-overs (CDNA4 guide, Guideline 16 R1): payload written through (sc1 stores), every storing wave drains, ONE flag word per item
//   (value = the step's epoch: a device counter advanced once per step, so nothing is ever re-zeroed); a consumer polls the flag words of
//   its producers with ONE 16-byte agent-scope load per lane and then reads the payload with sc1 loads.  Every wait is bounded (watchdog).
//
// LDS: 4 weight rings of 7 (6) tiles + 48 (64) KB shared (x ring | attention | finisher scratch) = 160 KB at <= 16 (<= 32) rows.
#include <stdlib.h>
#include "pcy_internal.h"
// (one workgroup per CU, as in the small-batch step: key tiles of four passes up front, V rows one pass ahead; same rows, same sums)
#define PCY_ATTN_DEC_VPF 1
#define PCY_ATTN_DEC_NP(DS) 4
#include "pcy_attn_dec.h"

namespace {

typedef __attribute__((address_space(3))) void* mb_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* mb_gptr_t;

constexpr int MBD = 4096, MBF = 14336, MBNQ = 6144;
// LDS split by batch tiles.  A CU's loads return in issue order, so the x tile of step s + n must be requested no later than the weight tiles of
// step s + n: the x ring has to look as many STEPS ahead as the weight rings.  16 rows (BT 1): 7 weight tiles per wave + 8 x steps of 4 KB;
// 32 rows (BT 2): 6 weight tiles per wave + 8 x steps of 8 KB (with 7 + 6 steps the O / D phases ran at the x loads' pace: 1.2 us per step).
template <int BT> struct MbCfg {
  static constexpr int RING = BT == 2 ? 6 : 7;          // tiles of a GEMV wave's ring; kept full (a step refills the slots it has just read)
  static constexpr int WBYTES = 4 * RING * 4096;        // 96 / 112 KB
  static constexpr int XBYTES = 160 * 1024 - WBYTES;    // 64 / 48 KB: x ring | attention | finisher scratch
  static constexpr int SX = 8;                          // x steps in the ring (12 at <= 16 rows measured the same: the single-tile phases are
                                                        // paced by two memory round trips -- 16 (Q) / 8 (O) tiles per wave against a ring of 7)
};
constexpr int MB_SMEM = 160 * 1024;
#ifndef MB_NLW
#define MB_NLW 1   // x loader waves (1: wave 4 alone; 4 measured slower: 4.1 -> 4.6 ms per 10-row step, 5.1 -> 5.8 at 32 rows)
#endif
// arrival flags of one layer (words; every group starts on a 256-byte line)
constexpr int MBF_XN = 0;      // [32]      F2 of the previous layer: row b's xn is there
constexpr int MBF_QKV = 64;    // [256]     Q items
constexpr int MBF_AO = 320;    // [4][64]   per K quarter of Wo: the attention units of its two kv heads
constexpr int MBF_O = 576;     // [256]     O items
constexpr int MBF_XN2 = 832;   // [32]      F1: row b's xn is there
constexpr int MBF_ACT = 896;   // [4][64]   per K quarter of Wdown: the 56 G items that produce its act columns
constexpr int MBF_DN = 1152;   // [256]     D items
constexpr int MBF_STRIDE = 1408;
constexpr int MB_G_WGS = 224, MB_F1_WG0 = 224, MB_F2_WG0 = 192;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mb_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); }
__device__ __forceinline__ uint4 mb_ld16(__amdgpu_buffer_rsrc_t rs, int byte_off) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16 /* sc1 */);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void mb_st16(__amdgpu_buffer_rsrc_t rs, int byte_off, const uint4& v) {
  const u32x4_t d = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, 16 /* sc1: written through */);
}

// ONE wave: until the `count` flag words at f (16-byte aligned) all hold `epoch`.  count <= 256.
__device__ __forceinline__ void mb_wait_flags(const unsigned* f, int count, unsigned epoch, unsigned* err, unsigned code, int lane) {
  const __amdgpu_buffer_rsrc_t rs = mb_rsrc(f);
  const int i4 = lane * 4;
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
    if (i4 < count) {
      const uint4 v = mb_ld16(rs, i4 * 4);
      ok = v.x == epoch && (i4 + 1 >= count || v.y == epoch) && (i4 + 2 >= count || v.z == epoch) && (i4 + 3 >= count || v.w == epoch);
    }
    if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
    if (pcy_wait_give_up(spins, 1u << 19, err, code, lane)) break;
    __builtin_amdgcn_s_sleep(2);
  }
}
__device__ __forceinline__ void mb_raise(unsigned* f, unsigned epoch) { __hip_atomic_store(f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Per-wave state of the weight stream (GEMV waves only; every field wave-uniform).
struct MbStream {
  const PcyLayerWeightsDev* layers;
  int n_layers, wg, wave;
  char* ring;                 // this wave's tiles
  int il, iph, itile, intiles, iK, irt2, islot, idone;
  int inss, ishift;           // 128-k steps of the phase's K range and the step the workgroup starts at (pcy_gemv_kshift: rotated K order)
  const bf16_t* ibase;        // first element of the phase's item for this wave: W + r0 * K + kbeg
  int cslot;                  // ring slot of the next tile to be consumed
  int wdef;                   // timing ablation: weight copies with the default cache policy instead of nt
};
// Q: 384 row tiles x 2 K halves = 768 wave items = 3 per workgroup (waves 0..2: tiles 3 (wg >> 1) + wave of half wg & 1) -- every CU streams in
// every phase but G (896 gate/up pairs = 3.5 per workgroup: 224 workgroups of 4).  A CU draws ~25 GB/s from HBM whatever it has in flight, so a
// phase is as long as its fullest CU.
__device__ __forceinline__ bool mb_has_item(int wg, int wave, int ph) { return ph == 0 ? wave < 3 : ph == 2 ? wg < MB_G_WGS : true; }
__device__ __forceinline__ void mb_stream_setup(MbStream& s) {
  const PcyLayerWeightsDev lw = s.layers[s.il];
  const int wg = s.wg, wave = s.wave;
  s.itile = 0;
  int r0wg;   // first weight row of the workgroup's item (all of its waves share the x tiles, hence the rotation)
  if (s.iph == 0) { r0wg = (wg >> 1) * 48; s.ibase = lw.wqkv + (size_t)(r0wg + wave * 16) * MBD + (wg & 1) * 2048; s.intiles = 16; s.iK = MBD; s.irt2 = 0; }
  else if (s.iph == 1) { r0wg = (wg >> 2) * 64; s.ibase = lw.wo + (size_t)(r0wg + wave * 16) * MBD + (wg & 3) * 1024; s.intiles = 8; s.iK = MBD; s.irt2 = 0; }
  else if (s.iph == 2) { r0wg = wg * 128; s.ibase = lw.wgu + (size_t)(r0wg + wave * 32) * MBD; s.intiles = 64; s.iK = MBD; s.irt2 = 1; }
  else { r0wg = (wg >> 2) * 64; s.ibase = lw.wdown + (size_t)(r0wg + wave * 16) * MBF + (wg & 3) * 3584; s.intiles = 28; s.iK = MBF; s.irt2 = 0; }
  s.inss = s.irt2 ? s.intiles >> 1 : s.intiles;
  s.ishift = pcy_gemv_kshift(r0wg, s.inss);
}
__device__ __forceinline__ void mb_stream_advance(MbStream& s) {   // the phase's last tile has been issued: the next phase that has an item
  for (;;) {
    if (++s.iph == 4) { s.iph = 0; ++s.il; }
    if (s.il >= s.n_layers) { s.idone = 1; s.itile = s.intiles - 1; return; }   // behind the last layer: the last tile again (the counts stay uniform)
    if (mb_has_item(s.wg, s.wave, s.iph)) break;
  }
  mb_stream_setup(s);
}
// one tile [16 rows][256 B] -> the ring; voff = this lane's element offset inside a tile for K = 4096 / 14336 (4 x 4 rows of 256 B, the 16-byte
// pieces of a row XOR-swizzled by the row on the source side: gemv_mfma4_kernel's layout)
template <int RING>
__device__ __forceinline__ void mb_stream_issue(MbStream& s, const int (&voff4)[4], const int (&voff14)[4]) {
  int ss = (s.irt2 ? s.itile >> 1 : s.itile) + s.ishift;
  ss = ss >= s.inss ? ss - s.inss : ss;
  const bf16_t* p = s.ibase + (s.irt2 ? (size_t)(s.itile & 1) * 16 * s.iK : (size_t)0) + (size_t)ss * 128;
  char* dst = s.ring + s.islot * 4096;
  const bool k4 = s.iK == MBD;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (s.wdef) __builtin_amdgcn_global_load_lds((mb_gptr_t)(p + (k4 ? voff4[q] : voff14[q])), (mb_lds_ptr_t)(dst + q * 1024), 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((mb_gptr_t)(p + (k4 ? voff4[q] : voff14[q])), (mb_lds_ptr_t)(dst + q * 1024), 16, 0, 2 /* nt */);
  s.islot = s.islot + 1 == RING ? 0 : s.islot + 1;
  if (!s.idone && ++s.itile == s.intiles) mb_stream_advance(s);
}

struct MbCtx {
  const PcyMbArgs* a;
  char* smem;
  char* xreg;                 // the shared 48 KB
  int wg;
  unsigned epoch;
  unsigned* lflags;           // this layer's flags
  unsigned long long* tr;     // this (layer, workgroup)'s stamps or nullptr
};
#define MB_T(i) if (c.tr && threadIdx.x == 0) c.tr[i] = wall_clock64();
// thread / lane / wave of the caller, through an optimisation barrier: every phase starts from a fresh value, so that nothing lane-derived of a
// phase (x sources, fragment addresses) is hoisted out of the layer loop and kept live across all the other phases (spills)
#define MB_IDS const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); (void)lane; (void)wave;

// One GEMV phase of a workgroup that has an item in it.  RT row tiles per wave and step, BT batch tiles; nsteps 128-k steps.
//   x: first element of the item's K range in row 0 of the activations [B][ldx]
//   EPI 0: the fp32 sums of rows [n0 + 16 wave, +16) go to ws_out [B][N] (this K split's slab); EPI 1: SwiGLU of the wave's gate / up
//   tiles -> act [B][ffn], features [16 (4 wg + wave), +16)
//   wait_f / wait_n: flags of the producers of x (nullptr: x was written before the launch); done_f: this item's flag
template <int RT, int BT, int EPI>
__device__ __forceinline__ void mb_gemv_phase(const MbCtx& c, MbStream& s, const int (&voff4)[4], const int (&voff14)[4], int nsteps, const bf16_t* x, int ldx,
                                              const unsigned* wait_f, int wait_n, unsigned wait_code, float* ws_out, int N, int n0, unsigned* done_f, int t_ready,
                                              int r0wg /* first weight row of the item: the K rotation */, int nact = 4 /* GEMV waves that have an item */) {
  constexpr int SX = MbCfg<BT>::SX, RING = MbCfg<BT>::RING;
  constexpr int NX = BT * 4;                        // copies per step
  MB_IDS
  const int B = c.a->B;
  const int abl = c.a->abl;   // timing ablations (tools only; wrong results): 1 x copies without sc1, 2 no MFMAs, 4 no x copies, 8 no attention, 16 no waits, 32 no finish arithmetic
  const int fr = lane & 15, fq = lane >> 4;
  char* xring = c.xreg;
  f32x4 acc[RT][BT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) acc[rt][bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // x loaders: waves 4 .. 4 + NLW - 1; loader lw copies the row groups g = lw, lw + NLW, ... (rows 4 g .. 4 g + 3; batch tile g / 4) of every
  // step.  Row groups beyond the batch are not copied at all (their MFMA columns are never stored).
  constexpr int NLW = MB_NLW;
  const int lw = wave - 4;
  int nlw = 0;                                       // copies per step of this loader wave
#pragma unroll
  for (int g = 0; g < NX; ++g) nlw += (lw >= 0 && lw < NLW && g % NLW == lw && g * 4 < B) ? 1 : 0;
  const int xshift = pcy_gemv_kshift(r0wg, nsteps);   // the workgroup's rotated K order: the same rule as the weight streams'
  auto issue_x = [&](int ss) __attribute__((always_inline)) {
    char* xb = xring + (ss % SX) * (BT * 4096);
    int rs = ss + xshift;
    rs = rs >= nsteps ? rs - nsteps : rs;
#pragma unroll
    for (int g = 0; g < NX; ++g) {
      if (g % NLW != lw || g * 4 >= B) continue;
      const int row = (g & 3) * 4 + (lane >> 4);
      int b = (g >> 2) * 16 + row;
      b = b < B ? b : B - 1;
      const bf16_t* src = x + (size_t)b * ldx + rs * 128 + ((lane & 15) ^ row) * 8;
      if (abl & 4) continue;
      if (abl & 1) __builtin_amdgcn_global_load_lds((mb_gptr_t)src, (mb_lds_ptr_t)(xb + g * 1024), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds((mb_gptr_t)src, (mb_lds_ptr_t)(xb + g * 1024), 16, 0, 16 /* sc1 */);
    }
  };
  if (wave == 4) {
    if (wait_f && !(abl & 16)) mb_wait_flags(wait_f, wait_n, c.epoch, c.a->err, wait_code, lane);
    if (c.tr && lane == 0) c.tr[t_ready] = wall_clock64();
  }
  if (NLW > 1) __builtin_amdgcn_s_barrier();   // the inputs are there: every loader wave may start
  if (lw >= 0 && lw < NLW) {
    for (int i = 0; i < SX - 1 && i < nsteps; ++i) issue_x(i);
  }
  for (int ss = 0; ss < nsteps; ++ss) {
    if (wave < nact) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - RT) * 4) : "memory");     // this step's tile(s) have landed
    } else if (lw >= 0 && lw < NLW) {
      if (ss + SX - 1 <= nsteps && !(abl & 4)) {                                  // this wave's copies of x(ss) have landed
        switch (nlw) {
          case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SX - 2) * 1) : "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SX - 2) * 2) : "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SX - 2) * 3) : "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SX - 2) * 4) : "memory"); break;
          case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SX - 2) * 5) : "memory"); break;
          case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SX - 2) * 6) : "memory"); break;
          case 7: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SX - 2) * 7) : "memory"); break;
          case 8: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SX - 2) * 8) : "memory"); break;
          default: break;
        }
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // x(ss) is there for everybody; slot (ss - 1) % SX is free
    if (c.tr && t_ready == 4 && ss == 0 && tid == 0) c.tr[12] = wall_clock64();
    if (wave < nact) {
      const char* wb0 = s.ring + s.cslot * 4096;
      const int cs1 = s.cslot + 1 == RING ? 0 : s.cslot + 1;
      const char* wb1 = s.ring + cs1 * 4096;
      const char* xb = xring + (ss % SX) * (BT * 4096);
      bf16x8 wf[RT][4];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[rt][j] = *reinterpret_cast<const bf16x8*>((rt ? wb1 : wb0) + fr * 256 + (((j * 4 + fq) ^ fr) << 4));
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        bf16x8 xf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xb + bt * 4096 + fr * 256 + (((j * 4 + fq) ^ fr) << 4));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (!(abl & 2)) acc[rt][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[rt][j], xf[j], acc[rt][bt], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // the tiles just read are refilled at once (the ring is wave-private: nobody else to wait for)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) mb_stream_issue<RING>(s, voff4, voff14);
      s.cslot = RT == 2 ? (cs1 + 1 == RING ? 0 : cs1 + 1) : cs1;
    } else if (lw >= 0 && lw < NLW) {
      if (ss + SX - 1 < nsteps) issue_x(ss + SX - 1);
    }
  }
  // ---- epilogue: GEMV wave w -> LDS -> wave w + 4 stores (written through), drains and the item's flag goes up ----
  if (c.tr && tid == 0 && (t_ready == 13 || t_ready == 4)) c.tr[t_ready == 13 ? 14 : 15] = wall_clock64();
  lds_barrier();                                    // everybody is done with the x ring
  uint4* stage = reinterpret_cast<uint4*>(xring);   // [4][BT][64]
  if (wave < nact) {
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      uint4 v;
      if (EPI == 0) {
        v = make_uint4(__float_as_uint(acc[0][bt][0]), __float_as_uint(acc[0][bt][1]), __float_as_uint(acc[0][bt][2]), __float_as_uint(acc[0][bt][3]));
      } else {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = rbf(silu_f(rbf(acc[0][bt][r]))) * rbf(acc[RT - 1][bt][r]);
        v = make_uint4(pack_bf(o[0], o[1]), pack_bf(o[2], o[3]), 0u, 0u);
      }
      stage[(wave * BT + bt) * 64 + lane] = v;
    }
  }
  lds_barrier();
  if (wave >= 4) {
    const int gw = wave - 4;
    if (gw >= nact) {
    } else if (EPI == 0) {
      const __amdgpu_buffer_rsrc_t rs = mb_rsrc(ws_out);
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        const int b = bt * 16 + fr;
        const uint4 v = stage[(gw * BT + bt) * 64 + lane];
        if (b < B) mb_st16(rs, (b * N + n0 + gw * 16 + fq * 4) * 4, v);
      }
    } else {
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        const int b = bt * 16 + fr;
        const uint4 v = stage[(gw * BT + bt) * 64 + lane];
        if (b < B)
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(c.a->act + (size_t)b * MBF + (c.wg * 4 + gw) * 16 + fq * 4),
                             (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its written-through stores have left
  }
  __builtin_amdgcn_s_barrier();
  if (tid == 256) mb_raise(done_f, c.epoch);
}

// F1 / F2: row b of the residual stream: x = bf16(bf16(sum of the four K-quarter slabs, in order) + x), xn = RMSNorm(x) * w -- element assignment,
// accumulation order and block reduction of gemv_splitk_finish_norm_kernel (256 threads; threads 256.. only keep the barriers company).
__device__ __forceinline__ void mb_finish_row(const MbCtx& c, const float* ws, const unsigned* wait_f, unsigned wait_code, int b, const bf16_t* w, unsigned* done_f) {
  const PcyMbArgs& a = *c.a;
  MB_IDS
  const int B = a.B;
  float* red = reinterpret_cast<float*>(c.xreg);
  // (the row's residual does not depend on the items this finisher waits for: requested in front of the wait)
  uint4 rr[2];
  if (tid < 256 && !(a.abl & 32)) {
    const __amdgpu_buffer_rsrc_t rx0 = mb_rsrc(a.x);
#pragma unroll
    for (int it = 0; it < 2; ++it) rr[it] = mb_ld16(rx0, (b * MBD + tid * 8 + it * 2048) * 2);
  }
  if (wave == 4 && !(a.abl & 16)) mb_wait_flags(wait_f, 256, c.epoch, a.err, wait_code, lane);
  __builtin_amdgcn_s_barrier();
  float xv[2][8];
  float ss = 0.f;
  if (tid < 256 && !(a.abl & 32)) {
    const __amdgpu_buffer_rsrc_t rws = mb_rsrc(ws), rx = mb_rsrc(a.x);
    uint4 p[2][2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int k = tid * 8 + it * 2048;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) p[it][h][s_] = mb_ld16(rws, ((s_ * B + b) * MBD + k + h * 4) * 4);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int k = tid * 8 + it * 2048;
      const uint32_t rw[4] = {rr[it].x, rr[it].y, rr[it].z, rr[it].w};
      uint32_t packed[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[4] = {__uint_as_float(p[it][h][0].x), __uint_as_float(p[it][h][0].y), __uint_as_float(p[it][h][0].z), __uint_as_float(p[it][h][0].w)};
#pragma unroll
        for (int s_ = 1; s_ < 4; ++s_) {
          v[0] += __uint_as_float(p[it][h][s_].x); v[1] += __uint_as_float(p[it][h][s_].y);
          v[2] += __uint_as_float(p[it][h][s_].z); v[3] += __uint_as_float(p[it][h][s_].w);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float o = rbf(v[r]);
          const uint32_t rword = rw[h * 2 + (r >> 1)];
          o = rbf(o + ((r & 1) ? hi_bf(rword) : lo_bf(rword)));
          xv[it][h * 4 + r] = o;
        }
        packed[h * 2] = pack_bf(xv[it][h * 4], xv[it][h * 4 + 1]);
        packed[h * 2 + 1] = pack_bf(xv[it][h * 4 + 2], xv[it][h * 4 + 3]);
      }
      mb_st16(rx, (b * MBD + k) * 2, make_uint4(packed[0], packed[1], packed[2], packed[3]));
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float p_ = xv[it][2 * j], q_ = xv[it][2 * j + 1]; ss += p_ * p_ + q_ * q_; }
    }
    ss = wave_sum(ss);
  }
  lds_barrier();
  if (tid < 256 && lane == 0) red[wave] = ss;
  lds_barrier();
  if (tid < 256 && !(a.abl & 32)) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) t += red[i];
    const float rstd = rsqrtf(t / (float)MBD + a.rms_eps);
    const __amdgpu_buffer_rsrc_t rxn = mb_rsrc(a.xn);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int k = tid * 8 + it * 2048;
      const uint4 g = ldg16(w + k);
      const uint32_t gg[4] = {g.x, g.y, g.z, g.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float p_ = xv[it][2 * j] * rstd, q_ = xv[it][2 * j + 1] * rstd;
        if (a.rms_cast == 0) { p_ = rbf(p_); q_ = rbf(q_); }
        o[j] = pack_bf(lo_bf(gg[j]) * p_, hi_bf(gg[j]) * q_);
      }
      mb_st16(rxn, (b * MBD + k) * 2, make_uint4(o[0], o[1], o[2], o[3]));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (tid == 0) mb_raise(done_f, c.epoch);
}

// A: the decode attention of unit (row b, kv head, column slice bx); the new token's q / k / v = the two K-half slabs of the Q phase added in
// split order (attn_dec_splitk_kernel's hook), fetched when the 24 Q items that hold them have arrived.
template <int DS>
__device__ __forceinline__ void mb_attention(const MbCtx& c, int layer, int unit) {
  constexpr int DH = 128, G = 4, SL = DH / DS;
  const PcyMbArgs& a = *c.a;
  const int kvh = unit & 7, bx = (unit >> 3) % SL, b = unit / (8 * SL);
  const size_t stage_off = (attn_dec_smem_bytes(G, DS, DH, a.Tmax) + 15) & ~(size_t)15;
  bf16_t* stage = reinterpret_cast<bf16_t*>(c.xreg + stage_off);   // [G + 2][DH]
  PcyDecAttnArgs t{};
  t.qkv = nullptr; t.ld = MBNQ;
  t.kcache = a.kcache + (size_t)layer * a.kv_layer_stride; t.vcache = a.vcache + (size_t)layer * a.kv_layer_stride;
  t.o = a.ao; t.ldo = MBD; t.pos_dev = a.pos_dev; t.cos_t = a.cos_t; t.sin_t = a.sin_t; t.keep = a.keep; t.ld_keep = a.ld_keep;
  t.B = a.B; t.H = 32; t.Hkv = 8; t.dh = DH; t.Tmax = a.Tmax; t.scale = a.scale;
  t.o_sc1 = 1; t.staged = stage;
  // the Q items that hold this kv head's rows: row tiles [32 kvh, +32) (q), [256 + 8 kvh, +8) (k), [320 + 8 kvh, +8) (v); tile t belongs to the
  // workgroups 2 (t / 3) and 2 (t / 3) + 1 (the two K halves).  Three contiguous flag ranges, widened to 16-byte granules.
  const unsigned* qf = c.lflags + MBF_QKV;
  const int tq = 32 * kvh, tk = 256 + 8 * kvh, tv = 320 + 8 * kvh;
  const int q_lo = (2 * (tq / 3)) & ~3, q_n = 2 * ((tq + 31) / 3) + 2 - q_lo;
  const int k_lo = (2 * (tk / 3)) & ~3, k_n = 2 * ((tk + 7) / 3) + 2 - k_lo;
  const int v_lo = (2 * (tv / 3)) & ~3, v_n = 2 * ((tv + 7) / 3) + 2 - v_lo;
  const float* ws = a.qkv_ws;
  MB_IDS
  const int B = a.B;
  const unsigned epoch = c.epoch;
  unsigned* err = a.err;
  unsigned long long* tr = c.tr;
  const int abl = a.abl;
  auto hook = [=]() __attribute__((always_inline)) {
    if (wave == 4 && !(abl & 16)) {
      mb_wait_flags(qf + q_lo, q_n, epoch, err, 31u, lane);
      mb_wait_flags(qf + k_lo, k_n, epoch, err, 37u, lane);
      mb_wait_flags(qf + v_lo, v_n, epoch, err, 38u, lane);
    }
    __builtin_amdgcn_s_barrier();
    if (tr && tid == 0) tr[2] = wall_clock64();
    constexpr int NV4 = (G + 2) * DH / 4;
    if (tid < NV4) {
      const int seg = tid / (DH / 4), e4 = tid % (DH / 4);
      const int n = (seg < G ? (kvh * G + seg) : (seg == G ? 32 + kvh : 32 + 8 + kvh)) * DH + e4 * 4;
      const __amdgpu_buffer_rsrc_t rs = mb_rsrc(ws);
      const uint4 p0 = mb_ld16(rs, (b * MBNQ + n) * 4), p1 = mb_ld16(rs, ((B + b) * MBNQ + n) * 4);
      const float v0 = __uint_as_float(p0.x) + __uint_as_float(p1.x), v1 = __uint_as_float(p0.y) + __uint_as_float(p1.y);
      const float v2 = __uint_as_float(p0.z) + __uint_as_float(p1.z), v3 = __uint_as_float(p0.w) + __uint_as_float(p1.w);
      *reinterpret_cast<uint2*>(stage + seg * DH + e4 * 4) = make_uint2(pack_bf(rbf(v0), rbf(v1)), pack_bf(rbf(v2), rbf(v3)));
    }
    lds_barrier();
  };
  if (!(abl & 8)) attn_dec_body<DH, G, DS>(t, c.xreg, bx, kvh, b, hook);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its written-through stores of the output have left
  __builtin_amdgcn_s_barrier();
  if (tid == 0) mb_raise(c.lflags + MBF_AO + (kvh >> 1) * 64 + (b * SL + bx) * 2 + (kvh & 1), epoch);
}

template <int BT, int DS>
__global__ __launch_bounds__(512) void decode_step_mb_kernel(PcyMbArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  MbCtx c;
  c.a = &a; c.smem = smem; c.xreg = smem + MbCfg<BT>::WBYTES;
  c.wg = (int)blockIdx.x;
  c.epoch = *a.epoch;
  MB_IDS
  const int wg = c.wg, B = a.B;
  constexpr int SL = 128 / DS;
  const int units = 8 * SL * B;
  int voff4[4], voff14[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = q * 4 + (lane >> 4);
    voff4[q] = row * MBD + ((lane & 15) ^ row) * 8;
    voff14[q] = row * MBF + ((lane & 15) ^ row) * 8;
  }
  MbStream s;
  s.layers = a.layers; s.n_layers = a.n_layers; s.wg = wg; s.wave = wave < 4 ? wave : 0;
  s.ring = smem + (wave < 4 ? wave : 0) * (MbCfg<BT>::RING * 4096);
  s.il = 0; s.iph = mb_has_item(wg, s.wave, 0) ? 0 : 1; s.islot = 0; s.cslot = 0; s.idone = 0; s.wdef = (a.abl & 64) ? 1 : 0;
  mb_stream_setup(s);
  if (wave < 4) {
#pragma unroll 1
    for (int i = 0; i < MbCfg<BT>::RING; ++i) mb_stream_issue<MbCfg<BT>::RING>(s, voff4, voff14);
  }
  for (int l = 0; l < a.n_layers; ++l) {
    const PcyLayerWeightsDev lw = a.layers[l];
    c.lflags = a.flags + (size_t)l * MBF_STRIDE;
    unsigned* nflags = a.flags + (size_t)(l + 1) * MBF_STRIDE;
    c.tr = a.trace ? a.trace + ((size_t)l * 256 + wg) * 16 : nullptr;
    MB_T(0)
    // ---- Q ----
    {
      const int kh = wg & 1;
      mb_gemv_phase<1, BT, 0>(c, s, voff4, voff14, 16, a.xn + kh * 2048, MBD, l ? c.lflags + MBF_XN : nullptr, B, 30u,
                              a.qkv_ws + (size_t)kh * B * MBNQ, MBNQ, (wg >> 1) * 48, c.lflags + MBF_QKV + wg, 13, (wg >> 1) * 48, 3);
    }
    MB_T(1)
    // ---- A ----
    if (wg < units) {
      mb_attention<DS>(c, l, wg);
      lds_barrier();
    }
    MB_T(3)
    // ---- O ----
    {
      const int rb = wg >> 2, kq = wg & 3;
      mb_gemv_phase<1, BT, 0>(c, s, voff4, voff14, 8, a.ao + kq * 1024, MBD, c.lflags + MBF_AO + kq * 64, 2 * SL * B, 32u,
                              a.sk_ws + (size_t)kq * B * MBD, MBD, rb * 64, c.lflags + MBF_O + wg, 4, rb * 64);
    }
    MB_T(5)
    // ---- F1 ----
    if (wg >= MB_F1_WG0 && wg < MB_F1_WG0 + B) {
      mb_finish_row(c, a.sk_ws, c.lflags + MBF_O, 33u, wg - MB_F1_WG0, lw.ln2, c.lflags + MBF_XN2 + (wg - MB_F1_WG0));
      lds_barrier();
    }
    MB_T(6)
    // ---- G ----
    if (wg < MB_G_WGS) {
      mb_gemv_phase<2, BT, 1>(c, s, voff4, voff14, 32, a.xn, MBD, c.lflags + MBF_XN2, B, 34u, nullptr, 0, 0,
                              c.lflags + MBF_ACT + (wg / 56) * 64 + wg % 56, 7, wg * 128);
    }
    MB_T(8)
    // ---- D ----
    {
      const int rb = wg >> 2, kq = wg & 3;
      mb_gemv_phase<1, BT, 0>(c, s, voff4, voff14, 28, a.act + kq * 3584, MBF, c.lflags + MBF_ACT + kq * 64, 56, 35u,
                              a.sk_ws + (size_t)kq * B * MBD, MBD, rb * 64, c.lflags + MBF_DN + wg, 9, rb * 64);
    }
    MB_T(10)
    // ---- F2 ----
    if (wg >= MB_F2_WG0 && wg < MB_F2_WG0 + B) {
      const bf16_t* nw = l + 1 < a.n_layers ? a.layers[l + 1].ln1 : a.final_norm;
      mb_finish_row(c, a.sk_ws, c.lflags + MBF_DN, 36u, wg - MB_F2_WG0, nw, nflags + MBF_XN + (wg - MB_F2_WG0));
      lds_barrier();
    }
    MB_T(11)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this workgroup's LDS may still be written when it retires
}

struct MbLaunchCache { int configured = 0; int resident = -1; };
MbLaunchCache g_mb_cache[16][3];

template <int BT, int DS>
bool launch_mb(hipStream_t s, int device, const PcyMbArgs& a, int n_cu, int slot) {
  if (device < 0 || device >= 16) return false;
  MbLaunchCache& lc = g_mb_cache[device][slot];
  if (!lc.configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_step_mb_kernel<BT, DS>), hipFuncAttributeMaxDynamicSharedMemorySize, MB_SMEM) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    lc.configured = 1;
  }
  // every workgroup waits for flags the others raise: all 256 must be resident at once
  if (lc.resident < 0) lc.resident = pcy_all_resident(decode_step_mb_kernel<BT, DS>, 512, MB_SMEM, 256, n_cu) ? 1 : 0;
  if (!lc.resident) return false;
  hipLaunchKernelGGL((decode_step_mb_kernel<BT, DS>), dim3(256), dim3(512), MB_SMEM, s, a);
  return true;
}

}  // namespace

size_t pcy_decode_mb_flag_words() { return MBF_STRIDE; }
int pcy_decode_mb_ds(int B) { return B >= 16 ? 128 : 64; }
bool pcy_decode_mb_fits(int B, int Tmax) {
  const int ds = pcy_decode_mb_ds(B);
  const size_t need = ((attn_dec_smem_bytes(4, ds, 128, Tmax) + 15) & ~(size_t)15) + (size_t)(4 + 2) * 128 * 2;
  return B >= 9 && B <= 32 && need <= (size_t)(B <= 16 ? MbCfg<1>::XBYTES : MbCfg<2>::XBYTES);
}

bool pcy_launch_decode_step_mb(hipStream_t s, int device, const PcyMbArgs& a, int n_cu) {
  if (a.B < 9 || a.B > 32 || n_cu < 256 || a.n_layers < 1 || !pcy_decode_mb_fits(a.B, a.Tmax)) return false;
  if (a.B <= 15) return launch_mb<1, 64>(s, device, a, n_cu, 0);
  if (a.B == 16) return launch_mb<1, 128>(s, device, a, n_cu, 1);
  return launch_mb<2, 128>(s, device, a, n_cu, 2);
}
