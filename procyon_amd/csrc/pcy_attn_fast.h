// Single-pass ("flash") attention for the ESM2 encoder: head_dim 64, bidirectional, packed varlen, no key mask
// (reference call: procyon/model/esm.py:517-538 -> ESM2 self-attention; row A2 of SURVEY.md section 8).
//
// Why a second kernel.  attn_kernel (pcy_attn.hip) reproduces the reference's eager bf16 pipeline op for op -- S rounded to
// bf16, P normalised BEFORE its bf16 rounding -- which forces two passes over the keys (two Q.K^T, two exponentials per score)
// and measured 18 % matrix-pipe busy at 25 x 1026 tokens (508 us per layer, a third of the encoder).  This kernel is the
// usual online-softmax form: ONE pass, S kept in fp32, P = bf16(exp(S - m)) unnormalised, O accumulated in fp32 and divided by
// the fp32 row sum at the end.  It is NOT bit-compatible with the reference's rounding points; it is closer to the fp32
// evaluation than the reference's own bf16 path (tests/test_gpu_fulldepth.py holds both to the same fp32 truth), and the exact
// kernel stays selectable (PCY_ESM_ATTN=exact).
//
// gfx950 design
//   * workgroup = 4 waves, each wave 64 queries (two 32-query groups), all four share the K / Vt tiles of one (sequence, head):
//     64 keys per tile, K [64 keys][64 dh] and Vt [64 dh][64 keys] = 8 KiB each, three buffers (48 KiB of LDS), brought in by
//     LDS-DMA (global_load_lds, 16 B per lane, no staging registers) one tile ahead, one barrier per tile.  128-byte rows: the 16-byte chunk c of row
//     r sits at position c ^ ((r >> 1) & 7) (applied on the SOURCE address and on the read), which makes every ds_read_b128
//     lane group hit 16 distinct 16-byte slots.
//   * 32x32x16 MFMAs.  S^T = K.Q^T with the key rows of a 32-key block PERMUTED over the MFMA's A rows
//     (row 8i + 4b + r  <->  key 16 (i >> 1) + 8 b + 4 (i & 1) + r), so that the 16 scores a lane ends up with are, per 16-key
//     chunk, 8 CONSECUTIVE keys of one query: after exp + bf16 packing they are that lane's B-operand fragment of
//     O^T = Vt.P^T as they stand (no LDS round trip, no shuffles), and the matching A operand is one 16-byte read of a Vt row.
//     A lane owns one query column of S^T and O^T: the row maximum needs the other half-wave only when the maximum grows.
//   * deferred maximum: O and l are rescaled only when a block's maximum exceeds the running one by more than 2^8 (see
//     fa_softmax_block); the row maximum then needs the other half-wave once (v_permlane32_swap).
//   * 1-D grid, XCD-aware: the q chunks of one (sequence, head) get consecutive logical ids on ONE XCD, so K / Vt (262 KB per
//     head at 1026 tokens) are fetched into one L2 instead of five.
#pragma once
#include "pcy_internal.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// Online softmax of one 32-key block of one 32-query group: s = this lane's 16 scores (query = lane & 31, keys per the permuted
// row map), m / l = running maximum (exp2 domain) / this lane's partial row sum, o = the group's O^T accumulators.
// The kernel is VALU-issue bound (measured by ablation: with every MFMA removed it loses 20 % of its time, with the exponentials
// 10 %, and what remains -- the fma / add / max / pack per score -- is 40 %), so this function is counted in instructions per score:
//   1 v_fma (s.log2e - m) + 1 v_exp + 1/2 v_pk_add (row sum, two running sums) + 1/2 v_max3 (inline asm: fmaxf on MFMA results gets a
//   canonicalising v_max per operand from hipcc) + 1/2 v_cvt_pk_bf16.
// Deferred maximum: m is raised -- and O, l rescaled -- only when some row's block maximum exceeds it by more than 2^8 (P stays
// <= 256: harmless for bf16's 8-bit exponent and the fp32 sums; m cancels in O / l).  After the first blocks no wave takes that
// branch on ordinary data; tests/test_gpu_round3.py forces it with spiked keys.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// (inline asm: hipcc's hazard recognizer does not see that it READS MFMA results -- an experiment that ran ONE query group per wave put
// an all-asm softmax right behind its own Q.K^T and read half-written accumulators, bits that changed from launch to launch.  The
// chain in fa_softmax_block therefore STARTS with a compiler-visible reader of the accumulators, see there.)
__device__ __forceinline__ float fa_max3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
template <bool MASKED, int ABL = 0>
__device__ __forceinline__ void fa_softmax_block(f32x16& s, float& m, float& l, f32x16 (&o)[2], bf16x8 (&pf)[2], int key0, int half, int len) {
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float THR = 8.0f;
  float v[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    v[e] = s[e];
    if (MASKED) {
      const int i = e >> 2, r = e & 3;
      const int key = key0 + 16 * (i >> 1) + 8 * half + 4 * (i & 1) + r;
      v[e] = key < len ? v[e] : -INFINITY;
    }
  }
  // The FIRST reader of the Q.K^T accumulators is a compiler-visible instruction (v_add_f32 with +0.0: not foldable without nsz, and
  // harmless -- it can only turn a -0.0 maximum into +0.0, which exp2(s - m) does not see): hipcc's hazard recognizer pads the
  // MFMA -> VALU wait states in front of IT, and every asm v_max3 below depends on its result, so none of them can be scheduled
  // ahead of the padding whatever the surrounding schedule looks like.  (fmed3(a, b, +inf) is folded into three canonicalising v_max.)
  float bm = fa_max3(v[0] + 0.0f, v[1], v[2]);
#pragma unroll
  for (int e = 3; e < 15; e += 2) bm = fa_max3(bm, v[e], v[e + 1]);
  bm = fa_max3(bm, v[15], v[15]);
  if (!(ABL & 1)) {
    if (__builtin_expect(__any(bm * LOG2E > m + THR), 0)) {
      // the other half-wave holds the other 16 keys of the same query; every row of the group moves to its true running maximum
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(bm), __float_as_uint(bm), false, false);
      const float row = fa_max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), m * (1.0f / LOG2E)) * LOG2E;
      const float mn = row > m ? row : m;               // (m = -inf in the first block)
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      l *= alpha;
#pragma unroll
      for (int t = 0; t < 2; ++t) o[t] *= alpha;
      m = mn;
    }
  }
  float p[16];
  f32x2_t sum2 = {0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 16; e += 2) {
    const float x0 = fmaf(v[e], LOG2E, -m), x1 = fmaf(v[e + 1], LOG2E, -m);
    p[e] = (ABL & 2) ? x0 : __builtin_amdgcn_exp2f(x0);   // (ABL: measurement variants, see attn_fast64_kernel)
    p[e + 1] = (ABL & 2) ? x1 : __builtin_amdgcn_exp2f(x1);
    sum2 += (f32x2_t){p[e], p[e + 1]};
  }
  l += sum2[0] + sum2[1];
  // chunk c (16 keys): registers i in {2c, 2c+1}  ->  slots j = 4 (i & 1) + r  <->  keys 16 c + 8 half + j
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const uint32_t w[4] = {pack_bf(p[8 * c + 0], p[8 * c + 1]), pack_bf(p[8 * c + 2], p[8 * c + 3]),
                           pack_bf(p[8 * c + 4], p[8 * c + 5]), pack_bf(p[8 * c + 6], p[8 * c + 7])};
    pf[c] = __builtin_bit_cast(bf16x8, w);
  }
}

// ABL != 0: measurement variants with parts of the work removed (wrong results; PCY_FA_ABL, tools/archive/bench_attn_esm.py):
//   1 no O rescale, 2 no exponentials, 4 no P.V MFMAs, 8 no Q.K^T MFMAs, 16 no Vt fragment reads, 32 no K fragment reads, 64 no DMA
// VROW: V is read where the qkv projection left it (token-major rows, a.v / a.ldv / a.vcol0) -- no transposed copy.  A V tile lands in
// LDS as [16 key quads][4 dh blocks][4 keys][16 dh] (128-byte blocks; the LDS-DMA image is lane-linear, so the order is made on the
// SOURCE addresses) and the A operand of O^T = Vt.P^T comes from two ds_read_b64_tr_b16: a 16-lane group supplies the addresses of one
// [4 keys][16 dh] block (lane L: key L >> 2, dh 4 (L & 3) .. = byte 8 L of the block) and lane i receives column i = 4 consecutive keys of
// ONE dh (lane map pinned by tools/probes/tr_read_map.hip).  The 32 lanes that share an LDS cycle read 256 contiguous bytes.
// (first form: [64 keys][64 dh] rows with the 64-byte halves swapped on bit 1 of the key -- conflict-free by the bank formula, but the
// kernel ran 17 us per layer slower than over the transposed copy)
typedef __attribute__((ext_vector_type(4))) short fa_s16x4;
// NG = query groups of 32 per wave.  2 (default): a wave alternates two groups, 256 queries per workgroup.  1: one group per wave, 128
// queries per workgroup, the two 32-key sub-blocks of a tile alternate instead -- for inputs with so few (sequence, head, chunk)
// workgroups that most CUs would idle (ONE 1024-residue protein: 100 workgroups of NG = 2, 180 of NG = 1, each half as long).  The
// arithmetic of a query group -- its key blocks in order, its rescale decisions (the __any is over the group's own 64 lanes) -- is the
// same in both forms: bit-identical results, so which form ran does not show in an embedding.
template <int ABL, bool VROW = false, int NG = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_fast64_kernel(PcyAttnArgs a, int nchunk) {
  constexpr int DH = 64, KT = 64, QW = 32 * NG, QB = 4 * QW, NBUF = 3;
  constexpr int TILE = KT * DH * 2;                     // 8 KiB: K tile, then Vt tile
  __shared__ __attribute__((aligned(1024))) char smem[NBUF * 2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  // XCD-aware logical id: consecutive logical ids run on one XCD
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int qc = lid % nchunk, h = (lid / nchunk) % a.H, sq = lid / (nchunk * a.H);
  const int t0 = a.cu[sq], len = a.cu[sq + 1] - t0;
  const int bq0 = qc * QB;
  if (bq0 >= len) return;                               // uniform per workgroup
  const int qr0 = bq0 + wave * QW;
  const bool active = qr0 < len;
  const int vt0 = VROW ? 0 : a.vt_cu[sq];
  const bf16_t* kglob = a.k + (size_t)t0 * a.ldk + a.kcol0 + h * DH;
  const bf16_t* vglob = VROW ? a.v + (size_t)t0 * a.ldv + a.vcol0 + h * DH : a.vt + (size_t)h * DH * a.vt_total + vt0;

  // Q fragments (B operand of S^T): lane holds Q[query col][16 ks + 8 half .. +8]
  bf16x8 qf[2][4];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    int qrow = qr0 + g * 32 + col;
    qrow = qrow < len ? qrow : len - 1;
    const bf16_t* qp = a.q + (size_t)(t0 + qrow) * a.ldq + a.qcol0 + h * DH + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[g][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  // Staging addresses are per-thread constants plus a tile offset (recomputing row / chunk / clamp per tile cost ~40 of the ~450
  // instructions of a tile); only the ragged last tile clamps its key rows.
  const bf16_t* kp[2]; const bf16_t* vp[2]; int srow[2], schunk[2], vrow[2], vchunk[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = i * 256 + tid, row = slot >> 3, cp = slot & 7;
    srow[i] = row; schunk[i] = (cp ^ ((row >> 1) & 7)) * 8;
    // (VROW) 16-byte slot of the tile image -> the (key, dh chunk) it holds: slot = ((key >> 2) * 4 + dh16) * 8 + (key & 3) * 2 + (dhc & 1)
    vrow[i] = (slot >> 5) * 4 + ((slot >> 1) & 3);
    vchunk[i] = (((slot >> 3) & 3) * 2 + (slot & 1)) * 8;
    kp[i] = kglob + (size_t)row * a.ldk + schunk[i];
    vp[i] = VROW ? vglob + (size_t)vrow[i] * a.ldv + vchunk[i] : vglob + (size_t)row * a.vt_total + schunk[i];
  }
  const size_t ktile_stride = (size_t)KT * a.ldk, vtile_stride = VROW ? (size_t)KT * a.ldv : (size_t)KT;
  // VROW form: the pieces as buffer loads -- resources based at the sequence's first K / V row of this head (SGPRs), the lane's byte
  // offset inside a tile (32 bits, constant), the tile as the instruction's scalar offset; the wave index in an SGPR so that the LDS
  // destination (M0) is scalar arithmetic.  (The Vt form keeps global_load_lds.)
  const int wave_s = VROW ? __builtin_amdgcn_readfirstlane(wave) : wave;
  __amdgpu_buffer_rsrc_t krs, vrs;
  int kvo[2], vvo[2];
  if constexpr (VROW) {
    krs = __builtin_amdgcn_make_buffer_rsrc((void*)kglob, 0, 0x7fffffff, 0x00020000);
    vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vglob, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) { kvo[i] = (srow[i] * a.ldk + schunk[i]) * 2; vvo[i] = (vrow[i] * a.ldv + vchunk[i]) * 2; }
  }
  auto stage = [&](int kt, char* buf) __attribute__((always_inline)) {
    if (ABL & 64) return;
    const bool clamp = (kt + 1) * KT > len;
    if constexpr (VROW) {
      if (!clamp) {
        const int ks_off = kt * KT * a.ldk * 2, vs_off = kt * KT * a.ldv * 2;   // (a sequence's K / V rows span < 2 GiB)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(krs, (lds_ptr_t)(buf + (i * 4 + wave_s) * 1024), 16, kvo[i], ks_off, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(vrs, (lds_ptr_t)(buf + TILE + (i * 4 + wave_s) * 1024), 16, vvo[i], vs_off, 0, 0);
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bf16_t* ks = kp[i] + kt * ktile_stride;
      const bf16_t* vs = vp[i] + kt * vtile_stride;
      if (clamp) {
        int key = kt * KT + srow[i]; key = key < len ? key : len - 1;
        ks = kglob + (size_t)key * a.ldk + schunk[i];
        if (VROW) {   // (rows beyond the sequence: any finite values, their P is 0)
          int vk = kt * KT + vrow[i]; vk = vk < len ? vk : len - 1;
          vs = vglob + (size_t)vk * a.ldv + vchunk[i];
        }
      }
      __builtin_amdgcn_global_load_lds((gptr_t)ks, (lds_ptr_t)(buf + (i * 4 + wave_s) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)vs, (lds_ptr_t)(buf + TILE + (i * 4 + wave_s) * 1024), 16, 0, 0);
    }
  };
  // A-operand row (lane & 31) of S^T  ->  key of the 32-key block
  const int krow = 16 * (col >> 4) + 8 * ((col >> 2) & 1) + 4 * ((col >> 3) & 1) + (col & 3);
  auto kfrag = [&](const char* buf, int sub, int ks) __attribute__((always_inline)) {
    if (ABL & 32) return qf[0][ks];
    const int row = krow + 32 * sub;
    return *reinterpret_cast<const bf16x8*>(buf + row * 128 + (((2 * ks + half) ^ ((row >> 1) & 7)) << 4));
  };
  // (VROW) byte offset of this lane's 8-byte piece: block (key quad 2 half + .., dh block (lane >> 4) & 1 + ..), byte 8 (lane & 15)
  const int vbase = half * 1024 + ((lane >> 4) & 1) * 128 + (lane & 15) * 8;
  auto vfrag = [&](const char* buf, int t, int sub, int c) __attribute__((always_inline)) {
    if (ABL & 16) return qf[1][t + 2 * c];
    if constexpr (VROW) {
      typedef __attribute__((address_space(3))) fa_s16x4* tr_ptr_t;
      const char* p = buf + TILE + vbase + ((8 * sub + 4 * c) * 4 + 2 * t) * 128;   // key quad 8 sub + 4 c (+ 2 half) + u, dh block 2 t (+ ..)
      const fa_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(p));
      const fa_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(p + 4 * 128));
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      return __builtin_bit_cast(bf16x8, (s16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
    } else {
      const int row = 32 * t + col;
      return *reinterpret_cast<const bf16x8*>(buf + TILE + row * 128 + (((4 * sub + 2 * c + half) ^ ((row >> 1) & 7)) << 4));
    }
  };

  float m0 = -INFINITY, m1 = -INFINITY, l0 = 0.f, l1 = 0.f;
  f32x16 o0[2], o1[2];   // O^T of query group 0 / 1: [dh tile]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[t][e] = 0.f; o1[t][e] = 0.f; }

  // Software pipeline INSIDE a wave (the matrix pipe runs asynchronously to the wave's VALU stream): the two query groups
  // alternate -- while the exponentials of one group's block run on the VALU, the S^T MFMAs of the other group and the P.V MFMAs
  // of the previous step are in flight:
  //   A(n): QK(n, g1) -> s1 | softmax(s0 = (n, g0)) | PV(n, g0)        B(n): QK(n+1, g0) -> s0 | softmax(s1 = (n, g1)) | PV(n, g1)
  // The K fragments of a block are read once and serve both groups; statically named score registers (s0 / s1).
  f32x16 s0, s1;
  bf16x8 kf[4];
#define FA_LOAD_K(BUF, SUB)                                                    \
  _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) kf[ks] = kfrag(BUF, SUB, ks);
#define FA_QK(S, G)                                                            \
  do {                                                                         \
    _Pragma("unroll") for (int e = 0; e < 16; ++e) S[e] = 0.f;                 \
    if (ABL & 8) {                                                             \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) { S[4 * ks] = __builtin_bit_cast(f32x4, kf[ks])[0]; S[4 * ks + 1] = __builtin_bit_cast(f32x4, qf[G][ks])[1]; } \
    } else {                                                                   \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[G][ks], S, 0, 0, 0); \
    }                                                                          \
  } while (0)
#define FA_SM_PV(BUF, KT_, SUB, S, M, L, O, TAIL)                                                        \
  do {                                                                                                   \
    bf16x8 pf[2];                                                                                        \
    fa_softmax_block<TAIL, ABL>(S, M, L, O, pf, (KT_) * KT + 32 * (SUB), half, len);                     \
    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                        \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                    \
        const bf16x8 vf_ = vfrag(BUF, t, SUB, c);                                                        \
        if (ABL & 4) { asm volatile("" :: "v"(vf_), "v"(pf[c])); O[t][c] += 1.0f; }                      \
        else O[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf_, pf[c], O[t], 0, 0, 0);                  \
      }                                                                                                  \
  } while (0)

  const int ntiles = (len + KT - 1) / KT;
  const int nfull = len / KT;                           // tiles without keys beyond the sequence: the pipelined, mask-free loop
  // Three tile buffers, ONE barrier per tile: the barrier at the end of tile u (before the first S^T of tile u+1) finds tile u+1
  // landed (its DMA was issued a whole tile earlier; the barrier's vmcnt(0) covers it) and every wave done with tile u-1, whose
  // buffer the DMA of tile u+2 then overwrites.
  stage(0, smem);
  if (ntiles > 1) stage(1, smem + 2 * TILE);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (NG == 1) {
    // one group per wave: the tile's two 32-key sub-blocks alternate through (s0, s1) --
    //   QK(u.b) -> s1 | softmax(s0 = u.a), PV(u.a) | [barrier, DMA of tile u+2] QK((u+1).a) -> s0 | softmax(s1 = u.b), PV(u.b)
    if (active && nfull > 0) { FA_LOAD_K(smem, 0); FA_QK(s0, 0); }
    for (int u = 0; u < nfull; ++u) {
      const char* buf = smem + (u % NBUF) * 2 * TILE;
      if (active) {
        FA_LOAD_K(buf, 1);
        FA_QK(s1, 0);                                      // (u.b)
        FA_SM_PV(buf, u, 0, s0, m0, l0, o0, false);
      }
      if (u + 1 < ntiles) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (see the two-group loop below)
        __syncthreads();
        if (u + 2 < ntiles) stage(u + 2, smem + ((u + 2) % NBUF) * 2 * TILE);
        if (active && u + 1 < nfull) { FA_LOAD_K(smem + ((u + 1) % NBUF) * 2 * TILE, 0); FA_QK(s0, 0); }   // ((u+1).a)
      }
      if (active) FA_SM_PV(buf, u, 1, s1, m0, l0, o0, false);
    }
    if (active && nfull < ntiles) {                        // the ragged last tile: masked, not pipelined
      const char* buf = smem + (nfull % NBUF) * 2 * TILE;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if (nfull * KT + 32 * sub >= len) break;
        FA_LOAD_K(buf, sub);
        FA_QK(s0, 0);
        FA_SM_PV(buf, nfull, sub, s0, m0, l0, o0, true);
      }
    }
  } else {
  if (active && nfull > 0) { FA_LOAD_K(smem, 0); FA_QK(s0, 0); }
  for (int u = 0; u < nfull; ++u) {
    const char* buf = smem + (u % NBUF) * 2 * TILE;
    if (active) {
      FA_QK(s1, 1);                                      // (u.a, g1), K fragments of u.a still in kf
      FA_SM_PV(buf, u, 0, s0, m0, l0, o0, false);
      FA_LOAD_K(buf, 1);
      FA_QK(s0, 0);                                      // (u.b, g0)
      FA_SM_PV(buf, u, 0, s1, m1, l1, o1, false);
      FA_QK(s1, 1);                                      // (u.b, g1)
      FA_SM_PV(buf, u, 1, s0, m0, l0, o0, false);
    }
    if (u + 1 < ntiles) {
      // hipcc does NOT put a vmcnt wait in front of this barrier (the loop's only VMEM operations are LDS-DMA, which its waitcnt
      // pass does not count against the barrier: the ISA had `s_waitcnt lgkmcnt(0); s_barrier`): tile u+1 could then be read before
      // it has landed -- seen once as NaNs in ~100 runs.  The wait must be explicit.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (u + 2 < ntiles) stage(u + 2, smem + ((u + 2) % NBUF) * 2 * TILE);
      if (active && u + 1 < nfull) { FA_LOAD_K(smem + ((u + 1) % NBUF) * 2 * TILE, 0); FA_QK(s0, 0); }   // ((u+1).a, g0)
    }
    if (active) FA_SM_PV(buf, u, 1, s1, m1, l1, o1, false);
  }
  if (active && nfull < ntiles) {                        // the ragged last tile: masked, not pipelined (one tile of ~17)
    const char* buf = smem + (nfull % NBUF) * 2 * TILE;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (nfull * KT + 32 * sub >= len) break;           // a block entirely beyond the sequence
      FA_LOAD_K(buf, sub);
      FA_QK(s0, 0);
      FA_QK(s1, 1);
      FA_SM_PV(buf, nfull, sub, s0, m0, l0, o0, true);
      FA_SM_PV(buf, nfull, sub, s1, m1, l1, o1, true);
    }
  }
  }
#undef FA_LOAD_K
#undef FA_QK
#undef FA_SM_PV
  if (!active) return;
  // O^T: lane (query col, half) holds dh = 32 t + 8 i + 4 half + r  ->  four consecutive features per (t, i)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int qq = qr0 + g * 32 + col;
    const float lg = g ? l1 : l0;
    const float lt = lg + __shfl_xor(lg, 32, 64);
    const float inv = 1.0f / lt;
    if (qq >= len) continue;
    bf16_t* op = a.o + (size_t)(t0 + qq) * a.ldo + h * DH + 4 * half;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x16& ot = g ? o1[t] : o0[t];
        const uint32_t w0 = pack_bf(ot[4 * i + 0] * inv, ot[4 * i + 1] * inv);
        const uint32_t w1 = pack_bf(ot[4 * i + 2] * inv, ot[4 * i + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * t + 8 * i) = make_uint2(w0, w1);
      }
  }
}

// The caller has checked pcy_attn_fast_eligible (shape + alignment; pcy_attn.hip) and, for the token-major V form, pcy_attn_fast_vrow;
// the tests below restate it and can only fail for a caller that skipped the predicate.  Vt form: every sequence's slice padded to a
// multiple of 64 keys (so that a whole 64-key tile of Vt can be fetched; the pad columns are zero), vt_total a multiple of 8.
inline bool pcy_launch_attn_fast64(hipStream_t s, const PcyAttnArgs& a, bool vt_pad64) {
  if (a.dh != 64 || a.causal || a.keep || a.scale != 1.0f || a.H != a.Hkv || !vt_pad64) return false;
  if ((a.ldq | a.ldk | a.qcol0 | a.kcol0) % 8 || a.ldo % 4) return false;
  const int nchunk = (a.max_len + 255) / 256;
  const dim3 grid(nchunk * a.H * a.nseq);
  if (a.v != nullptr) {   // V token-major as the projection wrote it (no transposed copy)
    if ((a.ldv | a.vcol0) % 8) return false;
    // few workgroups (one or two proteins): one query group per wave, twice the workgroups of half the length (bit-identical; ESM2-650M,
    // one 1024-residue protein: 29.2 -> 21.2 us per layer, encoder 4.77 -> 4.53 ms)
    if ((long)nchunk * a.H * a.nseq <= 224) {
      const int nchunk1 = (a.max_len + 127) / 128;
      hipLaunchKernelGGL((attn_fast64_kernel<0, true, 1>), dim3(nchunk1 * a.H * a.nseq), dim3(256), 0, s, a, nchunk1);
      return true;
    }
    hipLaunchKernelGGL((attn_fast64_kernel<0, true>), grid, dim3(256), 0, s, a, nchunk);
    return true;
  }
  if (a.vt_total % 8) return false;
#ifdef PCY_FA_ABL_BUILD   // measurement variants with parts of the work removed (WRONG results; tools/archive/bench_attn_abl.py builds with it)
  const char* e = getenv("PCY_FA_ABL");
  const int abl = e ? atoi(e) : 0;
  switch (abl) {
#define FA_CASE(V) case V: hipLaunchKernelGGL(attn_fast64_kernel<V>, grid, dim3(256), 0, s, a, nchunk); return true;
    FA_CASE(1) FA_CASE(2) FA_CASE(3) FA_CASE(4) FA_CASE(8) FA_CASE(12) FA_CASE(48) FA_CASE(64) FA_CASE(112) FA_CASE(15) FA_CASE(127)
#undef FA_CASE
    default: break;
  }
#endif
  hipLaunchKernelGGL(attn_fast64_kernel<0>, grid, dim3(256), 0, s, a, nchunk);
  return true;
}

}  // namespace
