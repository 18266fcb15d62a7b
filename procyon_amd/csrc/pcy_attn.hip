// Attention kernels of the ProCyon path on gfx950.
//
// (1) attn_kernel<DH>: exact-rounding two-pass MFMA attention for ESM2 (bidirectional, packed varlen)
//     and Llama prefill (causal, GQA, key-keep mask).  "Exact-rounding" = the reference's eager bf16
//     pipeline is reproduced op for op:  S = bf16(Q.K^T) ; S = bf16(S*scale) ; (+mask) ;
//     P = bf16(softmax_fp32(S)) ; O = bf16(P.V)  -- P is normalised BEFORE the bf16 rounding, so the
//     row max / sum are computed in a first pass over the keys and P.V in a second pass.
//     Layout trick: S^T = K.Q^T is computed with the keys of each 32-key block permuted over the two
//     16-row MFMA tiles so that every lane ends with 8 CONSECUTIVE keys of one query; those 8
//     probabilities are exactly the lane's A-operand fragment of the P.V MFMA (no LDS, no shuffles),
//     and the matching B operand is one 16-byte load from a pre-transposed V (Vt[d][key]).
// (2) attn_dec_scores / attn_dec_pv: one new token per row against the KV cache, RoPE + cache append fused.
#include <stdlib.h>
#include "pcy_internal.h"
#include "pcy_handover.h"
#include "pcy_mlp_chain.h"

#ifndef PCY_STEP_SPLIT_WQKV
#define PCY_STEP_SPLIT_WQKV 1
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// QT = 16-row query tiles per wave: the K / Vt fragments of a 32-key block are loaded once and reused by all QT tiles
// (QT x fewer L2 requests per MFMA); the next block's fragments are prefetched while the current one is consumed.
// SCALED = false when the score scale is 1 (ESM: q arrives pre-scaled): the scale-and-round step disappears at compile time.
//
// Loop structure: the key blocks that need no masking for ANY query row of the wave (inside the sequence, below the causal
// diagonal, no key mask) run in a loop whose body is straight-line code -- no branch between the QT independent
// softmax chains, so the scheduler can interleave their MFMA / VALU / cross-lane steps; the remaining (boundary or
// masked) blocks run in the general loop.  Measured: a handful of uniform branches inside the block body cost 18 %.
// VALU budget per score and pass: half a v_cvt_pk_bf16_f32 + one unpack for the bf16 rounding, one raw v_max (pass 1),
// one v_fma (log2e fold: exp(s - m) = exp2(s.log2e - m.log2e)), one v_exp_f32, one add (pass 1) or one multiply by 1/l
// and half a cvt_pk (pass 2).
template <int DH, int QT, bool PF, bool SCALED>
__global__ __launch_bounds__(256) void attn_kernel(PcyAttnArgs a) {
  constexpr int KB = DH / 32;   // k-blocks of the QK^T contraction
  constexpr int NT = DH / 16;   // 16-wide output tiles of P.V
  constexpr int QROWS = QT * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int sq = blockIdx.z, h = blockIdx.y;
  const int t0 = a.cu[sq], len = a.cu[sq + 1] - t0;
  const int qr0 = (blockIdx.x * 4 + wave) * QROWS;
  if (qr0 >= len) return;
  const int G = a.H / a.Hkv;
  const int kvh = h / G;
  const int vt0 = a.vt_cu[sq];

  // Q fragments (B operand of S^T = K.Q^T): lane holds Q[q = fr][kb*32 + fq*8 .. +8] of each q tile
  bf16x8 qf[QT][KB];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qrow = (qr0 + qt * 16 + fr) < len ? (qr0 + qt * 16 + fr) : len - 1;
    const bf16_t* qp = a.q + (size_t)(t0 + qrow) * a.ldq + a.qcol0 + h * DH + fq * 8;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) qf[qt][kb] = *reinterpret_cast<const bf16x8*>(qp + kb * 32);
  }
  const bf16_t* kbase = a.k + (size_t)t0 * a.ldk + a.kcol0 + kvh * DH + fq * 8;
  const uint8_t* keep = a.keep ? a.keep + t0 : nullptr;
  // A-operand row fr of tile a / tile b maps to key (fr/4)*8 + (fr%4) (+4 for tile b) of the block
  const int krow_a = (fr >> 2) * 8 + (fr & 3);

  auto load_k = [&](int kb0, bf16x8 (&fa)[KB], bf16x8 (&fb)[KB]) {
    int ka = kb0 + krow_a, kbk = ka + 4;
    ka = ka < len ? ka : len - 1;
    kbk = kbk < len ? kbk : len - 1;
    const bf16_t* pa = kbase + (size_t)ka * a.ldk;
    const bf16_t* pb = kbase + (size_t)kbk * a.ldk;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      fa[kb] = *reinterpret_cast<const bf16x8*>(pa + kb * 32);
      fb[kb] = *reinterpret_cast<const bf16x8*>(pb + kb * 32);
    }
  };
  constexpr float LOG2E = 1.4426950408889634f;
  // masked keys: the reference's finfo.min additive mask.  A power of two, so MASKV * log2e is exact and a fully masked
  // row gets exp2(0) = 1 for every key (uniform softmax); keys beyond the sequence get -inf (weight exactly 0)
  constexpr float MASKV = -0x1p126f;
  auto round2 = [&](float x0, float x1, float& y0, float& y1) {   // two bf16 roundings: one cvt_pk + two unpacks
    const uint32_t w = pack_bf(x0, x1);
    y0 = lo_bf(w); y1 = hi_bf(w);
  };
  // bf16-rounded (and scaled-and-rounded) scores of one 32-key block for q tile qt: this lane gets keys kb0 + fq*8 + 0..7
  // of query qr0 + qt*16 + fr
  auto scores_raw = [&](int qt, const bf16x8 (&fa)[KB], const bf16x8 (&fb)[KB], float (&s)[8]) {
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kb], qf[qt][kb], sa, 0, 0, 0);
      sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kb], qf[qt][kb], sb, 0, 0, 0);
    }
    round2(sa[0], sa[1], s[0], s[1]); round2(sa[2], sa[3], s[2], s[3]);
    round2(sb[0], sb[1], s[4], s[5]); round2(sb[2], sb[3], s[6], s[7]);
    if (SCALED) {
#pragma unroll
      for (int r = 0; r < 8; r += 2) round2(s[r] * a.scale, s[r + 1] * a.scale, s[r], s[r + 1]);
    }
  };
  auto mask_block = [&](int kb0, int qt, float (&s)[8]) {
    const int qpos = qr0 + qt * 16 + fr;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int j = kb0 + fq * 8 + r;
      bool allowed = !(a.causal && j > qpos);
      if (keep && j < len) allowed = allowed && (keep[j] != 0);
      s[r] = j < len ? (allowed ? s[r] : MASKV) : -INFINITY;
    }
  };
  auto vmax = [](float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };   // never NaN here
  auto vmax3 = [](float x, float y, float z) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z)); return r; };

  int kend = a.causal ? ((qr0 + QROWS) < len ? (qr0 + QROWS) : len) : len;
  // leading key blocks that need no masking for any query row of this wave
  int kint = keep ? 0 : (len / 32) * 32;
  if (a.causal) { const int kc = qr0 >= 31 ? ((qr0 + 1) / 32) * 32 : 0; kint = kint < kc ? kint : kc; }
  float m[QT], l[QT];
  // Online max / sum PER LANE: a lane sees keys fq*8 .. fq*8+7 of every block of its query, so the running (m, l) of the
  // four lanes of a query are merged once after the loop instead of exchanging partial maxima and sums through the LDS
  // crossbar in every block (12 ds_bpermute per block and their waits inside the dependent chain).
  auto p1_update = [&](int qt, const float (&s)[8], bool guard) {
    float bm = vmax3(s[0], s[1], s[2]);
    bm = vmax3(bm, s[3], s[4]);
    bm = vmax3(bm, s[5], s[6]);
    bm = vmax(bm, s[7]);
    const float mn = vmax(m[qt], bm);
    // guard (boundary blocks only): a lane whose keys are all beyond the sequence keeps m = -inf; 0 * inf must not appear
    const float nmz = (guard && mn == -INFINITY) ? 0.f : -mn * LOG2E;
    float bs = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) bs += __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, nmz));
    l[qt] = l[qt] * __builtin_amdgcn_exp2f(fmaf(m[qt], LOG2E, nmz)) + bs;
    m[qt] = mn;
  };
  auto p1_merge = [&]() {   // (m, l) of the lanes {x, x^16, x^32, x^48} -> the row's max and sum on all four
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mm = vmax(m[qt], __shfl_xor(m[qt], 16, 64));
      mm = vmax(mm, __shfl_xor(mm, 32, 64));
      float ll = (m[qt] == -INFINITY) ? 0.f : l[qt] * __builtin_amdgcn_exp2f((m[qt] - mm) * LOG2E);
      ll += __shfl_xor(ll, 16, 64);
      ll += __shfl_xor(ll, 32, 64);
      m[qt] = mm; l[qt] = ll;
    }
  };
  for (int attempt = 0; attempt < 2; ++attempt) {
    // pass 1: online row max / sum of exp over keys [0, kend)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m[qt] = -INFINITY; l[qt] = 0.f; }
    bf16x8 ka[KB], kb_[KB];
    load_k(0, ka, kb_);
    int kb0 = 0;
    const int ki = kint < kend ? kint : kend;
    for (; kb0 < ki; kb0 += 32) {   // straight-line body
      bf16x8 na[KB], nb[KB];
      load_k(kb0 + 32, na, nb);     // row indices are clamped: the (unused) load past the end is harmless
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float s[8];
        scores_raw(qt, ka, kb_, s);
        p1_update(qt, s, false);
      }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) { ka[kb] = na[kb]; kb_[kb] = nb[kb]; }
    }
    for (; kb0 < kend; kb0 += 32) {
      bf16x8 na[KB], nb[KB];
      const bool more = kb0 + 32 < kend;
      if (more) load_k(kb0 + 32, na, nb);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float s[8];
        scores_raw(qt, ka, kb_, s);
        mask_block(kb0, qt, s);
        p1_update(qt, s, true);
      }
      if (more) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) { ka[kb] = na[kb]; kb_[kb] = nb[kb]; }
      }
    }
    p1_merge();
    // a row whose allowed-key set is empty (a left-pad query): the reference's additive finfo.min mask
    // makes its softmax uniform over ALL keys of the sequence, causal or not -> redo over the full range
    bool empty_row = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) empty_row = empty_row || ((m[qt] == MASKV) && (qr0 + qt * 16 + fr) < len);
    if (attempt == 0 && kend < len && __any(empty_row)) { kend = len; continue; }
    break;
  }

  // pass 2: P = bf16(exp(S - m) / l), O += P.V   (one reciprocal per row; a product instead of a division per score)
  float rl[QT], nmz[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) { rl[qt] = 1.0f / l[qt]; nmz[qt] = -m[qt] * LOG2E; }
  f32x4 oacc[QT][NT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int n = 0; n < NT; ++n) oacc[qt][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16_t* vbase = a.vt + ((size_t)kvh * DH + fr) * a.vt_total + vt0 + fq * 8;
  auto load_v = [&](int kb0, bf16x8 (&vf)[NT]) {
#pragma unroll
    for (int n = 0; n < NT; ++n) vf[n] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)n * 16 * a.vt_total + kb0);
  };
  auto p2_update = [&](int qt, const float (&s)[8], const bf16x8 (&vf)[NT]) {
    float p[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, nmz[qt])) * rl[qt];
    const uint32_t pw[4] = {pack_bf(p[0], p[1]), pack_bf(p[2], p[3]), pack_bf(p[4], p[5]), pack_bf(p[6], p[7])};
    const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
    for (int n = 0; n < NT; ++n) oacc[qt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf[n], oacc[qt][n], 0, 0, 0);
  };
  {
    bf16x8 ka[KB], kb_[KB], vf[NT];
    load_k(0, ka, kb_);
    load_v(0, vf);
    int kb0 = 0;
    // the V prefetch of the block after the last one would leave the sequence's (32-padded) column range
    const int last = ((kend + 31) / 32 - 1) * 32;
    const int ki = kint < last ? kint : last;
    for (; kb0 < ki; kb0 += 32) {   // straight-line body
      bf16x8 na[KB], nb[KB], nv[NT];
      load_k(kb0 + 32, na, nb);
      load_v(kb0 + 32, nv);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float s[8];
        scores_raw(qt, ka, kb_, s);
        p2_update(qt, s, vf);
      }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) { ka[kb] = na[kb]; kb_[kb] = nb[kb]; }
#pragma unroll
      for (int n = 0; n < NT; ++n) vf[n] = nv[n];
    }
    for (; kb0 < kend; kb0 += 32) {
      bf16x8 na[KB], nb[KB], nv[NT];
      const bool more = kb0 + 32 < kend;
      if (more) { load_k(kb0 + 32, na, nb); load_v(kb0 + 32, nv); }
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float s[8];
        scores_raw(qt, ka, kb_, s);
        if (!(kb0 < kint)) mask_block(kb0, qt, s);
        p2_update(qt, s, vf);
      }
      if (more) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) { ka[kb] = na[kb]; kb_[kb] = nb[kb]; }
#pragma unroll
        for (int n = 0; n < NT; ++n) vf[n] = nv[n];
      }
    }
  }
  // O[q = fq*4 + r][d = n*16 + fr]
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qr0 + qt * 16 + fq * 4 + r;
      if (qq >= len) continue;
      bf16_t* op = a.o + (size_t)(t0 + qq) * a.ldo + h * DH + fr;
#pragma unroll
      for (int n = 0; n < NT; ++n) op[n * 16] = f2bf(oacc[qt][n][r]);
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-shared variant (the shipped one for dh <= 128): the 4 waves of a workgroup walk the key blocks together; each
// 32-key block of K ([32][DH]) and Vt ([DH][32]) is fetched from L2 ONCE per workgroup (one or two 16-byte loads per
// thread), staged in a double-buffered, XOR-swizzled LDS image and read back as MFMA fragments with ds_read_b128.
// The per-wave global fragment loads of attn_kernel made the kernel L2-bandwidth bound (5.6 GB of fragment traffic per
// ESM layer at batch 32); this cuts it 4x.  Arithmetic and rounding are identical to attn_kernel.
template <int DH, int QT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_lds_kernel(PcyAttnArgs a) {
  constexpr int KB = DH / 32, NT = DH / 16, QROWS = QT * 16;
  constexpr int KCH = DH / 8;                  // 16-B chunks per key row
  constexpr int KLD = (32 * KCH) / 256;        // K chunks per thread (1 for DH=64, 2 for DH=128)
  constexpr int VLD = (DH * 4) / 256;          // Vt chunks per thread
  constexpr int KTILE = 32 * DH * 2, VTILE = DH * 32 * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * (KTILE + VTILE)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int sq = blockIdx.z, h = blockIdx.y;
  const int t0 = a.cu[sq], len = a.cu[sq + 1] - t0;
  const int bq0 = blockIdx.x * 4 * QROWS;
  if (bq0 >= len) return;                      // uniform per workgroup
  const int qr0 = bq0 + wave * QROWS;
  const bool active = qr0 < len;
  const int G = a.H / a.Hkv;
  const int kvh = h / G;
  const int vt0 = a.vt_cu[sq];
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float MASKZ = -3.0e38f;
  const uint8_t* keep = a.keep ? a.keep + t0 : nullptr;

  bf16x8 qf[QT][KB];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int qrow = qr0 + qt * 16 + fr;
    qrow = qrow < len ? qrow : len - 1;
    const bf16_t* qp = a.q + (size_t)(t0 + qrow) * a.ldq + a.qcol0 + h * DH + fq * 8;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) qf[qt][kb] = *reinterpret_cast<const bf16x8*>(qp + kb * 32);
  }
  const bf16_t* kglob = a.k + (size_t)t0 * a.ldk + a.kcol0 + kvh * DH;
  const bf16_t* vglob = a.vt + (size_t)kvh * DH * a.vt_total + vt0;
  auto kswz = [](int row) __attribute__((always_inline)) { return DH == 64 ? (row & 7) : (row & 15); };
  auto vswz = [](int d) __attribute__((always_inline)) { return (d >> 2) & 3; };
  // cooperative global -> register fetch of key block kb0 (K part / Vt part)
  // (register sets travel by value: arrays handed to the lambdas by reference ended up in scratch memory -- a load, a wait and
  // a scratch store per prefetch, 40 of the 48 us of the single-prompt launch)
  struct KRegs { uint4 v[KLD]; };
  struct VRegs { uint4 v[VLD]; };
  auto fetch_k = [&](int kb0) __attribute__((always_inline)) {
    KRegs rr;
    uint4 (&r)[KLD] = rr.v;
#pragma unroll
    for (int i = 0; i < KLD; ++i) {
      const int c = tid + i * 256, key = c / KCH, ch = c % KCH;
      int kj = kb0 + key;
      kj = kj < len ? kj : len - 1;
      r[i] = *reinterpret_cast<const uint4*>(kglob + (size_t)kj * a.ldk + ch * 8);
    }
    return rr;
  };
  auto put_k = [&](char* buf, const KRegs& rr) __attribute__((always_inline)) {
    const uint4 (&r)[KLD] = rr.v;
#pragma unroll
    for (int i = 0; i < KLD; ++i) {
      const int c = tid + i * 256, key = c / KCH, ch = c % KCH;
      *reinterpret_cast<uint4*>(buf + (key * KCH + (ch ^ kswz(key))) * 16) = r[i];
    }
  };
  const int vlast = len > 0 ? ((len - 1) / 32) * 32 : 0;   // first key of the sequence's last (32-padded) block
  auto fetch_v = [&](int kb0) __attribute__((always_inline)) {
    VRegs rr;
    uint4 (&r)[VLD] = rr.v;
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
      const int c = tid + i * 256, d = c >> 2, ch = c & 3;
      r[i] = *reinterpret_cast<const uint4*>(vglob + (size_t)d * a.vt_total + (kb0 < vlast ? kb0 : vlast) + ch * 8);
    }
    return rr;
  };
  auto put_v = [&](char* buf, const VRegs& rr) __attribute__((always_inline)) {
    const uint4 (&r)[VLD] = rr.v;
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
      const int c = tid + i * 256, d = c >> 2, ch = c & 3;
      *reinterpret_cast<uint4*>(buf + (d * 4 + (ch ^ vswz(d))) * 16) = r[i];
    }
  };
  const int krow_a = (fr >> 2) * 8 + (fr & 3);
  auto kfrag = [&](const char* buf, int tile, int kb) __attribute__((always_inline)) {
    const int row = krow_a + tile * 4;
    return *reinterpret_cast<const bf16x8*>(buf + (row * KCH + ((kb * 4 + fq) ^ kswz(row))) * 16);
  };
  auto vfrag = [&](const char* buf, int n) __attribute__((always_inline)) {
    const int d = n * 16 + fr;
    return *reinterpret_cast<const bf16x8*>(buf + (d * 4 + (fq ^ vswz(d))) * 16);
  };
  auto scores = [&](const char* kbuf, int kb0, int qt, float (&s)[8]) __attribute__((always_inline)) {
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag(kbuf, 0, kb), qf[qt][kb], sa, 0, 0, 0);
      sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag(kbuf, 1, kb), qf[qt][kb], sb, 0, 0, 0);
    }
    const int qlo = qr0 + qt * 16;
    const bool interior = (kb0 + 32 <= len) && !keep && !(a.causal && kb0 + 31 > qlo);
    if (interior) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float v = rbf(r < 4 ? sa[r & 3] : sb[r & 3]);
        if (a.scale != 1.0f) v = rbf(v * a.scale);
        s[r] = v * LOG2E;
      }
    } else {
      const int qpos = qlo + fr;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int j = kb0 + fq * 8 + r;
        float v = rbf(r < 4 ? sa[r & 3] : sb[r & 3]);
        if (a.scale != 1.0f) v = rbf(v * a.scale);
        bool allowed = !(a.causal && j > qpos);
        if (keep && j < len) allowed = allowed && (keep[j] != 0);
        s[r] = j < len ? (allowed ? v * LOG2E : MASKZ) : -INFINITY;
      }
    }
  };

  // key range of the WORKGROUP (uniform): causal -> up to its last query row
  int kend = a.causal ? ((bq0 + 4 * QROWS) < len ? (bq0 + 4 * QROWS) : len) : len;
  float m[QT], l[QT];
  for (int attempt = 0; attempt < 2; ++attempt) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m[qt] = -INFINITY; l[qt] = 0.f; }
    // Key blocks travel global -> registers -> LDS.  A block is requested TWO iterations before it is read (two register sets
    // ka / kb_): with one iteration of look-ahead every 32-key step waited for a full L2 / HBM round trip (the matrix pipe
    // was busy 8.7 % of the time on the Llama prefill shape, PMC).
    auto p1_step = [&](int kb0, int cur) __attribute__((always_inline)) {
      const char* kbuf = smem + cur * KTILE;
      if (active) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          float s[8];
          scores(kbuf, kb0, qt, s);
          float bm = s[0];
#pragma unroll
          for (int r = 1; r < 8; ++r) bm = fmaxf(bm, s[r]);
          bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
          bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
          const float mn = fmaxf(m[qt], bm);
          float bs = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) bs += __builtin_amdgcn_exp2f(s[r] - mn);
          bs += __shfl_xor(bs, 16, 64);
          bs += __shfl_xor(bs, 32, 64);
          l[qt] = l[qt] * __builtin_amdgcn_exp2f(m[qt] - mn) + bs;
          m[qt] = mn;
        }
      }
    };
    {
      KRegs ka = fetch_k(0), kb_;
      lds_barrier();                         // previous readers of buffer 0 are done
      put_k(smem, ka);
      ka = fetch_k(32);                        // block 1 -> ka
      lds_barrier();
#define PCY_P1_STEP(NEXT, FAR)                                                          \
      {                                                                                 \
        FAR = fetch_k(kb0 + 64);      /* block i+2; past the end: clamped rows, never read (unconditional: a conditional */ \
        p1_step(kb0, cur);            /* definition made the compiler merge the two register sets and wait for the load */ \
        put_k(smem + (cur ^ 1) * KTILE, NEXT);   /* block i+1                              it had just issued) */       \
        lds_barrier();                                                                   \
        kb0 += 32; cur ^= 1;                                                            \
      }
      int kb0 = 0, cur = 0;
      while (kb0 < kend) {
        PCY_P1_STEP(ka, kb_)
        if (kb0 >= kend) break;
        PCY_P1_STEP(kb_, ka)
      }
#undef PCY_P1_STEP
    }
    bool empty_row = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) empty_row = empty_row || (active && (m[qt] == MASKZ) && (qr0 + qt * 16 + fr) < len);
    // workgroup-uniform decision (every wave must walk the same key blocks)
    if (attempt == 0 && kend < len && __syncthreads_or(empty_row ? 1 : 0)) { kend = len; continue; }
    break;
  }

  float rl[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) rl[qt] = 1.0f / l[qt];
  f32x4 oacc[QT][NT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int n = 0; n < NT; ++n) oacc[qt][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  char* kb_base = smem;
  char* vb_base = smem + 2 * KTILE;
  auto p2_step = [&](int kb0, int cur) __attribute__((always_inline)) {
    const char* kbuf = kb_base + cur * KTILE;
    const char* vbuf = vb_base + cur * VTILE;
    if (active) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float s[8];
        scores(kbuf, kb0, qt, s);
        bf16x8 pf;
#pragma unroll
        for (int r = 0; r < 8; ++r) pf[r] = (short)f2bf(__builtin_amdgcn_exp2f(s[r] - m[qt]) * rl[qt]);
#pragma unroll
        for (int n = 0; n < NT; ++n) oacc[qt][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vfrag(vbuf, n), oacc[qt][n], 0, 0, 0);
      }
    }
  };
  {
    KRegs ka = fetch_k(0), kb_;
    VRegs va = fetch_v(0), vb_;
    lds_barrier();
    put_k(kb_base, ka); put_v(vb_base, va);
    ka = fetch_k(32); va = fetch_v(32);
    lds_barrier();
#define PCY_P2_STEP(KN, VN, KF, VF)                                                                        \
    {                                                                                                      \
      KF = fetch_k(kb0 + 64); VF = fetch_v(kb0 + 64);                                                        \
      p2_step(kb0, cur);                                                                                   \
      put_k(kb_base + (cur ^ 1) * KTILE, KN); put_v(vb_base + (cur ^ 1) * VTILE, VN);                        \
      lds_barrier();                                                                                        \
      kb0 += 32; cur ^= 1;                                                                                 \
    }
    int kb0 = 0, cur = 0;
    while (kb0 < kend) {
      PCY_P2_STEP(ka, va, kb_, vb_)
      if (kb0 >= kend) break;
      PCY_P2_STEP(kb_, vb_, ka, va)
    }
#undef PCY_P2_STEP
  }
  if (!active) return;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qr0 + qt * 16 + fq * 4 + r;
      if (qq >= len) continue;
      bf16_t* op = a.o + (size_t)(t0 + qq) * a.ldo + h * DH + fr;
#pragma unroll
      for (int n = 0; n < NT; ++n) op[n * 16] = f2bf(oacc[qt][n][r]);
    }
}

#include "pcy_attn_dec.h"
#include "pcy_attn_fast.h"

template <int DH, int G, int DS>
__global__ __launch_bounds__(512) void attn_dec_kernel(PcyDecAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  attn_dec_body<DH, G, DS>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The same with the new token's q / k / v taken from the K-split partial sums of the batched qkv projection (a.qkv_partials):
// the cache rows are requested first, then 192 threads add the splits of the kv head's G + 2 rows in split order into LDS.
template <int DH, int G, int DS>
__global__ __launch_bounds__(512) void attn_dec_splitk_kernel(PcyDecAttnArgs a, int stage_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* stage = reinterpret_cast<bf16_t*>(smem + stage_off);   // [G + 2][DH]
  const int kvh = blockIdx.y, b = blockIdx.z;
  const float* ws = a.qkv_partials;
  const int splits = a.qkv_splits, H = a.H, Hkv = a.Hkv, B = a.B, ld = a.ld;
  a.staged = stage;
  // The first two splits are requested HERE, in front of the cache rows the body asks for (a CU's loads return in order: requested from the
  // hook, behind ~200 KB of K tiles and V rows, they came back last and the rope -- the head of the chain -- waited for every prefetched
  // cache row), and the barrier of the hook leaves the cache rows in flight (round 6).
  const int tid0 = pcy_tid();
  constexpr int NV4 = (G + 2) * DH / 4;
  static_assert(NV4 <= 512, "one quad per thread");
  const int seg = tid0 / (DH / 4), e4 = tid0 % (DH / 4);
  const int n = (seg < G ? (kvh * G + seg) : (seg == G ? H + kvh : H + Hkv + kvh)) * DH + e4 * 4;
  f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
  if (tid0 < NV4) {
    p0 = *reinterpret_cast<const f32x4*>(ws + (size_t)b * ld + n);
    if (splits > 1) p1 = *reinterpret_cast<const f32x4*>(ws + ((size_t)B + b) * ld + n);
  }
  auto hook = [&]() __attribute__((always_inline)) {
    if (tid0 < NV4) {
      f32x4 v = p0;
      if (splits > 1) { v[0] += p1[0]; v[1] += p1[1]; v[2] += p1[2]; v[3] += p1[3]; }
      for (int s_ = 2; s_ < splits; ++s_) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(ws + ((size_t)s_ * B + b) * ld + n);
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
      }
      *reinterpret_cast<uint2*>(stage + seg * DH + e4 * 4) = make_uint2(pack_bf(rbf(v[0]), rbf(v[1])), pack_bf(rbf(v[2]), rbf(v[3])));
    }
    lds_barrier();
  };
  attn_dec_body<DH, G, DS>(a, smem, blockIdx.x, kvh, b, hook);
}

template <int DH, int G, int DS>
void launch_dec_ds(hipStream_t s, const PcyDecAttnArgs& a) {
  if (a.qkv_partials) {
    const int stage_off = (int)((attn_dec_smem_bytes(G, DS, DH, a.Tmax) + 15) & ~(size_t)15);
    const size_t smem = (size_t)stage_off + (size_t)(G + 2) * DH * 2;
    static PcyLdsAttr lds;
    lds.ensure(&attn_dec_splitk_kernel<DH, G, DS>, smem);
    hipLaunchKernelGGL((attn_dec_splitk_kernel<DH, G, DS>), dim3(DH / DS, a.Hkv, a.B), dim3(512), smem, s, a, stage_off);
    return;
  }
  const size_t smem = attn_dec_smem_bytes(G, DS, DH, a.Tmax);
  static PcyLdsAttr lds;
  lds.ensure(&attn_dec_kernel<DH, G, DS>, smem);
  hipLaunchKernelGGL((attn_dec_kernel<DH, G, DS>), dim3(DH / DS, a.Hkv, a.B), dim3(512), smem, s, a);
}

// ------------------------------------------------------------------------------------------------
// Decode attention + output projection in ONE launch (batch 1).  The attention of a decode step is a latency chain
// (~14 us at t = 512..768) during which the weight stream stands still, and the o projection that follows needs 33.5 MB
// of weights that do not depend on the attention at all.  Here workgroups [0, n_attn) run the attention body unchanged
// (output written through to memory, then one flag per workgroup), and the remaining workgroups -- one per CU, 220 VGPRs
// keep two of these from sharing a CU -- pull their RW x 8 waves rows of Wo into REGISTERS while the attention runs
// (RW x K x 2 B / 64 lanes = RW x 8 x 16 B per lane for K = 4096), wait for the flags, fetch the attention output with
// agent-scope loads and finish with the same per-lane accumulation order, reduction tree and rounding points as
// gemv_stream_kernel<1, EPI_RESID, false, 2> -- bit-identical to the two-launch path.
// Attention workgroups have the lowest indices (dispatched first) and never wait for anyone, so the launch cannot
// dead-lock whatever the residency; the wait carries a watchdog that sets *err and lets the kernel finish.
__device__ __forceinline__ void ld4_asm_nt(const void* p, u32x4_t& r0, u32x4_t& r1, u32x4_t& r2, u32x4_t& r3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off nt\n\t"
      "global_load_dwordx4 %1, %4, off offset:1024 nt\n\t"
      "global_load_dwordx4 %2, %4, off offset:2048 nt\n\t"
      "global_load_dwordx4 %3, %4, off offset:3072 nt"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld4_asm_sc1(const void* p, u32x4_t& r0, u32x4_t& r1, u32x4_t& r2, u32x4_t& r3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
      "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
      "global_load_dwordx4 %3, %4, off offset:3072 sc1"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(p) : "memory");
}

template <int DH, int G, int RW>
__global__ __launch_bounds__(512) void attn_o_kernel(PcyDecAttnArgs a, PcyGemvArgs o, int n_attn, const unsigned* epoch_p,
                                                      unsigned* flags, unsigned* err, int dbg, int delay) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x < n_attn) {
    constexpr int slices = DH / 16;
    const int unit = blockIdx.x;
    a.xepoch = *epoch_p;   // (a.xflags / a.xmin set by the launcher; the epoch lives in device memory: graph replay freezes arguments)
    a.xerr = err;
    // unit -> (column slice, kv head): workgroup b runs on XCD b % 8, so with the kv head in the LOW digits the DH/16 slices of one
    // kv head share an XCD -- and its L2: the head's K panel and V rows leave HBM once instead of once per slice
    if (dbg != 3) {
      if (a.unit_map) attn_dec_body<DH, G, 16>(a, smem, (unit / a.Hkv) % slices, unit % a.Hkv, unit / (slices * a.Hkv));
      else attn_dec_body<DH, G, 16>(a, smem, unit % slices, (unit / slices) % a.Hkv, unit / (slices * a.Hkv));
    }
    // every wave: its (agent-scope, written-through) output stores have left; then the workgroup's flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + unit, *epoch_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // ---- output projection: K == 4096 (8 x 16 B per lane and row), rows [r0, r0 + RW) of this wave ----
  constexpr int KC = 8;
  const int K = o.K;
  const int r0 = ((blockIdx.x - n_attn) * 8 + wave) * RW;
  const bool active = r0 < o.N;
  u32x4_t wv[RW][KC];
  float res[RW], bia[RW];
  if (dbg == 2) return;
  if (delay > 0) {   // let the latency-critical first loads of the attention workgroups go ahead of the weight stream
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)delay) __builtin_amdgcn_s_sleep(8);
  }
  if (active) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int r = r0 + i < o.N ? r0 + i : o.N - 1;
      // (rotated k order of row r, PcyGemvArgs::krot: register c holds k-iteration (c + r / 4) % 8)
      const bf16_t* p = o.W + (size_t)r * K + lane * 8;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const uint4 t = ldg_nt(p + ((c + (r >> 2)) & 7) * 512);
        wv[i][c] = (u32x4_t){t.x, t.y, t.z, t.w};
      }
      res[i] = o.resid ? bf2f(o.resid[r]) : 0.f;
      bia[i] = o.bias ? bf2f(o.bias[r]) : 0.f;
    }
  }
  const unsigned epoch = *epoch_p;
  if (wave == 0 && dbg != 1) {
    unsigned spins = 0;
    for (;;) {
      bool ok = true;
      for (int i = lane; i < n_attn; i += 64)
        ok = ok && (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch);
      if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
      __builtin_amdgcn_s_sleep(2);
      if (pcy_wait_give_up(spins, 1u << 18, err, 2u, lane)) break;
    }
  }
  __syncthreads();
  u32x4_t xv[KC];
  ld4_asm_sc1(o.x + lane * 8, xv[0], xv[1], xv[2], xv[3]);
  ld4_asm_sc1(o.x + 2048 + lane * 8, xv[4], xv[5], xv[6], xv[7]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int c = 0; c < KC; ++c) {
    asm volatile("" : "+v"(xv[c]));
#pragma unroll
    for (int i = 0; i < RW; ++i) asm volatile("" : "+v"(wv[i][c]));
  }
  // x through LDS (the attention's buffers are not used by these workgroups): row i wants chunk (c + r_i / 4) % 8 beside weight register c,
  // an index that is not known at compile time
  uint4* xl = reinterpret_cast<uint4*>(smem);   // [KC][64] = 8 KB, one copy for the workgroup (every wave holds the same x)
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < KC; ++c) xl[c * 64 + lane] = make_uint4(xv[c][0], xv[c][1], xv[c][2], xv[c][3]);
  }
  __syncthreads();
  if (!active) return;
  float acc[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) acc[i] = 0.f;
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int r = r0 + i < o.N ? r0 + i : o.N - 1;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const uint4 x4 = xl[((c + (r >> 2)) & 7) * 64 + lane];
      acc[i] = dot8(make_uint4(wv[i][c][0], wv[i][c][1], wv[i][c][2], wv[i][c][3]), x4, acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < RW; ++i) acc[i] = wave_sum(acc[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      if (r0 + i >= o.N) continue;
      float v = rbf(acc[i] + bia[i]);
      if (o.resid) v = rbf(v + res[i]);
      o.y[r0 + i] = f2bf(v);
    }
  }
}

template <int DH, int G>
bool launch_attn_o_rw(hipStream_t s, PcyDecAttnArgs a, const PcyGemvArgs& o, int n_cu, const unsigned* epoch, unsigned* flags, unsigned* err,
                      unsigned* xflags) {
  const int n_attn = (DH / 16) * a.Hkv * a.B;
  const int n_o = n_cu - n_attn;
  if (n_o < 64) return false;
  const int rw = (o.N + n_o * 8 - 1) / (n_o * 8);
  if (rw > 4) return false;
  a.o_sc1 = 1;
  // key split between the slice workgroups from PCY_AO_XMIN cached keys on (0 = never; read per call so that tests can compare):
  // the exchange costs ~3 us per layer (eight gather loads per thread in flight; ~4 us with a load -> LDS store pair per
  // iteration), the K reads it saves 9 ns per key -- measured decode step at t ~ 540 / 660 / 1700 / 3100: 3.228 / 3.264 / 3.67 /
  // 4.29 ms without, 3.238 / 3.262 / 3.44 / 3.92 ms with the split; on from 768 keys
  const char* xe = getenv("PCY_AO_XMIN");
  const int xmin = xe ? atoi(xe) : 768;
  a.xflags = (xmin > 0 && a.scratch) ? xflags : nullptr;
  a.xmin = xmin;
  a.unit_map = 1;
  // The attention issues all of its cache reads (<= 1024 keys) in its first microsecond; 33.5 MB of weight reads queued at
  // the same moment delay them (attention workgroups alone 15.2 us, beside the immediate weight stream 17.9 us).  The o
  // workgroups therefore start ~5 us late: decode step 3.335 (no delay) -> 3.285 (4 us) -> 3.277 ms (8 us) at t = 512..768;
  // 5 us still leaves the stream (~6 us) inside the shortest attention.
  constexpr int dbg = 0, delay = 500;   // delay in 10 ns ticks
  size_t smem = attn_dec_smem_bytes(G, 16, DH, a.Tmax);
  if (smem < 8192) smem = 8192;   // (the o workgroups keep x in 8 KB of it)
  const dim3 grid(n_attn + (o.N + rw * 8 - 1) / (rw * 8)), block(512);
#define PCY_AO_LAUNCH(RWV)                                                                                          \
  do {                                                                                                              \
    static PcyLdsAttr lds;                                                                                          \
    lds.ensure(&attn_o_kernel<DH, G, RWV>, smem);                                                                   \
    static PcyResidentCache res;                                                                                    \
    if (!res.check(smem, [&] { return pcy_all_resident(attn_o_kernel<DH, G, RWV>, 512, smem, (int)grid.x, n_cu); })) return false; \
    hipLaunchKernelGGL((attn_o_kernel<DH, G, RWV>), grid, block, smem, s, a, o, n_attn, epoch, flags, err, dbg, delay); \
  } while (0)
  switch (rw) {
    case 1: PCY_AO_LAUNCH(1); break;
    case 2: PCY_AO_LAUNCH(2); break;
    case 3: PCY_AO_LAUNCH(3); break;
    default: PCY_AO_LAUNCH(4); break;
  }
#undef PCY_AO_LAUNCH
  return true;
}


// ------------------------------------------------------------------------------------------------
// Decode layer (batch 1): qkv projection + attention + o projection + MLP of one decoder layer in ONE launch.
//
// The attention of a decode step is a latency chain, the projections around it are bandwidth.  As launches (qkv GEMV,
// attention + o, gate/up, down) the chain starts only when the whole qkv vector is in memory and a kernel boundary later, and
// every stage pays a boundary and a ramp.  Here
//   workgroups [0, n_attn)   run attn_dec_body unchanged: the cache rows (keys, V slices), the rotary rows and the position are
//                            requested in the first cycle -- they do not depend on the new token -- and only then the body's
//                            inputs_ready hook waits for the G + 2 rows of the new token's q / k / v that THIS kv head needs
//                            (tagged words, pcy_handover.h), staged in LDS;
//   workgroups [n_attn, 256) hold their 4 rows of Wqkv in registers (32 KB per wave, requested right behind x), project,
//                            store tagged; then (the first 128 of them) pull 4 rows of Wo into the same registers while the
//                            attention runs -- and everyone the first 16 KB of its gate/up rows -- take the attention output
//                            (tagged) and finish with the residual epilogue.
// The o projection hands the residual stream over as tagged words (p.xo_tag), every workgroup -- attention ones included --
// takes it into LDS (the second RMSNorm needs all of it: a barrier in all but name) and runs mc_mlp_body (pcy_mlp_chain.h); the
// weight stream restarts from registers.  The kernel boundary that remains is the first RMSNorm of the next layer.
// Attention workgroups have the lowest indices and wait only for projection workgroups, which wait for nobody before their
// stores: no dead-lock whatever the residency.  Per-row arithmetic, accumulation order and rounding points are those of
// gemv_stream_kernel (RMSNorm statistics summed with the stand-alone launches' thread counts) and of attn_o_kernel.
// x_in_lines / x_out_lines (decode_step_kernel: all layers in one launch): the residual stream enters / leaves the layer as a
// tagged vector with one line per producing workgroup (mc_fetch_vector_lines) instead of through p.x / mc.x_out.
template <int DH, int G>
__device__ __forceinline__ void decode_layer_body(PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, int n_attn,
                                                  unsigned xepoch, int vthr_qkv, size_t stage_off, int vthr_gu, char* smem,
                                                  const uint32_t* x_in_lines, uint32_t* x_out_lines, unsigned long long* tr_base, int pf_off) {
  const int tid = pcy_tid(), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t tag = *p.epoch & 0xffffu;
  unsigned long long* tr = tr_base ? tr_base + (size_t)blockIdx.x * 16 : nullptr;
#define AB_T(i) if (tr && tid == 0) tr[i] = wall_clock64();
  AB_T(0)
  if ((int)blockIdx.x < n_attn) {
    constexpr int slices = DH / 16;
    const int unit = blockIdx.x;
    const int bx = (unit / a.Hkv) % slices, kvh = unit % a.Hkv;   // kv head in the low digits: the slices of a head share an XCD's L2
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
    attn_dec_body<DH, G, 16>(a, smem, bx, kvh, 0, hook);
    AB_T(2)
    {
      uint4 wa[16], wb[16];
      if (wave < 7) mc_prime_gate_up(mc, lane, (int)blockIdx.x * 7 + wave, wa, wb, true);   // 32 KB per wave while x is on its way
      __syncthreads();                                   // the attention's LDS is dead
      bf16_t* xr = reinterpret_cast<bf16_t*>(smem) + mc.d + mc.F;
      mc_fetch_vector(p.xo_tag, mc.d, 7, tag, xr, p.err, 12u);   // by the wave that has no gate/up rows (and nothing in flight)
      AB_T(3)
      mc_mlp_body<true>(mc, smem, vthr_gu, tag, gridDim.x, blockIdx.x, 3, wa, wb, tr ? tr + 8 : nullptr, x_out_lines);
      AB_T(5)
    }
    return;
  }
  // ---- projection workgroups ----
  const int d = p.d, K = a.H * DH;
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);             // [d]  RMSNorm(x) * ln1
  bf16_t* xa = xs + d;                                      // [K]  attention output
  float* red = reinterpret_cast<float*>(xa + K);
  bf16_t* xin = reinterpret_cast<bf16_t*>(red + 64);        // [d]  the layer's input when it arrives as a tagged vector
  const int gwo = ((int)blockIdx.x - n_attn) * 8 + wave;
  uint4 w[32];
  // qkv rows [rq0, rq0 + 4): all 32 KB of this wave requested right behind x (in front of the wait for a tagged x)
  const int rq0 = gwo * 4;
  auto load_wqkv = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int it = 0; it < 8; ++it) w[i * 8 + it] = ldg_nt(p.wqkv + (size_t)(rq0 + i) * d + ((((it + (rq0 >> 2)) & 7) * 64 + lane) * 8));
  };
  const bf16_t* xsrc = p.x;
  if (x_in_lines) {
    // half of the rows in front of the loads that fetch x, half behind them (a CU's loads return in order: the fetch waits for whatever
    // was requested before it; round 5, measured on the small-batch step: all in front / all behind / half and half = 3.41 / 3.36 / 3.32 ms)
    auto load_rows = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
      for (int i = i0; i < i0 + 2; ++i)
#pragma unroll
        for (int it = 0; it < 8; ++it) w[i * 8 + it] = ldg_nt(p.wqkv + (size_t)(rq0 + i) * d + ((((it + (rq0 >> 2)) & 7) * 64 + lane) * 8));
    };
    if (PCY_STEP_SPLIT_WQKV) {
      load_rows(0);
      mc_fetch_vector_lines(x_in_lines, d, 7, tag, xin, p.err, 14u, [&]() __attribute__((always_inline)) { load_rows(2); });
    } else {
      load_wqkv();
      mc_fetch_vector_lines(x_in_lines, d, 7, tag, xin, p.err, 14u);
    }
    xsrc = xin;
    mc_rms_stage(xin, p.ln1, d, vthr_qkv, p.rms_eps, p.rms_cast, xs, red, []() __attribute__((always_inline)) {});
  } else {
    mc_rms_stage(p.x, p.ln1, d, vthr_qkv, p.rms_eps, p.rms_cast, xs, red, load_wqkv);
  }
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const uint4 xv = *reinterpret_cast<const uint4*>(xs + (((it + (rq0 >> 2)) & 7) * 64 + lane) * 8);   // (rotated k order, PcyGemvArgs::krot)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = dot8(w[i * 8 + it], xv, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);
    // the workgroup's 32 rows = one 128-byte line of the tagged vector, stored by one instruction
    uint32_t* line = reinterpret_cast<uint32_t*>(red) + 16;
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) line[wave * 4 + i] = (tag << 16) | f2bf(rbf(acc[i]));
    }
    __syncthreads();
    if (wave == 0 && lane < 32) __hip_atomic_store(p.qkv_tag + (rq0 & ~31) + lane, line[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  AB_T(1)
  // o rows [r0, r0 + 4) (the first d / 32 projection workgroups: 32 rows = ONE 128-byte line of the result per workgroup -- words
  // of a line stored one by one from several CUs took 3.6 us to become visible, a line written by one instruction ~1): into
  // the same registers while the attention runs.  (While the q / k / v rows were stored word by word a 2 us pause in front of
  // these 33 MB helped the attention workgroups' requests through -- 2.91 -> 2.83 ms/token; with one line per workgroup the
  // pause no longer matters: 0 / 0.5 / 1 / 2 / 3 us = 2.64 / 2.63 / 2.64 / 2.65 / 2.67 ms.)
  const int r0 = gwo * 4;
  const bool active = r0 < d;                 // (workgroup-uniform: d % 32 == 0)
  uint4 wa[16], wb[16];
  float res[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = r0 + i;
#pragma unroll
      for (int c = 0; c < 8; ++c) w[i * 8 + c] = ldg_nt(p.wo + (size_t)r * K + (((c + (r0 >> 2)) & 7) * 64 + lane) * 8);
      res[i] = bf2f(xsrc[r]);
    }
  } else if (wave < 7) {   // no o rows here: the SECOND 16 KB of the wave's gate/up rows wait in the Wo registers instead
    const McRowG row_g{mc.F, mc.d};
    const int gidx = (int)blockIdx.x * 7 + wave;
#pragma unroll
    for (int un = 0; un < 2; ++un)
#pragma unroll
      for (int i = 0; i < 8; ++i) w[un * 8 + i] = ldg_nt(mc.wgu + row_g(gidx, i) + (mc_rot(2 + un, gidx & 7, 8) * 64 + lane) * 8);
  }
  // gate/up rows of the MLP while the attention runs: 16 KB per wave beside the Wo rows (both batches: 256 VGPRs and spills)
  if (wave < 7) mc_prime_gate_up(mc, lane, (int)blockIdx.x * 7 + wave, wa, wb, false);
  // ... and the THIRD batch (k-iterations 4, 5) into LDS, which is idle here (pf_off: behind everything this workgroup keeps in LDS; 16 KB
  // per wave): the registers are full, and without it the weight stream stood still for the last ~5 us of the attention (in-kernel stamps:
  // 73 MB requested at 14 us have landed by 26, the MLP starts at 33).  The second batch is requested as before once the Wo registers are free.
  if (pf_off > 0 && wave < 7 && (int)blockIdx.x * 7 + wave < (mc.F + 3) / 4)
    mc_lds_prefetch(mc, lane, (int)blockIdx.x * 7 + wave, MC_LDS_IT0, smem + pf_off + wave * 16384);
  // the attention output: one wave watches a 1 KB sample (192 workgroups asking for all 16 KB in a loop would load the fabric
  // while the attention workgroups are inside their latency chain), then every wave takes its share
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
  __syncthreads();
  AB_T(2)
  {
    uint4 tq[2];
    mc_fetch_issue<2>(p.ao_tag, wave * 512, lane, tq);
    mc_fetch_finish<2>(p.ao_tag, wave * 512, lane, tag, xa, tq, p.err, 11u);
  }
  __syncthreads();
  AB_T(3)
  if (active) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 xv = *reinterpret_cast<const uint4*>(xa + (((c + (r0 >> 2)) & 7) * 64 + lane) * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = dot8(w[i * 8 + c], xv, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);
    uint32_t* line = reinterpret_cast<uint32_t*>(red);   // the workgroup's 32 results, stored as one line by one wave
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = rbf(acc[i]);
        v = rbf(v + res[i]);
        line[wave * 4 + i] = f2bf(v);
      }
    }
    __syncthreads();
    if (wave == 0 && lane < 32) {
      const int r = (r0 & ~31) + lane;   // r0 of wave 0 is the workgroup's first row
      __hip_atomic_store(p.xo_tag + r, (tag << 16) | line[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  AB_T(4)
  {
    // the Wo registers are free: the second 16 KB of this wave's gate/up rows while the residual stream is on its way
    if (wave < 7) {
      if (active) mc_prime<8, 2, 2>(mc.wgu, mc.d, lane, (int)blockIdx.x * 7 + wave, (int)gridDim.x * 7, (mc.F + 3) / 4, wa, wb, McRowG{mc.F, mc.d}, McShiftG{mc.d >> 9});
      else {
#pragma unroll
        for (int i = 0; i < 16; ++i) wb[i] = w[i];
      }
    }
    __syncthreads();                                     // every wave is done with the attention output in LDS
    bf16_t* xr = reinterpret_cast<bf16_t*>(smem) + mc.d + mc.F;
    mc_fetch_vector(p.xo_tag, mc.d, 7, tag, xr, p.err, 13u);
    mc_mlp_body<true>(mc, smem, vthr_gu, tag, gridDim.x, blockIdx.x, 3, wa, wb, tr ? tr + 8 : nullptr, x_out_lines, pf_off);
    AB_T(5)
  }
#undef AB_T
}

template <int DH, int G>
__global__ __launch_bounds__(512) void decode_layer_kernel(PcyDecAttnArgs a, PcyAttnBlockArgs p, PcyMlpChainArgs mc, int n_attn,
                                                           const unsigned* step_epoch, int vthr_qkv, size_t stage_off, int vthr_gu, int pf_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  decode_layer_body<DH, G>(a, p, mc, n_attn, *step_epoch, vthr_qkv, stage_off, vthr_gu, smem, nullptr, nullptr, p.trace, pf_off);
}

// All decoder layers of a batch-1 decode step in ONE launch: workgroup b runs layer after layer; the residual stream crosses the
// layer boundary as a tagged vector (one line per workgroup, pcy_handover.h) instead of a kernel boundary, so a workgroup that
// has stored its down rows requests its next Wqkv rows (or cache rows) at once: no launch ramp, and the spread of the
// workgroups' finishing times (6-8 us per layer) is absorbed by the next layer's prefetch instead of waited out.
template <int DH, int G>
__global__ __launch_bounds__(512) void decode_step_kernel(PcyDecAttnArgs a, PcyAttnBlockArgs p, PcyMlpChainArgs mc, PcyDecodeStepArgs st, int n_attn,
                                                          const unsigned* step_epoch, int vthr_qkv, size_t stage_off, int vthr_gu, int pf_off) {
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
    mc.act_tag = tags; p.qkv_tag = tags + mc.F; p.ao_tag = p.qkv_tag + p.Nq; p.xo_tag = p.ao_tag + a.H * DH;
    const uint32_t* xin = l > 0 ? st.x_lines + (size_t)(l - 1) * st.x_lines_stride : nullptr;
    uint32_t* xout = l + 1 < st.n_layers ? st.x_lines + (size_t)l * st.x_lines_stride : nullptr;
    if (l > 0) __syncthreads();   // the previous layer's LDS is dead
    decode_layer_body<DH, G>(al, p, mc, n_attn, xepoch, vthr_qkv, stage_off, vthr_gu, smem, xin, xout,
                             p.trace ? p.trace + (size_t)l * 256 * 16 : nullptr, pf_off);
  }
}

template <int DH, int G>
bool launch_decode_layer(hipStream_t s, PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, const unsigned* step_epoch,
                         unsigned* xflags, int n_cu, const PcyDecodeStepArgs* st = nullptr) {
  const int n_attn = (DH / 16) * a.Hkv, n_o = 256 - n_attn;
  if (n_o < 64 || p.Nq != n_o * 8 * 4 || p.d > n_o * 8 * 4 || a.H * DH != 8 * 512 || p.d != 4096) return false;
  // the MLP body: same geometry conditions as pcy_launch_mlp_chain
  if (mc.d != p.d || mc.d != 2 * 256 * 8 || mc.F != 2 * 7 * 1024) return false;
  a.o_sc1 = 0;
  const char* xe = getenv("PCY_AO_XMIN");   // key split between the slice workgroups (see launch_attn_o_rw)
  const int xmin = xe ? atoi(xe) : 768;
  a.xflags = (xmin > 0 && a.scratch) ? xflags : nullptr;
  a.xmin = xmin;
  a.unit_map = 1;
  const size_t stage_off = (attn_dec_smem_bytes(G, 16, DH, a.Tmax) + 15) & ~(size_t)15;
  const size_t smem_attn = stage_off + (size_t)(G + 2) * DH * 2, smem_o = (size_t)(2 * p.d + a.H * DH) * 2 + 512;
  const size_t smem_mlp = (size_t)(2 * mc.d + mc.F) * 2 + 128;
  size_t smem = smem_attn > smem_o ? smem_attn : smem_o;
  smem = smem > smem_mlp ? smem : smem_mlp;
  // 7 x 16 KB behind the projection workgroups' own LDS for the gate/up batch they prefetch while the attention runs (PCY_DISABLE=lds_prefetch:
  // without; same bits); the attention workgroups do not touch it
  size_t pf_off = ((smem_o > smem_mlp ? smem_o : smem_mlp) + 1023) & ~(size_t)1023;
  if (pcy_off("lds_prefetch") || pf_off + 7 * 16384 > 160 * 1024) pf_off = 0;
  if (pf_off && pf_off + 7 * 16384 > smem) smem = pf_off + 7 * 16384;
  static PcyLdsAttr lds[2];
  if (st) lds[1].ensure(&decode_step_kernel<DH, G>, smem);
  else lds[0].ensure(&decode_layer_kernel<DH, G>, smem);
  // every workgroup waits for words the others write: all 256 must be resident at once (occupancy query, cached per kernel / device / LDS size)
  static PcyResidentCache res[2];
  if (!res[st ? 1 : 0].check(smem, [&] {
        return st ? pcy_all_resident(decode_step_kernel<DH, G>, 512, smem, 256, n_cu) : pcy_all_resident(decode_layer_kernel<DH, G>, 512, smem, 256, n_cu);
      }))
    return false;
  if (st)
    hipLaunchKernelGGL((decode_step_kernel<DH, G>), dim3(256), dim3(512), smem, s, a, p, mc, *st, n_attn, step_epoch, pcy_gemv_rms_threads(p.Nq),
                       stage_off, pcy_gemv_rms_threads(mc.F), (int)pf_off);
  else
    hipLaunchKernelGGL((decode_layer_kernel<DH, G>), dim3(256), dim3(512), smem, s, a, p, mc, n_attn, step_epoch, pcy_gemv_rms_threads(p.Nq),
                       stage_off, pcy_gemv_rms_threads(mc.F), (int)pf_off);
  return true;
}

template <int DH, int G>
void launch_dec(hipStream_t s, const PcyDecAttnArgs& a) {
  // widest slice that still leaves >= ~1 workgroup per CU 
  constexpr int force = 0;
  int ds = 16;
  if constexpr (DH == 128) {
    const int units = a.Hkv * a.B;
    if (units >= 128) ds = 128;
    else if (units >= 32) ds = 64;   // (from 4 rows of 8 kv heads on: 16-column workgroups score every key 8 times -- 5-7 rows 4.17-4.23 -> 3.68-3.69 ms per step)
    if (force == 16 || force == 64 || force == 128) ds = force;
    if (a.force_ds == 16 || a.force_ds == 32 || a.force_ds == 64 || a.force_ds == 128) ds = a.force_ds;   // (the twin of the small-batch step)
    if (ds == 128) return launch_dec_ds<DH, G, 128>(s, a);
    if (ds == 64) return launch_dec_ds<DH, G, 64>(s, a);
    if constexpr (G == 4) { if (ds == 32) return launch_dec_ds<DH, G, 32>(s, a); }
  }
  launch_dec_ds<DH, G, 16>(s, a);
}

template <int DH>
void launch_dec_g(hipStream_t s, const PcyDecAttnArgs& a) {
  switch (a.H / a.Hkv) {
    case 1: launch_dec<DH, 1>(s, a); break;
    case 2: launch_dec<DH, 2>(s, a); break;
    case 4: launch_dec<DH, 4>(s, a); break;
    default: launch_dec<DH, 8>(s, a); break;
  }
}

}  // namespace

// ONE predicate for the single-pass kernel, shared by pcy_attention, pcy_esm_encode and the launcher: shape (head_dim 64, bidirectional,
// unmasked, unit scale, no grouped heads) AND every alignment the kernel's 16-byte / 8-byte accesses need.  A caller that lays out Vt
// (or skips the transposed copy) for the fast kernel therefore knows the launcher will take it -- it cannot decline afterwards and
// leave the two-pass kernels with a buffer that was never written.
bool pcy_attn_fast_eligible(int dh, int causal, bool has_keep, float scale, int H, int Hkv, int ldq, int qcol0, int ldk, int kcol0, int ldo) {
  const char* e = getenv("PCY_ESM_ATTN");   // "exact" = the reference's rounding points (two-pass kernel); read per call: tests compare
  if (e && e[0] == 'e') return false;
  if (!(dh == 64 && !causal && !has_keep && scale == 1.0f && H == Hkv)) return false;
  return ((ldq | ldk | qcol0 | kcol0) % 8) == 0 && ldo % 4 == 0;
}
// ... and for reading V token-major where the projection wrote it (PCY_DISABLE=fa_vrow: from the transposed copy)
bool pcy_attn_fast_vrow(int ldv, int vcol0) { return !pcy_off("fa_vrow") && ((ldv | vcol0) % 8) == 0; }
void pcy_launch_attn(hipStream_t s, const PcyAttnArgs& a) {
  if (a.nseq <= 0) return;
  if (a.vt_pad64 && pcy_attn_fast_eligible(a.dh, a.causal, a.keep != nullptr, a.scale, a.H, a.Hkv, a.ldq, a.qcol0, a.ldk, a.kcol0, a.ldo) &&
      pcy_launch_attn_fast64(s, a, true)) {
    ++g_pcy_dispatch[PCY_DISPATCH_ATTN_FAST];
    return;
  }
  // head_dim 128 (Llama prefill): the LDS-shared kernel.  One q tile per wave leaves 16 K/Vt fragment loads of 1 KiB per 16 MFMAs
  // to every wave of the register-fragment kernel -- L1/L2 bound (44 TFLOP/s at B = 64, T = 450); fetching each key block once
  // per workgroup: Llama-3-8B pair-scoring prefill 870 -> 932 TFLOP/s (bf16), 1350 -> 1490 (fp8 weights).
  if (a.dh == 128) {
    // two q tiles per wave halve the K / Vt fragment reads per MFMA (the LDS pipe is the limiter at one); capped at 256
    // VGPRs (amdgpu_waves_per_eu(2): no spills -- uncapped the compiler took 260 and one wave per SIMD, slower):
    // 938 -> 975 TFLOP/s (bf16), 1555 -> 1664 (fp8 weights) on the pair-scoring prefill
    // ... but a small grid (one 512-token prompt: 4 x 32 workgroups of two tiles) is a latency chain per workgroup: keep one
    // tile per wave until two-tile workgroups alone fill the chip twice
    const long wg2 = (long)((a.max_len + 127) / 128) * a.H * a.nseq;
    if (wg2 >= 512) hipLaunchKernelGGL((attn_lds_kernel<128, 2>), dim3((a.max_len + 127) / 128, a.H, a.nseq), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_lds_kernel<128, 1>), dim3((a.max_len + 63) / 64, a.H, a.nseq), dim3(256), 0, s, a);
    return;
  }
  // head_dim 64 / 32 (ESM): the register-fragment kernel (the LDS-shared one measured equal at head_dim 64, 341 vs 346 proteins/s:
  // the kernel is VALU-issue bound, not L2 bound).  q rows per block = 4 waves x QT x 16: more q tiles per wave amortise the
  // K / Vt fragment loads and the loop overhead (QT 2 -> 3: 343 -> 396 proteins/s; QT = 4 needs 174 VGPRs and is slower again, 372;
  // forcing 4 waves/SIMD spills: 322)
  const bool scaled = a.scale != 1.0f;
#define PCY_ATTN_LAUNCH(DHV, QTV, ROWS)                                                                                    \
  do {                                                                                                                      \
    const dim3 grid((a.max_len + ROWS - 1) / ROWS, a.H, a.nseq);                                                            \
    if (scaled) hipLaunchKernelGGL((attn_kernel<DHV, QTV, true, true>), grid, dim3(256), 0, s, a);                         \
    else hipLaunchKernelGGL((attn_kernel<DHV, QTV, true, false>), grid, dim3(256), 0, s, a);                               \
  } while (0)
  if (a.dh == 64) PCY_ATTN_LAUNCH(64, 3, 192);
  else PCY_ATTN_LAUNCH(32, 2, 128);
#undef PCY_ATTN_LAUNCH
}

void pcy_launch_attn_decode(hipStream_t s, const PcyDecAttnArgs& a) {
  if (a.dh == 128) launch_dec_g<128>(s, a);
  else if (a.dh == 64) launch_dec_g<64>(s, a);
  else launch_dec_g<32>(s, a);
}

// Fused decode attention + o projection (see attn_o_kernel).  Returns false (nothing launched) when the shape is not
// covered: batch 1, head_dim 128, G in {1,2,4,8}, K = H*dh = 4096 contiguous, residual epilogue without x staging.
bool pcy_launch_attn_o(hipStream_t s, const PcyDecAttnArgs& a, const PcyGemvArgs& o, int n_cu, const unsigned* epoch,
                       unsigned* flags, int max_flags, unsigned* err, unsigned* xflags) {
  if (a.B != 1 || o.B != 1 || a.dh != 128 || o.K != 4096 || o.rms_w || o.epi != EPI_RESID || a.dbg) return false;
  if ((a.dh / 16) * a.Hkv * a.B > max_flags || n_cu > 256) return false;
  switch (a.H / a.Hkv) {
    case 1: return launch_attn_o_rw<128, 1>(s, a, o, n_cu, epoch, flags, err, xflags);
    case 2: return launch_attn_o_rw<128, 2>(s, a, o, n_cu, epoch, flags, err, xflags);
    case 4: return launch_attn_o_rw<128, 4>(s, a, o, n_cu, epoch, flags, err, xflags);
    case 8: return launch_attn_o_rw<128, 8>(s, a, o, n_cu, epoch, flags, err, xflags);
  }
  return false;
}

// One decoder layer of a batch-1 decode step (see decode_layer_kernel).  Returns false (nothing launched) when the shape is not
// covered: batch 1, head_dim 128, H * dh = d = 4096, ffn = 14336, G in {1,2,4,8}, Nq = 4 rows per projection wave, 256 CUs.
bool pcy_launch_decode_layer(hipStream_t s, const PcyDecAttnArgs& a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, int n_cu,
                             const unsigned* step_epoch, unsigned* xflags) {
  if (a.B != 1 || a.dh != 128 || a.dbg || n_cu < 256) return false;
  if (pcy_launch_decode_mha(s, a, p, mc, nullptr, n_cu, step_epoch, xflags)) return true;   // 32 kv heads, ffn 11008 (pcy_decode_mha.hip)
  switch (a.H / a.Hkv) {
    case 1: return launch_decode_layer<128, 1>(s, a, p, mc, step_epoch, xflags, n_cu);
    case 2: return launch_decode_layer<128, 2>(s, a, p, mc, step_epoch, xflags, n_cu);
    case 4: return launch_decode_layer<128, 4>(s, a, p, mc, step_epoch, xflags, n_cu);
    case 8: return launch_decode_layer<128, 8>(s, a, p, mc, step_epoch, xflags, n_cu);
  }
  return false;
}

// All decoder layers of a batch-1 decode step in one launch (see decode_step_kernel); same coverage as pcy_launch_decode_layer.
// a.xflags = base of the per-layer key-split flags (st.xflags_stride apart).
bool pcy_launch_decode_step(hipStream_t s, const PcyDecAttnArgs& a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, const PcyDecodeStepArgs& st,
                            int n_cu, const unsigned* step_epoch) {
  if (a.B != 1 || a.dh != 128 || a.dbg || n_cu < 256 || st.n_layers < 1) return false;
  if (pcy_launch_decode_mha(s, a, p, mc, &st, n_cu, step_epoch, a.xflags)) return true;
  switch (a.H / a.Hkv) {
    case 1: return launch_decode_layer<128, 1>(s, a, p, mc, step_epoch, a.xflags, n_cu, &st);
    case 2: return launch_decode_layer<128, 2>(s, a, p, mc, step_epoch, a.xflags, n_cu, &st);
    case 4: return launch_decode_layer<128, 4>(s, a, p, mc, step_epoch, a.xflags, n_cu, &st);
    case 8: return launch_decode_layer<128, 8>(s, a, p, mc, step_epoch, a.xflags, n_cu, &st);
  }
  return false;
}
