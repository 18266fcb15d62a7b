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
#include "pcy_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attn_kernel(PcyAttnArgs a) {
  constexpr int KB = DH / 32;   // k-blocks of the QK^T contraction
  constexpr int NT = DH / 16;   // 16-wide output tiles of P.V
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int sq = blockIdx.z, h = blockIdx.y;
  const int t0 = a.cu[sq], len = a.cu[sq + 1] - t0;
  const int qr0 = blockIdx.x * 64 + wave * 16;
  if (qr0 >= len) return;
  const int G = a.H / a.Hkv;
  const int kvh = h / G;
  const int vt0 = a.vt_cu[sq];

  // Q fragments (B operand of S^T = K.Q^T): lane holds Q[q = fr][kb*32 + fq*8 .. +8]
  const int qrow = (qr0 + fr) < len ? (qr0 + fr) : len - 1;
  const bf16_t* qp = a.q + (size_t)(t0 + qrow) * a.ldq + a.qcol0 + h * DH + fq * 8;
  bf16x8 qf[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) qf[kb] = *reinterpret_cast<const bf16x8*>(qp + kb * 32);

  const bf16_t* kbase = a.k + (size_t)t0 * a.ldk + a.kcol0 + kvh * DH + fq * 8;
  const uint8_t* keep = a.keep ? a.keep + t0 : nullptr;
  const int qpos = qr0 + fr;
  // A-operand row fr of tile a / tile b maps to key (fr/4)*8 + (fr%4) (+4 for tile b) of the block
  const int krow_a = (fr >> 2) * 8 + (fr & 3);

  // scores of one 32-key block for this lane: keys kb0 + fq*8 + 0..7 of query fr
  auto scores = [&](int kb0, float (&s)[8]) {
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
    int ka = kb0 + krow_a, kbk = ka + 4;
    ka = ka < len ? ka : len - 1;
    kbk = kbk < len ? kbk : len - 1;
    const bf16_t* pa = kbase + (size_t)ka * a.ldk;
    const bf16_t* pb = kbase + (size_t)kbk * a.ldk;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(pa + kb * 32);
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(pb + kb * 32);
      sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, qf[kb], sa, 0, 0, 0);
      sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, qf[kb], sb, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int j = kb0 + fq * 8 + r;
      float v = rbf(r < 4 ? sa[r & 3] : sb[r & 3]);
      if (a.scale != 1.0f) v = rbf(v * a.scale);
      bool allowed = !(a.causal && j > qpos);
      if (keep && j < len) allowed = allowed && (keep[j] != 0);
      v = allowed ? v : PCY_BF16_MIN;
      s[r] = j < len ? v : -INFINITY;
    }
  };

  int kend = a.causal ? ((qr0 + 16) < len ? (qr0 + 16) : len) : len;
  float m, l;
  for (int attempt = 0; attempt < 2; ++attempt) {
    // pass 1: online row max / sum of exp over keys [0, kend)
    m = -INFINITY; l = 0.f;
    for (int kb0 = 0; kb0 < kend; kb0 += 32) {
      float s[8];
      scores(kb0, s);
      float bm = s[0];
#pragma unroll
      for (int r = 1; r < 8; ++r) bm = fmaxf(bm, s[r]);
      bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
      const float mn = fmaxf(m, bm);
      float bs = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) bs += expf(s[r] - mn);
      bs += __shfl_xor(bs, 16, 64);
      bs += __shfl_xor(bs, 32, 64);
      l = l * expf(m - mn) + bs;
      m = mn;
    }
    // a row whose allowed-key set is empty (a left-pad query): the reference's additive finfo.min mask
    // makes its softmax uniform over ALL keys of the sequence, causal or not -> redo over the full range
    const bool empty_row = (m == PCY_BF16_MIN) && (qr0 + fr) < len;
    if (attempt == 0 && kend < len && __any(empty_row)) { kend = len; continue; }
    break;
  }

  // pass 2: P = bf16(exp(S - m) / l), O += P.V
  f32x4 oacc[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) oacc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bf16_t* vbase = a.vt + ((size_t)kvh * DH + fr) * a.vt_total + vt0 + fq * 8;
  for (int kb0 = 0; kb0 < kend; kb0 += 32) {
    float s[8];
    scores(kb0, s);
    bf16x8 pf;
#pragma unroll
    for (int r = 0; r < 8; ++r) pf[r] = (short)f2bf(expf(s[r] - m) / l);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vbase + (size_t)n * 16 * a.vt_total + kb0);
      oacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, oacc[n], 0, 0, 0);
    }
  }
  // O[q = fq*4 + r][d = n*16 + fr]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qq = qr0 + fq * 4 + r;
    if (qq >= len) continue;
    bf16_t* op = a.o + (size_t)(t0 + qq) * a.ldo + h * DH + fr;
#pragma unroll
    for (int n = 0; n < NT; ++n) op[n * 16] = f2bf(oacc[n][r]);
  }
}

// ------------------------------------------------------------------------------------------------
// Decode attention (row A7): one new token per row against the KV cache, exact softmax rounding.
// Two launches so that the work spreads over the chip instead of Hkv*B workgroups:
//   (A) attn_dec_scores: grid (key chunks of 64, Hkv, B).  Every block ropes the G query heads of its kv head in
//       registers (three bf16 roundings, HF Llama), streams its 64 cached keys with 16 lanes x 16 B per key row,
//       writes s = bf16(bf16(q.k)*scale) as fp32 to scratch[B,H,Tmax+1].  The block that owns slot t also ropes the
//       new key, appends K and V to the cache and scores it from registers.
//   (B) attn_dec_pv: grid (DH/16 column slices, Hkv, B).  Every block re-reads the G score rows (a few KB, L2),
//       computes max / sum / p = bf16(exp(s-m)/l) in LDS, then accumulates P.V for its 16 output columns over all keys
//       (fp32), reduces across its key groups in LDS and rounds once -> no cross-workgroup reduction.
template <int DH>
__device__ __forceinline__ void rope8(const bf16_t* __restrict__ x, const bf16_t* __restrict__ cs, const bf16_t* __restrict__ sn,
                                      int e0, float (&out)[8]) {
  // elements e0..e0+7 of one head; partner = e +- DH/2
  constexpr int HALF = DH / 2;
  const bool lo = e0 < HALF;
  const int p0 = lo ? e0 + HALF : e0 - HALF;
  const uint4 a = *reinterpret_cast<const uint4*>(x + e0);
  const uint4 b = *reinterpret_cast<const uint4*>(x + p0);
  const uint4 c = *reinterpret_cast<const uint4*>(cs + e0);
  const uint4 s = *reinterpret_cast<const uint4*>(sn + e0);
  const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w}, cw[4] = {c.x, c.y, c.z, c.w}, sw[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = lo_bf(aw[i]), x1 = hi_bf(aw[i]);
    float r0 = lo_bf(bw[i]), r1 = hi_bf(bw[i]);   // rotate_half partner: -x2 for the low half, +x1 for the high half
    if (lo) { r0 = -r0; r1 = -r1; }
    out[2 * i] = rbf(rbf(x0 * lo_bf(cw[i])) + rbf(r0 * lo_bf(sw[i])));
    out[2 * i + 1] = rbf(rbf(x1 * hi_bf(cw[i])) + rbf(r1 * hi_bf(sw[i])));
  }
}

template <int DH, int G>
__global__ __launch_bounds__(256) void attn_dec_scores_kernel(PcyDecAttnArgs a) {
  constexpr int LPK = DH / 8;    // lanes per key row
  constexpr int NG = 256 / LPK;  // key groups per block
  constexpr int CH = 64;         // keys per block
  const int t = *a.pos_dev;
  const int j0 = blockIdx.x * CH;
  if (j0 > t) return;
  const int kvh = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, grp = tid / LPK, sub = tid % LPK;
  const bf16_t* row = a.qkv + (size_t)b * a.ld;
  bf16_t* kc = a.kcache + ((size_t)b * a.Hkv + kvh) * a.Tmax * DH;
  bf16_t* vc = a.vcache + ((size_t)b * a.Hkv + kvh) * a.Tmax * DH;
  const bf16_t* cs = a.cos_t + (size_t)t * DH;
  const bf16_t* sn = a.sin_t + (size_t)t * DH;
  float q[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g) rope8<DH>(row + (kvh * G + g) * DH, cs, sn, sub * 8, q[g]);
  const uint8_t* keep = a.keep ? a.keep + (size_t)b * a.ld_keep : nullptr;
  const int scld = a.Tmax + 1;
  float* sc = a.scratch + ((size_t)b * a.H + kvh * G) * scld;
#pragma unroll
  for (int i = 0; i < CH / NG; ++i) {
    const int j = j0 + i * NG + grp;
    if (j > t) continue;   // uniform per key group (LPK lanes)
    float kf[8];
    if (j < t) {
      const uint4 kv = *reinterpret_cast<const uint4*>(kc + (size_t)j * DH + sub * 8);
      kf[0] = lo_bf(kv.x); kf[1] = hi_bf(kv.x); kf[2] = lo_bf(kv.y); kf[3] = hi_bf(kv.y);
      kf[4] = lo_bf(kv.z); kf[5] = hi_bf(kv.z); kf[6] = lo_bf(kv.w); kf[7] = hi_bf(kv.w);
    } else {  // the new token: rope its key, append K and V
      rope8<DH>(row + (a.H + kvh) * DH, cs, sn, sub * 8, kf);
      *reinterpret_cast<uint4*>(kc + (size_t)t * DH + sub * 8) =
          make_uint4(pack_bf(kf[0], kf[1]), pack_bf(kf[2], kf[3]), pack_bf(kf[4], kf[5]), pack_bf(kf[6], kf[7]));
      *reinterpret_cast<uint4*>(vc + (size_t)t * DH + sub * 8) =
          *reinterpret_cast<const uint4*>(row + (a.H + a.Hkv + kvh) * DH + sub * 8);
    }
    const bool kept = (keep && j < t) ? keep[j] != 0 : true;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d += q[g][e] * kf[e];
#pragma unroll
      for (int o = LPK / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
      if (sub == 0) sc[(size_t)g * scld + j] = kept ? rbf(rbf(d) * a.scale) : PCY_BF16_MIN;
    }
  }
}

template <int DH, int G>
__global__ __launch_bounds__(256) void attn_dec_pv_kernel(PcyDecAttnArgs a) {
  constexpr int DS = 16;          // output columns per block
  constexpr int LPR = DS / 2;     // lanes per V row (2 columns = 4 B each)
  constexpr int NG = 256 / LPR;   // key groups
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* p = reinterpret_cast<float*>(smem);          // [G][nk]
  const int t = *a.pos_dev;
  const int nk = t + 1;
  float* red = p + (size_t)G * (a.Tmax + 1);           // [NG][G][DS]
  float* wred = red + NG * G * DS;                     // [8]
  const int c0 = blockIdx.x * DS, kvh = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const int scld = a.Tmax + 1;
  const float* sc = a.scratch + ((size_t)b * a.H + kvh * G) * scld;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float mx = -INFINITY;
    for (int j = tid; j < nk; j += 256) { const float v = sc[(size_t)g * scld + j]; p[g * scld + j] = v; mx = fmaxf(mx, v); }
    mx = block_max<256>(mx, wred);
    float se = 0.f;
    for (int j = tid; j < nk; j += 256) se += expf(p[g * scld + j] - mx);
    se = block_sum<256>(se, wred);
    for (int j = tid; j < nk; j += 256) p[g * scld + j] = rbf(expf(p[g * scld + j] - mx) / se);
  }
  __syncthreads();
  const int grp = tid / LPR, sub = tid % LPR;
  const bf16_t* vc = a.vcache + ((size_t)b * a.Hkv + kvh) * a.Tmax * DH + c0 + sub * 2;
  float acc[G][2];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g][0] = acc[g][1] = 0.f;
  for (int j = grp; j < nk; j += NG) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(vc + (size_t)j * DH);
    const float v0 = lo_bf(w), v1 = hi_bf(w);
#pragma unroll
    for (int g = 0; g < G; ++g) { const float pj = p[g * scld + j]; acc[g][0] += pj * v0; acc[g][1] += pj * v1; }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) { red[(grp * G + g) * DS + sub * 2] = acc[g][0]; red[(grp * G + g) * DS + sub * 2 + 1] = acc[g][1]; }
  __syncthreads();
  if (tid < G * DS) {
    float s = 0.f;
    for (int gg = 0; gg < NG; ++gg) s += red[gg * G * DS + tid];
    const int g = tid / DS, c = tid % DS;
    a.o[(size_t)b * a.ldo + (kvh * G + g) * DH + c0 + c] = f2bf(s);
  }
}

template <int DH, int G>
void launch_dec(hipStream_t s, const PcyDecAttnArgs& a) {
  constexpr int NGB = 256 / 8;
  const size_t smem = sizeof(float) * ((size_t)G * (a.Tmax + 1) + (size_t)NGB * G * 16 + 8);
  static size_t configured = 0;
  if (smem > 65536 && smem > configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_dec_pv_kernel<DH, G>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = smem;
  }
  hipLaunchKernelGGL((attn_dec_scores_kernel<DH, G>), dim3((a.Tmax + 63) / 64, a.Hkv, a.B), dim3(256), 0, s, a);
  hipLaunchKernelGGL((attn_dec_pv_kernel<DH, G>), dim3(DH / 16, a.Hkv, a.B), dim3(256), smem, s, a);
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

void pcy_launch_attn(hipStream_t s, const PcyAttnArgs& a) {
  if (a.nseq <= 0) return;
  const dim3 grid((a.max_len + 63) / 64, a.H, a.nseq);
  if (a.dh == 128) hipLaunchKernelGGL((attn_kernel<128>), grid, dim3(256), 0, s, a);
  else if (a.dh == 64) hipLaunchKernelGGL((attn_kernel<64>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_kernel<32>), grid, dim3(256), 0, s, a);
}

void pcy_launch_attn_decode(hipStream_t s, const PcyDecAttnArgs& a) {
  if (a.dh == 128) launch_dec_g<128>(s, a);
  else if (a.dh == 64) launch_dec_g<64>(s, a);
  else launch_dec_g<32>(s, a);
}
