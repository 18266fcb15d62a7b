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
//     64 keys per tile, K [64 keys][64 dh] and Vt [64 dh][64 keys] = 8 KiB each, double buffered (32 KiB of LDS), brought in by
//     LDS-DMA (global_load_lds, 16 B per lane, no staging registers) one tile ahead.  128-byte rows: the 16-byte chunk c of row
//     r sits at position c ^ ((r >> 1) & 7) (applied on the SOURCE address and on the read), which makes every ds_read_b128
//     lane group hit 16 distinct 16-byte slots.
//   * 32x32x16 MFMAs.  S^T = K.Q^T with the key rows of a 32-key block PERMUTED over the MFMA's A rows
//     (row 8i + 4b + r  <->  key 16 (i >> 1) + 8 b + 4 (i & 1) + r), so that the 16 scores a lane ends up with are, per 16-key
//     chunk, 8 CONSECUTIVE keys of one query: after exp + bf16 packing they are that lane's B-operand fragment of
//     O^T = Vt.P^T as they stand (no LDS round trip, no shuffles), and the matching A operand is one 16-byte read of a Vt row.
//     A lane owns one query column of S^T and O^T: the row maximum needs the other half-wave only when the maximum grows.
//   * deferred maximum: the running maximum m of a row is only raised when a block's maximum exceeds it by more than 2^10 (in
//     the exp2 domain) -- P then stays <= 1024, harmless for bf16's 8-bit exponent and the fp32 accumulators -- so the
//     O / l rescale (32 multiplies per lane) sits in a branch that random and real score distributions take in the first
//     blocks only.  Exactness does not depend on it: m cancels in O / l.
//   * 1-D grid, XCD-aware: the q chunks of one (sequence, head) get consecutive logical ids on ONE XCD, so K / Vt (262 KB per
//     head at 1026 tokens) are fetched into one L2 instead of five.
#pragma once
#include "pcy_internal.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool MASKED>
__device__ __forceinline__ void fa_softmax_block(f32x16& s, float& m, float& l, f32x16 (&o)[2], bf16x8 (&pf)[2], int key0, int half, int len) {
  constexpr float LOG2E = 1.4426950408889634f;
  constexpr float THR = 10.0f;   // exp2 domain
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
  float bm = fmaxf(fmaxf(v[0], v[1]), v[2]);
#pragma unroll
  for (int e = 3; e < 15; e += 2) bm = fmaxf(fmaxf(bm, v[e]), v[e + 1]);
  bm = fmaxf(bm, v[15]);
  const float bm2 = bm * LOG2E;
  if (__builtin_expect(__any(bm2 > m + THR), 0)) {
    // raise the maximum of EVERY row of the group to its true running maximum (rows that did not grow get alpha = 1)
    const float other = __shfl_xor(bm2, 32, 64);
    const float mn = fmaxf(m, fmaxf(bm2, other));
    const float alpha = __builtin_amdgcn_exp2f(m - mn);   // m = -inf (first block): 0
    l *= alpha;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
    m = mn;
  }
  float p[16], sum = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    p[e] = __builtin_amdgcn_exp2f(fmaf(v[e], LOG2E, -m));
    sum += p[e];
  }
  l += sum;
  // chunk c (16 keys): registers i in {2c, 2c+1}  ->  slots j = 4 (i & 1) + r  <->  keys 16 c + 8 half + j
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const uint32_t w[4] = {pack_bf(p[8 * c + 0], p[8 * c + 1]), pack_bf(p[8 * c + 2], p[8 * c + 3]),
                           pack_bf(p[8 * c + 4], p[8 * c + 5]), pack_bf(p[8 * c + 6], p[8 * c + 7])};
    pf[c] = __builtin_bit_cast(bf16x8, w);
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attn_fast64_kernel(PcyAttnArgs a, int nchunk) {
  constexpr int DH = 64, KT = 64, QW = 64, QB = 4 * QW;
  constexpr int TILE = KT * DH * 2;                     // 8 KiB: K tile, then Vt tile
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * TILE];
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
  const int vt0 = a.vt_cu[sq];
  const bf16_t* kglob = a.k + (size_t)t0 * a.ldk + a.kcol0 + h * DH;
  const bf16_t* vglob = a.vt + (size_t)h * DH * a.vt_total + vt0;

  // Q fragments (B operand of S^T): lane holds Q[query col][16 ks + 8 half .. +8]
  bf16x8 qf[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    int qrow = qr0 + g * 32 + col;
    qrow = qrow < len ? qrow : len - 1;
    const bf16_t* qp = a.q + (size_t)(t0 + qrow) * a.ldq + a.qcol0 + h * DH + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[g][ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  auto stage = [&](int kt, char* buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int slot = i * 256 + tid, row = slot >> 3, cp = slot & 7;
      const int c = cp ^ ((row >> 1) & 7);
      int key = kt * KT + row;
      key = key < len ? key : len - 1;
      __builtin_amdgcn_global_load_lds((gptr_t)(kglob + (size_t)key * a.ldk + c * 8), (lds_ptr_t)(buf + (i * 4 + wave) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(vglob + (size_t)row * a.vt_total + kt * KT + c * 8),
                                       (lds_ptr_t)(buf + TILE + (i * 4 + wave) * 1024), 16, 0, 0);
    }
  };
  // A-operand row (lane & 31) of S^T  ->  key of the 32-key block
  const int krow = 16 * (col >> 4) + 8 * ((col >> 2) & 1) + 4 * ((col >> 3) & 1) + (col & 3);
  auto kfrag = [&](const char* buf, int sub, int ks) __attribute__((always_inline)) {
    const int row = krow + 32 * sub;
    return *reinterpret_cast<const bf16x8*>(buf + row * 128 + (((2 * ks + half) ^ ((row >> 1) & 7)) << 4));
  };
  auto vfrag = [&](const char* buf, int t, int sub, int c) __attribute__((always_inline)) {
    const int row = 32 * t + col;
    return *reinterpret_cast<const bf16x8*>(buf + TILE + row * 128 + (((4 * sub + 2 * c + half) ^ ((row >> 1) & 7)) << 4));
  };

  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
  f32x16 o[2][2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[g][t][e] = 0.f;

  const int ntiles = (len + KT - 1) / KT;
  stage(0, smem);
  __syncthreads();
  for (int kt = 0; kt < ntiles; ++kt) {
    const char* buf = smem + (kt & 1) * 2 * TILE;
    if (kt + 1 < ntiles) stage(kt + 1, smem + ((kt + 1) & 1) * 2 * TILE);
    if (active) {
      const bool tail = (kt + 1) * KT > len;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        bf16x8 kf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = kfrag(buf, sub, ks);
        f32x16 s[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
          for (int e = 0; e < 16; ++e) s[g][e] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) s[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[g][ks], s[g], 0, 0, 0);
        }
        bf16x8 pf[2][2];
        const int key0 = kt * KT + 32 * sub;
        if (tail) {
#pragma unroll
          for (int g = 0; g < 2; ++g) fa_softmax_block<true>(s[g], m[g], l[g], o[g], pf[g], key0, half, len);
        } else {
#pragma unroll
          for (int g = 0; g < 2; ++g) fa_softmax_block<false>(s[g], m[g], l[g], o[g], pf[g], key0, half, len);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const bf16x8 vf = vfrag(buf, t, sub, c);
#pragma unroll
            for (int g = 0; g < 2; ++g) o[g][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[g][c], o[g][t], 0, 0, 0);
          }
      }
    }
    __syncthreads();   // the next tile has landed (vmcnt drained) and every wave has left this one
  }
  if (!active) return;
  // O^T: lane (query col, half) holds dh = 32 t + 8 i + 4 half + r  ->  four consecutive features per (t, i)
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int qq = qr0 + g * 32 + col;
    const float lt = l[g] + __shfl_xor(l[g], 32, 64);
    const float inv = 1.0f / lt;
    if (qq >= len) continue;
    bf16_t* op = a.o + (size_t)(t0 + qq) * a.ldo + h * DH + 4 * half;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t w0 = pack_bf(o[g][t][4 * i + 0] * inv, o[g][t][4 * i + 1] * inv);
        const uint32_t w1 = pack_bf(o[g][t][4 * i + 2] * inv, o[g][t][4 * i + 3] * inv);
        *reinterpret_cast<uint2*>(op + 32 * t + 8 * i) = make_uint2(w0, w1);
      }
  }
}

// true when the launch was taken: head_dim 64, bidirectional, unmasked, unit scale, every sequence's Vt slice padded to a multiple
// of 64 keys (so that a whole 64-key tile of Vt can be fetched; the pad columns are zero)
inline bool pcy_launch_attn_fast64(hipStream_t s, const PcyAttnArgs& a, bool vt_pad64) {
  if (a.dh != 64 || a.causal || a.keep || a.scale != 1.0f || a.H != a.Hkv || !vt_pad64) return false;
  if ((a.ldq | a.ldk | a.qcol0 | a.kcol0 | a.vt_total) % 8 || a.ldo % 4) return false;
  const int nchunk = (a.max_len + 255) / 256;
  hipLaunchKernelGGL(attn_fast64_kernel, dim3(nchunk * a.H * a.nseq), dim3(256), 0, s, a, nchunk);
  return true;
}

}  // namespace
