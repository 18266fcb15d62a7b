// Memory-bound glue kernels of the ProCyon path: norms, embeddings / soft-token splice, rotary,
// KV scatter, V transpose, protein pooler, greedy pick.  All 16-byte vectorised along the feature
// dimension, fp32 math, bf16 rounding exactly where the reference's torch ops round.
#include "pcy_internal.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------ RMSNorm (HF LlamaRMSNorm)
__global__ __launch_bounds__(NT) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                     bf16_t* __restrict__ y, int d, float eps, int cast) {
  __shared__ float red[NT / 64];
  const bf16_t* xr = x + (size_t)blockIdx.x * d;
  bf16_t* yr = y + (size_t)blockIdx.x * d;
  float ss = 0.f;
  for (int k = threadIdx.x * 8; k < d; k += NT * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { float a = lo_bf(u[j]), b = hi_bf(u[j]); ss += a * a + b * b; }
  }
  ss = block_sum<NT>(ss, red);
  const float rstd = rsqrtf(ss / (float)d + eps);
  for (int k = threadIdx.x * 8; k < d; k += NT * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
    const uint4 g = *reinterpret_cast<const uint4*>(w + k);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = lo_bf(u[j]) * rstd, b = hi_bf(u[j]) * rstd;
      if (cast == 0) { a = rbf(a); b = rbf(b); }
      o[j] = pack_bf(lo_bf(gg[j]) * a, hi_bf(gg[j]) * b);
    }
    *reinterpret_cast<uint4*>(yr + k) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------ LayerNorm (nn.LayerNorm, fp32 math, one rounding)
__global__ __launch_bounds__(NT) void layernorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                       const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int d,
                                                       float eps) {
  __shared__ float red[NT / 64];
  const bf16_t* xr = x + (size_t)blockIdx.x * d;
  bf16_t* yr = y + (size_t)blockIdx.x * d;
  float s = 0.f;
  for (int k = threadIdx.x * 8; k < d; k += NT * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) s += lo_bf(u[j]) + hi_bf(u[j]);
  }
  const float mean = block_sum<NT>(s, red) / (float)d;
  float q = 0.f;
  for (int k = threadIdx.x * 8; k < d; k += NT * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { float a = lo_bf(u[j]) - mean, c = hi_bf(u[j]) - mean; q += a * a + c * c; }
  }
  const float rstd = rsqrtf(block_sum<NT>(q, red) / (float)d + eps);
  for (int k = threadIdx.x * 8; k < d; k += NT * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
    const uint4 g = *reinterpret_cast<const uint4*>(w + k);
    const uint4 h = *reinterpret_cast<const uint4*>(b + k);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w}, hh[4] = {h.x, h.y, h.z, h.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = pack_bf((lo_bf(u[j]) - mean) * rstd * lo_bf(gg[j]) + lo_bf(hh[j]),
                     (hi_bf(u[j]) - mean) * rstd * hi_bf(gg[j]) + hi_bf(hh[j]));
    *reinterpret_cast<uint4*>(yr + k) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// one WAVE per row, the row held in registers (VPL x 16 B per lane): one global read, no workgroup barrier
template <int VPL>
__global__ __launch_bounds__(NT) void layernorm_wave_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int rows,
                                                            int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * d;
  uint4 v[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int k = (i * 64 + lane) * 8;
    v[i] = k < d ? *reinterpret_cast<const uint4*>(xr + k) : make_uint4(0, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) s += lo_bf(u[j]) + hi_bf(u[j]);
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    if ((i * 64 + lane) * 8 < d) {
      const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float a = lo_bf(u[j]) - mean, c = hi_bf(u[j]) - mean; q += a * a + c * c; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  bf16_t* yr = y + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int k = (i * 64 + lane) * 8;
    if (k < d) {
      const uint4 g = *reinterpret_cast<const uint4*>(w + k);
      const uint4 h = *reinterpret_cast<const uint4*>(b + k);
      const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w}, gg[4] = {g.x, g.y, g.z, g.w}, hh[4] = {h.x, h.y, h.z, h.w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack_bf((lo_bf(u[j]) - mean) * rstd * lo_bf(gg[j]) + lo_bf(hh[j]),
                       (hi_bf(u[j]) - mean) * rstd * hi_bf(gg[j]) + hi_bf(hh[j]));
      *reinterpret_cast<uint4*>(yr + k) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
}

// ------------------------------------------------------------------ embedding lookup + soft-token splice (A5)
__global__ __launch_bounds__(NT) void embed_gather_kernel(const bf16_t* __restrict__ table, const int32_t* __restrict__ ids,
                                                          const bf16_t* __restrict__ soft, const int32_t* __restrict__ soft_map,
                                                          bf16_t* __restrict__ out, int d, unsigned* __restrict__ epoch, unsigned* __restrict__ epoch2) {
  const int r = blockIdx.x;
  // first launch of a decode step: also advances the epoch words of the step's in-launch hand-overs (was a launch of its own)
  if (r == 0 && threadIdx.x == 0) {
    if (epoch) *epoch += 1;
    if (epoch2) *epoch2 += 1;
  }
  const int sm = soft_map ? soft_map[r] : -1;
  const bf16_t* src = sm >= 0 ? soft + (size_t)sm * d : table + (size_t)ids[r] * d;
  for (int k = threadIdx.x * 8; k < d; k += NT * 8)
    *reinterpret_cast<uint4*>(out + (size_t)r * d + k) = *reinterpret_cast<const uint4*>(src + k);
}

// ------------------------------------------------------------------ ESM embedding (HF EsmEmbeddings, token_dropout)
// grid (nseq, SL): block (q, sl) handles tokens sl, sl+SL, ... of sequence q
__global__ __launch_bounds__(NT) void esm_embed_kernel(const bf16_t* __restrict__ table, const int32_t* __restrict__ toks,
                                                       const int32_t* __restrict__ cu, bf16_t* __restrict__ out, int d,
                                                       int mask_pads) {
  __shared__ float red[NT / 64];
  const int q = blockIdx.x;
  const int t0 = cu[q], len = cu[q + 1] - t0;
  float nmask = 0.f, nkeep = 0.f;
  for (int j = threadIdx.x; j < len; j += NT) {
    const int t = toks[t0 + j];
    nmask += (t == 32) ? 1.f : 0.f;
    nkeep += (t != 1) ? 1.f : 0.f;
  }
  nmask = block_sum<NT>(nmask, red);
  nkeep = block_sum<NT>(nkeep, red);
  const float src_len = mask_pads ? nkeep : (float)len;
  const float denom = 1.0f - nmask / src_len;  // 1 - mask_ratio_observed (fp32)
  for (int j = blockIdx.y; j < len; j += gridDim.y) {
    const int t = toks[t0 + j];
    const bool zero = (t == 32) || (mask_pads && t == 1);
    const bf16_t* src = table + (size_t)t * d;
    bf16_t* dst = out + (size_t)(t0 + j) * d;
    for (int k = threadIdx.x * 8; k < d; k += NT * 8) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (!zero) {
        const uint4 e = *reinterpret_cast<const uint4*>(src + k);
        const uint32_t u[4] = {e.x, e.y, e.z, e.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)  // bf16(x * 0.88) then fp32 / (1 - ratio) then bf16
          o[i] = pack_bf(rbf(lo_bf(u[i]) * 0.88f) / denom, rbf(hi_bf(u[i]) * 0.88f) / denom);
        v = make_uint4(o[0], o[1], o[2], o[3]);
      }
      *reinterpret_cast<uint4*>(dst + k) = v;
    }
  }
}

// ------------------------------------------------------------------ rotary on a token-major buffer
// one block per token; a thread handles 8 consecutive elements i..i+7 of the low half of a head and their partners
// i+dh/2.. (16-byte loads/stores)
// rope of token `tok` over heads [0, nh) of `row` (in place); heads >= kh0 (the key heads) are also copied, roped, to kdst_row
// (cache row of the token: kdst_row + (h - kh0) * kstride_h), nullptr = no copy
__device__ __forceinline__ void rope_row(bf16_t* row, int nh, int dh, int p, const bf16_t* __restrict__ cos_t, const bf16_t* __restrict__ sin_t,
                                         int mode, float prescale, int kh0, bf16_t* kdst_row, size_t kstride_h) {
  const int half = dh >> 1;
  const int cph = half >> 3;  // 8-element chunks per half head
  const bf16_t* c = cos_t + (size_t)p * dh;
  const bf16_t* s = sin_t + (size_t)p * dh;
  for (int e = threadIdx.x; e < nh * cph; e += NT) {
    const int h = e / cph, i = (e - h * cph) * 8;
    bf16_t* x = row + h * dh;
    const uint4 a1 = *reinterpret_cast<const uint4*>(x + i), a2 = *reinterpret_cast<const uint4*>(x + i + half);
    const uint4 c1 = *reinterpret_cast<const uint4*>(c + i), c2 = *reinterpret_cast<const uint4*>(c + i + half);
    const uint4 s1 = *reinterpret_cast<const uint4*>(s + i), s2 = *reinterpret_cast<const uint4*>(s + i + half);
    const uint32_t A1[4] = {a1.x, a1.y, a1.z, a1.w}, A2[4] = {a2.x, a2.y, a2.z, a2.w};
    const uint32_t C1[4] = {c1.x, c1.y, c1.z, c1.w}, C2[4] = {c2.x, c2.y, c2.z, c2.w};
    const uint32_t S1[4] = {s1.x, s1.y, s1.z, s1.w}, S2[4] = {s2.x, s2.y, s2.z, s2.w};
    uint32_t O1[4], O2[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float o1[2], o2[2];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float x1 = hh ? hi_bf(A1[w]) : lo_bf(A1[w]), x2 = hh ? hi_bf(A2[w]) : lo_bf(A2[w]);
        if (prescale != 0.f) { x1 = rbf(x1 * prescale); x2 = rbf(x2 * prescale); }
        const float cc1 = hh ? hi_bf(C1[w]) : lo_bf(C1[w]), cc2 = hh ? hi_bf(C2[w]) : lo_bf(C2[w]);
        const float ss1 = hh ? hi_bf(S1[w]) : lo_bf(S1[w]), ss2 = hh ? hi_bf(S2[w]) : lo_bf(S2[w]);
        if (mode == 0) {  // (x*cos) + (rotate_half(x)*sin), every op a bf16 tensor
          o1[hh] = rbf(rbf(x1 * cc1) + rbf(-x2 * ss1));
          o2[hh] = rbf(rbf(x2 * cc2) + rbf(x1 * ss2));
        } else {          // fp32, rounded once
          o1[hh] = x1 * cc1 + (-x2) * ss1;
          o2[hh] = x2 * cc2 + x1 * ss2;
        }
      }
      O1[w] = pack_bf(o1[0], o1[1]);
      O2[w] = pack_bf(o2[0], o2[1]);
    }
    const uint4 r1 = make_uint4(O1[0], O1[1], O1[2], O1[3]), r2 = make_uint4(O2[0], O2[1], O2[2], O2[3]);
    *reinterpret_cast<uint4*>(x + i) = r1;
    *reinterpret_cast<uint4*>(x + i + half) = r2;
    if (kdst_row && h >= kh0) {
      bf16_t* kd = kdst_row + (size_t)(h - kh0) * kstride_h;
      *reinterpret_cast<uint4*>(kd + i) = r1;
      *reinterpret_cast<uint4*>(kd + i + half) = r2;
    }
  }
}
__global__ __launch_bounds__(NT) void rope_kernel(bf16_t* __restrict__ buf, int ld, int col0, int nh, int dh,
                                                  const int32_t* __restrict__ pos, const bf16_t* __restrict__ cos_t,
                                                  const bf16_t* __restrict__ sin_t, int mode, float prescale) {
  const int tok = blockIdx.x;
  rope_row(buf + (size_t)tok * ld + col0, nh, dh, pos[tok], cos_t, sin_t, mode, prescale, nh, nullptr, 0);
}

// ------------------------------------------------------------------ K/V scatter into the [B,Hkv,Tmax,dh] cache
__global__ __launch_bounds__(NT) void kv_scatter_kernel(const bf16_t* __restrict__ qkv, int ld, int kcol0, int vcol0,
                                                        int Hkv, int dh, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                        int T, int Tmax) {
  const int tok = blockIdx.x;
  const int b = tok / T, t = tok - b * T;
  const int n8 = Hkv * dh / 8;
  for (int e = threadIdx.x; e < 2 * n8; e += NT) {
    const int isv = e >= n8;
    const int c = (isv ? e - n8 : e) * 8;
    const int h = c / dh, i = c - h * dh;
    const uint4 v = *reinterpret_cast<const uint4*>(qkv + (size_t)tok * ld + (isv ? vcol0 : kcol0) + c);
    bf16_t* dst = (isv ? vc : kc) + (((size_t)b * Hkv + h) * Tmax + t) * dh + i;
    *reinterpret_cast<uint4*>(dst) = v;
  }
}

// ------------------------------------------------------------------ V transpose: token-major -> Vt[nh][dh][vt_total]
// grid (tiles of 64 tokens, nh*dh/64, nseq); 64x64 tile through LDS
// vc != nullptr (rectangular batches of T tokens per sequence): the V rows are also copied into the [B,Hkv,Tmax,dh] cache
__device__ __forceinline__ void transpose_v_tile(const bf16_t* __restrict__ buf, int ld, int vcol0, int dh,
                                                 const int32_t* __restrict__ cu, const int32_t* __restrict__ vt_cu,
                                                 bf16_t* __restrict__ vt, int vt_total, int bx, int by, int q,
                                                 bf16_t* __restrict__ vc, int Hkv, int Tmax) {
  // 16-byte global loads (8 features of a token) and 16-byte global stores (8 tokens of a feature); the 64 x 64 tile
  // turns in LDS (row stride 72 elements = 144 B keeps the 16-byte rows aligned and spreads the column reads over banks)
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
  const int t0 = cu[q], len = cu[q + 1] - t0;
  const int j0 = bx * 64;
  if (j0 >= len) return;
  const int f0 = by * 64;  // feature (h*dh + e) tile
  const int padlen = vt_cu[q + 1] - vt_cu[q];
  const bool vec = ((ld | vcol0) % 8 == 0) && (vt_total % 8 == 0);
  if (!vec) {
    for (int e = threadIdx.x; e < 64 * 64; e += NT) {
      const int j = e >> 6, f = e & 63;
      tile[j][f] = (j0 + j < len) ? buf[(size_t)(t0 + j0 + j) * ld + vcol0 + f0 + f] : (bf16_t)0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += NT) {
      const int f = e >> 6, j = e & 63;
      if (j0 + j < padlen) vt[(size_t)(f0 + f) * vt_total + vt_cu[q] + j0 + j] = tile[j][f];
    }
    return;
  }
  for (int e = threadIdx.x; e < 64 * 8; e += NT) {
    const int j = e >> 3, fc = (e & 7) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (j0 + j < len) {
      v = *reinterpret_cast<const uint4*>(buf + (size_t)(t0 + j0 + j) * ld + vcol0 + f0 + fc);
      if (vc) { const int f = f0 + fc, h = f / dh; *reinterpret_cast<uint4*>(vc + (((size_t)q * Hkv + h) * Tmax + j0 + j) * dh + (f - h * dh)) = v; }
    }
    *reinterpret_cast<uint4*>(&tile[j][fc]) = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 8; e += NT) {
    const int f = e >> 3, jc = (e & 7) * 8;
    if (j0 + jc >= padlen) continue;   // padlen is a multiple of 32: whole 8-token groups
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (uint32_t)tile[jc + 2 * k][f] | ((uint32_t)tile[jc + 2 * k + 1][f] << 16);
    *reinterpret_cast<uint4*>(vt + (size_t)(f0 + f) * vt_total + vt_cu[q] + j0 + jc) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}
__global__ __launch_bounds__(NT) void transpose_v_kernel(const bf16_t* __restrict__ buf, int ld, int vcol0, int dh,
                                                         const int32_t* __restrict__ cu, const int32_t* __restrict__ vt_cu,
                                                         bf16_t* __restrict__ vt, int vt_total) {
  transpose_v_tile(buf, ld, vcol0, dh, cu, vt_cu, vt, vt_total, blockIdx.x, blockIdx.y, blockIdx.z, nullptr, 0, 0);
}

// Llama prefill, everything between the qkv projection and the attention in ONE launch (was rope, kv_scatter, transpose_v):
//   blocks [0, B*T):  rope of the token's q and k heads in place; the roped k heads also go to the K cache
//   the rest:         64 x 64 tiles of V -> Vt (the attention's B operand) and -> the V cache
// Rectangular batch: token b*T + t, cu[q] = q*T.  Same arithmetic and the same bytes as the three launches.
__global__ __launch_bounds__(NT) void prefill_post_qkv_kernel(bf16_t* __restrict__ qkv, int ld, int H, int Hkv, int dh,
                                                              const int32_t* __restrict__ pos, const bf16_t* __restrict__ cos_t,
                                                              const bf16_t* __restrict__ sin_t, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                              int B, int T, int Tmax, const int32_t* __restrict__ cu,
                                                              const int32_t* __restrict__ vt_cu, bf16_t* __restrict__ vt, int vt_total,
                                                              int tiles_x, int tiles_y) {
  const int ntok = B * T;
  if ((int)blockIdx.x < ntok) {
    const int tok = blockIdx.x, b = tok / T, t = tok - b * T;
    rope_row(qkv + (size_t)tok * ld, H + Hkv, dh, pos[tok], cos_t, sin_t, 0, 0.f, H, kc + ((size_t)b * Hkv * Tmax + t) * dh, (size_t)Tmax * dh);
    return;
  }
  const int r = blockIdx.x - ntok;
  const int bx = r % tiles_x, by = (r / tiles_x) % tiles_y, q = r / (tiles_x * tiles_y);
  transpose_v_tile(qkv, ld, (H + Hkv) * dh, dh, cu, vt_cu, vt, vt_total, bx, by, q, vc, Hkv, Tmax);
}

// ------------------------------------------------------------------ ProteinPooler (A3)
// Two stages so that a 32-protein batch fills the chip: stage 1, grid (nprot, ceil(d/512), POOL_CH): a thread owns 8
// consecutive features (one 16-byte load per token) of the tokens t with t mod (4*POOL_CH) == 4*chunk + threadIdx.x/64;
// the four token partitions of a workgroup meet in LDS, the chunk results (fp32 sum / max + count) go to a workspace;
// stage 2 adds the POOL_CH chunks in order and applies torch.nanmean's two roundings.
constexpr int POOL_CH = 8;
__global__ __launch_bounds__(NT) void pool_partial_kernel(const bf16_t* __restrict__ h, int d, const int32_t* __restrict__ seg,
                                                          const int32_t* __restrict__ rng, int mode, float* __restrict__ wsum,
                                                          int* __restrict__ wcnt) {
  __shared__ float part[3][64][8];
  __shared__ int pcnt[3][64][8];
  const int p = blockIdx.x, ch = blockIdx.z;
  const int fg = threadIdx.x & 63, tp = threadIdx.x >> 6;
  const int c = blockIdx.y * 512 + fg * 8;
  const bool live = c < d;   // d % 8 == 0
  const int r0 = seg[p], r1 = seg[p + 1];
  constexpr int STEP = 4 * POOL_CH;
  const int mine = 4 * ch + tp;
  float acc[8];
  int cnt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { acc[e] = (mode == 2) ? -INFINITY : 0.f; cnt[e] = 0; }
  if (live) {
    int tok = 0;   // running token index over the concatenated ranges
    for (int r = r0; r < r1; ++r) {
      int st = rng[2 * r], ln = rng[2 * r + 1];
      if (mode == 1) {  // x[1:-1] of the concatenation: drop first token of first range, last of last
        if (r == r0) { st += 1; ln -= 1; }
        if (r == r1 - 1) ln -= 1;
      }
      for (int j = ((mine - tok) % STEP + STEP) % STEP; j < ln; j += STEP) {
        const uint4 u = *reinterpret_cast<const uint4*>(h + (size_t)(st + j) * d + c);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = (e & 1) ? hi_bf(w[e >> 1]) : lo_bf(w[e >> 1]);
          if (mode == 2) acc[e] = fmaxf(acc[e], v);
          else if (v == v) { acc[e] += v; ++cnt[e]; }  // nanmean skips NaNs
        }
      }
      tok += ln > 0 ? ln : 0;
    }
  }
  if (tp > 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[tp - 1][fg][e] = acc[e]; pcnt[tp - 1][fg][e] = cnt[e]; }
  }
  __syncthreads();
  if (tp == 0 && live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[e];
      int n = cnt[e];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (mode == 2) a = fmaxf(a, part[q][fg][e]);
        else { a += part[q][fg][e]; n += pcnt[q][fg][e]; }
      }
      wsum[((size_t)p * POOL_CH + ch) * d + c + e] = a;
      wcnt[((size_t)p * POOL_CH + ch) * d + c + e] = n;
    }
  }
}
__global__ __launch_bounds__(NT) void pool_finish_kernel(const float* __restrict__ wsum, const int* __restrict__ wcnt, int d, int mode,
                                                         bf16_t* __restrict__ out) {
  const int p = blockIdx.x, c = blockIdx.y * NT + threadIdx.x;
  if (c >= d) return;
  float a = wsum[(size_t)p * POOL_CH * d + c];
  int n = wcnt[(size_t)p * POOL_CH * d + c];
  for (int ch = 1; ch < POOL_CH; ++ch) {
    const float v = wsum[((size_t)p * POOL_CH + ch) * d + c];
    if (mode == 2) a = fmaxf(a, v);
    else { a += v; n += wcnt[((size_t)p * POOL_CH + ch) * d + c]; }
  }
  // torch.nanmean on bf16: nansum (fp32 accumulate, rounded to bf16) / count, rounded again
  out[(size_t)p * d + c] = (mode == 2) ? f2bf(a) : f2bf(rbf(a) / (float)n);
}

// ------------------------------------------------------------------ greedy pick (A8): argmax + log-prob
// two stages so the 128k-entry vocabulary is scanned by 64 workgroups instead of one:
//   stage 1: block c of row b scans a contiguous chunk -> (max, lowest argmax, sum exp(x - local max))
//   stage 2: one wave per row merges the 64 partials, picks the token, adds bf16 log_softmax(logits)[tok]
constexpr int PICK_NT = 256;
constexpr int PICK_NB = 64;
struct PickPartial { float mx; int idx; float se; int pad; };

__global__ __launch_bounds__(PICK_NT) void pick_stage1_kernel(const bf16_t* __restrict__ logits, int V, PickPartial* __restrict__ part) {
  __shared__ float redv[PICK_NT / 64];
  __shared__ int redi[PICK_NT / 64];
  __shared__ float red[PICK_NT / 64];
  const int b = blockIdx.y, c = blockIdx.x;
  const int chunk = (V + PICK_NB - 1) / PICK_NB;
  const int lo = c * chunk, hi = (lo + chunk) < V ? (lo + chunk) : V;
  const bf16_t* lg = logits + (size_t)b * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lo + threadIdx.x; i < hi; i += PICK_NT) {
    const float v = bf2f(lg[i]);
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { redv[w] = best; redi[w] = bi; }
  __syncthreads();
  best = redv[0]; bi = redi[0];
  for (int i = 1; i < PICK_NT / 64; ++i)
    if (redv[i] > best || (redv[i] == best && redi[i] < bi)) { best = redv[i]; bi = redi[i]; }
  float se = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += PICK_NT) se += expf(bf2f(lg[i]) - best);
  se = block_sum<PICK_NT>(se, red);
  if (threadIdx.x == 0) part[b * PICK_NB + c] = PickPartial{best, bi, (lo < hi) ? se : 0.f, 0};
}

// one workgroup for all rows (wave w takes rows w, w + 4, ...), so that the step / position counters can be advanced by the
// same launch once every row has read them (was a launch of its own)
__global__ __launch_bounds__(256) void pick_stage2_kernel(const bf16_t* __restrict__ logits, int V, const PickPartial* __restrict__ part,
                                                          int32_t* __restrict__ next_tok, int32_t* __restrict__ tokens_out, int max_steps,
                                                          float* __restrict__ logprob, int32_t* __restrict__ step_dev, int32_t* __restrict__ pos_dev,
                                                          int advance_pos, int B) {
  const int l = threadIdx.x & 63;
  const int step = *step_dev;
  for (int b = threadIdx.x >> 6; b < B; b += 4) {
    const PickPartial pp = part[b * PICK_NB + l];
    float best = pp.mx;
    int bi = pp.idx;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    float se = pp.se * expf(pp.mx - best);   // empty chunks: mx = -inf -> exp(-inf) = 0
    se = wave_sum(se);
    if (l == 0) {
      // log_softmax in model dtype: bf16( x - max - log(sum exp(x - max)) ), x[tok] == max
      const float lsm = rbf((bf2f(logits[(size_t)b * V + bi]) - best) - logf(se));
      logprob[b] += lsm;
      next_tok[b] = bi;
      tokens_out[(size_t)b * max_steps + step] = bi;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) { if (advance_pos) *pos_dev += 1; *step_dev = step + 1; }
}

// ------------------------------------------------------------------ sampling / nucleus step (A8, model_unified.py:896-906)
// One decode step's selection when the reference does not take the argmax:
//   probs = softmax(logits / temperature)                         [a bf16 tensor: fp32 softmax, rounded once]
//   nucleus: probs = softmax(logits) * mask, mask = tokens whose ASCENDING cumulative probability has reached 1 - p
//            (`_get_nucleus_mask`, :844-858: sort ascending, cumsum (fp32 accumulate, bf16 values), keep where >= 1 - p;
//            the kept probabilities are NOT renormalised)
//   token ~ multinomial(probs)   -> here: inverse CDF in vocabulary order, token = first i with cdf_i > u * total, for a
//            uniform variate u in [0,1) supplied by the caller (torch's own RNG stream cannot be reproduced; parity is
//            defined on the probability vector and on the token for an injected u)
//   logprob += log_softmax(logits)[token]   (of the UNSCALED logits, bf16: :893,906)
// The ascending sort of 128k probabilities is replaced by a histogram over their 65536 possible bf16 values: the cumulative
// sum at the end of each value's run is a prefix sum over the histogram, the threshold falls inside exactly one run, and
// inside that run equal probabilities are ordered by token index (a stable sort; torch.sort leaves ties open).
constexpr int SMP_NT = 1024;

// partial (max, sum exp) of the raw and of the temperature-scaled logits per chunk: grid (PICK_NB, B)
__global__ __launch_bounds__(PICK_NT) void sample_stage1_kernel(const bf16_t* __restrict__ logits, int V, float temperature,
                                                                float4* __restrict__ part) {
  __shared__ float red[PICK_NT / 64];
  const int b = blockIdx.y, c = blockIdx.x;
  const int chunk = (V + PICK_NB - 1) / PICK_NB;
  const int lo = c * chunk, hi = (lo + chunk) < V ? (lo + chunk) : V;
  const bf16_t* lg = logits + (size_t)b * V;
  const bool scaled = temperature != 1.0f;
  float mr = -INFINITY, ms = -INFINITY;
  for (int i = lo + threadIdx.x; i < hi; i += PICK_NT) {
    const float v = bf2f(lg[i]);
    mr = fmaxf(mr, v);
    ms = fmaxf(ms, scaled ? rbf(v / temperature) : v);
  }
  mr = block_max<PICK_NT>(mr, red);
  ms = block_max<PICK_NT>(ms, red);
  float sr = 0.f, ss = 0.f;
  for (int i = lo + threadIdx.x; i < hi; i += PICK_NT) {
    const float v = bf2f(lg[i]);
    sr += expf(v - mr);
    ss += expf((scaled ? rbf(v / temperature) : v) - ms);
  }
  sr = block_sum<PICK_NT>(sr, red);
  ss = block_sum<PICK_NT>(ss, red);
  if (threadIdx.x == 0) part[b * PICK_NB + c] = make_float4(mr, lo < hi ? sr : 0.f, ms, lo < hi ? ss : 0.f);
}

// block-wide exclusive prefix sums of one value per thread (SMP_NT threads, thread order); *total = the sum
__device__ __forceinline__ float block_excl_scan(float v, float* lds /* [SMP_NT / 64] */, float* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
  __syncthreads();
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  float base = 0.f, tot = 0.f;
  for (int i = 0; i < SMP_NT / 64; ++i) { const float x = lds[i]; if (i < w) base += x; tot += x; }
  *total = tot;
  return base + inc - v;
}
__device__ __forceinline__ int block_excl_scan_i(int v, int* lds) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
  __syncthreads();
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += lds[i];
  return base + inc - v;
}

// bf16( softmax_fp32( logits / temperature ) ) of every token as bits ([B][V] uint16; >= +0, so the bits order like the values)
// and, for the nucleus mask, the histogram over those bits: grid (PICK_NB, B) -- the 128k exponentials and divisions of a row
// on 64 CUs instead of the one that runs the selection.
__global__ __launch_bounds__(PICK_NT) void sample_probs_kernel(const bf16_t* __restrict__ logits, int V, const float4* __restrict__ part,
                                                               float temperature, int nucleus, unsigned* __restrict__ hist,
                                                               bf16_t* __restrict__ pbits) {
  __shared__ float s_ml[2];
  const int b = blockIdx.y, c = blockIdx.x, lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {   // merge the chunk partials (as sample_stage2_kernel does)
    const float4 pp = part[b * PICK_NB + lane];
    const float ms = wave_max(pp.z);
    const float ss = wave_sum(pp.w * expf(pp.z - ms));
    if (lane == 0) { s_ml[0] = ms; s_ml[1] = ss; }
  }
  __syncthreads();
  const float m_s = s_ml[0], l_s = s_ml[1];
  const int chunk = (V + PICK_NB - 1) / PICK_NB;
  const int lo = c * chunk, hi = (lo + chunk) < V ? (lo + chunk) : V;
  const bf16_t* lg = logits + (size_t)b * V;
  const bool scaled = temperature != 1.0f;
  unsigned* hrow = hist + (size_t)b * 65536;
  for (int i = lo + threadIdx.x; i < hi; i += PICK_NT) {
    const float v = bf2f(lg[i]);
    const bf16_t pb = f2bf(expf((scaled ? rbf(v / temperature) : v) - m_s) / l_s);
    pbits[(size_t)b * V + i] = pb;
    if (nucleus) atomicAdd(&hrow[pb], 1u);
  }
}

// wave-wide sum with four DPP row rotations + two cross-row exchanges (every lane gets the total)
template <int CTRL>
__device__ __forceinline__ float smp_dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(r);
}
__device__ __forceinline__ float smp_wave_sum(float v) {
  v = smp_dpp_add<0x128>(v); v = smp_dpp_add<0x124>(v); v = smp_dpp_add<0x122>(v); v = smp_dpp_add<0x121>(v);   // row_ror 8, 4, 2, 1
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// wave-wide inclusive prefix sum in lane order
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
  return v;
}

// one workgroup per row.  hist: [B][65536] uint32 scratch, all zero on entry and on exit (nucleus only); probs_out optional [B,V].
// Wave w owns the K * 64 consecutive vocabulary indices from w * K * 64 (K = ceil(V / 1024)); lane l handles index
// base + k * 64 + l, k < K: every load is a coalesced 128-byte row, the probability bits of a lane's <= 128 elements stay
// packed in 64 registers for all passes, and index order = (wave, k, lane) order, so ranks and the CDF are wave scans plus
// one exchange of 16 per-wave totals.  (First version: a contiguous run of V / 1024 indices per THREAD -- 2-byte loads 252
// bytes apart, three passes each recomputing exp and division: 60 us per sampling step, 290 with the nucleus mask.)
constexpr int SMP_KMAX = 128;   // V <= SMP_NT * SMP_KMAX
__global__ __launch_bounds__(SMP_NT) void sample_stage2_kernel(const bf16_t* __restrict__ logits, int V, const float4* __restrict__ part,
                                                               float temperature, float nucleus_p, const float* __restrict__ uniforms,
                                                               unsigned* __restrict__ hist, bf16_t* __restrict__ probs_out,
                                                               int32_t* __restrict__ next_tok, int32_t* __restrict__ tokens_out, int max_steps,
                                                               float* __restrict__ logprob, const int32_t* __restrict__ step_dev, int B,
                                                               const bf16_t* __restrict__ pbits) {
  constexpr int NWV = SMP_NT / 64;
  __shared__ float fl[NWV];
  __shared__ int il[NWV];
  __shared__ float s_stat[4];
  __shared__ int s_first, s_kstar, s_jstar, s_tok, s_last;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* lg = logits + (size_t)b * V;
  const int step = *step_dev;
  if (tid < 64) {   // merge the chunk partials
    const float4 pp = part[b * PICK_NB + lane];
    const float mr = wave_max(pp.x), ms = wave_max(pp.z);
    const float sr = wave_sum(pp.y * expf(pp.x - mr)), ss = wave_sum(pp.w * expf(pp.z - ms));
    if (lane == 0) { s_stat[0] = mr; s_stat[1] = sr; s_stat[2] = ms; s_stat[3] = ss; }
  }
  if (tid == 0) { s_first = 0x7fffffff; s_kstar = -1; s_jstar = 0; s_tok = -1; s_last = -1; }
  __syncthreads();
  const float m_raw = s_stat[0], l_raw = s_stat[1];
  const int K = (V + SMP_NT - 1) / SMP_NT;
  const int wbase = wave * K * 64;
  // the probability bits of this lane's elements (sample_probs_kernel)
  const bf16_t* pbrow = pbits + (size_t)b * V;
  unsigned pk[SMP_KMAX / 2];
#pragma unroll
  for (int kk = 0; kk < SMP_KMAX / 2; ++kk) {
    unsigned w = 0;
    if (2 * kk < K) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = wbase + (2 * kk + e) * 64 + lane;
        if (2 * kk + e < K && i < V) w |= (unsigned)pbrow[i] << (16 * e);
      }
    }
    pk[kk] = w;
  }
#define SMP_PB(k) ((int)((pk[(k) >> 1] >> (((k) & 1) * 16)) & 0xffffu))
  const bool nucleus = nucleus_p > 0.f;
  unsigned* hrow = hist + (size_t)b * 65536;
  if (nucleus) {   // (the histogram was filled by sample_probs_kernel)
    // ascending cumulative sum over the 65536 values: wave w owns the values [4096 w, 4096 w + 4096), lane l value 4096 w + 64 k + l
    const int vbase = wave * 4096;
    float wsum = 0.f;
    for (int k = 0; k < 64; ++k) {
      const int val = vbase + k * 64 + lane;
      const unsigned c = __hip_atomic_load(&hrow[val], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wsum += (float)c * bf2f((bf16_t)val);
    }
    wsum = wave_sum(wsum);
    __syncthreads();
    if (lane == 0) fl[wave] = wsum;
    __syncthreads();
    float run = 0.f;
    for (int i = 0; i < wave; ++i) run += fl[i];
    const float thr = 1.0f - nucleus_p;
    // the first element (ascending order) whose cumulative sum, as a bf16 value, has reached the threshold
    int my_k = -1, my_j = 0;
    for (int k = 0; k < 64 && my_k < 0; ++k) {
      const int val = vbase + k * 64 + lane;
      const unsigned c = __hip_atomic_load(&hrow[val], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float v = bf2f((bf16_t)val);
      const float mine = (float)c * v;
      const float csum = smp_wave_sum(mine);
      if (rbf(run + csum) >= thr || run + csum >= thr) {   // (wave-uniform) the threshold may fall inside these 64 values: scan them
        const float incl = wave_incl_scan(mine, lane);
        const float before = run + (incl - mine);
        const bool hit = c != 0u && rbf(before + mine) >= thr;
        const unsigned long long hm = __ballot(hit);
        if (hm) {   // the lowest lane that reached it owns the threshold run
          const int fl_lane = __ffsll((long long)hm) - 1;
          if (lane == fl_lane) {
            int j = 1;
            while (j < (int)c && rbf(before + (float)j * v) < thr) ++j;
            my_k = val; my_j = j;
          }
          my_k = __shfl(my_k, fl_lane, 64); my_j = __shfl(my_j, fl_lane, 64);
        }
      }
      run += csum;
    }
    if (my_k >= 0 && lane == 0) atomicMin(&s_first, wave);
    __syncthreads();
    if (wave == s_first && lane == 0) { s_kstar = my_k; s_jstar = my_j; }
    // leave the histogram zeroed for the next call
    for (int k = 0; k < 64; ++k) hrow[vbase + k * 64 + lane] = 0u;
    __syncthreads();
  }
  const int kstar = nucleus ? s_kstar : -1, jstar = s_jstar;   // kstar < 0: keep everything (no nucleus, or nothing reached 1 - p)
  // index-order ranks inside the threshold run: ties of this wave, then the exclusive prefix over the waves
  int wties = 0;
  if (kstar >= 0) {
#pragma unroll
    for (int k = 0; k < SMP_KMAX; ++k)
      if (k < K) wties += __popcll(__ballot(SMP_PB(k) == kstar && wbase + k * 64 + lane < V));
    __syncthreads();
    if (lane == 0) il[wave] = wties;
    __syncthreads();
  }
  int rank0 = 1;   // rank (1-based, index order) of the first tie of this wave
  if (kstar >= 0) for (int i = 0; i < wave; ++i) rank0 += il[i];
  // the masked probabilities (pk is overwritten with them: 0 where masked) and the wave's sum
  float wsum = 0.f;
  int last_kept = -1;
  {
    int r = rank0;
#pragma unroll
    for (int kk = 0; kk < SMP_KMAX / 2; ++kk) {
      unsigned w = pk[kk], wn = 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = 2 * kk + e;
        if (k < K) {
          const int i = wbase + k * 64 + lane;
          const int pb = (int)((w >> (16 * e)) & 0xffffu);
          bool keep = i < V;
          if (kstar >= 0) {
            const bool tie = pb == kstar && i < V;
            const unsigned long long tm = __ballot(tie);
            const int myrank = r + __popcll(tm & ((1ull << lane) - 1ull));
            keep = keep && (pb > kstar || (tie && myrank >= jstar));
            r += __popcll(tm);
          }
          const int kept = keep ? pb : 0;
          wn |= (unsigned)kept << (16 * e);
          const float pv = bf2f((bf16_t)kept);
          if (probs_out && i < V) probs_out[(size_t)b * V + i] = (bf16_t)kept;
          if (pv > 0.f) last_kept = i;
          wsum += pv;
        }
      }
      pk[kk] = wn;
    }
  }
  wsum = wave_sum(wsum);
  last_kept = (int)wave_max((float)last_kept);   // indices < 2^24: exact in fp32
  __syncthreads();
  if (lane == 0) { fl[wave] = wsum; if (last_kept >= 0) atomicMax(&s_last, last_kept); }
  __syncthreads();
  float excl = 0.f, total = 0.f;
  for (int i = 0; i < NWV; ++i) { const float x = fl[i]; if (i < wave) excl += x; total += x; }
  const float target = uniforms[(size_t)step * B + b] * total;
  if (target >= excl && target < excl + wsum) {   // (wave-uniform) this wave's range holds the draw
    float cum = excl;
    int tok = -1;
#pragma unroll
    for (int k = 0; k < SMP_KMAX; ++k) {
      if (k < K && tok < 0) {
        const float pv = bf2f((bf16_t)SMP_PB(k));
        const float csum = smp_wave_sum(pv);
        if (cum + csum > target) {   // (wave-uniform) the draw may fall inside these 64 tokens: scan them
          const float incl = wave_incl_scan(pv, lane);
          const unsigned long long hm = __ballot(pv > 0.f && cum + incl > target);
          if (hm) tok = wbase + k * 64 + (__ffsll((long long)hm) - 1);
        }
        cum += csum;
      }
    }
    if (tok < 0) tok = last_kept;      // rounding at the end of the range
    if (lane == 0) atomicMax(&s_tok, tok);
  }
  __syncthreads();
  if (tid == 0) {
    int tok = s_tok >= 0 ? s_tok : s_last;   // u * total landed on the very end of the CDF
    if (tok < 0) tok = 0;
    const float lsm = rbf((bf2f(lg[tok]) - m_raw) - logf(l_raw));   // bf16 log_softmax of the unscaled logits
    logprob[b] += lsm;
    next_tok[b] = tok;
    tokens_out[(size_t)b * max_steps + step] = tok;
  }
#undef SMP_PB
}
// after every row has read *step_dev
__global__ void sample_advance_kernel(int32_t* pos_dev, int32_t* step_dev, int advance_pos) {
  if (threadIdx.x == 0) { if (advance_pos) *pos_dev += 1; *step_dev += 1; }
}

// ------------------------------------------------------------------ F.normalize(x, dim=-1) on a bf16 tensor
// torch: x / x.norm(2, -1, keepdim).clamp_min(eps): the norm is itself a bf16 tensor (fp32 accumulate, one rounding)
__global__ __launch_bounds__(NT) void l2norm_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int d, float eps) {
  __shared__ float red[NT / 64];
  const bf16_t* xr = x + (size_t)blockIdx.x * d;
  bf16_t* yr = y + (size_t)blockIdx.x * d;
  float ss = 0.f;
  for (int k = threadIdx.x * 8; k < d; k += NT * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float a = lo_bf(u[j]), b = hi_bf(u[j]); ss += a * a + b * b; }
  }
  ss = block_sum<NT>(ss, red);
  const float nrm = fmaxf(rbf(sqrtf(ss)), eps);
  for (int k = threadIdx.x * 8; k < d; k += NT * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf(lo_bf(u[j]) / nrm, hi_bf(u[j]) / nrm);
    *reinterpret_cast<uint4*>(yr + k) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------ retrieval ranking (row f3): top-k of a similarity row
// `get_proteins_from_embedding` (data/inference_utils.py:955-978) argsorts the whole similarity row to keep 20 entries.  Here
// every entry computes its own RANK -- the number of entries that sort in front of it: a larger score, or an equal score at a
// lower index (a stable descending order, deterministic where torch.argsort leaves ties open) -- and writes itself to slot
// `rank` if rank < k.  N^2 comparisons of 16-bit keys from LDS tiles: 18174 targets (the reference's cached matrix) = 3.3e8,
// 100k targets = 1e10 (< 1 ms of VALU), no sort, no atomics, k up to N (top_k=None: the full ranking).
// Order-preserving key of a bf16: flip all bits of negatives, the sign bit of positives; NaNs land at the extremes like in torch.
// (-0.0 and +0.0 compare equal in torch: both get the key of +0.0 and the index decides)
__device__ __forceinline__ uint32_t bf16_order_key(bf16_t b) { if (b == 0x8000u) b = 0; return (b & 0x8000u) ? (uint32_t)(b ^ 0xffffu) & 0xffffu : (uint32_t)(b | 0x8000u); }
constexpr int RANK_TILE = 4096;
__global__ __launch_bounds__(NT) void retrieval_rank_kernel(const bf16_t* __restrict__ sims, int N, int k, int32_t* __restrict__ idx_out,
                                                            bf16_t* __restrict__ score_out) {
  __shared__ uint16_t keys[RANK_TILE];
  const int q = blockIdx.y;
  const bf16_t* row = sims + (size_t)q * N;
  const int i0 = blockIdx.x * NT, i = i0 + threadIdx.x;
  const bf16_t mine = i < N ? row[i] : (bf16_t)0;
  const uint32_t ki = bf16_order_key(mine);
  int rank = 0;
  for (int t0 = 0; t0 < N; t0 += RANK_TILE) {
    const int tn = (N - t0) < RANK_TILE ? (N - t0) : RANK_TILE;
    __syncthreads();
    for (int j = threadIdx.x; j < tn; j += NT) keys[j] = (uint16_t)bf16_order_key(row[t0 + j]);
    __syncthreads();
    if (t0 + tn <= i0) {                       // every j of the tile lies before every i of this block: ties count
      for (int j = 0; j < tn; ++j) rank += keys[j] >= ki;
    } else if (t0 >= i0 + NT) {                // every j lies behind: ties do not count
      for (int j = 0; j < tn; ++j) rank += keys[j] > ki;
    } else {
      for (int j = 0; j < tn; ++j) rank += (keys[j] > ki) | ((keys[j] == ki) & (t0 + j < i));
    }
  }
  if (i < N && rank < k) {
    idx_out[(size_t)q * k + rank] = i;
    score_out[(size_t)q * k + rank] = mine;
  }
}

// ---- fp32 scoring for the shim's retrieval entry points (data/inference_utils.py:921-999: `get_proteins_from_batched_embeddings`
// works on `.float()` similarities, and a cached target matrix is fp32): cosine similarities with fp32 accumulation AND an
// fp32 result -- bf16 cosines keep ~3 digits, so among 18k-100k targets many scores would tie and be ordered by index.
//   sims[q][n] = (q / max(|q|, eps)) . (t_n / max(|t_n|, eps))
// One wave per target row (16-byte loads along the row), RET_QT queries per pass staged pre-normalised in LDS; the row's own
// squared norm rides along in the same pass.  HBM-bound: the target matrix is read once per RET_QT queries.
constexpr int RET_QT = 8;
template <bool TBF16>
__global__ __launch_bounds__(NT) void retrieval_dot_f32_kernel(const float* __restrict__ query, int Q, const void* __restrict__ targets, int N, int D,
                                                               float eps, float* __restrict__ sims) {
  extern __shared__ float qs[];                 // [RET_QT][D], normalised
  __shared__ float red[NT / 64];
  const int q0 = blockIdx.y * RET_QT;
  const int nq = (Q - q0) < RET_QT ? (Q - q0) : RET_QT;
  for (int qi = 0; qi < nq; ++qi) {             // stage + normalise (block-wide reduction per query)
    const float* qr = query + (size_t)(q0 + qi) * D;
    float ss = 0.f;
    for (int k = threadIdx.x; k < D; k += NT) { const float v = qr[k]; ss += v * v; }
    ss = block_sum<NT>(ss, red);
    const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
    for (int k = threadIdx.x; k < D; k += NT) qs[qi * D + k] = qr[k] * inv;
  }
  for (int qi = nq; qi < RET_QT; ++qi)
    for (int k = threadIdx.x; k < D; k += NT) qs[qi * D + k] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int n = blockIdx.x * (NT / 64) + wave; n < N; n += gridDim.x * (NT / 64)) {
    float acc[RET_QT], ss = 0.f;
#pragma unroll
    for (int qi = 0; qi < RET_QT; ++qi) acc[qi] = 0.f;
    for (int k = lane * 4; k < D; k += 64 * 4) {
      float t[4];
      if (TBF16) {
        const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(targets) + (size_t)n * D + k);
        t[0] = lo_bf(v.x); t[1] = hi_bf(v.x); t[2] = lo_bf(v.y); t[3] = hi_bf(v.y);
      } else {
        const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(targets) + (size_t)n * D + k);
        t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
      }
      ss += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
#pragma unroll
      for (int qi = 0; qi < RET_QT; ++qi) {
        const float4 qv = *reinterpret_cast<const float4*>(qs + qi * D + k);
        acc[qi] += t[0] * qv.x + t[1] * qv.y + t[2] * qv.z + t[3] * qv.w;
      }
    }
    ss = wave_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), eps);
#pragma unroll
    for (int qi = 0; qi < RET_QT; ++qi) {
      const float v = wave_sum(acc[qi]);
      if (lane == 0 && qi < nq) sims[(size_t)(q0 + qi) * N + n] = v * inv;
    }
  }
}
// order-preserving key of an fp32 (-0.0 == +0.0, NaN at the extremes like torch's sort)
__device__ __forceinline__ uint32_t f32_order_key(float f) {
  uint32_t b = __float_as_uint(f);
  if (b == 0x80000000u) b = 0;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ __launch_bounds__(NT) void retrieval_rank_f32_kernel(const float* __restrict__ sims, int N, int k, int32_t* __restrict__ idx_out,
                                                                float* __restrict__ score_out) {
  __shared__ uint32_t keys[RANK_TILE];
  const int q = blockIdx.y;
  const float* row = sims + (size_t)q * N;
  const int i0 = blockIdx.x * NT, i = i0 + threadIdx.x;
  const float mine = i < N ? row[i] : 0.f;
  const uint32_t ki = f32_order_key(mine);
  int rank = 0;
  for (int t0 = 0; t0 < N; t0 += RANK_TILE) {
    const int tn = (N - t0) < RANK_TILE ? (N - t0) : RANK_TILE;
    __syncthreads();
    for (int j = threadIdx.x; j < tn; j += NT) keys[j] = f32_order_key(row[t0 + j]);
    __syncthreads();
    if (t0 + tn <= i0) {
      for (int j = 0; j < tn; ++j) rank += keys[j] >= ki;
    } else if (t0 >= i0 + NT) {
      for (int j = 0; j < tn; ++j) rank += keys[j] > ki;
    } else {
      for (int j = 0; j < tn; ++j) rank += (keys[j] > ki) | ((keys[j] == ki) & (t0 + j < i));
    }
  }
  if (i < N && rank < k) {
    idx_out[(size_t)q * k + rank] = i;
    score_out[(size_t)q * k + rank] = mine;
  }
}

__global__ __launch_bounds__(NT) void copy_rows_kernel(const bf16_t* __restrict__ src, int lds_, bf16_t* __restrict__ dst,
                                                       int ldd, const int32_t* __restrict__ rows, int d) {
  const int r = blockIdx.x;
  const int sr = rows ? rows[r] : r;
  for (int k = threadIdx.x * 8; k < d; k += NT * 8)
    *reinterpret_cast<uint4*>(dst + (size_t)r * ldd + k) = *reinterpret_cast<const uint4*>(src + (size_t)sr * lds_ + k);
}

// fp32 running sum of selected rows (ret_token_access='all': torch.stack(hidden_states, -1).sum(-1) accumulates the
// L+1 bf16 states in fp32 and rounds once)
__global__ __launch_bounds__(NT) void acc_rows_kernel(const bf16_t* __restrict__ src, int lds_, const int32_t* __restrict__ rows,
                                                      float* __restrict__ acc, int d, int first) {
  const int r = blockIdx.x;
  const bf16_t* sp = src + (size_t)(rows ? rows[r] : r) * lds_;
  for (int k = threadIdx.x; k < d; k += NT) {
    const float v = bf2f(sp[k]);
    acc[(size_t)r * d + k] = first ? v : acc[(size_t)r * d + k] + v;
  }
}
__global__ __launch_bounds__(NT) void acc_finish_kernel(const float* __restrict__ acc, bf16_t* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) out[i] = f2bf(acc[i]);
}

// ------------------------------------------------------------------------------------------------
// fp8 path (BASELINE configs[4], no reference counterpart): per-row symmetric OCP e4m3 quantisation of a bf16 matrix.
// One workgroup per row, two passes over the (L2-resident) row: amax, then q = e4m3_rne(x / scale) with the smallest
// power-of-two scale >= 2^-126 that brings the row maximum to <= 448 (1.0 for an all-zero row): the division is exact, and
// a one-ulp difference in the row maximum cannot re-bucket the whole row.  oracle/fp8_ref.py::quant_rows restates exactly this.
__device__ __forceinline__ float clamp448(float v) { return fminf(fmaxf(v, -448.f), 448.f); }
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* __restrict__ x, int ldx, int K, unsigned char* __restrict__ q,
                                                             float* __restrict__ scale) {
  __shared__ float red[4];
  const size_t r = blockIdx.x;
  const bf16_t* xr = x + r * ldx;
  float amax = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(lo_bf(w[j])), fabsf(hi_bf(w[j]))));
  }
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sc = 1.f;
  if (amax > 0.f) {
    int e0 = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;     // amax = 1.m * 2^(e0+8); 448 = 1.75 * 2^8
    e0 = e0 < -126 ? -126 : e0;
    const float s0 = __uint_as_float((uint32_t)(e0 + 127) << 23);
    sc = amax <= 448.f * s0 ? s0 : 2.f * s0;
  }
  const float inv = 1.f / sc;   // exact: sc is a power of two
  if (threadIdx.x == 0) scale[r] = sc;
  unsigned char* qr = q + r * (size_t)K;
  for (int k = threadIdx.x * 8; k < K; k += 256 * 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[2 * j] = clamp448(lo_bf(w[j]) * inv); f[2 * j + 1] = clamp448(hi_bf(w[j]) * inv); }
    int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    *reinterpret_cast<uint2*>(qr + k) = make_uint2((uint32_t)lo, (uint32_t)hi);
  }
}

// ------------------------------------------------------------------------------------------------
// One step of the reference's diverse beam search bookkeeping (`_generate_beam_search`, model_unified.py:782-833) as ONE
// launch: grid = B prompts, one 1024-thread workgroup each, the beam groups of a prompt processed in order (group k's
// Hamming penalty depends on what groups < k chose at this step).  Per prompt:
//   row statistics   max and log-sum-exp of each of its `beam` logit rows (one wave per row)
//   candidates       s(r, v) = float(bf16((x[r][v] - max_r) - lse_r)) + cur[r] - penalty * #{earlier picks of this step == v}
//                    = the reference's  log_softmax(logits) (model dtype) + cur (fp32)  with the in-place bincount penalty
//   selection        top-g over the group's inc x V candidates (inc = 1 at step 0) by g rounds of a block-wide arg-max; ties
//                    go to the lowest flat index (torch.topk leaves the order of equal values unspecified)
//   commit           token histories (double-buffered rows: new row = parent's row + the token), running scores, parent
//                    slots (`src`, for the KV reorder and the logits record), next-step tokens, "row holds an EOS" flags
// The last workgroup to finish advances *step / *pos and raises *done when every row holds an EOS (model_unified.py:833);
// once *done is set a launch changes nothing, so the host may queue steps ahead and look at the flag only now and then.
struct BeamPick { float s; int c; };
__device__ __forceinline__ bool beam_better(float s, int c, float bs, int bc) { return s > bs || (s == bs && c < bc); }

// max and sum of exp(x - max) of one of BEAM_NCH slices of every logits row: grid (BB, BEAM_NCH).  (A single workgroup doing this
// for its `beam` rows with one dependent 2-byte load per iteration took 0.8 of the 1.27 ms of the first version of the step.)
constexpr int BEAM_NCH = 16;
__global__ __launch_bounds__(256) void beam_rowstats_kernel(const bf16_t* __restrict__ logits, int V, float2* __restrict__ part) {
  __shared__ float red[4];
  const int row = blockIdx.x, c = blockIdx.y;
  const int chunk = (V + BEAM_NCH - 1) / BEAM_NCH;
  const int lo = c * chunk, hi = (lo + chunk) < V ? (lo + chunk) : V;
  const bf16_t* x = logits + (size_t)row * V;
  float m = -INFINITY;
  for (int v0 = lo + threadIdx.x; v0 < hi; v0 += 256 * 8) {
    float xs[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int v = v0 + u * 256; xs[u] = v < hi ? bf2f(x[v]) : -INFINITY; }
#pragma unroll
    for (int u = 0; u < 8; ++u) m = fmaxf(m, xs[u]);
  }
  m = block_max<256>(m, red);
  float se = 0.f;
  if (lo < hi) {
    for (int v0 = lo + threadIdx.x; v0 < hi; v0 += 256 * 8) {
      float xs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int v = v0 + u * 256; xs[u] = v < hi ? bf2f(x[v]) : -INFINITY; }
#pragma unroll
      for (int u = 0; u < 8; ++u) se += expf(xs[u] - m);
    }
  }
  se = block_sum<256>(se, red);
  if (threadIdx.x == 0) part[(size_t)row * BEAM_NCH + c] = make_float2(m, lo < hi ? se : 0.f);
}

// Per (row, slice): the `n` best candidates of the slice in the order (log-probability descending, token ascending), as
// (bf16-rounded log-softmax value, token).  A group's final picks lie among the `beam` best unpenalised candidates of each of
// its rows (the Hamming penalty lowers at most beam - g tokens), so the sequential per-group logic only ever looks at these
// BB x BEAM_NCH x n entries instead of scanning V logits per round on ONE CU (0.84 ms per step, a CU pulls ~30 GB/s).
struct BeamCand { float s; int v; };
__global__ __launch_bounds__(256) void beam_topn_kernel(const bf16_t* __restrict__ logits, int V, const float2* __restrict__ part, int n,
                                                        BeamCand* __restrict__ cand) {
  __shared__ float red_s[4];
  __shared__ int red_v[4], red_t[4];
  __shared__ int win_t, win_v;
  __shared__ float win_s;
  const int row = blockIdx.x, c = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float2* pp = part + (size_t)row * BEAM_NCH;
  float m = -INFINITY;
  for (int q = 0; q < BEAM_NCH; ++q) m = fmaxf(m, pp[q].x);
  float se = 0.f;
  for (int q = 0; q < BEAM_NCH; ++q) se += pp[q].y * expf(pp[q].x - m);
  const float l = logf(se);
  const int chunk = (V + BEAM_NCH - 1) / BEAM_NCH;
  const int lo = c * chunk, hi = (lo + chunk) < V ? (lo + chunk) : V;
  const bf16_t* x = logits + (size_t)row * V;
  constexpr int EPT = 40;                      // elements per thread: slices of up to 10240 logits (V <= 163840)
  float xs[EPT];
#pragma unroll
  for (int u = 0; u < EPT; ++u) { const int v = lo + tid + u * 256; xs[u] = v < hi ? rbf((bf2f(x[v]) - m) - l) : -INFINITY; }
  unsigned long long taken = 0;
  for (int j = 0; j < n; ++j) {
    float bs = -INFINITY;
    int bu = -1;
#pragma unroll
    for (int u = 0; u < EPT; ++u)
      if (!((taken >> u) & 1) && lo + tid + u * 256 < hi && (bu < 0 || xs[u] > bs)) { bs = xs[u]; bu = u; }   // ascending u = ascending token
    int bv = bu >= 0 ? lo + tid + bu * 256 : 0x7fffffff, bt = tid;
    if (bu < 0) bs = -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(bs, o, 64);
      const int ov = __shfl_xor(bv, o, 64), ot = __shfl_xor(bt, o, 64);
      if (beam_better(os, ov, bs, bv)) { bs = os; bv = ov; bt = ot; }
    }
    if (lane == 0) { red_s[wave] = bs; red_v[wave] = bv; red_t[wave] = bt; }
    __syncthreads();
    if (tid == 0) {
      float fs = red_s[0];
      int fv = red_v[0], ft = red_t[0];
      for (int w = 1; w < 4; ++w)
        if (beam_better(red_s[w], red_v[w], fs, fv)) { fs = red_s[w]; fv = red_v[w]; ft = red_t[w]; }
      win_s = fs; win_v = fv; win_t = ft;
      cand[((size_t)row * BEAM_NCH + c) * n + j] = BeamCand{fs, fv};
    }
    __syncthreads();
    if (win_t == tid && win_v != 0x7fffffff) taken |= 1ull << ((win_v - lo - tid) / 256);
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void beam_step_kernel(const bf16_t* __restrict__ logits, int V, int beam, int g, float penalty,
                                                         PcyBeamState st, int B, const float2* __restrict__ st_part,
                                                         const BeamCand* __restrict__ cand, int ncand) {
  extern __shared__ __attribute__((aligned(16))) char bsm[];
  uint32_t* bitmap = reinterpret_cast<uint32_t*>(bsm);                    // [(V + 31) / 32] tokens picked earlier in this step
  const int nwords = (V + 31) / 32;
  __shared__ float rmax[32], rlse[32], red_s[16];
  __shared__ int red_c[16], sel_tok[32], nsel_s;
  __shared__ BeamPick pick_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, base = b * beam, BB = B * beam;
  if (*st.done) {
    if (tid < beam) st.src[base + tid] = base + tid;
    return;
  }
  const int i = *st.step;
  const int32_t* out_old = st.out + (size_t)(i & 1) * BB * st.max_len;
  int32_t* out_new = st.out + (size_t)((i + 1) & 1) * BB * st.max_len;
  const uint8_t* eos_old = st.has_eos + (size_t)(i & 1) * BB;
  uint8_t* eos_new = st.has_eos + (size_t)((i + 1) & 1) * BB;
  // ---- row statistics from the partials of beam_rowstats_kernel (BEAM_NCH chunks per row, combined in chunk order)
  if (tid < beam) {
    const float2* pp = st_part + (size_t)(base + tid) * BEAM_NCH;
    float m = -INFINITY;
    for (int c = 0; c < BEAM_NCH; ++c) m = fmaxf(m, pp[c].x);
    float se = 0.f;
    for (int c = 0; c < BEAM_NCH; ++c) se += pp[c].y * expf(pp[c].x - m);
    rmax[tid] = m; rlse[tid] = logf(se);
  }
  for (int w = tid; w < nwords; w += 1024) bitmap[w] = 0;
  if (tid == 0) nsel_s = 0;
  __syncthreads();
  const int groups = beam / g;
  for (int k = 0; k < groups; ++k) {
    const int gs = k * g, inc = (i == 0) ? 1 : g;
    float prev_s = INFINITY;
    int prev_c = -1;
    const int nsel = nsel_s;
    for (int j = 0; j < g; ++j) {
      float bs = -INFINITY;
      int bc = 0x7fffffff;
      // candidates: the BEAM_NCH x ncand best entries of each of the group's rows (beam_topn_kernel)
      const int per_row = BEAM_NCH * ncand;
      for (int e = tid; e < inc * per_row; e += 1024) {
        const int r = e / per_row;
        const BeamCand cd = cand[(size_t)(base + gs + r) * per_row + (e - r * per_row)];
        if (cd.v == 0x7fffffff) continue;
        float sc = cd.s + st.cur[base + gs + r];
        if (bitmap[cd.v >> 5] & (1u << (cd.v & 31))) {
          int cnt = 0;
          for (int q = 0; q < nsel; ++q) cnt += sel_tok[q] == cd.v;
          sc -= penalty * (float)cnt;
        }
        const int c = r * V + cd.v;
        const bool eligible = sc < prev_s || (sc == prev_s && c > prev_c);
        if (eligible && beam_better(sc, c, bs, bc)) { bs = sc; bc = c; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float os = __shfl_xor(bs, o, 64);
        const int oc = __shfl_xor(bc, o, 64);
        if (beam_better(os, oc, bs, bc)) { bs = os; bc = oc; }
      }
      if (lane == 0) { red_s[wave] = bs; red_c[wave] = bc; }
      __syncthreads();
      if (tid == 0) {
        float fs = red_s[0];
        int fc = red_c[0];
        for (int w = 1; w < 16; ++w)
          if (beam_better(red_s[w], red_c[w], fs, fc)) { fs = red_s[w]; fc = red_c[w]; }
        pick_s = BeamPick{fs, fc};
      }
      __syncthreads();
      const BeamPick pk = pick_s;
      prev_s = pk.s; prev_c = pk.c;
      // commit slot gs + j: parent row gs + pk.c / V (an OLD row: the old buffers are never written in this launch)
      const int parent = base + gs + pk.c / V, tok = pk.c % V, slot = base + gs + j;
      for (int q = tid; q < i; q += 1024) out_new[(size_t)slot * st.max_len + q] = out_old[(size_t)parent * st.max_len + q];
      if (tid == 0) {
        out_new[(size_t)slot * st.max_len + i] = tok;
        st.src[slot] = parent;
        st.next_tok[slot] = tok;
        st.cur_new[slot] = pk.s;
        eos_new[slot] = (uint8_t)((i > 0 ? eos_old[parent] : 0) | (tok == st.eos_id));
        if (st.anc) st.anc[(size_t)i * BB + slot] = parent;
      }
      __syncthreads();   // pick_s may be overwritten in the next round
    }
    // this group's picks join the penalty set of the later groups
    if (tid < g) {
      const int tok = out_new[(size_t)(base + gs + tid) * st.max_len + i];
      atomicOr(&bitmap[tok >> 5], 1u << (tok & 31));
      sel_tok[nsel + tid] = tok;
    }
    __syncthreads();
    if (tid == 0) nsel_s = nsel + g;
    __syncthreads();
  }
  // running scores: cur <- cur_new for this prompt's rows (a group reads only its OWN rows of cur, before it writes them)
  if (tid < beam) st.cur[base + tid] = st.cur_new[base + tid];
  __syncthreads();
  if (tid == 0) {
    // the reference tests (out == eos).any(dim=1) on rows that are ZERO-initialised up to max_len (model_unified.py:833):
    // with eos_id == 0 the padding itself satisfies it while the row is not full
    int all_eos = 1;
    for (int r = 0; r < beam; ++r) all_eos &= (eos_new[base + r] | (st.eos_id == 0 && i + 1 < st.max_len));
    st.blk_eos[b] = all_eos;
    __threadfence();
    const int ticket = atomicAdd(st.ticket, 1);
    if (ticket == B - 1) {       // last workgroup of the step
      __threadfence();
      int d = 1;
      for (int q = 0; q < B; ++q) d &= reinterpret_cast<volatile int32_t*>(st.blk_eos)[q];
      *st.ticket = 0;
      *st.step = i + 1;
      if (i > 0) *st.pos += 1;   // the step-0 logits come from the prefill: the cache length is still the prompt length
      if (d) *st.done = 1;
    }
  }
}

// RMSNorm + per-token e4m3 quantisation in one pass (fp8 prefill: the norm's bf16 output is only ever the quantiser's input).
// The normalised values are exactly those of rmsnorm_kernel (same statistic order, same rounding points) and the scale /
// codes exactly those of quant_rows_fp8_kernel on them, so the result is bit-identical to the two launches; d <= 8192.
__global__ __launch_bounds__(NT) void rmsnorm_quant_fp8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, int d, float eps,
                                                              int cast, unsigned char* __restrict__ q, float* __restrict__ scale) {
  __shared__ float red[NT / 64];
  __shared__ float redm[NT / 64];
  constexpr int MAXI = 4;
  const size_t r = blockIdx.x;
  const bf16_t* xr = x + r * d;
  uint4 xv[MAXI];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int k = threadIdx.x * 8 + it * NT * 8;
    if (k < d) {
      xv[it] = *reinterpret_cast<const uint4*>(xr + k);
      const uint32_t u[4] = {xv[it].x, xv[it].y, xv[it].z, xv[it].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { float a = lo_bf(u[j]), b = hi_bf(u[j]); ss += a * a + b * b; }
    }
  }
  ss = block_sum<NT>(ss, red);
  const float rstd = rsqrtf(ss / (float)d + eps);
  float nv[MAXI][8];
  float amax = 0.f;
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int k = threadIdx.x * 8 + it * NT * 8;
    if (k < d) {
      const uint4 g = *reinterpret_cast<const uint4*>(w + k);
      const uint32_t u[4] = {xv[it].x, xv[it].y, xv[it].z, xv[it].w}, gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = lo_bf(u[j]) * rstd, b = hi_bf(u[j]) * rstd;
        if (cast == 0) { a = rbf(a); b = rbf(b); }
        const uint32_t o = pack_bf(lo_bf(gg[j]) * a, hi_bf(gg[j]) * b);       // the bf16 tensor the reference materialises
        nv[it][2 * j] = lo_bf(o); nv[it][2 * j + 1] = hi_bf(o);
        amax = fmaxf(amax, fmaxf(fabsf(nv[it][2 * j]), fabsf(nv[it][2 * j + 1])));
      }
    }
  }
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) redm[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
  float sc = 1.f;
  if (amax > 0.f) {
    int e0 = (int)((__float_as_uint(amax) >> 23) & 0xff) - 127 - 8;
    e0 = e0 < -126 ? -126 : e0;
    const float s0 = __uint_as_float((uint32_t)(e0 + 127) << 23);
    sc = amax <= 448.f * s0 ? s0 : 2.f * s0;
  }
  const float inv = 1.f / sc;
  if (threadIdx.x == 0) scale[r] = sc;
  unsigned char* qr = q + r * (size_t)d;
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int k = threadIdx.x * 8 + it * NT * 8;
    if (k < d) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = clamp448(nv[it][e] * inv);
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
      int hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
      *reinterpret_cast<uint2*>(qr + k) = make_uint2((uint32_t)lo, (uint32_t)hi);
    }
  }
}

}  // namespace

bool pcy_launch_rmsnorm_quant_fp8(hipStream_t s, const bf16_t* x, const bf16_t* w, int rows, int d, float eps, int cast,
                                  unsigned char* q, float* scale) {
  if (d > 8192 || d % 8) return false;
  if (rows > 0) hipLaunchKernelGGL(rmsnorm_quant_fp8_kernel, dim3(rows), dim3(NT), 0, s, x, w, d, eps, cast, q, scale);
  return true;
}
void pcy_launch_quant_rows_fp8(hipStream_t s, const bf16_t* x, int ldx, int rows, int K, unsigned char* q, float* scale) {
  if (rows > 0) hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3(rows), dim3(256), 0, s, x, ldx, K, q, scale);
}

void pcy_launch_acc_rows(hipStream_t s, const bf16_t* src, int lds_, const int32_t* rows, float* acc, int nrows, int d, int first) {
  if (nrows > 0) hipLaunchKernelGGL(acc_rows_kernel, dim3(nrows), dim3(NT), 0, s, src, lds_, rows, acc, d, first);
}
void pcy_launch_acc_finish(hipStream_t s, const float* acc, bf16_t* out, size_t n) {
  if (n > 0) hipLaunchKernelGGL(acc_finish_kernel, dim3((unsigned)((n + NT - 1) / NT > 1024 ? 1024 : (n + NT - 1) / NT)), dim3(NT), 0, s, acc, out, n);
}

void pcy_launch_rmsnorm(hipStream_t s, const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int d, float eps, int cast) {
  if (rows > 0) hipLaunchKernelGGL(rmsnorm_kernel, dim3(rows), dim3(NT), 0, s, x, w, y, d, eps, cast);
}
void pcy_launch_layernorm(hipStream_t s, const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int rows, int d, float eps) {
  if (rows <= 0) return;
  const int vpl = (d + 511) / 512;
  const dim3 grid((rows + NT / 64 - 1) / (NT / 64));
  if (rows >= 64 && vpl <= 3) hipLaunchKernelGGL(layernorm_wave_kernel<3>, grid, dim3(NT), 0, s, x, w, b, y, rows, d, eps);
  else if (rows >= 64 && vpl <= 5) hipLaunchKernelGGL(layernorm_wave_kernel<5>, grid, dim3(NT), 0, s, x, w, b, y, rows, d, eps);
  else hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(NT), 0, s, x, w, b, y, d, eps);
}
void pcy_launch_embed_gather(hipStream_t s, const bf16_t* table, const int32_t* ids, const bf16_t* soft,
                             const int32_t* soft_map, bf16_t* out, int rows, int d) {
  if (rows > 0) hipLaunchKernelGGL(embed_gather_kernel, dim3(rows), dim3(NT), 0, s, table, ids, soft, soft_map, out, d, (unsigned*)nullptr, (unsigned*)nullptr);
}
// QA read-out (reference: data/inference_utils.py:582-604, training/train_utils.py:1048-1070): probabilities over the vocabulary at the answer
// row, `preds = logits.softmax(dim=-1)` on a tensor of the model dtype -- bf16: fp32 statistics, ONE rounding of exp(x - max) / sum (what
// torch.softmax does on a bf16 tensor); fp32: the same in fp32.  One workgroup per row; optional outputs: the whole vector, the yes / no
// columns (fp32 values of the stored probabilities), the argmax of the STORED probabilities (lowest index on ties, torch.argmax).
constexpr int QA_NT = 1024;
template <bool F32>
__global__ __launch_bounds__(QA_NT) void qa_probs_kernel(const void* __restrict__ logits, int V, int yes_id, int no_id, void* __restrict__ probs_out,
                                                         float* __restrict__ yes_no_out, int32_t* __restrict__ argmax_out) {
  __shared__ float red[QA_NT / 64];
  __shared__ int redi[QA_NT / 64];
  const int b = blockIdx.x;
  const bf16_t* lb = reinterpret_cast<const bf16_t*>(logits) + (size_t)b * V;
  const float* lf = reinterpret_cast<const float*>(logits) + (size_t)b * V;
  auto ld = [&](int i) { return F32 ? lf[i] : bf2f(lb[i]); };
  float m = -INFINITY;
  for (int i = threadIdx.x; i < V; i += QA_NT) m = fmaxf(m, ld(i));
  m = block_max<QA_NT>(m, red);
  float l = 0.f;
  for (int i = threadIdx.x; i < V; i += QA_NT) l += expf(ld(i) - m);
  l = block_sum<QA_NT>(l, red);
  float best = -1.f; int besti = 0x7fffffff;
  for (int i = threadIdx.x; i < V; i += QA_NT) {
    float p = expf(ld(i) - m) / l;
    if (!F32) p = rbf(p);
    if (probs_out) {
      if (F32) reinterpret_cast<float*>(probs_out)[(size_t)b * V + i] = p;
      else reinterpret_cast<bf16_t*>(probs_out)[(size_t)b * V + i] = f2bf(p);
    }
    if (yes_no_out && i == yes_id) yes_no_out[b * 2] = p;
    if (yes_no_out && i == no_id) yes_no_out[b * 2 + 1] = p;
    if (p > best) { best = p; besti = i; }       // (ascending i per thread: the first maximum)
  }
  // block arg-max, lowest index on ties
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = best; redi[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x == 0 && argmax_out) {
    for (int w = 1; w < QA_NT / 64; ++w)
      if (red[w] > best || (red[w] == best && redi[w] < besti)) { best = red[w]; besti = redi[w]; }
    argmax_out[b] = besti;
  }
}
void pcy_launch_qa_probs(hipStream_t s, const void* logits, int is_f32, int rows, int V, int yes_id, int no_id, void* probs_out, float* yes_no_out,
                         int32_t* argmax_out) {
  if (rows <= 0) return;
  if (is_f32) hipLaunchKernelGGL(qa_probs_kernel<true>, dim3(rows), dim3(QA_NT), 0, s, logits, V, yes_id, no_id, probs_out, yes_no_out, argmax_out);
  else hipLaunchKernelGGL(qa_probs_kernel<false>, dim3(rows), dim3(QA_NT), 0, s, logits, V, yes_id, no_id, probs_out, yes_no_out, argmax_out);
}
__global__ void bump_word_kernel(unsigned* w) { *w += 1; }
void pcy_launch_bump(hipStream_t s, unsigned* word) { hipLaunchKernelGGL(bump_word_kernel, dim3(1), dim3(1), 0, s, word); }

void pcy_launch_embed_tokens_dev(hipStream_t s, const bf16_t* table, const int32_t* ids, bf16_t* out, int rows, int d, unsigned* epoch,
                                 unsigned* epoch2) {
  if (rows > 0) hipLaunchKernelGGL(embed_gather_kernel, dim3(rows), dim3(NT), 0, s, table, ids, nullptr, nullptr, out, d, epoch, epoch2);
}
void pcy_launch_esm_embed(hipStream_t s, const bf16_t* table, const int32_t* toks, const int32_t* cu, int nseq,
                          int max_len, bf16_t* out, int d, int mask_pads) {
  if (nseq <= 0) return;
  int sl = max_len < 64 ? max_len : 64;
  if (sl < 1) sl = 1;
  hipLaunchKernelGGL(esm_embed_kernel, dim3(nseq, sl), dim3(NT), 0, s, table, toks, cu, out, d, mask_pads);
}
void pcy_launch_rope(hipStream_t s, bf16_t* buf, int ld, int col0, int nh, int dh, const int32_t* pos,
                     const bf16_t* cos_t, const bf16_t* sin_t, int ntok, int mode, float prescale) {
  if (ntok > 0) hipLaunchKernelGGL(rope_kernel, dim3(ntok), dim3(NT), 0, s, buf, ld, col0, nh, dh, pos, cos_t, sin_t, mode, prescale);
}
bool pcy_launch_prefill_post_qkv(hipStream_t s, bf16_t* qkv, int ld, int H, int Hkv, int dh, const int32_t* pos, const bf16_t* cos_t,
                                 const bf16_t* sin_t, bf16_t* kcache, bf16_t* vcache, int B, int T, int Tmax, const int32_t* cu,
                                 const int32_t* vt_cu, bf16_t* vt, int vt_total) {
  if (B * T <= 0 || (Hkv * dh) % 64 || dh % 16 || ld % 8 || ((H + Hkv) * dh) % 8 || vt_total % 8) return false;
  const int maxpad = (T + 31) / 32 * 32;
  const int tx = (maxpad + 63) / 64, ty = Hkv * dh / 64;
  hipLaunchKernelGGL(prefill_post_qkv_kernel, dim3(B * T + tx * ty * B), dim3(NT), 0, s, qkv, ld, H, Hkv, dh, pos, cos_t, sin_t, kcache, vcache,
                     B, T, Tmax, cu, vt_cu, vt, vt_total, tx, ty);
  return true;
}
void pcy_launch_kv_scatter(hipStream_t s, const bf16_t* qkv, int ld, int kcol0, int vcol0, int Hkv, int dh,
                           bf16_t* kcache, bf16_t* vcache, int B, int T, int Tmax) {
  if (B * T > 0) hipLaunchKernelGGL(kv_scatter_kernel, dim3(B * T), dim3(NT), 0, s, qkv, ld, kcol0, vcol0, Hkv, dh, kcache, vcache, T, Tmax);
}
__global__ void vt_offsets_kernel(const int32_t* __restrict__ cu, int nseq, int pad, int32_t* __restrict__ out) {
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int q = 0; q < nseq; ++q) { out[q] = acc; acc += (cu[q + 1] - cu[q] + pad - 1) / pad * pad; }
    out[nseq] = acc;
  }
}
void pcy_launch_vt_offsets(hipStream_t s, const int32_t* cu, int nseq, int pad, int32_t* vt_cu_out) {
  hipLaunchKernelGGL(vt_offsets_kernel, dim3(1), dim3(64), 0, s, cu, nseq, pad, vt_cu_out);
}
void pcy_launch_transpose_v(hipStream_t s, const bf16_t* buf, int ld, int vcol0, int nh, int dh, const int32_t* cu,
                            const int32_t* vt_cu, int nseq, int max_len, bf16_t* vt, int vt_total) {
  if (nseq <= 0) return;
  const int maxpad = (max_len + 63) / 64 * 64;   // (tiles past a sequence's padded length return at once)
  hipLaunchKernelGGL(transpose_v_kernel, dim3((maxpad + 63) / 64, nh * dh / 64, nseq), dim3(NT), 0, s, buf, ld, vcol0, dh, cu, vt_cu, vt, vt_total);
}
size_t pcy_pool_ws_bytes(int nprot, int d) { return (size_t)nprot * POOL_CH * d * 8; }
void pcy_launch_pool(hipStream_t s, const bf16_t* h, int d, const int32_t* seg, const int32_t* rng, int nprot, int mode, bf16_t* out,
                     void* ws) {
  if (nprot <= 0) return;
  float* wsum = reinterpret_cast<float*>(ws);
  int* wcnt = reinterpret_cast<int*>(wsum + (size_t)nprot * POOL_CH * d);
  hipLaunchKernelGGL(pool_partial_kernel, dim3(nprot, (d + 511) / 512, POOL_CH), dim3(NT), 0, s, h, d, seg, rng, mode, wsum, wcnt);
  hipLaunchKernelGGL(pool_finish_kernel, dim3(nprot, (d + NT - 1) / NT), dim3(NT), 0, s, wsum, wcnt, d, mode, out);
}
void pcy_launch_greedy_pick(hipStream_t s, const bf16_t* logits, int B, int V, int32_t* next_tok, int32_t* tokens_out,
                            int max_steps, float* logprob, int32_t* pos_dev, int32_t* step_dev, int advance_pos, void* partials) {
  hipLaunchKernelGGL(pick_stage1_kernel, dim3(PICK_NB, B), dim3(PICK_NT), 0, s, logits, V, reinterpret_cast<PickPartial*>(partials));
  hipLaunchKernelGGL(pick_stage2_kernel, dim3(1), dim3(256), 0, s, logits, V, reinterpret_cast<const PickPartial*>(partials), next_tok,
                     tokens_out, max_steps, logprob, step_dev, pos_dev, advance_pos, B);
}
int pcy_sample_max_vocab() { return SMP_NT * SMP_KMAX; }
void pcy_launch_sample_step(hipStream_t s, const bf16_t* logits, int B, int V, float temperature, float nucleus_p, const float* uniforms,
                            unsigned* hist, bf16_t* probs_out, int32_t* next_tok, int32_t* tokens_out, int max_steps, float* logprob,
                            int32_t* pos_dev, int32_t* step_dev, int advance_pos, void* partials, bf16_t* pbits) {
  hipLaunchKernelGGL(sample_stage1_kernel, dim3(PICK_NB, B), dim3(PICK_NT), 0, s, logits, V, temperature, reinterpret_cast<float4*>(partials));
  hipLaunchKernelGGL(sample_probs_kernel, dim3(PICK_NB, B), dim3(PICK_NT), 0, s, logits, V, reinterpret_cast<const float4*>(partials), temperature,
                     nucleus_p > 0.f ? 1 : 0, hist, pbits);
  hipLaunchKernelGGL(sample_stage2_kernel, dim3(B), dim3(SMP_NT), 0, s, logits, V, reinterpret_cast<const float4*>(partials), temperature,
                     nucleus_p, uniforms, hist, probs_out, next_tok, tokens_out, max_steps, logprob, step_dev, B, pbits);
  hipLaunchKernelGGL(sample_advance_kernel, dim3(1), dim3(64), 0, s, pos_dev, step_dev, advance_pos);
}
void pcy_launch_copy_rows(hipStream_t s, const bf16_t* src, int lds_, bf16_t* dst, int ldd, const int32_t* rows, int nrows, int d) {
  if (nrows > 0) hipLaunchKernelGGL(copy_rows_kernel, dim3(nrows), dim3(NT), 0, s, src, lds_, dst, ldd, rows, d);
}
void pcy_launch_l2norm_rows(hipStream_t s, const bf16_t* x, bf16_t* y, int rows, int d, float eps) {
  if (rows > 0) hipLaunchKernelGGL(l2norm_rows_kernel, dim3(rows), dim3(NT), 0, s, x, y, d, eps);
}
void pcy_launch_retrieval_rank(hipStream_t s, const bf16_t* sims, int Q, int N, int k, int32_t* idx_out, bf16_t* score_out) {
  if (Q > 0 && N > 0 && k > 0)
    hipLaunchKernelGGL(retrieval_rank_kernel, dim3((N + NT - 1) / NT, Q), dim3(NT), 0, s, sims, N, k, idx_out, score_out);
}
void pcy_launch_retrieval_dot_f32(hipStream_t s, const float* query, int Q, const void* targets, int targets_bf16, int N, int D, float eps, float* sims) {
  if (Q <= 0 || N <= 0) return;
  const size_t smem = (size_t)RET_QT * D * sizeof(float);
  const dim3 grid((unsigned)((N + NT / 64 - 1) / (NT / 64) < 2048 ? (N + NT / 64 - 1) / (NT / 64) : 2048), (Q + RET_QT - 1) / RET_QT);
  if (targets_bf16) {
    static bool conf = false;
    if (!conf && smem > 65536) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&retrieval_dot_f32_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); conf = true; }
    hipLaunchKernelGGL(retrieval_dot_f32_kernel<true>, grid, dim3(NT), smem, s, query, Q, targets, N, D, eps, sims);
  } else {
    static bool conf = false;
    if (!conf && smem > 65536) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&retrieval_dot_f32_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); conf = true; }
    hipLaunchKernelGGL(retrieval_dot_f32_kernel<false>, grid, dim3(NT), smem, s, query, Q, targets, N, D, eps, sims);
  }
}
size_t pcy_retrieval_dot_smem(int D) { return (size_t)RET_QT * D * sizeof(float); }
void pcy_launch_retrieval_rank_f32(hipStream_t s, const float* sims, int Q, int N, int k, int32_t* idx_out, float* score_out) {
  if (Q > 0 && N > 0 && k > 0)
    hipLaunchKernelGGL(retrieval_rank_f32_kernel, dim3((N + NT - 1) / NT, Q), dim3(NT), 0, s, sims, N, k, idx_out, score_out);
}
size_t pcy_beam_ws_bytes(int B, int beam) {
  return (size_t)B * beam * BEAM_NCH * sizeof(float2) + 256 + (size_t)B * beam * BEAM_NCH * beam * sizeof(BeamCand);
}
void pcy_launch_beam_step(hipStream_t s, const bf16_t* logits, int V, int B, int beam, int g, float penalty, const PcyBeamState& st,
                          void* ws) {
  const size_t smem = (size_t)((V + 31) / 32) * 4;
  float2* part = reinterpret_cast<float2*>(ws);
  BeamCand* cand = reinterpret_cast<BeamCand*>(reinterpret_cast<char*>(ws) + ((size_t)B * beam * BEAM_NCH * sizeof(float2) + 255) / 256 * 256);
  hipLaunchKernelGGL(beam_rowstats_kernel, dim3(B * beam, BEAM_NCH), dim3(256), 0, s, logits, V, part);
  hipLaunchKernelGGL(beam_topn_kernel, dim3(B * beam, BEAM_NCH), dim3(256), 0, s, logits, V, part, beam, cand);
  hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(1024), smem, s, logits, V, beam, g, penalty, st, B, part, cand, beam);
}
