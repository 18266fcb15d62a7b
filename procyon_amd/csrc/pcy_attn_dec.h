// Decode attention body shared by the stand-alone kernel (pcy_attn.hip) and the persistent decode kernel
// (pcy_decode.hip).  512 threads; `smem` = dynamic LDS of attn_dec_smem_bytes(G, DS, DH, Tmax) bytes.
#pragma once
#include <type_traits>
#include "pcy_common.h"
#include "pcy_internal.h"

#ifndef PCY_ATTN_DEC_NP
#define PCY_ATTN_DEC_NP(DS) ((DS) == 16 ? 4 : 2)
#endif
#ifndef PCY_ATTN_DEC_VPF
#define PCY_ATTN_DEC_VPF 0
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// Decode attention (row A7): one new token per row against the KV cache, exact softmax rounding, ONE pass per
// workgroup (kv head, sequence, DS output columns):
//   (A) rope the G query heads of the kv head and the new key once into LDS (three bf16 roundings, HF Llama); the
//       workgroup of slice 0 appends K and V to the cache; scores s = bf16(bf16(q.k)*scale) for all keys by MFMA
//       (A = the G roped heads, B = 16 cached key rows) into LDS;
//   (B) max / sum / p = bf16(exp(s-m)/l) in LDS (DPP wave reductions);
//   (C) P.V for the DS output columns over all keys (fp32 VALU, V rows requested at kernel start), reduced across
//       key groups by DPP + LDS, rounded once -> no cross-workgroup reduction.
struct Rope8In { uint4 a, b, c, s; };
// elements e0..e0+7 of one head and their rotate_half partners e +- DH/2, with the cos / sin rows of the position
template <int DH>
__device__ __forceinline__ Rope8In rope8_load(const bf16_t* __restrict__ x, const bf16_t* __restrict__ cs, const bf16_t* __restrict__ sn, int e0) {
  constexpr int HALF = DH / 2;
  const int p0 = e0 < HALF ? e0 + HALF : e0 - HALF;
  Rope8In r;
  r.a = *reinterpret_cast<const uint4*>(x + e0);
  r.b = *reinterpret_cast<const uint4*>(x + p0);
  r.c = *reinterpret_cast<const uint4*>(cs + e0);
  r.s = *reinterpret_cast<const uint4*>(sn + e0);
  return r;
}
// HF Llama apply_rotary_pos_emb on bf16 tensors: (x*cos) + (rotate_half(x)*sin), three roundings
template <int DH>
__device__ __forceinline__ void rope8_math(const Rope8In& in, int e0, float (&out)[8]) {
  const bool lo = e0 < DH / 2;
  const uint32_t aw[4] = {in.a.x, in.a.y, in.a.z, in.a.w}, bw[4] = {in.b.x, in.b.y, in.b.z, in.b.w};
  const uint32_t cw[4] = {in.c.x, in.c.y, in.c.z, in.c.w}, sw[4] = {in.s.x, in.s.y, in.s.z, in.s.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = lo_bf(aw[i]), x1 = hi_bf(aw[i]);
    float r0 = lo_bf(bw[i]), r1 = hi_bf(bw[i]);   // rotate_half partner: -x2 for the low half, +x1 for the high half
    if (lo) { r0 = -r0; r1 = -r1; }
    out[2 * i] = rbf(rbf(x0 * lo_bf(cw[i])) + rbf(r0 * lo_bf(sw[i])));
    out[2 * i + 1] = rbf(rbf(x1 * hi_bf(cw[i])) + rbf(r1 * hi_bf(sw[i])));
  }
}
template <int DH>
__device__ __forceinline__ void rope8(const bf16_t* __restrict__ x, const bf16_t* __restrict__ cs, const bf16_t* __restrict__ sn,
                                      int e0, float (&out)[8]) {
  rope8_math<DH>(rope8_load<DH>(x, cs, sn, e0), e0, out);
}

// Fused decode attention: grid (DH/16 column slices, Hkv, B), 512 threads.  Every block ropes q, scores ALL keys of
// its kv head (the K panel is small; the redundancy across the DH/16 slices buys a launch without any cross-workgroup
// dependency), normalises exactly, and accumulates P.V for its 16 output columns.
//   phase A  scores by MFMA: A operand = the G roped query heads (rows >= G are zero), B operand = 16 cached key rows
//            loaded straight from HBM in fragment layout (16 B per lane) -> no cross-lane reduction at all
//   phase B  exact softmax statistics (max, sum, p = bf16(exp(s-m)/l)) for all G heads in LDS
//   phase C  P.V on the VALU: lane pair (2 x 16 B) per V row slice, DPP row-rotate reduction over the key groups
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(r);
}
// sum over the 16 lanes of a DPP row; every lane of the row ends with the total
__device__ __forceinline__ float row16_sum(float v) {
  v = dpp_add<0x128>(v);  // row_ror:8
  v = dpp_add<0x124>(v);  // row_ror:4
  v = dpp_add<0x122>(v);  // row_ror:2
  v = dpp_add<0x121>(v);  // row_ror:1
  return v;
}

template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
  const int r = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false);
  return fmaxf(v, __int_as_float(r));
}
// wave-wide sum / max: four DPP row rotations + two cross-row exchanges (instead of six ds_bpermute round trips)
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = dpp_max<0x128>(v); v = dpp_max<0x124>(v); v = dpp_max<0x122>(v); v = dpp_max<0x121>(v);
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// DS = output columns per workgroup.  16 (DH/16 workgroups per (kv head, sequence), each recomputing the scores) fills
// the chip when B x Hkv is small; at beam / batch sizes where B x Hkv x DH/DS already covers the CUs the redundant score
// passes are the dominant cost and a wider slice (up to the whole head) is used instead.

struct AttnDecNoHook { __device__ __forceinline__ void operator()() const {} };

// `inputs_ready` runs after the cache rows of the first passes have been requested and before anything of the new
// token's q/k/v is read: the persistent decode kernel waits there for the projections of this layer.
template <int DH, int G, int DS, typename Hook = AttnDecNoHook>
__device__ __forceinline__ void attn_dec_body(const PcyDecAttnArgs& a, char* smem, const int bx, const int kvh, const int b,
                                              Hook inputs_ready = Hook()) {
  constexpr int NT = 512, NWV = NT / 64;
  constexpr int KB = DH / 32;
  constexpr int LPR = DS / 8;          // lanes that share one V row slice (16 bytes each)
  constexpr int RPW = 64 / LPR;        // V rows per wave per load
  const int scld = a.Tmax + 1;
  bf16_t* qk = reinterpret_cast<bf16_t*>(smem);                        // [(G+1)][DH] roped q heads, then roped k_new
  float* sc = reinterpret_cast<float*>(smem + (G + 1) * DH * 2);       // [G][scld] scores, then probabilities
  float* red = sc + (size_t)G * scld;           // [NWV][G][DS]
  float* wred = red + NWV * G * DS;             // [NWV][G]
  const int tid = pcy_tid(), lane = tid & 63, wave = tid >> 6;
  const int c0 = bx * DS;
  const int fr = lane & 15, fq = lane >> 4;
  const int t = a.t_plus1 ? a.t_plus1 - 1 : *a.pos_dev;
  const int nk = t + 1;
  if (a.dbg == 1) { if (tid == 0) a.o[(size_t)b * a.ldo + kvh * G * DH + c0] = (bf16_t)t; return; }
  const bf16_t* row = a.qkv + (size_t)b * a.ld;
  bf16_t* kc = a.kcache + ((size_t)b * a.Hkv + kvh) * a.Tmax * DH;
  bf16_t* vc = a.vcache + ((size_t)b * a.Hkv + kvh) * a.Tmax * DH;
  const bf16_t* cs = a.cos_t + (size_t)t * DH;
  const bf16_t* sn = a.sin_t + (size_t)t * DH;
  const uint8_t* keep = a.keep ? a.keep + (size_t)b * a.ld_keep : nullptr;

  // V row slices of the first P.V pass are requested now: they depend on nothing but t, and their HBM latency is
  // hidden behind the score and softmax phases
  // DS == 16: lanes 0-31 / 32-63 of a wave = the two 16-byte halves of 32 rows; wider slices: LPR consecutive lanes
  // cover one row slice (coalesced), RPW rows per wave
  // Vector loads return in order: the (L2-resident) q / k_new / cos / sin operands of the rope are requested BEFORE the
  // V rows and key tiles that come from HBM, otherwise the rope -- the head of the whole dependent chain -- waits for
  // every cache row first (measured: 6.8 us to the first barrier; the persistent kernel's hook has to keep its order)
  constexpr bool kPlainInputs = std::is_same<Hook, AttnDecNoHook>::value;
  static_assert((G + 1) * (DH / 8) <= NT, "one rope item per thread");
  const bool roper = tid < (G + 1) * (DH / 8);
  const int rh = tid / (DH / 8), rch = tid % (DH / 8);
  // (a.staged: the kv head's [G q heads][k][v] of the new token in LDS, filled by the inputs_ready hook)
  const bf16_t* rsrc = a.staged ? a.staged + rh * DH : ((rh < G) ? row + (kvh * G + rh) * DH : row + (a.H + kvh) * DH);
  Rope8In rin;
  if (kPlainInputs && roper) rin = rope8_load<DH>(rsrc, cs, sn, rch * 8);
  if (!kPlainInputs && roper) {   // the position's cos / sin rows now, q / k when the hook has delivered them
    rin.c = *reinterpret_cast<const uint4*>(cs + rch * 8);
    rin.s = *reinterpret_cast<const uint4*>(sn + rch * 8);
  }
  const int sub = DS == 16 ? lane >> 5 : lane % LPR;
  const int grp = DS == 16 ? wave * 32 + (lane & 31) : wave * RPW + lane / LPR;
  constexpr int NGV = NWV * RPW;
  constexpr int UV = DS == 16 ? 4 : 8;
  const bf16_t* vrow = a.staged ? a.staged + (G + 1) * DH : row + (a.H + a.Hkv + kvh) * DH;
  const bf16_t* vnew = vrow + c0 + sub * 8;
  const bf16_t* vsl = vc + c0 + sub * 8;
  uint4 vpre[UV];
#pragma unroll
  for (int u = 0; u < UV; ++u) {
    const int j = grp + u * NGV;
    vpre[u] = *reinterpret_cast<const uint4*>(vsl + (size_t)(j < t ? j : (t > 0 ? t - 1 : 0)) * DH);   // slot t: see phase C
  }

  // ---- phase A ----
  {
    // key fragments of the first two 16-key tiles of this wave, in flight while q / k_new are roped
    constexpr int PASS = NWV * 32;   // keys per block per iteration (2 tiles of 16 per wave)
    // Key split (batch-1 fused launch only, a.xflags != nullptr, long enough caches): the DH/DS column-slice workgroups of a kv
    // head all need ALL scores of the head, and so far each computed them itself from the head's whole K panel -- 256 B per
    // key through ONE CU's in-flight window (~30 GB/s: 9 ns per cached key and layer).  Instead slice `bx` scores only the
    // keys of chunk `bx` and the slices exchange their scores (fp32 bf16-valued, written through + one flag per workgroup,
    // like the attention -> o hand-over) before the softmax.  Same MFMA code on the same operands: identical scores.
    const bool xs_on = a.xflags != nullptr && nk >= a.xmin;
    constexpr int SLICES = DH / DS;
    const int ck = xs_on ? ((nk + SLICES * 16 - 1) / (SLICES * 16)) * 16 : nk;
    const int klo = xs_on ? (bx * ck < nk ? bx * ck : nk) : 0;
    const int khi = xs_on ? ((bx + 1) * ck < nk ? (bx + 1) * ck : nk) : nk;
    auto key_of = [&](int j0, int tile) { return klo + j0 + wave * 32 + tile * 16 + fr; };
    auto load_tile = [&](int j, bf16x8 (&f)[KB]) {
      const int jc = j < t ? j : (t > 0 ? t - 1 : 0);   // (requesting the tiles before the position load returns, i.e. clamping
                                                       // to Tmax instead of t, measured slower: 3.41 vs 3.38 ms/token)
      const bf16_t* p = kc + (size_t)jc * DH + fq * 8;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) f[kb] = *reinterpret_cast<const bf16x8*>(p + kb * 32);
    };
    // the key tiles of the first NP passes (NP x 256 keys) are all requested up front: each pass that fetched its own
    // tiles paid a full memory round trip inside the dependent chain rope -> scores -> softmax -> P.V
    constexpr int NP = PCY_ATTN_DEC_NP(DS);
    bf16x8 kt[NP][2][KB];
    auto load_group = [&](int j0) {
#pragma unroll
      for (int pp = 0; pp < NP; ++pp)
        if (pp == 0 || klo + j0 + pp * PASS < khi) {
          load_tile(key_of(j0 + pp * PASS, 0), kt[pp][0]);
          load_tile(key_of(j0 + pp * PASS, 1), kt[pp][1]);
        }
    };
    load_group(0);
    inputs_ready();
    // rope q (G heads) and the new key ONCE per block into LDS (bf16), then pick fragments from there
    if (!kPlainInputs && roper) {
      const int e0 = rch * 8, p0 = e0 < DH / 2 ? e0 + DH / 2 : e0 - DH / 2;
      rin.a = *reinterpret_cast<const uint4*>(rsrc + e0);
      rin.b = *reinterpret_cast<const uint4*>(rsrc + p0);
    }
    if (roper) {
      const int hh = rh, ch = rch;
      float tmp[8];
      rope8_math<DH>(rin, ch * 8, tmp);
      *reinterpret_cast<uint4*>(qk + hh * DH + ch * 8) =
          make_uint4(pack_bf(tmp[0], tmp[1]), pack_bf(tmp[2], tmp[3]), pack_bf(tmp[4], tmp[5]), pack_bf(tmp[6], tmp[7]));
      if (hh == G && bx == 0)   // append the new token's K
        *reinterpret_cast<uint4*>(kc + (size_t)t * DH + ch * 8) = *reinterpret_cast<const uint4*>(qk + hh * DH + ch * 8);
    }
    if (bx == 0 && tid >= NT - DH / 8) {  // ... and its V
      const int ch = tid - (NT - DH / 8);
      *reinterpret_cast<uint4*>(vc + (size_t)t * DH + ch * 8) = *reinterpret_cast<const uint4*>(vrow + ch * 8);
    }
    lds_barrier();
    // A operand: roped q of head `fr` (zero rows for fr >= G); new key in B-fragment layout
    bf16x8 qf[KB], knf[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      knf[kb] = *reinterpret_cast<const bf16x8*>(qk + G * DH + kb * 32 + fq * 8);
      if (fr < G) qf[kb] = *reinterpret_cast<const bf16x8*>(qk + fr * DH + kb * 32 + fq * 8);
      else qf[kb] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }

    for (int j0 = 0; klo + j0 < khi; j0 += NP * PASS) {
      if (j0 > 0) load_group(j0);
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const int jb = j0 + pp * PASS;
        if (klo + jb < khi) {
#pragma unroll
          for (int tile = 0; tile < 2; ++tile) {
            const int j = key_of(jb, tile);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
              bf16x8 kf = kt[pp][tile][kb];
              if (j == t) kf = knf[kb];
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kb], kf, acc, 0, 0, 0);
            }
            // D[row = head = fq*4 + r][col = key fr]
            if (j < khi) {
              const bool kept = (keep && j < t) ? keep[j] != 0 : true;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int head = fq * 4 + r;
                if (head < G) sc[head * scld + j] = kept ? rbf(rbf(acc[r]) * a.scale) : PCY_BF16_MIN;
              }
            }
          }
        }
      }
    }
  }
  lds_barrier();
  if (a.xflags != nullptr && nk >= a.xmin) {
    constexpr int SLICES = DH / DS;
    const int ck = ((nk + SLICES * 16 - 1) / (SLICES * 16)) * 16;
    const int klo = bx * ck < nk ? bx * ck : nk, khi = (bx + 1) * ck < nk ? (bx + 1) * ck : nk;
    unsigned* xs = reinterpret_cast<unsigned*>(a.scratch) + ((size_t)b * a.H + kvh * G) * scld;   // [G][Tmax + 1] of this kv head
    unsigned* fl = a.xflags + ((size_t)b * a.Hkv + kvh) * SLICES;
    const int cnt = khi - klo;
    for (int i = tid; i < G * cnt; i += NT) {
      const int g = i / cnt, j = klo + i % cnt;
      __hip_atomic_store(xs + (size_t)g * scld + j, __float_as_uint(sc[g * scld + j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its written-through stores have left
    __syncthreads();
    if (tid == 0) __hip_atomic_store(fl + bx, a.xepoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
      unsigned spins = 0;
      for (;;) {
        const bool ok = lane >= SLICES || __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.xepoch;
        if (__builtin_amdgcn_readfirstlane(__all(ok))) break;
        __builtin_amdgcn_s_sleep(1);
        if (pcy_wait_give_up(spins, 1u << 18, a.xerr, 3u, lane)) break;
      }
    }
    __syncthreads();
    // the other slices' scores: eight independent agent-scope loads per thread in flight, then the LDS stores (a load -> store
    // pair per iteration paid one memory round trip each)
    constexpr int UB = 8;
    for (int i0 = tid; i0 < G * nk; i0 += NT * UB) {
      unsigned v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i = i0 + u * NT;
        const int g = i / nk, j = i - g * nk;
        v[u] = 0;
        if (i < G * nk) asm volatile("global_load_dword %0, %1, off sc1" : "=&v"(v[u]) : "v"(xs + (size_t)g * scld + j) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        asm volatile("" : "+v"(v[u]));
        const int i = i0 + u * NT;
        const int g = i / nk, j = i - g * nk;
        if (i < G * nk && !(j >= klo && j < khi)) sc[g * scld + j] = __uint_as_float(v[u]);
      }
    }
    lds_barrier();
  }
  if (a.dbg == 2) { if (tid == 0) a.o[(size_t)b * a.ldo + kvh * G * DH + c0] = f2bf(sc[0]); return; }
  // ---- phase B ----
  {
    float mx[G], se[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      mx[g] = -INFINITY;
      for (int j = tid; j < nk; j += NT) mx[g] = fmaxf(mx[g], sc[g * scld + j]);
      mx[g] = wave_max_dpp(mx[g]);
      if (lane == 0) wred[wave * G + g] = mx[g];
    }
    lds_barrier();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      mx[g] = wred[g];
#pragma unroll
      for (int w = 1; w < NWV; ++w) mx[g] = fmaxf(mx[g], wred[w * G + g]);
      se[g] = 0.f;
      for (int j = tid; j < nk; j += NT) se[g] += expf(sc[g * scld + j] - mx[g]);
      se[g] = wave_sum_dpp(se[g]);
    }
    lds_barrier();
#pragma unroll
    for (int g = 0; g < G; ++g) if (lane == 0) wred[wave * G + g] = se[g];
    lds_barrier();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float l = wred[g];
#pragma unroll
      for (int w = 1; w < NWV; ++w) l += wred[w * G + g];
      for (int j = tid; j < nk; j += NT) sc[g * scld + j] = rbf(expf(sc[g * scld + j] - mx[g]) / l);
    }
  }
  lds_barrier();
  if (a.dbg == 3) { if (tid == 0) a.o[(size_t)b * a.ldo + kvh * G * DH + c0] = f2bf(sc[0]); return; }
  // ---- phase C ----
  float acc[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
  // (PCY_ATTN_DEC_VPF: the V rows of pass p + 1 are requested before pass p is consumed -- one workgroup per CU (the fused decode steps)
  // otherwise pays a memory round trip per pass of 256 / 512 keys; same rows, same order of the sums)
  uint4 vnx[UV];
  for (int j0 = 0; j0 < nk; j0 += UV * NGV) {
    uint4 vv[UV];
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      if (j0 == 0) {   // first pass: rows fetched at kernel start; slot t comes straight from the projection
        vv[u] = vpre[u];
        if (grp + u * NGV == t) vv[u] = *reinterpret_cast<const uint4*>(vnew);
        continue;
      }
      if (PCY_ATTN_DEC_VPF) { vv[u] = vnx[u]; continue; }
      const int j = j0 + grp + u * NGV;
      const bf16_t* src = (j < t) ? vsl + (size_t)j * DH : vnew;   // slot t comes straight from the projection
      vv[u] = *reinterpret_cast<const uint4*>(src);
    }
    if (PCY_ATTN_DEC_VPF && j0 + UV * NGV < nk) {
#pragma unroll
      for (int u = 0; u < UV; ++u) {
        const int j = j0 + UV * NGV + grp + u * NGV;
        const bf16_t* src = (j < t) ? vsl + (size_t)j * DH : vnew;
        vnx[u] = *reinterpret_cast<const uint4*>(src);
      }
    }
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const int j = j0 + grp + u * NGV;
      if (j < nk) {
        const float vf[8] = {lo_bf(vv[u].x), hi_bf(vv[u].x), lo_bf(vv[u].y), hi_bf(vv[u].y),
                             lo_bf(vv[u].z), hi_bf(vv[u].z), lo_bf(vv[u].w), hi_bf(vv[u].w)};
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float pj = sc[g * scld + j];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[g][e] += pj * vf[e];
        }
      }
    }
  }
  // reduce over the 32 key groups of a wave half: 16-lane DPP row sums, then the two rows of the half
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[g][e];
      if constexpr (DS == 16) {
        v = row16_sum(v);
        v += __shfl_xor(v, 16, 64);
      } else {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      }
      acc[g][e] = v;
    }
  if (DS == 16 ? (lane & 31) == 0 : lane < LPR) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(wave * G + g) * DS + sub * 8 + e] = acc[g][e];
  }
  lds_barrier();
  for (int i = tid; i < G * DS; i += NT) {
    float s = red[i];
#pragma unroll
    for (int w = 1; w < NWV; ++w) s += red[w * G * DS + i];
    const int g = i / DS, c = i % DS;
    const size_t oi = (size_t)b * a.ldo + (kvh * G + g) * DH + c0 + c;
    if (a.o_tag) __hip_atomic_store(a.o_tag + oi, (a.tag << 16) | f2bf(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (a.o_sc1) __hip_atomic_store(a.o + oi, f2bf(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else a.o[oi] = f2bf(s);
  }
}


__host__ __device__ inline size_t attn_dec_smem_bytes(int G, int DS, int DH, int Tmax) {
  return sizeof(float) * ((size_t)G * (Tmax + 1) + 8 * G * DS + 8 * G + 8) + (size_t)(G + 1) * DH * 2 + 64;
}

}  // namespace
