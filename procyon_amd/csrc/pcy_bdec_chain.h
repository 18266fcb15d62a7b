// Batched decode: the o projection, both RMSNorms, the MLP and the next layer's qkv projection in one launch (PcyBdChainArgs,
// pcy_internal.h).  Included by pcy_gemv.hip behind the batched MFMA GEMV kernels, whose copy / MFMA / epilogue code it reuses.
//
// A stage is the gemv_mfma4_kernel schedule for ONE unit per workgroup (64 weight rows -- 128 for gate/up -- over one K range; x
// and the weights in LDS rings of equal depth, pairs issued S - 1 super-steps ahead), split in two: `prefetch` issues the weight
// copies of the first S - 1 super-steps (they depend on nothing) BEFORE the grid barrier that ends the previous stage, `run`
// issues the x copies behind the barrier and takes up the steady schedule.  The counted waits of the first S - 2 iterations differ
// from the steady ones because the first x copies are younger than the first weight copies.
#pragma once

namespace {

constexpr int BD_WT = 16 * 256;      // one 16-row tile of one 128-k super-step
constexpr int BD_GRID = 256;

// Stores that other workgroups read in the next stage are written THROUGH to memory (agent scope, like pcy_handover.h): a grid
// barrier then only has to wait for their acknowledgement.  (First version: plain stores + a release fence = buffer_wbl2 sc1,
// which walks the XCD's whole L2 -- 20-25 us per barrier, measured with the in-kernel stamps; the launch ran 253 us per layer.)
// (compiler-generated agent-scope stores of 8 bytes: an inline-asm global_store is invisible to hipcc's hazard recognizer -- it
// overwrote the address / data VGPRs in the next cycles: memory aperture violation)
__device__ __forceinline__ void bd_st16(void* p, uint4 v) {
  st8_agent(p, v.x, v.y);
  st8_agent(reinterpret_cast<char*>(p) + 8, v.z, v.w);
}
__device__ __forceinline__ void bd_st16f(void* p, f32x4 v) {
  st8_agent(p, __float_as_uint(v[0]), __float_as_uint(v[1]));
  st8_agent(reinterpret_cast<char*>(p) + 8, __float_as_uint(v[2]), __float_as_uint(v[3]));
}
__device__ __forceinline__ void bd_st8(void* p, uint2 v) { st8_agent(p, v.x, v.y); }

// lane holds D[n = r0 + fq*4 + r][b = bt*16 + fr]: K-split partial sums -> ws[split][b][n .. n+3]   (N % 64 == 0)
__device__ __forceinline__ void bd_store_partial(float* ws, int split, int B, int N, f32x4 (&acc)[1][2], int r0, int fr, int fq) {
#pragma unroll
  for (int bt = 0; bt < 2; ++bt) {
    const int b = bt * 16 + fr;
    if (b < B) bd_st16f(ws + ((size_t)split * B + b) * N + r0 + fq * 4, acc[0][bt]);
  }
}
// ... SwiGLU of the wave's gate / up tiles -> act[b][f .. f+3]   (mfma_gemv_epilogue's EPI_SWIGLU arithmetic)
__device__ __forceinline__ void bd_store_swiglu(bf16_t* act, int B, int F, f32x4 (&acc)[2][2], int r0, int fr, int fq) {
#pragma unroll
  for (int bt = 0; bt < 2; ++bt) {
    const int b = bt * 16 + fr;
    if (b >= B) continue;
    const int f = (r0 >> 5) * 16 + fq * 4;
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = rbf(silu_f(rbf(acc[0][bt][r]))) * rbf(acc[1][bt][r]);
    bd_st8(act + (size_t)b * F + f, make_uint2(pack_bf(o[0], o[1]), pack_bf(o[2], o[3])));
  }
}

template <int RT> struct BdSrc { const bf16_t* w[RT][4]; };
struct BdXSrc { const bf16_t* p[2]; int dst[2]; };

template <int RT>
__device__ __forceinline__ BdSrc<RT> bd_wsrc(const bf16_t* W, int K, int r0, int nrows, int kbeg, int lane) {
  BdSrc<RT> s;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = q * 4 + (lane >> 4);
      int r = r0 + rt * 16 + row;
      r = r < nrows ? r : nrows - 1;
      s.w[rt][q] = W + (size_t)r * K + kbeg + ((lane & 15) ^ row) * 8;
    }
  return s;
}
__device__ __forceinline__ BdXSrc bd_xsrc(const bf16_t* x, int ldx, int B, int kbeg, int wave, int lane) {
  BdXSrc s;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int i = wave * 2 + t, bt = i >> 2, q = i & 3;
    const int row = q * 4 + (lane >> 4);
    int b = bt * 16 + row;
    b = b < B ? b : B - 1;
    s.p[t] = x + (size_t)b * ldx + kbeg + ((lane & 15) ^ row) * 8;
    s.dst[t] = bt * BD_WT + q * 1024;
  }
  return s;
}
template <int RT, int S>
__device__ __forceinline__ void bd_issue_w(const BdSrc<RT>& src, char* wring, int ss, int nss) {
  const int k = (ss < nss ? ss : nss - 1) * 128;
  char* dst = wring + (ss % S) * RT * BD_WT;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((gv_gptr_t)(src.w[rt][q] + k), (gv_lds_ptr_t)(dst + rt * BD_WT + q * 1024), 16, 0, 2);
}
template <int S>
__device__ __forceinline__ void bd_issue_x(const BdXSrc& src, char* xs, int ss, int nss) {
  const int k = (ss < nss ? ss : nss - 1) * 128;
  char* xb = xs + (ss % S) * 2 * BD_WT;
#pragma unroll
  for (int t = 0; t < 2; ++t) __builtin_amdgcn_global_load_lds((gv_gptr_t)(src.p[t] + k), (gv_lds_ptr_t)(xb + src.dst[t]), 16, 0, 0);
}
template <int RT, int S>
__device__ __forceinline__ void bd_prefetch(const BdSrc<RT>& src, char* wring, int nss) {
#pragma unroll
  for (int i = 0; i < S - 1; ++i) bd_issue_w<RT, S>(src, wring, i, nss);
}
// acc += W[unit rows][k range] . x[B][k range]^T ; the weight copies of super-steps 0 .. S-2 are in flight (bd_prefetch)
template <int RT, int S>
__device__ __forceinline__ void bd_run(const BdSrc<RT>& src, const BdXSrc& xsrc, char* wring, char* xs, int nss, f32x4 (&acc)[RT][2], int lane) {
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int i = 0; i < S - 1; ++i) bd_issue_x<S>(xsrc, xs, i, nss);
  for (int ss = 0; ss < nss; ++ss) {
    // outstanding, oldest first: W(0..S-2) x(0..S-2) [x(S-1) W(S-1)] [x(S) W(S)] ...; pair ss must have landed
    if (ss == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * 2) : "memory");
    else if (ss < S - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * 2 + RT * 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * (2 + RT * 4)) : "memory");
    __builtin_amdgcn_s_barrier();
    bd_issue_x<S>(xsrc, xs, ss + S - 1, nss);
    bd_issue_w<RT, S>(src, wring, ss + S - 1, nss);
    const int slot = ss % S;
    const char* wb = wring + slot * RT * BD_WT;
    const char* xb = xs + slot * 2 * BD_WT;
    bf16x8 wf[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[rt][j] = *reinterpret_cast<const bf16x8*>(wb + rt * BD_WT + fr * 256 + (((j * 4 + fq) ^ fr) << 4));
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      bf16x8 xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xb + bt * BD_WT + fr * 256 + (((j * 4 + fq) ^ fr) << 4));
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[rt][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[rt][j], xf[j], acc[rt][bt], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every wave is done with the rings: the next stage's prefetch may overwrite them
}

// Grid barrier, first half: this workgroup's (written-through) stores have been acknowledged, then it is counted.
__device__ __forceinline__ void bd_arrive(unsigned* ctr) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... second half: wait until `target` workgroups have been counted; what they stored is then visible to this workgroup.
__device__ __forceinline__ void bd_wait(unsigned* ctr, unsigned target, unsigned* err, unsigned code) {
  if (threadIdx.x == 0) {   // ONE lane asks: an agent-scope load is a request per lane, 64 x 256 of them per round starve the arrivals
    unsigned spins = 0;
    for (;;) {
      const unsigned v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int)(v - target) >= 0) break;
      if (pcy_wait_give_up(spins, 1u << 20, err, code, 0)) break;
      __builtin_amdgcn_s_sleep(4);
    }
  }
  if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __builtin_amdgcn_s_barrier();
}

template <int RT>
__device__ __forceinline__ void bd_zero(f32x4 (&acc)[RT][2]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) acc[rt][bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// K-split finish + residual + the RMSNorm that follows: gemv_splitk_finish_norm_kernel's arithmetic for batch row b
__device__ __forceinline__ void bd_finish_norm(const float* ws, int ksplit, int B, int N, int b, bf16_t* y /* resid in, y out */,
                                               const bf16_t* __restrict__ norm_w, bf16_t* xn, float eps, int cast, float* red) {
  constexpr int MAXI = 4;
  float xv[MAXI][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int k = threadIdx.x * 8 + it * 2048;
    if (k >= N) break;
    uint32_t packed[4];
    const uint4 rv = *reinterpret_cast<const uint4*>(y + (size_t)b * N + k);
    const uint32_t rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = k + h * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(ws + (size_t)b * N + n);
      for (int s_ = 1; s_ < ksplit; ++s_) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(ws + ((size_t)s_ * B + b) * N + n);
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t w = rr[h * 2 + (r >> 1)];
        float o = rbf(v[r]);
        o = rbf(o + ((r & 1) ? hi_bf(w) : lo_bf(w)));
        xv[it][h * 4 + r] = o;
      }
      packed[h * 2] = pack_bf(xv[it][h * 4], xv[it][h * 4 + 1]);
      packed[h * 2 + 1] = pack_bf(xv[it][h * 4 + 2], xv[it][h * 4 + 3]);
    }
    bd_st16(y + (size_t)b * N + k, make_uint4(packed[0], packed[1], packed[2], packed[3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float p = xv[it][2 * j], q = xv[it][2 * j + 1]; ss += p * p + q * q; }
  }
  ss = block_sum<256>(ss, red);
  const float rstd = rsqrtf(ss / (float)N + eps);
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int k = threadIdx.x * 8 + it * 2048;
    if (k >= N) break;
    const uint4 g = *reinterpret_cast<const uint4*>(norm_w + k);
    const uint32_t gg[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float p = xv[it][2 * j] * rstd, q = xv[it][2 * j + 1] * rstd;
      if (cast == 0) { p = rbf(p); q = rbf(q); }
      o[j] = pack_bf(lo_bf(gg[j]) * p, hi_bf(gg[j]) * q);
    }
    bd_st16(xn + (size_t)b * N + k, make_uint4(o[0], o[1], o[2], o[3]));
  }
}

constexpr int BD_S1 = 6, BD_S2 = 4;              // ring depths: one row tile per wave / gate + up tiles per wave
constexpr int BD_KS_O = 4, BD_KS_DOWN = 4, BD_KS_QKV = 2;   // the K splits of the launch-per-stage path at these shapes
constexpr size_t BD_SMEM = 163840;               // 4 waves x 4 x 2 x 4 KB + 4 x 8 KB  (>= 4 x 6 x 4 KB + 6 x 8 KB)

__global__ __launch_bounds__(256) void bdec_chain_kernel(PcyBdChainArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char bd_smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wg = blockIdx.x;
  char* wring1 = bd_smem + wave * BD_S1 * BD_WT;            // [S1][16 rows][256 B]
  char* xs1 = bd_smem + 4 * BD_S1 * BD_WT;                  // [S1][2][4 KB]
  char* wring2 = bd_smem + wave * BD_S2 * 2 * BD_WT;        // [S2][2][16 rows][256 B]
  char* xs2 = bd_smem + 4 * BD_S2 * 2 * BD_WT;
  float* red = reinterpret_cast<float*>(xs2);                // finish stages: nothing of the x rings is in flight then
  const int fr = lane & 15, fq = lane >> 4;
  // the counter is a multiple of the grid size between launches; early arrivals of THIS launch at its first barrier add < grid
  unsigned base;
  {
    unsigned* sh = reinterpret_cast<unsigned*>(xs2);
    if (threadIdx.x == 0) {
      const unsigned v = __hip_atomic_load(a.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh[0] = v - v % BD_GRID;
    }
    __syncthreads();
    base = sh[0];
    __syncthreads();
  }
  unsigned nbar = 0;
  unsigned long long* tr = a.trace ? a.trace + (size_t)wg * 16 : nullptr;
#define BD_T(i) if (tr && threadIdx.x == 0) tr[i] = wall_clock64();
  BD_T(0)

  // ---- o projection: 64 row groups x 4 K ranges ----
  {
    const int g = wg % (a.d / 64), sp = wg / (a.d / 64);
    const int ks = a.Ko / BD_KS_O, r0 = (g * 4 + wave) * 16;
    const BdSrc<1> src = bd_wsrc<1>(a.wo, a.Ko, r0, a.d, sp * ks, lane);
    const BdXSrc xsrc = bd_xsrc(a.ao, a.Ko, a.B, sp * ks, wave, lane);
    bd_prefetch<1, BD_S1>(src, wring1, ks / 128);
    f32x4 acc[1][2];
    bd_zero<1>(acc);
    bd_run<1, BD_S1>(src, xsrc, wring1, xs1, ks / 128, acc, lane);
    BD_T(1)
    bd_store_partial(a.ws, sp, a.B, a.d, acc, r0, fr, fq);
  }
  // gate/up: 224 units of 128 rows over the whole K
  const int gu_units = 2 * a.F / 128;
  const bool gu_mine = wg < gu_units;
  const int gu_r0 = (wg * 4 + wave) * 32;
  const BdSrc<2> gsrc = bd_wsrc<2>(a.wgu, a.d, gu_r0, 2 * a.F, 0, lane);
  bd_arrive(a.ctr);
  if (wg >= a.B && gu_mine) bd_prefetch<2, BD_S2>(gsrc, wring2, a.d / 128);   // (the finishing workgroups: behind their finish)
  BD_T(2)
  bd_wait(a.ctr, base + (++nbar) * BD_GRID, a.err, 20u);
  BD_T(3)
  if (wg < a.B) {
    bd_finish_norm(a.ws, BD_KS_O, a.B, a.d, wg, a.x, a.ln2, a.xn, a.rms_eps, a.rms_cast, red);
    __syncthreads();
  }
  bd_arrive(a.ctr);
  if (wg < a.B && gu_mine) bd_prefetch<2, BD_S2>(gsrc, wring2, a.d / 128);
  BD_T(4)
  bd_wait(a.ctr, base + (++nbar) * BD_GRID, a.err, 21u);
  BD_T(5)
  // ---- gate/up + SwiGLU ----
  const int ksd = a.F / BD_KS_DOWN;
  const int dg = wg % (a.d / 64), dsp = wg / (a.d / 64);
  const int d_r0 = (dg * 4 + wave) * 16;
  const BdSrc<1> dsrc = bd_wsrc<1>(a.wdown, a.F, d_r0, a.d, dsp * ksd, lane);
  if (gu_mine) {
    const BdXSrc xsrc = bd_xsrc(a.xn, a.d, a.B, 0, wave, lane);
    f32x4 acc[2][2];
    bd_zero<2>(acc);
    bd_run<2, BD_S2>(gsrc, xsrc, wring2, xs2, a.d / 128, acc, lane);
    bd_store_swiglu(a.act, a.B, a.F, acc, gu_r0, fr, fq);
  }
  bd_arrive(a.ctr);
  bd_prefetch<1, BD_S1>(dsrc, wring1, ksd / 128);
  BD_T(6)
  bd_wait(a.ctr, base + (++nbar) * BD_GRID, a.err, 22u);
  BD_T(7)
  // ---- down projection: 64 row groups x 4 K ranges ----
  {
    const BdXSrc xsrc = bd_xsrc(a.act, a.F, a.B, dsp * ksd, wave, lane);
    f32x4 acc[1][2];
    bd_zero<1>(acc);
    bd_run<1, BD_S1>(dsrc, xsrc, wring1, xs1, ksd / 128, acc, lane);
    bd_store_partial(a.ws, dsp, a.B, a.d, acc, d_r0, fr, fq);
  }
  // next layer's qkv: Nq / 64 row groups x 2 K ranges
  const int q_units = a.Nq ? (a.Nq / 64) * BD_KS_QKV : 0;
  const bool q_mine = wg < q_units;
  const int ksq = a.d / BD_KS_QKV;
  const int qg = a.Nq ? wg % (a.Nq / 64) : 0, qsp = a.Nq ? wg / (a.Nq / 64) : 0;
  const int q_r0 = (qg * 4 + wave) * 16;
  const BdSrc<1> qsrc = bd_wsrc<1>(a.next_wqkv ? a.next_wqkv : a.wo, a.d, q_r0, a.Nq ? a.Nq : 1, qsp * ksq, lane);
  bd_arrive(a.ctr);
  if (wg >= a.B && q_mine) bd_prefetch<1, BD_S1>(qsrc, wring1, ksq / 128);
  BD_T(8)
  bd_wait(a.ctr, base + (++nbar) * BD_GRID, a.err, 23u);
  BD_T(9)
  if (wg < a.B) {
    bd_finish_norm(a.ws, BD_KS_DOWN, a.B, a.d, wg, a.x, a.next_norm, a.xn, a.rms_eps, a.rms_cast, red);
    __syncthreads();
  }
  if (!a.Nq) return;
  bd_arrive(a.ctr);
  if (wg < a.B && q_mine) bd_prefetch<1, BD_S1>(qsrc, wring1, ksq / 128);
  BD_T(10)
  bd_wait(a.ctr, base + (++nbar) * BD_GRID, a.err, 24u);
  BD_T(11)
  if (q_mine) {
    const BdXSrc xsrc = bd_xsrc(a.xn, a.d, a.B, qsp * ksq, wave, lane);
    f32x4 acc[1][2];
    bd_zero<1>(acc);
    bd_run<1, BD_S1>(qsrc, xsrc, wring1, xs1, ksq / 128, acc, lane);
    bd_store_partial(a.ws, qsp, a.B, a.Nq, acc, q_r0, fr, fq);
  }
  BD_T(12)
#undef BD_T
}

}  // namespace

bool pcy_launch_bd_chain(hipStream_t s, const PcyBdChainArgs& a, int n_cu, int* qkv_splits) {
  if (qkv_splits) *qkv_splits = 0;
  if (n_cu != BD_GRID || a.B < 5 || a.B > 32 || !a.ctr) return false;
  if (a.d != 4096 || a.d % 64 || a.d / 64 * BD_KS_O != BD_GRID || a.d / 64 * BD_KS_DOWN != BD_GRID) return false;
  if (a.Ko % (BD_KS_O * 128) || a.F % (BD_KS_DOWN * 128) || (2 * a.F) % 128 || 2 * a.F / 128 > BD_GRID) return false;
  if (a.Ko / BD_KS_O / 128 < BD_S1 || a.F / BD_KS_DOWN / 128 < BD_S1 || a.d / 128 < BD_S1) return false;
  if (a.Nq && (a.Nq % 64 || a.Nq / 64 * BD_KS_QKV > BD_GRID || a.d % (BD_KS_QKV * 128) || a.d / BD_KS_QKV / 128 < BD_S1)) return false;
  if (a.ws_bytes < (size_t)4 * a.B * a.d * 4 || (a.Nq && a.ws_bytes < (size_t)BD_KS_QKV * a.B * a.Nq * 4)) return false;
  static int ok = 0;   // 1 usable, -1 not (LDS size refused or the grid is not resident at once)
  if (!ok) {
    ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&bdec_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)BD_SMEM) == hipSuccess &&
                 pcy_all_resident(bdec_chain_kernel, 256, BD_SMEM, BD_GRID, n_cu)
             ? 1 : -1;
    if (ok < 0) (void)hipGetLastError();
  }
  if (ok < 0) return false;
  hipLaunchKernelGGL(bdec_chain_kernel, dim3(BD_GRID), dim3(256), BD_SMEM, s, a);
  ++g_pcy_dispatch[PCY_DISPATCH_BD_CHAIN];
  if (qkv_splits) *qkv_splits = a.Nq ? BD_KS_QKV : 0;
  return true;
}
