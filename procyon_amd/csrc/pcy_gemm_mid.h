// Mid-M bf16 GEMM (included by pcy_gemm.hip inside its anonymous namespace, behind the epilogues).
//
// The regime: ONE 1024-residue protein through ESM2 (M = 1026) or ONE 512-token prompt through the Llama prefill (M = 512) --
// /root/reference/procyon/model/esm.py:517-538, /root/reference/procyon/model/pmc_llama.py:571-588 at batch 1.  A GEMM is then a
// single round of tiles: every workgroup runs prologue -> k-loop -> epilogue exactly once, nothing overlaps a neighbour's tail, and
// with the two-stage `__syncthreads()` loop of gemm_kernel each k-step waits for a full L2 / HBM round trip with ONE stage in flight
// (measured before this kernel: 31 us for the 10 GFLOP of the ESM qkv projection at M = 1026 = 13 % of the MFMA peak; 39 us for o / fc2).
//
// Design for that regime:
//   * an S-stage LDS ring (S = 3 .. 6, 64-k stages) filled by `buffer_load ... lds` pieces with a COUNTED s_waitcnt vmcnt -- S - 2
//     stages stay in flight across the raw s_barrier of a k-step, so the W panels (HBM misses: the weights of a layer are read once)
//     and the A panels (L2) are requested 2-4 k-steps before they are read; exact tail (no dummy refills: at K = 1280 a tile has only
//     20 k-steps);
//   * tile shape and wave layout are template parameters (TM x TN, WM x WN waves of FM x FN MFMA tiles) so that the launcher can pick,
//     per GEMM, the shape whose tile count fills the 256 CUs ONCE (e.g. 128 x 64 for N = 1280 at M = 1026: 180 tiles instead of 90);
//   * one workgroup per CU by LDS size; 4 or 8 waves;
//   * same per-element arithmetic as every other bf16 GEMM here (16x16x32 MFMA over ascending k, fp32 accumulation, the shared
//     epilogues): a row's bits do not depend on which kernel / tile shape computed it -- the packing invariance of the encoder holds.
//   * (tried, dropped: a software-pipelined form -- fragments of the next 32-k sub-step read under the MFMAs of the current one, two
//     register sets, the counted wait one stage earlier -- bit-identical and 8-25 % SLOWER at every shape (o 11.9 -> 13.6 us, fc2 33.9 ->
//     40.4, fc1 28.0 -> 30.9, Llama qkv 41.0 -> 43.7): two waves per SIMD already cover each other's LDS latency, the earlier wait costs a
//     stage of prefetch depth)
//   * (tried, dropped: a K-split form writing fp32 partial tiles for the Llama projections at M = 512 -- prefill 10.19 -> 10.33 / 11.2 ms)
#pragma once

template <int N>
__device__ __forceinline__ void mid_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// s_waitcnt vmcnt(left * PPS) for a run-time `left` in [0, MAXL] (the immediate has to be a constant: a chain of compares that
// folds away where `left` is an unrolled loop variable)
template <int PPS, int MAXL>
__device__ __forceinline__ void mid_wait_left(int left) {
  if constexpr (MAXL >= 0) {
    if (left == MAXL) { mid_wait_vmcnt<MAXL * PPS>(); return; }
    mid_wait_left<PPS, MAXL - 1>(left);
  }
}

// (the body is a __device__ function: buffer-resource builtins written directly inside a __global__ template make the HOST pass drop the
// kernel's stub without a diagnostic -- undefined symbol at load time)
template <int EPI, int TM, int TN, int WM, int WN, int S>
__device__ __forceinline__ void gemm_mid_body(const PcyGemmArgs& a) {
  constexpr int BK = 64, NW = WM * WN, NT = NW * 64;
  constexpr int FM = TM / WM / 16, FN = TN / WN / 16;            // MFMA tiles per wave: FM token tiles x FN feature tiles
  constexpr int TILE_A = TM * BK * 2, TILE_W = TN * BK * 2, STAGE = TILE_A + TILE_W;
  constexpr int PA = TM / 8, PT = (TM + TN) / 8;                  // 1-KiB pieces (8 rows x 128 B) of the A part / of a stage
  constexpr int PPS = (PT + NW - 1) / NW;                         // pieces per wave and stage (the last one repeated where PT % NW != 0)
  static_assert(TM % (WM * 16) == 0 && TN % (WN * 16) == 0 && S >= 2 && (S - 1) * PPS < 64, "tile / ring geometry");
  extern __shared__ __attribute__((aligned(1024))) char smem_dyn[];
  char* smem = smem_dyn;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  int m0, n0;
  tile_origin<TM, TN>(a, tile, m0, n0);
  const int nk = a.K / BK;
  constexpr int kbeg = 0;

  // this wave's pieces of a stage: piece p = wave + i * NW; p < PA: rows 8p .. 8p+7 of the A tile, else of the W tile.  Byte offset of the
  // lane inside its operand (row clamped at the matrix edge, 16-byte chunk XOR-swizzled by the row -- the read side applies the same
  // XOR) is the same for every k-step; the k offset travels as the instruction's scalar offset.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.A + (size_t)m0 * a.lda), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(a.W + (size_t)n0 * a.K), 0, 0x7fffffff, 0x00020000);
  int voff[PPS];
#pragma unroll
  for (int i = 0; i < PPS; ++i) {
    int p = wave + i * NW;
    p = p < PT ? p : PT - 1;
    const bool isA = p < PA;
    const int r = (isA ? p : p - PA) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (r & 7);
    const int valid = isA ? a.M - m0 : a.N - n0;
    const int rl = r < valid ? r : valid - 1;
    voff[i] = (rl * (isA ? a.lda : a.K) + c * 8) * 2;
  }
  auto issue = [&](int kt) __attribute__((always_inline)) {
    char* st = smem + (kt % S) * STAGE;
    const int koff = (kbeg + kt * BK) * 2;
#pragma unroll
    for (int i = 0; i < PPS; ++i) {
      int p = wave + i * NW;
      p = p < PT ? p : PT - 1;
      if (p < PA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(st + p * 1024), 16, voff[i], koff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(st + TILE_A + (p - PA) * 1024), 16, voff[i], koff, 0, 0);
    }
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int kt) __attribute__((always_inline)) {
    const char* As = smem + (kt % S) * STAGE;
    const char* Ws = As + TILE_A;
#pragma unroll
    for (int kb = 0; kb < BK / 32; ++kb) {
      bf16x8 xf[FM], wf[FN];
#pragma unroll
      for (int j = 0; j < FM; ++j) xf[j] = lds_frag<BK>(As, wm * FM * 16 + j * 16 + fr, kb * 4 + fq);
#pragma unroll
      for (int i = 0; i < FN; ++i) wf[i] = lds_frag<BK>(Ws, wn * FN * 16 + i * 16 + fr, kb * 4 + fq);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  };

  // prologue: stages 0 .. S-2
#pragma unroll
  for (int st = 0; st < S - 1; ++st)
    if (st < nk) issue(st);
  // steady state: stage kt has landed when at most S - 2 younger stages are in flight; behind the barrier every wave's pieces of it are
  // there and every wave has finished reading stage kt - 1, whose slot takes stage kt + S - 1
  int kt = 0;
  for (; kt + S - 1 < nk; ++kt) {
    mid_wait_vmcnt<(S - 2) * PPS>();
    __builtin_amdgcn_s_barrier();
    issue(kt + S - 1);
    compute(kt);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // tail: the last min(nk, S - 1) stages, nothing left to issue; stage kt has landed when only the nk - 1 - kt younger ones are in flight
#pragma unroll
  for (int left = S - 2; left >= 0; --left) {
    if (nk - 1 - kt == left) {
      mid_wait_left<PPS, S - 2>(left);
      __builtin_amdgcn_s_barrier();
      compute(kt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ++kt;
    }
  }

  if constexpr (EPI == EPI_GELU_ESM) {
    gelu_lut_to_lds<NT>(smem);   // (every stage buffer is dead: the tail waited for vmcnt(0) and every wave passed the last barrier + its reads)
    gemm_epilogue<EPI, FN, FM, false>(a, acc, m0, n0, wm, wn, fr, fq, reinterpret_cast<const uint16_t*>(smem));
    return;
  }
  gemm_epilogue<EPI, FN, FM, (FN % 4 == 0)>(a, acc, m0, n0, wm, wn, fr, fq);
}
template <int EPI, int TM, int TN, int WM, int WN, int S>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel_mid(PcyGemmArgs a) { gemm_mid_body<EPI, TM, TN, WM, WN, S>(a); }

// configurations (id -> TM, TN, WM, WN, S); LDS = S x (TM + TN) x 128 B <= 160 KiB
#define PCY_MID_CONFIGS(X)                                                                                        \
  X(1, 128, 128, 2, 2, 4) /* 4 waves of 64 x 64, 128 KiB */                                                        \
  X(2, 128, 128, 2, 4, 4) /* 8 waves of 64 tokens x 32 features */                                                 \
  X(3, 128, 128, 4, 2, 4) /* 8 waves of 32 tokens x 64 features */                                                 \
  X(4, 128, 64, 2, 2, 6)  /* 4 waves of 64 x 32, 144 KiB */                                                        \
  X(5, 128, 64, 4, 2, 6)  /* 8 waves of 32 x 32 */                                                                 \
  X(6, 256, 128, 4, 2, 3) /* 8 waves of 64 x 64, 144 KiB */                                                        \
  X(7, 128, 256, 2, 4, 3) /* 8 waves of 64 x 64 */                                                                 \
  X(8, 64, 128, 2, 2, 6)  /* 4 waves of 32 tokens x 64 features, 144 KiB */                                        \
  X(9, 64, 64, 2, 2, 8)   /* 4 waves of 32 x 32, 128 KiB */                                                        \
  X(10, 256, 128, 4, 2, 2) /* as 6, two stages (96 KiB) */                                                         \
  X(11, 128, 128, 2, 2, 3) /* as 1, 96 KiB */                                                                      \
  X(12, 128, 128, 2, 2, 5) /* as 1, 160 KiB */                                                                     \
  X(13, 128, 96, 4, 2, 5)  /* 8 waves of 32 tokens x 48 features, 140 KiB: N = 6144 at M = 512 in ONE full round of 256 tiles */ \
  X(14, 128, 96, 4, 2, 4)  /* as 13, 112 KiB */

template <int EPI, int TM, int TN, int WM, int WN, int S>
void launch_mid_cfg(hipStream_t s, const PcyGemmArgs& a0) {
  constexpr int smem = S * (TM + TN) * 128;
  static_assert(smem <= 160 * 1024, "LDS");
  static PcyLdsAttr lds;
  lds.ensure(&gemm_kernel_mid<EPI, TM, TN, WM, WN, S>, smem, 0);
  PcyGemmArgs a = a0;
  const int tm = (a.M + TM - 1) / TM, tn = (a.N + TN - 1) / TN;
  // rasterisation: an XCD runs a contiguous range of the tile order; groups of `gn` column tiles, all row tiles inside (tile_origin).
  // One round of tiles: what matters is that the tiles an XCD holds at once share panels -- gn ~ sqrt(tiles per XCD) columns.
  int gn = 1;
  const int per_xcd = (tm * tn + 7) / 8;
  while (gn * gn < per_xcd && gn < tn) ++gn;
  if (gn > tn) gn = tn;
  a.gn = gn < 1 ? 1 : gn;
  hipLaunchKernelGGL((gemm_kernel_mid<EPI, TM, TN, WM, WN, S>), dim3(tm * tn), dim3(WM * WN * 64), smem, s, a);
}

// cfg = one of PCY_MID_CONFIGS; false if the id is unknown or the epilogue does not fit the shape (rotary needs FN % 4 == 0)
template <int EPI>
bool launch_mid(hipStream_t s, const PcyGemmArgs& a, int cfg) {
  switch (cfg) {
#define X(ID, TM_, TN_, WM_, WN_, S_)                                                                   \
    case ID:                                                                                            \
      if (a.rope_cos != nullptr && ((TN_ / WN_ / 16) % 4 != 0)) return false;                           \
      if (EPI == EPI_SWIGLU && ((TN_ / WN_ / 16) % 2 != 0)) return false;                                \
      launch_mid_cfg<EPI, TM_, TN_, WM_, WN_, S_>(s, a);                                 \
      return true;
    PCY_MID_CONFIGS(X)
#undef X
    default: return false;
  }
}
