// Persistent decode-step kernel for batch 1 (row A7 of SURVEY.md section 8a; the layer loop of HF LlamaModel.forward
// behind procyon/model/pmc_llama.py:571-588 with one new token and the KV cache).  OPT-IN (PCY_DECODE_FUSED=1): correct
// and deterministic, but on MI355X it does not beat the layer-by-layer launches yet (3.41 vs 3.38 ms/token) -- see the
// measurements at the end of this comment.
//
// Idea: the layer-by-layer path spends ~5 us of every GEMV launch on ramp-up and drain (4 launches x 32 layers) and the
// weight stream stops during the attention.  Here ONE launch runs the whole step:
//
//   * grid = one 512-thread workgroup per CU.  Workgroups [0, n_stream) stream weights, the last n_attn do the decode
//     attention of every layer (same device code as the stand-alone attention kernel).
//   * a streaming workgroup = 7 streaming waves + 1 service wave.
//       - streaming waves only ever issue weight loads (16-byte non-temporal, NSET register sets x 4 KiB per wave) and
//         read x from LDS.  Their load queue runs NSET batches AHEAD of the arithmetic across phase and layer
//         boundaries -- weights do not depend on activations -- so vmcnt never has to drain (vector loads return in
//         order: a wave that also loaded activations or waited on its own stores would stall behind its look-ahead).
//         Row sums go to LDS.
//       - the service wave owns everything that depends on other workgroups: it waits for the producers of the next
//         phase's input (per-workgroup progress flags), stages x (RMSNorm fused) into LDS, applies the epilogue
//         (residual add / SwiGLU, reference rounding points) to the row sums of its workgroup, stores them and
//         publishes the workgroup's progress.
//   * phases per layer: qkv (RMS ln1) -> [attention workgroups] -> o (+residual) -> gate/up (RMS ln2, SwiGLU) -> down
//     (+residual); finally lm_head (final RMSNorm).  Rows of a phase are split contiguously over the streaming
//     workgroups; inside a workgroup the 2048-k batches of its rows are dealt evenly to the 7 waves.
//   * per-lane accumulation order is the one of gemv_stream_kernel (lane l sums k = (c*64 + l)*8.. over chunks c, then the
//     xor-shuffle tree); rows shared by two waves add two partial sums.  Results are deterministic and agree with the
//     layered path to the bf16 noise floor (not bitwise: RMS statistic and shared rows sum in a different order).
//
// Measured on MI355X, Llama-3-8B geometry, T = 512 (tools/check_fused.py, tools/fused_trace.py; PCY_FUSED_TRACE):
//   * with every cross-workgroup wait disabled (PCY_FUSED_NOWAIT=1, wrong results) the step takes 2.43 ms = 6.4 TB/s:
//     the streaming structure itself reaches the HBM rate of the best stand-alone GEMV.
//   * a hand-over of a phase's output vector to all workgroups costs 2.5 - 4 us (store written through -> flag -> poll ->
//     load -> RMSNorm -> LDS), 17 us through the attention; 4 + 1 per layer = 27 us against 69 us of streaming.
//     Release/acquire fences (buffer_wbl2 / buffer_inv) cost 4 - 7 us per hand-over, tagged-word polling of the data
//     itself 3.3 - 4.2 us; plain flags + agent-scope (sc1) loads and stores are the fastest of the three.
//   * the look-ahead does NOT hide those hand-overs: injecting d us of extra latency per hand-over lengthens the step by
//     0.85 d (PCY_FUSED_NOWAIT=1+10d).  A CU keeps only ~32 KiB of HBM reads in flight whatever the number of issued
//     loads (8 vs 12 register sets: 3.41 vs 3.31 ms), i.e. ~1.2 us of traffic chip-wide; the rest of the queue waits in
//     the CU.  A kernel boundary costs about the same as a hand-over, hence no gain over the layered path.
//   * next: hand over the output vector in two halves and order each phase (out-half, k-half) so that the second half
//     is needed a quarter into the next phase (hides the o->gate/up, gate/up->down and most of down->qkv hand-overs).
//
// Every spin loop carries a watchdog: a stuck dependency sets *err and the kernel runs to completion.
#include <stdlib.h>
#include "pcy_common.h"
#include "pcy_internal.h"
#include "pcy_attn_dec.h"

namespace {

constexpr int FD_NT = 512;
constexpr int FD_SW = 7;            // streaming waves per streaming workgroup (wave 7 = service)
#ifndef PCY_FUSED_NSET
#define PCY_FUSED_NSET 12
#endif
constexpr int NSET = PCY_FUSED_NSET, CH = 4;     // register sets in flight per wave x 16-byte chunks per lane per set
constexpr int BKE = CH * 512;       // k elements per batch: one set = 2048 consecutive k of one weight row
constexpr int TAB_BYTES = 4096;     // LDS copy of the per-layer pointer table (<= 85 layers)
constexpr int OUT_SLOTS = 1024;     // row sums of one workgroup and phase
constexpr unsigned SPIN_GLOBAL = 1u << 18, SPIN_LDS = 1u << 22;
enum { PH_QKV = 0, PH_O = 1, PH_GU = 2, PH_DOWN = 3, PH_LM = 4 };

#define LDS __attribute__((address_space(3)))
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

#define GLB __attribute__((address_space(1)))
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
// pointers that went through LDS / integer casts lose their address space: say "global" explicitly, a flat load would
// count on lgkmcnt AND vmcnt
__device__ __forceinline__ uint4 ldg16_nt(const bf16_t* p) {
  const u32x4_t v = __builtin_nontemporal_load((GLB const u32x4_t*)p);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 ldg16(const bf16_t* p) {
  const u32x4_t v = *(GLB const u32x4_t*)p;
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint4 lds_ld16(LDS const void* p) {
  const u32x4_t v = *reinterpret_cast<LDS const u32x4_t*>(p);
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_st16(LDS void* p, const uint4& v) {
  const u32x4_t t = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<LDS u32x4_t*>(p) = t;
}

__device__ __forceinline__ const bf16_t* lds_ptr(LDS const char* slot) {
  const u32x2_t v = *reinterpret_cast<LDS const u32x2_t*>(slot);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(v[0]), hi = __builtin_amdgcn_readfirstlane(v[1]);
  return reinterpret_cast<const bf16_t*>(((uint64_t)hi << 32) | lo);
}

struct Geo {
  int P, L;
  LDS const char* tab;   // [L][6] per-layer pointers, then [L*6] = lm_head
  LDS const int* rng;    // [8][5 kinds][4]: waves 0..6 = {first unit, first part, first chunk group, batches}; [7] = {n_lo, n_hi}
  LDS const int* kk;     // [5] reduction length of each phase kind
};
__device__ __forceinline__ int lds_int(LDS const int* p) { return __builtin_amdgcn_readfirstlane(*p); }

// One wave's position in its own batch sequence (all fields wave-uniform).  A batch = 2048 consecutive k of one weight
// row.  The batches of a workgroup's rows are dealt out contiguously and EVENLY to its 7 streaming waves, so a row may
// be shared by two neighbouring waves (partial sums meet in LDS): with whole rows per wave the waves of a phase
// differ by a row (3 vs 4 rows of `down` = 33 % longer), and the phase lasts as long as its slowest wave.
struct It {
  int p, kind, cpb, K;
  const bf16_t* W;
  int u, wg_lo, part, cg, left;
  const bf16_t* rowp;
};

__device__ __forceinline__ int unit_lo(int w, int U, int nsw) { return (int)(((long long)w * U) / nsw); }

__device__ __forceinline__ void set_row(It& it) {
  const int r = it.kind == PH_GU ? ((it.u >> 4) * 32 + (it.u & 15) + it.part * 16) : it.u;
  it.rowp = it.W + (size_t)r * it.K;
}

// `wave` = 0..6 streaming wave
__device__ __forceinline__ void enter_phase(It& it, const Geo& g, int p, int wave) {
  it.p = p;
  if (p >= g.P) return;
  const bool lm = p == 4 * g.L;
  it.kind = lm ? PH_LM : (p & 3);
  it.W = lds_ptr(g.tab + (lm ? g.L * 6 : (p >> 2) * 6 + it.kind) * 8);
  it.K = lds_int(g.kk + it.kind);
  it.cpb = it.K >> 11;
  LDS const int* r = g.rng + (wave * 5 + it.kind) * 4;
  it.u = lds_int(r); it.part = lds_int(r + 1); it.cg = lds_int(r + 2); it.left = lds_int(r + 3);
  it.wg_lo = lds_int(g.rng + (7 * 5 + it.kind) * 4);
  set_row(it);
}

// returns true when the batch just left was the last one of this wave in its phase
__device__ __forceinline__ bool advance(It& it, const Geo& g, int wave) {
  if (--it.left == 0) { enter_phase(it, g, it.p + 1, wave); return true; }
  if (++it.cg < it.cpb) return false;
  it.cg = 0;
  if (it.kind == PH_GU && it.part == 0) { it.part = 1; set_row(it); return false; }
  it.part = 0;
  ++it.u;
  set_row(it);
  return false;
}

__device__ __forceinline__ void flag_error(unsigned* err) {
  __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-wide wait until flags[0..n) >= target (agent-scope loads; no cache invalidate: all data read afterwards is
// fetched with agent-scope loads as well)
__device__ __forceinline__ void wait_flags(unsigned* flags, int n, unsigned target, int lane, bool& dead, unsigned* err) {
  if (dead) return;
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
    for (int i = lane; i < n; i += 64)
      ok = ok && (__hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target);
    if (rfl(__all(ok))) break;
    __builtin_amdgcn_s_sleep(2);
    if (++spins > SPIN_GLOBAL) { dead = true; if (lane == 0) flag_error(err); break; }
  }
}

// wave-uniform spin on an LDS word
__device__ __forceinline__ void lds_wait_ge(LDS int* p, int target, bool& dead, unsigned* err) {
  if (dead) return;
  unsigned spins = 0;
  while (rfl(*reinterpret_cast<volatile LDS int*>(p)) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > SPIN_LDS) { dead = true; flag_error(err); break; }
  }
}

// Cross-workgroup vectors (residual stream, qkv, attention output, act) are plain bf16 arrays written THROUGH to memory
// with agent-scope stores and read with agent-scope loads, guarded by per-workgroup progress flags: producer = stores,
// s_waitcnt vmcnt(0), flag store; consumer = poll the flags, then load.  No L2 write-back / invalidate (measured
// 4-7 us per hand-over with release/acquire fences, ~1.5 us this way); tagged-word polling of the data itself
// (no flag hop) measured slower -- 192 workgroups polling 16 KiB each compete with the stores they wait for.
// 4 x 16-byte agent-scope loads (4 consecutive 1-KiB rows of a wave's 64 x 16 B footprint), NOT waited for
__device__ __forceinline__ void ld4_sc1(const void* p, u32x4_t& r0, u32x4_t& r1, u32x4_t& r2, u32x4_t& r3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
      "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
      "global_load_dwordx4 %3, %4, off offset:3072 sc1"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(p)
      : "memory");
}
// ... and the wait: every register of the batch is re-defined after the s_waitcnt so that no use can move above it
template <int N>
__device__ __forceinline__ void wait_sc1(u32x4_t (&r)[N]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(r[i]));
}
__device__ __forceinline__ void st_sc1(bf16_t* p, bf16_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int DH, int G>
__device__ __forceinline__ void attn_role(const PcyFusedDecArgs& a, char* role) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int awg = blockIdx.x - a.n_stream;
  constexpr int slices = DH / 16;
  const int units = a.B * a.Hkv * slices;
  const int qkvw = (a.H + 2 * a.Hkv) * DH;
  bf16_t* stage = reinterpret_cast<bf16_t*>(role + attn_dec_smem_bytes(G, 16, DH, a.Tmax));   // [qkvw] this kv head's q/k/v
  bool dead = a.nowait != 0;
  PcyDecAttnArgs t{};
  t.qkv = stage; t.ld = qkvw; t.o = a.ao; t.ldo = a.H * DH; t.pos_dev = a.pos_dev;
  t.cos_t = a.cos_t; t.sin_t = a.sin_t; t.keep = a.keep; t.ld_keep = a.ld_keep; t.scratch = nullptr;
  t.B = a.B; t.H = a.H; t.Hkv = a.Hkv; t.dh = DH; t.Tmax = a.Tmax; t.scale = a.scale; t.dbg = 0;
  t.t_plus1 = *a.pos_dev + 1; t.o_sc1 = 1;
  for (int l = 0; l < a.L; ++l) {
    t.kcache = a.kcache + (size_t)l * a.layer_stride;
    t.vcache = a.vcache + (size_t)l * a.layer_stride;
    for (int unit = awg; unit < units; unit += a.n_attn) {
      const int sl = unit % slices, kvh = (unit / slices) % a.Hkv;
      attn_dec_body<DH, G, 16>(t, role, sl, kvh, 0, [&]() {
        // cache rows are on their way; now the qkv phase of layer l (flags), then this kv head's (G+2) x DH values
        if (wave == 0) wait_flags(a.flags, a.n_stream, 4 * l + 1, lane, dead, a.err);
        __syncthreads();
        if (a.trace && tid == 0) a.trace[(size_t)a.n_stream * (4 * a.L + 1) * 4 + (awg * a.L + l) * 2] = wall_clock64();
        if (tid < (G + 2) * DH / 8) {
          const int seg = tid / (DH / 8), c = (tid % (DH / 8)) * 8;   // seg < G: q heads; G: k; G+1: v
          const int col = seg < G ? (kvh * G + seg) * DH + c : (seg == G ? (a.H + kvh) * DH + c : (a.H + a.Hkv + kvh) * DH + c);
          u32x4_t v;
          asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(a.qkv + col) : "memory");
          *reinterpret_cast<u32x4_t*>(stage + col) = v;
        }
        __syncthreads();
      });
      // every wave: its stores are written through; then the workgroup's flag
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (tid == 0) __hip_atomic_store(a.flags + PCY_FUSED_NFLAGS + awg, (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.trace && tid == 0) a.trace[(size_t)a.n_stream * (4 * a.L + 1) * 4 + (awg * a.L + l) * 2 + 1] = wall_clock64();
  }
}

__device__ __forceinline__ void stream_role(unsigned* flags, const Geo& g, LDS const bf16_t* xs, LDS float* outbuf, LDS int* ctrl, int wave) {
  const int lane = threadIdx.x & 63;
  const int P = g.P;
  LDS int* xready = ctrl;
  LDS int* done = ctrl + 1;
  bool dead = false;
  It is, co;
  enter_phase(is, g, 0, wave);
  enter_phase(co, g, 0, wave);
  const bf16_t* dummy = is.W;
  bool co_first = true;
  float acc = 0.f;
  uint4 w[NSET][CH];

  auto issue = [&](uint4 (&wv)[CH]) {
    const bf16_t* src = (is.p < P ? is.rowp + is.cg * BKE : dummy) + lane * 8;   // exhausted: harmless re-read
#pragma unroll
    for (int c = 0; c < CH; ++c) wv[c] = ldg16_nt(src + c * 512);
    if (is.p < P) advance(is, g, wave);
  };
  auto consume = [&](const uint4 (&wv)[CH]) {
    if (co.p >= P) return;
    if (co_first) {
      lds_wait_ge(xready, co.p + 1, dead, flags);
      asm volatile("" ::: "memory");
      co_first = false;
    }
    LDS const bf16_t* xp = xs + co.cg * BKE + lane * 8;
#pragma unroll
    for (int c = 0; c < CH; ++c) acc = dot8(wv[c], lds_ld16(xp + c * 512), acc);
    if (co.cg == co.cpb - 1 || co.left == 1) {
      // end of the row, or of this wave's share of it: a row has at most two contributors, so the order of the two
      // additions onto the zeroed slot does not matter (a + b == b + a)
      const float v = wave_sum(acc);
      acc = 0.f;
      if (lane == 0)
        __hip_atomic_fetch_add(outbuf + (co.u - co.wg_lo) * (co.kind == PH_GU ? 2 : 1) + co.part, v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (advance(co, g, wave)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      co_first = true;
    }
  };

#pragma unroll
  for (int s = 0; s < NSET; ++s) issue(w[s]);
  while (co.p < P) {
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
      consume(w[s]);
      issue(w[s]);
    }
  }
}

__device__ __forceinline__ void service_role(const PcyFusedDecArgs& a, const Geo& g, LDS bf16_t* xs, LDS float* outbuf, LDS int* ctrl) {
  const int lane = threadIdx.x & 63;
  const int wg = blockIdx.x, P = g.P;
  LDS int* xready = ctrl;
  LDS int* done = ctrl + 1;
  bool dead = a.nowait != 0;
  // residual-stream rows owned by this workgroup stay in a register across the layers (lane i <-> row n_lo + i)
  float myres = 0.f;
  {
    const int lo = lds_int(g.rng + (7 * 5 + PH_O) * 4), hi = lds_int(g.rng + (7 * 5 + PH_O) * 4 + 1);
    if (lo + lane < hi) myres = bf2f(a.x[lo + lane]);
  }
  for (int p = 0; p < P; ++p) {
    const int kind = p == 4 * g.L ? PH_LM : (p & 3);
    const int K = lds_int(g.kk + kind);
    const int l = p >> 2;
    const int n_lo = lds_int(g.rng + (7 * 5 + kind) * 4), n_hi = lds_int(g.rng + (7 * 5 + kind) * 4 + 1);   // units of this workgroup
    const bool tr = a.trace && lane == 0;
    // (a) this phase's input vector -> LDS: flags of its producers, then agent-scope loads
    const bool rms = kind == PH_QKV || kind == PH_GU || kind == PH_LM;
    const bf16_t* gw_ = kind == PH_LM ? a.final_norm : lds_ptr(g.tab + (l * 6 + (kind == PH_QKV ? 4 : 5)) * 8);
    const int n = K >> 9;   // 1-KiB rows of 512 bf16; lane l holds elements (i*64 + l)*8 .. +7 of row i
    u32x4_t gv[8];
    if (rms) {
      // gains are constants: requested before the wait (d <= 4096 -> n <= 8)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < n) gv[i] = *(GLB const u32x4_t*)(gw_ + (i * 64 + lane) * 8);
    }
    if (p > 0) {
      if (kind == PH_O) wait_flags(a.flags + PCY_FUSED_NFLAGS, a.n_attn, l + 1, lane, dead, a.err);
      else wait_flags(a.flags, a.n_stream, p, lane, dead, a.err);
    }
    if (tr) a.trace[(wg * P + p) * 4 + 0] = wall_clock64();
    const bf16_t* src = kind == PH_O ? a.ao : (kind == PH_DOWN ? a.act : (p == 0 ? a.x : a.xres));
    if (!rms) {
      for (int base = 0; base < n; base += 16) {
        u32x4_t v[16];
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 4)
          if (base + i0 < n) ld4_sc1(src + (size_t)((base + i0) * 64 + lane) * 8, v[i0], v[i0 + 1], v[i0 + 2], v[i0 + 3]);
        wait_sc1(v);
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (base + i < n) *reinterpret_cast<LDS u32x4_t*>(xs + (base + i) * 512 + lane * 8) = v[i];
      }
    } else {
      u32x4_t v[8];
#pragma unroll
      for (int i0 = 0; i0 < 8; i0 += 4)
        if (i0 < n) ld4_sc1(src + (size_t)(i0 * 64 + lane) * 8, v[i0], v[i0 + 1], v[i0 + 2], v[i0 + 3]);
      wait_sc1(v);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < n) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float f0 = lo_bf(v[i][j]), f1 = hi_bf(v[i][j]); ss += f0 * f0 + f1 * f1; }
        }
      ss = wave_sum(ss);
      const float rstd = rsqrtf(ss / (float)K + a.rms_eps);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < n) {
          u32x4_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x0 = lo_bf(v[i][j]) * rstd, x1 = hi_bf(v[i][j]) * rstd;
            if (a.rms_cast == 0) { x0 = rbf(x0); x1 = rbf(x1); }
            o[j] = pack_bf(lo_bf(gv[i][j]) * x0, hi_bf(gv[i][j]) * x1);
          }
          *reinterpret_cast<LDS u32x4_t*>(xs + i * 512 + lane * 8) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (a.nowait > 1) {   // debug: artificial dependency latency of (nowait-1) x 0.1 us
      const unsigned long long t0 = wall_clock64();
      while (wall_clock64() - t0 < (unsigned long long)(a.nowait - 1) * 10) __builtin_amdgcn_s_sleep(1);
    }
    *reinterpret_cast<volatile LDS int*>(xready) = p + 1;
    if (tr) a.trace[(wg * P + p) * 4 + 1] = wall_clock64();
    // (b) row sums of the streaming waves
    { bool d2 = false; lds_wait_ge(done, FD_SW * (p + 1), d2, a.err); }
    asm volatile("" ::: "memory");
    if (tr) a.trace[(wg * P + p) * 4 + 2] = wall_clock64();
    // (c) epilogue (reference rounding points), written through; then this workgroup's progress flag
    if (kind == PH_O || kind == PH_DOWN) {
      if (n_lo + lane < n_hi) {
        myres = rbf(rbf(outbuf[lane]) + myres);
        st_sc1(a.xres + n_lo + lane, f2bf(myres));
      }
    } else if (kind == PH_GU) {
      for (int i = n_lo + lane; i < n_hi; i += 64) {
        const int s = i - n_lo;
        const float gt = rbf(outbuf[2 * s]), up = rbf(outbuf[2 * s + 1]);
        st_sc1(a.act + i, f2bf(rbf(silu_f(gt)) * up));
      }
    } else if (kind == PH_QKV) {
      for (int i = n_lo + lane; i < n_hi; i += 64) st_sc1(a.qkv + i, f2bf(outbuf[i - n_lo]));
    } else {
      for (int i = n_lo + lane; i < n_hi; i += 64) a.logits[i] = f2bf(outbuf[i - n_lo]);
    }
    if (kind != PH_LM) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(a.flags + wg, (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // row-sum slots of the next phase start from zero (its streaming waves wait for x_ready, set after this)
    for (int i = lane; i < OUT_SLOTS; i += 64) outbuf[i] = 0.f;
    if (tr) a.trace[(wg * P + p) * 4 + 3] = wall_clock64();
  }
}

template <int DH, int G>
__global__ __launch_bounds__(FD_NT) void decode_fused_kernel(PcyFusedDecArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  // per-layer pointer table -> LDS (scalar-ised reads from there; no vector loads in the streaming waves' loop)
  for (int i = tid; i < a.L * 6 * 2; i += FD_NT)
    reinterpret_cast<uint32_t*>(smem)[i] = reinterpret_cast<const uint32_t*>(a.layers)[i];
  if (tid == 0) reinterpret_cast<const bf16_t**>(smem)[a.L * 6] = a.lm_head;
  char* role = smem + TAB_BYTES;
  if (wg >= a.n_stream) { __syncthreads(); attn_role<DH, G>(a, role); return; }

  LDS char* lrole = (LDS char*)role;
  LDS bf16_t* xs = reinterpret_cast<LDS bf16_t*>(lrole);                         // [Kmax] staged input of the phase
  LDS float* outbuf = reinterpret_cast<LDS float*>(lrole + a.F * 2);             // [OUT_SLOTS] row sums of this workgroup
  LDS int* ctrl = reinterpret_cast<LDS int*>(lrole + a.F * 2 + OUT_SLOTS * 4);   // [0] x_ready [1] done [16..) unit ranges
  LDS int* rng = ctrl + 16;
  LDS int* kk = ctrl + 8;
  if (tid < 2) ctrl[tid] = 0;
  if (tid < 5) kk[tid] = tid == PH_DOWN ? a.F : a.d;
  for (int i = tid; i < OUT_SLOTS; i += FD_NT) outbuf[i] = 0.f;
  if (tid < 8 * 5) {
    // units of this workgroup (contiguous share of the phase's rows), then an even, contiguous deal of their batches
    // to the 7 streaming waves
    const int w = tid / 5, kind = tid % 5;
    const int U = kind == PH_LM ? a.vocab : (kind == PH_QKV ? (a.H + 2 * a.Hkv) * DH : (kind == PH_GU ? a.F : a.d));
    const int n_lo = unit_lo(wg, U, a.n_stream), n_hi = unit_lo(wg + 1, U, a.n_stream);
    const int cpb = (kind == PH_DOWN ? a.F : a.d) >> 11, ru = kind == PH_GU ? 2 : 1;
    if (w == 7) {
      rng[tid * 4] = n_lo; rng[tid * 4 + 1] = n_hi;
    } else {
      const int nb = (n_hi - n_lo) * ru * cpb;
      const int b0 = w * nb / FD_SW, b1 = (w + 1) * nb / FD_SW;
      rng[tid * 4] = n_lo + b0 / (ru * cpb);
      rng[tid * 4 + 1] = (b0 / cpb) % ru;
      rng[tid * 4 + 2] = b0 % cpb;
      rng[tid * 4 + 3] = b1 - b0;
    }
  }
  __syncthreads();
  Geo g;
  g.P = 4 * a.L + 1; g.L = a.L; g.tab = (LDS const char*)smem; g.rng = rng; g.kk = kk;
  if (wave < FD_SW) stream_role(a.err, g, xs, outbuf, ctrl, wave);
  else service_role(a, g, xs, outbuf, ctrl);
}

}  // namespace

bool pcy_fused_decode_supported(const PcyFusedDecArgs& a) {
  const int qkvw = (a.H + 2 * a.Hkv) * a.dh;
  const int nsw = a.n_stream * FD_SW;
  if (a.B != 1 || a.dh != 128 || a.H != 4 * a.Hkv) return false;
  if (a.d % BKE || a.F % BKE || a.H * a.dh != a.d || a.d > 4096) return false;
  if (a.L * 48 + 8 > TAB_BYTES) return false;
  // every wave needs at least a row's worth of batches per phase (a row is then shared by at most two waves)
  if (a.d / a.n_stream < FD_SW || qkvw / a.n_stream < FD_SW) return false;
  if (a.vocab / a.n_stream * (a.d / BKE) < FD_SW) return false;   // ... also of the lm_head rows
  if ((a.d + a.n_stream - 1) / a.n_stream + 1 > 64 || a.d % 1024) return false;   // residual rows of a workgroup live in one wave
  if ((a.vocab + a.n_stream - 1) / a.n_stream + 1 > OUT_SLOTS || 2 * ((a.F + a.n_stream - 1) / a.n_stream + 1) > OUT_SLOTS) return false;
  return a.n_stream > 0 && a.n_attn > 0;
}

size_t pcy_fused_decode_words(const PcyFusedDecArgs& a) {   // flags, then bf16 xres[d] qkv[qkvw] ao[d] act[F]
  return (size_t)2 * PCY_FUSED_NFLAGS + ((size_t)2 * a.d + (size_t)(a.H + 2 * a.Hkv) * a.dh + (size_t)a.F) / 2 + 64;
}

size_t pcy_fused_decode_smem(const PcyFusedDecArgs& a) {
  const size_t stream = (size_t)a.F * 2 + OUT_SLOTS * 4 + 64 + 8 * 5 * 4 * 4 + 64;
  const size_t attn = attn_dec_smem_bytes(4, 16, 128, a.Tmax) + (size_t)(a.H + 2 * a.Hkv) * a.dh * 2 + 64;
  return TAB_BYTES + (stream > attn ? stream : attn);
}

void pcy_launch_decode_fused(hipStream_t s, const PcyFusedDecArgs& a) {
  const size_t smem = pcy_fused_decode_smem(a);
  static size_t configured = 0;
  if (smem > 65536 && smem > configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_fused_kernel<128, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = smem;
  }
  hipLaunchKernelGGL((decode_fused_kernel<128, 4>), dim3(a.n_stream + a.n_attn), dim3(FD_NT), smem, s, a);
}
