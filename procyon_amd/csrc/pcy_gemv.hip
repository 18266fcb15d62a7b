// Decode-step weight-streaming GEMV for gfx950 (row A7 of SURVEY.md section 8a).
//
//   y[b][n] = epilogue( sum_k x[b][k] * W[n][k] ),  b < B <= 8, W in nn.Linear layout [N,K].
//
// HBM-bound: every weight byte is read exactly once per step (15.0 GB/token for Llama-3-8B).
// Design:
//   * 256-thread workgroup = 4 waves; each wave owns 4 output rows (8 weight rows for SwiGLU)
//     and streams them with 16-byte non-temporal loads, 64 lanes x 16 B = 1 KiB contiguous
//     per row per instruction (perfect coalescing), all loads of a K-pass issued before use.
//   * x is staged once per workgroup in LDS as bf16 (optionally RMS-normalised in the prologue
//     with the reference's rounding points) and re-read with ds_read_b128 -- LDS traffic is
//     1/rows of the HBM traffic.
//   * v_dot2c_f32_bf16 accumulates in fp32; one xor-shuffle tree per output at the end.
//   * epilogues reproduce the reference's bf16 materialisation points (residual add, SwiGLU,
//     bias+GELU), see pcy_common.h.
#include <stdlib.h>
#include "pcy_internal.h"
#include "pcy_handover.h"
#include "pcy_mlp_chain.h"

namespace {

#ifndef PCY_GEMV_DIRECTX
#define PCY_GEMV_DIRECTX 0  // reading x straight from L2 in the k-loop measured +2.3 us per o/down launch in situ
#endif
constexpr int GEMV_THREADS = 256;
constexpr int GEMV_WAVES = 4;
constexpr int XS_BYTES_MAX = 65536;

// RW = weight rows per wave (4, or 8 = 4 gate + 4 up for SwiGLU); UN = K-iterations in flight
template <int NB, int EPI, bool RMS, int UN>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(PcyGemvArgs a, int KC) {
  constexpr int RW = (EPI == EPI_SWIGLU) ? 8 : 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);              // [NB][KC]
  float* red = reinterpret_cast<float*>(smem + (size_t)NB * KC * 2);  // [GEMV_WAVES]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = a.K;

  // rows owned by this wave
  int rows[RW];
  bool rvalid[RW];
  if (EPI == EPI_SWIGLU) {
    const int f0 = blockIdx.x * 16 + wave * 4;  // first of 4 features; gate/up interleaved by 16 rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = f0 + i;
      rvalid[i] = rvalid[i + 4] = f < a.N;
      const int fc = f < a.N ? f : a.N - 1;
      rows[i] = (fc >> 4) * 32 + (fc & 15);
      rows[i + 4] = rows[i] + 16;
    }
  } else {
    const int r0 = blockIdx.x * 16 + wave * 4;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      rvalid[i] = (r0 + i) < a.N;
      rows[i] = rvalid[i] ? r0 + i : a.N - 1;
    }
  }
  const bf16_t* wrow[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) wrow[i] = a.W + (size_t)rows[i] * K;

  float acc[RW][NB];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[i][b] = 0.f;

  float rstd[NB];
  if (RMS) {
    // sum of squares over the whole row (RMS implies KC == K)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float ss = 0.f;
      for (int k = threadIdx.x * 8; k < K; k += GEMV_THREADS * 8) {
        uint4 v = *reinterpret_cast<const uint4*>(a.x + (size_t)b * a.ldx + k);
        float f;
        f = lo_bf(v.x); ss += f * f; f = hi_bf(v.x); ss += f * f;
        f = lo_bf(v.y); ss += f * f; f = hi_bf(v.y); ss += f * f;
        f = lo_bf(v.z); ss += f * f; f = hi_bf(v.z); ss += f * f;
        f = lo_bf(v.w); ss += f * f; f = hi_bf(v.w); ss += f * f;
      }
      ss = block_sum<GEMV_THREADS>(ss, red);
      rstd[b] = rsqrtf(ss / (float)K + a.rms_eps);
    }
  }

  for (int kc0 = 0; kc0 < K; kc0 += KC) {
    const int kc = (K - kc0) < KC ? (K - kc0) : KC;
    __syncthreads();
    // stage x[:, kc0:kc0+kc] into LDS
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      for (int k = threadIdx.x * 8; k < kc; k += GEMV_THREADS * 8) {
        uint4 v = *reinterpret_cast<const uint4*>(a.x + (size_t)b * a.ldx + kc0 + k);
        if (RMS) {
          const uint4 g = *reinterpret_cast<const uint4*>(a.rms_w + kc0 + k);
          const uint32_t xin[4] = {v.x, v.y, v.z, v.w};
          const uint32_t gin[4] = {g.x, g.y, g.z, g.w};
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x0 = lo_bf(xin[j]) * rstd[b], x1 = hi_bf(xin[j]) * rstd[b];
            float w0 = lo_bf(gin[j]), w1 = hi_bf(gin[j]);
            if (a.rms_cast == 0) { x0 = rbf(x0); x1 = rbf(x1); }
            o[j] = pack_bf(w0 * x0, w1 * x1);
          }
          v = make_uint4(o[0], o[1], o[2], o[3]);
        }
        *reinterpret_cast<uint4*>(xs + (size_t)b * KC + k) = v;
      }
    }
    __syncthreads();
    const int nit = (kc + 511) >> 9;  // 512 elements per wave-iteration
    for (int it0 = 0; it0 < nit; it0 += UN) {
      uint4 wv[UN][RW];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int k = ((it0 + u) * 64 + lane) * 8;
        const bool ok = k < kc;
#pragma unroll
        for (int i = 0; i < RW; ++i)
          wv[u][i] = ok ? ldg_nt(wrow[i] + kc0 + k) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int k = ((it0 + u) * 64 + lane) * 8;
        if (k < kc) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const uint4 xv = *reinterpret_cast<const uint4*>(xs + (size_t)b * KC + k);
#pragma unroll
            for (int i = 0; i < RW; ++i) acc[i][b] = dot8(wv[u][i], xv, acc[i][b]);
          }
        }
      }
    }
  }

#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[i][b] = wave_sum(acc[i][b]);

  if (lane == 0) {
    if (EPI == EPI_SWIGLU) {
      const int f0 = blockIdx.x * 16 + wave * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!rvalid[i]) continue;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float g = rbf(acc[i][b]), u = rbf(acc[i + 4][b]);
          const float s = rbf(silu_f(g));
          a.y[(size_t)b * a.ldy + f0 + i] = f2bf(s * u);
        }
      }
    } else {
      const int r0 = blockIdx.x * 16 + wave * 4;
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        if (!rvalid[i]) continue;
        const int n = r0 + i;
        const float bias = a.bias ? bf2f(a.bias[n]) : 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float v = rbf(acc[i][b] + bias);
          if (EPI == EPI_RESID) v = rbf(v + bf2f(a.resid[(size_t)b * a.ldy + n]));
          if (EPI == EPI_GELU_ERF) v = rbf(gelu_erf_f(v));
          if (EPI == EPI_GELU_ESM) v = gelu_esm_chain(v);
          a.y[(size_t)b * a.ldy + n] = f2bf(v);
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Streaming variant (whole x fits LDS): persistent waves, software-pipelined weight loads.
//   * a "unit" = R output rows (R features = 2R weight rows for SwiGLU); wave w of the launch handles units
//     w, w + n_waves, ...; the host sizes the grid so that every wave gets the same number of units
//   * the first batch of weight loads is issued BEFORE the x-staging / RMSNorm prologue, and each later batch
//     (16 x 16 B per lane) is issued before the previous one is consumed (two register sets, static indexing)
template <int NB, int EPI, bool RMS, int R>
__global__ __launch_bounds__(512) void gemv_stream_kernel(PcyGemvArgs a, int units) {
  constexpr bool DIRECTX = PCY_GEMV_DIRECTX && !RMS;
  constexpr int RW = (EPI == EPI_SWIGLU) ? 2 * R : R;
  constexpr int UN = 16 / RW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* xs = reinterpret_cast<bf16_t*>(smem);                          // [NB][K]
  float* red = reinterpret_cast<float*>(smem + (size_t)NB * a.K * 2);    // [waves per block]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = a.K;
  const int nit = (K + 511) >> 9;
  const int wpb = blockDim.x >> 6;
  const int nw = gridDim.x * wpb;
  const int nthr = blockDim.x;
  const int nrows = (EPI == EPI_SWIGLU) ? 2 * a.N : a.N;

  // element offset of weight row i of unit u (clamped into the matrix)
  auto row_off = [&](int u, int i) -> size_t {
    int r;
    if (EPI == EPI_SWIGLU) {
      const int f = u * R + (i % R);
      const int fc = f < a.N ? f : a.N - 1;
      r = (fc >> 4) * 32 + (fc & 15) + (i >= R ? 16 : 0);
    } else {
      r = u * R + i;
      r = r < nrows ? r : nrows - 1;
    }
    return (size_t)r * K;
  };
  // a.krot (K % 512 == 0): the unit of output row r walks its k-iterations rotated by (r / 4) % nit -- see PcyGemvArgs::krot
  const bool krot = a.krot && (K & 511) == 0;
  auto rot = [&](int u, int it) { if (!krot) return it; const int r = it + ((u * R) >> 2) % nit; return r >= nit ? r - nit : r; };
  auto issue = [&](int u, int it0, uint4 (&wv)[UN][RW]) {
#pragma unroll
    for (int un = 0; un < UN; ++un) {
      const int k = (rot(u, it0 + un) * 64 + lane) * 8;
      const bool ok = k < K && it0 + un < nit;
#pragma unroll
      for (int i = 0; i < RW; ++i)
        wv[un][i] = ok ? (a.plain_loads ? *reinterpret_cast<const uint4*>(a.W + row_off(u, i) + k) : ldg_nt(a.W + row_off(u, i) + k))
                       : make_uint4(0, 0, 0, 0);
    }
  };
  float acc[RW][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[i][b] = 0.f;
  };
  auto compute = [&](int u, int it0, const uint4 (&wv)[UN][RW]) {
#pragma unroll
    for (int un = 0; un < UN; ++un) {
      const int k = (rot(u, it0 + un) * 64 + lane) * 8;
      if (k < K && it0 + un < nit) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const uint4 xv = DIRECTX ? *reinterpret_cast<const uint4*>(a.x + (size_t)b * a.ldx + k)
                                   : *reinterpret_cast<const uint4*>(xs + (size_t)b * K + k);
#pragma unroll
          for (int i = 0; i < RW; ++i) acc[i][b] = dot8(wv[un][i], xv, acc[i][b]);
        }
      }
    }
  };
  auto finish = [&](int u) {
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[i][b] = wave_sum(acc[i][b]);
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int n = u * R + i;
        if (n >= a.N) continue;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float v;
          if (EPI == EPI_SWIGLU) {
            const float g = rbf(acc[i][b]), up = rbf(acc[i + R][b]);
            v = rbf(silu_f(g)) * up;
          } else {
            v = rbf(acc[i][b] + (a.bias ? bf2f(a.bias[n]) : 0.f));
            if (EPI == EPI_RESID) v = rbf(v + bf2f(a.resid[(size_t)b * a.ldy + n]));
            if (EPI == EPI_GELU_ERF) v = rbf(gelu_erf_f(v));
            if (EPI == EPI_GELU_ESM) v = gelu_esm_chain(v);
          }
          a.y[(size_t)b * a.ldy + n] = f2bf(v);
        }
      }
    }
    zero_acc();
  };

  uint4 wa[UN][RW], wb[UN][RW];
  int u = blockIdx.x * wpb + wave;
  int it0 = 0;
  bool have = u < units;
  // position of the batch after (u, it0)
  auto next_pos = [&](int cu, int cit, int& nu_, int& nit_) {
    nit_ = cit + UN; nu_ = cu;
    if (nit_ >= nit) { nit_ = 0; nu_ = cu + nw; }
  };

  // Batch 1 (NB == 1, the decode step): vector loads return in order, so whatever is requested first is what the prologue
  // waits for.  x (and the norm weight) go out FIRST -- a few KiB, back after one round trip -- then TWO weight batches
  // (32 x 16 B per lane); the RMSNorm / LDS staging then runs while the weights stream, instead of waiting behind the first
  // batch and leaving the second one to be requested only after it (qkv 12.46 -> 12.22 us, down 21.8 -> 21.0 us, decode step 3.27 -> 3.25 ms).
  constexpr int MAXX = RMS ? 4 : 8;
  const bool xfirst = NB == 1 && !DIRECTX && K <= MAXX * nthr * 8 && a.plain_loads != 2;
  uint4 xr[MAXX], gr[RMS ? MAXX : 1];
  int u1 = u, it1 = 0;
  bool have1 = false;
  if (xfirst) {
#pragma unroll
    for (int i = 0; i < MAXX; ++i) {
      const int k = (threadIdx.x + i * nthr) * 8;
      if (k < K) {
        xr[i] = *reinterpret_cast<const uint4*>(a.x + k);
        if (RMS) gr[i] = *reinterpret_cast<const uint4*>(a.rms_w + k);
      }
    }
    if (have) issue(u, 0, wa);
    next_pos(u, 0, u1, it1);
    have1 = have && u1 < units;
    if (have1) issue(u1, it1, wb);
    float rs = 1.f;
    if (RMS) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < MAXX; ++i) {
        const int k = (threadIdx.x + i * nthr) * 8;
        if (k < K) {
          const uint32_t w4[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { const float f0 = lo_bf(w4[j]), f1 = hi_bf(w4[j]); ss += f0 * f0 + f1 * f1; }
        }
      }
      ss = block_sum_rt(ss, red, wpb);
      rs = rsqrtf(ss / (float)K + a.rms_eps);
    }
#pragma unroll
    for (int i = 0; i < MAXX; ++i) {
      const int k = (threadIdx.x + i * nthr) * 8;
      if (k < K) {
        uint4 v = xr[i];
        if (RMS) {
          const uint32_t xin[4] = {v.x, v.y, v.z, v.w}, gin[4] = {gr[i].x, gr[i].y, gr[i].z, gr[i].w};
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x0 = lo_bf(xin[j]) * rs, x1 = hi_bf(xin[j]) * rs;
            if (a.rms_cast == 0) { x0 = rbf(x0); x1 = rbf(x1); }
            o[j] = pack_bf(lo_bf(gin[j]) * x0, hi_bf(gin[j]) * x1);
          }
          v = make_uint4(o[0], o[1], o[2], o[3]);
        }
        *reinterpret_cast<uint4*>(xs + k) = v;
      }
    }
    __syncthreads();
    zero_acc();
    // steady state, two batches deep: batch i sits in CUR, batch i+1 is on its way into NXT; after consuming i its registers
    // take batch i+2
#define PCY_GEMV_STEP2(CUR, NXT)                                 \
  {                                                              \
    compute(u, it0, CUR);                                        \
    if (it0 + UN >= nit) finish(u);                              \
    int u2, it2;                                                 \
    next_pos(u1, it1, u2, it2);                                  \
    const bool have2 = have1 && u2 < units;                      \
    if (have2) issue(u2, it2, CUR);                              \
    u = u1; it0 = it1; have = have1;                             \
    u1 = u2; it1 = it2; have1 = have2;                           \
  }
    while (have) {
      PCY_GEMV_STEP2(wa, wb)
      if (!have) break;
      PCY_GEMV_STEP2(wb, wa)
    }
#undef PCY_GEMV_STEP2
    return;
  }
  if (have) issue(u, 0, wa);

  // ---- prologue: x (optionally RMS-normalised) -> LDS, overlapping the first weight batch ----
  float rstd[NB];
  if (RMS) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float ss = 0.f;
      for (int k = threadIdx.x * 8; k < K; k += nthr * 8) {
        const uint4 v = *reinterpret_cast<const uint4*>(a.x + (size_t)b * a.ldx + k);
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float f0 = lo_bf(w4[j]), f1 = hi_bf(w4[j]); ss += f0 * f0 + f1 * f1; }
      }
      ss = block_sum_rt(ss, red, wpb);
      rstd[b] = rsqrtf(ss / (float)K + a.rms_eps);
    }
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (DIRECTX) break;
    for (int k = threadIdx.x * 8; k < K; k += nthr * 8) {
      uint4 v = *reinterpret_cast<const uint4*>(a.x + (size_t)b * a.ldx + k);
      if (RMS) {
        const uint4 g = *reinterpret_cast<const uint4*>(a.rms_w + k);
        const uint32_t xin[4] = {v.x, v.y, v.z, v.w}, gin[4] = {g.x, g.y, g.z, g.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x0 = lo_bf(xin[j]) * rstd[b], x1 = hi_bf(xin[j]) * rstd[b];
          if (a.rms_cast == 0) { x0 = rbf(x0); x1 = rbf(x1); }
          o[j] = pack_bf(lo_bf(gin[j]) * x0, hi_bf(gin[j]) * x1);
        }
        v = make_uint4(o[0], o[1], o[2], o[3]);
      }
      *reinterpret_cast<uint4*>(xs + (size_t)b * K + k) = v;
    }
  }
  if (!DIRECTX) __syncthreads();
  zero_acc();

  // ---- steady state: issue batch i+1, consume batch i (roles of wa/wb alternate) ----
#define PCY_GEMV_STEP(CUR, NXT)                                  \
  {                                                              \
    int nit0 = it0 + UN, nu = u;                                 \
    if (nit0 >= nit) { nit0 = 0; nu = u + nw; }                  \
    const bool nhave = nu < units;                               \
    if (nhave) issue(nu, nit0, NXT);                             \
    compute(u, it0, CUR);                                        \
    if (nit0 == 0) finish(u);                                    \
    u = nu; it0 = nit0; have = nhave;                            \
  }
  while (have) {
    PCY_GEMV_STEP(wa, wb)
    if (!have) break;
    PCY_GEMV_STEP(wb, wa)
  }
#undef PCY_GEMV_STEP
}

constexpr int GEMV_MAX_WAVES = 2048;  // 256 CUs x 2 waves/SIMD x 4 SIMDs
constexpr int GEMV_CUS = 256;

// Grid shape: a multiple of the CU count with 4..8 waves per workgroup, chosen so that every CU streams the same
// number of rows (a 448-block launch on 256 CUs leaves 3/4 of the chip idle for its second half).
inline void pick_grid(int units, int& blocks, int& wpb) {
  double best = 1e30;
  blocks = GEMV_CUS; wpb = 4;
  for (int bl = GEMV_CUS; bl <= 2 * GEMV_CUS; bl += GEMV_CUS)
    for (int w = 4; w <= 8; ++w) {
      const int waves = bl * w;
      if (waves > GEMV_MAX_WAVES) continue;
      const int per = (units + waves - 1) / waves;
      const double cost = (double)per * waves / units + 1e-3 * (GEMV_MAX_WAVES - waves) / GEMV_MAX_WAVES;
      if (cost < best) { best = cost; blocks = bl; wpb = w; }
    }
  if (units < GEMV_CUS * 4) { blocks = (units + 3) / 4; wpb = 4; }
}

template <int NB, int EPI, bool RMS, int R>
void launch_stream(hipStream_t s, const PcyGemvArgs& a) {
  const int units = (a.N + R - 1) / R;
  int blocks, wpb;
  pick_grid(units, blocks, wpb);
  const size_t smem = (size_t)NB * a.K * 2 + 64;
  hipLaunchKernelGGL((gemv_stream_kernel<NB, EPI, RMS, R>), dim3(blocks), dim3(wpb * 64), smem, s, a, units);
}

template <int NB, int EPI, bool RMS>
void launch_nb(hipStream_t s, const PcyGemvArgs& a) {
  if ((size_t)NB * a.K * 2 <= XS_BYTES_MAX) {
    // few rows -> 2-row units so that the launch still spreads over ~2k waves
    if ((a.N + 3) / 4 < GEMV_MAX_WAVES) launch_stream<NB, EPI, RMS, 2>(s, a);
    else launch_stream<NB, EPI, RMS, 4>(s, a);
    return;
  }
  int KC = a.K;
  if ((size_t)NB * KC * 2 > XS_BYTES_MAX) KC = (XS_BYTES_MAX / (NB * 2)) & ~511;
  const size_t smem = (size_t)NB * KC * 2 + 64;
  const int blocks = (a.N + 15) / 16;
  constexpr int UN = (EPI == EPI_SWIGLU) ? 2 : 4;
  hipLaunchKernelGGL((gemv_kernel<NB, EPI, RMS, UN>), dim3(blocks), dim3(GEMV_THREADS), smem, s, a, KC);
}

template <int NB>
void launch_epi(hipStream_t s, const PcyGemvArgs& a) {
  const bool rms = a.rms_w != nullptr;
  switch (a.epi) {
    case EPI_STORE: rms ? launch_nb<NB, EPI_STORE, true>(s, a) : launch_nb<NB, EPI_STORE, false>(s, a); break;
    case EPI_RESID: launch_nb<NB, EPI_RESID, false>(s, a); break;
    case EPI_GELU_ERF: launch_nb<NB, EPI_GELU_ERF, false>(s, a); break;
    case EPI_GELU_ESM: launch_nb<NB, EPI_GELU_ESM, false>(s, a); break;
    case EPI_SWIGLU: rms ? launch_nb<NB, EPI_SWIGLU, true>(s, a) : launch_nb<NB, EPI_SWIGLU, false>(s, a); break;
  }
}


// ------------------------------------------------------------------------------------------------
// Batched decode (B = 5..32 rows; e.g. beam 20): skinny MFMA GEMV.  Weights are read exactly once; the batch rides on the
// MFMA N dimension.  Were every 16-row workgroup to re-read all of x from L2, x : W traffic would be 2 : 1; here a workgroup is
// 4 waves x 16 output rows (32 weight rows for SwiGLU) over the SAME K range: x travels global -> LDS once per
// workgroup in 512-k chunks (LDS-DMA, double buffered) and is shared by the four waves (x : W = 1 : 2), the vector
// loads carry weights only.  K is split over gridDim.y workgroups when N/64 alone cannot fill the chip (o, down, qkv);
// partial sums then go to an fp32 workspace and a small finishing kernel adds them in split order and applies the
// epilogue (deterministic).
typedef __attribute__((address_space(3))) void* gv_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gv_gptr_t;


template <int EPI>
__global__ __launch_bounds__(256) void gemv_splitk_finish_kernel(PcyGemvArgs a, int ksplit) {
  const size_t quads = (size_t)a.B * (a.N / 4);
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)gridDim.x * 256) {
    const int b = (int)(q / (a.N / 4)), n = (int)(q % (a.N / 4)) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.splitk_ws + (size_t)b * a.N + n);
    for (int s_ = 1; s_ < ksplit; ++s_) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(a.splitk_ws + ((size_t)s_ * a.B + b) * a.N + n);
      v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
    }
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[r] = rbf(v[r] + (a.bias ? bf2f(a.bias[n + r]) : 0.f));
      if (EPI == EPI_RESID) o[r] = rbf(o[r] + bf2f(a.resid[(size_t)b * a.ldy + n + r]));
    }
    *reinterpret_cast<uint2*>(a.y + (size_t)b * a.ldy + n) = make_uint2(pack_bf(o[0], o[1]), pack_bf(o[2], o[3]));
  }
}

// K-split finish + the RMSNorm that follows it in the decoder layer (o projection -> post-attention norm, down projection ->
// next layer's input norm / final norm), one workgroup per batch row: saves a ~6 us launch per norm on a step of ~170 us per
// layer at batch 32.  Element assignment, accumulation order and block reduction are those of gemv_splitk_finish_kernel and
// rmsnorm_kernel (pcy_elem.hip, NT = 256), so y and next_xn are bit-identical to the two separate launches.
__global__ __launch_bounds__(256) void gemv_splitk_finish_norm_kernel(PcyGemvArgs a, int ksplit) {
  __shared__ float red[4];
  constexpr int MAXI = 4;                       // N <= 8192
  const int b = blockIdx.x;
  if (a.N == 4096 && ksplit <= 4 && !a.bias) {
    // The decode step's case (d = 4096, K split <= 4): EVERY load of the launch -- 16 partial-sum quads, the residual, the norm weights --
    // is requested before the first is used.  The general loop below asks per 2048-column block and fetches the norm weights behind the
    // block reduction: three dependent memory round trips in a launch of ~6.5 us, twice per decoder layer (round 6).  Same element
    // assignment, same order of every sum: same bits.
    f32x4 part[2][2][4];
    uint2 res[2][2];
    uint4 g[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int k = threadIdx.x * 8 + it * 2048;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int n = k + h * 4;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
          part[it][h][s_] = s_ < ksplit ? *reinterpret_cast<const f32x4*>(a.splitk_ws + ((size_t)s_ * a.B + b) * a.N + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        res[it][h] = *reinterpret_cast<const uint2*>(a.resid + (size_t)b * a.ldy + n);
      }
      g[it] = *reinterpret_cast<const uint4*>(a.next_rms_w + k);
    }
    float xq[2][8];
    float ssq = 0.f;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int k = threadIdx.x * 8 + it * 2048;
      uint32_t packed[4];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 v = part[it][h][0];
#pragma unroll
        for (int s_ = 1; s_ < 4; ++s_)
          if (s_ < ksplit) { v[0] += part[it][h][s_][0]; v[1] += part[it][h][s_][1]; v[2] += part[it][h][s_][2]; v[3] += part[it][h][s_][3]; }
        const uint32_t rw[2] = {res[it][h].x, res[it][h].y};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float o = rbf(v[r] + 0.f);
          o = rbf(o + ((r & 1) ? hi_bf(rw[r >> 1]) : lo_bf(rw[r >> 1])));
          xq[it][h * 4 + r] = o;
        }
        packed[h * 2] = pack_bf(xq[it][h * 4], xq[it][h * 4 + 1]);
        packed[h * 2 + 1] = pack_bf(xq[it][h * 4 + 2], xq[it][h * 4 + 3]);
      }
      *reinterpret_cast<uint4*>(a.y + (size_t)b * a.ldy + k) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float p = xq[it][2 * j], q = xq[it][2 * j + 1]; ssq += p * p + q * q; }
    }
    ssq = block_sum<256>(ssq, red);
    const float rstd = rsqrtf(ssq / (float)a.N + a.rms_eps);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int k = threadIdx.x * 8 + it * 2048;
      const uint32_t gg[4] = {g[it].x, g[it].y, g[it].z, g[it].w};
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float p = xq[it][2 * j] * rstd, q = xq[it][2 * j + 1] * rstd;
        if (a.rms_cast == 0) { p = rbf(p); q = rbf(q); }
        o[j] = pack_bf(lo_bf(gg[j]) * p, hi_bf(gg[j]) * q);
      }
      *reinterpret_cast<uint4*>(a.next_xn + (size_t)b * a.N + k) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    return;
  }
  float xv[MAXI][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int k = threadIdx.x * 8 + it * 2048;
    if (k >= a.N) break;
    uint32_t packed[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = k + h * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(a.splitk_ws + (size_t)b * a.N + n);
      for (int s_ = 1; s_ < ksplit; ++s_) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(a.splitk_ws + ((size_t)s_ * a.B + b) * a.N + n);
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float o = rbf(v[r] + (a.bias ? bf2f(a.bias[n + r]) : 0.f));
        o = rbf(o + bf2f(a.resid[(size_t)b * a.ldy + n + r]));
        xv[it][h * 4 + r] = o;
      }
      packed[h * 2] = pack_bf(xv[it][h * 4], xv[it][h * 4 + 1]);
      packed[h * 2 + 1] = pack_bf(xv[it][h * 4 + 2], xv[it][h * 4 + 3]);
    }
    *reinterpret_cast<uint4*>(a.y + (size_t)b * a.ldy + k) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float p = xv[it][2 * j], q = xv[it][2 * j + 1]; ss += p * p + q * q; }
  }
  ss = block_sum<256>(ss, red);
  const float rstd = rsqrtf(ss / (float)a.N + a.rms_eps);
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int k = threadIdx.x * 8 + it * 2048;
    if (k >= a.N) break;
    const uint4 g = *reinterpret_cast<const uint4*>(a.next_rms_w + k);
    const uint32_t gg[4] = {g.x, g.y, g.z, g.w};
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float p = xv[it][2 * j] * rstd, q = xv[it][2 * j + 1] * rstd;
      if (a.rms_cast == 0) { p = rbf(p); q = rbf(q); }
      o[j] = pack_bf(lo_bf(gg[j]) * p, hi_bf(gg[j]) * q);
    }
    *reinterpret_cast<uint4*>(a.next_xn + (size_t)b * a.N + k) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// Epilogue of the batched MFMA GEMVs: lane holds D[n = fq*4 + r][b = fr] of every (row tile, batch tile)
template <int EPI, int RT, int BT>
__device__ __forceinline__ void mfma_gemv_epilogue(const PcyGemvArgs& a, f32x4 (&acc)[RT][BT], int r0, int nrows, int ksplit, int fr, int fq,
                                                   int split = blockIdx.y) {
  if (r0 >= nrows) return;
  if (ksplit > 1) {
    float* ws = a.splitk_ws + (size_t)split * a.B * a.N;
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      const int b = bt * 16 + fr;
      if (b >= a.B) continue;
      const int n = r0 + fq * 4;
      if (n + 3 < a.N) *reinterpret_cast<f32x4*>(ws + (size_t)b * a.N + n) = acc[0][bt];
      else
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) ws[(size_t)b * a.N + n + r] = acc[0][bt][r];
    }
    return;
  }
#pragma unroll
  for (int bt = 0; bt < BT; ++bt) {
    const int b = bt * 16 + fr;
    if (b >= a.B) continue;
    if (EPI == EPI_SWIGLU) {
      const int f = (r0 >> 5) * 16 + fq * 4;
      if (f >= a.N) continue;
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = rbf(silu_f(rbf(acc[0][bt][r]))) * rbf(acc[RT - 1][bt][r]);
      *reinterpret_cast<uint2*>(a.y + (size_t)b * a.ldy + f) = make_uint2(pack_bf(o[0], o[1]), pack_bf(o[2], o[3]));
    } else {
      const int n = r0 + fq * 4;
      if (n >= a.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = (n + r) < a.N ? n + r : a.N - 1;
        v[r] = rbf(acc[0][bt][r] + (a.bias ? bf2f(a.bias[nn]) : 0.f));
        if (EPI == EPI_RESID) v[r] = rbf(v[r] + bf2f(a.resid[(size_t)b * a.ldy + nn]));
        if (EPI == EPI_GELU_ERF) v[r] = rbf(gelu_erf_f(v[r]));
        if (EPI == EPI_GELU_ESM) v[r] = gelu_esm_chain(v[r]);
      }
      if (n + 3 < a.N && (a.ldy & 3) == 0) {
        *reinterpret_cast<uint2*>(a.y + (size_t)b * a.ldy + n) = make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3]));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) a.y[(size_t)b * a.ldy + n + r] = f2bf(v[r]);
      }
    }
  }
}

template <int EPI, int BT>
__global__ __launch_bounds__(256) void gemv_mfma2_kernel(PcyGemvArgs a, int ksplit) {
  constexpr int RT = (EPI == EPI_SWIGLU) ? 2 : 1;
  constexpr int KC = 512;                      // k per x chunk
  constexpr int XROW = KC * 2 + 16;            // LDS bytes per x row: 16 B of padding spreads the 16 rows of a fragment read over the banks
  constexpr int XBUF = BT * 16 * XROW;
  __shared__ __attribute__((aligned(16))) char xs[2 * XBUF];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int K = a.K;
  const int nrows = (EPI == EPI_SWIGLU) ? 2 * a.N : a.N;
  const int r0 = (blockIdx.x * 4 + wave) * 16 * RT;
  const int ks = K / ksplit;                   // multiple of KC
  const int kbeg = blockIdx.y * ks;
  const int nchunk = ks / KC;
  const int csh = pcy_gemv_kshift(blockIdx.x * 4 * 16 * RT, ks / 128) >> 2;   // rotated K order, in chunks (pcy_gemv_kshift)
  auto rk = [&](int k) { int c = (k >> 9) + csh; c = c >= nchunk ? c - nchunk : c; return (c << 9) | (k & 511); };
  const bf16_t* wp[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int r = r0 + rt * 16 + fr;
    r = r < nrows ? r : nrows - 1;
    wp[rt] = a.W + (size_t)r * K + kbeg + fq * 8;   // natural fragment layout: the 4 lanes of a row read 64 contiguous bytes per load
  }
  // x chunk c -> LDS buffer c & 1: one 1-KiB DMA piece per x row (rows beyond B repeat the last row), BT*4 pieces per wave
  auto stage_x = [&](int c) {
    char* buf = xs + (c & 1) * XBUF;
#pragma unroll
    for (int i = 0; i < BT * 4; ++i) {
      const int row = wave * BT * 4 + i;
      const int b = row < a.B ? row : a.B - 1;
      const bf16_t* src = a.x + (size_t)b * a.ldx + kbeg + rk(c * KC) + lane * 8;
      __builtin_amdgcn_global_load_lds((gv_gptr_t)src, (gv_lds_ptr_t)(buf + row * XROW), 16, 0, 0);
    }
  };
  f32x4 acc[RT][BT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) acc[rt][bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto load_w = [&](int k0, bf16x8 (&wf)[RT][4]) {   // one 128-k super-step = 4 MFMA k-steps
    const int k = rk(k0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 v = ldg_nt(wp[rt] + k + j * 32);
        wf[rt][j] = __builtin_bit_cast(bf16x8, v);
      }
  };
  auto mma = [&](int c, int sstep, const bf16x8 (&wf)[RT][4]) {
    const char* buf = xs + (c & 1) * XBUF;
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      bf16x8 xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        xf[j] = *reinterpret_cast<const bf16x8*>(buf + (bt * 16 + fr) * XROW + (sstep * 128 + j * 32 + fq * 8) * 2);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[rt][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[rt][j], xf[j], acc[rt][bt], 0, 0, 0);
    }
  };
  // Four register sets, always three 128-k super-steps (3 x 64 B per lane and row tile) of weights ahead of the MFMAs: with two
  // sets (one super-step ahead) a CU had 64 KiB of weight reads in flight against 256 KiB in the batch-1 streaming kernel and
  // the batched GEMVs ran at 4.1 TB/s against its 5.6.
  stage_x(0);
  bf16x8 w0[RT][4], w1[RT][4], w2[RT][4], w3[RT][4];
  load_w(0, w0);
  load_w(128, w1);
  load_w(256, w2);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * RT * 4) : "memory");   // the x pieces are older than the weight loads
  __builtin_amdgcn_s_barrier();
  for (int c = 0; c < nchunk; ++c) {
    const bool more = c + 1 < nchunk;
    if (more) stage_x(c + 1);
    load_w(c * KC + 384, w3);
    mma(c, 0, w0);
    if (more) load_w((c + 1) * KC, w0);
    mma(c, 1, w1);
    if (more) load_w((c + 1) * KC + 128, w1);
    mma(c, 2, w2);
    if (more) load_w((c + 1) * KC + 256, w2);
    mma(c, 3, w3);
    if (more) {
      // chunk c+1's x pieces were issued before this iteration's 4 weight batches: those may stay in flight
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * RT * 4) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  mfma_gemv_epilogue<EPI, RT, BT>(a, acc, r0, nrows, ksplit, fr, fq);
}

// The same GEMV with the weights through LDS-DMA instead of registers.  A register load of an MFMA A fragment asks for 64
// contiguous bytes per weight row (16 rows per instruction); measured with the arithmetic in place, 256 contiguous bytes per
// row (4 rows per instruction) stream 25 % faster (gate/up at batch 32: 54.3 -> 43.3 us).  The DMA can have that shape: each
// wave copies ITS 16 (32) rows of a 128-k super-step as 4 (8) instructions of 4 rows x 256 B into a private LDS tile
// [16 rows][256 B] -- the 16-byte pieces of a row XOR-swizzled by the row index on the source side, so the fragment reads
// (16 rows x one piece) touch every bank once -- S super-steps ahead, and nobody else reads the tile: a counted vmcnt orders
// the wave's own copy and read, no barrier.  x travels as before (shared, 256-k chunks).  Same MFMA order: same bits.
template <int EPI, int BT, int S>
__global__ __launch_bounds__(256) void gemv_mfma3_kernel(PcyGemvArgs a, int ksplit) {
  constexpr int RT = (EPI == EPI_SWIGLU) ? 2 : 1;
  constexpr int KC = 256;                      // k per x chunk = two super-steps
  constexpr int XROW = KC * 2 + 16;
  constexpr int XBUF = BT * 16 * XROW;
  constexpr int WT = 16 * 256;                 // one row tile of one super-step
  extern __shared__ __attribute__((aligned(1024))) char smem3[];
  char* xs = smem3 + 4 * S * RT * WT;          // [2][XBUF] behind the four waves' weight rings
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* wring = smem3 + wave * S * RT * WT;    // [S][RT][16 rows][256 B]
  const int fr = lane & 15, fq = lane >> 4;
  const int K = a.K;
  const int nrows = (EPI == EPI_SWIGLU) ? 2 * a.N : a.N;
  const int r0 = (blockIdx.x * 4 + wave) * 16 * RT;
  const int ks = K / ksplit;                   // multiple of KC
  const int kbeg = blockIdx.y * ks;
  const int nss = ks / 128, nchunk = ks / KC;
  const int ksh = pcy_gemv_kshift(blockIdx.x * 4 * 16 * RT, nss);   // rotated K order (pcy_gemv_kshift): a multiple of 4 steps, so a 256-k x chunk never wraps
  auto rss = [&](int ss) { const int r = ss + ksh; return r >= nss ? r - nss : r; };
  // DMA source of this lane: row (lane >> 4) of each group of four rows, piece (lane & 15) ^ row of the 256-byte segment
  const bf16_t* wsrc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = q * 4 + (lane >> 4);
      int r = r0 + rt * 16 + row;
      r = r < nrows ? r : nrows - 1;
      wsrc[rt][q] = a.W + (size_t)r * K + kbeg + ((lane & 15) ^ row) * 8;
    }
  auto issue_w = [&](int ss) __attribute__((always_inline)) {   // (past the end: the last super-step again, so that the counts below stay uniform)
    const int k = rss(ss < nss ? ss : nss - 1) * 128;
    char* dst = wring + (ss % S) * RT * WT;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((gv_gptr_t)(wsrc[rt][q] + k), (gv_lds_ptr_t)(dst + rt * WT + q * 1024), 16, 0, 0);
  };
  auto stage_x = [&](int c) __attribute__((always_inline)) {
    const int cc = rss(2 * (c < nchunk ? c : nchunk - 1)) >> 1;
    char* buf = xs + (c & 1) * XBUF;
#pragma unroll
    for (int i = 0; i < BT * 4; ++i) {
      const int row = wave * BT * 4 + i;
      const int b = row < a.B ? row : a.B - 1;
      const bf16_t* src = a.x + (size_t)b * a.ldx + kbeg + cc * KC + (lane & 31) * 8;   // 512 B per row: lanes 32-63 repeat (land in the padding-free tail, unused)
      if (lane < 32) __builtin_amdgcn_global_load_lds((gv_gptr_t)src, (gv_lds_ptr_t)(buf + row * XROW), 16, 0, 0);
    }
  };
  f32x4 acc[RT][BT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) acc[rt][bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma = [&](int ss) __attribute__((always_inline)) {
    const char* wb = wring + (ss % S) * RT * WT;
    const char* buf = xs + ((ss >> 1) & 1) * XBUF;
    bf16x8 wf[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[rt][j] = *reinterpret_cast<const bf16x8*>(wb + rt * WT + fr * 256 + (((j * 4 + fq) ^ fr) << 4));
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      bf16x8 xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        xf[j] = *reinterpret_cast<const bf16x8*>(buf + (bt * 16 + fr) * XROW + ((ss & 1) * 128 + j * 32 + fq * 8) * 2);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[rt][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[rt][j], xf[j], acc[rt][bt], 0, 0, 0);
    }
  };
  // issue order per wave: X(0) W(0..S-2) | per chunk c: X(c+1) W(2c+S-1) [use 2c] W(2c+S) [use 2c+1] | ...
  stage_x(0);
#pragma unroll
  for (int i = 0; i < S - 1; ++i) issue_w(i);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * RT * 4) : "memory");   // x(0) is older than the weight copies
  __builtin_amdgcn_s_barrier();
  for (int c = 0; c < nchunk; ++c) {
    stage_x(c + 1);
    issue_w(2 * c + S - 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * RT * 4 + BT * 4) : "memory");   // W(2c) has landed
    mma(2 * c);
    issue_w(2 * c + S);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 1) * RT * 4 + BT * 4) : "memory");   // W(2c+1) has landed
    mma(2 * c + 1);
    // x(c+1) has landed (two weight copies are younger) and every wave is done with x(c)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * RT * 4) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this workgroup's LDS may still be written when it retires
  mfma_gemv_epilogue<EPI, RT, BT>(a, acc, r0, nrows, ksplit, fr, fq);
}

#ifndef PCY_GEMV_ABL
#define PCY_GEMV_ABL 0   // timing ablations (tools only): 1 = no x copies, 2 = no per-step barrier, 4 = no MFMAs
#endif
// Fourth version: the same copies and the same MFMA order (same bits), a different schedule.  LDS-DMA copies retire in issue
// order (one vmcnt), so "x(c+1) has landed" in gemv_mfma3_kernel also means "every weight copy issued before it has landed": with
// x staged one chunk ahead the waves drain their weight rings to two super-steps at every chunk end, whatever S is (S = 6 measured
// like S = 3).  Here x travels exactly like the weights -- a [16 batch rows][256 B] tile per 128-k super-step, pieces XOR-swizzled
// by the row on the source side, in a ring of the SAME depth S shared by the four waves -- and x(i), W(i) are issued as a pair
// S - 1 super-steps ahead: one counted wait per super-step ("pair ss has landed": (S-2) younger pairs may be in flight), one
// barrier (everyone's part of x(ss) is there, everyone is done with the slot about to be refilled).
//   LDS: 4 waves x S x RT x 4 KB + S x BT x 4 KB  (gate/up, batch 32: S = 4 -> exactly 160 KB, 96 KB of weights in flight per CU
//   instead of 64; single-row-tile kernels: S = 6 -> 144 KB, 80 KB in flight instead of 32).
template <int EPI, int BT, int S>
__global__ __launch_bounds__(256) void gemv_mfma4_kernel(PcyGemvArgs a, int ksplit) {
  constexpr int RT = (EPI == EPI_SWIGLU) ? 2 : 1;
  constexpr int WT = 16 * 256;                 // one 16-row tile of one super-step (weights and x alike)
#ifndef PCY_GEMV_W_AUX
#define PCY_GEMV_W_AUX 2
#endif
  constexpr int W_AUX = PCY_GEMV_W_AUX;        // cache policy of the weight copies: 2 = nt (read once: batch-32 step 4.81 -> 4.68 ms), 0 = default
  extern __shared__ __attribute__((aligned(1024))) char smem4[];
  char* xs = smem4 + 4 * S * RT * WT;          // [S][BT][WT] behind the four waves' weight rings
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* wring = smem4 + wave * S * RT * WT;    // [S][RT][16 rows][256 B]
  const int fr = lane & 15, fq = lane >> 4;
  const int K = a.K;
  const int nrows = (EPI == EPI_SWIGLU) ? 2 * a.N : a.N;
  const int r0 = (blockIdx.x * 4 + wave) * 16 * RT;
  const int ks = K / ksplit;                   // multiple of 128
  const int kbeg = blockIdx.y * ks;
  const int nss = ks / 128;
  const int ksh = pcy_gemv_kshift(blockIdx.x * 4 * 16 * RT, nss);   // rotated K order (pcy_gemv_kshift)
  const bf16_t* wsrc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = q * 4 + (lane >> 4);
      int r = r0 + rt * 16 + row;
      r = r < nrows ? r : nrows - 1;
      wsrc[rt][q] = a.W + (size_t)r * K + kbeg + ((lane & 15) ^ row) * 8;
    }
  // this wave's share of an x tile set: instructions wave*BT .. wave*BT + BT-1 of the BT*4 (tile bt = i >> 2, rows (i & 3)*4 .. +3).  Row groups
  // beyond the batch are not copied at all (round 6: their MFMA columns are never stored, and every byte a CU fetches -- L2-served activations
  // included -- goes through the same ~25 GB/s: 20 rows used to copy 32); nxw = this wave's copies per step
  const bf16_t* xsrc[BT];
  int xdst[BT];
  bool xon[BT];
  int nxw = 0;
#pragma unroll
  for (int t = 0; t < BT; ++t) {
    const int i = wave * BT + t, bt = i >> 2, q = i & 3;
    const int row = q * 4 + (lane >> 4);
    int b = bt * 16 + row;
    xon[t] = bt * 16 + q * 4 < a.B;
    nxw += xon[t] ? 1 : 0;
    b = b < a.B ? b : a.B - 1;
    xsrc[t] = a.x + (size_t)b * a.ldx + kbeg + ((lane & 15) ^ row) * 8;
    xdst[t] = bt * WT + q * 1024;
  }
  auto issue = [&](int ss) __attribute__((always_inline)) {   // (past the end: the last super-step again, the counts stay uniform)
    int rs = (ss < nss ? ss : nss - 1) + ksh;
    rs = rs >= nss ? rs - nss : rs;
    const int k = rs * 128;
    const int slot = ss % S;
    char* xb = xs + slot * BT * WT;
#pragma unroll
    for (int t = 0; t < BT; ++t)
      if (!(PCY_GEMV_ABL & 1) && xon[t]) __builtin_amdgcn_global_load_lds((gv_gptr_t)(xsrc[t] + k), (gv_lds_ptr_t)(xb + xdst[t]), 16, 0, 0);
    char* dst = wring + slot * RT * WT;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((gv_gptr_t)(wsrc[rt][q] + k), (gv_lds_ptr_t)(dst + rt * WT + q * 1024), 16, 0, W_AUX);
  };
  f32x4 acc[RT][BT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) acc[rt][bt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto mma = [&](int ss) __attribute__((always_inline)) {
    const int slot = ss % S;
    const char* wb = wring + slot * RT * WT;
    const char* xb = xs + slot * BT * WT;
    bf16x8 wf[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[rt][j] = *reinterpret_cast<const bf16x8*>(wb + rt * WT + fr * 256 + (((j * 4 + fq) ^ fr) << 4));
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
      bf16x8 xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const bf16x8*>(xb + bt * WT + fr * 256 + (((j * 4 + fq) ^ fr) << 4));
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[rt][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[rt][j], xf[j], acc[rt][bt], 0, 0, 0);
    }
  };
#pragma unroll
  for (int i = 0; i < S - 1; ++i) issue(i);
  for (int ss = 0; ss < nss; ++ss) {
    // this wave's copies of pair ss have landed ((S - 2) younger pairs of nxw + 4 RT copies may be in flight)
    if ((PCY_GEMV_ABL & 1) || nxw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * (RT * 4)) : "memory");
    else if (BT == 1 || nxw == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * (1 + RT * 4)) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * (2 + RT * 4)) : "memory");
    if (!(PCY_GEMV_ABL & 2)) __builtin_amdgcn_s_barrier();                                                     // ... and everybody else's; slot (ss-1) % S is free
    issue(ss + S - 1);
    if (!(PCY_GEMV_ABL & 4)) mma(ss);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing of this workgroup's LDS may still be written when it retires
  mfma_gemv_epilogue<EPI, RT, BT>(a, acc, r0, nrows, ksplit, fr, fq);
}

template <int EPI, int BT, int S>
bool launch_mfma4_s(hipStream_t s, const PcyGemvArgs& a, int bx, int ksplit) {
  constexpr int RT = (EPI == EPI_SWIGLU) ? 2 : 1;
  constexpr size_t smem = (size_t)4 * S * RT * 4096 + (size_t)S * BT * 4096;
  static PcyLdsAttr lds;
  if (!lds.ensure(&gemv_mfma4_kernel<EPI, BT, S>, smem, 0)) return false;   // (the device refused the LDS size)
  hipLaunchKernelGGL((gemv_mfma4_kernel<EPI, BT, S>), dim3(bx, ksplit), dim3(256), smem, s, a, ksplit);
  return true;
}
// ring depths of gemv_mfma4_kernel: the deepest that fit 160 KB of LDS
template <int EPI, int BT>
bool launch_mfma4(hipStream_t s, const PcyGemvArgs& a, int bx, int ksplit) {
  constexpr int RT = (EPI == EPI_SWIGLU) ? 2 : 1;
  if constexpr (RT == 2) return launch_mfma4_s<EPI, BT, 4>(s, a, bx, ksplit);                   // 128 + 16 BT KB
  else if constexpr (BT == 1) return launch_mfma4_s<EPI, BT, 8>(s, a, bx, ksplit);              // 128 + 32 KB
  else return launch_mfma4_s<EPI, BT, 6>(s, a, bx, ksplit);                                      // 96 + 48 KB
}

template <int EPI, int BT, int S>
void launch_mfma3_s(hipStream_t s, const PcyGemvArgs& a, int bx, int ksplit) {
  constexpr int RT = (EPI == EPI_SWIGLU) ? 2 : 1;
  constexpr size_t smem = (size_t)4 * S * RT * 4096 + 2 * BT * 16 * (256 * 2 + 16);
  static PcyLdsAttr lds;
  lds.ensure(&gemv_mfma3_kernel<EPI, BT, S>, smem, 0);
  hipLaunchKernelGGL((gemv_mfma3_kernel<EPI, BT, S>), dim3(bx, ksplit), dim3(256), smem, s, a, ksplit);
}
// Ring depth S (super-steps of 128 k kept in flight per wave = S - 1): the SwiGLU kernel streams two row tiles per wave (S = 3: 96 KB
// of rings, one workgroup per CU).  The single-row-tile kernels (qkv, o, down, lm_head) ran S = 6 (96 KB, one workgroup per CU);
// S = 3 halves the rings so that TWO workgroups share a CU (48 + 33 KB each) -- same bytes in flight per CU, twice the
// workgroups to split K over: measured no better, S = 6 stays.
inline int gemv_ring_depth_rt1() { return 6; }
template <int EPI, int BT>
void launch_mfma3(hipStream_t s, const PcyGemvArgs& a, int bx, int ksplit) {
  constexpr int RT = (EPI == EPI_SWIGLU) ? 2 : 1;
  if constexpr (RT == 2) launch_mfma3_s<EPI, BT, 3>(s, a, bx, ksplit);
  else if (gemv_ring_depth_rt1() == 3) launch_mfma3_s<EPI, BT, 3>(s, a, bx, ksplit);
  else launch_mfma3_s<EPI, BT, 6>(s, a, bx, ksplit);
}

template <int EPI>
void launch_mfma(hipStream_t s, const PcyGemvArgs& a) {
  const int nrows = (EPI == EPI_SWIGLU) ? 2 * a.N : a.N;
  {
    const int rpw = (EPI == EPI_SWIGLU) ? 32 : 16;          // weight rows per wave
    const int bx = (nrows + 4 * rpw - 1) / (4 * rpw);
    // K split until ~3/4 of the CUs have a workgroup (measured batch-32 decode step with the fill target at 128 / 192 / 256 / 512
    // workgroups: 4.46 / 4.21 / 4.28 / 4.56 ms); only with a plain / residual epilogue and a workspace
    int ksplit = 1;
    constexpr int kfill = 192;
    // K that is a multiple of 128 but not of 512 (Llama-2-7B's down projection, K = 11008 = 86 x 128: ProCyon-Split): only gemv_mfma4_kernel
    // steps in units of 128 -- the split is then counted in those units and the older kernels (the PCY_DISABLE twins) do not apply
    const int kunit = a.K % 512 == 0 ? 512 : 128;
    if ((EPI == EPI_STORE || EPI == EPI_RESID) && a.splitk_ws && a.N % 4 == 0 && (a.ldy & 3) == 0)
      while (ksplit < 8 && bx * ksplit < kfill && a.K % (ksplit * 2 * kunit) == 0 &&
             (size_t)(ksplit * 2) * a.B * a.N * 4 <= a.splitk_ws_bytes) ksplit *= 2;
    // PCY_DISABLE=gemv_lds: weights through registers (gemv_mfma2_kernel) instead of LDS-DMA; read per call, same bits
    const bool lds = (!pcy_off("gemv_lds") && a.K % (ksplit * 256) == 0) || kunit == 128;
    // PCY_DISABLE=gemv_mfma4: the previous schedule (gemv_mfma3_kernel: x one chunk ahead); read per call, same bits
    const bool v4 = lds && (!pcy_off("gemv_mfma4") || kunit == 128) && a.K % (ksplit * 128) == 0;
    if (v4 && (a.B <= 16 ? launch_mfma4<EPI, 1>(s, a, bx, ksplit) : launch_mfma4<EPI, 2>(s, a, bx, ksplit))) {
    } else if (lds) {
      if (a.B <= 16) launch_mfma3<EPI, 1>(s, a, bx, ksplit);
      else launch_mfma3<EPI, 2>(s, a, bx, ksplit);
    } else if (a.B <= 16) hipLaunchKernelGGL((gemv_mfma2_kernel<EPI, 1>), dim3(bx, ksplit), dim3(256), 0, s, a, ksplit);
    else hipLaunchKernelGGL((gemv_mfma2_kernel<EPI, 2>), dim3(bx, ksplit), dim3(256), 0, s, a, ksplit);
    if (a.defer_finish) *a.defer_finish = 0;
    if (ksplit > 1 && EPI == EPI_STORE && a.defer_finish && !a.bias) { *a.defer_finish = ksplit; return; }
    if (ksplit > 1) {
      const int eb = (int)(((size_t)a.B * (a.N / 4) + 255) / 256);
      if (EPI == EPI_RESID && a.next_rms_w && a.next_xn && a.fused_next && a.N % 8 == 0 && a.N <= 8192 && a.ldy == a.N &&
          !pcy_off("finish_norm")) {
        hipLaunchKernelGGL(gemv_splitk_finish_norm_kernel, dim3(a.B), dim3(256), 0, s, a, ksplit);
        *a.fused_next = 1;
        return;
      }
      if (EPI == EPI_RESID) hipLaunchKernelGGL(gemv_splitk_finish_kernel<EPI_RESID>, dim3(eb), dim3(256), 0, s, a, ksplit);
      else hipLaunchKernelGGL(gemv_splitk_finish_kernel<EPI_STORE>, dim3(eb), dim3(256), 0, s, a, ksplit);
    }
    return;
  }
}



// ------------------------------------------------------------------------------------------------
// Batch-1 decode: gate/up + SwiGLU -> down + residual as ONE launch (PcyMlpChainArgs; pcy_decode_mlp).  The body, shared with the
// decode layer launch (pcy_attn.hip), lives in pcy_mlp_chain.h.
//
// As two launches each stage pays a kernel boundary (~1.7 us) and its own ramp (~3.4 us: first bytes after the launch, uneven
// tail) on 22-42 us of streaming, and HBM idles meanwhile.  Here 256 workgroups (one per CU, 8 waves, all resident) run the
// stages back to back.  The down projection needs the WHOLE `act` vector, produced by all workgroups.
//
// Hand-over without flags (pcy_handover.h): a producer stores every element as ONE 32-bit word {tag : bf16 value}, written
// through to memory and never waited for; a consumer loads the words it wants with L1-bypassing loads and accepts them when
// every tag is the current one, else asks again (in-kernel stamps of a flag version: 5 us from the last flag to the vector in
// LDS, both trips queued behind the workgroup's own weight prefetch; flags bought nothing over three launches).
//
//   gate/up -> down: a wave produces its two units in slot order, so the first half of `act` is complete chip-wide when a wave
//     is half way through its rows.  A wave that has finished its gate/up rows requests its share of the first half of act and,
//     right behind it, the first two batches (28 KB) of its down rows; the second half of act is requested next and checked only
//     two batches later (~9 us), by when the slowest workgroup has delivered it.  Nobody waits for the hop and the spread of
//     the workgroups' finishing times (even XCDs get ~7 % less bandwidth than odd ones: 34-43 us) is absorbed instead of added.
//
// Arithmetic per output row = gemv_stream_kernel's (same k order of the accumulation, same reduction tree, same rounding
// points); the RMSNorm statistic is summed in the order of the stand-alone launch (`vthr` threads, block_sum_rt): bit-identical.
__global__ __launch_bounds__(MC_NT, 2) void mlp_chain_kernel(PcyMlpChainArgs a, int vthr_gu) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const uint32_t tag = *a.epoch & 0xffffu;
  uint4 wa[16], wb[16];
  unsigned long long* tr = a.trace ? a.trace + (size_t)blockIdx.x * 16 : nullptr;
  if (tr && tid == 0) tr[0] = wall_clock64();
  mc_mlp_body<false>(a, smem, vthr_gu, tag, gridDim.x, blockIdx.x, 0, wa, wb, tr);
  if (tr && tid == 0) tr[4] = wall_clock64();
}

// Launch-per-stage twin of the small-batch decode step's down projection (pcy_decode_nb.hip): K is walked as four interleaved sets of
// 512-element blocks {g + 4 j}, each set accumulated per lane in j order and reduced over the wave, the four partial sums added in g order:
//   y = bf16( bf16(((p0 + p1) + p2) + p3) + resid ).   One wave per output row; not a fast kernel.
__global__ __launch_bounds__(256) void gemv_kwin4_kernel(PcyGemvArgs a) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.N) return;
  const int nj = a.K / 2048;
  for (int b = 0; b < a.B; ++b) {
    float p[4];
    for (int g = 0; g < 4; ++g) {
      float acc = 0.f;
      for (int j = 0; j < nj; ++j) {
        const int k = ((g + 4 * j) * 64 + lane) * 8;
        acc = dot8(ldg_nt(a.W + (size_t)row * a.K + k), ldg16(a.x + (size_t)b * a.ldx + k), acc);
      }
      p[g] = wave_sum(acc);
    }
    float v = rbf(((p[0] + p[1]) + p[2]) + p[3]);
    if (a.resid) v = rbf(v + bf2f(a.resid[(size_t)b * a.ldy + row]));
    if (lane == 0) a.y[(size_t)b * a.ldy + row] = f2bf(v);
  }
}

}  // namespace

bool pcy_launch_splitk_finish_norm(hipStream_t s, const float* ws, int splits, int rows, int N, const bf16_t* resid, bf16_t* y,
                                   const bf16_t* next_rms_w, bf16_t* next_xn, float rms_eps, int rms_cast) {
  if (N % 8 || N > 8192 || rows <= 0 || splits < 1) return false;
  PcyGemvArgs a{};
  a.splitk_ws = const_cast<float*>(ws); a.B = rows; a.N = N; a.resid = resid; a.y = y; a.ldy = N;
  a.next_rms_w = next_rms_w; a.next_xn = next_xn; a.rms_eps = rms_eps; a.rms_cast = rms_cast;
  hipLaunchKernelGGL(gemv_splitk_finish_norm_kernel, dim3(rows), dim3(256), 0, s, a, splits);
  return true;
}

// threads of the stand-alone RMS-fused launch (launch_nb / pick_grid) for N output rows: the order of its statistic
int pcy_gemv_rms_threads(int N) {
  const int R = ((N + 3) / 4 < GEMV_MAX_WAVES) ? 2 : 4;
  int blocks, wpb;
  pick_grid((N + R - 1) / R, blocks, wpb);
  return wpb * 64;
}

bool pcy_launch_mlp_chain(hipStream_t s, const PcyMlpChainArgs& a, int n_cu) {
  // 256 workgroups, one per CU, must all be resident (the hand-over needs every workgroup's rows).  Geometry: one down unit
  // (two rows) per wave, act halves of whole k-batches, a wave's share of a half in 4 loads per lane.
  const int NW = GEMV_CUS * MC_WV;
  if (n_cu < GEMV_CUS || a.d != 2 * NW || a.d != MC_WV * 512 || a.F != 2 * 7 * 1024 || a.F % (2 * 512 * MC_UNB_D)) return false;
  const size_t smem = (size_t)(2 * a.d + a.F) * 2 + 128;
  if (smem > 64 * 1024) return false;
  static PcyResidentCache res;
  if (!res.check(smem, [&] { return pcy_all_resident(mlp_chain_kernel, MC_NT, smem, GEMV_CUS, n_cu); })) return false;
  hipLaunchKernelGGL(mlp_chain_kernel, dim3(GEMV_CUS), dim3(MC_NT), smem, s, a, pcy_gemv_rms_threads(a.F));
  return true;
}


// Smallest batch that takes the MFMA GEMVs (weights once per 16 / 32 rows) instead of the streaming kernel (x rows in LDS, one
// dot product per row and lane): decode step at T = 128, batch 2 / 3 / 4: 3.49 / 3.99 / 4.42 ms streaming, 4.01 / 4.01 / 4.02 MFMA
int pcy_mfma_min_batch() { return 4; }

bool pcy_launch_gemv_kwin4(hipStream_t s, const PcyGemvArgs& a) {
  if (a.K % 2048 || a.bias || a.rms_w || (a.epi != EPI_RESID && a.epi != EPI_STORE)) return false;
  hipLaunchKernelGGL(gemv_kwin4_kernel, dim3((a.N + 3) / 4), dim3(256), 0, s, a);
  return true;
}

void pcy_launch_gemv(hipStream_t s, const PcyGemvArgs& a00) {
  PcyGemvArgs a0 = a00;
  a0.plain_loads = 0;
  // batches from pcy_mfma_min_batch() on MFMA, 32 rows per pass over the weights (x already normalised by the caller: the fused
  // RMSNorm prologue is a feature of the streaming kernel)
  if (!a0.force_stream && (a0.B >= pcy_mfma_min_batch() || a0.force_mfma) && a0.K % 128 == 0 && a0.rms_w == nullptr && (a0.ldx % 8) == 0) {
    for (int b0 = 0; b0 < a0.B; b0 += 32) {
      PcyGemvArgs a = a0;
      a.B = (a0.B - b0) < 32 ? (a0.B - b0) : 32;
      a.x = a0.x + (size_t)b0 * a0.ldx;
      a.y = a0.y + (size_t)b0 * a0.ldy;
      if (a0.resid) a.resid = a0.resid + (size_t)b0 * a0.ldy;
      switch (a.epi) {
        case EPI_STORE: launch_mfma<EPI_STORE>(s, a); break;
        case EPI_RESID: launch_mfma<EPI_RESID>(s, a); break;
        case EPI_GELU_ERF: launch_mfma<EPI_GELU_ERF>(s, a); break;
        case EPI_GELU_ESM: launch_mfma<EPI_GELU_ESM>(s, a); break;
        case EPI_SWIGLU: launch_mfma<EPI_SWIGLU>(s, a); break;
      }
    }
    return;
  }
  // batch rows in groups of <= 4 (fused RMSNorm, or K not a multiple of 512)
  for (int b0 = 0; b0 < a0.B; b0 += 4) {
    PcyGemvArgs a = a0;
    const int nb = (a0.B - b0) < 4 ? (a0.B - b0) : 4;
    a.x = a0.x + (size_t)b0 * a0.ldx;
    a.y = a0.y + (size_t)b0 * a0.ldy;
    if (a0.resid) a.resid = a0.resid + (size_t)b0 * a0.ldy;
    a.B = nb;
    switch (nb) {
      case 1: launch_epi<1>(s, a); break;
      case 2: launch_epi<2>(s, a); break;
      case 3: launch_epi<3>(s, a); break;
      default: launch_epi<4>(s, a); break;
    }
  }
}
