// bf16 MFMA GEMM for the compute-bound side of the path (ESM2 encoder, Llama prefill; rows A2/A6):
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T ),  A and W both K-contiguous (activations x nn.Linear).
//
// gfx950 design (round 1: the "128x128, 2-barrier" structure of the CDNA4 guide):
//   * 128x128x64 tile per 256-thread workgroup (2x2 waves, 64x64 per wave = 4x4 MFMA 16x16x32 tiles)
//   * global -> LDS by `global_load_lds_dwordx4` (16 B/lane, LDS image lane-linear), double buffered;
//     the XOR bank swizzle is applied on the per-lane SOURCE address and on the ds_read_b128 address
//   * operands are fed swapped (W rows as the MFMA "A" operand) so every lane ends up with 4
//     consecutive output features of one token -> 8-byte epilogue loads/stores
//   * fused epilogues reproduce the reference's bf16 rounding points (bias, residual add, erf-GELU,
//     ESM op-by-op GELU, SwiGLU on 16-row interleaved gate/up weights)
//   * any M and N (row clamping + predicated stores), K % 64 == 0
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "pcy_internal.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int GEMM_THREADS = 256;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

// The ESM GELU chain (five bf16 tensors, pcy_common.h::gelu_esm_chain) is a FUNCTION bf16 -> bf16, so the fc1 epilogue looks
// it up instead of evaluating ~45 VALU instructions per output (139 of the 591 us of that GEMM at M = 32832):
// g_gelu_lut[s*4096 + (|bits| - (102 << 7))] for 2^-25 <= |x| < 2^7 (32 exponents: 4096 entries = 8 KiB per sign), filled once per
// device by the chain itself (bit-identical by construction); values outside the table take closed forms.  Once the mainloop has
// ended the table is copied over the dead tile buffers in LDS and gathered with ds_read_u16.
constexpr int GELU_LUT_E0 = 102, GELU_LUT_HALF = 32 * 128, GELU_LUT_N = 2 * GELU_LUT_HALF;
__device__ uint16_t g_gelu_lut[GELU_LUT_N];
__global__ void gelu_lut_build_kernel() {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= GELU_LUT_N) return;
  const uint32_t bits = (uint32_t)((i % GELU_LUT_HALF) + (GELU_LUT_E0 << 7)) | (i >= GELU_LUT_HALF ? 0x8000u : 0u);
  g_gelu_lut[i] = f2bf(gelu_esm_chain(__uint_as_float(bits << 16)));
}
// all threads of the workgroup; `lds` = start of the (no longer read) tile buffers
template <int NT>
__device__ __forceinline__ void gelu_lut_to_lds(char* lds) {
  __syncthreads();
  for (int i = threadIdx.x; i < GELU_LUT_N * 2 / 16; i += NT)
    reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(g_gelu_lut)[i];
  __syncthreads();
}
// SPARSE image (persistent 256 x 256 kernel): the positive half at `pos`, the negative half 64 KiB further, so that the entry of the
// 16-bit pattern b sits at  pos - 2 (E0 << 7) + 2 b  for either sign -- the gather address is the pattern shifted left by one.
constexpr int GELU_SPARSE_NEG = 32768;   // elements between the halves
template <int NT>
__device__ __forceinline__ void gelu_lut_to_lds_sparse(char* pos) {
  __syncthreads();
  for (int i = threadIdx.x; i < GELU_LUT_HALF * 2 / 16; i += NT) {
    reinterpret_cast<uint4*>(pos)[i] = reinterpret_cast<const uint4*>(g_gelu_lut)[i];
    reinterpret_cast<uint4*>(pos + GELU_SPARSE_NEG * 2)[i] = reinterpret_cast<const uint4*>(g_gelu_lut + GELU_LUT_HALF)[i];
  }
  __syncthreads();
}
// Branch-free (a divergent fall-back to the chain at each of the 128 unrolled call sites made hipcc keep the epilogue as a
// loop and move the accumulators to scratch: 591 -> 2593 us).  Outside the table the chain has closed forms:
//   |x| <  2^-17: erf term rounds away, 1 + t3 = 1          -> bf16(x * 0.5)
//   |x| >= 2^7  : erf = +-1 -> t4 = 2 or 0                  -> x   or   bf16(x * 0.5) * 0  (= -0, NaN for -inf, as the chain)
__device__ __forceinline__ float gelu_esm_lut(float v /* bf16-valued */, const uint16_t* lut, int neg_off = GELU_LUT_HALF) {
  const uint32_t b = __float_as_uint(v) >> 16;
  const uint32_t mag = b & 0x7fffu;
  const uint32_t idx = mag - (uint32_t)(GELU_LUT_E0 << 7);
  const bool in = idx < (uint32_t)GELU_LUT_HALF;
  const float t = __uint_as_float((uint32_t)lut[(in ? idx : 0u) + ((b >> 15) ? neg_off : 0)] << 16);
  const float half = rbf(v * 0.5f);
  const float big = (b >> 15) ? half * 0.0f : v;
  const float out = mag < (uint32_t)(GELU_LUT_E0 << 7) ? half : big;
  return in ? t : out;
}

// stage a [128 rows][BK k] bf16 tile with 1-KiB wave-instructions (LDS image lane-linear).  A row holds CPR = BK/8
// 16-byte chunks; LDS chunk position c' of row r holds global chunk c' ^ swz(r), the same XOR is applied on the read side:
//   BK=64 (128-B rows): swz = r & 7        BK=32 (64-B rows): swz = (r >> 2) & 3      -> conflict-free ds_read_b128
// SW = 1 (BK = 64 only): the swizzle of a W tile whose rows are read in the PERMUTED order of the 256 x 256 kernels (wperm_row
// below: the 16 lanes fr of a fragment read hit rows a*8 + h*4 + b, a = fr >> 2, b = fr & 3) -- s(r) = 2*((r >> 3) & 3) + ((r >> 1) & 1)
// puts the 16 lanes of every ds_read_b128 lane group on 16 distinct 16-byte slots, as r & 7 does for 16 consecutive rows.
// SW = 2: the same for the SwiGLU row order (wperm_row_swiglu: a = bits 5 and 3 of the row).
template <int BK, int SW = 0>
__device__ __forceinline__ int swz(int r) {
  if (SW == 1) return (((r >> 3) & 3) << 1) | ((r >> 1) & 1);
  if (SW == 2) return (((r >> 5) & 1) << 2) | (((r >> 3) & 1) << 1) | ((r >> 1) & 1);
  return BK == 64 ? (r & 7) : ((r >> 2) & 3);
}
// Row order of the W operand inside a wave's feature range (256 x 256 kernels).  MFMA tile i, operand row q (= lane & 15 on the
// read side, = fq*4 + r on the accumulator side) is the wave's feature (i>>1)*32 + (q>>2)*8 + (i&1)*4 + (q&3): a lane's
// accumulators of the tile pair (2p, 2p+1) are then EIGHT consecutive features of one token -> 16-byte epilogue loads / stores,
// four lanes = 64 contiguous bytes of a row per instruction instead of 32 (the write path of a CU is bound by requests: the
// plain epilogue of a tile round took 8 us).  A pure relabelling of which weight row sits in which MFMA row: every output element
// keeps its k order, same bits.
__device__ __forceinline__ int wperm_row(int i, int q) { return (i >> 1) * 32 + (q >> 2) * 8 + (i & 1) * 4 + (q & 3); }

// The SwiGLU form (packed W rows: 16 gate rows, then the 16 up rows of the same features, ...): a quad of tiles (4Q .. 4Q+3) = (gate,
// up) of features +0..3 and (gate, up) of features +4..7 of the lane's eight output features Q*32 + fq*8 + [0, 8), so the gate and
// the up value of a feature meet in one lane and the product leaves as a 16-byte store.
__device__ __forceinline__ int wperm_row_swiglu(int i, int q) {
  return (i >> 2) * 64 + (q >> 3) * 32 + (i & 1) * 16 + ((q >> 2) & 1) * 8 + ((i >> 1) & 1) * 4 + (q & 3);
}
template <int EPI>
__device__ __forceinline__ int wperm(int i, int q) { return EPI == EPI_SWIGLU ? wperm_row_swiglu(i, q) : wperm_row(i, q); }

// STG = 1: the pieces go out as `buffer_load_dwordx4 ... offen lds` -- a buffer resource based at the tile's first row (wave-uniform,
// SGPRs), the lane's byte offset inside the tile (32 bits, the same for every k-step: computed once per tile) and the k offset as
// the instruction's SCALAR offset.  The global_load_lds form (STG = 0) carries a 64-bit address per lane and piece, which the k-loop
// re-forms with a v_lshl_add_u64 per piece; with the wave index read into an SGPR (readfirstlane) the LDS destination (M0) is
// scalar arithmetic as well instead of a VGPR add + v_readfirstlane per piece.
template <int BK, int ROWS = 128, int NW = 4, int SW = 0, int STG = 0>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int ld, int row0, int nrows_valid, int k0,
                                           char* lds_tile, int wave, int lane) {
  constexpr int CPR = BK / 8, RPI = 64 / CPR, NINST = ROWS / RPI;
  if constexpr (STG >= 1) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g + (size_t)row0 * ld), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int i = 0; i < NINST / NW; ++i) {
      const int inst = wave * (NINST / NW) + i;
      const int r = inst * RPI + lane / CPR;
      const int c = (lane % CPR) ^ swz<BK, SW>(r);
      const int rl = row0 + r < nrows_valid ? r : nrows_valid - 1 - row0;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds_tile + inst * 1024), 16, (rl * ld + c * 8) * 2, k0 * 2, 0, 0);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NINST / NW; ++i) {
    const int inst = wave * (NINST / NW) + i;
    const int r = inst * RPI + lane / CPR;
    const int cp = lane % CPR;
    const int c = cp ^ swz<BK, SW>(r);
    int gr = row0 + r;
    gr = gr < nrows_valid ? gr : nrows_valid - 1;
    const bf16_t* src = g + (size_t)gr * ld + k0 + c * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(lds_tile + inst * 1024), 16, 0, 0);
  }
}

template <int BK, int SW = 0>
__device__ __forceinline__ bf16x8 lds_frag(const char* lds_tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds_tile + row * (BK * 2) + ((chunk ^ swz<BK, SW>(row)) << 4));
}

// Tile rasterisation for L2 reuse.  Each XCD (private 4 MiB L2) receives a contiguous run of the logical tile order
// (see the remap in the kernels); that order walks the N dimension in GROUPS of `gn` column tiles: for each group,
// all row tiles, inner loop over the group's columns.  The group's W panels (gn x 128 x K x 2 B <= ~2.5 MiB) stay
// L2-resident for the whole sweep over M, and every A panel is reused gn times back to back, instead of re-streaming
// all of W from the fabric once per row tile (PMC: 2.3 GB fetched for the 94 MB ESM qkv GEMM before this).
template <int BM = 128, int BN = 128>
__device__ __forceinline__ void tile_origin(const PcyGemmArgs& a, int tile, int& m0, int& n0) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const int gn = a.gn;
  const int per_group = tiles_m * gn;
  const int g = tile / per_group;
  const int gw = (g + 1) * gn <= tiles_n ? gn : tiles_n - g * gn;   // width of this (possibly last, narrower) group
  const int r = tile - g * per_group;
  m0 = (r / gw) * BM;
  n0 = (g * gn + r % gw) * BN;
}

// epilogue shared by the GEMM variants: lane holds D[n = fq*4 + r][m = fr] of each 16x16 tile.
// All global reads of the epilogue (bias: 16 values per lane, residual: 16 x 8 B per lane) are issued up front as
// independent vector loads -- element-wise loads inside the rounding chain serialised ~64 L2 round trips per tile
// (20 us of the 37 us a K=1280 tile took).
template <int EPI, int WTN = 4, int WTM = 4, bool ROPE_OK = true>
__device__ __forceinline__ void gemm_epilogue(const PcyGemmArgs& a, f32x4 (&acc)[WTN][WTM], int m0, int n0, int wm, int wn, int fr, int fq,
                                              const uint16_t* gelu_lut = nullptr) {
  m0 += wm * WTM * 16 - wm * 64;   // the code below adds wm * 64 / wn * 64 (the 4 x 4 layout)
  n0 += wn * WTN * 16 - wn * 64;
  auto emit = [&](int m, int n, uint2 w) __attribute__((always_inline)) {
    *reinterpret_cast<uint2*>(a.C + (size_t)m * a.ldc + n) = w;
  };
  const bool vec_ok = (a.ldc % 4 == 0) && (a.N % 4 == 0) && (a.resid == nullptr || a.ldr % 4 == 0);
  if (EPI == EPI_SWIGLU) {
#pragma unroll
    for (int j = 0; j < WTM; ++j) {
      const int m = m0 + wm * 64 + j * 16 + fr;
      if (m >= a.M) continue;
#pragma unroll
      for (int i = 0; i < WTN; i += 2) {  // tile i = gate, i+1 = up of the same 16 features
        const int nrow = n0 + wn * 64 + i * 16;         // packed row of the gate tile
        const int f = (nrow >> 5) * 16 + fq * 4;        // output feature
        if (nrow + 16 + fq * 4 >= a.N) continue;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = rbf(acc[i][j][r]), u = rbf(acc[i + 1][j][r]);
          o[r] = rbf(silu_f(g)) * u;
        }
        *reinterpret_cast<uint2*>(a.C + (size_t)m * a.ldc + f) = make_uint2(pack_bf(o[0], o[1]), pack_bf(o[2], o[3]));
      }
    }
    return;
  }
  if constexpr (EPI == EPI_STORE && ROPE_OK) if (a.rope_cos != nullptr) {
    // fused rotary: a head = 4 consecutive 16-feature tiles, this lane holds features e..e+3 (tile t) and their partners
    // e+32.. (tile t+2) of the same token.  All loads are issued in batches ahead of the arithmetic: bias once, the
    // positions of the WTM tokens once, then per token its four cos and four sin quads (shared by every head).
    constexpr int NH = WTN / 4;
    float b1[NH][2][4], b2[NH][2][4];
#pragma unroll
    for (int hg = 0; hg < NH; ++hg)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n1 = n0 + wn * 64 + hg * 64 + t * 16 + fq * 4;
        const int nc = (a.bias && n1 + 35 < a.N) ? n1 : 0;
        const uint2 u1 = a.bias ? *reinterpret_cast<const uint2*>(a.bias + nc) : make_uint2(0, 0);
        const uint2 u2 = a.bias ? *reinterpret_cast<const uint2*>(a.bias + nc + 32) : make_uint2(0, 0);
        b1[hg][t][0] = lo_bf(u1.x); b1[hg][t][1] = hi_bf(u1.x); b1[hg][t][2] = lo_bf(u1.y); b1[hg][t][3] = hi_bf(u1.y);
        b2[hg][t][0] = lo_bf(u2.x); b2[hg][t][1] = hi_bf(u2.x); b2[hg][t][2] = lo_bf(u2.y); b2[hg][t][3] = hi_bf(u2.y);
      }
    int pj[WTM];
#pragma unroll
    for (int j = 0; j < WTM; ++j) {
      const int m = m0 + wm * 64 + j * 16 + fr;
      pj[j] = a.rope_pos[m < a.M ? m : a.M - 1];
    }
    uint2 cs[WTM][2][4];   // [token][t][cos lo, cos hi, sin lo, sin hi]
#pragma unroll
    for (int j = 0; j < WTM; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int e = t * 16 + fq * 4;
        const bf16_t* crow = a.rope_cos + (size_t)pj[j] * 64 + e;
        const bf16_t* srow = a.rope_sin + (size_t)pj[j] * 64 + e;
        cs[j][t][0] = *reinterpret_cast<const uint2*>(crow);
        cs[j][t][1] = *reinterpret_cast<const uint2*>(crow + 32);
        cs[j][t][2] = *reinterpret_cast<const uint2*>(srow);
        cs[j][t][3] = *reinterpret_cast<const uint2*>(srow + 32);
      }
#pragma unroll
    for (int j = 0; j < WTM; ++j) {
      const int m = m0 + wm * 64 + j * 16 + fr;
      if (m >= a.M) continue;
#pragma unroll
      for (int hg = 0; hg < NH; ++hg) {
        const int nh = n0 + wn * 64 + hg * 64;   // first feature of this head
        if (nh >= a.N) continue;
        const bool rot = nh < a.rope_ncols, scl = nh < a.rope_qcols;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int i1 = hg * 4 + t, i2 = i1 + 2;
          const int n1 = nh + t * 16 + fq * 4, n2 = n1 + 32;
          const uint2 c1 = cs[j][t][0], c2 = cs[j][t][1], s1 = cs[j][t][2], s2 = cs[j][t][3];
          const float cc1[4] = {lo_bf(c1.x), hi_bf(c1.x), lo_bf(c1.y), hi_bf(c1.y)}, cc2[4] = {lo_bf(c2.x), hi_bf(c2.x), lo_bf(c2.y), hi_bf(c2.y)};
          const float ss1[4] = {lo_bf(s1.x), hi_bf(s1.x), lo_bf(s1.y), hi_bf(s1.y)}, ss2[4] = {lo_bf(s2.x), hi_bf(s2.x), lo_bf(s2.y), hi_bf(s2.y)};
          float o1[4], o2[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float y1 = rbf(acc[i1][j][r] + b1[hg][t][r]), y2 = rbf(acc[i2][j][r] + b2[hg][t][r]);
            if (rot) {
              if (scl) { y1 = rbf(y1 * a.rope_scale); y2 = rbf(y2 * a.rope_scale); }
              if (a.rope_mode == 0) {
                o1[r] = rbf(rbf(y1 * cc1[r]) + rbf(-y2 * ss1[r]));
                o2[r] = rbf(rbf(y2 * cc2[r]) + rbf(y1 * ss2[r]));
              } else {
                o1[r] = y1 * cc1[r] + (-y2) * ss1[r];
                o2[r] = y2 * cc2[r] + y1 * ss2[r];
              }
            } else {
              o1[r] = y1; o2[r] = y2;
            }
          }
          emit(m, n1, make_uint2(pack_bf(o1[0], o1[1]), pack_bf(o1[2], o1[3])));
          emit(m, n2, make_uint2(pack_bf(o2[0], o2[1]), pack_bf(o2[2], o2[3])));
        }
      }
    }
    return;
  }
  // bias of this lane's 4 x 4 output features
  float bias[WTN][4];
#pragma unroll
  for (int i = 0; i < WTN; ++i) {
    const int n = n0 + wn * 64 + i * 16 + fq * 4;
    if (a.bias && vec_ok && n < a.N) {
      const uint2 bb = *reinterpret_cast<const uint2*>(a.bias + n);
      bias[i][0] = lo_bf(bb.x); bias[i][1] = hi_bf(bb.x); bias[i][2] = lo_bf(bb.y); bias[i][3] = hi_bf(bb.y);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = (n + r) < a.N ? (n + r) : a.N - 1;
        bias[i][r] = a.bias ? bf2f(a.bias[nn]) : 0.f;
      }
    }
  }
  uint2 res[WTM][WTN];
  if (EPI == EPI_RESID && vec_ok) {
#pragma unroll
    for (int j = 0; j < WTM; ++j) {
      const int m = m0 + wm * 64 + j * 16 + fr;
      const int mc = m < a.M ? m : a.M - 1;
#pragma unroll
      for (int i = 0; i < WTN; ++i) {
        const int n = n0 + wn * 64 + i * 16 + fq * 4;
        const int nc = n < a.N ? n : 0;
        res[j][i] = *reinterpret_cast<const uint2*>(a.resid + (size_t)mc * a.ldr + nc);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < WTM; ++j) {
    const int m = m0 + wm * 64 + j * 16 + fr;
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < WTN; ++i) {
      const int n = n0 + wn * 64 + i * 16 + fq * 4;
      if (n >= a.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = rbf(acc[i][j][r] + bias[i][r]);
      if (EPI == EPI_RESID) {
        if (vec_ok) {
          v[0] = rbf(v[0] + lo_bf(res[j][i].x)); v[1] = rbf(v[1] + hi_bf(res[j][i].x));
          v[2] = rbf(v[2] + lo_bf(res[j][i].y)); v[3] = rbf(v[3] + hi_bf(res[j][i].y));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < a.N) v[r] = rbf(v[r] + bf2f(a.resid[(size_t)m * a.ldr + n + r]));
        }
      }
      if (EPI == EPI_GELU_ERF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(gelu_erf_f(v[r]));
      }
      if (EPI == EPI_GELU_ESM) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_lut ? gelu_esm_lut(v[r], gelu_lut) : gelu_esm_chain(v[r]);
      }
      if (vec_ok) {
        emit(m, n, make_uint2(pack_bf(v[0], v[1]), pack_bf(v[2], v[3])));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) a.C[(size_t)m * a.ldc + n + r] = f2bf(v[r]);
      }
    }
  }
}

// Epilogue of the 256 x 256 kernels with the W rows in wperm_row order: the lane (fr, fq) holds, in the tile pair (2p, 2p+1),
// features nw + p*32 + fq*8 + [0, 8) of token mw + j*16 + fr (tile 2p: +0..3, tile 2p+1: +4..7).  Same values, same rounding
// order as gemm_epilogue; bias / residual / rotary tables / output move as 16-byte pieces.  mw / nw = first token / feature of
// the wave's tile (callers with half a wave tile -- the fp8 kernels -- pass the half's first feature).
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = lo_bf(u.x); f[1] = hi_bf(u.x); f[2] = lo_bf(u.y); f[3] = hi_bf(u.y);
  f[4] = lo_bf(u.z); f[5] = hi_bf(u.z); f[6] = lo_bf(u.w); f[7] = hi_bf(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf(f[0], f[1]), pack_bf(f[2], f[3]), pack_bf(f[4], f[5]), pack_bf(f[6], f[7]));
}
template <int EPI, int WTN, int WTM, int JBR = 2>
__device__ __forceinline__ void gemm_epilogue_perm(const PcyGemmArgs& a, f32x4 (&acc)[WTN][WTM], int mw, int nw, int fr, int fq,
                                                   const uint16_t* gelu_lut = nullptr, const float* sx = nullptr,
                                                   int gelu_neg_off = GELU_LUT_HALF) {
  // sx != nullptr (fp8 kernels): the accumulators are still raw; token j's scale sx[j] and the feature scales a.sw are applied
  // here, (acc * sx) * sw as in oracle/fp8_ref.py, feature scales fetched per tile pair (all 32 of a wave tile up front spill)
  constexpr int NP = WTN / 2;
  const bool vec_ok = (a.ldc % 8 == 0) && (a.N % 8 == 0) && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 &&
                      (a.resid == nullptr || (a.ldr % 8 == 0 && (reinterpret_cast<uintptr_t>(a.resid) & 15) == 0)) &&
                      (a.bias == nullptr || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0);
  // (a wave tile that lies entirely behind the rotated columns -- the V third of a qkv projection -- takes the plain path below: no
  // position / table loads; rot = false gives the same bits)
  if constexpr (EPI == EPI_STORE) if (a.rope_cos != nullptr && nw < a.rope_ncols) {
    // fused rotary (head_dim 64; callers guarantee N % 64 == 0 and 16-byte aligned rows): a head = 4 tiles, the lane holds
    // e = fq*8 + [0, 8) in tiles (4hg, 4hg+1) and the partners e + 32 in tiles (4hg+2, 4hg+3).  Loads first: bias, positions,
    // then per token four 16-byte pieces of its cos / sin rows (shared by every head).
    constexpr int NH = WTN / 4;
    float b1[NH][8], b2[NH][8];
#pragma unroll
    for (int hg = 0; hg < NH; ++hg) {
      const int n1 = nw + hg * 64 + fq * 8;
      const int nc = (a.bias && n1 + 40 <= a.N) ? n1 : 0;
      uint4 u1 = make_uint4(0, 0, 0, 0), u2 = u1;
      if (a.bias) { u1 = *reinterpret_cast<const uint4*>(a.bias + nc); u2 = *reinterpret_cast<const uint4*>(a.bias + nc + 32); }
      unpack8(u1, b1[hg]);
      unpack8(u2, b2[hg]);
    }
    int pj[WTM];
#pragma unroll
    for (int j = 0; j < WTM; ++j) {
      const int m = mw + j * 16 + fr;
      pj[j] = a.rope_pos[m < a.M ? m : a.M - 1];
    }
    constexpr int JB = 2;   // tokens per batch of table loads (all four at once: 64 VGPRs of 16-byte pieces, spills)
#pragma unroll
    for (int j0 = 0; j0 < WTM; j0 += JB) {
      uint4 cs[JB][4];   // [token][cos lo, cos hi, sin lo, sin hi]
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        const bf16_t* crow = a.rope_cos + (size_t)pj[j0 + jj] * 64 + fq * 8;
        const bf16_t* srow = a.rope_sin + (size_t)pj[j0 + jj] * 64 + fq * 8;
        cs[jj][0] = *reinterpret_cast<const uint4*>(crow);
        cs[jj][1] = *reinterpret_cast<const uint4*>(crow + 32);
        cs[jj][2] = *reinterpret_cast<const uint4*>(srow);
        cs[jj][3] = *reinterpret_cast<const uint4*>(srow + 32);
      }
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        const int j = j0 + jj;
        const int m = mw + j * 16 + fr;
        if (m >= a.M) continue;
        float cc1[8], cc2[8], ss1[8], ss2[8];
        unpack8(cs[jj][0], cc1); unpack8(cs[jj][1], cc2); unpack8(cs[jj][2], ss1); unpack8(cs[jj][3], ss2);
#pragma unroll
        for (int hg = 0; hg < NH; ++hg) {
          const int nh = nw + hg * 64;   // first feature of this head
          if (nh >= a.N) continue;
          const bool rot = nh < a.rope_ncols, scl = nh < a.rope_qcols;
          float o1[8], o2[8];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int x = t * 4 + r;
              float a1 = acc[hg * 4 + t][j][r], a2 = acc[hg * 4 + 2 + t][j][r];
              if (sx) {
                const int n1 = nh + fq * 8 + x;
                a1 = (a1 * sx[j]) * a.sw[n1 < a.N ? n1 : a.N - 1];
                a2 = (a2 * sx[j]) * a.sw[n1 + 32 < a.N ? n1 + 32 : a.N - 1];
              }
              float y1 = rbf(a1 + b1[hg][x]), y2 = rbf(a2 + b2[hg][x]);
              if (rot) {
                if (scl) { y1 = rbf(y1 * a.rope_scale); y2 = rbf(y2 * a.rope_scale); }
                if (a.rope_mode == 0) {
                  o1[x] = rbf(rbf(y1 * cc1[x]) + rbf(-y2 * ss1[x]));
                  o2[x] = rbf(rbf(y2 * cc2[x]) + rbf(y1 * ss2[x]));
                } else {
                  o1[x] = y1 * cc1[x] + (-y2) * ss1[x];
                  o2[x] = y2 * cc2[x] + y1 * ss2[x];
                }
              } else {
                o1[x] = y1; o2[x] = y2;
              }
            }
          bf16_t* dst = a.C + (size_t)m * a.ldc + nh + fq * 8;
          *reinterpret_cast<uint4*>(dst) = pack8(o1);
          *reinterpret_cast<uint4*>(dst + 32) = pack8(o2);
        }
      }
    }
    return;
  }
  float bias[NP][8];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int n = nw + p * 32 + fq * 8;
    if (a.bias && vec_ok && n < a.N) {
      unpack8(*reinterpret_cast<const uint4*>(a.bias + n), bias[p]);
    } else {
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int nn = (n + x) < a.N ? (n + x) : a.N - 1;
        bias[p][x] = a.bias ? bf2f(a.bias[nn]) : 0.f;
      }
    }
  }
  constexpr int JB = JBR;   // tokens per batch of residual loads (a whole wave tile of 16-byte pieces at once spills; the fp8 kernels ask for 1)
#pragma unroll
  for (int j0 = 0; j0 < WTM; j0 += JB) {
    uint4 res[JB][NP];
    if (EPI == EPI_RESID && vec_ok) {
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        const int m = mw + (j0 + jj) * 16 + fr;
        const int mc = m < a.M ? m : a.M - 1;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int n = nw + p * 32 + fq * 8;
          res[jj][p] = *reinterpret_cast<const uint4*>(a.resid + (size_t)mc * a.ldr + (n < a.N ? n : 0));
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      const int j = j0 + jj;
      const int m = mw + j * 16 + fr;
      if (m >= a.M) continue;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int n = nw + p * 32 + fq * 8;
        if (n >= a.N) continue;
        float v[8];
        if (sx) {
          float sw8[8];
          if (n + 8 <= a.N && (reinterpret_cast<uintptr_t>(a.sw) & 15) == 0) {
            const float4 s0 = *reinterpret_cast<const float4*>(a.sw + n), s1 = *reinterpret_cast<const float4*>(a.sw + n + 4);
            sw8[0] = s0.x; sw8[1] = s0.y; sw8[2] = s0.z; sw8[3] = s0.w; sw8[4] = s1.x; sw8[5] = s1.y; sw8[6] = s1.z; sw8[7] = s1.w;
          } else {
#pragma unroll
            for (int x = 0; x < 8; ++x) sw8[x] = a.sw[n + x < a.N ? n + x : a.N - 1];
          }
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[t * 4 + r] = rbf((acc[2 * p + t][j][r] * sx[j]) * sw8[t * 4 + r] + bias[p][t * 4 + r]);
        } else {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[t * 4 + r] = rbf(acc[2 * p + t][j][r] + bias[p][t * 4 + r]);
        }
        if (EPI == EPI_RESID) {
          if (vec_ok) {
            float rr[8];
            unpack8(res[jj][p], rr);
#pragma unroll
            for (int x = 0; x < 8; ++x) v[x] = rbf(v[x] + rr[x]);
          } else {
#pragma unroll
            for (int x = 0; x < 8; ++x)
              if (n + x < a.N) v[x] = rbf(v[x] + bf2f(a.resid[(size_t)m * a.ldr + n + x]));
          }
        }
        if (EPI == EPI_GELU_ERF) {
#pragma unroll
          for (int x = 0; x < 8; ++x) v[x] = rbf(gelu_erf_f(v[x]));
        }
        if (EPI == EPI_GELU_ESM) {
#pragma unroll
          for (int x = 0; x < 8; ++x) v[x] = gelu_lut ? gelu_esm_lut(v[x], gelu_lut, gelu_neg_off) : gelu_esm_chain(v[x]);
        }
        if (vec_ok) {
          *reinterpret_cast<uint4*>(a.C + (size_t)m * a.ldc + n) = pack8(v);
        } else {
#pragma unroll
          for (int x = 0; x < 8; ++x)
            if (n + x < a.N) a.C[(size_t)m * a.ldc + n + x] = f2bf(v[x]);
        }
      }
    }
  }
}

// ESM-GELU epilogue, fast form (persistent kernel, W rows in wperm_row order, sparse table image): bias add + bf16 rounding as packed
// pairs, then per output  address = pattern << 1  and one ds_read_u16 -- no per-element range test, no closed forms: the lane keeps
// the packed minimum / maximum of the magnitudes it looked up and the wave votes once at the end; `false` = some value of this wave's
// tile lay outside 2^-25 <= |x| < 2^7 (or was not finite) and the caller redoes the tile with the select form (same stores again).
// An out-of-range pattern reads a stale tile byte or beyond the allocation (LDS returns 0 there): discarded either way.
// ~6 VALU + 1 gather per output instead of ~22 + 1.
typedef unsigned short gelu_u16x2 __attribute__((ext_vector_type(2)));
template <int WTN, int WTM>
__device__ __forceinline__ bool gemm_epilogue_perm_gelu_fast(const PcyGemmArgs& a, f32x4 (&acc)[WTN][WTM], int mw, int nw, int fr, int fq,
                                                             const char* lut_pos) {
  constexpr int NP = WTN / 2;
  const char* lut0 = lut_pos - 2 * (GELU_LUT_E0 << 7);   // entry of pattern b at lut0 + 2 b
  float bias[NP][8];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int n = nw + p * 32 + fq * 8;
    uint4 bb = make_uint4(0, 0, 0, 0);   // (`cond ? *p : zero` becomes a load through a selected POINTER, the zero in scratch)
    if (a.bias && n + 8 <= a.N) bb = *reinterpret_cast<const uint4*>(a.bias + n);
    unpack8(bb, bias[p]);
  }
  gelu_u16x2 mn = {0xffff, 0xffff}, mx = {0, 0};
#pragma unroll
  for (int j = 0; j < WTM; ++j) {
    const int m = mw + j * 16 + fr;
    if (m >= a.M) continue;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int n = nw + p * 32 + fq * 8;
      if (n + 8 > a.N) continue;
      uint32_t o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = q >> 1, r = (q & 1) * 2;
        const uint32_t w = pack_bf(acc[2 * p + t][j][r] + bias[p][t * 4 + r], acc[2 * p + t][j][r + 1] + bias[p][t * 4 + r + 1]);
        const gelu_u16x2 mag = __builtin_bit_cast(gelu_u16x2, w & 0x7fff7fffu);
        mn = __builtin_elementwise_min(mn, mag);
        mx = __builtin_elementwise_max(mx, mag);
        const uint32_t lo = *reinterpret_cast<const uint16_t*>(lut0 + ((w & 0xffffu) << 1));
        const uint32_t hi = *reinterpret_cast<const uint16_t*>(lut0 + ((w >> 15) & 0x1fffeu));
        o[q] = lo | (hi << 16);
      }
      *reinterpret_cast<uint4*>(a.C + (size_t)m * a.ldc + n) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
  constexpr unsigned LO = GELU_LUT_E0 << 7, HI = (GELU_LUT_E0 << 7) + GELU_LUT_HALF;
  const bool ok = mn[0] >= LO && mn[1] >= LO && mx[0] < HI && mx[1] < HI;
  return __all(ok);
}

// SwiGLU epilogue over W rows in wperm_row_swiglu order: nwp = first PACKED row of the wave's tile (or of its half), the lane's
// outputs are features nwp/2 + Q*32 + fq*8 + [0, 8) of token mw + j*16 + fr.  Rounding as gemm_epilogue's SwiGLU branch.
template <int WTN, int WTM>
__device__ __forceinline__ void gemm_epilogue_perm_swiglu(const PcyGemmArgs& a, f32x4 (&acc)[WTN][WTM], int mw, int nwp, int fr, int fq,
                                                          const float* sx = nullptr) {
  const int Nout = a.N >> 1;
  const bool vec_ok = (a.ldc % 8 == 0) && (Nout % 8 == 0) && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0;
#pragma unroll
  for (int Q = 0; Q < WTN / 4; ++Q) {
    const int nout = (nwp >> 1) + Q * 32 + fq * 8;
    if (nout >= Nout) continue;
    float swg[8], swu[8];
    if (sx) {
      const int rg = nwp + Q * 64 + (fq >> 1) * 32 + (fq & 1) * 8;   // packed row of the lane's first gate feature; up = +16
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        swg[x] = a.sw[rg + x < a.N ? rg + x : a.N - 1];
        swu[x] = a.sw[rg + 16 + x < a.N ? rg + 16 + x : a.N - 1];
      }
    }
#pragma unroll
    for (int j = 0; j < WTM; ++j) {
      const int m = mw + j * 16 + fr;
      if (m >= a.M) continue;
      float o[8];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int x = hf * 4 + r;
          float ga = acc[Q * 4 + hf * 2][j][r], ua = acc[Q * 4 + hf * 2 + 1][j][r];
          if (sx) { ga = (ga * sx[j]) * swg[x]; ua = (ua * sx[j]) * swu[x]; }
          const float g = rbf(ga), u = rbf(ua);
          o[x] = rbf(silu_f(g)) * u;
        }
      if (vec_ok && nout + 8 <= Nout) {
        *reinterpret_cast<uint4*>(a.C + (size_t)m * a.ldc + nout) = pack8(o);
      } else {
#pragma unroll
        for (int x = 0; x < 8; ++x)
          if (nout + x < Nout) a.C[(size_t)m * a.ldc + nout + x] = f2bf(o[x]);
      }
    }
  }
}

#ifndef PCY_STG_SMALL
#define PCY_STG_SMALL 1   // the 128 x 128 / 64 x 64 / K-split kernels stage with buffer loads too (stage_tile, STG = 1)
#endif
template <int EPI, int BK>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(PcyGemmArgs a) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * BM * BK * 2];  // [buf][A|W][128*BK] bf16: 64 KiB (BK=64) / 32 KiB (BK=32)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs, so give each XCD a
  // contiguous run of tiles that share A/W panels in its private L2 (bijective remap).
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  int m0, n0;
  tile_origin(a, tile, m0, n0);

  f32x4 acc[4][4];  // [n-sub][m-sub]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = a.K / BK;
  constexpr int TILE_B = BM * BK * 2;  // 16 KiB; buffer b: A at b*2*TILE_B, W right after it

  stage_tile<BK, 128, 4, 0, PCY_STG_SMALL>(a.A, a.lda, m0, a.M, 0, smem, wave, lane);
  stage_tile<BK, 128, 4, 0, PCY_STG_SMALL>(a.W, a.K, n0, a.N, 0, smem + TILE_B, wave, lane);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* Acur = smem + cur * 2 * TILE_B;
    const char* Wcur = Acur + TILE_B;
    // (the split + priority schedule of gemm_kernel_big buys nothing here -- two workgroups per CU already overlap: ESM wo 638 vs 641,
    // Llama qkv T512 482 vs 530 TFLOP/s -- and was removed)
    if (kt + 1 < nk) {
      char* Anext = smem + (cur ^ 1) * 2 * TILE_B;
      stage_tile<BK, 128, 4, 0, PCY_STG_SMALL>(a.A, a.lda, m0, a.M, (kt + 1) * BK, Anext, wave, lane);
      stage_tile<BK, 128, 4, 0, PCY_STG_SMALL>(a.W, a.K, n0, a.N, (kt + 1) * BK, Anext + TILE_B, wave, lane);
    }
#pragma unroll
    for (int kb = 0; kb < BK / 32; ++kb) {
      bf16x8 xf[4], wf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = lds_frag<BK>(Acur, wm * 64 + j * 16 + fr, kb * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = lds_frag<BK>(Wcur, wn * 64 + i * 16 + fr, kb * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  if constexpr (EPI == EPI_GELU_ESM) {
    gelu_lut_to_lds<GEMM_THREADS>(smem);
    gemm_epilogue<EPI>(a, acc, m0, n0, wm, wn, fr, fq, reinterpret_cast<const uint16_t*>(smem));
    return;
  }
  gemm_epilogue<EPI>(a, acc, m0, n0, wm, wn, fr, fq);
}

// ------------------------------------------------------------------------------------------------
// Small-problem variant: 64 x 64 tile (2x2 waves of 32 x 32), same mainloop and the same per-element accumulation order as
// gemm_kernel, so a row's result does not depend on which of the two ran (the packing invariance of the ESM encoder holds).
// For GEMMs whose 128 x 128 tiling leaves most of the chip idle: one 1024-residue protein through ESM2-650M gives fc2
// 9 x 10 = 90 tiles for 512 slots (80 us of a 217 us layer); 64 x 64 tiles give 340.
template <int EPI>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel_small(PcyGemmArgs a) {
  constexpr int BK = 64, TM = 64, TN = 64, WTM = 2, WTN = 2;
  constexpr int TILE_A = TM * BK * 2, TILE_W = TN * BK * 2;   // 8 KiB each
  __shared__ __attribute__((aligned(1024))) char smem[2 * (TILE_A + TILE_W)];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  int m0, n0;
  tile_origin<TM, TN>(a, tile, m0, n0);
  f32x4 acc[WTN][WTM];
#pragma unroll
  for (int i = 0; i < WTN; ++i)
#pragma unroll
    for (int j = 0; j < WTM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = a.K / BK;
  stage_tile<BK, TM, 4, 0, PCY_STG_SMALL>(a.A, a.lda, m0, a.M, 0, smem, wave, lane);
  stage_tile<BK, TN, 4, 0, PCY_STG_SMALL>(a.W, a.K, n0, a.N, 0, smem + TILE_A, wave, lane);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* Acur = smem + cur * (TILE_A + TILE_W);
    const char* Wcur = Acur + TILE_A;
    if (kt + 1 < nk) {
      char* Anext = smem + (cur ^ 1) * (TILE_A + TILE_W);
      stage_tile<BK, TM, 4, 0, PCY_STG_SMALL>(a.A, a.lda, m0, a.M, (kt + 1) * BK, Anext, wave, lane);
      stage_tile<BK, TN, 4, 0, PCY_STG_SMALL>(a.W, a.K, n0, a.N, (kt + 1) * BK, Anext + TILE_A, wave, lane);
    }
#pragma unroll
    for (int kb = 0; kb < BK / 32; ++kb) {
      bf16x8 xf[WTM], wf[WTN];
#pragma unroll
      for (int j = 0; j < WTM; ++j) xf[j] = lds_frag<BK>(Acur, wm * WTM * 16 + j * 16 + fr, kb * 4 + fq);
#pragma unroll
      for (int i = 0; i < WTN; ++i) wf[i] = lds_frag<BK>(Wcur, wn * WTN * 16 + i * 16 + fr, kb * 4 + fq);
#pragma unroll
      for (int i = 0; i < WTN; ++i)
#pragma unroll
        for (int j = 0; j < WTM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  if constexpr (EPI == EPI_GELU_ESM) {
    gelu_lut_to_lds<GEMM_THREADS>(smem);
    gemm_epilogue<EPI, WTN, WTM, false>(a, acc, m0, n0, wm, wn, fr, fq, reinterpret_cast<const uint16_t*>(smem));
    return;
  }
  gemm_epilogue<EPI, WTN, WTM, false>(a, acc, m0, n0, wm, wn, fr, fq);
}

// ------------------------------------------------------------------------------------------------
// Split-K for small M (Llama prefill of one 512-token prompt: o / down / qkv projections have 128-192 output tiles for 256
// CUs, `down` with K = 14336): grid = tiles x splits, every workgroup accumulates its K range with the mainloop of
// gemm_kernel and writes the fp32 partial tile; a second, element-wise kernel adds the partials in split order (fixed ->
// deterministic) and applies the epilogue.  Measured at M = 512: down 154 -> see DESIGN.md.
__global__ __launch_bounds__(GEMM_THREADS) void gemm_splitk_kernel(PcyGemmArgs a, int splits, int k_per_split) {
  constexpr int BK = 64;
  __shared__ __attribute__((aligned(1024))) char smem[2 * 2 * BM * BK * 2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles = gridDim.x / splits;
  const int split = blockIdx.x / tiles, bid = blockIdx.x % tiles;
  const int xq = tiles >> 3, xr = tiles & 7, xcd = bid & 7;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  int m0, n0;
  tile_origin(a, tile, m0, n0);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int kbeg = split * k_per_split;
  const int nk = k_per_split / BK;
  constexpr int TILE_B = BM * BK * 2;
  stage_tile<BK, 128, 4, 0, PCY_STG_SMALL>(a.A, a.lda, m0, a.M, kbeg, smem, wave, lane);
  stage_tile<BK, 128, 4, 0, PCY_STG_SMALL>(a.W, a.K, n0, a.N, kbeg, smem + TILE_B, wave, lane);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* Acur = smem + cur * 2 * TILE_B;
    const char* Wcur = Acur + TILE_B;
    if (kt + 1 < nk) {
      char* Anext = smem + (cur ^ 1) * 2 * TILE_B;
      stage_tile<BK, 128, 4, 0, PCY_STG_SMALL>(a.A, a.lda, m0, a.M, kbeg + (kt + 1) * BK, Anext, wave, lane);
      stage_tile<BK, 128, 4, 0, PCY_STG_SMALL>(a.W, a.K, n0, a.N, kbeg + (kt + 1) * BK, Anext + TILE_B, wave, lane);
    }
#pragma unroll
    for (int kb = 0; kb < BK / 32; ++kb) {
      bf16x8 xf[4], wf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = lds_frag<BK>(Acur, wm * 64 + j * 16 + fr, kb * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = lds_frag<BK>(Wcur, wn * 64 + i * 16 + fr, kb * 4 + fq);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // fp32 partial tile: lane holds D[n = fq*4 + r][m = fr] of each 16x16 tile -> one 16-byte store per tile
  float* ws = a.splitk_ws + (size_t)split * a.M * a.N;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + fr;
    if (m >= a.M) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + fq * 4;
      if (n + 3 < a.N) *reinterpret_cast<f32x4*>(ws + (size_t)m * a.N + n) = acc[i][j];
      else
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < a.N) ws[(size_t)m * a.N + n + r] = acc[i][j][r];
    }
  }
}

// sum of the partials in split order + bias (+ residual); 4 consecutive features per thread
template <int EPI>
__global__ __launch_bounds__(256) void gemm_splitk_epilogue(PcyGemmArgs a, int splits) {
  const size_t quads = (size_t)a.M * (a.N / 4);
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (size_t)gridDim.x * 256) {
    const int m = (int)(q / (a.N / 4)), n = (int)(q % (a.N / 4)) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(a.splitk_ws + (size_t)m * a.N + n);
    for (int s_ = 1; s_ < splits; ++s_) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(a.splitk_ws + ((size_t)s_ * a.M + m) * a.N + n);
      v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
    }
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[r] = rbf(v[r] + (a.bias ? bf2f(a.bias[n + r]) : 0.f));
      if (EPI == EPI_RESID) o[r] = rbf(o[r] + bf2f(a.resid[(size_t)m * a.ldr + n + r]));
    }
    *reinterpret_cast<uint2*>(a.C + (size_t)m * a.ldc + n) = make_uint2(pack_bf(o[0], o[1]), pack_bf(o[2], o[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// Large-M variant: 256 x 256 x 64 tile, 8 waves (4 along tokens x 2 along features), 64 tokens x 128 features per wave
// = 4 x 8 MFMA tiles.  Per 32-k step a wave reads 12 fragments for 32 MFMAs (the 128x128 kernel: 8 for 16), which takes
// the LDS read time from 100 % to 75 % of the MFMA time of a CU, and the 256-wide tile halves the L2 traffic per flop.
// 128 KiB of LDS (2 stages) -> one workgroup per CU, two waves per SIMD.
// F8 = true: the same kernel over OCP e4m3 operands.  A 128-byte LDS row then holds 128 k (instead of 64 bf16), the byte
// geometry of staging, swizzle and fragment reads is unchanged, and a lane's two 16-byte fragments of a row (pieces fq and
// 4 + fq) feed ONE v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) -- A and B use the same k -> slot map, which is all
// a dot product needs.  Twice the k per MFMA issue slot at the MX rate = 2x the bf16 flops for the same LDS / global bytes.
// The per-token / per-output-row dequantisation scales are applied to the fp32 accumulators in front of the epilogue.
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x8 cat_frag(const bf16x8& lo, const bf16x8& hi) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 a = __builtin_bit_cast(i32x4, lo), b = __builtin_bit_cast(i32x4, hi);
  return (i32x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// STG: 1 = lock-step k-loop over `buffer_load ... lds` staging (the fp8 kernels), 2 = the software-pipelined k-loop (bf16).
// NOPERM: W rows in their natural order + the 8-byte epilogue (SwiGLU, fp8 by default); otherwise wperm_row order + 16-byte epilogue.
template <int EPI, bool F8 = false, bool NOPERM = false, int STG = 2>
__global__ __launch_bounds__(512) void gemm_kernel_big(PcyGemmArgs a) {
  constexpr int BK = 64, TBM = 256, TBN = 256, WTM = 4, WTN = 8, NW = 8;
  constexpr int TILE_A = TBM * BK * 2, TILE_W = TBN * BK * 2;   // 32 KiB each
  constexpr bool PERM = !NOPERM;
  constexpr int SWW = PERM ? (EPI == EPI_SWIGLU ? 2 : 1) : 0;
  static_assert(STG == 1 || (STG == 2 && !F8), "bf16: pipelined loop; fp8: lock-step loop");
  extern __shared__ __attribute__((aligned(1024))) char smem_dyn[];
  char* smem = smem_dyn;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  int m0, n0;
  tile_origin<TBM, TBN>(a, tile, m0, n0);

  f32x4 acc[WTN][WTM];
#pragma unroll
  for (int i = 0; i < WTN; ++i)
#pragma unroll
    for (int j = 0; j < WTM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // leading dimensions and k offsets in 2-byte units (an e4m3 row of K bytes = K/2 units; one stage = 128 bytes of k)
  const int nk = F8 ? a.K / 128 : a.K / BK;
  constexpr int kbeg = 0;
  const int lda = F8 ? a.lda / 2 : a.lda, ldw = F8 ? a.K / 2 : a.K;
  stage_tile<BK, TBM, NW, 0, STG>(a.A, lda, m0, a.M, kbeg, smem, wave, lane);
  stage_tile<BK, TBN, NW, SWW, STG>(a.W, ldw, n0, a.N, kbeg, smem + TILE_A, wave, lane);
  const int fr = lane & 15, fq = lane >> 4;
  if constexpr (STG == 2 && !F8) {
    // Software-pipelined k-loop: the fragments of a 32-k sub-step are read from LDS while the MFMAs of the sub-step BEFORE it run --
    // the row of W fragments i is reloaded (for the next sub-step) as soon as its four MFMAs are issued, the four x fragments are
    // double buffered (+16 VGPRs) -- so no sub-step starts with every wave of the CU waiting for its first ds_read_b128 (the
    // lock-step loop: all eight waves issue 12 reads each right behind the barrier while the matrix pipes idle).  One barrier per
    // k-step as before, placed between the two sub-steps: behind it stage kt+1 has landed (read by the second sub-step's prefetch)
    // and every wave is done reading stage kt's buffer, into which the DMA of stage kt+2 then goes.
    if (nk > 1) {
      stage_tile<BK, TBM, NW, 0, STG>(a.A, lda, m0, a.M, kbeg + BK, smem + TILE_A + TILE_W, wave, lane);
      stage_tile<BK, TBN, NW, SWW, STG>(a.W, ldw, n0, a.N, kbeg + BK, smem + TILE_A + TILE_W + TILE_A, wave, lane);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // stage 0 (this wave's first eight pieces) has landed, stage 1 may be in flight
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
    bf16x8 xa[WTM], xb[WTM], wf[WTN];
    auto xrow = [&](int j) { return wm * WTM * 16 + j * 16 + fr; };
    auto wrow = [&](int i) { return wn * WTN * 16 + (PERM ? wperm<EPI>(i, fr) : i * 16 + fr); };
#pragma unroll
    for (int j = 0; j < WTM; ++j) xa[j] = lds_frag<BK>(smem, xrow(j), fq);
#pragma unroll
    for (int i = 0; i < WTN; ++i) wf[i] = lds_frag<BK, SWW>(smem + TILE_A, wrow(i), fq);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const char* Acur = smem + cur * (TILE_A + TILE_W);
      const char* Wcur = Acur + TILE_A;
      // (the last k-step prefetches from its own buffer: unused, but the loop body stays free of branches)
      const char* Anxt = smem + (kt + 1 < nk ? cur ^ 1 : cur) * (TILE_A + TILE_W);
      const char* Wnxt = Anxt + TILE_A;
      // sub-step 0: MFMAs on (xa, wf); prefetch sub-step 1 of this stage into (xb, wf)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < WTN; ++i) {
#pragma unroll
        for (int j = 0; j < WTM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xa[j], acc[i][j], 0, 0, 0);
        wf[i] = lds_frag<BK, SWW>(Wcur, wrow(i), 4 + fq);
        if (i < WTM) xb[i] = lds_frag<BK>(Acur, xrow(i), 4 + fq);
      }
      // (left alone hipcc sinks all twelve reads behind the 29th MFMA: pin 4 MFMAs : 2 or 1 reads)
#pragma unroll
      for (int i = 0; i < WTM; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
#pragma unroll
      for (int i = WTM; i < WTN; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      __builtin_amdgcn_s_setprio(0);
      __syncthreads();   // (lgkmcnt(0): this wave's reads of stage kt are complete; vmcnt(0): its pieces of stage kt+1 have landed)
      // (the eight pieces in front of the sub-step; spreading them over the MFMA rows with sched_group_barrier(0x020) made hipcc
      // cluster them -- each needs its own M0 -- and undo the MFMA : read interleave below)
      // sub-step 1: MFMAs on (xb, wf); prefetch sub-step 0 of stage kt+1 into (xa, wf).  The DMA of stage kt+2 goes out in two halves: A
      // here, W behind the fourth MFMA row (all eight pieces right behind the barrier: +2.7 % instead of +7 % on the Llama prefill)
      if (kt + 2 < nk) stage_tile<BK, TBM, NW, 0, STG>(a.A, lda, m0, a.M, kbeg + (kt + 2) * BK, const_cast<char*>(Acur), wave, lane);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < WTM; ++i) {
#pragma unroll
        for (int j = 0; j < WTM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xb[j], acc[i][j], 0, 0, 0);
        wf[i] = lds_frag<BK, SWW>(Wnxt, wrow(i), fq);
        xa[i] = lds_frag<BK>(Anxt, xrow(i), fq);
      }
#pragma unroll
      for (int i = 0; i < WTM; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
      __builtin_amdgcn_s_setprio(0);   // the W half of stage kt+2 behind the first four rows (the A half went out in front of them)
      if (kt + 2 < nk) stage_tile<BK, TBN, NW, SWW, STG>(a.W, ldw, n0, a.N, kbeg + (kt + 2) * BK, const_cast<char*>(Wcur), wave, lane);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = WTM; i < WTN; ++i) {
#pragma unroll
        for (int j = 0; j < WTM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xb[j], acc[i][j], 0, 0, 0);
        wf[i] = lds_frag<BK, SWW>(Wnxt, wrow(i), fq);
      }
#pragma unroll
      for (int i = WTM; i < WTN; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
      __builtin_amdgcn_s_setprio(0);
    }
  } else {
    // fp8: lock-step loop, the whole next stage requested at the top of the k-step (the split + priority schedule of the bf16 loop
    // measured no better here: 1609-1615 vs 1619-1620 TFLOP/s)
    static_assert(F8, "the lock-step loop is the fp8 form");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      const char* Acur = smem + cur * (TILE_A + TILE_W);
      const char* Wcur = Acur + TILE_A;
      if (kt + 1 < nk) {
        char* Anext = smem + (cur ^ 1) * (TILE_A + TILE_W);
        stage_tile<BK, TBM, NW, 0, STG>(a.A, lda, m0, a.M, kbeg + (kt + 1) * BK, Anext, wave, lane);
        stage_tile<BK, TBN, NW, SWW, STG>(a.W, ldw, n0, a.N, kbeg + (kt + 1) * BK, Anext + TILE_A, wave, lane);
      }
      i32x8 xf[WTM];
#pragma unroll
      for (int j = 0; j < WTM; ++j)
        xf[j] = cat_frag(lds_frag<BK>(Acur, wm * WTM * 16 + j * 16 + fr, fq), lds_frag<BK>(Acur, wm * WTM * 16 + j * 16 + fr, 4 + fq));
#pragma unroll
      for (int i = 0; i < WTN; ++i) {
        const int wr = wn * WTN * 16 + (PERM ? wperm<EPI>(i, fr) : i * 16 + fr);
        const i32x8 wf = cat_frag(lds_frag<BK, SWW>(Wcur, wr, fq), lds_frag<BK, SWW>(Wcur, wr, 4 + fq));
#pragma unroll
        for (int j = 0; j < WTM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, xf[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
      __syncthreads();
    }
  }
  if constexpr (F8) {
    // dequantise: lane holds D[n = fq*4 + r][m = fr] of tile (i, j); (acc * sa[m]) * sw[n], in this order (oracle/fp8_ref.py)
    float sx[WTM];
#pragma unroll
    for (int j = 0; j < WTM; ++j) {
      const int m = m0 + wm * WTM * 16 + j * 16 + fr;
      sx[j] = a.sa[m < a.M ? m : a.M - 1];
    }
    if constexpr (!PERM) {
#pragma unroll
    for (int i = 0; i < WTN; ++i) {
      const int n = n0 + wn * WTN * 16 + i * 16 + fq * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sw = a.sw[n + r < a.N ? n + r : a.N - 1];
#pragma unroll
        for (int j = 0; j < WTM; ++j) acc[i][j][r] = (acc[i][j][r] * sx[j]) * sw;
      }
    }
    }
    // the epilogue in two halves of 64 features: its up-front residual loads (64 VGPRs for the whole 64 x 128 wave tile)
    // plus the scale registers would spill
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (PERM && EPI == EPI_SWIGLU) gemm_epilogue_perm_swiglu<4, WTM>(a, reinterpret_cast<f32x4(&)[4][WTM]>(acc[h * 4]), m0 + wm * 64, n0 + wn * 128 + h * 64, fr, fq, sx);
      else if constexpr (PERM) gemm_epilogue_perm<EPI, 4, WTM, 1>(a, reinterpret_cast<f32x4(&)[4][WTM]>(acc[h * 4]), m0 + wm * 64, n0 + wn * 128 + h * 64, fr, fq, nullptr, sx);
      else gemm_epilogue<EPI, 4, WTM, false>(a, reinterpret_cast<f32x4(&)[4][WTM]>(acc[h * 4]), m0, n0 + wn * 64 + h * 64, wm, wn, fr, fq);
    }
    return;
  }
  if constexpr (EPI == EPI_GELU_ESM) {
    if constexpr (PERM) {
      // the fast table epilogue of the persistent kernel (sparse image: positive half over stage buffer 0, negative half over buffer 1:
      // every buffer is dead here); the select form as the fallback
      __syncthreads();
      gelu_lut_to_lds_sparse<512>(smem);
      const bool vec16 = (a.ldc % 8 == 0) && (a.N % 8 == 0) && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 &&
                         (a.bias == nullptr || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) && !a.gelu_select;
      bool done = false;
      if (vec16) done = gemm_epilogue_perm_gelu_fast<WTN, WTM>(a, acc, m0 + wm * 64, n0 + wn * 128, fr, fq, smem);
      if (!done)
        gemm_epilogue_perm<EPI, WTN, WTM>(a, acc, m0 + wm * 64, n0 + wn * 128, fr, fq, reinterpret_cast<const uint16_t*>(smem), nullptr, GELU_SPARSE_NEG);
    } else {
      gelu_lut_to_lds<512>(smem);
      gemm_epilogue<EPI, WTN, WTM>(a, acc, m0, n0, wm, wn, fr, fq, reinterpret_cast<const uint16_t*>(smem));
    }
    return;
  }
  if constexpr (PERM && EPI == EPI_SWIGLU) gemm_epilogue_perm_swiglu<WTN, WTM>(a, acc, m0 + wm * 64, n0 + wn * 128, fr, fq);
  else if constexpr (PERM) gemm_epilogue_perm<EPI, WTN, WTM>(a, acc, m0 + wm * 64, n0 + wn * 128, fr, fq);
  else gemm_epilogue<EPI, WTN, WTM>(a, acc, m0, n0, wm, wn, fr, fq);
}

}  // namespace
// which kernel a launch went to (test instrumentation behind pcy_debug_dispatch_count: parity tests assert that they reach the
// kernel they claim to test)
unsigned long long g_pcy_dispatch[PCY_DISPATCH_N] = {};
namespace {

// PCY_GEMM_PERM=0 (read per call: interleaved A/B in one process): the 256 x 256 kernels with the natural W row order and the
// 8-byte epilogue of the first rounds; same bits either way
// PCY_GEMM_PERM = mask of the epilogues that use the permuted order: 1 STORE, 2 RESID, 4 ESM GELU, 8 SwiGLU, 16 the fp8 kernels.
// Measured in one process, interleaved (tools/archive/ab_esm_env.py, tools/archive/ab_llama_env.py): ESM2-650M encoder at 25 x 1026 tokens 535 -> 553
// proteins/s with 1 | 2 | 4 (K = 1280: a tile is 20 k-steps, the epilogue a fifth of it); Llama-3-8B prefill (K = 4096 / 14336) 64 x 450
// tokens 1048 TFLOP/s with every mask, one 512-token prompt 10.43 -> 10.53 ms with the SwiGLU form (8), fp8 1804 -> 1793 TFLOP/s with
// 16 -- so the default is 7: the bf16 STORE / RESID / ESM-GELU epilogues.
#ifndef PCY_GEMM_PERM_DEFAULT
#define PCY_GEMM_PERM_DEFAULT 7
#endif
inline bool gemm_noperm(int epi, bool f8) {
  const char* e = getenv("PCY_GEMM_PERM");
  const int mask = e ? atoi(e) : PCY_GEMM_PERM_DEFAULT;
  const int bit = epi == EPI_STORE ? 1 : epi == EPI_RESID ? 2 : epi == EPI_GELU_ESM ? 4 : epi == EPI_SWIGLU ? 8 : 0;
  return !(mask & bit) || (f8 && !(mask & 16));
}
// (History of the 256 x 256 k-loop, all interleaved A/Bs, now the only forms left: LDS-DMA pieces as `buffer_load ... lds` with a scalar k
// offset instead of global_load_lds -- ESM2-650M batch 41.2 -> 40.6 ms, 16-40 VGPRs fewer; the software-pipelined loop with the stage
// DMA in two halves -- Llama-3-8B prefill 64 x 450 tokens 1068 -> 1143 TFLOP/s, ESM batch 40.3 -> 39.55 ms.  The lock-step bf16 loops,
// the persistent tile loop for the ESM-GELU GEMM (39.60 vs 39.36 ms per batch), the per-XCD tile queue, the epilogue through LDS and
// the 256 x 256 K-split form were measured slower or equal and REMOVED in round 4; see DESIGN.md for their numbers.)
template <int EPI>
void launch_fp8(hipStream_t s, const PcyGemmArgs& a) {
  constexpr int smem = 2 * (256 + 256) * 64 * 2;
  const int tiles_big = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  PcyGemmArgs b = a;
  const int tn = (a.N + 255) / 256;
  long gnb = (5L << 19) / ((long)256 * a.K);
  if (gnb < 2) gnb = 2;
  b.gn = (int)(gnb > tn ? tn : gnb);
  ++g_pcy_dispatch[PCY_DISPATCH_GEMM_FP8];
  if (gemm_noperm(EPI, true)) {
    static PcyLdsAttr lds_n;
    lds_n.ensure(&gemm_kernel_big<EPI, true, true, 1>, smem, 0);
    hipLaunchKernelGGL((gemm_kernel_big<EPI, true, true, 1>), dim3(tiles_big), dim3(512), smem, s, b);
    return;
  }
  static PcyLdsAttr lds;
  lds.ensure(&gemm_kernel_big<EPI, true, false, 1>, smem, 0);
  hipLaunchKernelGGL((gemm_kernel_big<EPI, true, false, 1>), dim3(tiles_big), dim3(512), smem, s, b);
}

template <int EPI, bool NOPERM>
void launch_big_variant(hipStream_t s, const PcyGemmArgs& b, dim3 grid, int smem) {
  static PcyLdsAttr lds;
  lds.ensure(&gemm_kernel_big<EPI, false, NOPERM, 2>, smem, 0);
  hipLaunchKernelGGL((gemm_kernel_big<EPI, false, NOPERM, 2>), grid, dim3(512), smem, s, b);
}

#include "pcy_gemm_mid.h"

#ifndef PCY_BIG_MIN_N
#define PCY_BIG_MIN_N 1280   // with the split + priority schedule the 256 x 256 kernel also wins at N = K = 1280 (ESM wo: 167-177 -> 152-157 us)
#endif
template <int EPI>
void launch(hipStream_t s, const PcyGemmArgs& a) {
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  if constexpr (EPI != EPI_GELU_ERF) {
    if (a.mid_cfg > 0 && launch_mid<EPI>(s, a, a.mid_cfg)) { ++g_pcy_dispatch[PCY_DISPATCH_GEMM_MID]; return; }
  }
  constexpr int big_min_m = 2048;
  // 256x256 tiles pay off where the mainloop dominates (measured, M = 32832: qkv 734 -> 804, fc2 827 -> 911 TFLOP/s);
  // for N = K = 1280 they do not; the ESM GELU epilogue is a wash since the rational erf (fc1 679 vs 700)
  // Below big_min_m rows the choice follows how well each tiling fills the chip: 128x128 tiles run two per CU (512 slots),
  // 256x256 one per CU with a ~14 % better mainloop.  Llama-3-8B gate/up at one 512-token prompt: 896 small tiles = 1.75
  // rounds vs 224 big tiles = one round -> 148 -> 113 us (prefill 12.7 -> 11.6 ms); at M = 256 or 768 the small tiles fit better.
  bool size_ok = a.M >= big_min_m;
  if (!size_ok && a.M >= 256) {
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128), t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    const double e128 = (double)t128 / (double)(((t128 + 511) / 512) * 512) * a.M / (double)(((a.M + 127) / 128) * 128);
    const double e256 = (double)t256 / (double)(((t256 + 255) / 256) * 256) * a.M / (double)(((a.M + 255) / 256) * 256);
    size_ok = e256 * 1.14 > e128;
  }
  const bool big_ok = size_ok && EPI != EPI_GELU_ERF &&
                      (a.N >= PCY_BIG_MIN_N || (a.N >= 256 && a.K >= 2560));
  if (big_ok) {
    constexpr int smem = 2 * (256 + 256) * 64 * 2;
    const int tiles_big = ((a.M + 255) / 256) * ((a.N + 255) / 256);
    PcyGemmArgs b = a;
    const int tn = (a.N + 255) / 256;
    long gnb = (5L << 19) / ((long)256 * a.K * 2);
    if (gnb < 2) gnb = 2;
    // The 32 workgroups an XCD runs at a time walk K together, so what its L2 has to hold is the current k-window of their panels, not
    // whole panels: 8 row tiles x 4 column tiles share 12 panels where 16 x 2 share 18.  Measured (interleaved, PCY_GEMM_GN): ESM2-650M
    // batch 40.2 -> 39.7 ms with groups of 5 (N = 1280: the whole row) instead of 4 / 2; Llama-3-8B prefill 64 x 450 tokens 375.8 ->
    // 372.3 ms with 4 instead of 2 (5, 6: neutral; 16: -4 %); one 512-token prompt (two row tiles) is best left at 2.
    if (a.M >= 2048) {
      long want = a.K <= 2560 ? 6 : 4;   // (6 vs 5 after the loop changes: 39.5 vs 39.86 ms per ESM batch, medians of 8)
      if (tn <= 6) want = tn;
      if (gnb < want) gnb = want;
    }
    b.gn = (int)(gnb > tn ? tn : gnb);
    // row order (PCY_GEMM_PERM mask: 16-byte epilogue over permuted W rows, or the natural order) -- both are shipped and compared by tests
    const bool noperm = gemm_noperm(EPI, false);
    if constexpr (EPI == EPI_GELU_ESM) {
      ++g_pcy_dispatch[PCY_DISPATCH_GEMM_BIG_PERSIST];   // (the slot keeps its name: "the ESM-GELU 256 x 256 kernel")
      b.gelu_select = pcy_off("gelu_fast");   // every wave takes the select form of the table epilogue (tests)
    } else {
      ++g_pcy_dispatch[PCY_DISPATCH_GEMM_BIG];
    }
    if (noperm) launch_big_variant<EPI, true>(s, b, dim3(tiles_big), smem);
    else launch_big_variant<EPI, false>(s, b, dim3(tiles_big), smem);
    return;
  }
  // Mid-M (128 <= M < 2048 and the 256 x 256 tiling under-fills the chip: one 1024-residue protein, one 512-token prompt): the GEMM is
  // ONE round of tiles, so the tile shape is chosen for CU fill and the k-loop for latency -- gemm_kernel_mid (pcy_gemm_mid.h), shape
  // by a cost model fitted on tools/archive/bench_gemm_mid.py (cold weights): rounds(tiles / 256 CUs) x tile area / efficiency, efficiency
  // 0.6 / 0.7 / 0.8 / 1.0 for 8 waves of 32 x 32 / 32 x 48 / 32 x 64 / 64 x 64.  Same bits as the kernels below (same k order per element).
  // (128 x 96: N = 6144 at M <= 512 -- the Llama qkv projection of one prompt -- is ONE full round of 256 tiles instead of 192 of 128 x 128:
  // 42.5 -> 38.0 us, prefill of a 512-token prompt 9.67 -> 9.55 ms)
  // In the encoder (rocprofv3 kernel time per 1024-residue protein, tools/insitu_ab.py): 5.97 -> 4.78 ms; o + fc2 38.9 -> 22.5 us avg.
  if constexpr (EPI != EPI_GELU_ERF) {
    if (a.mid_cfg == 0 && a.M >= 128) {
      static const int cand[4][4] = {{5, 128, 64, 6}, {3, 128, 128, 8}, {6, 256, 128, 10}, {14, 128, 96, 7}};   // id, TM, TN, 10 x efficiency
      int best = 0;
      double best_cost = 0;
      for (int i = 0; i < 4; ++i) {
        const int fn = cand[i][2] / 2 / 16;   // feature tiles per wave (two waves along N in all of them)
        if (a.rope_cos != nullptr && fn % 4 != 0) continue;
        if (EPI == EPI_SWIGLU && fn % 2 != 0) continue;
        const long t = (long)((a.M + cand[i][1] - 1) / cand[i][1]) * ((a.N + cand[i][2] - 1) / cand[i][2]);
        const double cost = (double)((t + 255) / 256) * cand[i][1] * cand[i][2] * 10.0 / cand[i][3];
        if (!best || cost < best_cost) { best = cand[i][0]; best_cost = cost; }
      }
      if (best && launch_mid<EPI>(s, a, best)) { ++g_pcy_dispatch[PCY_DISPATCH_GEMM_MID]; return; }
    }
  }
  // fewer than 192 tiles of 128 x 128 (of 512 slots): 64 x 64 tiles -- four times the workgroups, same arithmetic per element
  if constexpr (EPI != EPI_SWIGLU) {
    if (tiles < 192 && a.rope_cos == nullptr) {
      const int tiles64 = ((a.M + 63) / 64) * ((a.N + 63) / 64);
      PcyGemmArgs b = a;
      const int tn64 = (a.N + 63) / 64;
      long g64 = (5L << 19) / ((long)64 * a.K * 2);
      if (g64 < 4) g64 = 4;
      b.gn = (int)(g64 > tn64 ? tn64 : g64);
      ++g_pcy_dispatch[PCY_DISPATCH_GEMM_64];
      hipLaunchKernelGGL((gemm_kernel_small<EPI>), dim3(tiles64), dim3(GEMM_THREADS), 0, s, b);
      return;
    }
  }
  ++g_pcy_dispatch[PCY_DISPATCH_GEMM_128];
  hipLaunchKernelGGL((gemm_kernel<EPI, 64>), dim3(tiles), dim3(GEMM_THREADS), 0, s, a);
}

}  // namespace

void pcy_gemm_prepare(hipStream_t s) {
  static bool lut_built[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !lut_built[dev]) {   // once per device, on the caller's stream (ordered before the first use)
    hipLaunchKernelGGL(gelu_lut_build_kernel, dim3((GELU_LUT_N + 255) / 256), dim3(256), 0, s);
    lut_built[dev] = true;
  }
}

void pcy_launch_gemm(hipStream_t s, const PcyGemmArgs& a0) {
  if (a0.M <= 0 || a0.N <= 0) return;
  PcyGemmArgs a = a0;
  if (a.mid_cfg == 0) {   // tools / tests: force a gemm_kernel_mid configuration -- "5" for every GEMM, or per shape "3840x1280=6,1280x5120=5" (N x K)
    const char* e = getenv("PCY_GEMM_MID");
    if (e && !strchr(e, 'x')) a.mid_cfg = atoi(e);
    else if (e) {
      char key[48];
      snprintf(key, sizeof(key), "%dx%d=", a.N, a.K);
      const char* hit = strstr(e, key);
      if (hit && (hit == e || hit[-1] == ',')) a.mid_cfg = atoi(hit + strlen(key));
      else a.mid_cfg = -1;   // (-1: the launcher's pre-round-4 choice, for the shapes the map does not name)
    }
  }
  const int tiles_n = (a.N + BN - 1) / BN;
  const long panel = (long)BN * a.K * 2;
  long gn = (5L << 19) / panel;           // 2.5 MiB of W panels per group
  if (gn < 4) gn = 4;                      // long-K panels: a narrower group loses more A reuse than it saves (measured)
  a.gn = (int)(gn < 1 ? 1 : (gn > tiles_n ? tiles_n : gn));
  if (a.fp8) {   // e4m3 operands: the 256 x 256 kernel for every M (callers check K % 128, lda % 16)
    switch (a.epi) {
      case EPI_RESID: launch_fp8<EPI_RESID>(s, a); break;
      case EPI_SWIGLU: launch_fp8<EPI_SWIGLU>(s, a); break;
      default: launch_fp8<EPI_STORE>(s, a); break;
    }
    return;
  }
  // split-K: few tiles, long K, plain / residual epilogue, a workspace supplied by the caller
  if (a.mid_cfg <= 0 && a.splitk_ws && (a.epi == EPI_STORE || a.epi == EPI_RESID) && a.rope_cos == nullptr && a.N % 4 == 0 && a.ldc % 4 == 0 &&
      (a.resid == nullptr || a.ldr % 4 == 0)) {
    const int tiles = ((a.M + BM - 1) / BM) * tiles_n;
    // the split count depends on (N, K) only -- sized for four row tiles (one 512-token prompt) -- so that a row's result
    // does not change with the number of rows in the batch (tests: batch invariance); callers pass the workspace only
    // for M <= 1024.  (Tried and removed: a rule sized for 256 x 256 tiles with twice the splits -- the projections 44.8 / 29.8 / 80.3
    // -> 42 / 30.5 / 67.3 us, but twice the partial sums make the finish launches 6.5 + 2 x 10.7 -> 9.8 + 2 x 21.1 us: prefill 10.66 ->
    // 11.01 ms; gemm_kernel_mid tiles for the partial GEMMs: 10.19 -> 10.33 / 11.2 ms.)
    const int tiles_ref = 4 * tiles_n;
    int splits = 1;
    while (splits < 8 && tiles_ref * splits * 2 <= 512 && a.K % (splits * 2 * 64) == 0 && a.K / (splits * 2) >= 512) splits *= 2;
    // (a 2-way split buys less than its finish launch costs: Llama-3-8B qkv at one 512-token prompt 41.3 + 6.4 us split vs 41.9 us as one
    // gemm_kernel_mid launch -- and the rule stays a function of (N, K) only)
    if (splits == 2) splits = 1;
    if (splits > 1 && (size_t)splits * a.M * a.N * 4 <= a.splitk_ws_bytes) {
      ++g_pcy_dispatch[PCY_DISPATCH_GEMM_SPLITK];
      hipLaunchKernelGGL(gemm_splitk_kernel, dim3(tiles * splits), dim3(GEMM_THREADS), 0, s, a, splits, a.K / splits);
      const int eb = (int)(((size_t)a.M * (a.N / 4) + 255) / 256);
      // residual epilogue + the RMSNorm that follows in ONE finish launch (one workgroup per row) where the caller asks for it
      if (a.epi == EPI_RESID && a.next_rms_w && a.next_xn && a.fused_next && a.bias == nullptr && a.ldc == a.N && a.ldr == a.N &&
          !pcy_off("finish_norm") &&
          pcy_launch_splitk_finish_norm(s, a.splitk_ws, splits, a.M, a.N, a.resid, a.C, a.next_rms_w, a.next_xn, a.rms_eps, a.rms_cast)) {
        *a.fused_next = 1;
        return;
      }
      if (a.epi == EPI_RESID) hipLaunchKernelGGL(gemm_splitk_epilogue<EPI_RESID>, dim3(eb < 2048 ? eb : 2048), dim3(256), 0, s, a, splits);
      else hipLaunchKernelGGL(gemm_splitk_epilogue<EPI_STORE>, dim3(eb < 2048 ? eb : 2048), dim3(256), 0, s, a, splits);
      return;
    }
  }
  switch (a.epi) {
    case EPI_STORE: launch<EPI_STORE>(s, a); break;
    case EPI_RESID: launch<EPI_RESID>(s, a); break;
    case EPI_GELU_ERF: launch<EPI_GELU_ERF>(s, a); break;
    case EPI_GELU_ESM: {
      pcy_gemm_prepare(s);
      launch<EPI_GELU_ESM>(s, a);
      break;
    }
    case EPI_SWIGLU: launch<EPI_SWIGLU>(s, a); break;
  }
}
