// fp32 arithmetic for the callers of the reference that never call `.bfloat16()`:
//   /root/reference/examples/paper_analyses/protpep_qa_scores.py:55-58 (the loop that defines BASELINE configs[4]),
//   /root/reference/scripts/qa_filter_captions.py:17-18, /root/reference/scripts/caption_bulk.py:72-73.
// Their model holds fp32 weights and every torch op runs in fp32; the bf16 engine cannot reproduce that to better than 1e-2.  This
// file is the fp32 operator family those paths need -- encoder, pooler, projectors, splice, decoder PREFILL (QA scoring and retrieval
// read one forward pass; cached decode stays bf16-only) -- written for correctness first:
//   * Linear on the f32-input matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation in ascending k == an fmaf
//     chain; 155 TFLOP/s peak, 1/16 of the bf16 rate), 128 x 128 x 16 tiles, register-staged double buffer;
//   * attention as one wave per (sequence, head, query row): exact softmax in fp32 (scores in LDS), GQA, causal / key-keep masks;
//   * LayerNorm / RMSNorm / rotary / embedding / ESM token-dropout embedding / pooling / SiLU-mul as plain fp32 kernels.
// The Python side (procyon_amd/engine_f32.py) strings them together exactly as the oracle's fp32 evaluation does (oracle/esm_ref.py,
// oracle/llama_ref.py); tests hold every op and the stacks to <= 1e-4 of it.
#include "pcy_internal.h"
#include "../../include/pcy.h"

namespace {

constexpr int NT = 256;

// ------------------------------------------------------------------------------------------------ Linear
// C[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ resid);  act 0 none, 1 gelu (x * 0.5 * (1 + erf(x / sqrt 2)): nn.GELU and the ESM gelu are
// the same function in fp32).  Operands are fed swapped (W rows as the MFMA A operand) so a lane ends up with 4 consecutive output
// features of one token: 16-byte stores.
constexpr int F_TM = 128, F_TN = 128, F_BK = 16, F_LD = F_BK + 1;   // LDS row stride 17 floats: conflict-free ds_read_b32 fragments
__global__ __launch_bounds__(NT) void linear_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                        const float* __restrict__ bias, const float* resid, int ldr, float* C, int ldc,
                                                        int M, int N, int K, int act) {
  __shared__ float As[2][F_TM * F_LD], Ws[2][F_TN * F_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves of 64 tokens x 64 features
  const int tiles_n = (N + F_TN - 1) / F_TN;
  const int m0 = (blockIdx.x / tiles_n) * F_TM, n0 = (blockIdx.x % tiles_n) * F_TN;
  // staging: 128 rows x 16 floats = 512 float4 per operand, two per thread
  const int sr = tid >> 2, sc = (tid & 3) * 4;              // rows sr and sr + 64, columns sc .. sc+3
  const float* ap[2]; const float* wp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ra = m0 + sr + 64 * i; ra = ra < M ? ra : M - 1;
    int rw = n0 + sr + 64 * i; rw = rw < N ? rw : N - 1;
    ap[i] = A + (size_t)ra * lda + sc;
    wp[i] = W + (size_t)rw * K + sc;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra[2], rw[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { ra[i] = *reinterpret_cast<const float4*>(ap[i] + k0); rw[i] = *reinterpret_cast<const float4*>(wp[i] + k0); }
  };
  auto sstore = [&](int b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float* a = &As[b][(sr + 64 * i) * F_LD + sc];
      float* w = &Ws[b][(sr + 64 * i) * F_LD + sc];
      a[0] = ra[i].x; a[1] = ra[i].y; a[2] = ra[i].z; a[3] = ra[i].w;
      w[0] = rw[i].x; w[1] = rw[i].y; w[2] = rw[i].z; w[3] = rw[i].w;
    }
  };
  const int nk = K / F_BK;
  gload(0);
  sstore(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int b = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * F_BK);
#pragma unroll
    for (int k4 = 0; k4 < F_BK / 4; ++k4) {
      float xf[4], wf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = As[b][(wm * 64 + j * 16 + fr) * F_LD + k4 * 4 + fq];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = Ws[b][(wn * 64 + i * 16 + fr) * F_LD + k4 * 4 + fq];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) sstore(b ^ 1);
    __syncthreads();
  }
  // lane holds D[feature n = fq*4 + r][token m = fr] of tile (i, j)
  const bool vec = (ldc % 4 == 0) && (N % 4 == 0) && (resid == nullptr || ldr % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                   (resid == nullptr || (reinterpret_cast<uintptr_t>(resid) & 15) == 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wm * 64 + j * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wn * 64 + i * 16 + fq * 4;
      if (n >= N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n + r < N ? n + r : N - 1;
        float y = acc[i][j][r] + (bias ? bias[nn] : 0.f);
        if (act == 1) y = y * 0.5f * (1.0f + erff(y / 1.4142135623730951f));
        v[r] = y;
      }
      if (vec && n + 3 < N) {
        if (resid) {
          const float4 rr = *reinterpret_cast<const float4*>(resid + (size_t)m * ldr + n);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        *reinterpret_cast<float4*>(C + (size_t)m * ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (n + r < N) C[(size_t)m * ldc + n + r] = v[r] + (resid ? resid[(size_t)m * ldr + n + r] : 0.f);
      }
    }
  }
}

// small-K / unaligned fallback: one thread per output, fmaf chain over k (K % 16 != 0 or rows not 16-byte aligned)
__global__ __launch_bounds__(NT) void linear_f32_naive_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                              const float* __restrict__ bias, const float* resid, int ldr, float* C, int ldc,
                                                              int M, int N, int K, int act) {
  const size_t idx = (size_t)blockIdx.x * NT + threadIdx.x;
  if (idx >= (size_t)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx % N);
  float y = 0.f;
  for (int k = 0; k < K; ++k) y = fmaf(A[(size_t)m * lda + k], W[(size_t)n * K + k], y);
  y += bias ? bias[n] : 0.f;
  if (act == 1) y = y * 0.5f * (1.0f + erff(y / 1.4142135623730951f));
  if (resid) y += resid[(size_t)m * ldr + n];
  C[(size_t)m * ldc + n] = y;
}

// ------------------------------------------------------------------------------------------------ norms
// LayerNorm as torch's fp32 kernel: mean, then biased variance around it, (x - mean) * rstd * w + b.  One workgroup per row.
__global__ __launch_bounds__(NT) void layernorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                           float* __restrict__ y, int d, float eps) {
  __shared__ float red[NT / 64];
  const float* xr = x + (size_t)blockIdx.x * d;
  float s = 0.f;
  for (int k = threadIdx.x; k < d; k += NT) s += xr[k];
  const float mean = block_sum<NT>(s, red) / (float)d;
  float v = 0.f;
  for (int k = threadIdx.x; k < d; k += NT) { const float t = xr[k] - mean; v += t * t; }
  const float rstd = rsqrtf(block_sum<NT>(v, red) / (float)d + eps);
  float* yr = y + (size_t)blockIdx.x * d;
  for (int k = threadIdx.x; k < d; k += NT) yr[k] = (xr[k] - mean) * rstd * w[k] + b[k];
}
// RMSNorm (HF Llama, fp32 model: the cast orders coincide): w * (x * rsqrt(mean(x^2) + eps))
__global__ __launch_bounds__(NT) void rmsnorm_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int d, float eps) {
  __shared__ float red[NT / 64];
  const float* xr = x + (size_t)blockIdx.x * d;
  float s = 0.f;
  for (int k = threadIdx.x; k < d; k += NT) s += xr[k] * xr[k];
  const float r = rsqrtf(block_sum<NT>(s, red) / (float)d + eps);
  float* yr = y + (size_t)blockIdx.x * d;
  for (int k = threadIdx.x; k < d; k += NT) yr[k] = w[k] * (xr[k] * r);
}

// ------------------------------------------------------------------------------------------------ rotary, embeddings, SiLU-mul
// heads [col0, col0 + nh * dh) of row t: x <- (x * prescale?) * cos[pos] + rotate_half(x * prescale?) * sin[pos]   (tables fp32 [n_pos, dh])
__global__ __launch_bounds__(NT) void rope_f32_kernel(float* __restrict__ buf, int ld, int col0, int nh, int dh, const int32_t* __restrict__ pos,
                                                      const float* __restrict__ cos_t, const float* __restrict__ sin_t, float prescale) {
  const int t = blockIdx.x, p = pos[t], half = dh / 2;
  float* row = buf + (size_t)t * ld + col0;
  for (int i = threadIdx.x; i < nh * half; i += NT) {
    const int h = i / half, e = i % half;
    float x1 = row[h * dh + e], x2 = row[h * dh + e + half];
    if (prescale != 0.f) { x1 *= prescale; x2 *= prescale; }
    const float c1 = cos_t[(size_t)p * dh + e], s1 = sin_t[(size_t)p * dh + e];
    const float c2 = cos_t[(size_t)p * dh + e + half], s2 = sin_t[(size_t)p * dh + e + half];
    row[h * dh + e] = x1 * c1 + (-x2) * s1;
    row[h * dh + e + half] = x2 * c2 + x1 * s2;
  }
}
// out[r] = soft_map[r] >= 0 ? soft[soft_map[r]] : table[ids[r]]   (token embedding + soft-token splice in one gather)
__global__ __launch_bounds__(NT) void embed_f32_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids, const float* __restrict__ soft,
                                                       const int32_t* __restrict__ soft_map, float* __restrict__ out, int d) {
  const int r = blockIdx.x;
  const int sm = soft_map ? soft_map[r] : -1;
  const float* src = sm >= 0 ? soft + (size_t)sm * d : table + (size_t)ids[r] * d;
  for (int k = threadIdx.x; k < d; k += NT) out[(size_t)r * d + k] = src[k];
}
// HF EsmEmbeddings with token_dropout (oracle/esm_ref.py::embed): <mask> rows zero, x * 0.88 / (1 - observed mask ratio), pads zero
__global__ __launch_bounds__(NT) void esm_embed_f32_kernel(const float* __restrict__ table, const int32_t* __restrict__ toks, const int32_t* __restrict__ cu,
                                                           float* __restrict__ out, int d, int mask_pads) {
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
  const float denom = 1.0f - nmask / (mask_pads ? nkeep : (float)len);
  for (int j = blockIdx.y; j < len; j += gridDim.y) {
    const int t = toks[t0 + j];
    const bool zero = (t == 32) || (mask_pads && t == 1);
    for (int k = threadIdx.x; k < d; k += NT)
      out[(size_t)(t0 + j) * d + k] = zero ? 0.f : table[(size_t)t * d + k] * (1.0f - 0.15f * 0.8f) / denom;
  }
}
// dst[r] += src[rows[r]]  (sum over the L+1 hidden states at the [PROT] rows: ret_token_access='all', model_unified.py:560-563)
__global__ __launch_bounds__(NT) void acc_rows_f32_kernel(const float* __restrict__ src, int lds_, const int32_t* __restrict__ rows, float* __restrict__ dst, int d) {
  const int r = blockIdx.x;
  const float* s = src + (size_t)rows[r] * lds_;
  for (int k = threadIdx.x; k < d; k += NT) dst[(size_t)r * d + k] += s[k];
}
__global__ __launch_bounds__(NT) void silu_mul_f32_kernel(const float* __restrict__ g, const float* __restrict__ u, float* __restrict__ o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
    const float x = g[i];
    o[i] = (x / (1.0f + expf(-x))) * u[i];
  }
}
// ProteinPooler (/root/reference/procyon/model/esm.py:131-173) over token ranges, as pcy_pool: mode 0 mean, 1 mean of x[1:-1], 2 max.
// One workgroup per (protein, 256-feature slab); sequential over the tokens (fp32 sum in token order).
__global__ __launch_bounds__(NT) void pool_f32_kernel(const float* __restrict__ h, int d, const int32_t* __restrict__ seg, const int32_t* __restrict__ rng,
                                                      int mode, float* __restrict__ out) {
  const int p = blockIdx.x, c = blockIdx.y * NT + threadIdx.x;
  if (c >= d) return;
  const int r0 = seg[p], r1 = seg[p + 1];
  float acc = mode == 2 ? -INFINITY : 0.f;
  int cnt = 0;
  for (int r = r0; r < r1; ++r) {
    int st = rng[2 * r], ln = rng[2 * r + 1];
    if (mode == 1) {
      if (r == r0) { st += 1; ln -= 1; }
      if (r == r1 - 1) ln -= 1;
    }
    for (int j = 0; j < ln; ++j) {
      const float v = h[(size_t)(st + j) * d + c];
      if (mode == 2) acc = fmaxf(acc, v);
      else if (v == v) { acc += v; ++cnt; }
    }
  }
  out[(size_t)p * d + c] = mode == 2 ? acc : acc / (float)cnt;
}

// ------------------------------------------------------------------------------------------------ attention
// One wave per (query row i, head h, sequence q).  Packed rows: token t0 + j of sequence q.  keys j < (causal ? i + 1 : len), dropped
// where keep[t0 + j] == 0.  scores = (q . k) * scale in LDS, exact softmax, o = P . V.  A row without any kept key writes zeros (only
// pad query rows: the reference's value there -- a uniform average over every key -- is never read on the prefill-only path).
__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ q, int ldq, int qcol0, const float* __restrict__ k, int ldk, int kcol0,
                                                      const float* __restrict__ v, int ldv, int vcol0, float* __restrict__ o, int ldo,
                                                      const int32_t* __restrict__ cu, const uint8_t* __restrict__ keep, int H, int Hkv, int dh,
                                                      int causal, float scale) {
  extern __shared__ float smem_f[];
  const int i = blockIdx.x, h = blockIdx.y, sq = blockIdx.z, lane = threadIdx.x;
  const int t0 = cu[sq], len = cu[sq + 1] - t0;
  if (i >= len) return;
  float* qs = smem_f;            // [dh]
  float* sc = smem_f + dh;       // [nkeys]
  const int hk = h / (H / Hkv);
  const float* qrow = q + (size_t)(t0 + i) * ldq + qcol0 + h * dh;
  for (int e = lane; e < dh; e += 64) qs[e] = qrow[e];
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const int nkeys = causal ? i + 1 : len;
  float mx = -INFINITY;
  for (int j = lane; j < nkeys; j += 64) {
    float s = -INFINITY;
    if (!keep || keep[t0 + j]) {
      const float* kr = k + (size_t)(t0 + j) * ldk + kcol0 + hk * dh;
      float a = 0.f;
      for (int e = 0; e < dh; e += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(kr + e);
        a = fmaf(qs[e], kv.x, a); a = fmaf(qs[e + 1], kv.y, a); a = fmaf(qs[e + 2], kv.z, a); a = fmaf(qs[e + 3], kv.w, a);
      }
      s = a * scale;
    }
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float* orow = o + (size_t)(t0 + i) * ldo + h * dh;
  if (mx == -INFINITY) {
    for (int e = lane; e < dh; e += 64) orow[e] = 0.f;
    return;
  }
  float sum = 0.f;
  for (int j = lane; j < nkeys; j += 64) { const float p = expf(sc[j] - mx); sc[j] = p; sum += p; }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int e = lane; e < dh; e += 64) {
    float a = 0.f;
    const float* vc = v + (size_t)t0 * ldv + vcol0 + hk * dh + e;
    for (int j = 0; j < nkeys; ++j) a = fmaf(sc[j] * inv, vc[(size_t)j * ldv], a);
    orow[e] = a;
  }
}

// Decode attention in fp32: one wave per (head h, row b); the new token's query against the first `nkeys` slots of the row's cache
// (token-major [B][Tmax][Hkv dh]; every slot is attended: the reference passes no mask after the prefill, model_unified.py:769, :887).
__global__ __launch_bounds__(64) void attn_dec_f32_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ kc, const float* __restrict__ vc,
                                                          int ldkv, int Tmax, float* __restrict__ o, int ldo, int H, int Hkv, int dh, int nkeys,
                                                          float scale) {
  extern __shared__ float smem_f[];
  const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  float* qs = smem_f;            // [dh]
  float* sc = smem_f + dh;       // [nkeys]
  const int hk = h / (H / Hkv);
  const float* qrow = q + (size_t)b * ldq + h * dh;
  for (int e = lane; e < dh; e += 64) qs[e] = qrow[e];
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const float* kb = kc + (size_t)b * Tmax * ldkv + hk * dh;
  const float* vb = vc + (size_t)b * Tmax * ldkv + hk * dh;
  float mx = -INFINITY;
  for (int j = lane; j < nkeys; j += 64) {
    const float* kr = kb + (size_t)j * ldkv;
    float a = 0.f;
    for (int e = 0; e < dh; e += 4) {
      const float4 kv = *reinterpret_cast<const float4*>(kr + e);
      a = fmaf(qs[e], kv.x, a); a = fmaf(qs[e + 1], kv.y, a); a = fmaf(qs[e + 2], kv.z, a); a = fmaf(qs[e + 3], kv.w, a);
    }
    const float s_ = a * scale;
    sc[j] = s_;
    mx = fmaxf(mx, s_);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < nkeys; j += 64) { const float p = expf(sc[j] - mx); sc[j] = p; sum += p; }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  float* orow = o + (size_t)b * ldo + h * dh;
  for (int e = lane; e < dh; e += 64) {
    float a = 0.f;
    for (int j = 0; j < nkeys; ++j) a = fmaf(sc[j] * inv, vb[(size_t)j * ldkv + e], a);
    orow[e] = a;
  }
}

int launch_ok(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { pcy_set_error("kernel launch failed in %s: %s", what, hipGetErrorString(e)); return 3; }
  return 0;
}

}  // namespace

extern "C" {

int pcy_f32_linear(pcy_ctx* c, const float* A, int lda, const float* W, const float* bias, const float* resid, int ldr, float* C, int ldc,
                   int M, int N, int K, int act) {
  if (M <= 0 || N <= 0) return 0;
  if (act != 0 && act != 1) { pcy_set_error("pcy_f32_linear: act %d (0 none, 1 gelu)", act); return 1; }
  hipStream_t s = pcy_ctx_stream(c);
  const bool fast = K % F_BK == 0 && lda % 4 == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W)) & 15) == 0;
  if (fast) {
    const int tiles = ((M + F_TM - 1) / F_TM) * ((N + F_TN - 1) / F_TN);
    hipLaunchKernelGGL(linear_f32_kernel, dim3(tiles), dim3(NT), 0, s, A, lda, W, bias, resid, ldr, C, ldc, M, N, K, act);
  } else {
    const size_t n = (size_t)M * N;
    hipLaunchKernelGGL(linear_f32_naive_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, s, A, lda, W, bias, resid, ldr, C, ldc, M, N, K, act);
  }
  return launch_ok("pcy_f32_linear");
}
int pcy_f32_layernorm(pcy_ctx* c, const float* x, const float* w, const float* b, float* y, int rows, int d, float eps) {
  if (rows > 0) hipLaunchKernelGGL(layernorm_f32_kernel, dim3(rows), dim3(NT), 0, pcy_ctx_stream(c), x, w, b, y, d, eps);
  return launch_ok("pcy_f32_layernorm");
}
int pcy_f32_rmsnorm(pcy_ctx* c, const float* x, const float* w, float* y, int rows, int d, float eps) {
  if (rows > 0) hipLaunchKernelGGL(rmsnorm_f32_kernel, dim3(rows), dim3(NT), 0, pcy_ctx_stream(c), x, w, y, d, eps);
  return launch_ok("pcy_f32_rmsnorm");
}
int pcy_f32_rope(pcy_ctx* c, float* buf, int ld, int col0, int nh, int dh, const int32_t* pos, const float* cos_t, const float* sin_t, int ntok,
                 float prescale) {
  if (dh % 2) { pcy_set_error("pcy_f32_rope: odd head_dim"); return 1; }
  if (ntok > 0) hipLaunchKernelGGL(rope_f32_kernel, dim3(ntok), dim3(NT), 0, pcy_ctx_stream(c), buf, ld, col0, nh, dh, pos, cos_t, sin_t, prescale);
  return launch_ok("pcy_f32_rope");
}
int pcy_f32_embed(pcy_ctx* c, const float* table, const int32_t* ids, const float* soft, const int32_t* soft_map, float* out, int rows, int d) {
  if (rows > 0) hipLaunchKernelGGL(embed_f32_kernel, dim3(rows), dim3(NT), 0, pcy_ctx_stream(c), table, ids, soft, soft_map, out, d);
  return launch_ok("pcy_f32_embed");
}
int pcy_f32_esm_embed(pcy_ctx* c, const float* table, const int32_t* tokens, const int32_t* cu, int nseq, int max_len, float* out, int d, int mask_pads) {
  if (nseq > 0) {
    int sl = max_len < 64 ? max_len : 64;
    if (sl < 1) sl = 1;
    hipLaunchKernelGGL(esm_embed_f32_kernel, dim3(nseq, sl), dim3(NT), 0, pcy_ctx_stream(c), table, tokens, cu, out, d, mask_pads);
  }
  return launch_ok("pcy_f32_esm_embed");
}
int pcy_f32_silu_mul(pcy_ctx* c, const float* gate, const float* up, float* out, size_t n) {
  if (n > 0) {
    const size_t nb = (n + NT - 1) / NT;
    hipLaunchKernelGGL(silu_mul_f32_kernel, dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(NT), 0, pcy_ctx_stream(c), gate, up, out, n);
  }
  return launch_ok("pcy_f32_silu_mul");
}
int pcy_f32_acc_rows(pcy_ctx* c, const float* src, int ld, const int32_t* rows, float* dst, int nrows, int d) {
  if (nrows > 0) hipLaunchKernelGGL(acc_rows_f32_kernel, dim3(nrows), dim3(NT), 0, pcy_ctx_stream(c), src, ld, rows, dst, d);
  return launch_ok("pcy_f32_acc_rows");
}
int pcy_f32_pool(pcy_ctx* c, const float* hidden, int d, const int32_t* seg, const int32_t* rng, int nprot, int mode, float* out) {
  if (mode < 0 || mode > 2) { pcy_set_error("pcy_f32_pool: mode %d", mode); return 1; }
  if (nprot > 0) hipLaunchKernelGGL(pool_f32_kernel, dim3(nprot, (d + NT - 1) / NT), dim3(NT), 0, pcy_ctx_stream(c), hidden, d, seg, rng, mode, out);
  return launch_ok("pcy_f32_pool");
}
int pcy_f32_attention(pcy_ctx* c, const float* q, int ldq, int qcol0, const float* k, int ldk, int kcol0, const float* v, int ldv, int vcol0,
                      float* o, int ldo, const int32_t* cu, const uint8_t* keep, int nseq, int max_len, int H, int Hkv, int dh, int causal,
                      float scale) {
  if (dh % 4 || H % Hkv) { pcy_set_error("pcy_f32_attention: head_dim %% 4 and Hkv | H required"); return 1; }
  if ((ldk | kcol0) % 4) { pcy_set_error("pcy_f32_attention: K rows must be 16-byte aligned"); return 1; }
  const size_t smem = (size_t)(dh + max_len) * sizeof(float);
  if (smem > 160 * 1024 - 1024) { pcy_set_error("pcy_f32_attention: %d keys exceed the LDS score buffer", max_len); return 1; }
  if (nseq > 0 && max_len > 0) {
    static PcyLdsAttr lds_attn_f32_kernel;   // (configured once per device and size, refused sizes fail the call)
    if (!lds_attn_f32_kernel.ensure(&attn_f32_kernel, smem)) { pcy_set_error("pcy_f32_attention: the device refused %zu bytes of LDS", smem); return 1; }
    hipLaunchKernelGGL(attn_f32_kernel, dim3(max_len, H, nseq), dim3(64), smem, pcy_ctx_stream(c), q, ldq, qcol0, k, ldk, kcol0, v, ldv, vcol0, o, ldo,
                       cu, keep, H, Hkv, dh, causal, scale);
  }
  return launch_ok("pcy_f32_attention");
}

int pcy_f32_attn_decode(pcy_ctx* c, const float* q, int ldq, const float* kcache, const float* vcache, int ldkv, int Tmax, float* o, int ldo, int B,
                        int H, int Hkv, int dh, int nkeys, float scale) {
  if (dh % 4 || H % Hkv || ldkv % 4) { pcy_set_error("pcy_f32_attn_decode: head_dim %% 4, Hkv | H and 16-byte aligned cache rows required"); return 1; }
  if (nkeys < 1 || nkeys > Tmax) { pcy_set_error("pcy_f32_attn_decode: %d keys outside the cache (%d slots)", nkeys, Tmax); return 1; }
  const size_t smem = (size_t)(dh + nkeys) * sizeof(float);
  if (smem > 160 * 1024 - 1024) { pcy_set_error("pcy_f32_attn_decode: %d keys exceed the LDS score buffer", nkeys); return 1; }
  if (B > 0) {
    static PcyLdsAttr lds_attn_dec_f32_kernel;   // (configured once per device and size, refused sizes fail the call)
    if (!lds_attn_dec_f32_kernel.ensure(&attn_dec_f32_kernel, smem)) { pcy_set_error("pcy_f32_attn_decode: the device refused %zu bytes of LDS", smem); return 1; }
    hipLaunchKernelGGL(attn_dec_f32_kernel, dim3(H, B), dim3(64), smem, pcy_ctx_stream(c), q, ldq, kcache, vcache, ldkv, Tmax, o, ldo, H, Hkv, dh, nkeys, scale);
  }
  return launch_ok("pcy_f32_attn_decode");
}

}  // extern "C"
