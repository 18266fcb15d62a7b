// Host side of libpcy.so: context, workspace, layer loops, hipGraph capture, and the extern "C" ABI
// declared in include/pcy.h.  No torch types; everything is raw device pointers on one HIP stream.
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <dlfcn.h>
#include <vector>

#include "../../include/pcy.h"
#include "pcy_internal.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return fail(2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// for the other translation units of the library (pcy_f32.hip): the thread-local error text behind pcy_last_error
void pcy_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct pcy_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t cap_stream = nullptr;  // capture needs a non-legacy stream; replays go to `stream`
  char* ws = nullptr;
  size_t ws_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // captured decode step
  hipGraphExec_t graph = nullptr;
  static constexpr int GRAPH_KEY_N = 19;
  const void* graph_key[GRAPH_KEY_N] = {};
  int graph_B = 0;
  int graph_mode = 0;
  int graph_kind = 0;                 // 0: decode + greedy pick (pcy_llama_greedy), 1: decode only (pcy_llama_decode_graph), 2: a beam-search step (pcy_llama_beam_steps)
  int n_cu = 0;
  // sticky error word: a cross-workgroup hand-over inside a launch hit its watchdog.  PINNED HOST memory (device-visible): the
  // kernels store it with system scope, every ABI entry looks at it without touching the stream (take_sticky_error below)
  unsigned* xwg_err = nullptr;
  // fused attention + o projection launches of the layered decode step: [0] = step epoch, [64 + 64*l ...] = flags of layer l
  unsigned* ao_sync = nullptr;
  // tagged hand-over vectors of the MLP chain launches: [layer][ffn + d] words, owned by one model geometry at a time
  uint32_t* mc_tags = nullptr;
  const void* mc_tags_model = nullptr;
  size_t mc_tags_words = 0;
  int mc_tags_mode = -1;
  uint64_t layers_fp = 0;              // fingerprint of the weight pointers dev_layers was built from
  PcyLayerWeightsDev* dev_layers = nullptr;   // device copy of the layers' weight pointers (decode_step_kernel)
  // small-batch decode step (pcy_decode_nb.hip): hand-over slots and tag counter PER BATCH SIZE (a slot is rewritten in every step of its
  // own batch size and its counter only advances with those steps, so a word with the current tag can only come from the current step)
  uint32_t* nb_tags[9] = {};
  unsigned* nb_sync = nullptr;        // [16] tag counters, index = batch size
  // mid-batch decode step (pcy_decode_mb.hip): arrival flags [(layers + 1)][pcy_decode_mb_flag_words()] (a flag holds the epoch of the step that
  // raised it) and the epoch word, advanced once per step and never reset
  unsigned* mb_flags = nullptr;
  size_t mb_flags_words = 0;
  unsigned* mb_sync = nullptr;
  uint32_t* op_tags = nullptr;        // tagged `act` vector of pcy_decode_mlp ([ffn] words, its own counter)
  size_t op_tags_words = 0;
  unsigned* smp_hist = nullptr;       // [rows][65536] histogram scratch of the nucleus step (kept all-zero between calls)
  int smp_hist_rows = 0;
  bf16_t* smp_pbits = nullptr;        // [rows][vocab] bits of the probabilities of the sampling step
  size_t smp_pbits_elems = 0;
  char* beam_ws = nullptr;            // scratch of pcy_beam_step (its own allocation: never aliases the decode workspace)
  size_t beam_ws_bytes = 0;

  // Replayed launch chains of the short-sequence encoder (pcy_esm_encode of ONE protein = 230 launches of 5-35 us: on a busy host the
  // launch rate, not the GPU, sets the time -- measured 6.1 to 11 ms per call across gpurun boxes for 4.8 ms of kernels).  A slot is keyed
  // on every pointer and size that is baked into the launches; a key is captured the SECOND time it is seen (a stream of proteins of
  // ever-changing length never pays for captures it would not reuse).
  struct GraphSlot { uint64_t key[16]; hipGraphExec_t exec; unsigned long long stamp; };
  static constexpr int N_GSLOTS = 8;
  GraphSlot gslots[N_GSLOTS] = {};
  unsigned long long gstamp = 0;
  void drop_gslots() {
    for (auto& g : gslots) {
      if (g.exec) hipGraphExecDestroy(g.exec);
      g = GraphSlot{};
    }
  }

  int reserve(size_t bytes) {
    // PCY_DEBUG_POISON_WS=1 (tests): fill the workspace with NaN patterns before every use -- a kernel that reads a workspace
    // location before writing it then fails deterministically instead of depending on what the allocator handed out
    static const bool poison = [] { const char* e = getenv("PCY_DEBUG_POISON_WS"); return e && atoi(e) != 0; }();
    if (poison && ws && bytes <= ws_bytes) HIP_TRY(hipMemsetAsync(ws, 0xFF, ws_bytes, stream));
    if (bytes <= ws_bytes) return 0;
    if (ws) {
      HIP_TRY(hipStreamSynchronize(stream));
      HIP_TRY(hipFree(ws));
      ws = nullptr; ws_bytes = 0;
    }
    drop_graph();
    drop_gslots();
    bytes = align_up(bytes + (bytes >> 3), 1 << 20);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ws), bytes));
    ws_bytes = bytes;
    if (poison) HIP_TRY(hipMemsetAsync(ws, 0xFF, ws_bytes, stream));
    return 0;
  }
  void drop_graph() {
    if (graph) { hipGraphExecDestroy(graph); graph = nullptr; }
  }
};

namespace {

struct Carver {
  char* p; size_t off = 0;
  explicit Carver(char* base) : p(base) {}
  template <typename T> T* take(size_t n) {
    T* r = reinterpret_cast<T*>(p + off);
    off = align_up(off + n * sizeof(T), 256);
    return r;
  }
};

// A watchdog of an in-launch wait has fired since the last look (results of that launch and of everything that consumed them are
// invalid): report it ONCE, from whichever ABI call comes next -- no stream synchronisation, the word lives in host memory.
int take_sticky_error(pcy_ctx* c) {
  if (!c->xwg_err) return 0;
  const unsigned code = *reinterpret_cast<volatile unsigned*>(c->xwg_err);
  if (!code) return 0;
  *reinterpret_cast<volatile unsigned*>(c->xwg_err) = 0;
  return fail(4, "decode kernel: a cross-workgroup dependency wait timed out (code %u): its workgroups were not all resident -- is "
                 "another kernel running on this device? -- results since the last successful pcy_ctx_sync are invalid", code);
}
#define PCY_STICKY(c)                                 \
  do {                                                \
    if (int r_ = take_sticky_error(c)) return r_;     \
  } while (0)

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(3, "kernel launch failed in %s: %s", what, hipGetErrorString(e));
  return 0;
}

// GEMM for any M: MFMA tiles when M is large enough to fill them, the streaming GEMV otherwise
// next_w / next_xn / fused (optional): RMSNorm(C) * next_w -> next_xn inside the projection's finish launch where that exists
// (*fused = 1), see PcyGemmArgs
void linear(hipStream_t s, const bf16_t* A, int lda, const bf16_t* W, const bf16_t* bias, const bf16_t* resid, int ldr,
            bf16_t* C, int ldc, int M, int N, int K, int epi, float* splitk_ws = nullptr, size_t splitk_ws_bytes = 0,
            const bf16_t* next_w = nullptr, bf16_t* next_xn = nullptr, int* fused = nullptr, float rms_eps = 0.f, int rms_cast = 0) {
  if (M <= 8 && epi != EPI_SWIGLU && (resid == nullptr || ldr == ldc)) {
    PcyGemvArgs g{};
    g.W = W; g.x = A; g.y = C; g.bias = bias; g.resid = resid; g.rms_w = nullptr;
    g.N = N; g.K = K; g.B = M; g.ldx = lda; g.ldy = ldc; g.epi = epi;
    pcy_launch_gemv(s, g);
    return;
  }
  PcyGemmArgs a{};
  a.A = A; a.W = W; a.C = C; a.bias = bias; a.resid = resid;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr; a.epi = epi;
  a.splitk_ws = splitk_ws; a.splitk_ws_bytes = splitk_ws_bytes;
  a.next_rms_w = next_w; a.next_xn = next_xn; a.fused_next = fused; a.rms_eps = rms_eps; a.rms_cast = rms_cast;
  pcy_launch_gemm(s, a);
}

// ------------------------------------------------------------------ decode step (enqueue only)
struct DecodeWs { bf16_t *x, *qkv, *ao, *act; };

// PCY_DISABLE=attn_o switches the fused attention + o-projection launch of the layered decode step off (default on; batch 1,
// head_dim 128, d = 4096): 3.35 -> 3.28 ms/token.  PCY_DISABLE=attn_o: the two launches (pcy_switch.h).
bool attn_o_enabled() { return !pcy_off("attn_o"); }
// PCY_DISABLE=mlp_chain: pcy_decode_mlp runs the two GEMV launches instead of mlp_chain_kernel.  PCY_DISABLE=decode_layer: the batch-1
// decode step runs launch by launch (qkv GEMV, attention + o, gate/up, down) instead of one decode_layer_kernel per layer.  Both are
// read on every call: tests compare the paths in one process (bit-identical).
unsigned long long* g_mc_trace = nullptr;
bool mlp_chain_enabled() { return !pcy_off("mlp_chain"); }
bool decode_layer_enabled() { return !pcy_off("decode_layer"); }
// PCY_DISABLE=decode_step: one launch per decoder layer instead of one for all layers (decode_step_kernel); bit-identical
bool decode_step_enabled() { return !pcy_off("decode_step"); }
// (round 3 had everything of a batched-decode layer behind the attention as one launch with grid barriers: bit-identical, 133 us per
// layer against 110 us launch by launch at batch 32 -- a grid barrier under a saturated memory system costs ~10 us, more than the kernel
// boundary it replaces; removed in round 4, numbers in DESIGN.md)
bool qkv_finish_launch() { return pcy_off("attn_qkv_finish"); }   // the qkv K-split finish as its own launch instead of inside the attention's
// PCY_DISABLE=decode_nb: batches of 2..8 rows take the pre-round-5 path (streaming GEMVs below 4 rows, MFMA GEMVs from 4 on) instead of the
// small-batch step (pcy_decode_nb.hip).  PCY_DISABLE=decode_nb_step: its launch-per-stage twin (gemv_stream_kernel + attn_dec_kernel with the
// same column slices) instead of the one launch -- same bits, tests compare the two.
bool decode_nb_enabled() { return !pcy_off("decode_nb"); }
bool decode_nb_step_enabled() { return !pcy_off("decode_nb_step"); }
// PCY_DISABLE=decode_mb_step: batches of 9..32 rows run launch by launch (seven per layer) instead of the mid-batch step (pcy_decode_mb.hip:
// the same work items as phases of ONE launch) -- same bits, tests compare the two.
bool decode_mb_step_enabled() { return !pcy_off("decode_mb_step"); }
// ... for the small-batch step: the exchange costs more there than the K reads it saves until much longer caches (t ~ 800: 2 / 4 rows
// 2.935 / 3.385 ms per step with the split, 2.868 / 3.287 without; t ~ 1540: 3.158 / 3.653 with, 3.180 / 3.632 without)
int decode_xmin_nb(int B) {
  const char* xe = getenv("PCY_AO_XMIN");
  return xe ? atoi(xe) : (B == 2 ? 1536 : 4096);
}
int decode_xmin() {   // cached keys from which the decode attention splits its keys across the slice workgroups (read per call: tests compare)
  const char* xe = getenv("PCY_AO_XMIN");
  return xe ? atoi(xe) : 768;
}

// The mid-batch step is OFF by default (round 6): bit-identical to the launches, and -- once the launches' GEMVs walk their K ranges in rotated
// order (pcy_gemv_kshift) -- no faster: 3.98 / 4.17 ms per step at 10 / 16 rows against 3.88 / 4.02 launch by launch, 4.8 / 5.1 against 4.1 / 4.3
// at 20 / 32 rows (two batch tiles double the activation bytes every CU fetches per weight byte).  Its streaming phases run at the HBM rate
// (G 6.3 TB/s, D 6.1); what it loses is the Q / O phases (16 / 8 tiles per wave against a ring of 7: two memory round trips whatever the
// prefetch) and the flag hops.  PCY_MB_MAX=<rows> (9..32) runs it up to that batch size: tests, tools/bench_decode_mb.py.
int decode_mb_max_rows() {
  const char* e = getenv("PCY_MB_MAX");
  return e ? atoi(e) : 0;
}
int decode_nb_max_rows();
int decode_mode() {
  return (pcy_off("kv_permute") ? 2048 : 0) | (attn_o_enabled() ? 2 : 0) | (decode_layer_enabled() ? 32 : 0) | (decode_step_enabled() ? 64 : 0) | (qkv_finish_launch() ? 128 : 0) |
         (pcy_off("lds_prefetch") ? 256 : 0) | (decode_nb_enabled() ? 512 : 0) | (decode_nb_step_enabled() ? 1024 : 0) | (decode_mb_step_enabled() ? 4096 : 0) | (decode_mb_max_rows() << 16) | ((decode_nb_max_rows() & 3) << 28) |
         (int)((((unsigned)decode_xmin() * 2654435761u) ^ ((unsigned)decode_xmin_nb(2) * 40503u) ^ ((unsigned)decode_xmin_nb(4) * 69069u)) & 0x3fu) << 22;
}
constexpr int AO_MAX_LAYERS = 128, AO_FLAGS = 64;
// geometry of the small-batch step: Llama-3-8B, 256 CUs
// (7 and 8 rows: since the batched launches walk their K ranges in rotated order, ask for their finish loads at once and cut the attention
// into 64-column workgroups from 4 rows on, they take 3.69 / 3.70 ms per step against the fused step's 3.81 / 3.97 -- the fused step stops
// at 6 rows (3.65 against 3.69); PCY_NB_MAX=7 / 8 runs it there all the same: tests, tools)
int decode_nb_max_rows() {
  const char* e = getenv("PCY_NB_MAX");
  const int v = e ? atoi(e) : 6;
  return v < 8 ? v : 8;
}
bool decode_nb_covers(const pcy_ctx* c, const pcy_llama_desc* m, int B) {
  return B >= 2 && B <= decode_nb_max_rows() && m->d == 4096 && m->ffn == 14336 && m->n_heads == 32 && m->n_kv_heads == 8 && m->head_dim == 128 &&
         c->n_cu >= 256 && m->n_layers <= AO_MAX_LAYERS / 2;   // (score-exchange flags: 2 x AO_FLAGS words per layer)
}
// geometry of the mid-batch step (9..32 rows): the same model and chip
bool decode_mb_covers(const pcy_ctx* c, const pcy_llama_desc* m, int B) {
  return B >= 9 && B <= 32 && B <= decode_mb_max_rows() && m->d == 4096 && m->ffn == 14336 && m->n_heads == 32 && m->n_kv_heads == 8 && m->head_dim == 128 && c->n_cu >= 256;
}
// tagged vectors of one layer: act [ffn], qkv [(H + 2 Hkv) dh], attention output [H dh], x after o [d]
size_t tag_words_per_layer(const pcy_llama_desc* m) {
  return (size_t)m->ffn + (size_t)m->d + (size_t)(m->n_heads + 2 * m->n_kv_heads) * m->head_dim + (size_t)m->n_heads * m->head_dim;
}

// device words of the in-launch hand-overs; must run outside stream capture
int ensure_sync_words(pcy_ctx* c) {
  if (!c->n_cu) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount;
  }
  if (!c->xwg_err) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->xwg_err), 64, hipHostMallocMapped | hipHostMallocCoherent));
    memset(c->xwg_err, 0, 64);
  }
  if (!c->ao_sync) {
    // [0] step epoch, [1] tag counter of the decode step's fused launches, [2] tag counter of pcy_decode_mlp, attention->o flags,
    // score-exchange flags
    const size_t bytes = (size_t)(64 + 2 * AO_MAX_LAYERS * AO_FLAGS) * sizeof(unsigned);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ao_sync), bytes));
    HIP_TRY(hipMemset(c->ao_sync, 0, bytes));
  }
  return 0;
}
uint64_t layers_fingerprint(const pcy_llama_desc* m) {   // FNV-1a over every weight pointer the fused decode launches read
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p) { h = (h ^ (uint64_t)(uintptr_t)p) * 1099511628211ull; };
  mix(m->layers);
  for (int l = 0; l < m->n_layers; ++l) {
    const pcy_llama_layer& L = m->layers[l];
    mix(L.ln1); mix(L.wqkv); mix(L.wo); mix(L.ln2); mix(L.wgu); mix(L.wdown);
  }
  return h;
}
int ensure_decode_state(pcy_ctx* c, const pcy_llama_desc* m, int B = 1) {
  if (int r = ensure_sync_words(c)) return r;
  // A tagged word counts as delivered when its tag equals the chain epoch, so the slots must never hold anything but words of
  // earlier chain launches OF THE SAME LAYOUT: another model -> zeroed slots and a restarted epoch (next tag 1).
  // (a change of the launch mix as well: a slot the new mix reads may not have been rewritten for a while)
  // "Another model" is decided on the weight POINTERS, not on the descriptor's address: a new engine of the same geometry may
  // get the address of a freed descriptor, and a descriptor may be mutated in place -- either would leave dev_layers stale.
  const size_t words = (size_t)m->n_layers * (tag_words_per_layer(m) + 32 * 256);   // + the residual stream between the layers, one line per workgroup
  const uint64_t fp = layers_fingerprint(m);
  if (c->mc_tags_model != m || c->mc_tags_words != words || c->mc_tags_mode != decode_mode() || c->layers_fp != fp) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->drop_graph();   // a captured step holds mc_tags / dev_layers as kernel arguments
    if (c->mc_tags) HIP_TRY(hipFree(c->mc_tags));
    c->mc_tags = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->mc_tags), words * 4));
    HIP_TRY(hipMemset(c->mc_tags, 0, words * 4));
    HIP_TRY(hipMemset(c->ao_sync + 1, 0, 4));
    for (auto& t : c->nb_tags) { if (t) HIP_TRY(hipFree(t)); t = nullptr; }
    if (c->dev_layers) HIP_TRY(hipFree(c->dev_layers));
    c->dev_layers = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->dev_layers), (size_t)m->n_layers * sizeof(PcyLayerWeightsDev)));
    std::vector<PcyLayerWeightsDev> lw(m->n_layers);
    for (int l = 0; l < m->n_layers; ++l) {
      const pcy_llama_layer& L = m->layers[l];
      lw[l] = {(const bf16_t*)L.ln1, (const bf16_t*)L.wqkv, (const bf16_t*)L.wo, (const bf16_t*)L.ln2, (const bf16_t*)L.wgu, (const bf16_t*)L.wdown};
    }
    HIP_TRY(hipMemcpy(c->dev_layers, lw.data(), lw.size() * sizeof(PcyLayerWeightsDev), hipMemcpyHostToDevice));
    c->mc_tags_model = m; c->mc_tags_words = words; c->mc_tags_mode = decode_mode(); c->layers_fp = fp;
  }
  if (decode_nb_enabled() && decode_nb_covers(c, m, B) && !c->nb_tags[B]) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!c->nb_sync) {
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->nb_sync), 16 * sizeof(unsigned)));
      HIP_TRY(hipMemset(c->nb_sync, 0, 16 * sizeof(unsigned)));
    }
    const size_t nbw = (size_t)m->n_layers * (pcy_decode_nb_tag_words(B) + pcy_decode_nb_line_words(B));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->nb_tags[B]), nbw * 4));
    HIP_TRY(hipMemset(c->nb_tags[B], 0, nbw * 4));
    HIP_TRY(hipMemset(c->nb_sync + B, 0, 4));   // zeroed slots, next tag 1
  }
  if (decode_mb_covers(c, m, B)) {
    const size_t fw = (size_t)(m->n_layers + 1) * pcy_decode_mb_flag_words();
    if (!c->mb_sync) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->mb_sync), 64));
      HIP_TRY(hipMemset(c->mb_sync, 0, 64));
    }
    if (c->mb_flags_words < fw) {
      HIP_TRY(hipStreamSynchronize(c->stream));
      c->drop_graph();
      if (c->mb_flags) HIP_TRY(hipFree(c->mb_flags));
      c->mb_flags = nullptr; c->mb_flags_words = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->mb_flags), fw * 4));
      HIP_TRY(hipMemset(c->mb_flags, 0, fw * 4));   // (the epoch word keeps counting: a zeroed flag never equals a later epoch)
      c->mb_flags_words = fw;
    }
  }
  return 0;
}

size_t decode_ws_bytes(const pcy_llama_desc* m, int B, int Tmax) {
  const size_t qkvw = (size_t)(m->n_heads + 2 * m->n_kv_heads) * m->head_dim;
  return align_up((size_t)B * m->d * 2, 256) + align_up(B * qkvw * 2, 256) +
         align_up((size_t)B * m->n_heads * m->head_dim * 2, 256) + align_up((size_t)B * m->ffn * 2, 256) +
         align_up((size_t)B * m->n_heads * (Tmax + 1) * 4, 256) + align_up((size_t)B * 64 * 16, 256) +
         align_up((size_t)B * m->d * 2, 256) + (B >= pcy_mfma_min_batch() ? align_up((size_t)8 * B * qkvw * 4, 256) : 0) + 4096;
}

// layers_only (measurement, pcy_llama_decode_layers): the decoder layers without the token embedding (the residual stream is whatever
// the workspace holds) and without lm_head; the hand-over counters are advanced by two one-thread launches instead.
void enqueue_decode(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B, bool layers_only = false) {
  hipStream_t s = c->stream;
  const int d = m->d, H = m->n_heads, Hkv = m->n_kv_heads, dh = m->head_dim, F = m->ffn;
  const int qkvw = (H + 2 * Hkv) * dh;
  Carver cv(c->ws);
  bf16_t* x = cv.take<bf16_t>((size_t)B * d);
  bf16_t* qkv = cv.take<bf16_t>((size_t)B * qkvw);
  bf16_t* ao = cv.take<bf16_t>((size_t)B * H * dh);
  bf16_t* act = cv.take<bf16_t>((size_t)B * F);
  float* scores = cv.take<float>((size_t)B * H * (kv->Tmax + 1));
  cv.take<char>((size_t)B * 64 * 16);                 // pick partials (same carve as enqueue_pick)
  bf16_t* xn = cv.take<bf16_t>((size_t)B * d);        // normalised x for the batched (MFMA) GEMV path
  const size_t sk_bytes = B >= pcy_mfma_min_batch() ? (size_t)8 * B * qkvw * 4 : 0;   // K-split partial sums of the batched GEMVs
  float* sk_ws = sk_bytes ? cv.take<float>(sk_bytes / 4) : nullptr;
  const bool batched_head = B >= pcy_mfma_min_batch() && d % 128 == 0 && F % 128 == 0;   // skinny-MFMA GEMVs, 32 rows per pass over the weights
                                                                                         // (F = 11008, Llama-2-7B / ProCyon-Split: 86 x 128)
  // 2..8 rows: the small-batch step (pcy_decode_nb.hip) -- one launch for all layers, or its launch-per-stage twin; lm_head as before
  const bool nb_on = decode_nb_enabled() && decode_nb_covers(c, m, B) && c->nb_tags[B] && c->nb_sync && c->dev_layers && c->ao_sync && c->xwg_err &&
                     pcy_decode_nb_launchable(c->device, B, kv->Tmax, c->n_cu);
  const bool batched = batched_head && !nb_on;
  // 9..32 rows: the mid-batch step (pcy_decode_mb.hip) -- the batched path's work items as phases of one launch
  const bool mb_step = batched && decode_mb_step_enabled() && decode_mb_covers(c, m, B) && pcy_decode_mb_fits(B, kv->Tmax) && c->mb_flags && c->mb_sync &&
                       c->dev_layers && c->xwg_err && (size_t)(m->n_layers + 1) * pcy_decode_mb_flag_words() <= c->mb_flags_words;
  const bool try_ao = attn_o_enabled() && B == 1 && c->ao_sync && c->xwg_err && m->n_layers <= AO_MAX_LAYERS;
  bool try_layer = decode_layer_enabled() && try_ao && c->mc_tags;   // one launch per decoder layer
  const bool nb_step = nb_on && decode_nb_step_enabled();
  if (layers_only) {
    if (try_ao || nb_step) pcy_launch_bump(s, c->ao_sync);
    if (try_layer) pcy_launch_bump(s, c->ao_sync + 1);
    if (nb_step) pcy_launch_bump(s, c->nb_sync + B);
    if (mb_step) pcy_launch_bump(s, c->mb_sync);
  } else {
    pcy_launch_embed_tokens_dev(s, (const bf16_t*)m->embed, st->next_tok, x, B, d, (try_ao || nb_step) ? c->ao_sync : (mb_step ? c->mb_sync : nullptr),
                                try_layer ? c->ao_sync + 1 : (nb_step ? c->nb_sync + B : nullptr));
  }
  const size_t tag_stride = tag_words_per_layer(m);
  const size_t layer_stride = (size_t)kv->B * Hkv * kv->Tmax * dh;
  bool step_done = false;
  if (nb_step) {
    PcyDecAttnArgs t{};
    t.qkv = qkv; t.ld = qkvw; t.kcache = (bf16_t*)kv->k; t.vcache = (bf16_t*)kv->v;
    t.o = ao; t.ldo = H * dh; t.pos_dev = st->pos; t.cos_t = (const bf16_t*)m->rope_cos; t.sin_t = (const bf16_t*)m->rope_sin;
    t.keep = st->keep; t.ld_keep = kv->Tmax; t.scratch = scores; t.B = B; t.H = H; t.Hkv = Hkv; t.dh = dh; t.Tmax = kv->Tmax;
    t.scale = 1.0f / sqrtf((float)dh);
    t.xflags = c->ao_sync + 64 + AO_MAX_LAYERS * AO_FLAGS;
    PcyAttnBlockArgs bp{};
    bp.x = x; bp.d = d; bp.Nq = qkvw; bp.rms_eps = m->rms_eps; bp.rms_cast = m->rms_cast; bp.epoch = c->nb_sync + B; bp.err = c->xwg_err;
    PcyMlpChainArgs mc{};
    mc.x = x; mc.x_out = x; mc.d = d; mc.F = F; mc.rms_eps = m->rms_eps; mc.rms_cast = m->rms_cast; mc.epoch = c->nb_sync + B; mc.err = c->xwg_err;
    PcyDecodeStepArgs sa{};
    sa.layers = c->dev_layers; sa.n_layers = m->n_layers; sa.kv_layer_stride = layer_stride;
    sa.tags = c->nb_tags[B]; sa.tag_stride = pcy_decode_nb_tag_words(B); sa.xflags_stride = 2 * AO_FLAGS;   // up to 128 attention units per layer
    sa.x_lines = c->nb_tags[B] + (size_t)m->n_layers * sa.tag_stride; sa.x_lines_stride = pcy_decode_nb_line_words(B);
    if (getenv("PCY_MC_TRACE")) {   // measurement aid (tools/bench_decode_nb.py): in-kernel time stamps, [layer][workgroup][16]
      if (!g_mc_trace) { hipMalloc(&g_mc_trace, 2 * 128 * 256 * 16 * 8); hipMemset(g_mc_trace, 0, 2 * 128 * 256 * 16 * 8); }
      bp.trace = g_mc_trace + (size_t)128 * 256 * 16;
    }
    step_done = pcy_launch_decode_step_nb(s, c->device, t, bp, mc, sa, c->n_cu, c->ao_sync, B, decode_xmin_nb(B));
  }
  if (try_layer && decode_step_enabled() && c->dev_layers) {   // all layers in one launch
    PcyDecAttnArgs t{};
    t.qkv = qkv; t.ld = qkvw; t.kcache = (bf16_t*)kv->k; t.vcache = (bf16_t*)kv->v;
    t.o = ao; t.ldo = H * dh; t.pos_dev = st->pos; t.cos_t = (const bf16_t*)m->rope_cos; t.sin_t = (const bf16_t*)m->rope_sin;
    t.keep = st->keep; t.ld_keep = kv->Tmax; t.scratch = scores; t.B = B; t.H = H; t.Hkv = Hkv; t.dh = dh; t.Tmax = kv->Tmax;
    t.scale = 1.0f / sqrtf((float)dh);
    t.xflags = c->ao_sync + 64 + AO_MAX_LAYERS * AO_FLAGS;
    PcyAttnBlockArgs bp{};
    bp.x = x; bp.d = d; bp.Nq = qkvw; bp.rms_eps = m->rms_eps; bp.rms_cast = m->rms_cast; bp.epoch = c->ao_sync + 1; bp.err = c->xwg_err;
    PcyMlpChainArgs mc{};
    mc.x = x; mc.x_out = x; mc.d = d; mc.F = F; mc.rms_eps = m->rms_eps; mc.rms_cast = m->rms_cast; mc.epoch = c->ao_sync + 1; mc.err = c->xwg_err;
    PcyDecodeStepArgs sa{};
    sa.layers = c->dev_layers; sa.n_layers = m->n_layers; sa.kv_layer_stride = layer_stride;
    sa.tags = c->mc_tags; sa.tag_stride = tag_stride; sa.xflags_stride = AO_FLAGS;
    sa.x_lines = c->mc_tags + (size_t)m->n_layers * tag_stride; sa.x_lines_stride = 32 * 256;
    if (getenv("PCY_MC_TRACE")) {   // measurement aid (tools/bench_decode.py): in-kernel time stamps, [layer][workgroup][16]
      if (!g_mc_trace) { hipMalloc(&g_mc_trace, 2 * 128 * 256 * 16 * 8); hipMemset(g_mc_trace, 0, 2 * 128 * 256 * 16 * 8); }
      bp.trace = g_mc_trace + (size_t)128 * 256 * 16;
    }
    step_done = pcy_launch_decode_step(s, t, bp, mc, sa, c->n_cu, c->ao_sync);
  }
  int xn_ready = 0;   // batched path: xn = RMSNorm(x) of the NEXT projection already produced by a fused finish kernel
  if (mb_step) {
    PcyMbArgs ma{};
    ma.layers = c->dev_layers; ma.n_layers = m->n_layers; ma.final_norm = (const bf16_t*)m->final_norm;
    ma.x = x; ma.xn = xn; ma.ao = ao; ma.act = act;
    ma.qkv_ws = sk_ws; ma.sk_ws = sk_ws + (size_t)2 * B * qkvw;
    ma.kcache = (bf16_t*)kv->k; ma.vcache = (bf16_t*)kv->v; ma.kv_layer_stride = layer_stride; ma.Bcache = kv->B;
    ma.pos_dev = st->pos; ma.cos_t = (const bf16_t*)m->rope_cos; ma.sin_t = (const bf16_t*)m->rope_sin; ma.keep = st->keep; ma.ld_keep = kv->Tmax;
    ma.B = B; ma.Tmax = kv->Tmax; ma.scale = 1.0f / sqrtf((float)dh); ma.rms_eps = m->rms_eps; ma.rms_cast = m->rms_cast;
    ma.flags = c->mb_flags; ma.epoch = c->mb_sync; ma.err = c->xwg_err;
    { const char* e = getenv("PCY_MB_ABL"); ma.abl = e ? atoi(e) : 0; }
    if (getenv("PCY_MC_TRACE")) {   // measurement aid (tools/bench_decode_mb.py): in-kernel time stamps, [layer][workgroup][16]
      if (!g_mc_trace) { hipMalloc(&g_mc_trace, 2 * 128 * 256 * 16 * 8); hipMemset(g_mc_trace, 0, 2 * 128 * 256 * 16 * 8); }
      ma.trace = g_mc_trace + (size_t)128 * 256 * 16;
    }
    pcy_launch_rmsnorm(s, x, (const bf16_t*)m->layers[0].ln1, xn, B, d, m->rms_eps, m->rms_cast);
    if (pcy_launch_decode_step_mb(s, c->device, ma, c->n_cu)) { step_done = true; xn_ready = 1; }
  }
  for (int l = 0; l < (step_done ? 0 : m->n_layers); ++l) {
    const pcy_llama_layer& L = m->layers[l];
    PcyGemvArgs g{};
    g.W = (const bf16_t*)L.wqkv; g.x = x; g.y = qkv; g.rms_w = (const bf16_t*)L.ln1; g.rms_eps = m->rms_eps;
    g.rms_cast = m->rms_cast; g.N = qkvw; g.K = d; g.B = B; g.ldx = d; g.ldy = qkvw; g.epi = EPI_STORE;
    g.splitk_ws = sk_ws; g.splitk_ws_bytes = sk_bytes; g.force_stream = nb_on;
    if (batched) {   // (the previous layer's down projection may have written xn already, fused into its K-split finish)
      if (!xn_ready) pcy_launch_rmsnorm(s, x, (const bf16_t*)L.ln1, xn, B, d, m->rms_eps, m->rms_cast);
      xn_ready = 0;
      g.x = xn; g.rms_w = nullptr;
    }
    PcyDecAttnArgs t{};
    t.qkv = qkv; t.ld = qkvw; t.kcache = (bf16_t*)kv->k + l * layer_stride; t.vcache = (bf16_t*)kv->v + l * layer_stride;
    t.o = ao; t.ldo = H * dh; t.pos_dev = st->pos; t.cos_t = (const bf16_t*)m->rope_cos; t.sin_t = (const bf16_t*)m->rope_sin;
    t.keep = st->keep; t.ld_keep = kv->Tmax; t.scratch = scores; t.B = B; t.H = H; t.Hkv = Hkv; t.dh = dh; t.Tmax = kv->Tmax;
    t.scale = 1.0f / sqrtf((float)dh);
    t.force_ds = nb_on ? pcy_decode_nb_ds(B) : 0;
    // (multi-head geometry, one row: the column split of the fused step's attention workgroups, whatever the launch mix -- one set of bits)
    const bool mha1 = B == 1 && pcy_decode_mha_covers(d, H, Hkv, dh, F, c->n_cu);
    if (mha1) t.force_ds = pcy_decode_mha_ds();
    // one row: the projections over d walk their k-iterations rotated (PcyGemvArgs::krot) -- in the streaming launches, in the launches that
    // fuse them (attn_o, mlp_chain) and in the one-launch steps alike: one set of bits
    const int krot1 = B == 1 ? 1 : 0;
    g.krot = krot1;   // (+ the rotated k order of its projections, PcyGemvArgs::krot)
    PcyGemvArgs o{};
    o.W = (const bf16_t*)L.wo; o.x = ao; o.y = x; o.resid = x; o.N = d; o.K = H * dh; o.B = B; o.ldx = H * dh; o.ldy = d; o.epi = EPI_RESID;
    o.splitk_ws = sk_ws; o.splitk_ws_bytes = sk_bytes; o.force_stream = nb_on; o.krot = krot1;
    // attention and o projection in one launch (Wo rows wait in registers while the attention runs) where covered
    if (batched && B <= 32) { o.next_rms_w = (const bf16_t*)L.ln2; o.next_xn = xn; o.fused_next = &xn_ready; o.rms_eps = m->rms_eps; o.rms_cast = m->rms_cast; }
    if (try_layer) {   // the whole layer as one launch
      uint32_t* tags = c->mc_tags + (size_t)l * tag_stride;
      PcyAttnBlockArgs bp{};
      bp.x = x; bp.ln1 = (const bf16_t*)L.ln1; bp.wqkv = (const bf16_t*)L.wqkv; bp.wo = (const bf16_t*)L.wo;
      bp.d = d; bp.Nq = qkvw; bp.rms_eps = m->rms_eps; bp.rms_cast = m->rms_cast;
      bp.qkv_tag = tags + F; bp.ao_tag = bp.qkv_tag + qkvw; bp.xo_tag = bp.ao_tag + H * dh;
      bp.epoch = c->ao_sync + 1; bp.err = c->xwg_err;
      if (getenv("PCY_MC_TRACE")) {   // measurement aid (tools/bench_decode.py): in-kernel time stamps, [layer][workgroup][16]
        if (!g_mc_trace) { hipMalloc(&g_mc_trace, 2 * 128 * 256 * 16 * 8); hipMemset(g_mc_trace, 0, 2 * 128 * 256 * 16 * 8); }
        bp.trace = g_mc_trace + (size_t)(128 + l) * 256 * 16;
      }
      PcyMlpChainArgs mc{};
      mc.x = x; mc.x_out = x; mc.ln2 = (const bf16_t*)L.ln2; mc.wgu = (const bf16_t*)L.wgu; mc.wdown = (const bf16_t*)L.wdown;
      mc.d = d; mc.F = F; mc.rms_eps = m->rms_eps; mc.rms_cast = m->rms_cast;
      mc.act_tag = tags; mc.epoch = c->ao_sync + 1; mc.err = c->xwg_err;
      if (pcy_launch_decode_layer(s, t, bp, mc, c->n_cu, c->ao_sync, c->ao_sync + 64 + (AO_MAX_LAYERS + l) * AO_FLAGS)) continue;
      try_layer = false;   // geometry not covered: the same for every layer
    }
    int qkv_splits = 0;   // batched: the attention adds up the K-split partial sums of ITS rows (no finish launch); PCY_DISABLE=attn_qkv_finish: separate launch
    if (batched && sk_ws && !try_ao && !qkv_finish_launch()) g.defer_finish = &qkv_splits;
    pcy_launch_gemv(s, g);
    if (qkv_splits > 1) { t.qkv_partials = sk_ws; t.qkv_splits = qkv_splits; }
    if (!(try_ao && pcy_launch_attn_o(s, t, o, c->n_cu, c->ao_sync, c->ao_sync + 64 + l * AO_FLAGS, AO_FLAGS, c->xwg_err,
                                       c->ao_sync + 64 + (AO_MAX_LAYERS + l) * AO_FLAGS))) {
      pcy_launch_attn_decode(s, t);
      pcy_launch_gemv(s, o);
    }
    PcyGemvArgs u{};
    u.W = (const bf16_t*)L.wgu; u.x = x; u.y = act; u.rms_w = (const bf16_t*)L.ln2; u.rms_eps = m->rms_eps; u.rms_cast = m->rms_cast;
    u.N = F; u.K = d; u.B = B; u.ldx = d; u.ldy = F; u.epi = EPI_SWIGLU; u.force_stream = nb_on; u.krot = krot1;
    if (batched) {
      if (!xn_ready) pcy_launch_rmsnorm(s, x, (const bf16_t*)L.ln2, xn, B, d, m->rms_eps, m->rms_cast);
      xn_ready = 0;
      u.x = xn; u.rms_w = nullptr;
    }
    pcy_launch_gemv(s, u);
    PcyGemvArgs w{};
    w.W = (const bf16_t*)L.wdown; w.x = act; w.y = x; w.resid = x; w.N = d; w.K = F; w.B = B; w.ldx = F; w.ldy = d; w.epi = EPI_RESID;
    w.splitk_ws = sk_ws; w.splitk_ws_bytes = sk_bytes; w.force_stream = nb_on;
    if (batched && B <= 32) {
      w.next_rms_w = (const bf16_t*)(l + 1 < m->n_layers ? m->layers[l + 1].ln1 : m->final_norm);
      w.next_xn = xn; w.fused_next = &xn_ready; w.rms_eps = m->rms_eps; w.rms_cast = m->rms_cast;
    }
    if (nb_on && pcy_launch_gemv_kwin4(s, w)) continue;   // (the four-way K split of the small-batch step's down projection)
    pcy_launch_gemv(s, w);
  }
  if (layers_only) return;
  PcyGemvArgs h{};
  h.W = (const bf16_t*)m->lm_head; h.x = x; h.y = (bf16_t*)st->logits; h.rms_w = (const bf16_t*)m->final_norm; h.rms_eps = m->rms_eps;
  h.rms_cast = m->rms_cast; h.N = m->vocab; h.K = d; h.B = B; h.ldx = d; h.ldy = m->vocab; h.epi = EPI_STORE;
  // (4 rows on the small-batch step: the streaming kernel with the final norm fused -- one pass over the matrix for up to 4 rows -- instead of
  // norm + MFMA GEMV: 3.178 -> 3.153 ms per step; 5 and 8 rows would take two passes: 3.51 -> 3.65, 4.00 -> 4.15)
  if (batched_head && !(nb_on && B <= 4)) {
    if (!xn_ready) pcy_launch_rmsnorm(s, x, (const bf16_t*)m->final_norm, xn, B, d, m->rms_eps, m->rms_cast);
    h.x = xn; h.rms_w = nullptr;
  } else if (nb_on) {
    h.force_stream = 1;
  }
  pcy_launch_gemv(s, h);
}

// Record of the per-step logits (the reference returns it on the CPU, model_unified.py:892,917-921).  `all` may be PINNED
// HOST memory: the rows then cross PCIe while the next decode step runs, instead of as one 65 MB (batch 1, 256 tokens) to
// 4.2 GB (batch 32, 512 tokens) device-to-host copy after the loop -- batch-32 generation 8.3 -> see DESIGN.md ms per token end to end.
// Row (step, b) starts at all + (step*B + b)*ld; with ld % 8 == 0 every row is 16-byte aligned and is written with 16-byte
// stores (the source rows have the odd stride V, so they are gathered with 2-byte loads from L2).
__global__ void store_logits_kernel(const bf16_t* __restrict__ logits, bf16_t* __restrict__ all, const int32_t* step_dev,
                                    int B, int V, int ld) {
  const size_t step = (size_t)(*step_dev);
  const int chunks = (V + 7) / 8;
  const bool vec = (ld % 8 == 0) && ((reinterpret_cast<uintptr_t>(all) & 15) == 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)B * chunks; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / chunks), c = (int)(i % chunks);
    const bf16_t* src = logits + (size_t)b * V + c * 8;
    bf16_t* dst = all + (step * B + b) * ld + c * 8;
    if (vec && c * 8 + 8 <= V) {
      const uint32_t w0 = src[0] | ((uint32_t)src[1] << 16), w1 = src[2] | ((uint32_t)src[3] << 16);
      const uint32_t w2 = src[4] | ((uint32_t)src[5] << 16), w3 = src[6] | ((uint32_t)src[7] << 16);
      *reinterpret_cast<uint4*>(dst) = make_uint4(w0, w1, w2, w3);
    } else {
      for (int e = 0; e < 8 && c * 8 + e < V; ++e) dst[e] = src[e];
    }
  }
}
void enqueue_pick(pcy_ctx* c, const pcy_llama_desc* m, const pcy_gen_state* st, int B, int advance_pos, int Tmax) {
  hipStream_t s = c->stream;
  if (st->logits_all)
    hipLaunchKernelGGL(store_logits_kernel, dim3(B >= 8 ? 256 : 64), dim3(256), 0, s, (const bf16_t*)st->logits,
                       (bf16_t*)st->logits_all, st->step, B, m->vocab, st->logits_all_ld > 0 ? st->logits_all_ld : m->vocab);
  // partials live at the tail of the decode workspace carve (same offsets as enqueue_decode)
  Carver cv(c->ws);
  const int qkvw = (m->n_heads + 2 * m->n_kv_heads) * m->head_dim;
  cv.take<bf16_t>((size_t)B * m->d); cv.take<bf16_t>((size_t)B * qkvw); cv.take<bf16_t>((size_t)B * m->n_heads * m->head_dim);
  cv.take<bf16_t>((size_t)B * m->ffn); cv.take<float>((size_t)B * m->n_heads * (Tmax + 1));
  void* partials = cv.take<char>((size_t)B * 64 * 16);
  pcy_launch_greedy_pick(s, (const bf16_t*)st->logits, B, m->vocab, st->next_tok, st->tokens_out, st->max_steps,
                         st->logprob, st->pos, st->step, advance_pos, partials);
}

// sampling / nucleus selection on state->logits (the non-greedy branch of `_generate_sampling`, model_unified.py:896-906)
void enqueue_sample(pcy_ctx* c, const pcy_llama_desc* m, const pcy_gen_state* st, int B, int advance_pos, int Tmax, float temperature,
                    float nucleus_p, const float* uniforms, bf16_t* probs_out) {
  hipStream_t s = c->stream;
  // the nucleus branch of the reference is softmax(logits) * mask -- the temperature is not applied there (model_unified.py:899-901)
  if (nucleus_p > 0.f) temperature = 1.0f;
  if (st->logits_all)
    hipLaunchKernelGGL(store_logits_kernel, dim3(B >= 8 ? 256 : 64), dim3(256), 0, s, (const bf16_t*)st->logits,
                       (bf16_t*)st->logits_all, st->step, B, m->vocab, st->logits_all_ld > 0 ? st->logits_all_ld : m->vocab);
  Carver cv(c->ws);
  const int qkvw = (m->n_heads + 2 * m->n_kv_heads) * m->head_dim;
  cv.take<bf16_t>((size_t)B * m->d); cv.take<bf16_t>((size_t)B * qkvw); cv.take<bf16_t>((size_t)B * m->n_heads * m->head_dim);
  cv.take<bf16_t>((size_t)B * m->ffn); cv.take<float>((size_t)B * m->n_heads * (Tmax + 1));
  void* partials = cv.take<char>((size_t)B * 64 * 16);
  pcy_launch_sample_step(s, (const bf16_t*)st->logits, B, m->vocab, temperature, nucleus_p, uniforms, c->smp_hist, probs_out, st->next_tok,
                         st->tokens_out, st->max_steps, st->logprob, st->pos, st->step, advance_pos, partials, c->smp_pbits);
}
int ensure_sample_state(pcy_ctx* c, int B, int V) {
  if ((size_t)B * V > c->smp_pbits_elems) {
    if (c->smp_pbits) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->smp_pbits)); c->smp_pbits = nullptr; c->smp_pbits_elems = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->smp_pbits), (size_t)B * V * 2));
    c->smp_pbits_elems = (size_t)B * V;
  }
  if (B <= c->smp_hist_rows) return 0;
  if (c->smp_hist) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->smp_hist)); c->smp_hist = nullptr; }
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->smp_hist), (size_t)B * 65536 * sizeof(unsigned)));
  HIP_TRY(hipMemset(c->smp_hist, 0, (size_t)B * 65536 * sizeof(unsigned)));
  c->smp_hist_rows = B;
  return 0;
}

// Beam-search cache reorder, cache[:, b] = cache[:, rows[b]] over slots [0, t), for EVERY layer and for K and V in two
// launches (gather into a scratch copy, copy back): grid (B, Hkv, 2L).  Rows that keep their place (rows[b] == b: most rows
// once the beams of a group have settled) are skipped in both passes.  The per-layer version took 4 launches per layer
// (128 launches, ~1 ms of a 8.8 ms beam step at beam 10).
// The same reorder in ONE pass and in place for up to 32 rows: a thread owns one 16-byte piece of every row of a (layer, K | V, head) slab --
// it loads the piece of ALL rows (through LDS: the source row of a destination is only known at run time), then stores row b's piece from
// row rows[b]'s.  The rows that are somebody's source are read once and the moved rows written once (the two-launch form reads and writes
// the moved rows twice and needs a scratch copy); nothing is touched when no row moves.  Vanilla beam search (beam_group_size = beam_size, the reference's default)
// re-ranks most rows at every step: 3.87 -> see DESIGN.md ms per beam-5 step at a 570-token cache.
template <int MAXB>
__global__ __launch_bounds__(128) void kv_permute_kernel(bf16_t* __restrict__ kbase, bf16_t* __restrict__ vbase, const int32_t* __restrict__ rows, int B,
                                                         int Bcache, int Hkv, int Tmax, int t, int dh, const int32_t* __restrict__ t_dev, int t0) {
  __shared__ uint4 stage[MAXB * 128];
  const int h = blockIdx.x, lw = blockIdx.y, l = lw >> 1;
  unsigned need = 0;                                       // rows some OTHER row takes its contents from (uniform)
  bool moves = false;
  for (int b = 0; b < B; ++b) {
    const int sb = rows[b];
    if (sb != b) { moves = true; if (sb >= 0 && sb < B) need |= 1u << sb; }
  }
  if (!moves) return;                                      // no row moves
  if (t_dev) { t = *t_dev; t = t < Tmax ? t : Tmax; }
  bf16_t* cache = ((lw & 1) ? vbase : kbase) + (size_t)l * Bcache * Hkv * Tmax * dh + (size_t)h * Tmax * dh;
  const size_t rstride = (size_t)Hkv * Tmax * dh;          // elements between two rows of the slab
  const size_t n8 = (size_t)t * dh / 8;
  // slots [t0, t) only: the caller knows that the rows hold the same contents below t0 (beam search: the beams of a prompt share its prefix)
  for (size_t i = (size_t)t0 * dh / 8 + (size_t)blockIdx.z * 128 + threadIdx.x; i < n8; i += (size_t)gridDim.z * 128) {
#pragma unroll 4
    for (int b = 0; b < B; ++b)
      if ((need >> b) & 1u) stage[b * 128 + threadIdx.x] = *reinterpret_cast<const uint4*>(cache + (size_t)b * rstride + i * 8);
#pragma unroll 4
    for (int b = 0; b < B; ++b) {
      const int sb = rows[b];
      if (sb == b) continue;
      // (a source beyond the permuted rows -- pcy_kv_reorder accepts any cache row -- is not staged, and is nobody's destination: straight from memory)
      const uint4 v = (sb >= 0 && sb < B) ? stage[sb * 128 + threadIdx.x] : *reinterpret_cast<const uint4*>(cache + (size_t)sb * rstride + i * 8);
      *reinterpret_cast<uint4*>(cache + (size_t)b * rstride + i * 8) = v;
    }
  }
}
// enqueue: the one-pass form for <= 32 rows (PCY_DISABLE=kv_permute: the two launches through the scratch copy; same result)
static bool enqueue_kv_permute(hipStream_t s, const pcy_llama_desc* m, const pcy_kv_cache* kv, const int32_t* src_rows, int B, int t, const int32_t* t_dev,
                               int t0 = 0) {
  if (B > 32 || pcy_off("kv_permute")) return false;
  const dim3 grid(m->n_kv_heads, 2 * m->n_layers, 4);
  if (B <= 8)
    hipLaunchKernelGGL(kv_permute_kernel<8>, grid, dim3(128), 0, s, (bf16_t*)kv->k, (bf16_t*)kv->v, src_rows, B, kv->B, m->n_kv_heads, kv->Tmax, t, m->head_dim, t_dev, t0);
  else
    hipLaunchKernelGGL(kv_permute_kernel<32>, grid, dim3(128), 0, s, (bf16_t*)kv->k, (bf16_t*)kv->v, src_rows, B, kv->B, m->n_kv_heads, kv->Tmax, t, m->head_dim, t_dev, t0);
  return true;
}

// t_dev != nullptr (the replayed beam step): the number of slots is read from the device (the position counter the beam step has just
// advanced) and the scratch rows are Tmax slots apart -- nothing in the launch depends on the step.
__global__ void kv_gather_kernel(bf16_t* __restrict__ kbase, bf16_t* __restrict__ vbase, bf16_t* __restrict__ tmp,
                                 const int32_t* __restrict__ rows, int B, int Bcache, int Hkv, int Tmax, int t, int dh, int to_tmp,
                                 const int32_t* __restrict__ t_dev, int t0) {
  const int b = blockIdx.x, h = blockIdx.y, lw = blockIdx.z, l = lw >> 1;
  const int src_b = rows[b];
  if (src_b == b) return;
  int ts = t;                 // slots between two scratch rows
  if (t_dev) { t = *t_dev; t = t < Tmax ? t : Tmax; ts = Tmax; }
  bf16_t* cache = ((lw & 1) ? vbase : kbase) + (size_t)l * Bcache * Hkv * Tmax * dh;
  bf16_t* scratch = tmp + (size_t)lw * B * Hkv * ts * dh;
  const size_t n8 = (size_t)t * dh / 8;
  const uint4* sp; uint4* dp;
  if (to_tmp) {
    sp = reinterpret_cast<const uint4*>(cache + ((size_t)src_b * Hkv + h) * Tmax * dh);
    dp = reinterpret_cast<uint4*>(scratch + ((size_t)b * Hkv + h) * ts * dh);
  } else {
    sp = reinterpret_cast<const uint4*>(scratch + ((size_t)b * Hkv + h) * ts * dh);
    dp = reinterpret_cast<uint4*>(cache + ((size_t)b * Hkv + h) * Tmax * dh);
  }
  for (size_t i = (size_t)t0 * dh / 8 + threadIdx.x; i < n8; i += blockDim.x) dp[i] = sp[i];
}

}  // namespace

// ====================================================================================== C ABI
extern "C" {

}  // extern "C"
hipStream_t pcy_ctx_stream(pcy_ctx* c) { return c->stream; }
extern "C" {
int pcy_abi_version(void) { return PCY_ABI_VERSION; }
unsigned long long pcy_debug_dispatch_count(int kind) { return (kind >= 0 && kind < PCY_DISPATCH_N) ? g_pcy_dispatch[kind] : 0; }
const char* pcy_last_error(void) { return g_err; }

int pcy_ctx_create(int device_id, void* stream, pcy_ctx** out) {
  if (!out) return fail(1, "pcy_ctx_create: out is NULL");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device_id < 0 || device_id >= n) return fail(1, "pcy_ctx_create: device %d of %d", device_id, n);
  HIP_TRY(hipSetDevice(device_id));
  pcy_ctx* c = new pcy_ctx();
  c->device = device_id;
  c->stream = reinterpret_cast<hipStream_t>(stream);
  HIP_TRY(hipEventCreate(&c->ev0));
  HIP_TRY(hipEventCreate(&c->ev1));
  HIP_TRY(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
  *out = c;
  return 0;
}
void pcy_ctx_destroy(pcy_ctx* c) {
  if (!c) return;
  hipStreamSynchronize(c->stream);
  c->drop_graph();
  c->drop_gslots();
  if (c->ws) hipFree(c->ws);
  if (c->xwg_err) hipHostFree(c->xwg_err);
  if (c->ao_sync) hipFree(c->ao_sync);
  if (c->mc_tags) hipFree(c->mc_tags);
  if (c->dev_layers) hipFree(c->dev_layers);
  for (auto& t : c->nb_tags) if (t) hipFree(t);
  if (c->nb_sync) hipFree(c->nb_sync);
  if (c->mb_flags) hipFree(c->mb_flags);
  if (c->mb_sync) hipFree(c->mb_sync);
  if (c->op_tags) hipFree(c->op_tags);
  if (c->beam_ws) hipFree(c->beam_ws);
  if (c->smp_hist) hipFree(c->smp_hist);
  if (c->smp_pbits) hipFree(c->smp_pbits);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->cap_stream) hipStreamDestroy(c->cap_stream);
  delete c;
}
int pcy_ctx_sync(pcy_ctx* c) {
  HIP_TRY(hipStreamSynchronize(c->stream));
  return take_sticky_error(c);
}
int pcy_timer_start(pcy_ctx* c) { HIP_TRY(hipEventRecord(c->ev0, c->stream)); return 0; }
int pcy_timer_stop(pcy_ctx* c, float* ms) {
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}

int pcy_gemm(pcy_ctx* c, const void* A, int lda, const void* W, const void* bias, const void* resid, int ldr, void* C,
             int ldc, int M, int N, int K, int epi) {
  if (K % 64 != 0) return fail(1, "pcy_gemm: K=%d must be a multiple of 64", K);
  if (epi == EPI_SWIGLU && N % 32 != 0) return fail(1, "pcy_gemm: SwiGLU needs N %% 32 == 0");
  if (epi == EPI_RESID && !resid) return fail(1, "pcy_gemm: EPI_RESID without residual");
  PcyGemmArgs a{};
  a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = (const bf16_t*)bias; a.resid = (const bf16_t*)resid;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.ldr = ldr; a.epi = epi;
  pcy_launch_gemm(c->stream, a);
  return check_launch("pcy_gemm");
}

int pcy_gemv(pcy_ctx* c, const void* W, const void* x, int ldx, const void* bias, const void* resid, void* y, int ldy,
             const void* rms_w, float rms_eps, int rms_cast, int N, int K, int B, int epi) {
  if (K % 8 != 0 || ldx % 8 != 0) return fail(1, "pcy_gemv: K and ldx must be multiples of 8");
  if (rms_w && !(epi == EPI_STORE || epi == EPI_SWIGLU)) return fail(1, "pcy_gemv: fused RMSNorm only with STORE/SWIGLU");
  if (rms_w && (size_t)K * 2 * (B < 4 ? B : 4) > 65536) return fail(1, "pcy_gemv: fused RMSNorm needs x to fit LDS");
  PcyGemvArgs g{};
  g.W = (const bf16_t*)W; g.x = (const bf16_t*)x; g.y = (bf16_t*)y; g.bias = (const bf16_t*)bias; g.resid = (const bf16_t*)resid;
  g.rms_w = (const bf16_t*)rms_w; g.rms_eps = rms_eps; g.rms_cast = rms_cast; g.N = N; g.K = K; g.B = B; g.ldx = ldx; g.ldy = ldy; g.epi = epi;
  pcy_launch_gemv(c->stream, g);
  return check_launch("pcy_gemv");
}

int pcy_decode_mlp(pcy_ctx* c, void* x, const void* ln2, const void* wgu, const void* wdown, int d, int ffn, float rms_eps, int rms_cast) {
  PCY_STICKY(c);
  if (d % 8 || ffn % 8 || (size_t)d * 2 > 65536) return fail(1, "pcy_decode_mlp: d=%d ffn=%d not covered", d, ffn);
  if (int r = ensure_sync_words(c)) return r;
  const size_t words = (size_t)ffn;
  if (c->op_tags_words != words) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->op_tags) HIP_TRY(hipFree(c->op_tags));
    c->op_tags = nullptr; c->op_tags_words = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->op_tags), words * 4));
    HIP_TRY(hipMemset(c->op_tags, 0, words * 4));
    HIP_TRY(hipMemset(c->ao_sync + 2, 0, 4));
    c->op_tags_words = words;
  }
  if (mlp_chain_enabled()) {
    PcyMlpChainArgs mc{};
    mc.x = (const bf16_t*)x; mc.x_out = (bf16_t*)x; mc.ln2 = (const bf16_t*)ln2; mc.wgu = (const bf16_t*)wgu; mc.wdown = (const bf16_t*)wdown;
    mc.d = d; mc.F = ffn; mc.rms_eps = rms_eps; mc.rms_cast = rms_cast;
    mc.act_tag = c->op_tags; mc.epoch = c->ao_sync + 2; mc.err = c->xwg_err;
    pcy_launch_bump(c->stream, c->ao_sync + 2);   // a fresh tag for every use of the slot
    if (pcy_launch_mlp_chain(c->stream, mc, c->n_cu)) return check_launch("pcy_decode_mlp");
  }
  if (int r = c->reserve((size_t)ffn * 2 + 256)) return r;
  PcyGemvArgs u{};
  u.W = (const bf16_t*)wgu; u.x = (const bf16_t*)x; u.y = (bf16_t*)c->ws; u.rms_w = (const bf16_t*)ln2; u.rms_eps = rms_eps; u.rms_cast = rms_cast;
  u.N = ffn; u.K = d; u.B = 1; u.ldx = d; u.ldy = ffn; u.epi = EPI_SWIGLU; u.krot = 1;   // (as mlp_chain_kernel walks it)
  pcy_launch_gemv(c->stream, u);
  PcyGemvArgs w{};
  w.W = (const bf16_t*)wdown; w.x = (const bf16_t*)c->ws; w.y = (bf16_t*)x; w.resid = (const bf16_t*)x; w.N = d; w.K = ffn; w.B = 1; w.ldx = ffn; w.ldy = d;
  w.epi = EPI_RESID;
  pcy_launch_gemv(c->stream, w);
  return check_launch("pcy_decode_mlp");
}

int pcy_rmsnorm(pcy_ctx* c, const void* x, const void* w, void* y, int rows, int d, float eps, int cast) {
  if (d % 8) return fail(1, "pcy_rmsnorm: d %% 8 != 0");
  pcy_launch_rmsnorm(c->stream, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rows, d, eps, cast);
  return check_launch("pcy_rmsnorm");
}
int pcy_layernorm(pcy_ctx* c, const void* x, const void* w, const void* b, void* y, int rows, int d, float eps) {
  if (d % 8) return fail(1, "pcy_layernorm: d %% 8 != 0");
  pcy_launch_layernorm(c->stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)b, (bf16_t*)y, rows, d, eps);
  return check_launch("pcy_layernorm");
}
int pcy_embed_splice(pcy_ctx* c, const void* table, const int32_t* ids, const void* soft, const int32_t* soft_map, void* out,
                     int rows, int d) {
  if (d % 8) return fail(1, "pcy_embed_splice: d %% 8 != 0");
  pcy_launch_embed_gather(c->stream, (const bf16_t*)table, ids, (const bf16_t*)soft, soft_map, (bf16_t*)out, rows, d);
  return check_launch("pcy_embed_splice");
}
int pcy_pool(pcy_ctx* c, const void* hidden, int d, const int32_t* seg, const int32_t* rng, int nprot, int mode, void* out) {
  if (mode < 0 || mode > 2) return fail(1, "pcy_pool: mode %d", mode);
  if (d % 8) return fail(1, "pcy_pool: d=%d must be a multiple of 8", d);
  if (int r = c->reserve(pcy_pool_ws_bytes(nprot, d) + 256)) return r;
  pcy_launch_pool(c->stream, (const bf16_t*)hidden, d, seg, rng, nprot, mode, (bf16_t*)out, c->ws);
  return check_launch("pcy_pool");
}

int pcy_rope(pcy_ctx* c, void* buf, int ld, int col0, int nh, int dh, const int32_t* pos, const void* cos_t, const void* sin_t,
             int ntok, int mode, float prescale) {
  if (dh % 2) return fail(1, "pcy_rope: odd head_dim");
  pcy_launch_rope(c->stream, (bf16_t*)buf, ld, col0, nh, dh, pos, (const bf16_t*)cos_t, (const bf16_t*)sin_t, ntok, mode, prescale);
  return check_launch("pcy_rope");
}

int pcy_attention(pcy_ctx* c, const void* q, int ldq, int qcol0, const void* k, int ldk, int kcol0, const void* v, int ldv,
                  int vcol0, void* o, int ldo, const int32_t* cu, const int32_t* vt_cu, const uint8_t* keep, int nseq,
                  int max_len, int vt_total, int H, int Hkv, int dh, int causal, float scale) {
  if (dh != 32 && dh != 64 && dh != 128) return fail(1, "pcy_attention: head_dim %d unsupported (32/64/128)", dh);
  if ((Hkv * dh) % 64 || H % Hkv) return fail(1, "pcy_attention: Hkv*dh must be a multiple of 64 and Hkv | H");
  // the single-pass kernel wants every sequence's Vt slice zero-padded to a multiple of 64 keys: it lays Vt out itself (offsets
  // computed on the device from cu), whatever vt_cu / vt_total the caller passed
  const bool fast = pcy_attn_fast_eligible(dh, causal, keep != nullptr, scale, H, Hkv, ldq, qcol0, ldk, kcol0, ldo);
  int32_t* vt_cu64 = nullptr;
  if (fast) {
    int ntok_ub = 0;   // upper bound of the token count: nseq * max_len
    ntok_ub = nseq * max_len;
    vt_total = (int)align_up((size_t)ntok_ub + 64 * (size_t)nseq, 8);
  }
  if (int r = c->reserve(align_up((size_t)Hkv * dh * vt_total * 2, 256) + align_up((size_t)(nseq + 1) * 4, 256) + 4096)) return r;
  bf16_t* vt = reinterpret_cast<bf16_t*>(c->ws);
  // the single-pass kernel reads V token-major where it is (PCY_DISABLE=fa_vrow: from a transposed copy, as the other kernels do)
  const bool vrow = fast && pcy_attn_fast_vrow(ldv, vcol0);
  if (fast && !vrow) {
    vt_cu64 = reinterpret_cast<int32_t*>(c->ws + align_up((size_t)Hkv * dh * vt_total * 2, 256));
    pcy_launch_vt_offsets(c->stream, cu, nseq, 64, vt_cu64);
    vt_cu = vt_cu64;
  }
  if (!vrow) pcy_launch_transpose_v(c->stream, (const bf16_t*)v, ldv, vcol0, Hkv, dh, cu, vt_cu, nseq, max_len, vt, vt_total);
  PcyAttnArgs t{};
  if (vrow) { t.v = (const bf16_t*)v; t.ldv = ldv; t.vcol0 = vcol0; }
  t.q = (const bf16_t*)q; t.ldq = ldq; t.qcol0 = qcol0; t.k = (const bf16_t*)k; t.ldk = ldk; t.kcol0 = kcol0; t.vt = vt;
  t.vt_total = vt_total; t.o = (bf16_t*)o; t.ldo = ldo; t.cu = cu; t.vt_cu = vt_cu; t.keep = keep; t.nseq = nseq; t.max_len = max_len;
  t.H = H; t.Hkv = Hkv; t.dh = dh; t.causal = causal; t.scale = scale; t.vt_pad64 = fast ? 1 : 0;
  pcy_launch_attn(c->stream, t);
  return check_launch("pcy_attention");
}

int pcy_attn_decode(pcy_ctx* c, void* qkv, int ld, void* kcache, void* vcache, void* o, int ldo, const int32_t* pos,
                    const void* cos_t, const void* sin_t, const uint8_t* keep, int B, int H, int Hkv, int dh, int Tmax) {
  if (dh != 32 && dh != 64 && dh != 128) return fail(1, "pcy_attn_decode: head_dim %d unsupported (32/64/128)", dh);
  const int G = H / Hkv;
  if (H % Hkv || !(G == 1 || G == 2 || G == 4 || G == 8)) return fail(1, "pcy_attn_decode: H/Hkv must be 1, 2, 4 or 8");
  PcyDecAttnArgs t{};
  t.qkv = (bf16_t*)qkv; t.ld = ld; t.kcache = (bf16_t*)kcache; t.vcache = (bf16_t*)vcache; t.o = (bf16_t*)o; t.ldo = ldo;
  t.pos_dev = pos; t.cos_t = (const bf16_t*)cos_t; t.sin_t = (const bf16_t*)sin_t; t.keep = keep; t.ld_keep = Tmax;
  t.scratch = nullptr; t.B = B; t.H = H; t.Hkv = Hkv; t.dh = dh; t.Tmax = Tmax; t.scale = 1.0f / sqrtf((float)dh);
  pcy_launch_attn_decode(c->stream, t);
  return check_launch("pcy_attn_decode");
}

int pcy_quant_rows_fp8(pcy_ctx* c, const void* x, int ldx, int rows, int K, void* q_out, float* scale_out) {
  if (rows < 0 || K <= 0 || K % 8 || ldx % 8) return fail(1, "pcy_quant_rows_fp8: K=%d ldx=%d must be multiples of 8", K, ldx);
  pcy_launch_quant_rows_fp8(c->stream, (const bf16_t*)x, ldx, rows, K, (unsigned char*)q_out, scale_out);
  return check_launch("pcy_quant_rows_fp8");
}
int pcy_gemm_fp8(pcy_ctx* c, const void* A8, const float* sa, const void* W8, const float* sw, const void* resid, int ldr,
                 void* C, int ldc, int M, int N, int K, int epi) {
  if (K <= 0 || K % 128) return fail(1, "pcy_gemm_fp8: K=%d must be a multiple of 128", K);
  if (epi != EPI_STORE && epi != EPI_RESID && epi != EPI_SWIGLU) return fail(1, "pcy_gemm_fp8: epilogue %d unsupported", epi);
  if (epi == EPI_SWIGLU && N % 32) return fail(1, "pcy_gemm_fp8: SwiGLU needs N %% 32 == 0");
  PcyGemmArgs a{};
  a.A = (const bf16_t*)A8; a.W = (const bf16_t*)W8; a.C = (bf16_t*)C; a.resid = (const bf16_t*)resid;
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldc = ldc; a.ldr = ldr; a.epi = epi; a.fp8 = 1; a.sa = sa; a.sw = sw;
  pcy_launch_gemm(c->stream, a);
  return check_launch("pcy_gemm_fp8");
}
// ---- RCCL all-gather of the sharded retrieval path (the reference's gather: training/trainIT.py:1594-1610)
// RCCL is resolved at first use with dlopen -- in a PyTorch process that is the copy torch has already loaded (torch/lib/librccl.so),
// so the engine and torch.distributed share one RCCL; libpcy.so itself carries no link-time dependency on it.
struct Id128 { char b[128]; };   // ncclUniqueId (passed by value)
namespace {
struct Rccl {
  void* h = nullptr;
  int (*get_unique_id)(void*) = nullptr;
  int (*comm_init_rank)(void**, int, Id128, int) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  const char* (*get_error_string)(int) = nullptr;
};
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (r.h) {
      r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.h, "ncclGetUniqueId"));
      r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.h, "ncclCommInitRank"));
      r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.h, "ncclAllGather"));
      r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.h, "ncclCommDestroy"));
      r.get_error_string = reinterpret_cast<decltype(r.get_error_string)>(dlsym(r.h, "ncclGetErrorString"));
    }
  }
  return (r.h && r.get_unique_id && r.comm_init_rank && r.all_gather && r.comm_destroy) ? &r : nullptr;
}
int rccl_fail(const char* what, int rc) {
  Rccl* r = rccl();
  return fail(5, "%s failed: %s", what, (r && r->get_error_string) ? r->get_error_string(rc) : "RCCL error");
}
}  // namespace
int pcy_comm_unique_id(void* id128) {
  Rccl* r = rccl();
  if (!r) return fail(5, "pcy_comm_unique_id: librccl.so not found");
  if (int rc = r->get_unique_id(id128)) return rccl_fail("ncclGetUniqueId", rc);
  return 0;
}
int pcy_comm_init(pcy_ctx* c, int nranks, int rank, const void* id128, void** comm_out) {
  Rccl* r = rccl();
  if (!r) return fail(5, "pcy_comm_init: librccl.so not found");
  HIP_TRY(hipSetDevice(c->device));
  Id128 id;
  memcpy(id.b, id128, sizeof(id.b));
  if (int rc = r->comm_init_rank(comm_out, nranks, id, rank)) return rccl_fail("ncclCommInitRank", rc);
  return 0;
}
void pcy_comm_destroy(void* comm) {
  Rccl* r = rccl();
  if (r && comm) r->comm_destroy(comm);
}
int pcy_allgather(pcy_ctx* c, void* comm, const void* sendbuf, void* recvbuf, size_t bytes) {
  Rccl* r = rccl();
  if (!r) return fail(5, "pcy_allgather: librccl.so not found");
  if (int rc = r->all_gather(sendbuf, recvbuf, bytes, /* ncclUint8 */ 1, comm, c->stream)) return rccl_fail("ncclAllGather", rc);
  return 0;
}

int pcy_retrieval_topk(pcy_ctx* c, const void* query, int Q, const void* targets, int N, int D, int k, int32_t* idx_out, void* score_out) {
  PCY_STICKY(c);
  if (D % 64) return fail(1, "pcy_retrieval_topk: D=%d must be a multiple of 64", D);
  if (k < 1 || k > N) return fail(1, "pcy_retrieval_topk: k=%d must be in [1, N=%d]", k, N);
  const size_t qb = align_up((size_t)Q * D * 2, 256), tb = align_up((size_t)N * D * 2, 256), sb = align_up((size_t)Q * N * 2, 256);
  if (int r = c->reserve(qb + tb + sb + 4096)) return r;
  bf16_t* qn = reinterpret_cast<bf16_t*>(c->ws);
  bf16_t* tn = reinterpret_cast<bf16_t*>(c->ws + qb);
  bf16_t* sims = reinterpret_cast<bf16_t*>(c->ws + qb + tb);
  pcy_launch_l2norm_rows(c->stream, (const bf16_t*)query, qn, Q, D, 1e-12f);
  pcy_launch_l2norm_rows(c->stream, (const bf16_t*)targets, tn, N, D, 1e-12f);
  linear(c->stream, qn, D, tn, nullptr, nullptr, 0, sims, N, Q, N, D, EPI_STORE);
  pcy_launch_retrieval_rank(c->stream, sims, Q, N, k, idx_out, (bf16_t*)score_out);
  return check_launch("pcy_retrieval_topk");
}
int pcy_retrieval_scores(pcy_ctx* c, const void* query, int Q, const void* targets, int N, int D, void* sims_out) {
  PCY_STICKY(c);
  if (D % 64) return fail(1, "pcy_retrieval_scores: D=%d must be a multiple of 64", D);
  const size_t qb = align_up((size_t)Q * D * 2, 256), tb = align_up((size_t)N * D * 2, 256);
  if (int r = c->reserve(qb + tb + 4096)) return r;
  bf16_t* qn = reinterpret_cast<bf16_t*>(c->ws);
  bf16_t* tn = reinterpret_cast<bf16_t*>(c->ws + qb);
  pcy_launch_l2norm_rows(c->stream, (const bf16_t*)query, qn, Q, D, 1e-12f);
  pcy_launch_l2norm_rows(c->stream, (const bf16_t*)targets, tn, N, D, 1e-12f);
  linear(c->stream, qn, D, tn, nullptr, nullptr, 0, (bf16_t*)sims_out, N, Q, N, D, EPI_STORE);
  return check_launch("pcy_retrieval_scores");
}

int pcy_retrieval_scores_f32(pcy_ctx* c, const float* query, int Q, const void* targets, int targets_bf16, int N, int D, float* sims_out) {
  PCY_STICKY(c);
  if (D % 4 || D <= 0) return fail(1, "pcy_retrieval_scores_f32: D=%d must be a positive multiple of 4", D);
  if (pcy_retrieval_dot_smem(D) > 159 * 1024) return fail(1, "pcy_retrieval_scores_f32: D=%d does not fit the query tile in LDS", D);
  pcy_launch_retrieval_dot_f32(c->stream, query, Q, targets, targets_bf16, N, D, 1e-12f, sims_out);
  return check_launch("pcy_retrieval_scores_f32");
}
int pcy_retrieval_topk_f32(pcy_ctx* c, const float* query, int Q, const void* targets, int targets_bf16, int N, int D, int k, int32_t* idx_out,
                           float* score_out) {
  PCY_STICKY(c);
  if (D % 4 || D <= 0) return fail(1, "pcy_retrieval_topk_f32: D=%d must be a positive multiple of 4", D);
  if (pcy_retrieval_dot_smem(D) > 159 * 1024) return fail(1, "pcy_retrieval_topk_f32: D=%d does not fit the query tile in LDS", D);
  if (k < 1 || k > N) return fail(1, "pcy_retrieval_topk_f32: k=%d must be in [1, N=%d]", k, N);
  if (int r = c->reserve(align_up((size_t)Q * N * 4, 256) + 4096)) return r;
  float* sims = reinterpret_cast<float*>(c->ws);
  pcy_launch_retrieval_dot_f32(c->stream, query, Q, targets, targets_bf16, N, D, 1e-12f, sims);
  pcy_launch_retrieval_rank_f32(c->stream, sims, Q, N, k, idx_out, score_out);
  return check_launch("pcy_retrieval_topk_f32");
}

int pcy_qa_probs(pcy_ctx* c, const void* logits, int is_f32, int rows, int V, int yes_id, int no_id, void* probs_out, float* yes_no_out,
                 int32_t* argmax_out) {
  PCY_STICKY(c);
  if (!logits || rows < 0 || V <= 0) return fail(1, "pcy_qa_probs: logits NULL or bad shape (rows %d, V %d)", rows, V);
  if (yes_no_out && (yes_id < 0 || yes_id >= V || no_id < 0 || no_id >= V)) return fail(1, "pcy_qa_probs: yes / no token outside the vocabulary");
  pcy_launch_qa_probs(c->stream, logits, is_f32, rows, V, yes_id, no_id, probs_out, yes_no_out, argmax_out);
  return check_launch("pcy_qa_probs");
}

int pcy_mlp_forward(pcy_ctx* c, const pcy_mlp_desc* m, const void* x, int M, void* out) {
  PCY_STICKY(c);
  if (m->n_layers < 1 || m->n_layers > 8) return fail(1, "pcy_mlp_forward: n_layers %d", m->n_layers);
  int maxw = 0;
  for (int i = 0; i <= m->n_layers; ++i) {
    if (m->dims[i] % 8) return fail(1, "pcy_mlp_forward: dims must be multiples of 8");
    if (i > 0 && i < m->n_layers && m->dims[i] > maxw) maxw = m->dims[i];
  }
  if (int r = c->reserve(2 * align_up((size_t)M * (maxw + 8) * 2, 256) + 4096)) return r;
  Carver cv(c->ws);
  bf16_t* t[2] = {cv.take<bf16_t>((size_t)M * (maxw + 8)), cv.take<bf16_t>((size_t)M * (maxw + 8))};
  const bf16_t* cur = (const bf16_t*)x;
  for (int i = 0; i < m->n_layers; ++i) {
    const bool last = i == m->n_layers - 1;
    bf16_t* dst = last ? (bf16_t*)out : t[i & 1];
    const int K = m->dims[i], N = m->dims[i + 1];
    if (M > 8 && K % 64 != 0) return fail(1, "pcy_mlp_forward: K=%d must be a multiple of 64 for M>8", K);
    linear(c->stream, cur, K, (const bf16_t*)m->w[i], (const bf16_t*)m->b[i], nullptr, 0, dst, N, M, N, K,
           last ? EPI_STORE : EPI_GELU_ERF);
    cur = dst;
  }
  return check_launch("pcy_mlp_forward");
}

static int esm_encode_enqueue(pcy_ctx* c, const pcy_esm_desc* m, const int32_t* tokens, const int32_t* pos, const int32_t* cu,
                              const int32_t* vt_cu, int ntok, int nseq, int max_len, int vt_total, int mask_pads, void* hidden_out);
// FNV-1a over EVERY model pointer and constant a captured pcy_esm_encode chain bakes in (the CONTENTS of the host layer array, not its address:
// a freed engine's descriptor array and device blocks can come back at the same addresses with other weights behind them)
static uint64_t esm_weights_fingerprint(const pcy_esm_desc* m) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
  mix((uint64_t)(uintptr_t)m->layers); mix((uint64_t)(uintptr_t)m->final_ln_w); mix((uint64_t)(uintptr_t)m->final_ln_b);
  mix((uint64_t)(uintptr_t)m->rope_cos); mix((uint64_t)(uintptr_t)m->rope_sin);
  uint32_t eps; memcpy(&eps, &m->ln_eps, 4); mix(eps); mix((uint64_t)(uint32_t)m->vocab);
  for (int l = 0; l < m->n_layers; ++l) {
    const unsigned char* p = reinterpret_cast<const unsigned char*>(&m->layers[l]);
    for (size_t i = 0; i < sizeof(pcy_esm_layer); ++i) mix(p[i]);
  }
  return h;
}
// tokens at or below which pcy_esm_encode replays a captured launch chain (PCY_DISABLE=esm_graph: always launch by launch; read per call)
static int esm_graph_max_tokens() {
  return pcy_off("esm_graph") ? 0 : 40000;   // (round 5: bulk batches too -- see EsmEngine.GRAPH_MAX_TOKENS)
}
int pcy_esm_encode(pcy_ctx* c, const pcy_esm_desc* m, const int32_t* tokens, const int32_t* pos, const int32_t* cu,
                   const int32_t* vt_cu, int ntok, int nseq, int max_len, int vt_total, int mask_pads, void* hidden_out) {
  PCY_STICKY(c);
  if (ntok <= 0 || ntok > esm_graph_max_tokens() || !c->cap_stream)
    return esm_encode_enqueue(c, m, tokens, pos, cu, vt_cu, ntok, nseq, max_len, vt_total, mask_pads, hidden_out);
  // everything a launch of the chain bakes in: argument pointers, sizes, the model's arrays, the workspace base, and the run-time
  // switches that select kernels (read per launch from the environment: a changed switch is another key)
  auto envh = [](const char* n) { const char* e = getenv(n); uint64_t h = 1469598103934665603ull; for (; e && *e; ++e) h = (h ^ (unsigned char)*e) * 1099511628211ull; return h; };
  uint64_t key[16] = {(uint64_t)(uintptr_t)tokens, (uint64_t)(uintptr_t)pos, (uint64_t)(uintptr_t)cu, (uint64_t)(uintptr_t)vt_cu,
                      (uint64_t)(uintptr_t)hidden_out, esm_weights_fingerprint(m), (uint64_t)(uintptr_t)m->embed,
                      ((uint64_t)(uint32_t)ntok << 32) | (uint32_t)nseq, ((uint64_t)(uint32_t)max_len << 32) | (uint32_t)vt_total,
                      ((uint64_t)(uint32_t)m->d << 32) | (uint32_t)m->ffn, ((uint64_t)(uint32_t)m->n_layers << 32) | (uint32_t)m->n_heads,
                      (uint64_t)mask_pads | ((uint64_t)m->rope_mode << 8), 0 /* workspace base, below */,
                      envh("PCY_ESM_ATTN") ^ (envh("PCY_DISABLE") << 1), envh("PCY_GEMM_MID") ^ (envh("PCY_GEMM_PERM") << 1),
                      envh("PCY_DEBUG_POISON_WS")};
  // the workspace is sized (and possibly re-allocated: every slot is dropped then) before the key is final
  {
    const int d = m->d, F = m->ffn;
    const size_t vtt = align_up((size_t)ntok + 64 * (size_t)nseq, 8) > (size_t)vt_total ? align_up((size_t)ntok + 64 * (size_t)nseq, 8) : (size_t)vt_total;
    const size_t need = align_up((size_t)ntok * d * 2, 256) * 3 + align_up((size_t)ntok * 3 * d * 2, 256) + align_up((size_t)ntok * F * 2, 256) +
                        align_up((size_t)d * vtt * 2, 256) + align_up((size_t)(nseq + 1) * 4, 256) + 4096;
    if (int r = c->reserve(need)) return r;
  }
  key[12] = (uint64_t)(uintptr_t)c->ws;
  pcy_ctx::GraphSlot* slot = nullptr;
  for (auto& g : c->gslots)
    if (g.stamp && memcmp(g.key, key, sizeof(key)) == 0) { slot = &g; break; }
  if (slot && slot->exec) {
    slot->stamp = ++c->gstamp;
    ++g_pcy_dispatch[PCY_DISPATCH_ESM_GRAPH];
    HIP_TRY(hipGraphLaunch(slot->exec, c->stream));
    return 0;
  }
  if (!slot) {   // first sight of this key: remember it (evicting the least recently used slot), run launch by launch
    slot = &c->gslots[0];
    for (auto& g : c->gslots)
      if (g.stamp < slot->stamp) slot = &g;
    if (slot->exec) hipGraphExecDestroy(slot->exec);
    *slot = pcy_ctx::GraphSlot{};
    memcpy(slot->key, key, sizeof(key));
    slot->stamp = ++c->gstamp;
    return esm_encode_enqueue(c, m, tokens, pos, cu, vt_cu, ntok, nseq, max_len, vt_total, mask_pads, hidden_out);
  }
  // second sight: capture (on the capture stream -- the caller's may be the legacy stream), then replay on the caller's stream
  pcy_gemm_prepare(c->stream);   // one-time device tables must not be built inside the capture
  hipGraph_t g = nullptr;
  hipStream_t user = c->stream;
  c->stream = c->cap_stream;
  hipError_t e0 = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal);
  int rc = 0;
  if (e0 == hipSuccess) {
    rc = esm_encode_enqueue(c, m, tokens, pos, cu, vt_cu, ntok, nseq, max_len, vt_total, mask_pads, hidden_out);
    e0 = hipStreamEndCapture(c->cap_stream, &g);
  }
  c->stream = user;
  if (e0 != hipSuccess || rc != 0 || !g) {   // could not capture: forget the key, run launch by launch
    (void)hipGetLastError();
    if (g) hipGraphDestroy(g);
    *slot = pcy_ctx::GraphSlot{};
    return esm_encode_enqueue(c, m, tokens, pos, cu, vt_cu, ntok, nseq, max_len, vt_total, mask_pads, hidden_out);
  }
  e0 = hipGraphInstantiate(&slot->exec, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e0 != hipSuccess) {
    (void)hipGetLastError();
    *slot = pcy_ctx::GraphSlot{};
    return esm_encode_enqueue(c, m, tokens, pos, cu, vt_cu, ntok, nseq, max_len, vt_total, mask_pads, hidden_out);
  }
  slot->stamp = ++c->gstamp;
  ++g_pcy_dispatch[PCY_DISPATCH_ESM_GRAPH];
  HIP_TRY(hipGraphLaunch(slot->exec, c->stream));
  return 0;
}
static int esm_encode_enqueue(pcy_ctx* c, const pcy_esm_desc* m, const int32_t* tokens, const int32_t* pos, const int32_t* cu,
                              const int32_t* vt_cu, int ntok, int nseq, int max_len, int vt_total, int mask_pads, void* hidden_out) {
  const int d = m->d, H = m->n_heads, F = m->ffn;
  if (d % H) return fail(1, "pcy_esm_encode: d %% heads");
  const int dh = d / H;
  if (dh != 32 && dh != 64 && dh != 128) return fail(1, "pcy_esm_encode: head_dim %d unsupported (32/64/128)", dh);
  if (d % 64 || F % 64) return fail(1, "pcy_esm_encode: d and ffn must be multiples of 64");
  if (ntok <= 0) return 0;
  // single-pass attention (default at head_dim 64): Vt slices zero-padded to 64 keys, laid out by the engine itself
  const bool fast = pcy_attn_fast_eligible(dh, 0, false, 1.0f, H, H, 3 * d, 0, 3 * d, d, d);
  if (fast) vt_total = (int)align_up((size_t)ntok + 64 * (size_t)nseq, 8);
  const size_t need = align_up((size_t)ntok * d * 2, 256) * 3 + align_up((size_t)ntok * 3 * d * 2, 256) +
                      align_up((size_t)ntok * F * 2, 256) + align_up((size_t)d * vt_total * 2, 256) + align_up((size_t)(nseq + 1) * 4, 256) + 4096;
  if (int r = c->reserve(need)) return r;
  Carver cv(c->ws);
  bf16_t* x = cv.take<bf16_t>((size_t)ntok * d);
  bf16_t* xn = cv.take<bf16_t>((size_t)ntok * d);
  bf16_t* ao = cv.take<bf16_t>((size_t)ntok * d);
  bf16_t* qkv = cv.take<bf16_t>((size_t)ntok * 3 * d);
  bf16_t* act = cv.take<bf16_t>((size_t)ntok * F);
  bf16_t* vt = cv.take<bf16_t>((size_t)d * vt_total);
  int32_t* vt_cu64 = cv.take<int32_t>((size_t)nseq + 1);
  const bool vrow = fast && pcy_attn_fast_vrow(3 * d, 2 * d);   // the single-pass kernel reads V out of qkv: no transposed copy
  if (fast && !vrow) {
    pcy_launch_vt_offsets(c->stream, cu, nseq, 64, vt_cu64);
    vt_cu = vt_cu64;
  }
  hipStream_t s = c->stream;
  pcy_launch_esm_embed(s, (const bf16_t*)m->embed, tokens, cu, nseq, max_len, x, d, mask_pads);
  const float qscale = 1.0f / sqrtf((float)dh);
  for (int l = 0; l < m->n_layers; ++l) {
    const pcy_esm_layer& L = m->layers[l];
    pcy_launch_layernorm(s, x, (const bf16_t*)L.ln1_w, (const bf16_t*)L.ln1_b, xn, ntok, d, m->ln_eps);
    if (dh == 64 && ntok > 8 && d % 64 == 0) {
      // rotary (q pre-scaled) fused into the projection's epilogue: saves a read+write pass over q and k
      PcyGemmArgs g{};
      g.A = xn; g.W = (const bf16_t*)L.wqkv; g.C = qkv; g.bias = (const bf16_t*)L.bqkv;
      g.M = ntok; g.N = 3 * d; g.K = d; g.lda = d; g.ldc = 3 * d; g.epi = EPI_STORE;
      g.rope_pos = pos; g.rope_cos = (const bf16_t*)m->rope_cos; g.rope_sin = (const bf16_t*)m->rope_sin;
      g.rope_ncols = 2 * d; g.rope_qcols = d; g.rope_mode = m->rope_mode; g.rope_scale = qscale;
      pcy_launch_gemm(s, g);
    } else {
      linear(s, xn, d, (const bf16_t*)L.wqkv, (const bf16_t*)L.bqkv, nullptr, 0, qkv, 3 * d, ntok, 3 * d, d, EPI_STORE);
      pcy_launch_rope(s, qkv, 3 * d, 0, H, dh, pos, (const bf16_t*)m->rope_cos, (const bf16_t*)m->rope_sin, ntok, m->rope_mode, qscale);
      pcy_launch_rope(s, qkv, 3 * d, d, H, dh, pos, (const bf16_t*)m->rope_cos, (const bf16_t*)m->rope_sin, ntok, m->rope_mode, 0.f);
    }
    if (!vrow) pcy_launch_transpose_v(s, qkv, 3 * d, 2 * d, H, dh, cu, vt_cu, nseq, max_len, vt, vt_total);
    PcyAttnArgs t{};
    if (vrow) { t.v = qkv; t.ldv = 3 * d; t.vcol0 = 2 * d; }
    t.q = qkv; t.ldq = 3 * d; t.qcol0 = 0; t.k = qkv; t.ldk = 3 * d; t.kcol0 = d; t.vt = vt; t.vt_total = vt_total;
    t.o = ao; t.ldo = d; t.cu = cu; t.vt_cu = vt_cu; t.keep = nullptr; t.nseq = nseq; t.max_len = max_len; t.H = H; t.Hkv = H;
    t.dh = dh; t.causal = 0; t.scale = 1.0f; t.vt_pad64 = fast ? 1 : 0;
    pcy_launch_attn(s, t);
    linear(s, ao, d, (const bf16_t*)L.wo, (const bf16_t*)L.bo, x, d, x, d, ntok, d, d, EPI_RESID);
    pcy_launch_layernorm(s, x, (const bf16_t*)L.ln2_w, (const bf16_t*)L.ln2_b, xn, ntok, d, m->ln_eps);
    linear(s, xn, d, (const bf16_t*)L.w1, (const bf16_t*)L.b1, nullptr, 0, act, F, ntok, F, d, EPI_GELU_ESM);
    linear(s, act, F, (const bf16_t*)L.w2, (const bf16_t*)L.b2, x, d, x, d, ntok, d, F, EPI_RESID);
  }
  pcy_launch_layernorm(s, x, (const bf16_t*)m->final_ln_w, (const bf16_t*)m->final_ln_b, (bf16_t*)hidden_out, ntok, d, m->ln_eps);
  return check_launch("pcy_esm_encode");
}

}  // extern "C"
namespace {
int llama_prefill_impl(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const void* embeds, const uint8_t* keep,
                       const int32_t* pos, const int32_t* cu, const int32_t* vt_cu, int B, int T, const int32_t* logit_rows,
                       int n_logit_rows, void* logits_out, void* hidden_out, const int32_t* sum_rows, int n_sum_rows,
                       void* hidden_sum_out, void* hidden_all_out) {
  PCY_STICKY(c);
  const int d = m->d, H = m->n_heads, Hkv = m->n_kv_heads, dh = m->head_dim, F = m->ffn;
  if (dh != 32 && dh != 64 && dh != 128) return fail(1, "pcy_llama_prefill: head_dim %d unsupported (32/64/128)", dh);
  if (d % 64 || F % 64 || (H * dh) % 64 || (Hkv * dh) % 64) return fail(1, "pcy_llama_prefill: d, ffn, H*dh, Hkv*dh must be multiples of 64");
  if (F % 16) return fail(1, "pcy_llama_prefill: ffn %% 16");
  if (B > kv->B || T > kv->Tmax) return fail(1, "pcy_llama_prefill: B=%d T=%d exceed cache (%d,%d)", B, T, kv->B, kv->Tmax);
  if (T > m->max_pos) return fail(1, "pcy_llama_prefill: T=%d exceeds rope table %d", T, m->max_pos);
  const int M = B * T, qkvw = (H + 2 * Hkv) * dh, Tp = (T + 31) / 32 * 32;
  // Row stride of the transposed V ([Hkv*dh][vt_total]): a power-of-two stride (one 512-token prompt: 1 KB) sends the 128 rows of a
  // key block's V tile to the same few L2 channels; pad it to an odd multiple of 64 bytes
  int vt_total = B * Tp;
  if ((vt_total / 32) % 2 == 0) vt_total += 32;
  const size_t need = align_up((size_t)M * d * 2, 256) * 2 + align_up((size_t)M * qkvw * 2, 256) + align_up((size_t)M * H * dh * 2, 256) +
                      align_up((size_t)M * F * 2, 256) + align_up((size_t)Hkv * dh * vt_total * 2, 256) +
                      align_up((size_t)(n_logit_rows + 1) * d * 2, 256) + align_up((size_t)(n_sum_rows + 1) * d * 6, 256) +
                      (M <= 1024 ? align_up((size_t)8 * M * qkvw * 4, 256) : 0) + 4096 +
                      (m->layers_fp8 ? align_up((size_t)M * (F > H * dh ? F : H * dh), 256) + align_up((size_t)M * 4, 256) : 0);
  if (n_sum_rows > 0 && (!sum_rows || !hidden_sum_out)) return fail(1, "pcy_llama_prefill: sum_rows / hidden_sum_out missing");
  if (m->layers_fp8 && (d % 128 || F % 128 || (H * dh) % 128)) return fail(1, "pcy_llama_prefill: the fp8 path needs d, ffn, H*dh %% 128 == 0");
  if (int r = c->reserve(need)) return r;
  Carver cv(c->ws);
  bf16_t* x = cv.take<bf16_t>((size_t)M * d);
  bf16_t* xn = cv.take<bf16_t>((size_t)M * d);
  bf16_t* qkv = cv.take<bf16_t>((size_t)M * qkvw);
  bf16_t* ao = cv.take<bf16_t>((size_t)M * H * dh);
  bf16_t* act = cv.take<bf16_t>((size_t)M * F);
  bf16_t* vt = cv.take<bf16_t>((size_t)Hkv * dh * vt_total);
  bf16_t* lastx = cv.take<bf16_t>((size_t)(n_logit_rows + 1) * d);
  const size_t sk_bytes = M <= 1024 ? (size_t)8 * M * qkvw * 4 : 0;   // split-K partials for the projections that under-fill the chip
  float* sk_ws = sk_bytes ? cv.take<float>(sk_bytes / 4) : nullptr;
  float* hsum = cv.take<float>((size_t)(n_sum_rows + 1) * d);      // ret_token_access='all': fp32 sum of the L+1 hidden states
  bf16_t* hsum_tmp = cv.take<bf16_t>((size_t)(n_sum_rows + 1) * d);
  hipStream_t s = c->stream;
  // fp8 weight path: every projection = per-token e4m3 quantisation of its bf16 input + the fp8 MFMA GEMM
  unsigned char* a8 = m->layers_fp8 ? cv.take<unsigned char>((size_t)M * (F > H * dh ? F : H * dh)) : nullptr;
  float* sa8 = m->layers_fp8 ? cv.take<float>((size_t)M) : nullptr;
  // ln != nullptr: A is the raw hidden state, RMSNorm(A) * ln is what gets quantised (one fused pass; PCY_DISABLE=fp8_fused_norm = two launches)
  auto linear8 = [&](const bf16_t* A, int K, const void* W8, const float* sw, const bf16_t* resid, bf16_t* Cout, int ldc, int N, int epi,
                     const bf16_t* ln = nullptr) {
    if (ln && !pcy_off("fp8_fused_norm") && pcy_launch_rmsnorm_quant_fp8(s, A, ln, M, K, m->rms_eps, m->rms_cast, a8, sa8)) {
    } else {
      if (ln) { pcy_launch_rmsnorm(s, A, ln, xn, M, K, m->rms_eps, m->rms_cast); A = xn; }
      pcy_launch_quant_rows_fp8(s, A, K, M, K, a8, sa8);
    }
    PcyGemmArgs g{};
    g.A = (const bf16_t*)a8; g.W = (const bf16_t*)W8; g.C = Cout; g.resid = resid; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldc = ldc;
    g.ldr = ldc; g.epi = epi; g.fp8 = 1; g.sa = sa8; g.sw = sw;
    pcy_launch_gemm(s, g);
  };
  HIP_TRY(hipMemcpyAsync(x, embeds, (size_t)M * d * 2, hipMemcpyDeviceToDevice, s));
  // hidden_all_out [L+1][M][d]: HF's `hidden_states` tuple -- the embeddings, the output of layers 0..L-2, and the FINAL-NORMED
  // output of layer L-1 (pmc_llama.py:575,584 always asks for it; only materialised here when the caller does)
  bf16_t* hall = (bf16_t*)hidden_all_out;
  if (hall) HIP_TRY(hipMemcpyAsync(hall, embeds, (size_t)M * d * 2, hipMemcpyDeviceToDevice, s));
  // hidden_states = (embeddings, output of layers 0..L-2, final-normed output of layer L-1)  [HF LlamaModel.forward]
  pcy_launch_acc_rows(s, x, d, sum_rows, hsum, n_sum_rows, d, 1);
  const size_t layer_stride = (size_t)kv->B * Hkv * kv->Tmax * dh;
  int xn_ready = 0;   // xn = RMSNorm(x) of the next projection already written by a K-split finish launch
  for (int l = 0; l < m->n_layers; ++l) {
    const pcy_llama_layer& L = m->layers[l];
    const pcy_llama_layer_fp8* L8 = m->layers_fp8 ? &m->layers_fp8[l] : nullptr;
    if (L8) {
      linear8(x, d, L8->wqkv, L8->sqkv, nullptr, qkv, qkvw, qkvw, EPI_STORE, (const bf16_t*)L.ln1);
    } else {
      if (!xn_ready) pcy_launch_rmsnorm(s, x, (const bf16_t*)L.ln1, xn, M, d, m->rms_eps, m->rms_cast);
      xn_ready = 0;
      linear(s, xn, d, (const bf16_t*)L.wqkv, nullptr, nullptr, 0, qkv, qkvw, M, qkvw, d, EPI_STORE, sk_ws, sk_bytes);
    }
    if (pcy_off("prefill_post_qkv") ||   // (the three launches: the test compares both)
        !pcy_launch_prefill_post_qkv(s, qkv, qkvw, H, Hkv, dh, pos, (const bf16_t*)m->rope_cos, (const bf16_t*)m->rope_sin,
                                     (bf16_t*)kv->k + l * layer_stride, (bf16_t*)kv->v + l * layer_stride, B, T, kv->Tmax, cu, vt_cu, vt, vt_total)) {
      pcy_launch_rope(s, qkv, qkvw, 0, H + Hkv, dh, pos, (const bf16_t*)m->rope_cos, (const bf16_t*)m->rope_sin, M, 0, 0.f);
      pcy_launch_kv_scatter(s, qkv, qkvw, H * dh, (H + Hkv) * dh, Hkv, dh, (bf16_t*)kv->k + l * layer_stride,
                            (bf16_t*)kv->v + l * layer_stride, B, T, kv->Tmax);
      pcy_launch_transpose_v(s, qkv, qkvw, (H + Hkv) * dh, Hkv, dh, cu, vt_cu, B, T, vt, vt_total);
    }
    PcyAttnArgs t{};
    t.q = qkv; t.ldq = qkvw; t.qcol0 = 0; t.k = qkv; t.ldk = qkvw; t.kcol0 = H * dh; t.vt = vt; t.vt_total = vt_total;
    t.o = ao; t.ldo = H * dh; t.cu = cu; t.vt_cu = vt_cu; t.keep = keep; t.nseq = B; t.max_len = T; t.H = H; t.Hkv = Hkv; t.dh = dh;
    t.causal = 1; t.scale = 1.0f / sqrtf((float)dh);
    pcy_launch_attn(s, t);
    if (L8) linear8(ao, H * dh, L8->wo, L8->so, x, x, d, d, EPI_RESID);
    else linear(s, ao, H * dh, (const bf16_t*)L.wo, nullptr, x, d, x, d, M, d, H * dh, EPI_RESID, sk_ws, sk_bytes,
                (const bf16_t*)L.ln2, xn, &xn_ready, m->rms_eps, m->rms_cast);   // (M <= 1024: the K-split finish also writes RMSNorm(x) * ln2)
    if (!L8 && !xn_ready) pcy_launch_rmsnorm(s, x, (const bf16_t*)L.ln2, xn, M, d, m->rms_eps, m->rms_cast);
    xn_ready = 0;
    if (L8) {
      linear8(x, d, L8->wgu, L8->sgu, nullptr, act, F, 2 * F, EPI_SWIGLU, (const bf16_t*)L.ln2);
      linear8(act, F, L8->wdown, L8->sdown, x, x, d, d, EPI_RESID);
      if (l + 1 < m->n_layers) pcy_launch_acc_rows(s, x, d, sum_rows, hsum, n_sum_rows, d, 0);
      if (hall && l + 1 < m->n_layers) HIP_TRY(hipMemcpyAsync(hall + (size_t)(l + 1) * M * d, x, (size_t)M * d * 2, hipMemcpyDeviceToDevice, s));
      continue;
    }
    if (M <= 8) {
      PcyGemvArgs u{};
      u.W = (const bf16_t*)L.wgu; u.x = xn; u.y = act; u.N = F; u.K = d; u.B = M; u.ldx = d; u.ldy = F; u.epi = EPI_SWIGLU;
      pcy_launch_gemv(s, u);
    } else {
      linear(s, xn, d, (const bf16_t*)L.wgu, nullptr, nullptr, 0, act, F, M, 2 * F, d, EPI_SWIGLU);
    }
    if (l + 1 < m->n_layers && !m->layers_fp8)   // ... and the next layer's input norm
      linear(s, act, F, (const bf16_t*)L.wdown, nullptr, x, d, x, d, M, d, F, EPI_RESID, sk_ws, sk_bytes,
             (const bf16_t*)m->layers[l + 1].ln1, xn, &xn_ready, m->rms_eps, m->rms_cast);
    else linear(s, act, F, (const bf16_t*)L.wdown, nullptr, x, d, x, d, M, d, F, EPI_RESID, sk_ws, sk_bytes);
    if (l + 1 < m->n_layers) pcy_launch_acc_rows(s, x, d, sum_rows, hsum, n_sum_rows, d, 0);
    if (hall && l + 1 < m->n_layers) HIP_TRY(hipMemcpyAsync(hall + (size_t)(l + 1) * M * d, x, (size_t)M * d * 2, hipMemcpyDeviceToDevice, s));
  }
  if (hall) pcy_launch_rmsnorm(s, x, (const bf16_t*)m->final_norm, hall + (size_t)m->n_layers * M * d, M, d, m->rms_eps, m->rms_cast);
  if (hidden_out) pcy_launch_rmsnorm(s, x, (const bf16_t*)m->final_norm, (bf16_t*)hidden_out, M, d, m->rms_eps, m->rms_cast);
  if (n_sum_rows > 0) {
    pcy_launch_copy_rows(s, x, d, hsum_tmp, d, sum_rows, n_sum_rows, d);
    pcy_launch_rmsnorm(s, hsum_tmp, (const bf16_t*)m->final_norm, hsum_tmp, n_sum_rows, d, m->rms_eps, m->rms_cast);
    pcy_launch_acc_rows(s, hsum_tmp, d, nullptr, hsum, n_sum_rows, d, 0);
    pcy_launch_acc_finish(s, hsum, (bf16_t*)hidden_sum_out, (size_t)n_sum_rows * d);
  }
  if (hall && n_logit_rows > 64 && logits_out) {
    // pcy_llama_prefill_all with many rows (the reference's full [B,T,V] logits): final norm of those rows, then lm_head as an MFMA
    // GEMM -- the GEMV would stream the 1 GB matrix once per 32 rows.  ONLY on this entry point: the GEMM accumulates in another
    // order than the fused-norm GEMV, and a row's logits from pcy_llama_prefill must not depend on how many rows were asked for
    // (QA / pair scoring with 64 or 65 rows: same bits per row; tests/test_gpu_round4.py).
    pcy_launch_copy_rows(s, x, d, lastx, d, logit_rows, n_logit_rows, d);
    pcy_launch_rmsnorm(s, lastx, (const bf16_t*)m->final_norm, lastx, n_logit_rows, d, m->rms_eps, m->rms_cast);
    linear(s, lastx, d, (const bf16_t*)m->lm_head, nullptr, nullptr, 0, (bf16_t*)logits_out, m->vocab, n_logit_rows, m->vocab, d, EPI_STORE);
  } else if (n_logit_rows > 0 && logits_out) {
    // final norm of the selected rows, then lm_head on the MFMA GEMV (32 rows per pass over the matrix) for EVERY row count, one row
    // included: the same arithmetic per row whatever the number of rows asked for (the fused-norm streaming GEMV it replaces ran 4 rows
    // per pass: QA / pair scoring with more than 64 rows streamed the 1 GB matrix rows / 4 times -- advisor finding, round 4)
    pcy_launch_copy_rows(s, x, d, lastx, d, logit_rows, n_logit_rows, d);
    PcyGemvArgs h{};
    h.W = (const bf16_t*)m->lm_head; h.x = lastx; h.y = (bf16_t*)logits_out; h.N = m->vocab; h.K = d; h.B = n_logit_rows; h.ldx = d; h.ldy = m->vocab;
    h.epi = EPI_STORE;
    if (d % 512 == 0) {
      pcy_launch_rmsnorm(s, lastx, (const bf16_t*)m->final_norm, lastx, n_logit_rows, d, m->rms_eps, m->rms_cast);
      h.force_mfma = 1;
    } else {   // (toy geometries: the streaming kernel with the norm fused)
      h.rms_w = (const bf16_t*)m->final_norm; h.rms_eps = m->rms_eps; h.rms_cast = m->rms_cast;
    }
    pcy_launch_gemv(s, h);
  }
  return check_launch("pcy_llama_prefill");
}
}  // namespace
extern "C" {
int pcy_llama_prefill(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const void* embeds, const uint8_t* keep,
                      const int32_t* pos, const int32_t* cu, const int32_t* vt_cu, int B, int T, const int32_t* logit_rows,
                      int n_logit_rows, void* logits_out, void* hidden_out, const int32_t* sum_rows, int n_sum_rows,
                      void* hidden_sum_out) {
  return llama_prefill_impl(c, m, kv, embeds, keep, pos, cu, vt_cu, B, T, logit_rows, n_logit_rows, logits_out, hidden_out, sum_rows,
                            n_sum_rows, hidden_sum_out, nullptr);
}
int pcy_llama_prefill_all(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const void* embeds, const uint8_t* keep,
                          const int32_t* pos, const int32_t* cu, const int32_t* vt_cu, int B, int T, const int32_t* logit_rows,
                          int n_logit_rows, void* logits_out, void* hidden_all_out) {
  if (!hidden_all_out) return fail(1, "pcy_llama_prefill_all: hidden_all_out is NULL");
  return llama_prefill_impl(c, m, kv, embeds, keep, pos, cu, vt_cu, B, T, logit_rows, n_logit_rows, logits_out, nullptr, nullptr, 0,
                            nullptr, hidden_all_out);
}

int pcy_llama_decode(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B) {
  PCY_STICKY(c);
  if (B > kv->B) return fail(1, "pcy_llama_decode: B=%d exceeds cache rows %d", B, kv->B);
  if (int r = c->reserve(decode_ws_bytes(m, B, kv->Tmax))) return r;
  if (int r = ensure_decode_state(c, m, B)) return r;
  enqueue_decode(c, m, kv, st, B);
  return check_launch("pcy_llama_decode");
}

int pcy_llama_decode_layers(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B, int reps) {
  PCY_STICKY(c);
  if (B > kv->B) return fail(1, "pcy_llama_decode_layers: B=%d exceeds cache rows %d", B, kv->B);
  if (int r = c->reserve(decode_ws_bytes(m, B, kv->Tmax))) return r;
  if (int r = ensure_decode_state(c, m, B)) return r;
  for (int i = 0; i < reps; ++i) enqueue_decode(c, m, kv, st, B, true);
  return check_launch("pcy_llama_decode_layers");
}

int pcy_greedy_pick(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B, int advance_pos) {
  PCY_STICKY(c);
  if (int r = c->reserve(decode_ws_bytes(m, B, kv->Tmax))) return r;
  enqueue_pick(c, m, st, B, advance_pos, kv->Tmax);
  return check_launch("pcy_greedy_pick");
}

namespace {
// capture (once per model / cache / state / batch) and replay the decode step; kind 0 = decode + greedy pick, 1 = decode only
int replay_decode_graph(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B, int n_steps, int kind) {
  // EVERYTHING a captured kernel argument was derived from: every pointer of the state / cache / model the enqueue functions
  // read, AND the geometry (an allocator may hand a new state or a cache of another capacity the addresses of the previous one
  // while e.g. the `keep` mask or the token buffer differ -- a stale graph would then run with dangling arguments)
  const void* key[pcy_ctx::GRAPH_KEY_N] = {m, m->layers, m->embed, kv->k, kv->v, st->pos, st->step, st->next_tok, st->tokens_out, st->logprob,
                                           st->logits, st->logits_all, st->keep, c->ws,
                                           (const void*)(intptr_t)(((int64_t)st->logits_all_ld << 32) ^ kv->Tmax),
                                           (const void*)(intptr_t)(((int64_t)kv->B << 32) ^ st->max_steps),
                                           (const void*)((uintptr_t)c->mc_tags ^ ((uintptr_t)c->mb_flags << 1)), c->dev_layers, (const void*)(uintptr_t)c->layers_fp};
  if (!c->graph || memcmp(key, c->graph_key, sizeof(key)) != 0 ||
      c->graph_B != B || c->graph_mode != decode_mode() || c->graph_kind != kind) {
    c->drop_graph();
    hipGraph_t g = nullptr;
    hipStream_t user = c->stream;
    c->stream = c->cap_stream;
    hipError_t e0 = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal);
    if (e0 == hipSuccess) {
      enqueue_decode(c, m, kv, st, B);
      if (kind == 0) enqueue_pick(c, m, st, B, 1, kv->Tmax);
      e0 = hipStreamEndCapture(c->cap_stream, &g);
    }
    c->stream = user;
    if (e0 != hipSuccess) return fail(2, "decode-step graph capture failed: %s", hipGetErrorString(e0));
    HIP_TRY(hipGraphInstantiate(&c->graph, g, nullptr, nullptr, 0));
    hipGraphDestroy(g);
    memcpy(c->graph_key, key, sizeof(key));
    c->graph_B = B;
    c->graph_mode = decode_mode();
    c->graph_kind = kind;
  }
  for (int i = 0; i < n_steps; ++i) HIP_TRY(hipGraphLaunch(c->graph, c->stream));
  return 0;
}
}  // namespace

int pcy_llama_greedy(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B, int n_steps,
                     int use_graph) {
  PCY_STICKY(c);
  if (B > kv->B) return fail(1, "pcy_llama_greedy: B=%d exceeds cache rows %d", B, kv->B);
  if (int r = c->reserve(decode_ws_bytes(m, B, kv->Tmax))) return r;
  if (int r = ensure_decode_state(c, m, B)) return r;
  if (!use_graph) {
    for (int i = 0; i < n_steps; ++i) {
      enqueue_decode(c, m, kv, st, B);
      enqueue_pick(c, m, st, B, 1, kv->Tmax);
    }
    return check_launch("pcy_llama_greedy");
  }
  return replay_decode_graph(c, m, kv, st, B, n_steps, 0);
}

int pcy_llama_decode_graph(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B) {
  PCY_STICKY(c);
  if (B > kv->B) return fail(1, "pcy_llama_decode_graph: B=%d exceeds cache rows %d", B, kv->B);
  if (int r = c->reserve(decode_ws_bytes(m, B, kv->Tmax))) return r;
  if (int r = ensure_decode_state(c, m, B)) return r;
  return replay_decode_graph(c, m, kv, st, B, 1, 1);
}

int pcy_sample_pick(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B, int advance_pos,
                    float temperature, float nucleus_prob, const float* uniforms, void* probs_out) {
  PCY_STICKY(c);
  if (!(temperature > 0.f)) return fail(1, "pcy_sample_pick: temperature must be > 0 (greedy: pcy_greedy_pick)");
  if (nucleus_prob >= 1.f) return fail(1, "pcy_sample_pick: nucleus_prob must be < 1 (<= 0 switches the nucleus mask off)");
  if (m->vocab > pcy_sample_max_vocab()) return fail(1, "pcy_sample_pick: vocabulary %d unsupported (<= %d)", m->vocab, pcy_sample_max_vocab());
  if (int r = c->reserve(decode_ws_bytes(m, B, kv->Tmax))) return r;
  if (int r = ensure_sample_state(c, B, m->vocab)) return r;
  enqueue_sample(c, m, st, B, advance_pos, kv->Tmax, temperature, nucleus_prob, uniforms, (bf16_t*)probs_out);
  return check_launch("pcy_sample_pick");
}

int pcy_llama_sample(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B, int n_steps,
                     float temperature, float nucleus_prob, const float* uniforms) {
  PCY_STICKY(c);
  if (B > kv->B) return fail(1, "pcy_llama_sample: B=%d exceeds cache rows %d", B, kv->B);
  if (!(temperature > 0.f) || nucleus_prob >= 1.f) return fail(1, "pcy_llama_sample: temperature > 0 and nucleus_prob < 1 required");
  if (m->vocab > pcy_sample_max_vocab()) return fail(1, "pcy_llama_sample: vocabulary %d unsupported (<= %d)", m->vocab, pcy_sample_max_vocab());
  if (int r = c->reserve(decode_ws_bytes(m, B, kv->Tmax))) return r;
  if (int r = ensure_decode_state(c, m, B)) return r;
  if (int r = ensure_sample_state(c, B, m->vocab)) return r;
  for (int i = 0; i < n_steps; ++i) {
    enqueue_decode(c, m, kv, st, B);
    enqueue_sample(c, m, st, B, 1, kv->Tmax, temperature, nucleus_prob, uniforms, nullptr);
  }
  return check_launch("pcy_llama_sample");
}

int pcy_beam_step(pcy_ctx* c, const void* logits, int vocab, int B, int beam, int group_size, float diversity_penalty,
                  const pcy_beam_state* st) {
  PCY_STICKY(c);
  if (B <= 0 || beam <= 0 || beam > 32 || group_size <= 0 || beam % group_size)
    return fail(1, "pcy_beam_step: beam=%d (1..32) must be a multiple of group_size=%d", beam, group_size);
  if (vocab <= 0 || vocab > 163840) return fail(1, "pcy_beam_step: vocab %d unsupported (<= 163840)", vocab);
  PcyBeamState b{};
  b.out = st->out; b.max_len = st->max_len; b.cur = st->cur; b.cur_new = st->cur_new; b.next_tok = st->next_tok; b.src = st->src;
  b.anc = st->anc; b.has_eos = st->has_eos; b.blk_eos = st->blk_eos; b.ticket = st->ticket; b.pos = st->pos; b.step = st->step;
  b.done = st->done; b.eos_id = st->eos_id;
  // row-statistics partials in an allocation of their own: never inside the decode workspace (a captured decode graph points
  // into that one, and its tail belongs to the KV-reorder scratch)
  const size_t need = align_up(pcy_beam_ws_bytes(B, beam), 256);
  if (need > c->beam_ws_bytes) {
    if (c->beam_ws) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->beam_ws)); c->beam_ws = nullptr; c->beam_ws_bytes = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->beam_ws), need * 2));
    c->beam_ws_bytes = need * 2;
  }
  void* ws = c->beam_ws;
  pcy_launch_beam_step(c->stream, (const bf16_t*)logits, vocab, B, beam, group_size, diversity_penalty, b, ws);
  return check_launch("pcy_beam_step");
}

int pcy_kv_reorder(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const int32_t* src_rows, int B, int t) {
  return pcy_kv_reorder_range(c, m, kv, src_rows, B, 0, t);
}
int pcy_kv_reorder_range(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const int32_t* src_rows, int B, int t0, int t) {
  PCY_STICKY(c);
  const int Hkv = m->n_kv_heads, dh = m->head_dim, L = m->n_layers;
  if (t <= 0 || t0 >= t) return 0;
  if (t0 < 0) return fail(1, "pcy_kv_reorder_range: t0 < 0");
  if ((t * dh) % 8 || (t0 * dh) % 8) return fail(1, "pcy_kv_reorder: t * head_dim (and t0 * head_dim) must be multiples of 8");
  // scratch sized for the cache capacity, not for t: a workspace that grows step by step would be re-allocated (and the
  // captured decode graph dropped) several times per beam search
  const size_t tmp_elems = (size_t)2 * L * B * Hkv * kv->Tmax * dh;
  if (int r = c->reserve(decode_ws_bytes(m, B, kv->Tmax) + align_up(tmp_elems * 2, 256) + 4096)) return r;
  bf16_t* tmp = reinterpret_cast<bf16_t*>(c->ws + align_up(decode_ws_bytes(m, B, kv->Tmax), 256));
  if (enqueue_kv_permute(c->stream, m, kv, src_rows, B, t, nullptr, t0)) return check_launch("pcy_kv_reorder");
  const dim3 grid(B, Hkv, 2 * L);
  hipLaunchKernelGGL(kv_gather_kernel, grid, dim3(256), 0, c->stream, (bf16_t*)kv->k, (bf16_t*)kv->v, tmp, src_rows, B, kv->B, Hkv,
                     kv->Tmax, t, dh, 1, (const int32_t*)nullptr, t0);
  hipLaunchKernelGGL(kv_gather_kernel, grid, dim3(256), 0, c->stream, (bf16_t*)kv->k, (bf16_t*)kv->v, tmp, src_rows, B, kv->B, Hkv,
                     kv->Tmax, t, dh, 0, (const int32_t*)nullptr, t0);
  return check_launch("pcy_kv_reorder");
}

// Steps 1 .. of the diverse beam search as ONE replayed launch chain per step: decode step (the graph of pcy_llama_decode_graph's launches)
// -> record of the step's logits (row (step, b) of logits_rec, step read from the device) -> pcy_beam_step -> pcy_kv_reorder over the
// slots the position counter names.  Same kernels, same order, same bits as the four calls (PCY_DISABLE=beam_graph: the callers keep
// them); it removes the ~12 launch boundaries of a step (3.43 -> see DESIGN.md ms per step at beam 5).
int pcy_llama_beam_steps(pcy_ctx* c, const pcy_llama_desc* m, const pcy_kv_cache* kv, const pcy_gen_state* st, int B, int beam, int group_size,
                         float diversity_penalty, const pcy_beam_state* bs, void* logits_rec, int n_steps, int kv_t0) {
  PCY_STICKY(c);
  if (kv_t0 < 0 || (kv_t0 * m->head_dim) % 8) return fail(1, "pcy_llama_beam_steps: kv_t0 = %d (>= 0, kv_t0 * head_dim a multiple of 8)", kv_t0);
  const int BB = B * beam;
  if (B <= 0 || beam <= 0 || beam > 32 || group_size <= 0 || beam % group_size)
    return fail(1, "pcy_llama_beam_steps: beam=%d (1..32) must be a multiple of group_size=%d", beam, group_size);
  if (BB > kv->B) return fail(1, "pcy_llama_beam_steps: %d rows exceed cache rows %d", BB, kv->B);
  if (m->vocab <= 0 || m->vocab > 163840) return fail(1, "pcy_llama_beam_steps: vocab %d unsupported (<= 163840)", m->vocab);
  if (st->pos != bs->pos || st->next_tok != bs->next_tok) return fail(1, "pcy_llama_beam_steps: the decode state must read the beam state's position and tokens");
  if ((kv->Tmax * m->head_dim) % 8) return fail(1, "pcy_llama_beam_steps: Tmax * head_dim must be a multiple of 8");
  if (n_steps <= 0) return 0;
  // everything the chain allocates, before the capture: decode workspace + the reorder scratch behind it, beam scratch, decode state
  const int Hkv = m->n_kv_heads, dh = m->head_dim, L = m->n_layers;
  const size_t tmp_elems = (size_t)2 * L * BB * Hkv * kv->Tmax * dh;
  if (int r = c->reserve(decode_ws_bytes(m, BB, kv->Tmax) + align_up(tmp_elems * 2, 256) + 4096)) return r;
  if (int r = ensure_decode_state(c, m, BB)) return r;
  const size_t need = align_up(pcy_beam_ws_bytes(B, beam), 256);
  if (need > c->beam_ws_bytes) {
    if (c->beam_ws) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->beam_ws)); c->beam_ws = nullptr; c->beam_ws_bytes = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->beam_ws), need * 2));
    c->beam_ws_bytes = need * 2;
    c->drop_graph();
  }
  uint32_t pen_bits; memcpy(&pen_bits, &diversity_penalty, 4);
  // every array of the beam state is baked into the chain: all of them go into the key (FNV-1a), not only the ones listed by name below
  uint64_t bsh = 1469598103934665603ull;
  {
    const unsigned char* pb = reinterpret_cast<const unsigned char*>(bs);
    for (size_t i = 0; i < sizeof(pcy_beam_state); ++i) bsh = (bsh ^ pb[i]) * 1099511628211ull;
  }
  // ... and what enqueue_decode bakes into the step besides: the key-keep mask, the hand-over slots / flags of the fused steps
  uint64_t fold = 1469598103934665603ull;
  for (const void* q : {(const void*)st->keep, (const void*)c->mc_tags, (const void*)c->dev_layers, (const void*)c->mb_flags,
                        (const void*)(BB <= 8 ? c->nb_tags[BB] : nullptr)})
    fold = (fold ^ (uint64_t)(uintptr_t)q) * 1099511628211ull;
  const void* key[pcy_ctx::GRAPH_KEY_N] = {m, m->layers, m->embed, kv->k, kv->v, st->pos, st->step, st->next_tok, bs->out, bs->cur,
                                           st->logits, logits_rec, bs->src, c->ws,
                                           (const void*)(intptr_t)(((int64_t)beam << 40) ^ ((int64_t)group_size << 32) ^ kv->Tmax),
                                           (const void*)(intptr_t)(((int64_t)kv->B << 32) ^ pen_bits ^ ((int64_t)kv_t0 << 44)),
                                           c->beam_ws, bs->anc, (const void*)(uintptr_t)(c->layers_fp ^ bsh ^ fold)};
  if (!c->graph || memcmp(key, c->graph_key, sizeof(key)) != 0 || c->graph_B != BB || c->graph_mode != decode_mode() || c->graph_kind != 2) {
    c->drop_graph();
    PcyBeamState b{};
    b.out = bs->out; b.max_len = bs->max_len; b.cur = bs->cur; b.cur_new = bs->cur_new; b.next_tok = bs->next_tok; b.src = bs->src;
    b.anc = bs->anc; b.has_eos = bs->has_eos; b.blk_eos = bs->blk_eos; b.ticket = bs->ticket; b.pos = bs->pos; b.step = bs->step;
    b.done = bs->done; b.eos_id = bs->eos_id;
    bf16_t* tmp = reinterpret_cast<bf16_t*>(c->ws + align_up(decode_ws_bytes(m, BB, kv->Tmax), 256));
    hipGraph_t g = nullptr;
    hipStream_t user = c->stream;
    c->stream = c->cap_stream;
    hipError_t e0 = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal);
    if (e0 == hipSuccess) {
      hipStream_t s = c->stream;
      enqueue_decode(c, m, kv, st, BB);
      if (logits_rec)
        hipLaunchKernelGGL(store_logits_kernel, dim3(BB >= 8 ? 256 : 64), dim3(256), 0, s, (const bf16_t*)st->logits, (bf16_t*)logits_rec, bs->step, BB,
                           m->vocab, m->vocab);
      pcy_launch_beam_step(s, (const bf16_t*)st->logits, m->vocab, B, beam, group_size, diversity_penalty, b, c->beam_ws);
      if (!enqueue_kv_permute(s, m, kv, bs->src, BB, 0, (const int32_t*)bs->pos, kv_t0)) {
        const dim3 grid(BB, Hkv, 2 * L);
        hipLaunchKernelGGL(kv_gather_kernel, grid, dim3(256), 0, s, (bf16_t*)kv->k, (bf16_t*)kv->v, tmp, bs->src, BB, kv->B, Hkv, kv->Tmax, 0, dh, 1,
                           (const int32_t*)bs->pos, kv_t0);
        hipLaunchKernelGGL(kv_gather_kernel, grid, dim3(256), 0, s, (bf16_t*)kv->k, (bf16_t*)kv->v, tmp, bs->src, BB, kv->B, Hkv, kv->Tmax, 0, dh, 0,
                           (const int32_t*)bs->pos, kv_t0);
      }
      e0 = hipStreamEndCapture(c->cap_stream, &g);
    }
    c->stream = user;
    if (e0 != hipSuccess) return fail(2, "beam-step graph capture failed: %s", hipGetErrorString(e0));
    HIP_TRY(hipGraphInstantiate(&c->graph, g, nullptr, nullptr, 0));
    hipGraphDestroy(g);
    memcpy(c->graph_key, key, sizeof(key));
    c->graph_B = BB;
    c->graph_mode = decode_mode();
    c->graph_kind = 2;
  }
  for (int i = 0; i < n_steps; ++i) HIP_TRY(hipGraphLaunch(c->graph, c->stream));
  return 0;
}

}  // extern "C"

extern "C" int pcy_debug_mc_trace(unsigned long long* out, int n) {
  if (!g_mc_trace) return 0;
  hipDeviceSynchronize();
  hipMemcpy(out, g_mc_trace, (size_t)n * 8, hipMemcpyDeviceToHost);
  return n;
}
