// Internal launcher prototypes shared by the engine translation units.
#pragma once
#include "pcy_common.h"
#include "pcy_switch.h"

struct PcyGemvArgs {
  const bf16_t* W;      // [N,K] row-major (nn.Linear layout); EPI_SWIGLU: [2N,K], 16-row gate/up interleave
  const bf16_t* x;      // [B,K] (ldx elements between rows)
  bf16_t* y;            // [B,N] (ldy)
  const bf16_t* bias;   // [N] or null
  const bf16_t* resid;  // [B,N] (ldy) or null (EPI_RESID); may alias y
  const bf16_t* rms_w;  // non-null: x is the raw hidden state; RMSNorm(x)*rms_w is fused in the prologue
  float rms_eps;
  int rms_cast;         // 0: w * bf16(x_hat) (transformers>=4.32) ; 1: bf16(w * x_hat) (4.31)
  int N, K, B, ldx, ldy, epi;
  int plain_loads;      // debug A/B: 0 = non-temporal weight loads (default), 1 = default cache policy
  // optional fp32 workspace [ksplit][B][N] for the batched (B > 4) kernel's K-split partial sums
  float* splitk_ws; size_t splitk_ws_bytes;
  // optional (batched K-split path with the residual epilogue only): the finish kernel also writes
  // next_xn = RMSNorm(y) * next_rms_w (rms_eps / rms_cast above) and sets *fused_next = 1; otherwise *fused_next stays 0 and
  // the caller launches the norm itself.  Same summation order as rmsnorm_kernel -> identical bits.
  const bf16_t* next_rms_w; bf16_t* next_xn; int* fused_next;
  // optional (batched K-split path, plain epilogue without bias): leave the partial sums in splitk_ws, launch no finish kernel and
  // report the number of splits in *defer_finish (0: y has been written as usual) -- the consumer adds them up itself
  // (decode attention: PcyDecAttnArgs::qkv_partials)
  int* defer_finish;
  // 1: the streaming kernel (one dot product per row and lane, rows in groups of <= 4) whatever the batch -- the launch-per-stage twin of the
  // small-batch decode step (pcy_decode_nb.hip) is built from it
  int force_stream;
  // 1: the MFMA kernel (x already normalised, rms_w == NULL) whatever the batch, one row included -- the prefill's lm_head, so that a row's
  // logits are the same bits for every number of rows asked for AND 32 rows share a pass over the 1 GB matrix
  int force_mfma;
  // 1 (streaming kernel, one row, K % 512 == 0): the dot product of output row r (feature r for EPI_SWIGLU) walks its 512-element k-iterations
  // rotated by (r / 4) % (K / 512).  All waves of the chip otherwise read the same 1 KB window of their rows at the same moment, and with the
  // 8 KB row stride of K = 4096 those windows fall on the same HBM channels (round 6, tools/probes/stream_rows.hip).  The order of the sum is
  // a function of the row alone: the fused multi-head decode step (pcy_decode_mha.hip) uses the same one, so the two stay bit-identical twins.
  int krot;
};
void pcy_launch_gemv(hipStream_t s, const PcyGemvArgs& a);

// Rotated K order of the batched GEMVs (round 6).  Every workgroup of these kernels walks ITS K range front to back in step with all the
// others, so at any moment every wave of the chip reads the same 256-byte window of its rows -- and with a power-of-two row stride (K = 4096:
// 8 KB) those windows fall on the same few HBM channels: tools/probes/stream_rows.hip measures the gate/up stream at 4.86 TB/s, and 6.26 TB/s
// when workgroups start at different places of the range (padding the rows by 256 B gives the same 6.3).  A workgroup therefore starts
// `shift` 128-k steps into its range and wraps around: a function of the 192-row group of its FIRST weight row only (192 = lcm of the row
// blocks the launches (64) and the mid-batch step (48 / 64) cut the qkv matrix into: a row gets the same order whoever computes it, whatever
// the batch), in units of 512 k (the x chunk of gemv_mfma2_kernel), zero when the range is not a multiple of 512.  The summation order of a
// row's dot product is rotated with it -- the same for every kernel of the family and for the fused step, so they stay bit-identical twins.
__host__ __device__ inline int pcy_gemv_kshift(int r0_wg, int nss) {
  const int nch = nss >> 2;
  return ((nss & 3) == 0 && nch > 1) ? ((r0_wg / 192 * 5) % nch) * 4 : 0;
}

// y = bf16(bf16(sum of four interleaved K-block partial sums) [+ resid]): the down projection of the small-batch decode step's launch-per-stage
// twin (pcy_decode_nb.hip); K % 2048 == 0, no bias / norm; false = not covered
bool pcy_launch_gemv_kwin4(hipStream_t s, const PcyGemvArgs& a);

// Launches whose workgroups wait for each other INSIDE the launch need every workgroup resident at once: true when the occupancy
// query says `grid` workgroups of `block` threads with `smem` bytes of dynamic LDS fit on `n_cu` compute units at the same time
// (the check a cooperative launch would make, without its +15-19 us per launch; MI355X_MICROARCH.md, coop-launch row).
template <typename Kernel>
inline bool pcy_all_resident(Kernel kernel, int block, size_t smem, int grid, int n_cu) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, smem) != hipSuccess) { (void)hipGetLastError(); return false; }
  return nb >= 1 && (long)nb * n_cu >= (long)grid;
}
int pcy_mfma_min_batch();
// finish of a K-split projection with the residual epilogue + RMSNorm of the result (gemv_splitk_finish_norm_kernel): partial sums
// ws [splits][rows][N] fp32; y = bf16(bf16(sum) + resid) (ldy == N), xn = RMSNorm(y) * w.  false = shape not covered.
bool pcy_launch_splitk_finish_norm(hipStream_t s, const float* ws, int splits, int rows, int N, const bf16_t* resid, bf16_t* y,
                                   const bf16_t* next_rms_w, bf16_t* next_xn, float rms_eps, int rms_cast);   // smallest batch that takes the MFMA GEMVs

// Batch-1 decode, the MLP of a layer in ONE launch (pcy_gemv.hip, mlp_chain_kernel; the body also runs inside the decode layer launch):
//   act = SwiGLU(RMSNorm(x) * ln2 . Wgu^T) ;  x_out = x + act . Wdown^T
// One workgroup per CU, all resident; `act` travels between the stages as {tag : bf16} words (no flags, no drains), see
// pcy_handover.h / pcy_mlp_chain.h.  Bit-identical to the two stand-alone GEMV launches.
struct PcyMlpChainArgs {
  const bf16_t* x; bf16_t* x_out;          // residual stream [d] (may alias)
  const bf16_t* ln2; const bf16_t* wgu;    // [d], [2F, d] gate/up interleaved in blocks of 16 rows
  const bf16_t* wdown;                     // [d, F]
  int d, F; float rms_eps; int rms_cast;
  uint32_t* act_tag;                       // [F] tagged hand-over vector, private to THIS launch's layer
  const unsigned* epoch;                   // device word advanced once per decode step that runs these launches (tag = low 16 bits)
  unsigned* err;                           // watchdog word
  unsigned long long* trace;               // measurement aid: [grid][16] time stamps (nullptr: none)
};
void pcy_launch_qa_probs(hipStream_t s, const void* logits, int is_f32, int rows, int V, int yes_id, int no_id, void* probs_out, float* yes_no_out,
                         int32_t* argmax_out);   // softmax over the vocabulary at the QA answer row (+ yes / no columns, argmax)
void pcy_launch_bump(hipStream_t s, unsigned* word);   // *word += 1 (tag counters, pcy_handover.h)
// false = geometry not covered (nothing launched)
bool pcy_launch_mlp_chain(hipStream_t s, const PcyMlpChainArgs& a, int n_cu);

struct PcyGemmArgs {
  const bf16_t* A;      // [M,K] lda
  const bf16_t* W;      // [N,K] row-major; EPI_SWIGLU: N counts interleaved gate/up rows (output width N/2)
  bf16_t* C;            // [M,N] ldc (or [M,N/2])
  const bf16_t* bias;   // [N] or null
  const bf16_t* resid;  // [M,N] ldr or null; may alias C
  int M, N, K, lda, ldc, ldr, epi;
  int gn;               // column tiles per rasterisation group (set by pcy_launch_gemm)
  // optional rotary embedding fused into the EPI_STORE epilogue (head_dim 64 only): columns [0, rope_ncols) are heads
  // of 64 features rotated with cos/sin rows of pos[token]; columns < rope_qcols are first multiplied by rope_scale and
  // rounded (ESM q.dh^-1/2); rope_mode 0 = every product a bf16 tensor (HF Llama), 1 = fp32, rounded once (HF ESM)
  const int32_t* rope_pos; const bf16_t* rope_cos; const bf16_t* rope_sin; int rope_ncols, rope_qcols, rope_mode; float rope_scale;
  // optional split-K workspace (fp32 [splits][M][N]); the launcher splits K when the tile count under-fills the chip
  float* splitk_ws; size_t splitk_ws_bytes;
  // fp8 path (BASELINE configs[4]): A and W point to OCP e4m3 bytes ([M,K] lda bytes / [N,K]), K % 128 == 0;
  // C = epi( bf16-rounding chain of ((acc * sa[m]) * sw[n]) ), sa / sw = per-token / per-output-row dequantisation scales
  int fp8; const float* sa; const float* sw;
  // optional (split-K path with the residual epilogue only): the finish kernel also writes next_xn = RMSNorm(C) * next_rms_w and
  // sets *fused_next = 1; otherwise *fused_next stays 0 and the caller launches the norm itself.  Same bits as the two launches.
  const bf16_t* next_rms_w; bf16_t* next_xn; int* fused_next; float rms_eps; int rms_cast;
  int gelu_select;      // set by the launcher (PCY_DISABLE=gelu_fast): the ESM-GELU kernel skips the fast table epilogue
  int mid_cfg;          // > 0: this configuration of gemm_kernel_mid (pcy_gemm_mid.h); 0: the launcher's own choice
};
struct pcy_ctx;
hipStream_t pcy_ctx_stream(pcy_ctx* c);               // the context's stream (pcy_engine.hip owns the struct)
void pcy_set_error(const char* fmt, ...);             // sets the text pcy_last_error() returns
void pcy_launch_gemm(hipStream_t s, const PcyGemmArgs& a);
// builds the one-time device tables of the GEMM epilogues (the ESM GELU table) on `s` if this device has none yet -- callers that
// CAPTURE a chain of launches call it first, so that the build is not recorded into the graph
void pcy_gemm_prepare(hipStream_t s);
// launch counters per kernel family (pcy_debug_dispatch_count)
enum { PCY_DISPATCH_GEMM_128 = 0, PCY_DISPATCH_GEMM_64 = 1, PCY_DISPATCH_GEMM_BIG = 2, PCY_DISPATCH_GEMM_BIG_PERSIST = 3,
       PCY_DISPATCH_GEMM_SPLITK = 4, PCY_DISPATCH_GEMM_FP8 = 5, PCY_DISPATCH_ATTN_FAST = 6, PCY_DISPATCH_UNUSED_7 = 7, PCY_DISPATCH_GEMM_MID = 8, PCY_DISPATCH_ESM_GRAPH = 9, PCY_DISPATCH_N = 10 };
extern unsigned long long g_pcy_dispatch[PCY_DISPATCH_N];

// per-row symmetric e4m3 quantisation: scale[r] = smallest power of two with amax|x[r,:]| / scale <= 448 (1 for an all-zero row), q = e4m3_rne(x / scale)
void pcy_launch_quant_rows_fp8(hipStream_t s, const bf16_t* x, int ldx, int rows, int K, unsigned char* q, float* scale);
// RMSNorm(x) * w -> per-token e4m3 codes + scales in one pass, bit-identical to pcy_launch_rmsnorm + pcy_launch_quant_rows_fp8;
// false (nothing launched) for d > 8192
bool pcy_launch_rmsnorm_quant_fp8(hipStream_t s, const bf16_t* x, const bf16_t* w, int rows, int d, float eps, int cast,
                                  unsigned char* q, float* scale);
void pcy_launch_rmsnorm(hipStream_t s, const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int d, float eps, int cast);
void pcy_launch_layernorm(hipStream_t s, const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int rows, int d, float eps);
void pcy_launch_embed_gather(hipStream_t s, const bf16_t* table, const int32_t* ids, const bf16_t* soft,
                             const int32_t* soft_map, bf16_t* out, int rows, int d);
// epoch (optional): device word incremented by the launch (decode step: epoch of the in-launch hand-overs)
void pcy_launch_embed_tokens_dev(hipStream_t s, const bf16_t* table, const int32_t* ids, bf16_t* out, int rows, int d, unsigned* epoch = nullptr,
                                 unsigned* epoch2 = nullptr);
void pcy_launch_esm_embed(hipStream_t s, const bf16_t* table, const int32_t* toks, const int32_t* cu, int nseq,
                          int max_len, bf16_t* out, int d, int mask_pads);
// rope on heads [0,nh) located at column col0 of a token-major buffer; pos[tok] = rotary position.
// mode 0: three roundings in bf16 (HF Llama); mode 1: fp32 once (HF ESM). prescale != 0: x = bf16(x*prescale) first.
void pcy_launch_rope(hipStream_t s, bf16_t* buf, int ld, int col0, int nh, int dh, const int32_t* pos,
                     const bf16_t* cos_t, const bf16_t* sin_t, int ntok, int mode, float prescale);
// scatter roped K and V of a token-major qkv buffer into the [B,Hkv,Tmax,dh] cache at slots [0,T)
// rope (q, k in place) + K/V cache scatter + V transpose of a rectangular prefill batch in ONE launch; false = shape not covered
bool pcy_launch_prefill_post_qkv(hipStream_t s, bf16_t* qkv, int ld, int H, int Hkv, int dh, const int32_t* pos, const bf16_t* cos_t,
                                 const bf16_t* sin_t, bf16_t* kcache, bf16_t* vcache, int B, int T, int Tmax, const int32_t* cu,
                                 const int32_t* vt_cu, bf16_t* vt, int vt_total);
void pcy_launch_kv_scatter(hipStream_t s, const bf16_t* qkv, int ld, int kcol0, int vcol0, int Hkv, int dh,
                           bf16_t* kcache, bf16_t* vcache, int B, int T, int Tmax);
// V (token-major, column vcol0, nh heads) -> Vt[nh][dh][vt_total], sequence q occupies columns vt_cu[q]+j
void pcy_launch_transpose_v(hipStream_t s, const bf16_t* buf, int ld, int vcol0, int nh, int dh, const int32_t* cu,
                            const int32_t* vt_cu, int nseq, int max_len, bf16_t* vt, int vt_total);

struct PcyAttnArgs {
  const bf16_t* q; int ldq; int qcol0;   // token-major, head h at column qcol0 + h*dh
  const bf16_t* k; int ldk; int kcol0;   // token-major, kv head at kcol0 + kvh*dh
  const bf16_t* vt; int vt_total;        // [Hkv][dh][vt_total]
  bf16_t* o; int ldo;                    // token-major [ntok, H*dh]
  const int32_t* cu; const int32_t* vt_cu;  // [nseq+1]
  const uint8_t* keep;                   // per token key-keep flag (attention_mask) or null
  int nseq, max_len, H, Hkv, dh, causal;
  float scale;                           // multiplied into bf16 scores, then rounded (1.0 = none)
  int vt_pad64;                          // every sequence's Vt slice starts at a multiple of 8 and is zero-padded to a multiple of 64 keys
  // optional, single-pass kernel only: V token-major (head h at column vcol0 + h*dh) -- set instead of vt / vt_cu / vt_total, no
  // transposed copy is made at all (vt_pad64 still says "the caller may take the single-pass kernel")
  const bf16_t* v; int ldv; int vcol0;
};
// PCY_DISABLE=fa_vrow (read per call): the single-pass attention reads a transposed copy of V (the first form) instead of V itself
bool pcy_attn_fast_vrow(int ldv, int vcol0);
void pcy_launch_attn(hipStream_t s, const PcyAttnArgs& a);
// the single-pass kernel (pcy_attn_fast.h) covers this call unless PCY_ESM_ATTN=exact asks for the reference's rounding points
bool pcy_attn_fast_eligible(int dh, int causal, bool has_keep, float scale, int H, int Hkv, int ldq, int qcol0, int ldk, int kcol0, int ldo);
// vt_cu_out[q] = sum over earlier sequences of their length rounded up to `pad` (vt_cu_out[nseq] = total), on the device
void pcy_launch_vt_offsets(hipStream_t s, const int32_t* cu, int nseq, int pad, int32_t* vt_cu_out);

struct PcyDecAttnArgs {
  bf16_t* qkv; int ld;            // [B, (H+2Hkv)*dh] un-roped projections of the new token
  bf16_t* kcache; bf16_t* vcache; // [B,Hkv,Tmax,dh]
  bf16_t* o; int ldo;             // [B, H*dh]
  const int32_t* pos_dev;         // device scalar: cache length t == rotary position of the new token
  const bf16_t* cos_t; const bf16_t* sin_t;  // [max_pos, dh]
  const uint8_t* keep; int ld_keep;  // optional [B,Tmax] key-keep mask ("clean" mode) or null (reference quirk Q1)
  float* scratch;                 // [B*H*Tmax] fp32 probabilities workspace
  int B, H, Hkv, dh, Tmax; float scale; int dbg;
  // persistent decode kernel only (zero otherwise): cache length + 1 if already known; `o` written through to memory
  // with agent-scope stores
  int t_plus1; int o_sc1;
  // key split across the column-slice workgroups of a kv head (fused batch-1 launch only; nullptr = every workgroup scores all
  // keys): flags [B*Hkv*DH/DS] for THIS launch, epoch value of this decode step, minimum cache length, watchdog word;
  // `scratch` ([B*H*(Tmax+1)] fp32) carries the exchanged scores
  unsigned* xflags; unsigned xepoch; int xmin; unsigned* xerr;
  int unit_map;                   // fused launch: 1 = kv head in the low digits of the workgroup index (slices of a head share an XCD)
  // attention block launch only (device-side fields, zero otherwise): the new token's roped-to-be q heads / k / v of this kv head
  // staged in LDS as [G + 2][dh] (nullptr: read from `qkv`); `o` stored as {tag : bf16} words into o_tag instead
  const bf16_t* staged; uint32_t* o_tag; uint32_t tag;
  // batched decode: the qkv projection's K-split partial sums [qkv_splits][B][ld] fp32 (nullptr: `qkv` holds the finished rows).
  // The attention workgroup adds the G + 2 rows of ITS kv head in split order (what gemv_splitk_finish_kernel does: same bits)
  // into LDS -- one launch less per decoder layer.
  const float* qkv_partials; int qkv_splits;
  // stand-alone launch only: output columns per workgroup (16 / 32 / 64 / 128; 0 = the launcher's choice) -- the summation order of P.V
  // depends on it, and the launch-per-stage twin of the small-batch decode step must use the fused launch's
  int force_ds;
};
void pcy_launch_attn_decode(hipStream_t s, const PcyDecAttnArgs& a);
struct PcyGemvArgs;
// decode attention + o projection (EPI_RESID GEMV over the attention output) in one launch; false = shape not covered,
// nothing launched.  epoch: device word that differs between consecutive calls on the same `flags` (max_flags words).
bool pcy_launch_attn_o(hipStream_t s, const PcyDecAttnArgs& a, const PcyGemvArgs& o, int n_cu, const unsigned* epoch,
                       unsigned* flags, int max_flags, unsigned* err, unsigned* xflags = nullptr);

// Batch-1 decode, ONE launch per decoder layer (pcy_attn.hip, decode_layer_kernel):
//   qkv = RMSNorm(x) * ln1 . Wqkv^T ;  attention over the cache (+ append) ;  x' = x + attn . Wo^T ;  x_out = x' + MLP(x')
// Workgroups [0, n_attn) run the decode attention body (cache rows requested at once, the new token's q/k/v taken from the
// tagged qkv vector when it arrives); the others project qkv (all their rows in registers from the first cycle), pull their Wo
// rows and the first gate/up rows while the attention runs and finish the o projection; then every workgroup runs the MLP
// body (mc).  Bit-identical to the launch-per-stage step.
struct PcyAttnBlockArgs {
  const bf16_t* x;                          // residual stream [d] (the result goes to mc.x_out, which may alias)
  const bf16_t* ln1; const bf16_t* wqkv;    // [d], [Nq, d]
  const bf16_t* wo;                         // [d, H*dh]
  int d, Nq; float rms_eps; int rms_cast;
  uint32_t* qkv_tag; uint32_t* ao_tag;      // [Nq], [H*dh] tagged hand-over vectors, private to THIS layer's launch
  uint32_t* xo_tag;                         // [d] the residual stream after the o projection
  const unsigned* epoch;                    // tag counter (see pcy_handover.h)
  unsigned* err;
  unsigned long long* trace;                // measurement aid: [grid][16] time stamps (nullptr: none)
};
// All layers in one launch (decode_step_kernel): device array of the layers' weights, per-layer strides of the cache / tag slots,
// and the tagged vectors that carry the residual stream across the layer boundaries.
struct PcyLayerWeightsDev { const bf16_t *ln1, *wqkv, *wo, *ln2, *wgu, *wdown; };
struct PcyDecodeStepArgs {
  const PcyLayerWeightsDev* layers; int n_layers;   // device memory
  size_t kv_layer_stride;                           // elements between the K (V) caches of consecutive layers
  uint32_t* tags; size_t tag_stride;                // per-layer hand-over slots: act [F] | qkv [Nq] | attention output [H dh] | x after o [d]
  size_t xflags_stride;
  uint32_t* x_lines; size_t x_lines_stride;         // [n_layers - 1][32 * 256] words: the residual stream behind layer l
};
// false = geometry not covered (nothing launched).  xflags / step_epoch: key-split exchange of the attention workgroups
// (as pcy_launch_attn_o).  mc: the layer's MLP (mc.x unused).
// p / mc: geometry and the per-step pointers (x, epoch, err); their per-layer fields are filled in by the kernel.
bool pcy_launch_decode_step(hipStream_t s, const PcyDecAttnArgs& a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, const PcyDecodeStepArgs& st,
                            int n_cu, const unsigned* step_epoch);
bool pcy_launch_decode_layer(hipStream_t s, const PcyDecAttnArgs& a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, int n_cu,
                             const unsigned* step_epoch, unsigned* xflags);
// The same for the multi-head geometry (pcy_decode_mha.hip: 32 kv heads of 128, d 4096, 7168 < ffn <= 14336 -- Llama-2-7B / ProCyon-Split):
// one layer (st == nullptr) or all layers in one launch; tried first by the two launchers above.  The launch-per-stage twin's attention must
// cut its output into pcy_decode_mha_ds() columns per workgroup (the summation order of P.V).
bool pcy_decode_mha_covers(int d, int H, int Hkv, int dh, int F, int n_cu);
int pcy_decode_mha_ds();
bool pcy_launch_decode_mha(hipStream_t s, PcyDecAttnArgs a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc, const PcyDecodeStepArgs* st, int n_cu,
                           const unsigned* step_epoch, unsigned* xflags);
// Small-batch decode step (pcy_decode_nb.hip): every decoder layer for 2..8 rows in ONE launch, the weights streamed once.  Hand-over slots
// per layer: pcy_decode_nb_tag_words(B) words (act | qkv | attention output | x after o, B rows each), residual stream between two layers:
// pcy_decode_nb_line_words(B) words; p.epoch = the tag counter of THIS batch size's slots.  false = not covered, nothing launched.
size_t pcy_decode_nb_tag_words(int B);
size_t pcy_decode_nb_line_words(int B);
int pcy_decode_nb_ds(int B);   // output columns per attention workgroup of the B-row step (what the launch-per-stage twin must use)
bool pcy_launch_decode_step_nb(hipStream_t s, int device, const PcyDecAttnArgs& a, const PcyAttnBlockArgs& p, const PcyMlpChainArgs& mc,
                               const PcyDecodeStepArgs& st, int n_cu, const unsigned* step_epoch, int B, int xmin);
bool pcy_decode_nb_launchable(int device, int B, int Tmax, int n_cu);   // pcy_launch_decode_step_nb would launch (LDS for this Tmax, residency)
// threads of the stand-alone RMS-fused GEMV launch for N output rows (the summation order of its statistic)
int pcy_gemv_rms_threads(int N);

// Mid-batch decode step (pcy_decode_mb.hip): every decoder layer of a step for 9..32 rows in ONE launch -- the batched (skinny-MFMA) GEMVs
// of the launch-per-stage path, its decode attention and its K-split finish + RMSNorm as phases of one persistent kernel, the weight rings
// running ahead across the phase boundaries.  Geometry: Llama-3-8B (d 4096, ffn 14336, 32 / 8 heads of 128), 256 CUs.
struct PcyMbArgs {
  const PcyLayerWeightsDev* layers; int n_layers;   // device table of the layers' weights
  const bf16_t* final_norm;                         // [d] the norm behind the last layer (its result is what lm_head reads)
  bf16_t* x;                                        // [B][d] residual stream (in: the embedded tokens; out: the last layer's output)
  bf16_t* xn;                                       // [B][d] in: RMSNorm(x) * ln1 of layer 0; out: RMSNorm(x_out) * final_norm
  bf16_t* ao;                                       // [B][d] attention output
  bf16_t* act;                                      // [B][ffn] SwiGLU output
  float* qkv_ws;                                    // [2][B][Nq] K-split partial sums of the qkv projection
  float* sk_ws;                                     // [4][B][d] K-split partial sums of the o / down projections
  bf16_t* kcache; bf16_t* vcache; size_t kv_layer_stride; int Bcache;
  const int32_t* pos_dev; const bf16_t* cos_t; const bf16_t* sin_t; const uint8_t* keep; int ld_keep;
  int B, Tmax; float scale, rms_eps; int rms_cast;
  unsigned* flags;                                  // [(n_layers + 1)][pcy_decode_mb_flag_words()] arrival flags (value = step epoch)
  const unsigned* epoch;                            // device word advanced once per step that runs this launch
  unsigned* err;                                    // watchdog word
  unsigned long long* trace;                        // measurement aid: [layer][256][16] time stamps (nullptr: none)
  int abl;                                          // timing ablations (PCY_MB_ABL, tools only: the results are wrong)
};
size_t pcy_decode_mb_flag_words();
int pcy_decode_mb_ds(int B);                        // output columns per attention workgroup of the B-row step (the twin must use the same)
bool pcy_decode_mb_fits(int B, int Tmax);           // LDS of the attention phase fits beside the weight rings
// false = not covered, nothing launched
bool pcy_launch_decode_step_mb(hipStream_t s, int device, const PcyMbArgs& a, int n_cu);

// pooled[i] over token ranges rng[seg[i]..seg[i+1]) = (start,len) pairs; mode 0 mean, 1 mean-corrected, 2 max
size_t pcy_pool_ws_bytes(int nprot, int d);
void pcy_launch_pool(hipStream_t s, const bf16_t* h, int d, const int32_t* seg, const int32_t* rng, int nprot,
                     int mode, bf16_t* out, void* ws);
// per-row argmax (lowest index on ties) over bf16 logits [B,V]; accumulates log_softmax(logits)[tok] into
// logprob[B] (bf16 log-softmax, fp32 running sum), appends tok to tokens_out[b*max_steps + step], writes
// next_tok[b], then (one thread) ++*step and, if advance_pos, ++*pos.
void pcy_launch_greedy_pick(hipStream_t s, const bf16_t* logits, int B, int V, int32_t* next_tok, int32_t* tokens_out,
                            int max_steps, float* logprob, int32_t* pos_dev, int32_t* step_dev, int advance_pos,
                            void* partials /* B*64*16 bytes of scratch */);
// sampling / nucleus selection of one step (model_unified.py:896-906): token ~ multinomial(probs) by inverse CDF with the caller's
// uniform variate uniforms[step * B + b]; nucleus_p <= 0: plain temperature sampling.  hist: [B][65536] uint32, zero on entry / exit;
// partials: B * 64 * 16 bytes; probs_out: optional [B,V] record of the pre-sampling probability vector
void pcy_launch_retrieval_dot_f32(hipStream_t s, const float* query, int Q, const void* targets, int targets_bf16, int N, int D, float eps, float* sims);
size_t pcy_retrieval_dot_smem(int D);
void pcy_launch_retrieval_rank_f32(hipStream_t s, const float* sims, int Q, int N, int k, int32_t* idx_out, float* score_out);
int pcy_sample_max_vocab();   // largest vocabulary the selection workgroup of the sampling step covers (SMP_NT * SMP_KMAX)
void pcy_launch_sample_step(hipStream_t s, const bf16_t* logits, int B, int V, float temperature, float nucleus_p, const float* uniforms,
                            unsigned* hist, bf16_t* probs_out, int32_t* next_tok, int32_t* tokens_out, int max_steps, float* logprob,
                            int32_t* pos_dev, int32_t* step_dev, int advance_pos, void* partials /* 64 x float4 per row */,
                            bf16_t* pbits /* [B][V] scratch: the probabilities' bits */);
// device-side state of the diverse beam search (pcy_beam_step); every pointer is device memory, BB = B * beam rows
struct PcyBeamState {
  int32_t* out; int32_t max_len;   // [2][BB][max_len] token histories, buffer (step & 1) is current
  float* cur; float* cur_new;      // [BB] running scores (+ scratch)
  int32_t* next_tok;               // [BB] tokens fed to the next decode step
  int32_t* src;                    // [BB] parent slot of every slot after this step (KV reorder, logits record)
  int32_t* anc;                    // optional [max_len][BB] record of `src` per step
  uint8_t* has_eos;                // [2][BB]
  int32_t* blk_eos; int32_t* ticket;   // [B], [1]
  int32_t* pos; int32_t* step; int32_t* done;   // device scalars: cache length, step index, all-rows-hold-an-EOS flag
  int32_t eos_id;
};
size_t pcy_beam_ws_bytes(int B, int beam);   // scratch of pcy_launch_beam_step (row-statistics partials)
void pcy_launch_beam_step(hipStream_t s, const bf16_t* logits, int V, int B, int beam, int g, float penalty, const PcyBeamState& st,
                          void* ws);
void pcy_launch_copy_rows(hipStream_t s, const bf16_t* src, int lds, bf16_t* dst, int ldd, const int32_t* rows,
                          int nrows, int d);
void pcy_launch_l2norm_rows(hipStream_t s, const bf16_t* x, bf16_t* y, int rows, int d, float eps);
// per row of sims [Q,N]: the k best entries in stable descending order (ties: lower index first) -> idx_out / score_out [Q,k]
void pcy_launch_retrieval_rank(hipStream_t s, const bf16_t* sims, int Q, int N, int k, int32_t* idx_out, bf16_t* score_out);
// acc[r][:] (fp32) = / += src[rows[r]][:]; out = bf16(acc)
void pcy_launch_acc_rows(hipStream_t s, const bf16_t* src, int lds, const int32_t* rows, float* acc, int nrows, int d, int first);
void pcy_launch_acc_finish(hipStream_t s, const float* acc, bf16_t* out, size_t n);
