/* pcy.h -- C ABI of the MI355X-native ProCyon forward/generation engine (libpcy.so).
 *
 * The reference (mims-harvard/ProCyon) is pure Python and has no FFI; this ABI is the boundary
 * the build introduces BELOW the reference's sub-module waist:
 *
 *   LlamaPostTokenization.forward   procyon/model/pmc_llama.py:546-596     -> pcy_llama_prefill / pcy_llama_decode*
 *   ESM_PLM.forward                 procyon/model/esm.py:504-558           -> pcy_esm_encode + pcy_pool
 *   ProteinPooler.forward           procyon/model/esm.py:131-173           -> pcy_pool
 *   create_mlp stacks               procyon/model/model_utils.py:13-41     -> pcy_mlp_forward
 *   _prepare_input_embeddings       procyon/model/model_unified.py:1135    -> pcy_embed_splice
 *   _generate_sampling (greedy)     procyon/model/model_unified.py:861-921 -> pcy_llama_greedy
 *
 * Conventions: plain pointers and sizes only.  Every `const void*` / `void*` tensor argument is a
 * DEVICE pointer to bf16 (raw 16-bit) data unless stated; int32 index arrays are device pointers
 * too.  The caller owns all tensors (the engine never frees them); the engine owns only its
 * scratch workspace.  Every call enqueues on the context's HIP stream and returns without
 * synchronising.  Return value 0 = ok, otherwise an error code with text in pcy_last_error()
 * (thread-local).  Nothing throws across the ABI.  One context per device/stream; a context is
 * not thread-safe.
 */
#ifndef PCY_H
#define PCY_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCY_ABI_VERSION 10

typedef struct pcy_ctx pcy_ctx;

/* epilogue selectors of pcy_gemm / pcy_gemv (bf16 rounding points in procyon_amd/csrc/pcy_common.h) */
enum { PCY_EPI_STORE = 0, PCY_EPI_RESID = 1, PCY_EPI_GELU_ERF = 2, PCY_EPI_GELU_ESM = 3, PCY_EPI_SWIGLU = 4 };
/* pooling modes of ProteinPooler (esm.py:138-149) */
enum { PCY_POOL_MEAN = 0, PCY_POOL_MEAN_CORRECTED = 1, PCY_POOL_MAX = 2 };

int pcy_abi_version(void);
/* Test instrumentation: number of GEMM launches that went to kernel family `kind` since the library was loaded
 * (0: 128x128 tiles, 1: 64x64, 2: 256x256, 3: 256x256 persistent (ESM fc1 + GELU), 4: split-K, 5: fp8 256x256).  Parity
 * tests use it to assert that they reach the kernel they claim to test. */
unsigned long long pcy_debug_dispatch_count(int kind);
const char* pcy_last_error(void);
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for the default stream */
int pcy_ctx_create(int device_id, void* stream, pcy_ctx** out);
void pcy_ctx_destroy(pcy_ctx* ctx);
int pcy_ctx_sync(pcy_ctx* ctx);
/* HIP-event timer on the context's stream (bench.py roofline leg): start, stop -> elapsed ms */
int pcy_timer_start(pcy_ctx* ctx);
int pcy_timer_stop(pcy_ctx* ctx, float* ms_out);

/* ---- primitive ops (unit parity tests; also what the host composes for odd shapes) ---------- */
/* C[M,N] = epi(A[M,K] . W[N,K]^T + bias); EPI_SWIGLU: W is [N,K] with 16-row gate/up interleave, C is [M,N/2] */
int pcy_gemm(pcy_ctx*, const void* A, int lda, const void* W, const void* bias, const void* resid, int ldr,
             void* C, int ldc, int M, int N, int K, int epi);
/* y[B,N] = epi(x[B,K] . W[N,K]^T); rms_w != NULL fuses RMSNorm(x)*rms_w in front (rms_cast 0: >=4.32, 1: 4.31) */
int pcy_gemv(pcy_ctx*, const void* W, const void* x, int ldx, const void* bias, const void* resid, void* y, int ldy,
             const void* rms_w, float rms_eps, int rms_cast, int N, int K, int B, int epi);
/* One token through a Llama MLP, in place: x[d] += (silu(g) * u) . Wdown^T with [g; u] = (RMSNorm(x) * ln2) . Wgu^T
   (transformers LlamaDecoderLayer.forward: post_attention_layernorm -> LlamaMLP -> residual, as run per generated token by
   model_unified.py:883-915).  wgu: [2*ffn, d] in the 16-row gate/up interleave of pcy_gemv(EPI_SWIGLU); wdown: [d, ffn].
   Llama-3-8B geometry runs as ONE launch (the decode step's MLP chain kernel), anything else as the two pcy_gemv launches;
   same bits either way. */
int pcy_decode_mlp(pcy_ctx*, void* x, const void* ln2, const void* wgu, const void* wdown, int d, int ffn, float rms_eps, int rms_cast);
int pcy_rmsnorm(pcy_ctx*, const void* x, const void* w, void* y, int rows, int d, float eps, int cast);
int pcy_layernorm(pcy_ctx*, const void* x, const void* w, const void* b, void* y, int rows, int d, float eps);
/* out[r] = soft_map[r] >= 0 ? soft[soft_map[r]] : table[ids[r]]   (model_unified.py:1146-1167) */
int pcy_embed_splice(pcy_ctx*, const void* table, const int32_t* ids, const void* soft, const int32_t* soft_map,
                     void* out, int rows, int d);
/* In-place rotary on heads [0,nh) at column col0 of a token-major buffer [ntok, ld]; pos[tok] = position.
 * mode 0: (x*cos)+(rotate_half(x)*sin) with every op rounded to bf16 (HF Llama); mode 1: fp32, rounded once
 * (HF Esm).  prescale != 0 multiplies (and rounds) first: ESM's q * head_dim^-0.5. */
int pcy_rope(pcy_ctx*, void* buf, int ld, int col0, int nh, int dh, const int32_t* pos, const void* cos_t,
             const void* sin_t, int ntok, int mode, float prescale);
/* Exact-rounding eager attention over packed sequences (cu/vt_cu as in pcy_esm_encode): q,k,v token-major with
 * head h at column col0 + h*dh; O = bf16( bf16(softmax_fp32( bf16(bf16(QK^T)*scale) + mask )) . V ).
 * keep: per-token key mask (attention_mask) or NULL; causal != 0 adds the causal mask; GQA when Hkv < H. */
int pcy_attention(pcy_ctx*, const void* q, int ldq, int qcol0, const void* k, int ldk, int kcol0, const void* v, int ldv,
                  int vcol0, void* o, int ldo, const int32_t* cu, const int32_t* vt_cu, const uint8_t* keep, int nseq,
                  int max_len, int vt_total, int H, int Hkv, int dh, int causal, float scale);
/* Decode attention of ONE new token per row against the [B,Hkv,Tmax,dh] cache: ropes q and the new k at position
 * *pos (device scalar = cache length), appends K,V at slot *pos, attends slots [0,*pos] (keep: optional [B,Tmax]
 * key mask, NULL = the reference's unmasked decode, quirk Q1).  qkv [B,(H+2Hkv)*dh] un-roped projections. */
int pcy_attn_decode(pcy_ctx*, void* qkv, int ld, void* kcache, void* vcache, void* o, int ldo, const int32_t* pos,
                    const void* cos_t, const void* sin_t, const uint8_t* keep, int B, int H, int Hkv, int dh, int Tmax);
/* pooled[i] over the token ranges rng[2*r],rng[2*r+1] = (start,len), r in [seg[i], seg[i+1])  (esm.py:131-173) */
int pcy_pool(pcy_ctx*, const void* hidden, int d, const int32_t* seg, const int32_t* rng, int nprot, int mode, void* out);

/* Retrieval scoring (evaluate/framework/procyon.py:400-406, data/inference_utils.py:955-960):
 * sims[Q,N] = F.normalize(query) @ F.normalize(targets)^T in bf16 (norm rounded to bf16, then the division, then an
 * fp32-accumulating matmul rounded once).  D must be a multiple of 64. */
int pcy_retrieval_scores(pcy_ctx*, const void* query, int Q, const void* targets, int N, int D, void* sims_out);

/* ---- the one collective of the path: all-gather of the per-rank embedding blocks over RCCL / xGMI --------------------------
 * (reference pattern: training/trainIT.py:1594-1610, torch.distributed.all_gather of the shards of SequentialDistributedSampler).
 * One rank calls pcy_comm_unique_id (128 bytes) and distributes the id by any host channel (torch.distributed's store, MPI,
 * a file); every rank then calls pcy_comm_init.  pcy_allgather: recvbuf [nranks * bytes] = every rank's sendbuf [bytes] in
 * rank order, enqueued on the context's stream.  RCCL is dlopen'ed at first use (no link-time dependency). */
int pcy_comm_unique_id(void* id128);
int pcy_comm_init(pcy_ctx*, int nranks, int rank, const void* id128, void** comm_out);
void pcy_comm_destroy(void* comm);
int pcy_allgather(pcy_ctx*, void* comm, const void* sendbuf, void* recvbuf, size_t bytes);

/* Retrieval ranking (data/inference_utils.py:921-978 `get_proteins_from_embedding`): the same similarities, then per query the
 * k best targets in stable descending order (ties: the lower index first; torch.argsort leaves them open) -> idx_out [Q,k]
 * int32, score_out [Q,k] bf16.  1 <= k <= N; k == N is the full ranking (top_k=None). */
int pcy_retrieval_topk(pcy_ctx*, const void* query, int Q, const void* targets, int N, int D, int k, int32_t* idx_out, void* score_out);

/* The same two operations with fp32 similarities (the shim's `get_proteins_from_embedding` / `get_proteins_from_batched_embeddings`,
 * data/inference_utils.py:921-999: cached target matrices are fp32 and the batched form returns `.float()` scores):
 * sims[q][n] = (q / max(|q|, 1e-12)) . (t_n / max(|t_n|, 1e-12)), fp32 accumulation, fp32 result; query [Q,D] fp32, targets [N,D]
 * fp32 (targets_bf16 = 0) or bf16 (1); D % 4 == 0.  topk: idx_out [Q,k] int32, score_out [Q,k] fp32, stable descending order
 * (ties, incl. -0.0 vs +0.0: the lower index first). */
int pcy_retrieval_scores_f32(pcy_ctx*, const float* query, int Q, const void* targets, int targets_bf16, int N, int D, float* sims_out);
int pcy_retrieval_topk_f32(pcy_ctx*, const float* query, int Q, const void* targets, int targets_bf16, int N, int D, int k, int32_t* idx_out,
                           float* score_out);

/* QA read-out (reference: procyon/data/inference_utils.py:582-604, procyon/training/train_utils.py:1048-1070: `preds = logits.softmax(dim=-1)`
 * at the answer row, then the yes / no columns or the argmax).  logits [rows, V] bf16 (is_f32 = 0: fp32 statistics, ONE rounding of
 * exp(x - max) / sum, torch.softmax on a bf16 tensor) or fp32 (is_f32 = 1).  Optional outputs (NULL = not wanted): probs_out [rows, V] in
 * the logits' dtype; yes_no_out [rows, 2] fp32 = the stored probabilities of yes_id / no_id; argmax_out [rows] int32 = argmax of the stored
 * probabilities, lowest index on ties. */
int pcy_qa_probs(pcy_ctx*, const void* logits, int is_f32, int rows, int V, int yes_id, int no_id, void* probs_out, float* yes_no_out,
                 int32_t* argmax_out);

/* ---- fp8 weight path (BASELINE.json configs[4]: "fp8 MFMA weight path"; the reference has no fp8 counterpart) ----
 * Per-row symmetric OCP e4m3 quantisation of a bf16 matrix x[rows,K] (ldx elements between rows, K % 8 == 0):
 * scale_out[r] = the smallest power of two >= 2^-126 with amax|x[r,:]| / scale <= 448 (1 for an all-zero row),
 * q_out[r,k] = e4m3_rne(x / scale) (exact division), q_out [rows,K] bytes.  Used once per weight matrix (per output
 * channel) and per call on activations (per token). */
int pcy_quant_rows_fp8(pcy_ctx*, const void* x, int ldx, int rows, int K, void* q_out, float* scale_out);
/* C[M,N] = epi( ((A8[M,K] . W8[N,K]^T) * sa[m]) * sw[n] ) on the MX-scaled 16x16x128 fp8 MFMA (unit block scales), fp32
 * accumulation, then the same bf16 rounding points as pcy_gemm.  epi in {STORE, RESID, SWIGLU}; K % 128 == 0. */
int pcy_gemm_fp8(pcy_ctx*, const void* A8, const float* sa, const void* W8, const float* sw, const void* resid, int ldr,
                 void* C, int ldc, int M, int N, int K, int epi);

/* ---- create_mlp projector (model_utils.py:13-41) -------------------------------------------- */
typedef struct {
  int32_t n_layers;          /* 1 (bias-free Linear) or >= 2 (Linear+bias -> GELU ... -> Linear+bias) */
  int32_t dims[9];           /* dims[0]=in, dims[i+1]=out of layer i */
  const void* w[8];          /* [dims[i+1], dims[i]] */
  const void* b[8];          /* [dims[i+1]] or NULL */
} pcy_mlp_desc;
int pcy_mlp_forward(pcy_ctx*, const pcy_mlp_desc*, const void* x, int M, void* out);

/* ---- ESM2 encoder (esm.py:504-538 -> fair-esm ESM2 / HF EsmModel) ---------------------------- */
typedef struct {
  const void *wqkv, *bqkv;   /* [3d,d],[3d]: query,key,value rows stacked */
  const void *wo, *bo;       /* [d,d],[d] */
  const void *ln1_w, *ln1_b; /* attention LayerNorm */
  const void *w1, *b1;       /* [F,d],[F] */
  const void *w2, *b2;       /* [d,F],[d] */
  const void *ln2_w, *ln2_b; /* FFN LayerNorm */
} pcy_esm_layer;
typedef struct {
  int32_t d, n_layers, n_heads, ffn, vocab;
  float ln_eps;
  int32_t rope_mode;         /* 1: fp32, rounded once (HF Esm) ; 0: three bf16 roundings (fair-esm) */
  const void* embed;         /* [vocab,d] */
  const void *final_ln_w, *final_ln_b;
  const void *rope_cos, *rope_sin; /* [max_len, d/n_heads] bf16 */
  const pcy_esm_layer* layers;     /* host array [n_layers] */
} pcy_esm_desc;
/* Packed varlen batch: tokens[ntok], sequence q = tokens[cu[q]..cu[q+1]); pos[tok] = index inside its
 * sequence; vt_cu[q] = sum over earlier sequences of roundup(len,32), vt_total = vt_cu[nseq].
 * mask_pads=1: fair-esm semantics (callers pack WITHOUT pad tokens); mask_pads=0: the HF-"official"
 * call without attention_mask (esm.py:533): pack the padded rows, pads take part as ordinary tokens.
 * hidden_out [ntok,d] = representations[repr_layer] (post final LayerNorm). */
int pcy_esm_encode(pcy_ctx*, const pcy_esm_desc*, const int32_t* tokens, const int32_t* pos, const int32_t* cu,
                   const int32_t* vt_cu, int ntok, int nseq, int max_len, int vt_total, int mask_pads, void* hidden_out);

/* ---- Llama decoder (pmc_llama.py:546-596 -> HF LlamaForCausalLM) ----------------------------- */
typedef struct {
  const void* wqkv;          /* [(H+2Hkv)*dh, d]: q,k,v rows stacked */
  const void* wo;            /* [d, H*dh] */
  const void* wgu;           /* [2F, d]: gate/up interleaved in blocks of 16 rows */
  const void* wdown;         /* [d, F] */
  const void *ln1, *ln2;     /* [d] */
} pcy_llama_layer;
/* optional e4m3 copies of the four projection matrices (same row order / interleave as the bf16 ones) and their per-row
 * scales (pcy_quant_rows_fp8).  When pcy_llama_desc.layers_fp8 != NULL, pcy_llama_prefill quantises the input of every
 * projection per token and runs it through pcy_gemm_fp8; norms, rope, attention, residuals, lm_head and the whole
 * decode path stay bf16. */
typedef struct {
  const void *wqkv, *wo, *wgu, *wdown;
  const float *sqkv, *so, *sgu, *sdown;
} pcy_llama_layer_fp8;
typedef struct {
  int32_t vocab, d, n_layers, n_heads, n_kv_heads, head_dim, ffn, max_pos;
  float rms_eps;
  int32_t rms_cast;          /* 0: w*bf16(x_hat) ; 1: bf16(w*x_hat) (transformers 4.31, SURVEY App.B Q11) */
  const void* embed;         /* [vocab,d] */
  const void* final_norm;    /* [d] */
  const void* lm_head;       /* [vocab,d] */
  const void *rope_cos, *rope_sin; /* [max_pos, head_dim] bf16 (table precision is the caller's choice, Q9/Q10) */
  const pcy_llama_layer* layers;   /* host array [n_layers] */
  const pcy_llama_layer_fp8* layers_fp8; /* host array [n_layers] or NULL (bf16 weights everywhere) */
} pcy_llama_desc;
typedef struct {
  void* k;                   /* [L,B,Hkv,Tmax,dh] bf16: layer l rows = past_key_values[l][0] */
  void* v;                   /* same for values */
  int32_t B, Tmax;
} pcy_kv_cache;
/* Prefill: embeds [B,T,d]; keep [B*T] uint8 attention_mask or NULL; pos[B*T] rotary positions
 * (reference: arange(T) per row, Q2); cu/vt_cu [B+1] = b*T / b*roundup(T,32).
 * logit_rows[n_logit_rows] = token rows (b*T+t) whose logits are wanted -> logits_out [n_logit_rows, vocab];
 * hidden_out (optional) [B*T,d] = hidden_states[-1] (final-normed). K/V of slots [0,T) are written.
 * sum_rows[n_sum_rows] (optional) -> hidden_sum_out [n_sum_rows, d] = sum over ALL L+1 hidden states of those token rows
 * (ret_token_access='all', model_unified.py:560-563: fp32 accumulation, rounded once). */
int pcy_llama_prefill(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const void* embeds, const uint8_t* keep,
                      const int32_t* pos, const int32_t* cu, const int32_t* vt_cu, int B, int T,
                      const int32_t* logit_rows, int n_logit_rows, void* logits_out, void* hidden_out,
                      const int32_t* sum_rows, int n_sum_rows, void* hidden_sum_out);
/* The same prefill that additionally materialises what `LlamaPostTokenization.forward` hands back besides the logits
 * (pmc_llama.py:575-596, `output_hidden_states=True`): hidden_all_out [L+1][B*T][d] bf16 = the embeddings, the output of layers
 * 0..L-2 and the final-normed output of layer L-1.  logit_rows may name every row (the reference's [B,T,V] logits): more than 64
 * rows take the lm_head as an MFMA GEMM -- on THIS entry point only; pcy_llama_prefill always uses the fused-norm GEMV, so a row's
 * logits there do not depend on the number of rows requested. */
int pcy_llama_prefill_all(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const void* embeds, const uint8_t* keep,
                          const int32_t* pos, const int32_t* cu, const int32_t* vt_cu, int B, int T, const int32_t* logit_rows,
                          int n_logit_rows, void* logits_out, void* hidden_all_out);
typedef struct {
  int32_t* pos;              /* device scalar: cache length == rotary position of the next token (Q2) */
  int32_t* step;             /* device scalar: index of the next generated token */
  int32_t* next_tok;         /* [B] token fed to the next decode step */
  int32_t* tokens_out;       /* [B, max_steps] */
  float* logprob;            /* [B] running sum of log_softmax(logits)[token] */
  void* logits;              /* [B, vocab] logits of the latest step */
  void* logits_all;          /* optional record [max_steps, B, logits_all_ld] of every step's logits; may be PINNED HOST memory
                              * (hipHostMalloc): the rows then cross PCIe during the following decode steps */
  const uint8_t* keep;       /* optional [B, Tmax] decode key mask ("clean" mode); NULL = reference quirk Q1 */
  int32_t max_steps;
  int32_t logits_all_ld;     /* row stride of logits_all in elements; 0 = vocab.  A multiple of 8 enables 16-byte stores */
} pcy_gen_state;
/* one decode step: next_tok -> logits (and K/V appended at slot *pos); does not pick or advance */
int pcy_llama_decode(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const pcy_gen_state*, int B);
/* Measurement aid (tools/bench_decode.py with PCY_MC_TRACE=1 in the environment): copies the in-kernel time stamps of the last
 * fused decode launches ([2][128 layers][256 workgroups][16] ticks of 10 ns) to `out` (n words); returns n, or 0 when no stamps
 * were taken. */
int pcy_debug_mc_trace(unsigned long long* out, int n);
/* Measurement aid (bench.py roofline leg): `reps` passes over the decoder layers of a decode step -- no token embedding (the
 * residual stream is whatever the workspace holds), no lm_head, no pick; K/V of slot *pos are rewritten each pass.  With HIP
 * events around it: time per layer launch = elapsed / (reps * n_layers). */
int pcy_llama_decode_layers(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const pcy_gen_state*, int B, int reps);
/* the same step as ONE replayed hipGraph (captured on first use per model / cache / state / batch): for host-driven loops that
 * do their own selection between steps (sampling, diverse beam search).  next_tok and *pos are read from device memory. */
int pcy_llama_decode_graph(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const pcy_gen_state*, int B);
/* argmax of state->logits (lowest index on ties) -> next_tok / tokens_out[step], logprob, ++pos? no: ++step only
 * when advance_pos == 0 (used on the prefill logits), ++pos and ++step otherwise */
int pcy_greedy_pick(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const pcy_gen_state*, int B, int advance_pos);
/* n_steps x (decode + pick) with no host synchronisation; use_graph != 0 replays a captured hipGraph */
int pcy_llama_greedy(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const pcy_gen_state*, int B, int n_steps,
                     int use_graph);
/* Sampling / nucleus selection (`_generate_sampling` without greedy, model_unified.py:896-906) on state->logits:
 *   probs = bf16 softmax(logits / temperature); nucleus_prob in (0,1): probs = softmax(logits) * mask of the tokens whose ascending
 *   cumulative probability has reached 1 - nucleus_prob (`_get_nucleus_mask`, not renormalised); nucleus_prob <= 0: no mask;
 *   token = first index whose cumulative masked probability exceeds uniforms[step * B + b] * total (inverse CDF with the CALLER's
 *   uniform variates in [0,1): torch's multinomial stream is not reproducible, the probability vector and the draw for a given u
 *   are); logprob += bf16 log_softmax(unscaled logits)[token]; then next_tok / tokens_out / ++step (/ ++pos) like pcy_greedy_pick.
 * probs_out: optional [B, vocab] bf16 record of the pre-sampling probability vector.  uniforms: device fp32 [max_steps * B]. */
int pcy_sample_pick(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const pcy_gen_state*, int B, int advance_pos,
                    float temperature, float nucleus_prob, const float* uniforms, void* probs_out);
/* n_steps x (decode + sample) with no host synchronisation */
int pcy_llama_sample(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const pcy_gen_state*, int B, int n_steps,
                     float temperature, float nucleus_prob, const float* uniforms);
/* Diverse beam search bookkeeping of ONE step on the device (`_generate_beam_search`, model_unified.py:782-833): per prompt
 * and beam group, top-`group_size` of  log_softmax(logits) (model dtype) + running score - diversity_penalty * (count of the
 * token among the picks of the earlier groups at this step);  at step 0 only the first beam of a group is expanded.  Updates
 * the token histories, the running scores, `src` (parent slot of every slot: feed it to pcy_kv_reorder), `next_tok`, ++*step,
 * ++*pos (from step 1 on) and raises *done once every row holds `eos_id`; a launch with *done set changes nothing.
 * logits [B*beam, vocab] bf16.  All pointers are device memory; state layout in the struct below (BB = B * beam). */
typedef struct {
  int32_t* out; int32_t max_len;   /* [2][BB][max_len] token histories; buffer (*step & 1) is current; zero-initialised */
  float* cur; float* cur_new;      /* [BB] running scores (zero-initialised) + scratch */
  int32_t* next_tok;               /* [BB] */
  int32_t* src;                    /* [BB] */
  int32_t* anc;                    /* optional [max_len][BB]: src of every step (to permute a per-slot logits record afterwards) */
  uint8_t* has_eos;                /* [2][BB] zero-initialised */
  int32_t* blk_eos; int32_t* ticket;   /* [B], [1] zero-initialised */
  int32_t* pos; int32_t* step; int32_t* done;   /* device scalars */
  int32_t eos_id;
} pcy_beam_state;
int pcy_beam_step(pcy_ctx*, const void* logits, int vocab, int B, int beam, int group_size, float diversity_penalty,
                  const pcy_beam_state* state);
/* KV rows gather for beam search: cache[:, dst] = cache[:, src[dst]] over slots [0,t) (model_unified.py:830-832) */
int pcy_kv_reorder(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const int32_t* src_rows, int B, int t);
/* ... over slots [t0, t) only: for callers that know the rows hold the same contents below t0 -- the beams of a prompt share its prefix (the
 * reference moves the whole history of every row in every group of every step, model_unified.py:830-832; moving equal bytes changes nothing).
 * src_rows[b] may be any row of the cache (rows >= B are read straight from memory). */
int pcy_kv_reorder_range(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const int32_t* src_rows, int B, int t0, int t);
/* n_steps iterations (steps >= 1) of the loop body of `_generate_beam_search` (model_unified.py:751-842): decode step for the B * beam
 * rows -> logits_rec[step][row][vocab] = the step's logits (NULL: no record) -> pcy_beam_step -> pcy_kv_reorder over the slots the state's
 * position counter names, from slot kv_t0 on (0: all of them; the prompt length when the beams of a prompt share its prefix rows: see
 * pcy_kv_reorder_range); each iteration is ONE replayed launch chain.  gen_state->pos / next_tok must be the beam state's arrays.  Same
 * results as pcy_llama_decode_graph + a copy + pcy_beam_step + pcy_kv_reorder per step. */
int pcy_llama_beam_steps(pcy_ctx*, const pcy_llama_desc*, const pcy_kv_cache*, const pcy_gen_state*, int B, int beam, int group_size,
                         float diversity_penalty, const pcy_beam_state* state, void* logits_rec, int n_steps, int kv_t0);

/* ---- fp32 operator family: the arithmetic of the reference's callers that never call `.bfloat16()` ----
 * /root/reference/examples/paper_analyses/protpep_qa_scores.py:55-58 (`model.eval().to(device)`: fp32 weights, every torch op in
 * fp32 -- the loop that defines BASELINE configs[4]), /root/reference/scripts/qa_filter_captions.py:17-18 and
 * /root/reference/scripts/caption_bulk.py:72-73.  Every pointer is a DEVICE pointer to fp32 (ids / pos / cu: int32, keep: uint8);
 * calls enqueue on the context's stream.  The host side (procyon_amd/engine_f32.py) strings these together op for op like the fp32
 * evaluation of the reference's modules: ESM2 encoder (esm.py:504-538) -> ProteinPooler (:131-173) -> create_mlp projectors
 * (model_utils.py:13-41) -> splice (model_unified.py:1135-1175) -> Llama prefill (pmc_llama.py:546-596).  Cached decode stays bf16. */
/* C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) (+ resid[M,N]); W is nn.Linear's [out, in]; act 0 = none, 1 = gelu
 * x * 0.5 * (1 + erf(x / sqrt 2)) (nn.GELU and the ESM gelu coincide in fp32).  bias / resid may be NULL; resid may alias C.
 * Matrix cores on exact fp32 products (v_mfma_f32_16x16x4_f32) when K % 16 == 0 and rows are 16-byte aligned, a plain kernel otherwise. */
int pcy_f32_linear(pcy_ctx*, const float* A, int lda, const float* W, const float* bias, const float* resid, int ldr, float* C, int ldc,
                   int M, int N, int K, int act);
/* torch.nn.functional.layer_norm / HF LlamaRMSNorm (w * (x * rsqrt(mean(x^2) + eps))) over rows of d */
int pcy_f32_layernorm(pcy_ctx*, const float* x, const float* w, const float* b, float* y, int rows, int d, float eps);
int pcy_f32_rmsnorm(pcy_ctx*, const float* x, const float* w, float* y, int rows, int d, float eps);
/* rotary embedding in place on heads [col0, col0 + nh*dh) of every row: x <- x' cos[pos] + rotate_half(x') sin[pos], x' = x * prescale
 * (prescale 0 = none; ESM scales q by dh^-1/2 BEFORE the rotation, esm2 multihead attention); cos / sin tables fp32 [n_pos, dh] */
int pcy_f32_rope(pcy_ctx*, float* buf, int ld, int col0, int nh, int dh, const int32_t* pos, const float* cos_t, const float* sin_t, int ntok,
                 float prescale);
/* out[r] = soft_map && soft_map[r] >= 0 ? soft[soft_map[r]] : table[ids[r]]: token embedding + soft-token splice as one gather */
int pcy_f32_embed(pcy_ctx*, const float* table, const int32_t* ids, const float* soft, const int32_t* soft_map, float* out, int rows, int d);
/* HF EsmEmbeddings with token_dropout over packed sequences (cu [nseq+1]): <mask> rows zero, x * 0.88 / (1 - mask ratio), pads zero */
int pcy_f32_esm_embed(pcy_ctx*, const float* table, const int32_t* tokens, const int32_t* cu, int nseq, int max_len, float* out, int d, int mask_pads);
int pcy_f32_silu_mul(pcy_ctx*, const float* gate, const float* up, float* out, size_t n);      /* out = silu(gate) * up */
int pcy_f32_acc_rows(pcy_ctx*, const float* src, int ld, const int32_t* rows, float* dst, int nrows, int d);   /* dst[r] += src[rows[r]] */
/* ProteinPooler over token ranges, arguments as pcy_pool (mode 0 mean, 1 mean of x[1:-1], 2 max) */
int pcy_f32_pool(pcy_ctx*, const float* hidden, int d, const int32_t* seg, const int32_t* rng, int nprot, int mode, float* out);
/* softmax((q . k) * scale + mask) . v over packed sequences (cu), grouped heads (Hkv | H), exact fp32 softmax; causal = 1: keys <= query;
 * keep (may be NULL): one byte per packed token, 0 = the key is masked out.  q / k / v are column windows of row-major buffers. */
int pcy_f32_attention(pcy_ctx*, const float* q, int ldq, int qcol0, const float* k, int ldk, int kcol0, const float* v, int ldv, int vcol0,
                      float* o, int ldo, const int32_t* cu, const uint8_t* keep, int nseq, int max_len, int H, int Hkv, int dh, int causal,
                      float scale);

/* fp32 decode attention (callers that never call .bfloat16() and generate: /root/reference/scripts/caption_bulk.py:70-73, 123-132): the new
 * token's q [B, H*dh] (roped) against slots [0, nkeys) of token-major caches [B][Tmax][Hkv*dh] (row stride ldkv floats); every slot is
 * attended (/root/reference/procyon/model/model_unified.py:769, :887 pass no mask after the prefill). */
int pcy_f32_attn_decode(pcy_ctx*, const float* q, int ldq, const float* kcache, const float* vcache, int ldkv, int Tmax, float* o, int ldo, int B,
                        int H, int Hkv, int dh, int nkeys, float scale);

#ifdef __cplusplus
}
#endif
#endif /* PCY_H */
