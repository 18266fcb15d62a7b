// Run-time switches of libpcy.so, all read per call (tests flip them inside one process; a launch costs ~3.5 us, a getenv ~50 ns, and the
// replayed graphs of the decode step and of the short-input encoder read nothing at all):
//
//   PCY_DISABLE=a,b,...   switches OFF the named fused / alternative path; each one has a slower twin with the same bits, and a test that
//                         compares the two:  attn_o  mlp_chain  decode_layer  decode_step  attn_qkv_finish  finish_norm  fp8_fused_norm
//                         prefill_post_qkv  gemv_lds  gemv_mfma4  fa_vrow  gelu_fast  esm_graph  lds_prefetch  decode_nb_step  beam_graph  kv_permute
//                         decode_mb_step  beam_prefill_once  beam_kv_suffix   (decode_step / decode_layer also select the twins of the
//                         ProCyon-Split step, pcy_decode_mha.hip)
//                         (decode_nb: batches of 2..8 rows back on the round-4 launches -- another arithmetic, compared to bf16 noise)
//   PCY_NB_MAX=<rows>     largest batch on the small-batch decode step (default 6; 7, 8: tests, tools)
//   PCY_MB_MAX=<rows>     largest batch on the opt-in mid-batch decode step, 9..32 (default 0: off);  PCY_MB_ABL=<mask>  its timing ablations (tools)
//   PCY_ESM_ATTN=exact    the two-pass attention with the reference's bf16 rounding points (default: the single-pass kernel)
//   PCY_GEMM_PERM=<mask>  256 x 256 epilogues on the permuted W row order (1 STORE, 2 RESID, 4 ESM GELU, 8 SwiGLU, 16 fp8; default 7)
//   PCY_GEMM_MID=<cfg>    force a gemm_kernel_mid configuration (-1: the pre-round-4 kernels; "NxK=cfg,...": per shape)
//   PCY_AO_XMIN=<keys>    cache length from which the decode attention splits its keys across the slice workgroups (default 768)
//   PCY_NB_UB=1|2         k-iterations per weight batch of the small-batch decode step's MLP streams (measurement; default 2)
//   PCY_DEBUG_POISON_WS=1 fill the workspace with NaN patterns before every call;  PCY_MC_TRACE=1  in-kernel time stamps (tools)
#pragma once
#include <stdlib.h>
#include <string.h>

inline bool pcy_off(const char* name) {
  const char* e = getenv("PCY_DISABLE");
  if (!e) return false;
  const size_t n = strlen(name);
  for (const char* p = e; (p = strstr(p, name)) != nullptr; p += n)
    if ((p == e || p[-1] == ',') && (p[n] == 0 || p[n] == ',')) return true;
  return false;
}
