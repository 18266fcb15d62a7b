"""ctypes binding of libpcy.so (include/pcy.h).  The product path has NO fallback: if the HIP
library is missing or fails to load, importing the engine raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCY_LIB") or os.path.join(_HERE, "libpcy.so")   # PCY_LIB: A/B of experimental builds

EPI_STORE, EPI_RESID, EPI_GELU_ERF, EPI_GELU_ESM, EPI_SWIGLU = range(5)
POOL_MEAN, POOL_MEAN_CORRECTED, POOL_MAX = range(3)
ABI_VERSION = 10
# pcy_debug_dispatch_count kinds
DISPATCH_GEMM_128, DISPATCH_GEMM_64, DISPATCH_GEMM_BIG, DISPATCH_GEMM_BIG_PERSIST, DISPATCH_GEMM_SPLITK, DISPATCH_GEMM_FP8, DISPATCH_ATTN_FAST, _DISPATCH_UNUSED_7, DISPATCH_GEMM_MID, DISPATCH_ESM_GRAPH = range(10)

vp = C.c_void_p
i32 = C.c_int32
ci = C.c_int


class MlpDesc(C.Structure):
    _fields_ = [("n_layers", i32), ("dims", i32 * 9), ("w", vp * 8), ("b", vp * 8)]


class EsmLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_w", "ln1_b", "w1", "b1", "w2", "b2", "ln2_w", "ln2_b")]


class EsmDesc(C.Structure):
    _fields_ = [("d", i32), ("n_layers", i32), ("n_heads", i32), ("ffn", i32), ("vocab", i32), ("ln_eps", C.c_float),
                ("rope_mode", i32), ("embed", vp), ("final_ln_w", vp), ("final_ln_b", vp), ("rope_cos", vp),
                ("rope_sin", vp), ("layers", C.POINTER(EsmLayer))]


class LlamaLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("wqkv", "wo", "wgu", "wdown", "ln1", "ln2")]


class LlamaLayerFp8(C.Structure):
    _fields_ = [(n, vp) for n in ("wqkv", "wo", "wgu", "wdown", "sqkv", "so", "sgu", "sdown")]


class LlamaDesc(C.Structure):
    _fields_ = [("vocab", i32), ("d", i32), ("n_layers", i32), ("n_heads", i32), ("n_kv_heads", i32), ("head_dim", i32),
                ("ffn", i32), ("max_pos", i32), ("rms_eps", C.c_float), ("rms_cast", i32), ("embed", vp),
                ("final_norm", vp), ("lm_head", vp), ("rope_cos", vp), ("rope_sin", vp), ("layers", C.POINTER(LlamaLayer)),
                ("layers_fp8", C.POINTER(LlamaLayerFp8))]


class KvCache(C.Structure):
    _fields_ = [("k", vp), ("v", vp), ("B", i32), ("Tmax", i32)]


class GenState(C.Structure):
    _fields_ = [("pos", vp), ("step", vp), ("next_tok", vp), ("tokens_out", vp), ("logprob", vp), ("logits", vp),
                ("logits_all", vp), ("keep", vp), ("max_steps", i32), ("logits_all_ld", i32)]


class BeamState(C.Structure):
    _fields_ = [("out", vp), ("max_len", i32), ("cur", vp), ("cur_new", vp), ("next_tok", vp), ("src", vp), ("anc", vp),
                ("has_eos", vp), ("blk_eos", vp), ("ticket", vp), ("pos", vp), ("step", vp), ("done", vp), ("eos_id", i32)]


# name -> (restype, argtypes); every symbol include/pcy.h declares
SIGNATURES = {
    "pcy_abi_version": (ci, []),
    "pcy_debug_dispatch_count": (C.c_ulonglong, [ci]),
    "pcy_last_error": (C.c_char_p, []),
    "pcy_ctx_create": (ci, [ci, vp, C.POINTER(vp)]),
    "pcy_ctx_destroy": (None, [vp]),
    "pcy_ctx_sync": (ci, [vp]),
    "pcy_timer_start": (ci, [vp]),
    "pcy_timer_stop": (ci, [vp, C.POINTER(C.c_float)]),
    "pcy_gemm": (ci, [vp, vp, ci, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci]),
    "pcy_gemv": (ci, [vp, vp, vp, ci, vp, vp, vp, ci, vp, C.c_float, ci, ci, ci, ci, ci]),
    "pcy_decode_mlp": (ci, [vp, vp, vp, vp, vp, ci, ci, C.c_float, ci]),
    "pcy_rmsnorm": (ci, [vp, vp, vp, vp, ci, ci, C.c_float, ci]),
    "pcy_layernorm": (ci, [vp, vp, vp, vp, vp, ci, ci, C.c_float]),
    "pcy_embed_splice": (ci, [vp, vp, vp, vp, vp, vp, ci, ci]),
    "pcy_rope": (ci, [vp, vp, ci, ci, ci, ci, vp, vp, vp, ci, ci, C.c_float]),
    "pcy_attention": (ci, [vp, vp, ci, ci, vp, ci, ci, vp, ci, ci, vp, ci, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, C.c_float]),
    "pcy_attn_decode": (ci, [vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci]),
    "pcy_pool": (ci, [vp, vp, ci, vp, vp, ci, ci, vp]),
    "pcy_retrieval_scores": (ci, [vp, vp, ci, vp, ci, ci, vp]),
    "pcy_comm_unique_id": (ci, [vp]),
    "pcy_comm_init": (ci, [vp, ci, ci, vp, C.POINTER(vp)]),
    "pcy_comm_destroy": (None, [vp]),
    "pcy_allgather": (ci, [vp, vp, vp, vp, C.c_size_t]),
    "pcy_retrieval_topk": (ci, [vp, vp, ci, vp, ci, ci, ci, vp, vp]),
    "pcy_retrieval_scores_f32": (ci, [vp, vp, ci, vp, ci, ci, ci, vp]),
    "pcy_retrieval_topk_f32": (ci, [vp, vp, ci, vp, ci, ci, ci, ci, vp, vp]),
    "pcy_qa_probs": (ci, [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]),
    # fp32 operator family (include/pcy.h: callers that never call .bfloat16())
    "pcy_f32_linear": (ci, [vp, vp, ci, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci]),
    "pcy_f32_layernorm": (ci, [vp, vp, vp, vp, vp, ci, ci, C.c_float]),
    "pcy_f32_rmsnorm": (ci, [vp, vp, vp, vp, ci, ci, C.c_float]),
    "pcy_f32_rope": (ci, [vp, vp, ci, ci, ci, ci, vp, vp, vp, ci, C.c_float]),
    "pcy_f32_embed": (ci, [vp, vp, vp, vp, vp, vp, ci, ci]),
    "pcy_f32_esm_embed": (ci, [vp, vp, vp, vp, ci, ci, vp, ci, ci]),
    "pcy_f32_silu_mul": (ci, [vp, vp, vp, vp, C.c_size_t]),
    "pcy_f32_acc_rows": (ci, [vp, vp, ci, vp, vp, ci, ci]),
    "pcy_f32_pool": (ci, [vp, vp, ci, vp, vp, ci, ci, vp]),
    "pcy_f32_attention": (ci, [vp, vp, ci, ci, vp, ci, ci, vp, ci, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, C.c_float]),
    "pcy_f32_attn_decode": (ci, [vp, vp, ci, vp, vp, ci, ci, vp, ci, ci, ci, ci, ci, ci, C.c_float]),
    "pcy_quant_rows_fp8": (ci, [vp, vp, ci, ci, ci, vp, vp]),
    "pcy_gemm_fp8": (ci, [vp, vp, vp, vp, vp, vp, ci, vp, ci, ci, ci, ci, ci]),
    "pcy_mlp_forward": (ci, [vp, C.POINTER(MlpDesc), vp, ci, vp]),
    "pcy_esm_encode": (ci, [vp, C.POINTER(EsmDesc), vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "pcy_llama_prefill": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), vp, vp, vp, vp, vp, ci, ci, vp, ci, vp, vp, vp, ci, vp]),
    "pcy_llama_prefill_all": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), vp, vp, vp, vp, vp, ci, ci, vp, ci, vp, vp]),
    "pcy_llama_decode": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), C.POINTER(GenState), ci]),
    "pcy_debug_mc_trace": (ci, [vp, ci]),
    "pcy_llama_decode_layers": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), C.POINTER(GenState), ci, ci]),
    "pcy_llama_decode_graph": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), C.POINTER(GenState), ci]),
    "pcy_greedy_pick": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), C.POINTER(GenState), ci, ci]),
    "pcy_llama_greedy": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), C.POINTER(GenState), ci, ci, ci]),
    "pcy_sample_pick": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), C.POINTER(GenState), ci, ci, C.c_float, C.c_float, vp, vp]),
    "pcy_llama_sample": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), C.POINTER(GenState), ci, ci, C.c_float, C.c_float, vp]),
    "pcy_beam_step": (ci, [vp, vp, ci, ci, ci, ci, C.c_float, C.POINTER(BeamState)]),
    "pcy_kv_reorder": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), vp, ci, ci]),
    "pcy_kv_reorder_range": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), vp, ci, ci, ci]),
    "pcy_llama_beam_steps": (ci, [vp, C.POINTER(LlamaDesc), C.POINTER(KvCache), C.POINTER(GenState), ci, ci, ci, C.c_float, C.POINTER(BeamState), vp, ci, ci]),
}

_lib = None


class PcyError(RuntimeError):
    pass


def load():
    """dlopen libpcy.so and type every entry point.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PcyError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(the ProCyon engine has no CPU/torch fallback)")
    # torch first: it ships its own libamdhip64; loading libpcy.so before it binds the engine to /opt/rocm's copy and the
    # process ends up with two HIP runtimes (the second one then reports "no ROCm-capable device")
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.pcy_abi_version() != ABI_VERSION:
        raise PcyError("libpcy.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pcy_last_error()
        raise PcyError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
