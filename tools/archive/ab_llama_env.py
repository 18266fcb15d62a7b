"""Interleaved A/B of an environment switch that the engine reads per call, on the Llama-3-8B prefill: one 512-token prompt, 64 x 450
tokens in bf16 and the same through the fp8 weight path:
    VAR=PCY_GEMM_PERM VALS=0,1 python tools/ab_llama_env.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, LlamaConfig, LlamaEngine
kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
var, vals = os.environ["VAR"], os.environ.get("VALS", "0,1").split(",")
emb1 = (torch.randn(1, 512, 4096, device="cuda") * 0.02).bfloat16()
cache1 = eng.new_cache(1, 520)
embB = (torch.randn(64, 450, 4096, device="cuda") * 0.02).bfloat16()
cacheB = eng.new_cache(64, 450)
def fl(B, T): return B * (2 * 6979584000 * T + 262144 * T * T + 2 * kw["vocab"] * 4096)
def leg(name, emb, cache, n):
    for v in vals:
        os.environ[var] = v
        eng.prefill(emb, None, cache, "last")
    for rnd in range(3):
        for v in vals:
            os.environ[var] = v
            ctx.timer_start()
            for _ in range(n): eng.prefill(emb, None, cache, "last")
            ms = ctx.timer_stop() / n
            print(f"{name} round {rnd} {var}={v}: {ms:8.3f} ms  {fl(emb.shape[0], emb.shape[1]) / ms / 1e9:7.0f} TFLOP/s", flush=True)
leg("prefill 1x512 bf16", emb1, cache1, 10)
leg("prefill 64x450 bf16", embB, cacheB, 2)
eng.quantize_fp8()
leg("prefill 64x450 fp8", embB, cacheB, 2)
