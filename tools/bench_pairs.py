"""BASELINE configs[4] shape: pair scoring = Llama-3-8B prefill of 256 QA prompts (T = 450, logits at one row each), bf16
weights vs the fp8 (e4m3, MX MFMA) weight path.  Prints pairs/s and the algorithmic TFLOP/s of the prefill."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, LlamaConfig, LlamaEngine
kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
pairs, T, chunk = int(os.environ.get("PAIRS", 256)), 450, int(os.environ.get("CHUNK", 64))
emb = (torch.randn(chunk, T, 4096, device="cuda") * 0.02).bfloat16()
cache = eng.new_cache(chunk, T)
flop = pairs * (2 * 6979584000 * T + 262144 * T * T + 2 * kw["vocab"] * 4096)
for mode in ("bf16", "fp8"):
    if mode == "fp8":
        eng.quantize_fp8()
    eng.prefill(emb, None, cache, "last")
    ctx.timer_start()
    for _ in range(pairs // chunk):
        logits, _ = eng.prefill(emb, None, cache, "last")
    ms = ctx.timer_stop()
    print(f"{mode}: {pairs} pairs in {ms:.0f} ms = {pairs / ms * 1e3:.1f} pairs/s, {flop / ms / 1e9:.0f} TFLOP/s algorithmic", flush=True)
