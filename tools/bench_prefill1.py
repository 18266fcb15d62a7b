"""Single-prompt Llama-3-8B prefill (B = 1, T = 512): time per call, for the kernel table of the M = 512 path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, LlamaConfig, LlamaEngine
kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
T = int(os.environ.get("T", 512))
emb = (torch.randn(1, T, 4096, device="cuda") * 0.02).bfloat16()
cache = eng.new_cache(1, T + 8)
for _ in range(2): eng.prefill(emb, None, cache, "last")
ctx.timer_start(); n = 10
for _ in range(n): eng.prefill(emb, None, cache, "last")
ms = ctx.timer_stop() / n
fl = 2 * 6979584000 * T + 262144 * T * T + 2 * kw["vocab"] * 4096
print(f"prefill T={T}: {ms:.2f} ms  {fl / ms / 1e9:.0f} TFLOP/s algorithmic")
