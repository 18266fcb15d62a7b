"""Decode-only ms per token for the three selection kernels (greedy graph, sampling, nucleus) at Llama-3-8B geometry, batch 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
T, N = 512, 200
emb = (torch.randn(1, T, 4096, device="cuda") * 0.02).bfloat16()
u = torch.rand(N + 8, device="cuda")
for name, nuc in (("greedy", None), ("sampling", None), ("nucleus", 0.9)):
    cache = eng.new_cache(1, T + N + 8)
    st = GenState(1, kw["vocab"], N + 8, "cuda")
    logits, _ = eng.prefill(emb, None, cache, "last")
    st.logits.copy_(logits); st.pos.fill_(T)
    if name == "greedy":
        eng.pick(cache, st, 1, advance_pos=False)
        eng.greedy_steps(cache, st, 1, 8)
        ctx.timer_start(); eng.greedy_steps(cache, st, 1, 64); ms = ctx.timer_stop() / 64
    else:
        eng.sample_pick(cache, st, 1, False, u, 1.0, nuc)
        eng.sample_steps(cache, st, 1, 8, u, 1.0, nuc)
        ctx.timer_start(); eng.sample_steps(cache, st, 1, 64, u, 1.0, nuc); ms = ctx.timer_stop() / 64
    print(f"{name:9s}: {ms:.3f} ms/token", flush=True)
