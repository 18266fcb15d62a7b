"""Decode-only tokens/s for Llama-3-8B geometry, used for A/B of decode-path changes (PCY_DECODE_PIPE=0/1, GRAPH=0/1):
GPU time per token (HIP events) and host time to enqueue a step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
T, N = int(os.environ.get("T", 512)), 256
graph = os.environ.get("GRAPH", "1") != "0"
emb = (torch.randn(1, T, 4096, device="cuda") * 0.02).bfloat16()
cache = eng.new_cache(1, T + N)
st = GenState(1, kw["vocab"], N, "cuda")
logits, _ = eng.prefill(emb, None, cache, "last")
st.logits.copy_(logits); st.pos.fill_(T)
eng.pick(cache, st, 1, advance_pos=False)
eng.greedy_steps(cache, st, 1, 8, use_graph=graph)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    ctx.timer_start(); eng.greedy_steps(cache, st, 1, 60, use_graph=graph); host = (time.perf_counter() - t0) / 60
    ms = ctx.timer_stop() / 60
    print(f"decode {ms:.3f} ms/token  {1e3/ms:.1f} tok/s  {15.09/ms:.2f} TB/s   host enqueue {host*1e3:.3f} ms/step", flush=True)
ctx.sync()
