import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
T, N, B = 512, 40, int(os.environ.get("B", 20))
emb = (torch.randn(B, T, 4096, device="cuda") * 0.02).bfloat16()
cache = eng.new_cache(B, T + N)
st = GenState(B, kw["vocab"], N, "cuda")
logits, _ = eng.prefill(emb, None, cache, "last")
st.logits.copy_(logits); st.pos.fill_(T)
eng.pick(cache, st, B, advance_pos=False)
eng.greedy_steps(cache, st, B, 4)
ctx.timer_start(); eng.greedy_steps(cache, st, B, 30); ms = ctx.timer_stop() / 30
print(f"B={B:2d} T={T}: {ms:.3f} ms/step  {B*1e3/ms:.0f} tok/s", flush=True)
