"""ProCyon-Split geometry (Llama-2-7B) at beam-search batch sizes: ms per decode step per batch size (launch-per-stage MFMA GEMVs; the one-launch
step covers one row).  BATCHES=1,2,4,5,10,20  T=<prompt tokens>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
kw = dict(vocab=32007, d=4096, n_layers=32, n_heads=32, n_kv_heads=32, ffn=11008)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
T, N = int(os.environ.get("T", 700)), 80
wbytes = (32 * (12288 * 4096 + 4096 * 4096 + 3 * 11008 * 4096) * 2 + 32007 * 4096 * 2)
for B in [int(x) for x in os.environ.get("BATCHES", "1,2,4,5,10,20").split(",")]:
    emb = (torch.randn(B, T, 4096, device="cuda") * 0.02).bfloat16()
    cache = eng.new_cache(B, T + N)
    st = GenState(B, kw["vocab"], N, "cuda")
    logits, _ = eng.prefill(emb, None, cache, "last")
    st.logits.copy_(logits); st.pos.fill_(T)
    eng.pick(cache, st, B, advance_pos=False)
    eng.greedy_steps(cache, st, B, 4)
    ctx.timer_start(); eng.greedy_steps(cache, st, B, 64); ms = ctx.timer_stop() / 64
    ctx.sync()
    gb = (wbytes + B * 32 * (T + N // 2) * 2 * 4096 * 2) / 1e9
    print(f"B={B:2d} t~{T + N // 2}: {ms:.3f} ms/step ({B * 1e3 / ms:.0f} tok/s, {gb / ms:.2f} TB/s = {gb / ms / 8:.3f} of peak)", flush=True)
