"""Decode steps only (for the rocprofv3 --pmc passes behind profiles/r0N_pmc_decode_step*.json): Llama-3-8B geometry (GEO=split: Llama-2-7B,
ProCyon-Split -- decode_step_mha_kernel), 24 greedy steps at t = 512."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
if os.environ.get("GEO") == "split":
    kw = dict(vocab=32007, d=4096, n_layers=32, n_heads=32, n_kv_heads=32, ffn=11008)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
T, N = 512, 32
B = int(os.environ.get("ROWS", 1))     # ROWS=4: the small-batch step (decode_step_nb_kernel)
emb = (torch.randn(B, T, 4096, device="cuda") * 0.02).bfloat16()
cache = eng.new_cache(B, T + N)
st = GenState(B, kw["vocab"], N, "cuda")
logits, _ = eng.prefill(emb, None, cache, "last")
st.logits.copy_(logits); st.pos.fill_(T)
eng.pick(cache, st, B, advance_pos=False)
eng.greedy_steps(cache, st, B, 24, use_graph=False)
Context.get().sync()
