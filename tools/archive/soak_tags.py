"""Tag wrap soak: > 65536 passes over the decoder layers (the 16-bit tag of the in-launch hand-overs wraps), then the next ordinary
steps must still equal a run that never wrapped, and the watchdog word must be clear."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
kw = dict(vocab=4096, d=4096, n_layers=2, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=512))
ctx = Context.get()
emb = (torch.randn(1, 70, 4096, device="cuda") * 0.02).bfloat16()
def run(passes):
    cache = eng.new_cache(1, 96)
    st = GenState(1, kw["vocab"], 16, "cuda")
    logits, _ = eng.prefill(emb, None, cache, "last")
    st.logits.copy_(logits); st.pos.fill_(70)
    eng.pick(cache, st, 1, advance_pos=False)
    eng.greedy_steps(cache, st, 1, 3)
    t0 = time.perf_counter()
    done = 0
    while done < passes:
        n = min(4096, passes - done)
        eng.decode_layers(cache, st, 1, n); ctx.sync(); done += n
    dt = time.perf_counter() - t0
    eng.greedy_steps(cache, st, 1, 3)
    ctx.sync()
    return st.logits.cpu().clone(), st.tokens_out.cpu().clone(), dt
a = run(0)
b = run(int(os.environ.get("PASSES", 70000)))
assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
print(f"tag wrap soak ok: {os.environ.get('PASSES', 70000)} passes in {b[2]:.1f} s")
