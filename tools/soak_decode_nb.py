"""Soak of the fused decode steps (1 row and 2..8 rows): batch sizes in random order on ONE context, eager and replayed, every run compared
bit for bit with the first run of its (batch, prompt) -- a stale tag, a slot shared between batch sizes or a lost hand-over shows as a
difference or as a watchdog error.  MINUTES=<wall clock> (default 3), LAYERS=<n> (default 8), GEO=split (the Llama-2-7B geometry: one-launch step at 1 row), ROWS=<max batch> (default 8)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("PCY_NB_MAX", "8")   # the fused step at 8 rows is opt-in since round 6
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
kw = dict(vocab=4096, d=4096, n_layers=int(os.environ.get("LAYERS", 8)), n_heads=32, n_kv_heads=8, ffn=14336)
if os.environ.get("GEO") == "split":   # ProCyon-Split (Llama-2-7B): the one-launch step at 1 row (pcy_decode_mha.hip), launches above
    kw.update(n_kv_heads=32, ffn=11008)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
random.seed(0)
N = 12
cases = [(B, T) for B in range(1, int(os.environ.get("ROWS", 8)) + 1) for T in (40, 333, 900, 1700)]
embs = {}
for B, T in cases:
    torch.manual_seed(B * 10007 + T)
    embs[(B, T)] = (torch.randn(B, T, 4096, device="cuda") * 0.02).bfloat16()
first, runs, bad = {}, 0, 0
t_end = time.time() + 60 * float(os.environ.get("MINUTES", 3))
while time.time() < t_end:
    B, T = random.choice(cases)
    use_graph = random.random() < 0.5
    cache = eng.new_cache(B, T + N + 2)
    st = GenState(B, kw["vocab"], N + 2, "cuda")
    logits, _ = eng.prefill(embs[(B, T)], None, cache, "last")
    st.logits.copy_(logits); st.pos.fill_(T)
    eng.pick(cache, st, B, advance_pos=False)
    k = random.choice((1, 3, N))      # steps per call (a replayed chain of k steps, or k eager steps)
    done = 0
    while done < N:
        n = min(k, N - done)
        eng.greedy_steps(cache, st, B, n, use_graph=use_graph)
        done += n
    ctx.sync()
    got = (st.tokens_out[:, :N + 1].cpu(), st.logits.cpu())
    if (B, T) not in first:
        first[(B, T)] = got
    elif not all(torch.equal(a, b) for a, b in zip(got, first[(B, T)])):
        bad += 1
        print(f"MISMATCH B={B} T={T} graph={use_graph} k={k}", flush=True)
    runs += 1
print(f"soak: {runs} generations of {N} steps over {len(first)} (batch, prompt) cases, mismatches {bad}, finite {all(bool(torch.isfinite(v[1].float()).all()) for v in first.values())}")
print("SOAK", "PASS" if bad == 0 else "FAIL")
