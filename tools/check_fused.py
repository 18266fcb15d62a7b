"""A/B of the persistent decode kernel (PCY_DECODE_FUSED=1) against the layer-by-layer launches (=0): same seeded
Llama-3-8B-geometry model and prompt; dumps greedy tokens + logits of every step to gpurun_out/fused_<mode>.pt,
and with `compare` reports the differences.  The env var is read once per process, hence two runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if len(sys.argv) > 1 and sys.argv[1] == "compare":
    fa, fb = (sys.argv[2], sys.argv[3]) if len(sys.argv) > 3 else ("gpurun_out/fused_0.pt", "gpurun_out/fused_1.pt")
    a, b = torch.load(fa), torch.load(fb)
    la, lb = a["logits"].float(), b["logits"].float()
    same = (a["tokens"] == b["tokens"]).all().item()
    rel = ((la - lb).norm(dim=-1) / la.norm(dim=-1))
    print(f"tokens identical: {same}  first mismatch: {(a['tokens'] != b['tokens']).nonzero()[:1].tolist()}")
    print(f"logits rel err per step: max {rel.max():.2e} mean {rel.mean():.2e}; step0 {rel[0]:.2e}")
    print("rel err by step:", " ".join(f"{v:.1e}" for v in rel[:40].tolist()))
    print(f"ms/token layered {a['ms']:.3f}  fused {b['ms']:.3f}")
    for nm in ("k", "v"):
        d = (a[nm] != b[nm])          # [L, Hkv, N, dh]
        if d.any():
            idx = d.nonzero()
            print(f"{nm} cache: {d.sum().item()} differing elements; first (layer, kvh, pos, col): {idx[0].tolist()}")
            l0, h0, p0 = idx[0][:3].tolist()
            cols = d[l0, h0, p0].nonzero().flatten().tolist()
            print(f"   row ({l0},{h0},{p0}): {len(cols)} cols differ: {cols[:40]}")
            per_layer = d.flatten(1).sum(1)
            print("   per-layer counts:", per_layer.tolist())
            print("   per-pos counts in first bad layer:", d[l0].sum((0, 2)).tolist())
        else:
            print(f"{nm} cache identical")
    sys.exit(0)

from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
mode = os.environ.get("PCY_DECODE_FUSED", "1")
L = int(os.environ.get("LAYERS", 32))
kw = dict(vocab=128263, d=4096, n_layers=L, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
T, N = 512, 48
torch.manual_seed(0)
emb = (torch.randn(1, T, 4096, device="cuda") * 0.02).bfloat16()
cache = eng.new_cache(1, T + N + 80)
st = GenState(1, kw["vocab"], N + 80, "cuda")
logits, _ = eng.prefill(emb, None, cache, "last")
st.logits.copy_(logits); st.pos.fill_(T)
eng.pick(cache, st, 1, advance_pos=False)
all_logits = []
for i in range(N):
    eng.greedy_steps(cache, st, 1, 1, use_graph=False)
    all_logits.append(st.logits[0].clone())
ctx.sync()
toks = st.tokens_out[0, :N + 1].cpu().clone()
ctx.timer_start(); eng.greedy_steps(cache, st, 1, 60); ms = ctx.timer_stop() / 60
ctx.sync()
print(f"mode {mode}: {ms:.3f} ms/token  {1e3/ms:.1f} tok/s  tokens {toks[:12].tolist()}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
torch.save({"tokens": toks, "logits": torch.stack(all_logits).cpu(), "ms": ms,
            "k": cache.k[:, 0, :, T:T + N].cpu(), "v": cache.v[:, 0, :, T:T + N].cpu()}, os.environ.get("OUT", f"gpurun_out/fused_{mode}.pt"))
