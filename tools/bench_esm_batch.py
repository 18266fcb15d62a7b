"""ESM2-650M encoder at the retrieval batch (25 x 1024 residues): median ms per batch over N calls, one call per timed region."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, EsmConfig, EsmEngine
kw = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
eng = EsmEngine(synth.esm_state_dict(**kw, device="cuda"), EsmConfig(**kw))
ctx = Context.get()
B, N = int(os.environ.get("B", 25)), int(os.environ.get("N", 12))
toks = synth.protein_tokens([1024] * B, seed=1)
for _ in range(2): eng.forward(toks)
ctx.sync()
ts = []
for _ in range(N):
    ctx.timer_start(); eng.forward(toks); ts.append(ctx.timer_stop())
ts.sort()
ms = ts[len(ts) // 2]
fl = (2 * 648806400 * 1026 + 168960 * 1026 ** 2) * B
print(f"B={B}: median {ms:.2f} ms (min {ts[0]:.2f}, max {ts[-1]:.2f})  {B / ms * 1e3:.1f} proteins/s  {fl / ms / 2.5e12 * 100:.2f} % of 2.5 PF")
