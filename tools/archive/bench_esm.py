"""ESM2-650M encode throughput (retrieval unit of work: 1024-residue proteins) at a given batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import EsmConfig, EsmEngine
kw = dict(d=1280, n_layers=int(os.environ.get("LAYERS", 33)), n_heads=20, ffn=5120)
eng = EsmEngine(synth.esm_state_dict(**kw, device="cuda"), EsmConfig(**kw))
for B in [int(x) for x in os.environ.get("BATCHES", "1,16,64").split(",")]:
    toks = synth.protein_tokens([1024] * B, seed=1)
    z = eng.forward(toks); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 3
    for _ in range(n):
        z = eng.forward(toks)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    fl = (2 * 648806400 * 1026 + 168960 * 1026 ** 2) * B * kw["n_layers"] / 33
    print(f"B={B:3d}: {dt*1e3:8.2f} ms  {B/dt:8.1f} proteins/s  {fl/dt/1e12:7.1f} TFLOP/s algorithmic ({fl/dt/2.5e15*100:.1f}% of 2.5 PF)", flush=True)
