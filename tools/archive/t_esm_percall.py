"""EsmEngine.forward: back-to-back calls (host packing overlaps GPU encoding) and per-call timing with a sync after each,
for the A/B of the index-upload path (PCY_STAGER=0: pinned copy per array, 1: persistent staging ring)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import EsmConfig, EsmEngine
kw = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
eng = EsmEngine(synth.esm_state_dict(**kw, device="cuda"), EsmConfig(**kw))
for B in (8, 25, 8, 25):
    toks = synth.protein_tokens([1024] * B, seed=1)
    eng.forward(toks); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); z = eng.forward(toks); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    for _ in range(8): z = eng.forward(toks)
    torch.cuda.synchronize(); bb = (time.perf_counter() - t0) / 8 * 1e3
    print(f"B={B:2d} per-call {[round(t, 1) for t in ts]}  back-to-back {bb:.1f} ms/call", flush=True)
