"""Interleaved A/B of an environment switch that the engine reads per call, on the ESM2-650M encoder (batch 25 x 1024 residues):
    VAR=PCY_GEMM_PERM VALS=0,7 python tools/ab_esm_env.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import EsmConfig, EsmEngine
kw = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
eng = EsmEngine(synth.esm_state_dict(**kw, device="cuda"), EsmConfig(**kw))
var, vals = os.environ["VAR"], os.environ.get("VALS", "0,1").split(",")
B = int(os.environ.get("B", 25))
toks = synth.protein_tokens([1024] * B, seed=1)
for v in vals:
    os.environ[var] = v
    eng.forward(toks)
torch.cuda.synchronize()
for rnd in range(4):
    for v in vals:
        os.environ[var] = v
        t0 = time.perf_counter()
        for _ in range(3):
            eng.forward(toks)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"round {rnd} {var}={v}: {dt*1e3:7.2f} ms  {B/dt:7.1f} proteins/s", flush=True)
