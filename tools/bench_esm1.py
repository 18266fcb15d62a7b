"""ESM2-650M encode of ONE 1024-residue protein (M = 1026 tokens: the headline config's encoder phase): kernel time per call
(device timer on the engine's stream), for the per-kernel table of the small-M path.  N=<proteins> changes the batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, EsmConfig, EsmEngine
kw = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
eng = EsmEngine(synth.esm_state_dict(**kw, device="cuda"), EsmConfig(**kw))
ctx = Context.get()
B = int(os.environ.get("N", 1))
L = int(os.environ.get("L", 1024))
toks = synth.protein_tokens([L] * B, seed=1)
for _ in range(3): eng.forward(toks)
torch.cuda.synchronize()
n = 10
ctx.timer_start()
for _ in range(n): z = eng.forward(toks)
ms = ctx.timer_stop() / n
S = L + 2
fl = B * (2 * 648806400 * S + 168960 * S * S)
print(f"esm encode {B} x {L} residues: {ms:.3f} ms per call  {fl / ms / 1e9:.0f} TFLOP/s algorithmic = {fl / ms / 1e9 / 2500:.3f} of 2.5 PF")
