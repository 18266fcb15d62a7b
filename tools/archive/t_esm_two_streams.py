"""Retrieval-style bulk encoding (ESM2-650M, 1024-residue proteins): ONE launch chain after another on one stream vs TWO chains of
independent batches on two HIP streams (one engine context per stream, shared weights), so that one chain's partial tile rounds,
ramps and tails are filled by the other's workgroups.  NB = proteins per engine call (comma list), N = proteins per pass."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, EsmConfig, EsmEngine
kw = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
eng = EsmEngine(synth.esm_state_dict(**kw, device="cuda"), EsmConfig(**kw))
L = int(os.environ.get("L", 1024))
N = int(os.environ.get("N", 200))
S = L + 2
fl1 = 2 * 648806400 * S + 168960 * S * S
streams = [torch.cuda.Stream() for _ in range(3)]
engs = []
for s in streams:
    with torch.cuda.stream(s):
        e = copy.copy(eng)
        e.ctx = Context.get()
        e._enc_slots = {}
        engs.append(e)


def run(nb, ns):
    toks = [synth.protein_tokens([L] * nb, seed=i) for i in range(ns)]
    nbatch = -(-N // nb)

    def one_pass():
        outs = []
        for i in range(nbatch):
            k = i % ns
            with torch.cuda.stream(streams[k]):
                outs.append(engs[k].forward(toks[k]))
        return outs
    one_pass(); torch.cuda.synchronize()
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        one_pass()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    pps = nbatch * nb / best
    print(f"batch {nb:3d} x {ns} stream(s): {pps:7.1f} proteins/s = {pps * fl1 / 2.5e15:.4f} of 2.5 PF   ({best / nbatch * 1e3:.2f} ms per call)", flush=True)


for nb in [int(x) for x in os.environ.get("NB", "25,13,16,20").split(",")]:
    for ns in (1, 2, 3):
        run(nb, ns)
