"""pcy_esm_encode's replayed launch chain: how many of the calls of a bench-style loop (three unsynchronised warm-up calls, then ten timed
ones) are replays, and the time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth, _lib as L
from procyon_amd.engine import Context, EsmConfig, EsmEngine
kw = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
eng = EsmEngine(synth.esm_state_dict(**kw, device="cuda"), EsmConfig(**kw))
ctx = Context.get()
toks = synth.protein_tokens([1024], seed=1)
lib = L.load()
cnt = lambda: int(lib.pcy_debug_dispatch_count(9))
for _ in range(3): eng.forward(toks)
torch.cuda.synchronize()
print("replays after 3 warm-up calls:", cnt())
for rep in range(3):
    c0 = cnt()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(10): z = eng.forward(toks)
    ms = ctx.timer_stop() / 10
    torch.cuda.synchronize()
    print(f"rep {rep}: {ms:.3f} ms per call (device timer), wall {(time.perf_counter() - t0) * 100:.3f} ms per call, replays {cnt() - c0} of 10", flush=True)
# per-call host time and device completion of a fresh shape (another protein length): where do the slow calls sit?
toks2 = synth.protein_tokens([1000], seed=2)
for _ in range(3): eng.forward(toks2)
torch.cuda.synchronize()
ts = []
for i in range(12):
    t0 = time.perf_counter(); z = eng.forward(toks2); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e3, 2), round((t2 - t0) * 1e3, 2)))
print("per call (host returns, device done) ms:", ts)
