import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
T, N = int(os.environ.get("T", 128)), 64
for B in [int(x) for x in os.environ.get("BATCHES", "1,4,5,8,16,20,32").split(",")]:
    emb = (torch.randn(B, T, 4096, device="cuda") * 0.02).bfloat16()
    cache = eng.new_cache(B, T + N)
    st = GenState(B, kw["vocab"], N, "cuda")
    logits, _ = eng.prefill(emb, None, cache, "last")
    st.logits.copy_(logits); st.pos.fill_(T)
    eng.pick(cache, st, B, advance_pos=False)
    if os.environ.get("PCY_MC_TRACE"): eng.greedy_steps(cache, st, B, 1, use_graph=False)   # the trace buffer is allocated outside capture
    eng.greedy_steps(cache, st, B, 4)
    ctx.timer_start(); eng.greedy_steps(cache, st, B, 32); ms = ctx.timer_stop() / 32
    print(f"B={B:2d}: {ms:.3f} ms/step  {B*1e3/ms:.0f} tok/s", flush=True)
    if os.environ.get("PCY_MC_TRACE"):
        import ctypes, numpy as np
        from procyon_amd import _lib as L
        n = 32 * 256 * 16
        buf = np.zeros(n, dtype=np.uint64)
        L.load().pcy_debug_mc_trace(buf.ctypes.data, n)
        tr = buf.reshape(32, 256, 16).astype(np.int64)
        lay = tr[5]                                    # one middle layer
        t0 = lay[:, 0].min()
        names = ["start", "o run end", "arrive1", "pass1", "arrive2", "pass2", "gate/up end+arrive3", "pass3", "arrive4", "pass4", "arrive5", "pass5", "end"]
        for i, nm in enumerate(names):
            col = (lay[:, i] - t0) / 100.0             # 100 MHz clock -> us
            print(f"  {nm:22s} min {col.min():7.2f}  med {np.median(col):7.2f}  max {col.max():7.2f} us")
