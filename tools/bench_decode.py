"""Decode-only tokens/s for Llama-3-8B geometry, used for A/B of decode-path changes (PCY_DECODE_PIPE=0/1, GRAPH=0/1):
GPU time per token (HIP events) and host time to enqueue a step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
ctx = Context.get()
T, N = int(os.environ.get("T", 512)), 256
graph = os.environ.get("GRAPH", "1") != "0"
emb = (torch.randn(1, T, 4096, device="cuda") * 0.02).bfloat16()
cache = eng.new_cache(1, T + N)
st = GenState(1, kw["vocab"], N, "cuda")
logits, _ = eng.prefill(emb, None, cache, "last")
st.logits.copy_(logits); st.pos.fill_(T)
eng.pick(cache, st, 1, advance_pos=False)
eng.greedy_steps(cache, st, 1, 8, use_graph=graph)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    ctx.timer_start(); eng.greedy_steps(cache, st, 1, 60, use_graph=graph); host = (time.perf_counter() - t0) / 60
    ms = ctx.timer_stop() / 60
    print(f"decode {ms:.3f} ms/token  {1e3/ms:.1f} tok/s  {15.09/ms:.2f} TB/s   host enqueue {host*1e3:.3f} ms/step", flush=True)
ctx.sync()
if os.environ.get("PCY_MC_TRACE"):
    import ctypes, numpy as np
    from procyon_amd import _lib
    L = _lib.lib() if hasattr(_lib, "lib") else None
    lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libpcy.so"))
    n = 32 * 256 * 16
    buf = np.zeros(n, dtype=np.uint64)
    lib.pcy_debug_mc_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.pcy_debug_mc_trace(buf.ctypes.data, n)
    t = buf.reshape(32, 256, 16)[:, :, :8].astype(np.int64)
    for l in (1, 10, 20, 31):
        tl = t[l]
        t0 = tl[:, 0].min()
        r = (tl - t0) / 100.0   # us (100 MHz)
        names = ["start", "s1 done(w0)", "act A in LDS", "act B in LDS", "s2 done", "x in LDS", "rms done", "end"]
        print(f"layer {l}: (us since first WG start)  min / median / max over 256 WGs")
        for i, nm in enumerate(names):
            if tl[:, i].max() == 0: continue
            print(f"  {nm:14s} {r[:, i].min():7.2f} {np.median(r[:, i]):7.2f} {r[:, i].max():7.2f}")
    s1 = (t[1:32, :, 1] - t[1:32, :, 0].min(axis=1, keepdims=True)) / 100.0   # [layers, WG] stage-1 finishing time of wave 0
    dev = s1 - np.median(s1, axis=1, keepdims=True)
    print("stage-1 finishing time minus the layer median, mean over layers, per XCD (workgroup % 8):", np.round([dev[:, x::8].mean() for x in range(8)], 2))
    m = dev.mean(axis=0)
    print("per-workgroup mean deviation: min %.2f max %.2f std %.2f ; std of the residual (per layer noise) %.2f" % (m.min(), m.max(), m.std(), (dev - m).std()))
    order = np.argsort(m)
    print("slowest workgroups:", order[-16:], np.round(m[order[-16:]], 1))
    print("fastest workgroups:", order[:16], np.round(m[order[:16]], 1))
    s2 = (t[1:31, :, 4] - t[1:31, :, 0].min(axis=1, keepdims=True)) / 100.0
    dev2 = s2 - np.median(s2, axis=1, keepdims=True)
    m2 = dev2.mean(axis=0)
    print("stage-2: per-workgroup mean deviation: min %.2f max %.2f std %.2f ; residual std %.2f ; corr with stage 1 %.2f" % (m2.min(), m2.max(), m2.std(), (dev2 - m2).std(), np.corrcoef(m, m2)[0, 1]))
    print("per XCD stage 2:", np.round([dev2[:, x::8].mean() for x in range(8)], 2))
