"""Decode-only tokens/s for Llama-3-8B geometry, used for A/B of decode-path changes (PCY_DISABLE=decode_layer/1, PCY_DISABLE=attn_o/1, GRAPH=0/1; PCY_MC_TRACE=1 GRAPH=0: in-kernel stamps):
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
    lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libpcy.so"))
    n = 2 * 128 * 256 * 16
    buf = np.zeros(n, dtype=np.uint64)
    lib.pcy_debug_mc_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.pcy_debug_mc_trace(buf.ctypes.data, n)
    t = buf.reshape(2, 128, 256, 16).astype(np.int64)
    def show(tl, names, title):
        t0 = tl[:, 0].min()
        r = (tl - t0) / 100.0   # us (100 MHz)
        print(title, "(us since first WG start)  min / median / max")
        for i, nm in enumerate(names):
            if nm == "-" or tl[:, i].max() == 0: continue
            print(f"  {nm:22s} {r[:, i].min():7.2f} {np.median(r[:, i]):7.2f} {r[:, i].max():7.2f}")
    for l in (1, 16, 30):
        show(t[1, l, :64], ["start", "q/k/v staged", "attention done", "x after o in LDS", "-", "layer end"], f"layer {l} attention block, 64 attention WGs")
        show(t[1, l, 64:], ["start", "qkv rows stored", "ao sample seen", "ao in LDS", "o done", "layer end", "x sample seen", "x fetched", "-",
                            "gate/up done (wave 0)", "act first half in LDS", "act second half in LDS"], f"layer {l} attention block, 192 projection WGs")
        show(t[1, l, :64], ["start", "-", "-", "-", "-", "-", "-", "-", "-", "gate/up done (wave 0)", "act first half in LDS", "act second half in LDS"], f"layer {l} MLP stamps, 64 attention WGs")
        if t[0, l].max() > 0:
            show(t[0, l], ["start", "s1 done(w0)", "act A in LDS", "act B in LDS", "s2 done"], f"layer {l} MLP chain")
            print("  gap attention block end -> chain start: %.2f us ; chain end -> next block start: %.2f us" % (
                (t[0, l, :, 0].min() - t[1, l, :, :5].max()) / 100.0, (t[1, l + 1, :, 0].min() - t[0, l, :, :5].max()) / 100.0))
        else:
            print("  last layer end -> first next-layer start: %.2f us ; per workgroup, own layer end -> own next start: median %.2f us" % (
                (t[1, l + 1, :, 0].min() - t[1, l, :, :6].max()) / 100.0, np.median(t[1, l + 1, :, 0] - t[1, l, :, 5]) / 100.0))
            nx = (t[1, l + 1] - t[1, l, :, 0].min()) / 100.0
            print("  next layer (same clock): qkv rows stored min/median/max %.2f %.2f %.2f" % (nx[64:, 1].min(), np.median(nx[64:, 1]), nx[64:, 1].max()))
