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
if os.environ.get("PCY_PIPE_TRACE"):
    import ctypes as C
    buf = (C.c_ulonglong * 4096)()
    n = ctx.lib.pcy_debug_pipe_trace(ctx.h, buf, 4096)
    t = [buf[i] for i in range(n)]
    t0 = t[0]
    names = ["embed"] + ["qkv", "attn", "o", "gateup", "down"] * 32 + ["lm_head"]
    print("stage           entry    ready     done     flag   (us from the step's first stamp; workgroup 0)")
    for s_ in list(range(0, 12)) + list(range(n // 4 - 7, n // 4)):
        e, r, d, f = [(x - t0) / 100.0 for x in t[4 * s_:4 * s_ + 4]]
        print(f"{s_:4d} {names[s_]:8s} {e:8.2f} {r:8.2f} {d:8.2f} {f:8.2f}   wait {r - e:6.2f}  work {d - r:6.2f}  publish {f - d:5.2f}")
    L5 = [t[4 * (1 + 5 * l) + 0] for l in range(32)]
    print("layer period (us):", [round((L5[i + 1] - L5[i]) / 100.0, 1) for i in range(0, 31, 5)])
