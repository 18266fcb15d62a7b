"""Micro-benchmark of the decode GEMVs at Llama-3-8B shapes (B=1), rotating over distinct weight copies so
neither the 256 MiB Infinity Cache nor L2 can serve them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
dev = "cuda"
d, F, QKV, V = 4096, 14336, 6144, 128263
shapes = [("qkv+rms", QKV, d, 0, True), ("o+res", d, d, 1, False), ("gateup+rms", F, d, 4, True), ("down+res", d, F, 1, False), ("lm_head+rms", V, d, 0, True)]
for name, N, K, epi, rms in shapes:
    rows = 2 * N if epi == 4 else N
    nbytes = rows * K * 2
    ncopy = max(2, int(1.2e9 // nbytes) + 1)
    Ws = [torch.randn(rows, K, device=dev).bfloat16() for _ in range(ncopy)]
    x = torch.randn(1, K, device=dev).bfloat16()
    w = torch.ones(K, device=dev).bfloat16() if rms else None
    res = torch.randn(1, N, device=dev).bfloat16() if epi == 1 else None
    out = torch.empty(1, N, device=dev, dtype=torch.bfloat16)
    for W in Ws:
        ctx.gemv(W, x, resid=res, epi=epi, rms_w=w, out=out)
    reps = max(8, 64 // ncopy)
    ctx.timer_start()
    for _ in range(reps):
        for W in Ws:
            ctx.gemv(W, x, resid=res, epi=epi, rms_w=w, out=out)
    ms = ctx.timer_stop() / (reps * ncopy)
    print(f"{name:12s} N={N:6d} K={K:5d} {nbytes/1e6:8.1f} MB  {ms*1e3:7.2f} us  {nbytes/1e9/(ms/1e3):7.1f} GB/s", flush=True)
