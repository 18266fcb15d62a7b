"""Batched decode GEMVs (gemv_mfma2_kernel) at Llama-3-8B shapes, rotating weights: time and weight bandwidth per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
dev = "cuda"
d, F, QKV = 4096, 14336, 6144
B = int(os.environ.get("B", 32))
shapes = [("gateup", F, d, 4), ("qkv", QKV, d, 0), ("o+res", d, d, 1), ("down+res", d, F, 1)]
for name, N, K, epi in shapes:
    rows = 2 * N if epi == 4 else N
    nbytes = rows * K * 2
    ncopy = max(2, int(1.2e9 // nbytes) + 1)
    Ws = [torch.randn(rows, K, device=dev).bfloat16() for _ in range(ncopy)]
    x = torch.randn(B, K, device=dev).bfloat16()
    res = torch.randn(B, N, device=dev).bfloat16() if epi == 1 else None
    out = torch.empty(B, N, device=dev, dtype=torch.bfloat16)
    for W in Ws:
        ctx.gemv(W, x, resid=res, epi=epi, out=out)
    reps = max(8, 64 // ncopy)
    ctx.timer_start()
    for _ in range(reps):
        for W in Ws:
            ctx.gemv(W, x, resid=res, epi=epi, out=out)
    ms = ctx.timer_stop() / (reps * ncopy)
    print(f"B={B} {name:9s} N={N:6d} K={K:5d} {nbytes/1e6:8.1f} MB  {ms*1e3:7.2f} us  {nbytes/1e9/(ms/1e3):7.1f} GB/s", flush=True)
