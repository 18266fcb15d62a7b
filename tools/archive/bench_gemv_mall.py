"""Batched decode GEMVs at Llama-3-8B shapes: COLD weights (rotating over > 1 GB of copies) vs weights that a reader kernel has just
pulled through the memory-side cache (256 MB Infinity Cache) -- does a prefetch of the next launch's weights pay?  The reader is a torch
reduction over the matrix (plain loads), timed separately; the GEMV is timed alone (HIP events around the one launch)."""
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
    ncopy = max(4, int(1.2e9 // nbytes) + 1)
    Ws = [torch.randn(rows, K, device=dev).bfloat16() for _ in range(ncopy)]
    x = torch.randn(B, K, device=dev).bfloat16()
    res = torch.randn(B, N, device=dev).bfloat16() if epi == 1 else None
    out = torch.empty(B, N, device=dev, dtype=torch.bfloat16)
    for W in Ws:
        ctx.gemv(W, x, resid=res, epi=epi, out=out)

    def one(W):
        ctx.timer_start()
        ctx.gemv(W, x, resid=res, epi=epi, out=out)
        return ctx.timer_stop() * 1e3

    cold = sorted(one(W) for _ in range(3) for W in Ws)
    hot = sorted(one(Ws[0]) for _ in range(3 * ncopy))
    pre = []
    frac = float(os.environ.get("FRAC", 1.0))
    for _ in range(3):
        for W in Ws:
            v = W.view(torch.int32)[: int(rows * frac)]
            v.sum()                                  # the reader: pulls (a fraction of) W through the memory-side cache
            pre.append(one(W))
    pre.sort()
    med = lambda v: v[len(v) // 2]
    print(f"B={B} {name:9s} {nbytes/1e6:7.1f} MB  cold {med(cold):6.1f} us ({nbytes/1e3/med(cold):6.0f} GB/s)  same W again {med(hot):6.1f} us  "
          f"after a reader of {frac:.2f} W {med(pre):6.1f} us ({nbytes/1e3/med(pre):6.0f} GB/s)", flush=True)
