"""Feasibility probe: do two decode GEMV launches on two HIP streams overlap, i.e. does the second kernel's fill hide the
first one's drain + the kernel boundary?  Aggregate bandwidth of independent GEMVs alternating between two streams vs all on
one stream, at the four Llama-3-8B decode shapes (weights rotate over > 1 GB so nothing is cache-resident)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context

dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    c1 = Context.get(dev)
with torch.cuda.stream(s2):
    c2 = Context.get(dev)
assert c1 is not c2
d, F, QKV = 4096, 14336, 6144
shapes = [("qkv+rms", QKV, d, 0, True), ("o+res", d, d, 1, False), ("gateup+rms", F, d, 4, True), ("down+res", d, F, 1, False)]
for name, N, K, epi, rms in shapes:
    rows = 2 * N if epi == 4 else N
    nbytes = rows * K * 2
    ncopy = max(4, int(1.2e9 // nbytes) + 1)
    ncopy += ncopy & 1
    Ws = [torch.randn(rows, K, device=dev).bfloat16() for _ in range(ncopy)]
    x = torch.randn(1, K, device=dev).bfloat16()
    w = torch.ones(K, device=dev).bfloat16() if rms else None
    res = torch.randn(1, N, device=dev).bfloat16() if epi == 1 else None
    outs = [torch.empty(1, N, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    torch.cuda.synchronize()
    reps = max(8, 128 // ncopy)
    for mode in ("one stream", "two streams"):
        def run():
            for _ in range(reps):
                for i, W in enumerate(Ws):
                    if mode == "two streams" and (i & 1):
                        with torch.cuda.stream(s2):
                            c2.gemv(W, x, resid=res, epi=epi, rms_w=w, out=outs[1])
                    else:
                        with torch.cuda.stream(s1):
                            c1.gemv(W, x, resid=res, epi=epi, rms_w=w, out=outs[0])
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        per = dt / (reps * ncopy)
        print(f"{name:12s} {mode:11s} {nbytes/1e6:8.1f} MB  {per*1e6:7.2f} us/launch  {nbytes/1e9/per:7.1f} GB/s aggregate", flush=True)
