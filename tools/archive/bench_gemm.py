import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
shapes = [("esm qkv", 32832, 3840, 1280, 0), ("esm wo+res", 32832, 1280, 1280, 1), ("esm fc1+gelu", 32832, 5120, 1280, 3), ("esm fc2+res", 32832, 1280, 5120, 1),
          ("llama qkv T512", 512, 6144, 4096, 0), ("llama gateup T512", 512, 28672, 4096, 4), ("llama down T512", 512, 4096, 14336, 1), ("sq 4096", 4096, 4096, 4096, 0)]
for name, M, N, K, epi in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16() if epi in (0, 1, 3) else None
    Nout = N // 2 if epi == 4 else N
    r = torch.randn(M, Nout, device="cuda").bfloat16() if epi == 1 else None
    out = torch.empty(M, Nout, device="cuda", dtype=torch.bfloat16)
    for _ in range(2): ctx.gemm(A, W, b, r, epi, out=out)
    ctx.timer_start(); n = 10
    for _ in range(n): ctx.gemm(A, W, b, r, epi, out=out)
    ms = ctx.timer_stop() / n
    print(f"{name:18s} M={M:6d} N={N:6d} K={K:6d}  {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
