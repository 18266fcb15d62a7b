import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
M, N = 32832, 3840
for K in (64, 128, 256, 640, 1280, 2560):
    for epi in (0, 1, 3):
        A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        b = torch.randn(N, device="cuda").bfloat16(); r = torch.randn(M, N, device="cuda").bfloat16() if epi == 1 else None
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(2): ctx.gemm(A, W, b, r, epi, out=out)
        ctx.timer_start(); n = 10
        for _ in range(n): ctx.gemm(A, W, b, r, epi, out=out)
        ms = ctx.timer_stop() / n
        tiles = ((M + 127) // 128) * (N // 128)
        print(f"K={K:5d} epi={epi}: {ms*1e3:8.1f} us  per tile (x512 slots) {ms*1e3*512/tiles:6.2f} us  {2*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
