"""Per-tile fixed cost of the 256x256 GEMM: time over K at fixed M, N (plain store epilogue) -> t = rounds * (a + b * K/64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
for M, N in ((25650, 5120), (25600, 5120), (4096, 4096), (25650, 1280)):
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    for K in (128, 256, 640, 1280, 2560, 5120):
        A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(2): ctx.gemm(A, W, None, None, 0, out=out)
        ctx.timer_start(); n = 10
        for _ in range(n): ctx.gemm(A, W, None, None, 0, out=out)
        ms = ctx.timer_stop() / n
        rounds = -(-tiles // 256)
        print(f"M={M} N={N} K={K:5d}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s  tiles {tiles} rounds {rounds}  per tile-round {ms*1e3/rounds:6.1f} us  per k-step {ms*1e3/rounds/(K/64):5.2f}", flush=True)
