"""256x256 GEMM time over M around the ESM batch (25 x 1026 = 25650 rows: a ragged last row tile) at the encoder's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
for rep in range(2):
    for N, K in ((5120, 1280), (3840, 1280), (1280, 1280), (1280, 5120)):
        for M in (25600, 25650, 25856, 25601, 25728):
            A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            for _ in range(3): ctx.gemm(A, W, None, None, 0, out=out)
            ctx.timer_start(); n = 20
            for _ in range(n): ctx.gemm(A, W, None, None, 0, out=out)
            ms = ctx.timer_stop() / n
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            print(f"N={N} K={K} M={M}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s  tiles {tiles} ({tiles/256:.2f} rounds)", flush=True)
