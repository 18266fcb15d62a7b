"""The four ESM2-650M GEMM shapes at batch 32 (M = 32832): shipped epilogue vs plain store (what the epilogue costs),
and the mid-M kernel (PCY_GEMM_MID) vs the 256x256 one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
M = int(os.environ.get("M", 32832))
shapes = [("qkv", 3840, 1280, 0), ("o+res", 1280, 1280, 1), ("fc1+gelu", 5120, 1280, 3), ("fc2+res", 1280, 5120, 1)]
for name, N, K, epi in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16(); r = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for e in (epi, 0):
        rr = r if e == 1 else None
        for _ in range(2): ctx.gemm(A, W, b, rr, e, out=out)
        ctx.timer_start(); n = 10
        for _ in range(n): ctx.gemm(A, W, b, rr, e, out=out)
        ms = ctx.timer_stop() / n
        print(f"{name:9s} epi={e}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
