"""ESM / Llama GEMM shapes: the engine's kernels next to torch.matmul (hipBLASLt) on the same random bf16 data -- what a tuned
library reaches on this box for these shapes (a ceiling estimate for the plain-HIP kernels; never part of the product path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
from procyon_amd import _lib as L
ctx = Context.get()
M = 25650
shapes = [("esm qkv", M, 3840, 1280), ("esm o", M, 1280, 1280), ("esm fc1", M, 5120, 1280), ("esm fc2", M, 1280, 5120),
          ("llama qkv T512", 512, 6144, 4096), ("llama gate/up T512", 512, 28672, 4096), ("llama down T512", 512, 4096, 14336),
          ("llama gate/up M28800", 28800, 28672, 4096), ("4096^3", 4096, 4096, 4096)]
def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    for name, m, n, k in shapes:
        a = (torch.randn(m, k, device="cuda")).bfloat16()
        w = (torch.randn(n, k, device="cuda") * 0.03).bfloat16()
        out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
        t_ref = tm(lambda: torch.matmul(a, w.T, out=out))
        t_pcy = tm(lambda: ctx.gemm(a, w, out=out))
        fl = 2.0 * m * n * k
        print(f"{name:22s} M={m:6d} N={n:6d} K={k:6d}: hipBLASLt {t_ref:8.1f} us {fl/t_ref/1e6:7.0f} TF | pcy {t_pcy:8.1f} us {fl/t_pcy/1e6:7.0f} TF", flush=True)
