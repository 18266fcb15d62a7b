"""The decode MLP chain launch (pcy_decode_mlp: gate/up + SwiGLU + down + residual of one token, Llama-3-8B shapes), rotating
over distinct weight copies so neither the 256 MiB Infinity Cache nor L2 can serve them.  Used for the rocprofv3 --pmc passes
behind profiles/r02_pmc_mlp_chain.json and for A/B timing (PCY_DISABLE=mlp_chain: the two GEMV launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
dev = "cuda"
d, F = 4096, 14336
ncopy = 6
wgu = [(torch.randn(2 * F, d, device=dev) * 0.02).bfloat16() for _ in range(ncopy)]
wdn = [(torch.randn(d, F, device=dev) * 0.01).bfloat16() for _ in range(ncopy)]
ln = torch.ones(d, device=dev).bfloat16()
x0 = torch.randn(1, d, device=dev).bfloat16()
x = x0.clone()
for i in range(ncopy):
    ctx.decode_mlp(x, ln, wgu[i], wdn[i])
reps = 16
x.copy_(x0)
ctx.timer_start()
for _ in range(reps):
    for i in range(ncopy):
        ctx.decode_mlp(x, ln, wgu[i], wdn[i])
ms = ctx.timer_stop() / (reps * ncopy)
nbytes = 3 * F * d * 2
print(f"decode mlp (chain={'mlp_chain' not in os.environ.get('PCY_DISABLE', '')}) {nbytes/1e6:8.1f} MB  {ms*1e3:7.2f} us per call  {nbytes/1e9/(ms/1e3):7.1f} GB/s", flush=True)
