"""Diagnostic: who is closer to the exactly-rounded result, the CPU bf16 F.linear or the HIP kernel?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from procyon_amd.engine import Context
ctx = Context.get()
BF = torch.bfloat16
def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(BF)
def ulps(a, b):
    ai = a.view(torch.int16).int(); bi = b.view(torch.int16).int()
    # monotone integer mapping of bf16
    ai = torch.where(ai < 0, -(ai & 0x7fff), ai); bi = torch.where(bi < 0, -(bi & 0x7fff), bi)
    return (ai - bi).abs()
print(torch.__config__.show().split('\n')[0:3], torch.get_num_threads())
os.system("lscpu | grep -E 'Model name|Flags' | cut -c1-400 | sed -e 's/.*\\(amx[a-z_0-9]*\\).*\\(avx512_bf16\\).*/\\1 \\2/' | head -3")
for kind, M, N, K in [("gemv",1,4096,4096),("gemv",2,4096,4096),("gemv",4,4096,4096),("gemv",2,512,14336),("gemm",17,256,5120),("gemm",200,384,256),("gemm",640,3840,1280),("gemm",1026,1280,1280)]:
    A, W = rnd(M, K, seed=1), rnd(N, K, seed=2, std=0.05)
    cpu = F.linear(A, W)
    exact = (A.double() @ W.double().T).to(BF)
    f32 = (A.float() @ W.float().T).to(BF)
    gpu = (ctx.gemv(W.cuda(), A.cuda()) if kind == "gemv" else ctx.gemm(A.cuda(), W.cuda())).cpu()
    for nm, t in (("cpu_bf16", cpu), ("cpu_f32mm", f32), ("gpu", gpu)):
        u = ulps(t, exact)
        print(f"{kind} M{M} N{N} K{K} {nm:10s} vs exact: mism {int((u>0).sum())}/{u.numel()} max_ulp {int(u.max())}")
