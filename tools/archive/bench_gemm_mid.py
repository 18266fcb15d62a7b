"""gemm_kernel_mid configurations (PCY_GEMM_MID, read per call) against the launcher's own choice on the single-protein ESM2-650M
shapes (M = 1026) and the single-prompt Llama-3-8B shapes (M = 512): time per launch (interleaved rounds, medians) and bit-equality
with the default kernel's output (same k order per element: every non-split configuration must reproduce it exactly).
SHAPES=esm|llama|all, CFGS="1,2,3" to restrict."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import _lib as L
from procyon_amd.engine import Context
ctx = Context.get()
ESM = [("esm qkv", 1026, 3840, 1280, L.EPI_STORE), ("esm o", 1026, 1280, 1280, L.EPI_RESID),
       ("esm fc1", 1026, 5120, 1280, L.EPI_GELU_ESM), ("esm fc2", 1026, 1280, 5120, L.EPI_RESID)]
LLAMA = [("llama qkv", 512, 6144, 4096, L.EPI_STORE), ("llama o", 512, 4096, 4096, L.EPI_RESID),
         ("llama gate/up", 512, 28672, 4096, L.EPI_SWIGLU), ("llama down", 512, 4096, 14336, L.EPI_RESID)]
which = os.environ.get("SHAPES", "all")
shapes = (ESM if which in ("esm", "all") else []) + (LLAMA if which in ("llama", "all") else [])
cfgs = [int(c) for c in os.environ.get("CFGS", "1,2,3,4,5,6,7,8,9,10,11,12,13,14").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, epi in shapes:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    # COLD weights, as inside the encoder / the prefill (every layer has its own): rotate over enough copies to outgrow the 256 MiB
    # Infinity Cache, so that every launch streams its W panels from HBM; the activations stay hot (the previous kernel wrote them)
    nW = max(2, min(96, int(700e6 / (N * K * 2)) + 1)) if os.environ.get("COLD", "1") != "0" else 1
    Ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(nW)]
    W = Ws[0]
    bias = (torch.randn(N, device="cuda", generator=g) * 0.1).bfloat16() if epi != L.EPI_SWIGLU else None
    Nout = N // 2 if epi == L.EPI_SWIGLU else N
    resid = torch.randn(M, Nout, device="cuda", generator=g).bfloat16() if epi == L.EPI_RESID else None
    os.environ.pop("PCY_GEMM_MID", None)
    ref = ctx.gemm(A, W, bias, resid, epi).clone()
    out = torch.empty_like(ref)
    times = {c: [] for c in [0] + cfgs}
    ok = {}
    for c in cfgs:
        os.environ["PCY_GEMM_MID"] = str(c)
        out.zero_()
        ctx.gemm(A, W, bias, resid, epi, out=out)
        ok[c] = bool(torch.equal(out, ref))
    for rnd in range(5):
        for c in [0] + cfgs:
            os.environ["PCY_GEMM_MID"] = str(c)
            for i in range(2): ctx.gemm(A, Ws[i % nW], bias, resid, epi, out=out)
            ctx.timer_start(); n = 24
            for i in range(n): ctx.gemm(A, Ws[(i + 2) % nW], bias, resid, epi, out=out)
            times[c].append(ctx.timer_stop() / n * 1e3)
    fl = 2.0 * M * N * K
    print(f"--- {name}: M={M} N={N} K={K} epi={epi}  ({nW} weight copies)")
    for c in [0] + cfgs:
        t = statistics.median(times[c])
        print(f"  cfg {c:2d}: {t:7.1f} us (min {min(times[c]):7.1f})  {fl / t / 1e6:7.1f} TF/s  {'' if c == 0 else ('bit-equal' if ok[c] else 'DIFFERS')}", flush=True)
os.environ.pop("PCY_GEMM_MID", None)
