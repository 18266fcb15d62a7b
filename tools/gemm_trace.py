"""In-kernel stamps of gemm_kernel_big (PCY_GEMM_TRACE=1): prologue / mainloop / epilogue per tile and the gap between
consecutive workgroups on one CU."""
import os, sys, ctypes
os.environ["PCY_GEMM_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from procyon_amd import _lib
from procyon_amd.engine import Context
ctx = Context.get()
lib = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libpcy.so"))
lib.pcy_debug_mc_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
M, N, K = int(os.environ.get("M", 25600)), int(os.environ.get("N", 5120)), int(os.environ.get("K", 1280))
epi = int(os.environ.get("EPI", 0))
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(N, device="cuda").bfloat16(); r = torch.randn(M, N, device="cuda").bfloat16() if epi == 1 else None
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3): ctx.gemm(A, W, b, r, epi, out=out)
ctx.sync()
tiles = ((M + 255) // 256) * ((N + 255) // 256)
buf = np.zeros(tiles * 8, dtype=np.uint64)
lib.pcy_debug_mc_trace(buf.ctypes.data, tiles * 8)
t = buf.reshape(tiles, 8)
st = t[:, :4].astype(np.int64); st -= st[:, 0].min()
us = st / 100.0
print(f"M={M} N={N} K={K} epi={epi}: {tiles} tiles; kernel span {us[:, 3].max():.1f} us")
print("prologue (start -> first stage landed): median %.2f us; mainloop: median %.2f (%.2f per k-step); epilogue: median %.2f" % (
    np.median(us[:, 1] - us[:, 0]), np.median(us[:, 2] - us[:, 1]), np.median(us[:, 2] - us[:, 1]) / (K / 64), np.median(us[:, 3] - us[:, 2])))
cu = t[:, 4]
gaps, per = [], []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    o = idx[np.argsort(us[idx, 0])]
    per.append(len(o))
    for i in range(1, len(o)):
        gaps.append(us[o[i], 0] - us[o[i - 1], 3])
print("distinct CU ids %d; tiles per CU min %d max %d; gap between workgroups on a CU: median %.2f us, mean %.2f" % (len(per), min(per), max(per), np.median(gaps), np.mean(gaps)))
first = np.sort(us[:, 0])[:256]
print("first-round start times: min %.2f median %.2f max %.2f" % (first.min(), np.median(first), first.max()))
