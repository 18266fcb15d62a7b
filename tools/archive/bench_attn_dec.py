import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context, rope_tables
ctx = Context.get()
H, Hkv, dh, Tmax, L = 32, 8, 128, 768, 32
cos, sin = rope_tables(dh, 1e4, 1024, "cuda")
qkv = torch.randn(1, (H + 2 * Hkv) * dh, device="cuda").bfloat16()
kcs = [torch.randn(1, Hkv, Tmax, dh, device="cuda").bfloat16() for _ in range(L)]
vcs = [torch.randn(1, Hkv, Tmax, dh, device="cuda").bfloat16() for _ in range(L)]
out = torch.empty(1, H * dh, device="cuda", dtype=torch.bfloat16)
flush = torch.empty(1 << 28, device="cuda", dtype=torch.bfloat16)   # 512 MB
for t in (0, 64, 320, 640, 760):
    pos = torch.tensor([t], dtype=torch.int32, device="cuda")
    for mode in ("warm", "cold"):
        tot = 0.0
        for rep in range(3):
            if mode == "cold":
                flush.zero_()
            ctx.timer_start()
            for l in range(L):
                ctx.attn_decode(qkv, kcs[l], vcs[l], pos, cos, sin, H, Hkv, dh, out=out)
            tot += ctx.timer_stop()
        print(f"t={t:4d} {mode}: {tot / 3 / L * 1e3:7.2f} us per layer-launch (host-paced loop)", flush=True)
