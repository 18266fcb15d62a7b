"""ESM-shaped attention (20 heads x 64, 25 x 1026 tokens): exact two-pass vs single-pass kernel, HIP-event time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
H, dh, lens = 20, 64, [1026] * 25
n = sum(lens)
q, k, v = [(torch.randn(n, H * dh, device="cuda") * (0.3 if i == 0 else 1.0)).bfloat16() for i in range(3)]
for rnd_ in range(2):
    for mode in ("exact", "fast"):
        os.environ["PCY_ESM_ATTN"] = mode
        for _ in range(3): ctx.attention(q, k, v, lens, H, H, dh, False, 1.0)
        ctx.timer_start()
        for _ in range(10): ctx.attention(q, k, v, lens, H, H, dh, False, 1.0)
        ms = ctx.timer_stop() / 10
        fl = 4 * 1026 * 1026 * H * dh * 25
        print(f"{mode}: {ms*1e3:.1f} us per launch (incl. V transpose)  {fl/ms/1e9:.0f} TFLOP/s", flush=True)
