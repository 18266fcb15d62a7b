"""Ablation of the single-pass ESM attention kernel (PCY_FA_ABL variants; wrong results, timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
H, dh, lens = 20, 64, [1026] * 25
n = sum(lens)
q, k, v = [(torch.randn(n, H * dh, device="cuda") * (0.3 if i == 0 else 1.0)).bfloat16() for i in range(3)]
os.environ["PCY_ESM_ATTN"] = "fast"
names = {0: "full", 1: "no rescale", 2: "no exp", 3: "no rescale, no exp", 4: "no PV mfma", 8: "no QK mfma", 12: "no mfma", 48: "no LDS frag reads",
         64: "no DMA", 112: "no DMA, no LDS reads", 15: "no mfma/exp/rescale", 127: "nothing"}
for rnd_ in range(2):
    for abl in (0, 1, 2, 3, 4, 8, 12, 48, 64, 112, 15, 127):
        os.environ["PCY_FA_ABL"] = str(abl)
        for _ in range(2): ctx.attention(q, k, v, lens, H, H, dh, False, 1.0)
        ctx.timer_start()
        for _ in range(10): ctx.attention(q, k, v, lens, H, H, dh, False, 1.0)
        ms = ctx.timer_stop() / 10
        print(f"abl {abl:3d} {names[abl]:28s}: {ms*1e3:7.1f} us per launch (incl. ~34 us V transpose)", flush=True)
