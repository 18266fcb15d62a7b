"""Llama prefill attention (head_dim 128, GQA 32/8, causal) of ONE prompt over T, and of a batch: time per launch (includes the V transpose)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
H, Hkv, dh = 32, 8, 128
for lens in ([128], [256], [512], [1024], [2048], [512] * 8, [450] * 64):
    n = sum(lens)
    q = torch.randn(n, H * dh, device="cuda").bfloat16(); k = torch.randn(n, Hkv * dh, device="cuda").bfloat16(); v = torch.randn(n, Hkv * dh, device="cuda").bfloat16()
    for _ in range(2): ctx.attention(q, k, v, lens, H, Hkv, dh, True, dh ** -0.5)
    ctx.sync()
    ctx.timer_start(); r = 10
    for _ in range(r): ctx.attention(q, k, v, lens, H, Hkv, dh, True, dh ** -0.5)
    ms = ctx.timer_stop() / r
    fl = sum(4 * t * t * H * dh / 2 for t in lens)
    print(f"lens {lens[0]} x {len(lens)}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s (causal half counted)", flush=True)
