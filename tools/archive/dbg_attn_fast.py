import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd.engine import Context
ctx = Context.get()
H, dh = 3, 64
os.environ["PCY_ESM_ATTN"] = "fast"
def ref64(q, k, v, lens):
    outs, t0 = [], 0
    for n in lens:
        qs, ks, vs = (x[t0:t0 + n].double().view(n, H, dh).transpose(0, 1) for x in (q, k, v))
        outs.append((torch.softmax(qs @ ks.transpose(1, 2), -1) @ vs).transpose(0, 1).reshape(n, H * dh)); t0 += n
    return torch.cat(outs)
for lens in ([1], [33], [63], [64], [65], [1, 33], [64, 1, 33, 700, 257, 1026, 63, 65, 128]):
    n = sum(lens)
    g = torch.Generator().manual_seed(1)
    q, k, v = [(torch.randn(n, H * dh, generator=g) * s).bfloat16() for s in (0.35, 1, 1)]
    out = ctx.attention(q.cuda(), k.cuda(), v.cuda(), lens, H, H, dh, False, 1.0).cpu().double()
    ref = ref64(q, k, v, lens)
    t0 = 0
    for L_ in lens:
        o, r = out[t0:t0 + L_], ref[t0:t0 + L_]
        bad = torch.isnan(o).any(1).nonzero().flatten().tolist()
        err = float((torch.nan_to_num(o) - r).norm() / r.norm())
        print(f"lens={lens} seq len {L_}: nan rows {bad[:8]}{'...' if len(bad) > 8 else ''} ({len(bad)}) err {err:.3e}")
        t0 += L_
print("---- exact then fast, as the test does")
for rep in range(3):
    for lens in ([1026], [64, 1, 33, 700, 257, 1026, 63, 65, 128]):
        n = sum(lens)
        g = torch.Generator().manual_seed(1)
        q, k, v = [(torch.randn(n, H * dh, generator=g) * s).bfloat16() for s in (0.35, 1, 1)]
        junk = torch.full((64 << 20,), float("nan"), device="cuda"); del junk      # poison freed memory
        os.environ["PCY_ESM_ATTN"] = "exact"
        ex = ctx.attention(q.cuda(), k.cuda(), v.cuda(), lens, H, H, dh, False, 1.0).cpu().double()
        os.environ["PCY_ESM_ATTN"] = "fast"
        out = ctx.attention(q.cuda(), k.cuda(), v.cuda(), lens, H, H, dh, False, 1.0).cpu().double()
        ref = ref64(q, k, v, lens)
        bad = torch.isnan(out).any(1).nonzero().flatten().tolist()
        print(rep, lens[:3], "nan rows", bad[:10], len(bad), "err", float((torch.nan_to_num(out) - ref).norm() / ref.norm()), "exact err", float((ex - ref).norm() / ref.norm()))
