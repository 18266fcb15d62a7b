import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from procyon_amd.engine import Context
ctx = Context.get()
BF = torch.bfloat16
def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.randn(*shape, generator=g) * std).to(BF)
H, dh = 4, 64
lens = [int(x) for x in os.environ.get("LENS", "1026,191,192,193").split(",")]
n = sum(lens)
q, k, v = rnd(n, H * dh, seed=1), rnd(n, H * dh, seed=2), rnd(n, H * dh, seed=3)
q = (q.float() * dh ** -0.5).to(BF)
outs, t0 = [], 0
for L in lens:
    qs = q[t0:t0+L].view(L, H, dh).transpose(0, 1); ks = k[t0:t0+L].view(L, H, dh).transpose(0, 1); vs = v[t0:t0+L].view(L, H, dh).transpose(0, 1)
    s = torch.matmul(qs, ks.transpose(1, 2)); p = F.softmax(s, dim=-1, dtype=torch.float32).to(BF)
    outs.append(torch.matmul(p, vs).transpose(0, 1).reshape(L, H * dh)); t0 += L
ref = torch.cat(outs)
out = ctx.attention(q.cuda(), k.cuda(), v.cuda(), lens, H, H, dh, False, 1.0).cpu()
err = (out.float() - ref.float()).abs().view(n, H, dh).amax(-1)   # [n, H]
scale = ref.float().abs().view(n, H, dh).amax(-1)
bad = (err > 0.02 * scale + 2e-3).nonzero()
print("total rel err", float((out.float()-ref.float()).norm()/ref.float().norm()), "bad (row,head) pairs:", bad.shape[0])
t0 = 0
for si, L in enumerate(lens):
    rows = sorted(set(r - t0 for r, h in bad.tolist() if t0 <= r < t0 + L))
    print(f"seq {si} len {L}: bad rows {rows[:40]}{'...' if len(rows) > 40 else ''} (count {len(rows)})")
    t0 += L
