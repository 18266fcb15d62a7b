import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from procyon_amd.engine import Context
ctx = Context.get()
BF = torch.bfloat16
g = torch.Generator().manual_seed(1)
H, dh, n = 20, 64, 300
q = (torch.randn(n, H*dh, generator=g) * dh ** -0.5).to(BF); k = torch.randn(n, H*dh, generator=g).to(BF)
v = torch.ones(n, H*dh).to(BF)
out = ctx.attention(q.cuda(), k.cuda(), v.cuda(), [n], H, H, dh, False, 1.0).cpu().float()
qs = q.view(n, H, dh).transpose(0, 1); ks = k.view(n, H, dh).transpose(0, 1)
s = torch.matmul(qs, ks.transpose(1, 2))
p = F.softmax(s, dim=-1, dtype=torch.float32).to(BF)
ref = p.float().sum(-1).transpose(0, 1)  # [n,H]
o = out.view(n, H, dh)[:, :, 0]
print("sum p: ref mean", ref.mean().item(), "gpu mean", o.mean().item(), "max abs diff", (o - ref.to(BF).float()).abs().max().item())
d = (o - ref)
print("diff quantiles", torch.quantile(d.flatten(), torch.tensor([0.01, 0.5, 0.99])))
print("rows 0..3", o[:4, 0], ref[:4, 0], "rows 296..299", o[296:, 0], ref[296:, 0])
bad = (d.abs() > 0.012).nonzero()
print("bad count", bad.shape[0], "rows", sorted(set(bad[:, 0].tolist()))[:40], "heads", sorted(set(bad[:, 1].tolist()))[:20])
for r, h in bad[:6].tolist():
    print(r, h, "gpu", o[r, h].item(), "ref", ref[r, h].item(), "ref max p", p[h, r].float().max().item(), "n_big", int((p[h, r].float() > 0.05).sum()))
