import torch, torch.nn.functional as F
BF = torch.bfloat16
def ulps(a, b):
    ai = a.view(torch.int16).int(); bi = b.view(torch.int16).int()
    ai = torch.where(ai < 0, -(ai & 0x7fff), ai); bi = torch.where(bi < 0, -(bi & 0x7fff), bi)
    return (ai - bi).abs()
g = torch.Generator().manual_seed(0)
for (B,H,S,dh) in [(2,2,70,64),(1,20,70,64),(1,4,45,64),(1,32,24,128)]:
    q = (torch.randn(B,H,S,dh,generator=g)).to(BF); k = torch.randn(B,H,S,dh,generator=g).to(BF); v = torch.randn(B,H,S,dh,generator=g).to(BF)
    s_cpu = torch.matmul(q, k.transpose(2,3))
    s_ex = torch.matmul(q.double(), k.double().transpose(2,3)).to(BF)
    u = ulps(s_cpu, s_ex); print((B,H,S,dh), 'QK^T cpu vs exact mism', int((u>0).sum()), '/', u.numel(), 'max', int(u.max()))
    p = F.softmax(s_ex.float()*0.1, -1).to(BF)
    o_cpu = torch.matmul(p, v); o_ex = torch.matmul(p.double(), v.double()).to(BF)
    u = ulps(o_cpu, o_ex); print('   PV cpu vs exact mism', int((u>0).sum()), '/', u.numel(), 'max', int(u.max()))
    x = torch.randn(B*S, H*dh, generator=g).to(BF); w = (torch.randn(H*dh, H*dh, generator=g)*0.05).to(BF); b = torch.randn(H*dh, generator=g).to(BF)
    y = F.linear(x.view(B,S,-1), w, b); ye = (x.double()@w.double().T + b.double()).to(BF).view(B,S,-1)
    u = ulps(y, ye); print('   linear3d cpu vs exact mism', int((u>0).sum()), '/', u.numel(), 'max', int(u.max()))
