import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from procyon_amd import synth, _lib as L
from procyon_amd.engine import Context, EsmConfig, EsmEngine
kw = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
eng = EsmEngine(synth.esm_state_dict(**kw, device="cuda"), EsmConfig(**kw))
ctx = Context.get()
toks = synth.protein_tokens([1024], seed=1)
c0 = ctx.lib.pcy_debug_dispatch_count(L.DISPATCH_ESM_GRAPH)
for i in range(5):
    eng.forward(toks); torch.cuda.synchronize()
    print(i, ctx.lib.pcy_debug_dispatch_count(L.DISPATCH_ESM_GRAPH) - c0, ctx.lib.pcy_last_error())
