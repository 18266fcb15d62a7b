"""Soak: repeated generation (greedy / beam / nucleus), QA forward and retrieval embedding on the full-geometry synthetic model;
checks that device memory does not grow, that no hand-over watchdog fires and that greedy output is reproducible."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd import synthetic_model as SM
from procyon_amd.engine import Context
model = SM.build("full", device="cuda", max_new_tokens=1100)
ctx = Context.get()
prot = synth.protein_tokens([1024, 300], seed=0)
def inp(seed=1):
    d = SM.caption_inputs(model, prot, n_prompt_words=250, n_slots=2, seed=seed)
    d["input"]["seq"] = [[0, 1]]
    return d
ref = None
mem = []
t0 = time.perf_counter()
for it in range(int(os.environ.get("ITERS", 6))):
    tok, *_ = model.generate(inp(), max_len=96, method="greedy")
    if ref is None: ref = tok.clone()
    assert torch.equal(tok, ref), "greedy generation not reproducible"
    model.generate(inp(), max_len=24, method="beam", beam_size=6, beam_group_size=3)
    torch.manual_seed(it); model.generate(inp(), max_len=16, method="nucleus", nucleus_prob=0.9)
    model.forward_sequences(synth.protein_tokens([1024] * 25, seed=it))
    ctx.sync()
    torch.cuda.synchronize()
    mem.append(torch.cuda.memory_allocated() >> 20)
long_tok, *_ = model.generate(inp(), max_len=1024, method="greedy")     # t up to 1276: second key group of the fused attention launch
ctx.sync()
print(f"soak ok in {time.perf_counter() - t0:.1f} s; allocated MiB per iteration {mem}; long generation {tuple(long_tok.shape)}")
assert max(mem[1:]) - min(mem[1:]) <= 64, mem
