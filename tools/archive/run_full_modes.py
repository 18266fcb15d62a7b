"""Full-geometry sanity run of the generation modes the bench does not time: diverse beam search, nucleus sampling, QA
forward and retrieval forward on the synthetic ProCyon-Full model; prints wall times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd import synthetic_model as SM

model = SM.build("full", device="cuda", max_new_tokens=64)
prot = synth.protein_tokens([1024, 300], seed=0)
inp = SM.caption_inputs(model, prot, n_prompt_words=250, n_slots=2, seed=1)
inp["input"]["seq"] = [[0, 1]]
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print(f"{name}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    return out
model.generate(inp, max_len=3, method="beam", beam_size=10, beam_group_size=2)   # warm-up: workspace growth, first launches
tok, sc, lg, txt = timed("beam 10/2 x 24 tokens", lambda: model.generate(inp, max_len=24, method="beam", beam_size=10, beam_group_size=2))
timed("beam 10/2 x 24 tokens (again)", lambda: model.generate(inp, max_len=24, method="beam", beam_size=10, beam_group_size=2))
print("  tokens", tuple(tok.shape), "scores", sc[0, :4].tolist())
torch.manual_seed(0)
model.generate(inp, max_len=3, method="nucleus", nucleus_prob=0.9)
tok, lp, lg, txt = timed("nucleus x 24 tokens", lambda: model.generate(inp, max_len=24, method="nucleus", nucleus_prob=0.9))
timed("nucleus x 24 tokens (again)", lambda: model.generate(inp, max_len=24, method="nucleus", nucleus_prob=0.9))
print("  tokens", tuple(tok.shape))
tok, lp, lg, txt = timed("greedy x 24 tokens", lambda: model.generate(inp, max_len=24, method="greedy"))
z = timed("forward_sequences", lambda: model.forward_sequences(prot, get_soft_tokens=True))
print("  shared", tuple(z["shared"].shape), "finite", bool(torch.isfinite(z["shared"].float()).all()))
