"""Diverse beam search at full geometry: slope (ms per step) and intercept of generate(method="beam") over max_len."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd import synthetic_model as SM
model = SM.build("full", device="cuda", max_new_tokens=128)
prot = synth.protein_tokens([1024, 300], seed=0)
inp = SM.caption_inputs(model, prot, n_prompt_words=250, n_slots=2, seed=1)
inp["input"]["seq"] = [[0, 1]]
beam, g = int(os.environ.get("BEAM", 10)), int(os.environ.get("GROUP", 2))
model.generate(inp, max_len=3, method="beam", beam_size=beam, beam_group_size=g)
for n in (8, 8, 24, 56, 120):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.generate(inp, max_len=n, method="beam", beam_size=beam, beam_group_size=g)
    torch.cuda.synchronize(); print(f"beam {beam}/{g} max_len {n:3d}: {(time.perf_counter() - t0) * 1e3:7.1f} ms", flush=True)
