"""Randomised soak of `generate(method="beam")` on the small geometry: the replayed beam-step chain (pcy_llama_beam_steps) against the four calls
per step (PCY_DISABLE=beam_graph) for random beam sizes / groups / lengths / prompts: tokens, scores and the logits record must be equal."""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd import synthetic_model as SM
model = SM.build("small", device="cuda", max_new_tokens=64)
random.seed(1)
bad = runs = 0
t_end = time.time() + 90
while time.time() < t_end:
    nprot = random.choice((1, 2))
    prot = synth.protein_tokens([random.randint(20, 90) for _ in range(2)], seed=random.randint(0, 999))
    inp = SM.caption_inputs(model, prot, n_prompt_words=random.randint(5, 60), n_slots=nprot, seed=random.randint(0, 999))
    beam = random.choice((2, 3, 4, 5, 6, 8)); groups = [g for g in (1, 2, 3, 4) if beam % g == 0]; g = random.choice(groups)
    kw = dict(max_len=random.randint(2, 40), method="beam", beam_size=beam, beam_group_size=g, diversity_penalty=random.choice((0.0, 0.8)))
    import copy
    os.environ.pop("PCY_DISABLE", None)
    a = model.generate(copy.deepcopy(inp), **kw)
    os.environ["PCY_DISABLE"] = "beam_graph"
    b = model.generate(copy.deepcopy(inp), **kw)
    os.environ.pop("PCY_DISABLE", None)
    ok = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    runs += 1
    if not ok:
        bad += 1; print("MISMATCH", kw)
print(f"beam soak: {runs} generations, mismatches {bad}")
