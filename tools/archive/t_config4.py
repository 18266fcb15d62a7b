"""config 4 (batch 32, mixed lengths): where the end-to-end time goes -- generate(max_len) for several max_len."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd import synthetic_model as SM
model = SM.build("full", device="cuda", max_new_tokens=512)
g = torch.Generator().manual_seed(7)
lens = torch.randint(256, 2049, (32,), generator=g).tolist()
prot = synth.protein_tokens(lens, seed=7)
words = lambda b: " ".join([f"w{(17 * b + 31 * i) % 50000}" for i in range(254)] + ["<|protein|>"] + [f"w{(13 * b + 7 * i) % 50000}" for i in range(254)]) + " [ANSWER]"
def inputs4():
    return {"data": {"seq": prot.clone(), "seq_idx": torch.arange(32), "text": [], "drug": None},
            "input": {"seq": [[b] for b in range(32)], "text": [[] for _ in range(32)], "drug": None},
            "target": {"seq": None, "text": None, "drug": None}, "instructions": [words(b) for b in range(32)]}
model.generate(inputs4(), max_len=4, method="greedy"); torch.cuda.synchronize()
for n in (2, 66, 258, 512, 512, 258):
    torch.cuda.synchronize(); t0 = time.perf_counter(); model.generate(inputs4(), max_len=n, method="greedy"); torch.cuda.synchronize()
    print(f"max_len {n:4d}: {time.perf_counter() - t0:.3f} s", flush=True)
t0 = time.perf_counter(); inp = inputs4(); emb, ids, mask, *_ = model._preprocessing(inp, crop_off=True, no_pad=True, left_pad=True); torch.cuda.synchronize()
print(f"preprocessing (ESM 49 chunks + projector + tokenise + splice): {time.perf_counter() - t0:.3f} s, prompt {tuple(emb.shape)}")
