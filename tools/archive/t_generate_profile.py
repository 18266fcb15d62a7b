"""Host-side profile of one UnifiedProCyon.generate call (bench workload): where the milliseconds outside the kernels go."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd import synthetic_model as SM
model = SM.build("full", device="cuda", max_new_tokens=256)
prot = synth.protein_tokens([1024], seed=0)
def step():
    inputs = SM.caption_inputs(model, prot, n_prompt_words=510, n_slots=2, seed=0)
    return model.generate(inputs, max_len=256, method="greedy")
step(); step(); torch.cuda.synchronize()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print(f"one step: {(time.perf_counter() - t0) * 1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
