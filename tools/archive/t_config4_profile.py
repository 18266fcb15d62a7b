"""Where the seconds of a BASELINE configs[3] call go (batch 32, mixed residues, 512-token prompts, 512 greedy tokens): wall time of
the call, the graph-replayed decode steps alone (HIP events), and a host-side cProfile of the rest."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synthetic_model as SM
from procyon_amd import workloads as W
model = SM.build("full", device="cuda", max_new_tokens=512)
make, lens, plen = W.config4_inputs(False, 32)
model.generate(make(), max_len=8, method="greedy"); torch.cuda.synchronize()
for n in (8, 512):
    t0 = time.perf_counter(); model.generate(make(), max_len=n, method="greedy"); torch.cuda.synchronize()
    print(f"generate max_len={n}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
pr = cProfile.Profile(); pr.enable(); model.generate(make(), max_len=8, method="greedy"); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
