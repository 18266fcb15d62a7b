"""Host time against GPU time of the retrieval leg's batches: does the host run ahead of the GPU (asynchronous enqueue) or in step with it?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd import synthetic_model as SM
model = SM.build("full", device="cuda", llama_layers=1)
rb = model.protein_seq_encoder.engine.preferred_batch(1026)
tok_fn = lambda idx: synth.protein_tokens([1024] * len(idx), seed=1000 + idx[0])
model.forward_sequences(tok_fn(list(range(rb))))["shared"]; torch.cuda.synchronize()
for rep in range(2):
    t_tok = t_fwd = 0.0
    t0 = time.perf_counter()
    outs = []
    for s in range(0, 250, rb):
        a = time.perf_counter(); toks = tok_fn(list(range(s, s + rb))); b = time.perf_counter()
        outs.append(model.forward_sequences(toks)["shared"]); c = time.perf_counter()
        t_tok += b - a; t_fwd += c - b
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"10 batches of {rb}: host token synthesis {t_tok * 100:.2f} ms/batch, forward_sequences call {t_fwd * 100:.2f} ms/batch, loop returned after {t_enq * 1e3:.1f} ms, GPU done after {t_all * 1e3:.1f} ms")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    toks = tok_fn(list(range(rb)))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): model.forward_sequences(toks)["shared"]
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
