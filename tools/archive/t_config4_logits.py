"""configs[3] call (batch 32, 512 + 512 tokens): cost of the per-step logits record (pinned host memory written by the decode steps vs a
device buffer copied at the end), second calls included (allocator reuse)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synthetic_model as SM
from procyon_amd import workloads as W
from procyon_amd.engine import Context, GenState
model = SM.build("full", device="cuda", max_new_tokens=512)
make, lens, plen = W.config4_inputs(False, 32)
model.generate(make(), max_len=8, method="greedy"); torch.cuda.synchronize()
for host in ("1", "0", "1", "0"):
    os.environ["PCY_LOGITS_HOST"] = host
    t0 = time.perf_counter(); out = model.generate(make(), max_len=512, method="greedy"); torch.cuda.synchronize()
    print(f"PCY_LOGITS_HOST={host}: generate 512 = {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    del out
# the decode steps alone, with and without a record
eng = model.text_encoder.engine; cfg = eng.cfg; ctx = Context.get(eng.device)
emb = (torch.randn(32, 512, cfg.d) * 0.02).to(torch.bfloat16).to(eng.device)
for keep, host in ((False, "1"), (True, "1"), (True, "0")):
    os.environ["PCY_LOGITS_HOST"] = host
    cache = eng.new_cache(32, 1024)
    t0 = time.perf_counter(); st = GenState(32, cfg.vocab, 512, eng.device, keep, None); t_alloc = time.perf_counter() - t0
    logits, _ = eng.prefill(emb, None, cache, "last")
    st.logits.copy_(logits); st.pos.fill_(512)
    eng.pick(cache, st, 32, advance_pos=False)
    eng.greedy_steps(cache, st, 32, 8)
    ctx.timer_start(); eng.greedy_steps(cache, st, 32, 496); ms = ctx.timer_stop() / 496
    print(f"keep_logits={keep} host={host}: GenState alloc {t_alloc * 1e3:.1f} ms, {ms:.3f} ms per step", flush=True)
    del st, cache
