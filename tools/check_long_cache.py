"""One-row fused decode steps (Llama-3-8B and ProCyon-Split geometry, 2 layers) at cache lengths 1900 / 3300 / 6000: step, replayed step and
one launch per layer against the launch-per-stage run, bit for bit."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
BF = torch.bfloat16
for name, geo in (("split", dict(d=4096, n_heads=32, n_kv_heads=32, ffn=11008)), ("llama3", dict(d=4096, n_heads=32, n_kv_heads=8, ffn=14336))):
    kw = dict(vocab=4096, n_layers=2, **geo)
    eng = LlamaEngine(synth.llama_state_dict(**kw), LlamaConfig(**kw, max_pos=8192))
    for T, N in ((1900, 4), (3300, 3), (6000, 3)):
        torch.manual_seed(T)
        emb = (torch.randn(1, T, 4096) * 0.02).to(BF).cuda()
        def run(dis, g):
            if dis: os.environ["PCY_DISABLE"] = dis
            else: os.environ.pop("PCY_DISABLE", None)
            cache = eng.new_cache(1, T + N + 2)
            st = GenState(1, kw["vocab"], N + 2, "cuda")
            logits, _ = eng.prefill(emb, None, cache, "last")
            st.logits.copy_(logits); st.pos.fill_(T)
            eng.pick(cache, st, 1, advance_pos=False)
            out = []
            for _ in range(N):
                eng.greedy_steps(cache, st, 1, 1, use_graph=g)
                out.append(st.logits.clone())
            Context.get().sync()
            return torch.stack(out).cpu(), st.tokens_out[:, :N + 1].cpu()
        ref = run("decode_step,decode_layer,attn_o,mlp_chain", False)
        res = [all(torch.equal(x, y) for x, y in zip(run(d, g), ref)) for d, g in (("", False), ("", True), ("decode_step", True))]
        print(name, T, res, bool(torch.isfinite(ref[0].float()).all()), flush=True)
    del eng
    torch.cuda.empty_cache()
