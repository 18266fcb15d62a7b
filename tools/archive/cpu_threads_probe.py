import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import llama_ref as LR
from procyon_amd import synth
lk = dict(vocab=1024, d=4096, n_layers=1, n_heads=32, n_kv_heads=8, ffn=14336)
for dt in (torch.bfloat16, torch.float32):
    lsd = {k: v.to(dt) for k, v in synth.llama_state_dict(**lk).items()}
    geom = LR.LlamaGeom(**lk)
    for nt in (8, 32, 64, 128, 256):
        torch.set_num_threads(nt)
        emb = (torch.randn(1, 512, 4096) * 0.02).to(dt)
        t0 = time.perf_counter(); r = LR.llama_forward(lsd, geom, inputs_embeds=emb, attn_mask=torch.ones(1, 512), logits_rows="last"); tp = time.perf_counter() - t0
        tok = r["logits"][:, -1].argmax(-1, keepdim=True); past = r["past_kv"]
        t0 = time.perf_counter()
        for _ in range(3):
            r = LR.llama_forward(lsd, geom, input_ids=tok, attn_mask=None, past_kv=past, logits_rows="last"); past = r["past_kv"]
        td = (time.perf_counter() - t0) / 3
        print(dt, "threads", nt, f"prefill/layer {tp*1e3:.0f} ms  decode/layer {td*1e3:.1f} ms", flush=True)
