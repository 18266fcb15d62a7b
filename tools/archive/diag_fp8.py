"""fp8 prefill vs the fp8 oracle on a small Llama: per-layer-count and per-row errors (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import llama_ref as LR
from procyon_amd import synth
from procyon_amd.engine import LlamaConfig, LlamaEngine
BF = torch.bfloat16
def rel(a, b): a, b = a.float(), b.float(); return ((a - b).norm() / b.norm()).item()
for nl in (1, 2):
    for masked in (False, True):
        kw = dict(vocab=320, d=256, n_layers=nl, n_heads=4, n_kv_heads=2, ffn=512)
        sd = synth.llama_state_dict(**kw)
        eng = LlamaEngine({k: v.clone() for k, v in sd.items()}, LlamaConfig(**kw, max_pos=512), fp8_prefill=True)
        torch.manual_seed(7)
        B, T = 3, 37
        emb = (torch.randn(B, T, 256) * 0.02).to(BF)
        mask = torch.ones(B, T)
        if masked: mask[1, :5] = 0
        ref = LR.llama_forward(sd, LR.LlamaGeom(**kw, weights="fp8"), inputs_embeds=emb, attn_mask=mask, want_hidden=True)
        ref16 = LR.llama_forward(sd, LR.LlamaGeom(**kw), inputs_embeds=emb, attn_mask=mask, want_hidden=True)
        cache = eng.new_cache(B, T)
        _, hid = eng.prefill(emb.cuda(), mask, cache, "last", want_hidden=True)
        hid = hid.cpu()
        print(f"layers {nl} masked {masked}: hidden vs fp8 oracle {rel(hid, ref['last_hidden']):.2e}  vs bf16 oracle {rel(hid, ref16['last_hidden']):.2e}  "
              f"(oracle fp8 vs bf16 {rel(ref['last_hidden'], ref16['last_hidden']):.2e})")
        rows = [(b, t, rel(hid[b, t], ref['last_hidden'][b, t])) for b in range(B) for t in range(T)]
        rows.sort(key=lambda r: -r[2])
        print("   worst rows", [(b, t, f"{e:.1e}") for b, t, e in rows[:5]], " median", f"{sorted(r[2] for r in rows)[len(rows)//2]:.1e}")
