"""Where does the fp8 (e4m3, W8A8 per-token / per-channel) prefill path leave the bf16 path?  Walks the 32 layers of the seeded
Llama-3-8B-geometry model and reports err(fp8, bf16) of the RESIDUAL STREAM after every layer (pcy_llama_prefill_all), next to the
CPU oracle's fp8-vs-bf16 distance for the first layers (same seeds) and to the random-walk model e_1 * sqrt(l).

    python tools/fp8_layer_walk.py [--layers 32] [--oracle-layers 4] [--T 64] [--json out.json]

Reading: if the curve follows e_1 * sqrt(l) (independent ~equal kicks per layer into an undamped random-init residual stream), the
0.66 distance of the answer-row logits after 32 layers is quantisation noise of a W8A8 path, not a defect of a kernel; a kernel
defect shows as a jump at one layer or as a GPU curve above the oracle's."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def rel(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / b.norm())


def walk(n_layers=32, oracle_layers=4, T=64, B=2, vocab=512, seed=11, verbose=True):
    from oracle import llama_ref as LR
    from procyon_amd import synth
    from procyon_amd.engine import LlamaConfig, LlamaEngine
    kw = dict(vocab=vocab, d=4096, n_heads=32, n_kv_heads=8, ffn=14336)
    g = torch.Generator().manual_seed(seed)
    emb = (torch.randn(B, T, 4096, generator=g) * 0.02).bfloat16()
    # weights seeded on the CPU (the device generator draws other numbers): the oracle's first layers are then the SAME tensors
    eng = LlamaEngine(synth.llama_state_dict(**kw, n_layers=n_layers, workers=16), LlamaConfig(**kw, n_layers=n_layers, max_pos=512),
                      free_source=True, fp8_prefill=True)
    eng.set_fp8(False)
    lg16, h16 = eng.prefill_all(emb.cuda(), None, eng.new_cache(B, T), "last")
    eng.set_fp8(True)
    lg8, h8 = eng.prefill_all(emb.cuda(), None, eng.new_cache(B, T), "last")
    gpu = [rel(h8[l], h16[l]) for l in range(1, n_layers)]          # residual stream after layers 0..L-2 (entry L is final-normed)
    gpu_final, gpu_logits = rel(h8[n_layers], h16[n_layers]), rel(lg8, lg16)
    orc = []
    if oracle_layers > 0:
        sd = synth.llama_state_dict(**kw, n_layers=oracle_layers + 1)     # the first layers of the same seeded model
        r16 = LR.llama_forward(sd, LR.LlamaGeom(**kw, n_layers=oracle_layers + 1), inputs_embeds=emb, attn_mask=torch.ones(B, T), want_hidden=True)
        r8 = LR.llama_forward(sd, LR.LlamaGeom(**kw, n_layers=oracle_layers + 1, weights="fp8"), inputs_embeds=emb, attn_mask=torch.ones(B, T), want_hidden=True)
        orc = [rel(r8["hidden_states"][l], r16["hidden_states"][l]) for l in range(1, oracle_layers + 1)]
        gpu_vs_orc8 = [rel(h8[l].cpu(), r8["hidden_states"][l]) for l in range(1, oracle_layers + 1)]
    else:
        gpu_vs_orc8 = []
    e1 = gpu[0]
    rows = []
    for i, e in enumerate(gpu):
        l = i + 1
        rows.append(dict(after_layer=l, gpu_fp8_vs_bf16=e, sqrt_model=e1 * l ** 0.5, oracle_fp8_vs_bf16=orc[i] if i < len(orc) else None,
                         gpu_fp8_vs_oracle_fp8=gpu_vs_orc8[i] if i < len(gpu_vs_orc8) else None))
        if verbose:
            o = f"  oracle fp8-vs-bf16 {orc[i]:.3e}  gpu-fp8 vs oracle-fp8 {gpu_vs_orc8[i]:.3e}" if i < len(orc) else ""
            print(f"after layer {l:2d}: err(fp8, bf16) {e:.3e}   e1*sqrt(l) {e1 * l ** 0.5:.3e}   ratio {e / (e1 * l ** 0.5):.2f}{o}", flush=True)
    out = dict(rows=rows, final_normed=gpu_final, last_row_logits=gpu_logits, T=T, B=B, n_layers=n_layers)
    if verbose:
        print(f"final-normed state: {gpu_final:.3e}   last-row logits: {gpu_logits:.3e}")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--oracle-layers", type=int, default=4)
    ap.add_argument("--T", type=int, default=64)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = walk(a.layers, a.oracle_layers, a.T)
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)
