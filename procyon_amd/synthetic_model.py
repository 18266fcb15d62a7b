"""Build a `UnifiedProCyon` of a given geometry with seeded random weights: the only
kind of model that exists on the build and GPU boxes (no checkpoints, no network)."""
from __future__ import annotations

import torch

from . import synth
from .engine import BF16, EsmConfig, LlamaConfig, MlpEngine
from .model import ESM_PLM, LlamaPostTokenization, ProCyonConfig, UnifiedProCyon
from .tokenizer import SyntheticTokenizer

GEOMETRIES = {
    # ProCyon-Full as BASELINE.json sizes it: ESM2-650M + Llama-3-8B (embedding rows = len(tokenizer) - 1)
    "full": dict(llama=dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336),
                 esm=dict(d=1280, n_layers=33, n_heads=20, ffn=5120), proj_hidden=2560, proj_layers=3),
    # ProCyon-Split (BASELINE configs[0]; /root/reference/README.md:50-51, procyon/training/training_args_IT.py:129-134): ESM2-150M + Llama-2-7B,
    # 32000 + 8 added tokens - [EXT]
    "split": dict(llama=dict(vocab=32007, d=4096, n_layers=32, n_heads=32, n_kv_heads=32, ffn=11008),
                  esm=dict(d=640, n_layers=30, n_heads=20, ffn=2560), proj_hidden=2560, proj_layers=3),
    # small geometry for tests / smoke (head_dim 64, same code paths)
    "small": dict(llama=dict(vocab=128263 - 128000 + 2048, d=256, n_layers=2, n_heads=4, n_kv_heads=2, ffn=512),
                  esm=dict(d=128, n_layers=2, n_heads=2, ffn=256), proj_hidden=192, proj_layers=3),
}


def build(geometry="full", device="cuda", pooling="mean", max_new_tokens=256, llama_layers=None, esm_layers=None,
          rope_theta=10000.0, return_weights=False, dtype=BF16, as_loaded=False):
    """dtype=torch.float32 generates fp32 weights -- an "fp32 checkpoint": the sub-modules keep them beside their bf16 engines until
    `.bfloat16()`, as a model loaded by the reference's `from_pretrained` holds fp32 parameters.  as_loaded=True returns the model
    WITHOUT the `.eval().bfloat16()` every shipped caller adds (what protpep_qa_scores.py:55-58 / caption_bulk.py:72-73 run)."""
    g = {k: (dict(v) if isinstance(v, dict) else v) for k, v in GEOMETRIES[geometry].items()}
    if llama_layers is not None:
        g["llama"]["n_layers"] = llama_layers
    if esm_layers is not None:
        g["esm"]["n_layers"] = esm_layers
    small = geometry == "small"
    if small:
        tok = SyntheticTokenizer(n_text=2000, base_vocab=2048 + 0, bos_token_id=2040, eos_token_id=2041)
    elif geometry == "split":
        tok = SyntheticTokenizer(n_text=30000, base_vocab=32000, bos_token_id=1, eos_token_id=2)     # Llama-2 vocabulary size and bos / eos ids
    else:
        tok = SyntheticTokenizer()
    if small:
        g["llama"]["vocab"] = len(tok) - 1
    assert g["llama"]["vocab"] == len(tok) - 1  # model_unified.py:166: [EXT] is never embedded
    gen_dev = device if not return_weights else "cpu"
    lsd = synth.llama_state_dict(**g["llama"], device=gen_dev, dtype=dtype)
    esd = synth.esm_state_dict(**g["esm"], device=gen_dev, dtype=dtype)
    D, d = g["esm"]["d"], g["llama"]["d"]
    projs = {"aaseq": synth.mlp_layers(g["proj_layers"], D, d, g["proj_hidden"], 0, device=gen_dev, dtype=dtype),
             "shared": synth.mlp_layers(g["proj_layers"], D, D, g["proj_hidden"], 20, device=gen_dev, dtype=dtype),
             "lm": synth.mlp_layers(g["proj_layers"], d, D, g["proj_hidden"], 40, device=gen_dev, dtype=dtype)}
    dev = torch.device(device)
    text_encoder = LlamaPostTokenization(lsd, LlamaConfig(**g["llama"], rope_theta=rope_theta, max_pos=4096), dev, max_new_tokens)
    plm = ESM_PLM(esd, EsmConfig(**g["esm"]), pooling_method=pooling, device=dev)
    def mk(layers):
        m = MlpEngine([(w.to(dev, BF16), None if b is None else b.to(dev, BF16)) for w, b in layers])
        if dtype == torch.float32:
            m.src_f32 = list(layers)
        return m
    cfg = ProCyonConfig(protein_pooling_opt=pooling, use_aaseq_embeddings=False)
    model = UnifiedProCyon(cfg, text_encoder, tok, protein_seq_encoder=plm, token_projectors={"aaseq": mk(projs["aaseq"])},
                           aaseq_shared_projector=mk(projs["shared"]), aaseq_lm_projector=mk(projs["lm"]))
    if not as_loaded:
        model.eval().bfloat16()      # what every shipped caller of the reference does after construction (procyon.py:64-65)
    if return_weights:
        return model, dict(llama=lsd, esm=esd, projs=projs, geom=g)
    return model


def caption_inputs(model, protein_tokens, n_prompt_words=500, n_slots=2, seed=0):
    """The nested `inputs` dict of the reference's collators (it_collator.py:1467-1501) for one
    caption-style prompt with `n_slots` <|protein|> slots and a trailing [ANSWER]."""
    g = torch.Generator().manual_seed(seed)
    words = [f"w{int(i)}" for i in torch.randint(0, 50000, (n_prompt_words,), generator=g)]
    cut = sorted(torch.randperm(n_prompt_words - 2, generator=g)[:n_slots].tolist())
    for c in cut:
        words[c] = "<|protein|>"
    instr = " ".join(words) + " [ANSWER]"
    B = 1
    return {"data": {"seq": protein_tokens, "seq_idx": torch.arange(protein_tokens.shape[0]), "text": [], "drug": None},
            "input": {"seq": [[0] * n_slots for _ in range(B)], "text": [[] for _ in range(B)], "drug": None},
            "target": {"seq": None, "text": None, "drug": None},
            "instructions": [instr for _ in range(B)]}
