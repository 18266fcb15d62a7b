"""Checkpoint ingest: map a ProCyon state dict (`txllm_model_ckpt.pt` or the fp32 dict merged from
DeepSpeed ZeRO shards, /root/reference/procyon/model/model_unified.py:1371-1382) onto the engine.

Key layout of the reference's `UnifiedProCyon.state_dict()`:
  text_encoder.model.<HF LlamaForCausalLM keys>          (pmc_llama.py:478-487: self.model = LlamaForCausalLM)
  protein_seq_encoder.model.<fair-esm ESM2 keys | HF Esm keys>   (esm.py:378-420)
  token_projectors.{aaseq,prot_structure,drug}.<i>.{weight,bias}   create_mlp Sequential: Linear at 0, 3, 6, ...
  aaseq_shared_projector.<i>.*, aaseq_lm_projector.<i>.*
  {protein_seq,domain,peptide,protein_struct,drug_structure}_embeddings.weight
No checkpoint exists on the build / GPU boxes; the key mapping is unit-tested on synthetic dicts (tests/test_host_cpu.py).
"""
from __future__ import annotations

import re

import torch

_FAIR2HF = [
    (r"^embed_tokens\.weight$", "esm.embeddings.word_embeddings.weight"),
    (r"^emb_layer_norm_after\.(weight|bias)$", r"esm.encoder.emb_layer_norm_after.\1"),
    (r"^layers\.(\d+)\.self_attn\.q_proj\.(weight|bias)$", r"esm.encoder.layer.\1.attention.self.query.\2"),
    (r"^layers\.(\d+)\.self_attn\.k_proj\.(weight|bias)$", r"esm.encoder.layer.\1.attention.self.key.\2"),
    (r"^layers\.(\d+)\.self_attn\.v_proj\.(weight|bias)$", r"esm.encoder.layer.\1.attention.self.value.\2"),
    (r"^layers\.(\d+)\.self_attn\.out_proj\.(weight|bias)$", r"esm.encoder.layer.\1.attention.output.dense.\2"),
    (r"^layers\.(\d+)\.self_attn_layer_norm\.(weight|bias)$", r"esm.encoder.layer.\1.attention.LayerNorm.\2"),
    (r"^layers\.(\d+)\.fc1\.(weight|bias)$", r"esm.encoder.layer.\1.intermediate.dense.\2"),
    (r"^layers\.(\d+)\.fc2\.(weight|bias)$", r"esm.encoder.layer.\1.output.dense.\2"),
    (r"^layers\.(\d+)\.final_layer_norm\.(weight|bias)$", r"esm.encoder.layer.\1.LayerNorm.\2"),
]


def fair_esm_to_hf(sd):
    """fair-esm ESM2 parameter names -> the HF Esm names the engine consumes.  Unused tensors (lm_head, contact head,
    rotary inv_freq buffers) are dropped.  Already-HF dicts pass through."""
    out = {}
    for k, v in sd.items():
        if k.startswith("esm."):
            out[k] = v
            continue
        for pat, rep in _FAIR2HF:
            nk, n = re.subn(pat, rep, k)
            if n:
                out[nk] = v
                break
    return out


def split_state_dict(sd):
    """ProCyon state dict -> dict(llama=HF-named, esm=HF-named or None, projectors={name: [(W,b|None),...]},
    tables={name: tensor})."""
    llama = {k[len("text_encoder.model."):]: v for k, v in sd.items() if k.startswith("text_encoder.model.")}
    esm_raw = {k[len("protein_seq_encoder.model."):]: v for k, v in sd.items() if k.startswith("protein_seq_encoder.model.")}
    esm = fair_esm_to_hf(esm_raw) if esm_raw else None

    def mlp(prefix):
        idx = sorted({int(m.group(1)) for k in sd for m in [re.match(re.escape(prefix) + r"\.(\d+)\.weight$", k)] if m})
        return [(sd[f"{prefix}.{i}.weight"], sd.get(f"{prefix}.{i}.bias")) for i in idx]

    projectors = {}
    for name in ("aaseq", "prot_structure", "drug"):
        layers = mlp(f"token_projectors.{name}")
        if layers:
            projectors["token_" + name] = layers
    for name in ("aaseq_shared_projector", "aaseq_lm_projector"):
        layers = mlp(name)
        if layers:
            projectors[name] = layers
    tables = {k[:-len(".weight")]: v for k, v in sd.items() if re.match(r"^(protein_seq|domain|peptide|protein_struct|drug_structure)_embeddings\.weight$", k)}
    return dict(llama=llama, esm=esm, projectors=projectors, tables=tables)


def infer_llama_config(llama_sd, **over):
    from .engine import LlamaConfig
    emb = llama_sd["model.embed_tokens.weight"]
    n_layers = 1 + max(int(m.group(1)) for k in llama_sd for m in [re.match(r"model\.layers\.(\d+)\.", k)] if m)
    d = emb.shape[1]
    kv = llama_sd["model.layers.0.self_attn.k_proj.weight"].shape[0]
    ffn = llama_sd["model.layers.0.mlp.gate_proj.weight"].shape[0]
    head_dim = over.pop("head_dim", 128)
    return LlamaConfig(vocab=emb.shape[0], d=d, n_layers=n_layers, n_heads=d // head_dim, n_kv_heads=kv // head_dim, ffn=ffn, **over)


def infer_esm_config(esm_sd, n_heads=None, **over):
    from .engine import EsmConfig
    emb = esm_sd["esm.embeddings.word_embeddings.weight"]
    n_layers = 1 + max(int(m.group(1)) for k in esm_sd for m in [re.match(r"esm\.encoder\.layer\.(\d+)\.", k)] if m)
    d = emb.shape[1]
    ffn = esm_sd["esm.encoder.layer.0.intermediate.dense.weight"].shape[0]
    if n_heads is None:
        n_heads = 40 if d >= 2560 else 20        # ESM2 family
    return EsmConfig(d=d, n_layers=n_layers, n_heads=n_heads, ffn=ffn, vocab=emb.shape[0], **over)


def build_model(sd, config, tokenizer, device="cuda", max_new_tokens=256, esm_heads=None, **llama_over):
    """state dict + `ProCyonConfig` + tokenizer -> engine-backed `UnifiedProCyon` (the tail of `from_pretrained`,
    model_unified.py:1370-1382 `load_state_dict(strict=False)`)."""
    from .engine import BF16, MlpEngine
    from .model import ESM_PLM, LlamaPostTokenization, UnifiedProCyon
    parts = split_state_dict(sd)
    dev = torch.device(device)
    text_encoder = LlamaPostTokenization(parts["llama"], infer_llama_config(parts["llama"], **llama_over), dev, max_new_tokens)
    plm = None
    if parts["esm"] and not config.use_aaseq_embeddings:
        plm = ESM_PLM(parts["esm"], infer_esm_config(parts["esm"], n_heads=esm_heads), pooling_method=config.protein_pooling_opt,
                      protein_pooling_correction_option=config.protein_pooling_correction_option,
                      max_protein_len=config.max_protein_len, device=dev)
    mk = lambda layers: MlpEngine([(w.to(dev, BF16), None if b is None else b.to(dev, BF16)) for w, b in layers])
    P = parts["projectors"]
    tok_proj = {n[len("token_"):]: mk(l) for n, l in P.items() if n.startswith("token_")}
    tabs = {k: v.to(dev, BF16) for k, v in parts["tables"].items()}
    return UnifiedProCyon(config, text_encoder, tokenizer, protein_seq_encoder=plm, token_projectors=tok_proj,
                          aaseq_shared_projector=mk(P["aaseq_shared_projector"]) if "aaseq_shared_projector" in P else None,
                          aaseq_lm_projector=mk(P["aaseq_lm_projector"]) if "aaseq_lm_projector" in P else None,
                          protein_seq_embeddings=tabs.get("protein_seq_embeddings"), domain_embeddings=tabs.get("domain_embeddings"),
                          peptide_embeddings=tabs.get("peptide_embeddings"), protein_struct_embeddings=tabs.get("protein_struct_embeddings"),
                          drug_structure_embeddings=tabs.get("drug_structure_embeddings"))
