"""Seeded synthetic weights and inputs (SURVEY.md section 8d "Synthetic weights").

There are no checkpoints, tokenizer files or datasets on the build or GPU boxes, so every
test and benchmark runs on random-init weights of the reference's architecture:
per-tensor `torch.Generator(seed = 1234 + tensor_index)`; Linear / embedding ~ N(0, 0.02^2),
norm gains = 1 + N(0, 0.02^2), biases N(0, 0.02^2); generated fp32 then cast once.
State-dict keys follow the HuggingFace names the reference's checkpoints use under
`text_encoder.model.*` / `protein_seq_encoder.model.*` (model_unified.py:1376-1382).
"""
from __future__ import annotations

import torch

BASE_SEED = 1234


def _gen(shape, idx, std=0.02, mean=0.0, device="cpu", dtype=torch.bfloat16):
    g = torch.Generator(device=device)
    g.manual_seed(BASE_SEED + idx)
    t = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std + mean
    return t.to(dtype)


def _materialise(todo, device, dtype, workers):
    """todo = [(name, shape, idx, kw)] -> {name: tensor}.  Every tensor has its own seeded generator, so the values do not
    depend on the order or the number of worker threads (torch.randn releases the GIL; 8 B parameters: ~110 s -> ~15 s)."""
    if workers and workers > 1 and str(device) == "cpu":
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            ts = list(ex.map(lambda t: _gen(t[1], t[2], device=device, dtype=dtype, **t[3]), todo))
    else:
        ts = [_gen(shape, idx, device=device, dtype=dtype, **kw) for _, shape, idx, kw in todo]
    return {t[0]: v for t, v in zip(todo, ts)}


def llama_state_dict(vocab, d, n_layers, n_heads, n_kv_heads, ffn, dtype=torch.bfloat16, device="cpu", workers=0):
    dh = d // n_heads
    todo = []

    def add(name, shape, **kw):
        todo.append((name, shape, len(todo), kw))

    add("model.embed_tokens.weight", (vocab, d))
    for l in range(n_layers):
        p = f"model.layers.{l}."
        add(p + "self_attn.q_proj.weight", (n_heads * dh, d))
        add(p + "self_attn.k_proj.weight", (n_kv_heads * dh, d))
        add(p + "self_attn.v_proj.weight", (n_kv_heads * dh, d))
        add(p + "self_attn.o_proj.weight", (d, n_heads * dh))
        add(p + "mlp.gate_proj.weight", (ffn, d))
        add(p + "mlp.up_proj.weight", (ffn, d))
        add(p + "mlp.down_proj.weight", (d, ffn))
        add(p + "input_layernorm.weight", (d,), mean=1.0)
        add(p + "post_attention_layernorm.weight", (d,), mean=1.0)
    add("model.norm.weight", (d,), mean=1.0)
    add("lm_head.weight", (vocab, d))
    return _materialise(todo, device, dtype, workers)


def damp_residual_branches(sd, scale=0.25):
    """Multiply every residual BRANCH's last projection (Llama: o_proj, down_proj; ESM: attention.output.dense, output.dense -- weights
    and biases) by `scale` IN PLACE and return sd.  With N(0, 0.02^2) weights every branch is as large as the stream it is added to,
    and a 32-layer stack amplifies a 2^-9 rounding into ~8 % of the logits (tests/test_gpu_fulldepth.py): any two bf16 arithmetics then
    disagree on a fifth of the argmaxes.  Trained models keep their branches a fraction of the stream; scale 0.25 (a power of two: exact
    in bf16, so the damped weights are the same bf16 values on every side) puts the synthetic model into that regime, where token
    agreement between two implementations is a meaningful number."""
    for k, v in sd.items():
        if k.endswith(("self_attn.o_proj.weight", "mlp.down_proj.weight", "attention.output.dense.weight", "attention.output.dense.bias")) or \
                (".output.dense." in k and ".attention." not in k):
            v.mul_(scale)
    return sd


def esm_state_dict(d, n_layers, n_heads, ffn, vocab=33, dtype=torch.bfloat16, device="cpu", workers=0):
    todo = []

    def add(name, shape, **kw):
        todo.append((name, shape, 100000 + len(todo), kw))

    add("esm.embeddings.word_embeddings.weight", (vocab, d))
    for l in range(n_layers):
        p = f"esm.encoder.layer.{l}."
        for nm in ("query", "key", "value"):
            add(p + f"attention.self.{nm}.weight", (d, d))
            add(p + f"attention.self.{nm}.bias", (d,))
        add(p + "attention.output.dense.weight", (d, d))
        add(p + "attention.output.dense.bias", (d,))
        add(p + "attention.LayerNorm.weight", (d,), mean=1.0)
        add(p + "attention.LayerNorm.bias", (d,))
        add(p + "intermediate.dense.weight", (ffn, d))
        add(p + "intermediate.dense.bias", (ffn,))
        add(p + "output.dense.weight", (d, ffn))
        add(p + "output.dense.bias", (d,))
        add(p + "LayerNorm.weight", (d,), mean=1.0)
        add(p + "LayerNorm.bias", (d,))
    add("esm.encoder.emb_layer_norm_after.weight", (d,), mean=1.0)
    add("esm.encoder.emb_layer_norm_after.bias", (d,))
    return _materialise(todo, device, dtype, workers)


def mlp_layers(n_layers, in_f, out_f, hidden, seed_off, dtype=torch.bfloat16, device="cpu"):
    """Weights of a `create_mlp` stack (model_utils.py:13-41) as [(W, b|None), ...]."""
    if n_layers == 1:
        return [(_gen((out_f, in_f), 200000 + seed_off, device=device, dtype=dtype), None)]
    out = []
    for i in range(n_layers):
        fi = hidden if i > 0 else in_f
        fo = hidden if i < n_layers - 1 else out_f
        out.append((_gen((fo, fi), 200000 + seed_off + 2 * i, device=device, dtype=dtype),
                    _gen((fo,), 200001 + seed_off + 2 * i, device=device, dtype=dtype)))
    return out


def protein_tokens(lengths, seed=0):
    """ESM token matrix for random proteins over the 20 standard residues (ids 4..23):
    <cls>=0 + residues + <eos>=2, right-padded with <pad>=1."""
    g = torch.Generator().manual_seed(seed)
    maxlen = max(lengths)
    toks = torch.full((len(lengths), maxlen + 2), 1, dtype=torch.int64)
    for i, n in enumerate(lengths):
        toks[i, 0] = 0
        toks[i, 1:n + 1] = torch.randint(4, 24, (n,), generator=g)
        toks[i, n + 1] = 2
    return toks


def prompt_ids(B, T, vocab_text, special_ids, n_protein, seed=0):
    """Random prompt ids in [0, vocab_text) holding `n_protein` <|protein|> slots and a trailing
    [ANSWER]; special_ids = dict(protein=, answer=)."""
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, vocab_text, (B, T), generator=g)
    for b in range(B):
        pos = torch.randperm(T - 2, generator=g)[:n_protein].sort().values
        ids[b, pos] = special_ids["protein"]
        ids[b, T - 1] = special_ids["answer"]
    return ids
