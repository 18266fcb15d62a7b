#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

Two sources of truth, neither of which travels to the GPU box:
  (1) the reference's own pure functions / method bodies, AST-extracted from
      /root/reference and exec'ed here: batched_split_long_seq,
      ProteinPooler, create_mlp, left_pad_tensors, multi_replace_tokens, mask_before,
      get_after_answer_tokens, UnifiedProCyon._prepare_input_embeddings /
      _get_nucleus_mask / _generate_sampling / _generate_beam_search (bound to a stub self);
  (2) the container's transformers 5.15 LlamaForCausalLM / EsmModel (eager attention) with
      the seeded synthetic weights of procyon_amd/synth.py.
The fixtures hold inputs and expected outputs only (data, no reference source).

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import ast
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from procyon_amd import synth  # noqa: E402


def extract(path, names, ns):
    """exec top-level defs/classes `names` (or Class.method as 'Class.method') from a reference file."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
        if isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and f"{node.name}.{sub.name}" in names:
                    mod = ast.Module(body=[sub], type_ignores=[])
                    exec(compile(mod, path, "exec"), ns)
    return ns


def np_(t):
    if isinstance(t, torch.Tensor):
        if t.dtype == torch.bfloat16:
            return t.view(torch.int16).numpy().view(np.uint16)  # raw bf16 bits
        return t.numpy()
    return np.asarray(t)


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: np_(v) for k, v in arrs.items()})
    print("wrote", name, {k: tuple(np_(v).shape) for k, v in arrs.items()})


def ref_ns():
    import torch.nn as nn
    import torch.nn.functional as F
    from typing import List, Optional
    ns = {"torch": torch, "nn": nn, "F": F, "math": math, "np": np, "List": List, "Optional": Optional}
    extract("procyon/training/train_utils.py", {"batched_split_long_seq", "get_after_answer_tokens", "reverse_batched_split"}, ns)
    extract("procyon/model/esm.py", {"ProteinPooler"}, ns)
    extract("procyon/model/model_utils.py", {"create_mlp", "left_pad_tensors"}, ns)
    extract("procyon/model/model_unified.py",
            {"mask_before", "multi_replace_tokens", "UnifiedProCyon._prepare_input_embeddings",
             "UnifiedProCyon._get_nucleus_mask", "UnifiedProCyon._generate_sampling",
             "UnifiedProCyon._generate_beam_search"}, ns)
    return ns


# --------------------------------------------------------------------------- g1 splitter
def g1(ns):
    out = {}
    for n, lens in enumerate([[300, 40], [1024, 7], [1025, 1024], [2049, 300, 1500], [3073, 10], [2050, 3074]]):
        toks = synth.protein_tokens(lens, seed=10 + n)
        new, keys, eos = ns["batched_split_long_seq"](toks.clone(), 1, 2, "split", 1024)
        out[f"in{n}"] = toks
        out[f"rows{n}"] = new
        out[f"keys{n}"] = keys
        out[f"eos{n}"] = torch.tensor([int(e) for e in eos])
    save("g1_split", **out)


# --------------------------------------------------------------------------- g2 pooler
def g2(ns):
    g = torch.Generator().manual_seed(2)
    out = {}
    keys = torch.tensor([0, 1, 2, 1, 1, 2])
    S, D = 12, 8
    lens = [12, 12, 12, 12, 5, 3]  # non-pad tokens per row
    pad = torch.zeros(6, S, dtype=torch.bool)
    for i, l in enumerate(lens):
        pad[i, l:] = True
    for dt, nm in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        z = torch.randn(6, S, D, generator=g).to(dt)
        for method, corr in (("mean", False), ("mean", True), ("max", False)):
            P = ns["ProteinPooler"](pooling_method=method, protein_pooling_correction_option=corr)
            r = P(z.clone(), batch_keys=keys, padding_mask=pad)
            out[f"out_{nm}_{method}_{int(corr)}"] = r
        out[f"z_{nm}"] = z
    out["keys"] = keys
    out["pad"] = pad
    save("g2_pool", **out)


# --------------------------------------------------------------------------- g3 create_mlp
def g3(ns):
    out = {}
    g = torch.Generator().manual_seed(3)
    for nl in (1, 3):
        for dt, nm in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            torch.manual_seed(30 + nl)
            m = ns["create_mlp"](nl, 24, 40, hidden_features=32, dropout_rate=0.25).eval().to(dt)
            x = torch.randn(5, 24, generator=g).to(dt)
            with torch.no_grad():
                y = m(x)
            lin = [l for l in m if isinstance(l, torch.nn.Linear)]
            for i, l in enumerate(lin):
                out[f"w_{nl}_{nm}_{i}"] = l.weight.detach()
                if l.bias is not None:
                    out[f"b_{nl}_{nm}_{i}"] = l.bias.detach()
            out[f"x_{nl}_{nm}"] = x
            out[f"y_{nl}_{nm}"] = y
    save("g3_mlp", **out)


# --------------------------------------------------------------------------- g4 pad / splice
def g4(ns):
    out = {}
    ts = [torch.arange(5), torch.arange(2) + 10, torch.arange(7) + 20]
    bt, am = ns["left_pad_tensors"](ts, pad_value=99)
    out["lp_tok"], out["lp_mask"] = bt, am
    a = [5, 9, 1, 9, 2, 9, 3]
    b = [[70, 71], [80], [90, 91, 92]]
    out["mr_train"] = torch.tensor(ns["multi_replace_tokens"](list(a), b, 9, eval=False))
    out["mr_eval"] = torch.tensor(ns["multi_replace_tokens"](list(a), b, 9, eval=True))
    # _prepare_input_embeddings on a stub self
    g = torch.Generator().manual_seed(4)
    V, d = 40, 16
    emb = torch.nn.Embedding(V, d)
    emb.weight.data = torch.randn(V, d, generator=g)
    ids = torch.randint(0, 30, (3, 10), generator=g)
    PROT, RET, STRUCT, DRUG = 31, 32, 33, 34
    ids[0, 2] = PROT; ids[0, 3] = STRUCT; ids[0, 7] = PROT; ids[0, 8] = STRUCT
    ids[1, 1] = PROT; ids[1, 2] = STRUCT; ids[1, 5] = DRUG; ids[1, 9] = RET
    ids[2, 4] = RET; ids[2, 6] = DRUG
    prot_soft = torch.randn(3, d, generator=g)
    struct_soft = [torch.randn(2, d, generator=g), torch.randn(1, d, generator=g), []]
    drug_soft = torch.randn(2, d, generator=g)
    for roll in (0, 1):
        self = types.SimpleNamespace(input_embeddings=emb, prot_replacement_idx=PROT, struct_idx=STRUCT,
                                     drug_idx=DRUG, prot_retrieval_idx=RET,
                                     config=types.SimpleNamespace(roll_num=roll))
        with torch.no_grad():
            z, ret = ns["_prepare_input_embeddings"](self, ids, protein_soft_tokens=prot_soft,
                                                     protein_struct_tokens=struct_soft, drug_soft_tokens=drug_soft)
        out[f"z_roll{roll}"] = z
        out[f"ret_roll{roll}"] = ret
    out.update(emb_w=emb.weight.data, ids=ids, prot_soft=prot_soft, struct_soft0=struct_soft[0],
               struct_soft1=struct_soft[1], drug_soft=drug_soft)
    save("g4_pad_splice", **out)


# --------------------------------------------------------------------------- g5 ESM (HF 5.15)
def hf_esm(d, L, H, F_, dtype):
    from transformers import EsmConfig, EsmModel
    cfg = EsmConfig(vocab_size=33, hidden_size=d, num_hidden_layers=L, num_attention_heads=H,
                    intermediate_size=F_, max_position_embeddings=1026, position_embedding_type="rotary",
                    token_dropout=True, emb_layer_norm_before=False, pad_token_id=1, mask_token_id=32,
                    layer_norm_eps=1e-5, attn_implementation="eager", hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0)
    m = EsmModel(cfg, add_pooling_layer=False).eval()
    sd = synth.esm_state_dict(d, L, H, F_, dtype=torch.float32)
    m.load_state_dict({k[4:]: v for k, v in sd.items()}, strict=False)
    return m.to(dtype)


def g5():
    out = {}
    toks = synth.protein_tokens([40, 17, 63, 5], seed=5)
    toks[0, 5] = 32  # one <mask>
    out["tokens"] = toks
    for dt, nm in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        m = hf_esm(64, 2, 4, 128, dt)
        with torch.no_grad():
            out[f"h_masked_{nm}"] = m(input_ids=toks, attention_mask=(toks != 1).long()).last_hidden_state
            out[f"h_nomask_{nm}"] = m(input_ids=toks).last_hidden_state
    # one full-width ESM2-650M layer (d1280, H20, F5120), bf16, S=70
    toks2 = synth.protein_tokens([68, 31], seed=6)
    m = hf_esm(1280, 1, 20, 5120, torch.bfloat16)
    with torch.no_grad():
        out["h_650m_1layer_bf16"] = m(input_ids=toks2, attention_mask=(toks2 != 1).long()).last_hidden_state
    out["tokens_650m"] = toks2
    save("g5_esm", **out)


# --------------------------------------------------------------------------- g6 Llama (HF 5.15)
def hf_llama(V, d, L, H, Hkv, F_, theta, dtype):
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(vocab_size=V, hidden_size=d, intermediate_size=F_, num_hidden_layers=L,
                      num_attention_heads=H, num_key_value_heads=Hkv, max_position_embeddings=8192,
                      rms_norm_eps=1e-5, rope_theta=theta, attn_implementation="eager",
                      tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    m.load_state_dict(synth.llama_state_dict(V, d, L, H, Hkv, F_, dtype=torch.float32))
    return m.to(dtype)


TINY = dict(V=300, d=128, L=2, H=8, Hkv=2, F_=256)


def g6():
    out = {}
    g = torch.Generator().manual_seed(6)
    B, T, NDEC = 2, 12, 8
    emb = torch.randn(B, T, TINY["d"], generator=g) * 0.02
    mask_eq = torch.ones(B, T)
    mask_rag = torch.ones(B, T); mask_rag[1, :3] = 0
    out["embeds"] = emb
    out["mask_rag"] = mask_rag
    for theta in (1e4, 5e5):
        for dt, nm in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            m = hf_llama(theta=theta, dtype=dt, **TINY)
            for mname, mask in (("eq", mask_eq), ("rag", mask_rag)):
                tag = f"{nm}_{int(theta)}_{mname}"
                with torch.no_grad():
                    o = m(inputs_embeds=emb.to(dt), attention_mask=mask, use_cache=True, output_hidden_states=True)
                    out[f"prefill_logits_{tag}"] = o.logits
                    out[f"prefill_hidden_{tag}"] = torch.stack(o.hidden_states)
                    out[f"prefill_k0_{tag}"] = o.past_key_values.layers[0].keys
                    out[f"prefill_v1_{tag}"] = o.past_key_values.layers[1].values
                    toks, lg = [], []
                    pk = o.past_key_values
                    nxt = o.logits[:, -1].argmax(-1, keepdim=True)
                    for _ in range(NDEC):  # decode: no mask, positions = cache length (Q1/Q2)
                        toks.append(nxt)
                        o = m(input_ids=nxt, past_key_values=pk, use_cache=True)
                        pk = o.past_key_values
                        lg.append(o.logits[:, -1])
                        nxt = o.logits[:, -1].argmax(-1, keepdim=True)
                    out[f"dec_in_tokens_{tag}"] = torch.cat(toks, 1)
                    out[f"dec_logits_{tag}"] = torch.stack(lg, 1)
    # one full-width Llama-3-8B layer (d4096 H32 Hkv8 F14336), small vocab, bf16
    FW = dict(V=512, d=4096, L=1, H=32, Hkv=8, F_=14336)
    m = hf_llama(theta=1e4, dtype=torch.bfloat16, **FW)
    embw = (torch.randn(1, 24, 4096, generator=g) * 0.02).bfloat16()
    with torch.no_grad():
        o = m(inputs_embeds=embw, attention_mask=torch.ones(1, 24), use_cache=True, output_hidden_states=True)
        out["fw_embeds"] = embw
        out["fw_prefill_hidden"] = o.hidden_states[-1]
        out["fw_prefill_logits_last"] = o.logits[:, -1]
        nxt = o.logits[:, -1].argmax(-1, keepdim=True)
        o2 = m(input_ids=nxt, past_key_values=o.past_key_values, use_cache=True)
        out["fw_dec_token"] = nxt
        out["fw_dec_logits"] = o2.logits[:, -1]
    save("g6_llama", **out)


# --------------------------------------------------------------------------- g7 generation loops
class _KVList(list):
    pass


def g7(ns):
    """Reference generation loops (extracted method bodies) driving the HF tiny Llama."""
    out = {}
    m = hf_llama(theta=1e4, dtype=torch.float32, **TINY)
    mb = hf_llama(theta=1e4, dtype=torch.bfloat16, **TINY)

    def make_self(model, eos):
        from transformers import DynamicCache

        def text_encoder(input_embeds=None, input_ids=None, attn_masks=None, past_key_values=None, use_cache=True):
            cache = None
            if past_key_values is not None:  # list of [K,V] -> DynamicCache
                cache = DynamicCache()
                for li, (k, v) in enumerate(past_key_values):
                    cache.update(k, v, li)
            with torch.no_grad():
                o = model(inputs_embeds=input_embeds, input_ids=input_ids, attention_mask=attn_masks,
                          past_key_values=cache, use_cache=True)
            pk = [[l.keys, l.values] for l in o.past_key_values.layers]
            return types.SimpleNamespace(logits=o.logits, past_key_values=pk)
        text_encoder.model = types.SimpleNamespace(vocab_size=TINY["V"])
        s = types.SimpleNamespace(text_encoder=text_encoder,
                                  tokenizer=types.SimpleNamespace(eos_token_id=eos))
        s._get_nucleus_mask = lambda probs, p: ns["_get_nucleus_mask"](s, probs, p)
        return s

    g = torch.Generator().manual_seed(7)
    B, T = 2, 10
    emb = torch.randn(B, T, TINY["d"], generator=g) * 0.02
    mask = torch.ones(B, T)
    out["embeds"] = emb
    for nm, model, dt in (("f32", m, torch.float32), ("bf16", mb, torch.bfloat16)):
        s = make_self(model, eos=299)
        tok, lp, lg = ns["_generate_sampling"](s, emb.to(dt), mask, max_len=16, num_text_per_instance=1,
                                               temperature=1.0, greedy=True, nucleus_prob=None)
        out[f"greedy_tokens_{nm}"] = tok
        out[f"greedy_logprob_{nm}"] = lp.float()
        out[f"greedy_logits_{nm}"] = lg
        for (bs, gs, pen) in ((4, 2, 0.8), (4, 4, 0.0), (6, 1, 0.5)):
            tok, sc, lg = ns["_generate_beam_search"](s, emb.to(dt), mask, max_len=10, beam_size=bs,
                                                      beam_group_size=gs, diversity_penalty=pen)
            out[f"beam_tokens_{nm}_{bs}_{gs}"] = tok
            out[f"beam_scores_{nm}_{bs}_{gs}"] = sc.float()
    # EOS early stop: pick eos = a token the f32 greedy run emits
    eos = int(out["greedy_tokens_f32"][0, 0, 2])
    s = make_self(m, eos=eos)
    tok, sc, lg = ns["_generate_beam_search"](s, emb[:1], mask[:1], max_len=12, beam_size=2,
                                              beam_group_size=2, diversity_penalty=0.0)
    out["beam_eos_id"] = torch.tensor(eos)
    out["beam_eos_tokens"] = tok
    out["beam_eos_steps"] = torch.tensor(lg.shape[2])
    # nucleus mask on a small prob table
    probs = torch.softmax(torch.randn(3, 50, generator=g) * 2, -1)
    s = make_self(m, eos=299)
    for p in (0.9, 0.5):
        out[f"nucleus_mask_{int(p * 100)}"] = ns["_get_nucleus_mask"](s, probs, p)
    out["nucleus_probs"] = probs
    save("g7_generate", **out)


# --------------------------------------------------------------------------- g8 QA helpers
def g8(ns):
    out = {}
    ANS = 50
    toks = torch.tensor([[1, 50, 3, 4, 50, 6, 7, 0], [50, 2, 3, 4, 5, 6, 7, 8], [9, 9, 9, 50, 9, 50, 50, 9]])
    out["toks"] = toks
    out["after_answer"] = ns["get_after_answer_tokens"](toks, ANS)
    out["mask_before_last"] = ns["mask_before"](toks.clone(), ANS, before_last_answer=True)
    save("g8_qa", **out)


# --------------------------------------------------------------------------- g9 end-to-end tiny (config-1 shaped)
def g9(ns):
    """256-residue protein -> ESM tiny -> mean pool -> 3-layer token projector -> splice ->
    Llama tiny -> 64 greedy tokens, fp32, composed from HF modules + the reference's functions."""
    out = {}
    dt = torch.float32
    esm = hf_esm(64, 2, 4, 128, dt)
    lm = hf_llama(theta=1e4, dtype=dt, **TINY)
    toks = synth.protein_tokens([256], seed=0)
    new, keys, _ = ns["batched_split_long_seq"](toks.clone(), 1, 2, "split", 1024)
    with torch.no_grad():
        z = esm(input_ids=new, attention_mask=(new != 1).long()).last_hidden_state
    pooled = ns["ProteinPooler"]("mean", False)(z, batch_keys=keys, padding_mask=(new == 1))
    layers = synth.mlp_layers(3, 64, TINY["d"], 96, seed_off=0, dtype=dt)
    x = pooled
    for i, (w, b) in enumerate(layers):
        x = torch.nn.functional.linear(x, w, b)
        if i < 2:
            x = torch.nn.functional.gelu(x)
    PROT, ANSWER = 290, 291
    ids = synth.prompt_ids(1, 32, 280, dict(protein=PROT, answer=ANSWER), n_protein=1, seed=9)
    emb = torch.nn.Embedding(TINY["V"], TINY["d"])
    emb.weight.data = lm.model.embed_tokens.weight.data
    self = types.SimpleNamespace(input_embeddings=emb, prot_replacement_idx=PROT, struct_idx=-1, drug_idx=-2,
                                 prot_retrieval_idx=-3, config=types.SimpleNamespace(roll_num=0))
    with torch.no_grad():
        zin, _ = ns["_prepare_input_embeddings"](self, ids, protein_soft_tokens=x)

    def text_encoder(input_embeds=None, input_ids=None, attn_masks=None, past_key_values=None, use_cache=True):
        with torch.no_grad():
            return lm(inputs_embeds=input_embeds, input_ids=input_ids, attention_mask=attn_masks,
                      past_key_values=past_key_values, use_cache=True)
    s = types.SimpleNamespace(text_encoder=text_encoder)
    tok, lp, lg = ns["_generate_sampling"](s, zin, torch.ones(1, 32), max_len=64, num_text_per_instance=1,
                                           temperature=1.0, greedy=True, nucleus_prob=None)
    out.update(protein_tokens=toks, prompt_ids=ids, pooled=pooled, soft_token=x, tokens=tok, logprob=lp.float(),
               logits_first=lg[:, :, 0], logits_last=lg[:, :, -1])
    save("g9_e2e_tiny", **out)


# --------------------------------------------------------------------------- g10 host text preparation
def g10(ns):
    """Reference `_prepare_text_inputs_and_tokenize` (model_unified.py:1177-1293) driven by the build's synthetic
    tokenizer: [EXT] splicing, per-description truncation budget, drug tail, EOS/pad, left padding."""
    import random
    from procyon_amd.tokenizer import SyntheticTokenizer
    extract("procyon/model/model_unified.py", {"UnifiedProCyon._prepare_text_inputs_and_tokenize"}, ns)
    ns["random"] = random
    tok = SyntheticTokenizer()
    out = {}
    instr = ["Definition: w1 w2 [EXT] then <|protein|> and [EXT] finally [ANSWER]",
             "Short one <|protein|> [ANSWER] [EXT]",
             "No description here <|protein|> [ANSWER]"]
    texts = [["alpha beta gamma delta " * 30, "Drug: <|drug|> tail words " + "filler " * 10], ["only one description " * 5], []]
    for max_text_len, tag in ((64, "short"), (2048, "long")):
        for no_pad, left_pad, nm in ((False, False, "pad"), (True, True, "leftpad")):
            self = types.SimpleNamespace(tokenizer=tok, config=types.SimpleNamespace(max_text_len=max_text_len),
                                         drug_idx=tok.convert_tokens_to_ids("<|drug|>"), ext_idx=tok.convert_tokens_to_ids("[EXT]"),
                                         training=False, context_crop_sampling=False)
            ids, mask = ns["_prepare_text_inputs_and_tokenize"](self, list(instr), [list(t) for t in texts], crop_off=True,
                                                                retrieval=False, no_pad=no_pad, left_pad=left_pad)
            out[f"ids_{tag}_{nm}"] = ids
            out[f"mask_{tag}_{nm}"] = mask
    save("g10_text_prep", **out)


def g11():
    """Reference `get_prompt` / `get_prompt_open_def` (instruct_constructor.py:111-365, imported from the file -- it only needs
    torch / numpy) on seven of the reference's task templates (data files): every category, PPI and non-PPI, the three
    sequence types.  The fixture holds the task dicts (inputs) and the returned tuples (expected outputs)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("ref_ic", os.path.join(REF, "procyon/data/instruct_tune/instruct_constructor.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cases = []
    tasks = {}
    for name in ("uniprot_all_caption", "disgenet_all_retrieval", "omim_all_qa", "protein_homology_qa", "protein_homology_retrieval",
                 "domain_pfam_all_qa", "peptide_all_retrieval"):
        task = json.load(open(os.path.join(REF, "procyon/data/instruct_tune/tasks", name + ".json")))
        tasks[name] = task
        ppi = "aaseq_1" in task["Positive Examples"][0]
        for aa in ("protein", "domain", "peptide", None):
            for ne in (None, 0, 1, 2):
                for samp in (False, True):
                    cases.append(dict(task=name, fn="get_prompt", aaseq_type=aa, num_examples=ne, sample_examples=samp, is_ppi=ppi,
                                      out=list(ref.get_prompt(task, ne, False, ppi, samp, aa))))
                cases.append(dict(task=name, fn="get_prompt_open_def", aaseq_type=aa, num_examples=ne, sample_examples=False, is_ppi=ppi,
                                  out=list(ref.get_prompt_open_def(task, ne, False, ppi, False, aa))))
    import gzip
    with gzip.open(os.path.join(HERE, "g11_prompts.json.gz"), "wt") as f:
        json.dump(dict(tasks=tasks, cases=cases), f)
    print("wrote g11_prompts.json.gz", len(cases), "cases")


# --------------------------------------------------------------------------- g12 reverse_batched_split (ESM_PLM.forward(aggregate=False))
def g12(ns):
    """per-position states of split proteins laid end to end again (train_utils.py:1599-1649), fed by the reference's own splitter"""
    out = {}
    g = torch.Generator().manual_seed(12)
    for n, lens in enumerate([[300, 40], [2049, 300, 1500], [3073, 10], [1025, 1024, 7]]):
        toks = synth.protein_tokens(lens, seed=120 + n)
        new, keys, eos = ns["batched_split_long_seq"](toks.clone(), 1, 2, "split", 1024)
        emb = torch.randn(new.shape[0], new.shape[1], 2, generator=g)
        out[f"toks{n}"] = toks
        out[f"emb{n}"] = emb
        out[f"keys{n}"] = keys
        out[f"eos{n}"] = torch.tensor([int(e) for e in eos])
        out[f"out{n}"] = ns["reverse_batched_split"](emb.clone(), keys, eos)
    save("g12_reverse_split", **out)


# --------------------------------------------------------------------------- g13 input builders (row f2)
def g13():
    """The REFERENCE's `create_caption_input_simple` / `create_qa_input_simple` / `create_input_retrieval`
    (procyon/data/inference_utils.py:67-244, 247-420, 663-843, AST-extracted; their helpers `construct_task_id`,
    `get_text_sequences_compositions`, `get_prompt[_open_def]` and the column tables come from the reference too) run over the
    synthetic ProCyon-Instruct tree of tests/synth_instruct.py.  The fixture holds the inputs the tree is built from (seven task
    templates are already in g11; here the description-column tables of the four datasets of the tree) and the returned
    dictionaries.  The CPU test rebuilds the identical tree and runs the shim's builders."""
    import gzip
    import importlib.util
    import json
    import tempfile
    import pandas as pd
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth_instruct as SI

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ref_ic = load("ref_ic", "procyon/data/instruct_tune/instruct_constructor.py")
    ref_const = load("ref_const", "procyon/data/constants.py")
    datasets = ("uniprot", "omim", "disgenet", "pfam")
    tables = {"ENTITY_DESCRIPTION_NAMES": {d: ref_const.ENTITY_DESCRIPTION_NAMES[d] for d in datasets}}
    for key in ("QA_SUBSETS", "RETRIEVAL_SUBSETS", "CAPTION_SUBSETS"):
        t = getattr(ref_const, key)
        tables[key] = {str(ver): {d: cols[d] for d in datasets if d in cols} for ver, cols in t.items()}
    with gzip.open(os.path.join(HERE, "g11_prompts.json.gz"), "rt") as f:
        tasks = json.load(f)["tasks"]
    with tempfile.TemporaryDirectory() as root:
        data_dir, home_dir = SI.build_tree(root, tasks, tables)
        from typing import Dict, List, Optional
        from procyon.training.training_args_IT import DataArgs       # the shim's argument class (same fields)
        ns = {"torch": torch, "pd": pd, "np": np, "os": os, "json": json, "List": List, "Dict": Dict, "Optional": Optional,
              "DataArgs": DataArgs, "DATA_DIR": data_dir, "HOME_DIR": home_dir, "get_prompt": ref_ic.get_prompt,
              "get_prompt_open_def": ref_ic.get_prompt_open_def,
              "QA_SUBSETS": ref_const.QA_SUBSETS, "RETRIEVAL_SUBSETS": ref_const.RETRIEVAL_SUBSETS, "CAPTION_SUBSETS": ref_const.CAPTION_SUBSETS,
              "ENTITY_DESCRIPTION_NAMES": ref_const.ENTITY_DESCRIPTION_NAMES, "DRUGMASK": None}
        extract("procyon/data/it_collator.py", {"construct_task_id"}, ns)
        extract("procyon/data/data_utils.py", {"get_text_sequences_compositions"}, ns)
        # module-level tables of the reference file, built the way it builds them (inference_utils.py:42-51)
        ns["UNIPROT_IDS"] = pd.read_pickle(os.path.join(data_dir, "integrated_data/v1/protein/", "protein_info_filtered.pkl"))[["index", "protein_id", "name"]]
        ns["functional_descriptions"] = pd.read_pickle(
            os.path.join(data_dir, "integrated_data/v1/protein/uniprot_functional_descriptions.pkl")).sort_values("index", axis=0)["function"]
        extract("procyon/data/inference_utils.py", {"create_caption_input_simple", "create_qa_input_simple", "create_input_retrieval",
                                                    "uniprot_id_to_index", "index_to_uniprot_id"}, ns)
        cases = []
        for fn, kw in SI.BUILDER_CASES:
            try:
                out = SI.jsonable(ns[fn](data_args=DataArgs(), **kw))
            except Exception as e:      # the reference's own behaviour on this argument combination (e.g. QA + task_definition: KeyError 'answer')
                out = {"__raises__": type(e).__name__}
            cases.append(dict(fn=fn, kwargs=kw, out=out))
        ids = dict(p530=ns["uniprot_id_to_index"]("P00530"), i5=ns["index_to_uniprot_id"](5))
    with gzip.open(os.path.join(HERE, "g13_input_builders.json.gz"), "wt") as f:
        json.dump(dict(tables=tables, cases=cases, ids=ids), f)
    print("wrote g13_input_builders.json.gz", len(cases), "cases", os.path.getsize(os.path.join(HERE, "g13_input_builders.json.gz")), "bytes")


# --------------------------------------------------------------------------- g14 _init_tokenizer on a real tokenizer object (row f2)
def g14(ns):
    """The REFERENCE's `_init_tokenizer` (model_unified.py:1088-1133), its yes / no id rule (:342-347) and
    `_prepare_text_inputs_and_tokenize` (:1177-1293), run over a real `PreTrainedTokenizerFast` (tests/synth_tokenizer.py: a
    Llama-3-style byte-level BPE built in the container, loaded through $LLAMA3_PATH like the reference does)."""
    import json
    import random
    import tempfile
    import transformers
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth_tokenizer as ST
    extract("procyon/model/model_unified.py", {"UnifiedProCyon._init_tokenizer", "UnifiedProCyon._prepare_text_inputs_and_tokenize"}, ns)
    ns.update(random=random, os=os, transformers=transformers)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        ST.build(d)
        os.environ["LLAMA3_PATH"] = d
        self = types.SimpleNamespace(config=types.SimpleNamespace(text_encoder_fname="llama-3-8b", max_text_len=96), pretrained_weights_dir=d,
                                     training=False, context_crop_sampling=False)
        ns["_init_tokenizer"](self)
        tk = self.tokenizer
        out["ids"] = {k: int(getattr(self, k)) for k in ("prot_replacement_idx", "prot_retrieval_idx", "answer_idx", "struct_idx", "drug_idx", "ext_idx")}
        out["sep_token_id"], out["pad_token_id"], out["len"] = int(tk.sep_token_id), int(tk.pad_token_id), len(tk)
        out["eos_token_id"], out["bos_token_id"], out["padding_side"] = int(tk.eos_token_id), int(tk.bos_token_id), tk.padding_side
        out["yes_token"] = int(tk.encode(" yes", add_special_tokens=False)[0])      # model_unified.py:342-344 ("llama-3" in the name)
        out["no_token"] = int(tk.encode(" no", add_special_tokens=False)[0])
        out["encode_samples"] = {t: tk(t, add_special_tokens=True)["input_ids"] for t in
                                 ("w <|protein|> x [ANSWER]", "[EXT][PROT] yes", " yes", "yes", "<|drug|> tail")}
        for no_pad, left_pad, nm in ((False, False, "pad"), (True, True, "leftpad")):
            ids, mask = ns["_prepare_text_inputs_and_tokenize"](self, list(ST.INSTRUCTIONS), [list(t) for t in ST.TEXTS], crop_off=True,
                                                                retrieval=False, no_pad=no_pad, left_pad=left_pad)
            out[f"prep_ids_{nm}"], out[f"prep_mask_{nm}"] = ids.tolist(), mask.tolist()
        out["decode"] = tk.batch_decode(torch.tensor(out["prep_ids_leftpad"])[:, -12:])
    with open(os.path.join(HERE, "g14_real_tokenizer.json"), "w") as f:
        json.dump(out, f)
    print("wrote g14_real_tokenizer.json", out["ids"], out["len"], os.path.getsize(os.path.join(HERE, "g14_real_tokenizer.json")), "bytes")


if __name__ == "__main__":
    torch.manual_seed(0)
    if sys.argv[1:] == ["g14"]:
        g14(ref_ns())
        sys.exit(0)
    if sys.argv[1:] == ["g13"]:
        g13()
        sys.exit(0)
    if sys.argv[1:] == ["g12"]:
        g12(ref_ns())
        sys.exit(0)
    if sys.argv[1:] == ["g11"]:
        g11()
        sys.exit(0)
    ns = ref_ns()
    g1(ns); g2(ns); g3(ns); g4(ns); g5(); g6(); g7(ns); g8(ns); g9(ns); g10(ns); g11(); g12(ns); g13(); g14(ns)
