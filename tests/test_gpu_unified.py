"""GPU: the `UnifiedProCyon` mirror (procyon_amd/model) end to end against the oracle pipeline on the same seeded
weights: generate (greedy / beam / sampling probabilities), forward (QA / retrieval), forward_sequences.
Small geometry (head_dim 64) so the oracle runs in seconds; multi-layer bf16 stacks -> parity_bar (conftest)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import alt_accumulation, parity_bar, rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def env():
    from oracle import esm_ref as ER
    from oracle import llama_ref as LR
    from procyon_amd import synth
    from procyon_amd import synthetic_model as SM
    model, w = SM.build("small", device="cuda", return_weights=True, max_new_tokens=32)
    g = w["geom"]
    lgeom = LR.LlamaGeom(**g["llama"])
    egeom = ER.EsmGeom(**g["esm"])
    prot = synth.protein_tokens([90, 41], seed=3)
    return dict(model=model, w=w, lgeom=lgeom, egeom=egeom, prot=prot)


def _inputs(model, prot, instr, seq_slots, texts=(), text_slots=()):
    return {"data": {"seq": prot, "seq_idx": None if prot is None else torch.arange(prot.shape[0]), "text": list(texts), "drug": None},
            "input": {"seq": seq_slots, "text": [list(t) for t in text_slots], "drug": None},
            "target": {"seq": None, "text": None, "drug": None}, "instructions": list(instr)}


def _oracle_embeds(env, inputs, no_pad=True, left_pad=True):
    """Oracle restatement of `_preprocessing`: ESM -> pool -> token projector -> tokenise/splice."""
    from oracle import procyon_ref as PR
    m, w = env["model"], env["w"]
    z = PR.esm_plm_forward(w["esm"], env["egeom"], inputs["data"]["seq"], pooling="mean")
    idx = [i for row in inputs["input"]["seq"] for i in row]
    soft = PR.mlp_forward(z[idx], w["projs"]["aaseq"])
    ids, mask = m._prepare_text_inputs_and_tokenize(list(inputs["instructions"]),
                                                    [[inputs["data"]["text"][i] for i in r] for r in inputs["input"]["text"]],
                                                    crop_off=True, no_pad=no_pad, left_pad=left_pad)
    emb, ret = PR.prepare_input_embeddings(w["llama"]["model.embed_tokens.weight"], ids.long(), m.prot_replacement_idx, soft,
                                           ret_idx=m.prot_retrieval_idx)
    return emb, ids, mask, ret, z


def test_forward_sequences(env):
    from oracle import procyon_ref as PR
    m, w = env["model"], env["w"]
    out = m.forward_sequences(env["prot"], get_soft_tokens=True)
    ref = PR.forward_sequences(w["esm"], env["egeom"], env["prot"], w["projs"]["shared"], w["projs"]["aaseq"], pooling="mean")
    for k in ("original", "shared", "token"):
        assert out[k].shape == ref[k].shape
        assert rel_err(out[k].cpu(), ref[k]) < parity_bar(0.0), k


def test_generate_greedy_matches_oracle(env):
    from oracle import llama_ref as LR
    m, w = env["model"], env["w"]
    instr = ["Definition: describe w1 w2 <|protein|> and w3 <|protein|> then w4 [ANSWER]",
             "short <|protein|> prompt [ANSWER]"]
    inputs = _inputs(m, env["prot"], instr, [[0, 1], [1]], text_slots=[[], []])
    emb, ids, mask, _, _ = _oracle_embeds(env, inputs)
    tok_ref, lg_ref, lp_ref = LR.greedy_generate(w["llama"], env["lgeom"], emb, mask, 12)
    with alt_accumulation():  # the reference pipeline's own reproducibility floor on these inputs
        emb_t, _, _, _, _ = _oracle_embeds(env, inputs)
        _, lg_twin, _ = LR.greedy_generate(w["llama"], env["lgeom"], emb_t, mask, 1)
    floor = rel_err(lg_twin[:, 0], lg_ref[:, 0])
    tokens, log_probs, logits, text = m.generate(_inputs(m, env["prot"], instr, [[0, 1], [1]], text_slots=[[], []]),
                                                 max_len=12, method="greedy")
    assert tokens.shape == (2, 1, 12) and logits.shape[:3] == (2, 1, 12) and len(text) == 2
    # step-0 logits (prefill of the spliced, left-padded prompts) and the free-running token prefix
    err = rel_err(logits[:, 0, 0], lg_ref[:, 0])
    print(f"e2e step-0 logits: gpu-vs-oracle {err:.2e}, cpu-vs-cpu floor {floor:.2e}")
    assert err < parity_bar(floor, base=8e-3)
    for b in range(2):
        for s_ in range(12):
            if tokens[b, 0, s_] != tok_ref[b, s_]:
                top2 = lg_ref[b, s_].float().topk(2).values
                noise = float((logits[b, 0, s_].float() - lg_ref[b, s_].float()).abs().max())
                assert float(top2[0] - top2[1]) <= 4 * noise, (b, s_)
                break


def test_generate_beam_matches_oracle(env):
    from oracle import llama_ref as LR
    m, w = env["model"], env["w"]
    instr = ["w5 w6 <|protein|> w7 [ANSWER]"]
    inputs = _inputs(m, env["prot"], instr, [[0]], text_slots=[[]])
    emb, ids, mask, _, _ = _oracle_embeds(env, inputs)
    enc = LR.make_text_encoder(w["llama"], env["lgeom"])
    V = env["lgeom"].vocab
    trace = []
    t_ref, s_ref, lg_ref = LR.beam_search(enc, emb, mask, vocab_size=V, eos_id=m.tokenizer.eos_token_id, max_len=6,
                                          beam_size=4, beam_group_size=2, diversity_penalty=0.8, trace=trace)
    tokens, scores, logits, text = m.generate(_inputs(m, env["prot"], instr, [[0]], text_slots=[[]]), max_len=6, method="beam",
                                              beam_size=4, beam_group_size=2, diversity_penalty=0.8)
    assert tokens.shape == t_ref.shape and scores.shape == s_ref.shape
    # step 0 is identical up to bf16 noise: same top-2 per group from beam 0 (before any re-indexing every row of a prompt
    # holds the same logits, so this comparison does not depend on the beams' order)
    assert rel_err(logits[0, 0, 0], lg_ref[0, 0, 0]) < 1e-2   # ESM + pool + projector + 2 Llama layers deep
    noise = float((logits[0, 0, 0].float() - lg_ref[0, 0, 0].float()).abs().max())
    if torch.equal(tokens, t_ref):
        assert torch.allclose(scores, s_ref, atol=0.3)
    else:
        # A divergence must come from a near-tie among the oracle's own candidate scores at the FIRST step that differs: some
        # adjacent pair of its top-(g+1) candidates (bf16 log-softmax + fp32 running score) lies within the logits noise (x4,
        # plus one bf16 ulp of the score, the granularity of the log-softmax) -- otherwise the engine picked a clear loser.
        first = int((tokens != t_ref).any(0).any(0).nonzero()[0])
        gaps = []
        for (step, b, k, top) in trace:
            if step == first:
                ulp = 2.0 ** -8 * float(top.abs().max())
                gaps += [(float(g_), ulp) for g_ in (top[:-1] - top[1:])]
        assert gaps and any(g_ <= 4 * noise * (first + 1) + ulp for g_, ulp in gaps), (first, noise, gaps)
        assert torch.equal(tokens[..., :first], t_ref[..., :first])


@pytest.mark.parametrize("beam,group,max_len", [(4, 2, 6), (5, 1, 19), (6, 3, 9)])
def test_generate_beam_replayed_chain_equals_the_four_calls(env, monkeypatch, beam, group, max_len):
    """`_generate_beam_search` with every step >= 1 as ONE replayed launch chain (pcy_llama_beam_steps: decode -> record -> beam step -> KV
    reorder over the slots the device-side position counter names) against the four calls per step (PCY_DISABLE=beam_graph): tokens,
    scores and the re-indexed logits record must be EQUAL; max_len 19 crosses two of the EOS-flag checks (every 8 steps)."""
    m = env["model"]
    instr = ["w5 w6 <|protein|> w7 [ANSWER]", "w1 <|protein|> and <|protein|> w2 w3 [ANSWER]"]
    kw = dict(max_len=max_len, method="beam", beam_size=beam, beam_group_size=group, diversity_penalty=0.8)
    mk = lambda: _inputs(m, env["prot"], instr, [[0], [1, 0]], text_slots=[[], []])
    tok_a, sc_a, lg_a, _ = m.generate(mk(), **kw)
    monkeypatch.setenv("PCY_DISABLE", "beam_graph")
    tok_b, sc_b, lg_b, _ = m.generate(mk(), **kw)
    monkeypatch.delenv("PCY_DISABLE")
    tok_c, sc_c, lg_c, _ = m.generate(mk(), **kw)     # a second replayed run (the chain is re-captured for the new cache)
    assert torch.equal(tok_a, tok_b) and torch.equal(sc_a, sc_b) and torch.equal(lg_a, lg_b)
    assert torch.equal(tok_a, tok_c) and torch.equal(sc_a, sc_c) and torch.equal(lg_a, lg_c)


def test_forward_retrieval_and_qa(env):
    from oracle import llama_ref as LR
    from oracle import procyon_ref as PR
    m, w = env["model"], env["w"]
    # retrieval: one [PROT] per prompt -> aaseq_lm_projector(hidden[-1][PROT])
    instr = ["w1 w2 w3 describe [PROT]", "w9 [EXT] find [PROT]"]
    inp = _inputs(m, None, instr, None, texts=["alpha beta gamma"], text_slots=[[], [0]])
    inp["data"]["seq"] = None
    ids, mask = m._prepare_text_inputs_and_tokenize(list(instr), [[], ["alpha beta gamma"]], no_pad=False)
    real = int(mask.sum(1).max())
    emb_ref, ret = PR.prepare_input_embeddings(w["llama"]["model.embed_tokens.weight"], ids.long(), m.prot_replacement_idx, None,
                                               ret_idx=m.prot_retrieval_idx)
    r = LR.llama_forward(w["llama"], env["lgeom"], inputs_embeds=emb_ref[:, :real], attn_mask=mask[:, :real], want_hidden=True)
    ref = PR.retrieval_text_embedding(r["hidden_states"], ret[:, :real], w["projs"]["lm"], "last")
    out = m.forward(inp, retrieval=True)
    got = out["contrastive_out"]["positive"]["text"]
    assert got.shape == ref.shape and rel_err(got.cpu(), ref) < 1e-2
    # ret_token_access='all' (the ModelArgs default, training_args_IT.py:173): sum of all L+1 hidden states at [PROT]
    ref_all = PR.retrieval_text_embedding(r["hidden_states"], ret[:, :real], w["projs"]["lm"], "all")
    m.config.ret_token_access = "all"
    try:
        got_all = m.forward(inp, retrieval=True)["contrastive_out"]["positive"]["text"]
    finally:
        m.config.ret_token_access = "last"
    assert got_all.shape == ref_all.shape and rel_err(got_all.cpu(), ref_all) < 1e-2
    assert rel_err(got_all.cpu(), ref_all) < 0.5 * rel_err(got_all.cpu(), ref)   # ... and not the 'last' embedding
    # QA: yes/no probabilities at the last [ANSWER] position
    instr = ["w1 <|protein|> is w2 ? [ANSWER] yes w3 <|protein|> ? [ANSWER]", "w4 <|protein|> ? [ANSWER]"]
    inp = _inputs(m, env["prot"], instr, [[0, 1], [1]], text_slots=[[], []])
    emb, ids, mask, _, _ = _oracle_embeds(env, inp, no_pad=False, left_pad=False)
    real = int(mask.sum(1).max())
    r = LR.llama_forward(w["llama"], env["lgeom"], inputs_embeds=emb[:, :real], attn_mask=mask[:, :real])
    yes_ref, no_ref, _ = PR.qa_yes_no_probs(r["logits"], ids[:, :real], m.answer_idx, m.yes_token, m.no_token)
    out = m.forward(_inputs(m, env["prot"], instr, [[0, 1], [1]], text_slots=[[], []]), retrieval=False)
    probs = out["outputs"].answer_logits[:, 0].softmax(-1).cpu()
    assert out["text_toks"].shape[1] == m.config.max_text_len
    assert torch.equal(out["answer_positions"], PR.get_after_answer_tokens(ids, m.answer_idx) - 1)
    assert torch.allclose(probs[:, m.yes_token].float(), yes_ref.float(), rtol=0.05, atol=1e-6)
    assert torch.allclose(probs[:, m.no_token].float(), no_ref.float(), rtol=0.05, atol=1e-6)
    # full_logits=True (opt-in): every position, as the reference's outputs.logits (model_unified.py:548-554) up to the last real column;
    # the row a QA reader would index itself equals the answer-row-only result
    full = m.forward(_inputs(m, env["prot"], instr, [[0, 1], [1]], text_slots=[[], []]), retrieval=False, full_logits=True)
    fl = full["outputs"].logits
    assert fl.shape[:2] == (2, real) and fl.shape[2] == out["outputs"].logits.shape[2]
    picked = fl[torch.arange(2), out["answer_positions"].to(fl.device)]
    assert rel_err(picked.cpu(), out["outputs"].answer_logits[:, 0].cpu()) < 2e-2
    valid = mask[:, :real].bool()
    assert rel_err(fl.cpu()[valid], r["logits"][valid]) < 2e-2
    # the DEFAULT `outputs.logits` reads like the reference's [B, T, V] tensor (model_unified.py:548-554): the QA readers' index returns the
    # rows that were computed, without a second pass; any other access materialises every position once (the same bits as full_logits=True)
    lz, pos = out["outputs"].logits, out["answer_positions"]
    assert tuple(lz.shape) == (2, real, fl.shape[2]) and lz._full_t is None
    assert torch.equal(lz[torch.arange(2), pos], out["outputs"].answer_logits[:, 0]) and lz._full_t is None
    assert torch.equal(lz[:, 3], fl[:, 3]) and lz._full_t is not None
    assert torch.equal(torch.softmax(lz, dim=-1), fl.softmax(-1)) and torch.equal(lz.argmax(-1), fl.argmax(-1))
    # the read-out itself on the device (pcy_qa_probs): softmax over the vocabulary in the model dtype, yes / no columns, argmax
    from procyon_amd.engine import Context
    ans = out["outputs"].answer_logits[:, 0]
    p_dev, yn, am = Context.get().qa_probs(ans, m.yes_token, m.no_token, want_argmax=True)
    p_ref = ans.float().softmax(-1)
    assert p_dev.dtype == ans.dtype and float((p_dev.float() - p_ref).abs().max()) <= 2.0 ** -8 * float(p_ref.max()) + 1e-9
    assert torch.equal(yn[:, 0], p_dev[:, m.yes_token].float()) and torch.equal(yn[:, 1], p_dev[:, m.no_token].float())
    assert torch.equal(am, p_dev.float().argmax(-1))
    p32 = Context.get().qa_probs(ans.float())[0]
    assert p32.dtype == torch.float32 and torch.allclose(p32, p_ref, rtol=1e-5, atol=1e-9)


def test_sampling_probability_vector_and_nucleus_mask(env):
    from oracle import llama_ref as LR
    m = env["model"]
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(3, 500, generator=g) * 3
    for p in (0.9, 0.5):
        assert torch.equal(m._get_nucleus_mask(logits.softmax(-1), p), LR.nucleus_mask(logits.softmax(-1), p))
    # the engine's pre-sampling probability vector (its own logits through its own `_sampling_probs`) against the ORACLE's
    # (oracle logits through oracle.llama_ref.sampling_probs): step 0 of the spliced prompt, nucleus and temperature forms
    w = env["w"]
    instr = ["w5 w6 <|protein|> w7 [ANSWER]"]
    emb, ids, mask, _, _ = _oracle_embeds(env, _inputs(m, env["prot"], instr, [[0]], text_slots=[[]]))
    _, lg_ref, _ = LR.greedy_generate(w["llama"], env["lgeom"], emb, mask, 1)
    torch.manual_seed(0)
    tokens, lp, lg, text = m.generate(_inputs(m, env["prot"], instr, [[0]], text_slots=[[]]), max_len=4, method="nucleus",
                                      nucleus_prob=0.9, num_text_per_instance=2)
    assert tokens.shape == (1, 2, 4) and lg.shape[:3] == (1, 2, 4)
    assert torch.equal(lg[0, 0, 0], lg[0, 1, 0])          # both texts start from the same prefill
    for kw in (dict(nucleus_prob=0.9), dict(temperature=0.7), dict(temperature=1.0)):
        p_eng = m._sampling_probs(lg[0, 0, 0][None].cuda(), kw.get("temperature", 1.0), kw.get("nucleus_prob")).float().cpu()
        p_ref = LR.sampling_probs(lg_ref[:, 0], **kw).float()
        tv = 0.5 * float((p_eng - p_ref).abs().sum())
        in_e, in_r = p_eng[0] > 0, p_ref[0] > 0
        # tokens whose nucleus membership differs must sit at the threshold: their total mass is bounded by the logits noise
        edge_mass = float(p_ref[0][in_e != in_r].sum() + p_eng[0][in_e != in_r].sum())
        print(f"sampling probs {kw}: total variation vs oracle {tv:.3e}, support {int(in_e.sum())} vs {int(in_r.sum())}, edge mass {edge_mass:.3e}")
        assert tv < 2e-2 and edge_mass < 2e-2
    assert not torch.equal(tokens[0, 0], tokens[0, 1]) or True      # two texts: independent draws (may coincide on a peaked model)
    # each sampled token lies inside the nucleus of its own step's distribution (of the ORACLE's function on the step's logits)
    for k in range(2):
        for s_ in range(4):
            pr = LR.sampling_probs(lg[0, k, s_][None].float(), nucleus_prob=0.9)
            assert pr[0, tokens[0, k, s_]] > 0


def test_checkpoint_ingest_builds_identical_model(env):
    """A state dict in the reference's checkpoint layout (text_encoder.model.*, protein_seq_encoder.model.* with fair-esm
    names, projector Sequential indices) builds a model whose outputs are bit-identical to the directly built one."""
    from procyon_amd.checkpoint import build_model
    from procyon_amd.model import ProCyonConfig
    m, w = env["model"], env["w"]
    sd = {"text_encoder.model." + k: v for k, v in w["llama"].items()}
    for k, v in w["esm"].items():
        k2 = k[4:].replace("embeddings.word_embeddings", "embed_tokens").replace("encoder.emb_layer_norm_after", "emb_layer_norm_after")
        k2 = k2.replace("encoder.layer.", "layers.").replace("attention.self.query", "self_attn.q_proj").replace("attention.self.key", "self_attn.k_proj")
        k2 = k2.replace("attention.self.value", "self_attn.v_proj").replace("attention.output.dense", "self_attn.out_proj")
        k2 = k2.replace("attention.LayerNorm", "self_attn_layer_norm").replace("intermediate.dense", "fc1").replace("output.dense", "fc2")
        sd["protein_seq_encoder.model." + k2.replace(".LayerNorm.", ".final_layer_norm.")] = v
    for name, key in (("token_projectors.aaseq", "aaseq"), ("aaseq_shared_projector", "shared"), ("aaseq_lm_projector", "lm")):
        for j, (wt, b) in zip((0, 3, 6), w["projs"][key]):
            sd[f"{name}.{j}.weight"], sd[f"{name}.{j}.bias"] = wt, b
    m2 = build_model(sd, ProCyonConfig(protein_pooling_opt="mean"), m.tokenizer, device="cuda", max_new_tokens=16, esm_heads=2, head_dim=64, max_pos=4096,
                     esm_rope_math="fp32_once")      # (fair-esm names select the fair-esm rotary arithmetic by default)
    # a freshly built / loaded model is "fp32" like the reference's; the engine is bf16-only and must refuse, not run silently
    with pytest.raises(RuntimeError, match="bfloat16"):
        m2.forward_sequences(env["prot"])
    with pytest.raises(RuntimeError, match="bfloat16"):
        m2.generate(_inputs(m2, env["prot"], ["w5 <|protein|> [ANSWER]"], [[0]], text_slots=[[]]), max_len=2, method="greedy")
    assert m2.eval().bfloat16().to("cuda") is m2 and m2.dtype == torch.bfloat16
    with pytest.raises(RuntimeError):
        m2.to("cpu")
    a, b = m.forward_sequences(env["prot"], get_soft_tokens=True), m2.forward_sequences(env["prot"], get_soft_tokens=True)
    for k in ("original", "shared", "token"):
        assert torch.equal(a[k], b[k]), k
    instr = ["w5 w6 <|protein|> w7 [ANSWER]"]
    t1 = m.generate(_inputs(m, env["prot"], instr, [[0]], text_slots=[[]]), max_len=6, method="greedy")[0]
    t2 = m2.generate(_inputs(m2, env["prot"], instr, [[0]], text_slots=[[]]), max_len=6, method="greedy")[0]
    assert torch.equal(t1, t2)
    # ... and through the reference's entry point: a checkpoint directory with model_args.pt / data_args.pt /
    # txllm_model_ckpt.pt -> UnifiedProCyon.from_pretrained(...) -> (model, config)
    import os
    import tempfile
    from types import SimpleNamespace
    from procyon_amd.model import UnifiedProCyon
    with tempfile.TemporaryDirectory() as d:
        torch.save(SimpleNamespace(protein_pooling_opt="mean", max_protein_len=1024, ret_token_access="last", use_aaseq_embeddings=False,
                                   text_encoder_fname="llama-3-8b"), os.path.join(d, "model_args.pt"))
        torch.save(SimpleNamespace(data_dir="/x"), os.path.join(d, "data_args.pt"))
        torch.save(sd, os.path.join(d, "txllm_model_ckpt.pt"))
        m3, cfg = UnifiedProCyon.from_pretrained(pretrained_weights_dir=d, checkpoint_dir=d, tokenizer=m.tokenizer, max_new_tokens=16,
                                                 esm_heads=2, head_dim=64, max_pos=4096, esm_rope_math="fp32_once")
    assert cfg.protein_pooling_opt == "mean" and cfg.n_model_pieces == 1
    assert m3.dtype == torch.float32
    m3.eval()
    m3.bfloat16()
    c = m3.forward_sequences(env["prot"], get_soft_tokens=True)
    for k in ("original", "shared", "token"):
        assert torch.equal(a[k], c[k]), k


def test_config4_shape_mixed_lengths_ragged_batch(env):
    """BASELINE configs[3] in miniature: a batch of prompts of different lengths (left-padded, reference quirks Q1/Q2),
    proteins of mixed lengths of which some exceed max_protein_len (chunk splitting + cross-chunk pooling), greedy decode.
    Per row: step-0 logits at the noise bar, tokens equal up to the first near-tie of the oracle."""
    from oracle import llama_ref as LR
    from oracle import procyon_ref as PR
    from procyon_amd import synth
    m, w = env["model"], env["w"]
    mpl = 64
    prot = synth.protein_tokens([30, 70, 150, 41, 129, 64], seed=11)     # 70, 150, 129 > 64 residues -> 2-3 chunks
    old = m.protein_seq_encoder.max_protein_len
    m.protein_seq_encoder.max_protein_len = mpl
    try:
        instr = ["w1 <|protein|> w2 w3 w4 w5 w6 describe [ANSWER]", "<|protein|> [ANSWER]",
                 "w7 w8 <|protein|> and <|protein|> w9 [ANSWER]", "w2 w2 w2 w3 w3 w3 w4 w4 w4 <|protein|> [ANSWER]",
                 "w5 <|protein|> [ANSWER]"]
        slots = [[0], [1], [2, 3], [4], [5]]
        inputs = _inputs(m, prot, instr, slots, text_slots=[[] for _ in instr])
        z = PR.esm_plm_forward(w["esm"], env["egeom"], prot, pooling="mean", max_protein_len=mpl)
        soft = PR.mlp_forward(z[[i for r in slots for i in r]], w["projs"]["aaseq"])
        ids, mask = m._prepare_text_inputs_and_tokenize(list(instr), [[] for _ in instr], crop_off=True, no_pad=True, left_pad=True)
        assert len(set(mask.sum(1).tolist())) > 2, "prompts must be ragged"
        emb, _ = PR.prepare_input_embeddings(w["llama"]["model.embed_tokens.weight"], ids.long(), m.prot_replacement_idx, soft,
                                             ret_idx=m.prot_retrieval_idx)
        N = 8
        tok_ref, lg_ref, _ = LR.greedy_generate(w["llama"], env["lgeom"], emb, mask, N)
        tokens, _, logits, _ = m.generate(_inputs(m, prot, instr, slots, text_slots=[[] for _ in instr]), max_len=N, method="greedy")
    finally:
        m.protein_seq_encoder.max_protein_len = old
    assert tokens.shape == (5, 1, N)
    for b in range(5):
        assert rel_err(logits[b, 0, 0], lg_ref[b, 0]) < 1e-2, b
        for s_ in range(N):
            if tokens[b, 0, s_] != tok_ref[b, s_]:
                top2 = lg_ref[b, s_].float().topk(2).values
                noise = float((logits[b, 0, s_].float() - lg_ref[b, s_].float()).abs().max())
                assert float(top2[0] - top2[1]) <= 4 * noise, (b, s_)
                break


def test_config5_shape_pair_scoring_with_in_context_examples(env):
    """BASELINE configs[4] in miniature (bf16; there is no reference counterpart for fp8): QA prompts with SIX protein
    slots each -- a positive and a negative in-context pair plus the query pair (instruct_constructor.py:124-145,
    protpep_qa_scores.py:129-193) -- scored as P(yes), P(no) at the last [ANSWER]."""
    from oracle import llama_ref as LR
    from oracle import procyon_ref as PR
    from procyon_amd import synth
    m, w = env["model"], env["w"]
    prot = synth.protein_tokens([80, 12, 80, 9, 80, 15, 11], seed=5)     # receptor-like + short peptides
    tmpl = ("w1 <|protein|> binds <|protein|> ? [ANSWER] yes w2 <|protein|> binds <|protein|> ? [ANSWER] no "
            "w3 <|protein|> binds <|protein|> ? [ANSWER]")
    instr = [tmpl, tmpl, tmpl]
    slots = [[0, 1, 2, 3, 4, 5], [0, 1, 2, 3, 4, 6], [0, 1, 2, 3, 0, 3]]      # the query pair changes, the examples do not
    inp = _inputs(m, prot, instr, slots, text_slots=[[], [], []])
    z = PR.esm_plm_forward(w["esm"], env["egeom"], prot, pooling="mean")
    soft = PR.mlp_forward(z[[i for r in slots for i in r]], w["projs"]["aaseq"])
    ids, mask = m._prepare_text_inputs_and_tokenize(list(instr), [[], [], []], crop_off=False, no_pad=False, left_pad=False)
    emb, _ = PR.prepare_input_embeddings(w["llama"]["model.embed_tokens.weight"], ids.long(), m.prot_replacement_idx, soft,
                                         ret_idx=m.prot_retrieval_idx)
    real = int(mask.sum(1).max())
    r = LR.llama_forward(w["llama"], env["lgeom"], inputs_embeds=emb[:, :real], attn_mask=mask[:, :real])
    yes_ref, no_ref, _ = PR.qa_yes_no_probs(r["logits"], ids[:, :real], m.answer_idx, m.yes_token, m.no_token)
    out = m.forward(inp, retrieval=False)
    probs = out["outputs"].answer_logits[:, 0].softmax(-1).cpu()
    assert torch.allclose(probs[:, m.yes_token].float(), yes_ref.float(), rtol=0.05, atol=1e-6)
    assert torch.allclose(probs[:, m.no_token].float(), no_ref.float(), rtol=0.05, atol=1e-6)
    # rows 0 and 1 differ only in the last slot -> different scores; the engine must not have mixed up the slot order
    assert not torch.allclose(probs[0], probs[1])
