"""GPU: the fp32 operator family and engines (procyon_amd/csrc/pcy_f32.hip, procyon_amd/engine_f32.py) -- the arithmetic of the
reference's callers that never call `.bfloat16()` (/root/reference/examples/paper_analyses/protpep_qa_scores.py:55-58,
/root/reference/scripts/qa_filter_captions.py:17-18, /root/reference/scripts/caption_bulk.py:72-73) -- against the oracle evaluated in
fp32 on the same fp32 weights.  Bar: <= 1e-4 relative on every stack (two fp32 evaluations differ by accumulation order only),
<= 2e-5 on single ops."""
import pytest
import torch
import torch.nn.functional as F

from conftest import record_parity, rel_err

pytestmark = pytest.mark.gpu
F32 = torch.float32


def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std


@pytest.fixture(scope="module")
def ops():
    from procyon_amd.engine_f32 import F32Ops
    return F32Ops()


@pytest.mark.parametrize("M,N,K", [(200, 384, 256), (1026, 1280, 1280), (130, 1001, 128), (17, 256, 5120), (77, 333, 100)])
@pytest.mark.parametrize("mode", ["plain", "bias_gelu", "bias_resid"])
def test_f32_linear(ops, M, N, K, mode):
    """pcy_f32_linear: the MFMA path (K % 16 == 0) and the plain fallback (K = 100), ragged M / N, bias + gelu, bias + residual"""
    A, W, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, std=0.05), rnd(N, seed=3, std=0.1), rnd(M, N, seed=4)
    bias = None if mode == "plain" else b
    ref = F.linear(A, W, bias)
    if mode == "bias_gelu":
        ref = F.gelu(ref)
    if mode == "bias_resid":
        ref = ref + r
    out = ops.linear(A.cuda(), W.cuda(), None if bias is None else bias.cuda(), r.cuda() if mode == "bias_resid" else None,
                     act=1 if mode == "bias_gelu" else 0).cpu()
    e = rel_err(out, ref)
    assert e < 2e-5, e
    assert float((out - ref).abs().max()) < 2e-4 * float(ref.abs().max())


def test_f32_norms_rope_silu(ops):
    x, w, b = rnd(37, 1280, seed=1) * 3 + 0.5, rnd(1280, seed=2) * 0.1 + 1, rnd(1280, seed=3) * 0.1
    assert rel_err(ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-5).cpu(), F.layer_norm(x, (1280,), w, b, 1e-5)) < 2e-6
    from oracle import llama_ref as LR
    assert rel_err(ops.rmsnorm(x.cuda(), w.cuda(), 1e-5).cpu(), LR.rms_norm(x, w, 1e-5)) < 2e-6
    g, u = rnd(50, 300, seed=4) * 2, rnd(50, 300, seed=5)
    assert rel_err(ops.silu_mul(g.cuda(), u.cuda()).cpu(), F.silu(g) * u) < 2e-6
    # rotary on a [tokens, 3 heads x 64] window with prescale, vs the oracle's formula
    from procyon_amd.engine_f32 import _rope_tables_f32
    dh, nh, T = 64, 3, 29
    cos, sin = _rope_tables_f32(dh, 10000.0, 64, "cuda")
    buf = rnd(T, 8 + nh * dh + 5, seed=6)
    pos = torch.randint(0, 64, (T,), generator=torch.Generator().manual_seed(7)).to(torch.int32)
    q = buf[:, 8:8 + nh * dh].reshape(T, nh, dh) * 0.125
    c, s_ = cos.cpu()[pos.long()][:, None], sin.cpu()[pos.long()][:, None]
    ref = buf.clone()
    ref[:, 8:8 + nh * dh] = (q * c + LR.rotate_half(q) * s_).reshape(T, nh * dh)
    d = buf.cuda().contiguous()
    ops.rope(d, 8, nh, dh, pos.cuda(), cos, sin, prescale=0.125)
    assert rel_err(d.cpu(), ref) < 1e-6 and torch.equal(d.cpu()[:, :8], buf[:, :8])


@pytest.mark.parametrize("causal,H,Hkv,dh", [(True, 8, 2, 16), (False, 4, 4, 64), (True, 4, 4, 128)])
def test_f32_attention(ops, causal, H, Hkv, dh):
    """packed sequences of ragged length, grouped heads, causal / bidirectional, a key-keep mask (left pads)"""
    lens = [33, 1, 70, 128]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    ntok = int(cu[-1])
    qkv = rnd(ntok, (H + 2 * Hkv) * dh, seed=1)
    keep = torch.ones(ntok, dtype=torch.uint8)
    keep[int(cu[2]):int(cu[2]) + 9] = 0            # sequence 2 starts with 9 pads
    scale = dh ** -0.5
    out = ops.attention(qkv.cuda(), 0, qkv.cuda(), H * dh, qkv.cuda(), (H + Hkv) * dh, cu.cuda(), keep.cuda(), len(lens), max(lens), H, Hkv, dh, causal, scale).cpu()
    for s_, n in enumerate(lens):
        t0 = int(cu[s_])
        q = qkv[t0:t0 + n, :H * dh].view(n, H, dh).transpose(0, 1)
        k = qkv[t0:t0 + n, H * dh:(H + Hkv) * dh].view(n, Hkv, dh).transpose(0, 1).repeat_interleave(H // Hkv, 0)
        v = qkv[t0:t0 + n, (H + Hkv) * dh:].view(n, Hkv, dh).transpose(0, 1).repeat_interleave(H // Hkv, 0)
        sc = (q @ k.transpose(1, 2)) * scale
        allowed = keep[t0:t0 + n].bool()[None, None, :].expand(1, n, n).clone()
        if causal:
            allowed &= torch.tril(torch.ones(n, n, dtype=torch.bool))[None]
        sc = sc.masked_fill(~allowed, float("-inf"))
        ref = (sc.softmax(-1).nan_to_num(0.0) @ v).transpose(0, 1).reshape(n, H * dh)
        live = allowed[0].any(-1)                   # rows with at least one kept key (pad query rows write zeros)
        assert rel_err(out[t0:t0 + n][live], ref[live]) < 2e-6, s_
        assert bool((out[t0:t0 + n][~live] == 0).all())


def test_f32_esm_engine_vs_oracle():
    """ESM2 (4 layers, d = 320, 5 heads of 64) on fp32 weights, ragged proteins incl. one that is split into chunks: hidden states and
    the mean / corrected-mean / max pooled embeddings vs the oracle in fp32"""
    from oracle import esm_ref as ER
    from oracle import procyon_ref as PR
    from procyon_amd import synth
    from procyon_amd.engine import EsmConfig
    from procyon_amd.engine_f32 import EsmEngineF32
    kw = dict(d=320, n_layers=4, n_heads=5, ffn=1280)
    sd = synth.esm_state_dict(**kw, dtype=F32)
    eng = EsmEngineF32(sd, EsmConfig(**kw))
    toks = synth.protein_tokens([100, 37, 260], seed=1)
    ref = ER.esm_forward(sd, ER.EsmGeom(**kw), toks)
    out = eng.hidden_states(toks).cpu()
    keep = toks != 1
    e = rel_err(out[keep], ref[keep])
    record_parity("fp32/esm_small_hidden", err_hip_oracle_fp32=e)
    assert e < 1e-4, e
    for pooling, corr in (("mean", False), ("mean", True), ("max", False)):
        z_ref = PR.esm_plm_forward(sd, ER.EsmGeom(**kw), toks, pooling=pooling, correction=corr, max_protein_len=128)
        z = eng.forward(toks, pooling=pooling, correction=corr, max_protein_len=128).cpu()
        assert z.shape == z_ref.shape and rel_err(z, z_ref) < 1e-4, (pooling, corr, rel_err(z, z_ref))


def test_f32_llama_engine_vs_oracle():
    """Llama prefill (3 layers, GQA 4/2, head_dim 64) on fp32 weights: logits at chosen rows, final hidden states, the L+1-state sum,
    with a left-padded row, vs the oracle in fp32"""
    from oracle import llama_ref as LR
    from procyon_amd import synth
    from procyon_amd.engine import LlamaConfig
    from procyon_amd.engine_f32 import LlamaEngineF32
    kw = dict(vocab=331, d=256, n_layers=3, n_heads=4, n_kv_heads=2, ffn=512)
    sd = synth.llama_state_dict(**kw, dtype=F32)
    eng = LlamaEngineF32(sd, LlamaConfig(**kw, max_pos=256))
    g = torch.Generator().manual_seed(3)
    B, T = 3, 41
    ids = torch.randint(0, 331, (B, T), generator=g)
    mask = torch.ones(B, T, dtype=torch.long); mask[1, :7] = 0
    emb = eng.embed_tokens(ids)
    assert torch.equal(emb.cpu(), F.embedding(ids, sd["model.embed_tokens.weight"]))
    r = LR.llama_forward(sd, LR.LlamaGeom(**kw), inputs_embeds=emb.cpu(), attn_mask=mask, want_hidden=True)
    rows = torch.tensor([40, 2 * T - 1, 2 * T + 5])
    logits, hidden, hsum = eng.prefill(emb, mask, rows, want_hidden=True, sum_rows=rows)
    valid = mask.bool()
    e_h = rel_err(hidden.cpu()[valid], r["hidden_states"][-1][valid])
    e_l = rel_err(logits.cpu(), r["logits"].reshape(B * T, -1)[rows])
    e_s = rel_err(hsum.cpu(), torch.stack(r["hidden_states"], -1).sum(-1).reshape(B * T, -1)[rows])
    record_parity("fp32/llama_small_prefill", err_hidden=e_h, err_logits=e_l, err_hidden_sum=e_s)
    assert e_h < 1e-4 and e_l < 1e-4 and e_s < 1e-4, (e_h, e_l, e_s)


def test_fp32_model_runs_as_the_unchanged_fp32_callers_drive_it():
    """The model exactly as protpep_qa_scores.py:55-58 / qa_filter_captions.py:17-18 use it -- built from an fp32 checkpoint, `.eval()`,
    NO `.bfloat16()` -- answers `forward` (QA yes/no at [ANSWER]; retrieval [PROT] embedding, ret_token_access last / all) and
    `forward_sequences` AND `generate` (greedy, diverse beam) in fp32 arithmetic, <= 1e-4 from the oracle pipeline in fp32; after
    `.bfloat16()` the same object runs the bf16 engine and its fp32 copies are gone."""
    from oracle import esm_ref as ER
    from oracle import llama_ref as LR
    from oracle import procyon_ref as PR
    from procyon_amd import synth
    from procyon_amd import synthetic_model as SM
    m, w = SM.build("small", device="cuda", return_weights=True, max_new_tokens=8, dtype=F32, as_loaded=True)
    m.eval()
    assert m.dtype == F32
    lgeom, egeom = LR.LlamaGeom(**w["geom"]["llama"]), ER.EsmGeom(**w["geom"]["esm"])
    prot = synth.protein_tokens([90, 41], seed=3)
    mk = lambda instr, seq, slots, texts=(), text_slots=(): {
        "data": {"seq": seq, "seq_idx": None if seq is None else torch.arange(seq.shape[0]), "text": list(texts), "drug": None},
        "input": {"seq": slots, "text": [list(t) for t in text_slots], "drug": None},
        "target": {"seq": None, "text": None, "drug": None}, "instructions": list(instr)}
    # forward_sequences
    out = m.forward_sequences(prot, get_soft_tokens=True)
    ref = PR.forward_sequences(w["esm"], egeom, prot, w["projs"]["shared"], w["projs"]["aaseq"], pooling="mean")
    for k in ("original", "shared", "token"):
        assert out[k].dtype == F32 and rel_err(out[k].cpu(), ref[k]) < 1e-4, k
    # QA
    instr = ["w1 <|protein|> is w2 ? [ANSWER] yes w3 <|protein|> ? [ANSWER]", "w4 <|protein|> ? [ANSWER]"]
    slots = [[0, 1], [1]]
    z = PR.esm_plm_forward(w["esm"], egeom, prot, pooling="mean")
    soft = PR.mlp_forward(z[[0, 1, 1]], w["projs"]["aaseq"])
    ids, mask = m._prepare_text_inputs_and_tokenize(list(instr), [[], []], crop_off=True, no_pad=False, left_pad=False)
    emb, _ = PR.prepare_input_embeddings(w["llama"]["model.embed_tokens.weight"], ids.long(), m.prot_replacement_idx, soft, ret_idx=m.prot_retrieval_idx)
    real = int(mask.sum(1).max())
    r = LR.llama_forward(w["llama"], lgeom, inputs_embeds=emb[:, :real], attn_mask=mask[:, :real])
    yes_ref, no_ref, _ = PR.qa_yes_no_probs(r["logits"], ids[:, :real], m.answer_idx, m.yes_token, m.no_token)
    o = m.forward(mk(instr, prot, slots, text_slots=[[], []]), retrieval=False)
    lg = o["outputs"].answer_logits[:, 0]
    assert lg.dtype == F32
    probs = lg.softmax(-1).cpu()
    pos = o["answer_positions"]
    e_qa = rel_err(lg.cpu(), torch.stack([r["logits"][i, pos[i]] for i in range(2)]))
    assert e_qa < 1e-4, e_qa
    assert torch.allclose(probs[:, m.yes_token], yes_ref, rtol=1e-3, atol=1e-9) and torch.allclose(probs[:, m.no_token], no_ref, rtol=1e-3, atol=1e-9)
    # retrieval, ret_token_access last and all
    instr = ["w1 w2 w3 describe [PROT]", "w9 [EXT] find [PROT]"]
    inp = mk(instr, None, None, texts=["alpha beta gamma"], text_slots=[[], [0]])
    ids, mask = m._prepare_text_inputs_and_tokenize(list(instr), [[], ["alpha beta gamma"]], no_pad=False)
    real = int(mask.sum(1).max())
    emb_ref, ret = PR.prepare_input_embeddings(w["llama"]["model.embed_tokens.weight"], ids.long(), m.prot_replacement_idx, None, ret_idx=m.prot_retrieval_idx)
    r = LR.llama_forward(w["llama"], lgeom, inputs_embeds=emb_ref[:, :real], attn_mask=mask[:, :real], want_hidden=True)
    errs = {}
    for access in ("last", "all"):
        m.config.ret_token_access = access
        got = m.forward(mk(instr, None, None, texts=["alpha beta gamma"], text_slots=[[], [0]]), retrieval=True)["contrastive_out"]["positive"]["text"]
        refv = PR.retrieval_text_embedding(r["hidden_states"], ret[:, :real], w["projs"]["lm"], access)
        errs[access] = rel_err(got.cpu(), refv)
        assert got.dtype == F32 and errs[access] < 1e-4, (access, errs[access])
    m.config.ret_token_access = "last"
    record_parity("fp32/model_small_as_loaded", err_qa_logits=e_qa, err_retrieval_last=errs["last"], err_retrieval_all=errs["all"])
    # generation in fp32 (/root/reference/scripts/caption_bulk.py:70-73 never casts the model and calls generate(method="beam")): greedy and
    # diverse beam search on the fp32 operator family against the oracle's loops in fp32 -- same tokens, logits <= 1e-4
    ginstr = ["w4 <|protein|> w5 [ANSWER]"]
    gslots = [[1]]
    gids, gmask = m._prepare_text_inputs_and_tokenize(list(ginstr), [[]], crop_off=True, no_pad=True, left_pad=True)
    gsoft = PR.mlp_forward(z[[1]], w["projs"]["aaseq"])
    gemb, _ = PR.prepare_input_embeddings(w["llama"]["model.embed_tokens.weight"], gids.long(), m.prot_replacement_idx, gsoft, ret_idx=m.prot_retrieval_idx)
    t_ref, lg_ref, _ = LR.greedy_generate(w["llama"], lgeom, gemb, gmask, 6)
    toks, lp, lg, _ = m.generate(mk(ginstr, prot, gslots, text_slots=[[]]), max_len=6, method="greedy")
    assert lg.dtype == F32 and torch.equal(toks[:, 0], t_ref), (toks, t_ref)
    e_gen = rel_err(lg[:, 0].float(), lg_ref)
    enc = LR.make_text_encoder(w["llama"], lgeom)
    tb_ref, sb_ref, lb_ref = LR.beam_search(enc, gemb, gmask, vocab_size=lgeom.vocab, eos_id=m.tokenizer.eos_token_id, max_len=5,
                                            beam_size=4, beam_group_size=2, diversity_penalty=0.8)
    tb, sb, lb, _ = m.generate(mk(ginstr, prot, gslots, text_slots=[[]]), max_len=5, method="beam", beam_size=4, beam_group_size=2, diversity_penalty=0.8)
    assert torch.equal(tb, tb_ref) and torch.allclose(sb, sb_ref, atol=1e-3), (tb, tb_ref, sb, sb_ref)
    e_beam = rel_err(lb.float(), lb_ref)
    record_parity("fp32/model_small_generate", err_greedy_logits=e_gen, err_beam_logits=e_beam, greedy_tokens_equal=True, beam_tokens_equal=True)
    assert e_gen < 1e-4 and e_beam < 1e-4, (e_gen, e_beam)
    # .bfloat16(): the bf16 engine takes over, the fp32 copies are dropped
    m.bfloat16()
    assert m.text_encoder._src_f32 is None and m.protein_seq_encoder._src_f32 is None and m.aaseq_shared_projector.src_f32 is None
    ob = m.forward_sequences(prot)
    assert ob["shared"].dtype == torch.bfloat16 and rel_err(ob["shared"].float().cpu(), ref["shared"]) < 2e-2
    with pytest.raises(RuntimeError, match="fp32"):      # the fp32 weights are gone: .float() says so at once
        m.float()
