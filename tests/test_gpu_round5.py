"""Round 5: the small-batch decode step (procyon_amd/csrc/pcy_decode_nb.hip) -- every decoder layer of a decode step for 2..8 rows in ONE
launch (beam search runs at batch = beam_size: /root/reference/procyon/model/model_unified.py:751-832, :887; the N-GPU points of
BASELINE configs[3] run 32 / N rows per GPU)."""
import pytest
import torch

from conftest import pcy_disable

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
KW = dict(vocab=4096, d=4096, n_layers=2, n_heads=32, n_kv_heads=8, ffn=14336)


@pytest.fixture(scope="module")
def eng2():
    from procyon_amd import synth
    from procyon_amd.engine import LlamaConfig, LlamaEngine
    return LlamaEngine(synth.llama_state_dict(**KW), LlamaConfig(**KW, max_pos=2048))


@pytest.mark.parametrize("B,T,N,xmin", [(2, 300, 9, 0), (3, 100, 7, 0), (4, 300, 9, 0), (5, 64, 7, 0), (6, 40, 5, 0), (7, 90, 5, 0), (8, 300, 7, 0),
                                        (4, 800, 6, 0), (2, 1100, 5, 0), (4, 420, 6, 384), (2, 500, 6, 384), (5, 420, 6, 384), (8, 450, 5, 384),
                                        (2, 1600, 4, 0)])
def test_decode_nb_step_bit_identical(eng2, monkeypatch, B, T, N, xmin):
    """decode_step_nb_kernel (all layers of a B-row step in one launch: NB dot products per weight row, tagged hand-overs, four-way K split
    of the down projection with {tag, fp32} partial sums) against its launch-per-stage twin (PCY_DISABLE=decode_nb_step: gemv_stream_kernel
    + attn_dec_kernel with the same column slices + gemv_kwin4_kernel): logits of every step, tokens and the appended K / V rows must be
    BIT-identical, eager and under hipGraph replay (the second replayed run starts at the tag / flag state the first one left); PCY_AO_XMIN =
    384 and T = 1600 at 2 rows (default threshold 1536) put the attention's key split + score exchange on (2 / 4 column slices; 5 and 8 rows:
    the units on the projection workgroups too); no watchdog (Context.sync raises)."""
    from procyon_amd.engine import Context, GenState
    monkeypatch.setenv("PCY_NB_MAX", "8")        # (round 6: 8 rows default to the batched launches, which are faster there)
    if xmin:
        monkeypatch.setenv("PCY_AO_XMIN", str(xmin))
    torch.manual_seed(B * 1000 + T)
    emb = (torch.randn(B, T, 4096) * 0.02).to(BF).cuda()

    def run(step, use_graph):
        pcy_disable(monkeypatch, "" if step else "decode_nb_step")
        cache = eng2.new_cache(B, T + N + 2)
        st = GenState(B, KW["vocab"], N + 2, "cuda")
        logits, _ = eng2.prefill(emb, None, cache, "last")
        st.logits.copy_(logits); st.pos.fill_(T)
        eng2.pick(cache, st, B, advance_pos=False)
        out = []
        for i in range(N):
            if i % 3 == 2:
                continue
            eng2.greedy_steps(cache, st, B, 2 if i % 3 == 1 else 1, use_graph=use_graph)
            out.append(st.logits.clone())
        Context.get().sync()
        return torch.stack(out).cpu(), st.tokens_out[:, :N + 1].cpu(), st.logprob.cpu().clone(), cache.k[:, :, :, T:T + N].cpu(), cache.v[:, :, :, T:T + N].cpu()

    ref = run(False, False)
    assert torch.isfinite(ref[0].float()).all()
    for use_graph in (False, True, True):
        got = run(True, use_graph)
        for x, y in zip(got, ref):
            assert torch.equal(x, y), (B, T, use_graph)


def test_decode_nb_rows_are_independent(eng2, monkeypatch):
    """A row's logits in a 4-row step do not depend on its batch mates (every row has its own accumulators, hand-over vectors and cache
    rows): four different prompts decoded together == the same four decoded as (row, three copies of another row)."""
    from procyon_amd.engine import Context, GenState
    pcy_disable(monkeypatch)
    torch.manual_seed(11)
    T, N = 200, 5
    emb = (torch.randn(4, T, 4096) * 0.02).to(BF).cuda()

    def run(e):
        cache = eng2.new_cache(4, T + N + 2)
        st = GenState(4, KW["vocab"], N + 2, "cuda")
        logits, _ = eng2.prefill(e, None, cache, "last")
        st.logits.copy_(logits); st.pos.fill_(T)
        eng2.pick(cache, st, 4, advance_pos=False)
        out = []
        for _ in range(N):
            eng2.greedy_steps(cache, st, 4, 1)
            out.append(st.logits.clone())
        Context.get().sync()
        return torch.stack(out).cpu()

    a = run(emb)
    b = run(torch.stack([emb[2], emb[0], emb[0], emb[0]]).contiguous())
    assert torch.equal(a[:, 2], b[:, 0]) and torch.equal(a[:, 0], b[:, 1]) and torch.equal(b[:, 1], b[:, 2])


def test_decode_nb_beam_loop_bit_identical(eng2, monkeypatch):
    """The reference's production path: diverse beam search, beam 5 (scripts/caption_bulk.py:123-132) = a 5-row decode step + pcy_beam_step
    + the K/V reorder of every step.  One launch per step vs the launch-per-stage twin: tokens, running scores and the parent chain equal."""
    from procyon_amd.engine import BeamState, Context, GenState
    torch.manual_seed(5)
    T, steps, beam = 120, 14, 5
    emb = (torch.randn(1, T, 4096) * 0.02).to(BF).cuda().repeat(beam, 1, 1).contiguous()

    def run(step):
        pcy_disable(monkeypatch, "" if step else "decode_nb_step")
        cache = eng2.new_cache(beam, T + steps + 2)
        logits, _ = eng2.prefill(emb, None, cache, "last")
        bs = BeamState(1, beam, steps, 2, prompt_len=T, device="cuda")
        st = GenState(beam, KW["vocab"], 1, "cuda")
        st.pos, st.next_tok = bs.pos, bs.next_tok
        st.c.pos, st.c.next_tok = bs.pos.data_ptr(), bs.next_tok.data_ptr()
        lg = logits.contiguous()
        for i in range(steps):
            if i > 0:
                eng2.decode_graph(cache, st, beam)
                lg = st.logits
            eng2.beam_step(lg, bs, 5, 0.8)
            eng2.kv_reorder(cache, bs.src, T + i)
        out, n = bs.tokens()
        Context.get().sync()
        return out.cpu(), bs.cur.cpu().clone(), bs.anc[:n].cpu().clone()

    ref, got = run(False), run(True)
    for x, y in zip(got, ref):
        assert torch.equal(x, y)


def test_decode_nb_switch_restores_round4_path(eng2, monkeypatch):
    """PCY_DISABLE=decode_nb: batches of 2..8 rows take the pre-round-5 launches (MFMA GEMVs from 4 rows on); a different arithmetic (other
    accumulation orders), so the logits agree to bf16 noise, not bit for bit -- and switching back and forth inside one process keeps every
    path's own results (per-batch-size tag slots and counters)."""
    from procyon_amd.engine import Context, GenState
    torch.manual_seed(3)
    B, T, N = 4, 150, 4
    emb = (torch.randn(B, T, 4096) * 0.02).to(BF).cuda()

    def run(*off):
        pcy_disable(monkeypatch, *off)
        cache = eng2.new_cache(B, T + N + 2)
        st = GenState(B, KW["vocab"], N + 2, "cuda")
        logits, _ = eng2.prefill(emb, None, cache, "last")
        st.logits.copy_(logits); st.pos.fill_(T)
        eng2.pick(cache, st, B, advance_pos=False)
        eng2.greedy_steps(cache, st, B, N)
        Context.get().sync()
        return st.logits.cpu().clone()

    new1, old, new2 = run(), run("decode_nb"), run()
    assert torch.equal(new1, new2)
    err = float((new1.float() - old.float()).norm() / old.float().norm())
    assert err < 2e-2, err


def test_esm_replayed_chain_is_keyed_on_the_weights(monkeypatch):
    """pcy_esm_encode replays a captured launch chain for short inputs; the chain bakes in every weight pointer.  Two engines of the same
    geometry with DIFFERENT weights built one after the other in one process (the second may get the freed addresses of the first's
    descriptor array and device blocks): each must reproduce its own launch-by-launch result (PCY_DISABLE=esm_graph) on a shape the other
    has already captured."""
    import gc
    from procyon_amd import synth
    from procyon_amd.engine import EsmConfig, EsmEngine
    kw = dict(d=640, n_layers=3, n_heads=20, ffn=2560)
    toks = synth.protein_tokens([200], seed=4)
    outs = []
    for seed in (1, 2, 1):
        sd = synth.esm_state_dict(**kw)
        if seed == 2:                            # other weights behind (possibly) the same addresses
            sd = {k: (v * 1.25).to(v.dtype) if v.dim() == 2 else v for k, v in sd.items()}
        eng = EsmEngine(sd, EsmConfig(**kw))
        pcy_disable(monkeypatch, "esm_graph")
        ref = eng.forward(toks).cpu()
        pcy_disable(monkeypatch)
        for _ in range(4):                      # first sight, capture, replays
            got = eng.forward(toks).cpu()
            assert torch.equal(got, ref), seed
        outs.append(ref)
        del eng
        gc.collect(); torch.cuda.empty_cache()
    assert not torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_create_mlp_reference_signature_on_the_engine():
    """`create_mlp(n_layers, in, out, hidden, dropout)` -> load_state_dict of the reference module's weights (fixture g3, made by the
    reference's own create_mlp) -> forward on the HIP projector: bf16 and fp32 parameters, 1 and 3 layers."""
    from conftest import load_golden, rel_err
    from procyon.model.model_utils import create_mlp
    g = load_golden("g3_mlp")
    for nl in (1, 3):
        for nm, dt in (("bf16", BF), ("f32", torch.float32)):
            m = create_mlp(nl, 24, 40, hidden_features=32, dropout_rate=0.25)
            keys = [k for k in m.state_dict()]
            sd = {}
            for i in range(nl):
                sd[keys[0].replace("0.", f"{3 * i}.") if nl > 1 else "0.weight"] = g[f"w_{nl}_{nm}_{i}"]
                if nl > 1:
                    sd[f"{3 * i}.bias"] = g[f"b_{nl}_{nm}_{i}"]
            m.load_state_dict(sd)
            m = m.to(dt).cuda().eval()
            y = m(g[f"x_{nl}_{nm}"].to(dt).cuda()).cpu()
            assert y.dtype == dt and rel_err(y, g[f"y_{nl}_{nm}"].to(dt)) < (2e-3 if dt == BF else 1e-5), (nl, nm)


def test_split_geometry_config0_full_depth(golden):
    """BASELINE configs[0] "full once" (SURVEY.md section 8d, config 1): ProCyon-Split geometry -- ESM2-150M (30 layers) + Llama-2-7B (32 layers,
    32 KV heads, ffn 11008, V = 32007; /root/reference/README.md:50-51, procyon/training/training_args_IT.py:129-134), one 256-residue
    protein, a 128-token prompt, 64 greedy tokens.  Fixture f6 = the oracle in fp32 (the reference's CPU arithmetic for this config).
    (a) the fp32 operator family (what a caller that never calls .bfloat16() gets): SAME 64 tokens, every logit <= 1e-4;
    (b) the bf16 engine on the bf16-rounded weights: pooled / soft token within bf16 noise, tokens equal up to the first step whose top-2
        margin lies inside 4 x the logit noise (the margin rule); since round 6 the 64 decode steps run in the one-launch step of this geometry
        (decode_step_mha_kernel, pcy_decode_mha.hip)."""
    from conftest import record_parity, rel_err
    from procyon_amd import synth
    from procyon_amd.engine import EsmConfig, EsmEngine, LlamaConfig, LlamaEngine, MlpEngine
    from procyon_amd.engine_f32 import EsmEngineF32, LlamaEngineF32, MlpEngineF32
    F32 = torch.float32
    g = golden("f6_split_config0_fp32")
    LL = dict(vocab=32007, d=4096, n_layers=32, n_heads=32, n_kv_heads=32, ffn=11008)
    ES = dict(d=640, n_layers=30, n_heads=20, ffn=2560)
    toks, ids = g["protein_tokens"].long(), g["prompt_ids"].long()
    T, N = ids.shape[1], g["tokens"].numel()
    cols = g["cols"].long()
    PROT = 32001
    esd = synth.esm_state_dict(**ES, dtype=F32)
    proj = synth.mlp_layers(3, 640, 4096, 2560, seed_off=0, dtype=F32)
    lsd = synth.llama_state_dict(**LL, dtype=F32, workers=16)
    smap = torch.full((ids.numel(),), -1, dtype=torch.int32)
    smap[(ids == PROT).view(-1)] = 0
    # ---- (a) fp32 family
    z = EsmEngineF32(esd, EsmConfig(**ES)).forward(toks)
    soft = MlpEngineF32(proj)(z)
    e_pool, e_soft = rel_err(z[0].cpu(), g["pooled"]), rel_err(soft[0].cpu(), g["soft_token"])
    lm = LlamaEngineF32(lsd, LlamaConfig(**LL, max_pos=4096))
    cache = lm.new_cache(1, T + N)
    logits, _ = lm.prefill(lm.embed_tokens(ids, soft, smap), None, "last", cache=cache)
    out, errs = [], []
    for s in range(N):
        errs.append(rel_err(logits[0, cols].cpu(), g["logits"][s]))
        tok = int(logits[0].argmax())
        out.append(tok)
        if s + 1 < N:
            logits = lm.decode(cache, torch.tensor([tok]), T + s)
    record_parity("split/config0_fp32_family", err_pooled=e_pool, err_soft_token=e_soft, err_logits_max=max(errs), tokens_equal=out == g["tokens"].tolist())
    assert e_pool < 1e-4 and e_soft < 1e-4 and max(errs) < 1e-4, (e_pool, e_soft, max(errs))
    assert out == g["tokens"].tolist()
    del lm, cache
    torch.cuda.empty_cache()
    # ---- (b) bf16 engine
    bf = lambda sd: {k: v.to(BF) for k, v in sd.items()}
    zb = EsmEngine(bf(esd), EsmConfig(**ES)).forward(toks)
    softb = MlpEngine([(w.to(BF).cuda(), b.to(BF).cuda()) for w, b in proj])(zb)
    eb_pool, eb_soft = rel_err(zb[0].float().cpu(), g["pooled"]), rel_err(softb[0].float().cpu(), g["soft_token"])
    eng = LlamaEngine(bf(lsd), LlamaConfig(**LL, max_pos=4096), free_source=True)
    tokb, _, lgb, _ = eng.generate_greedy(eng.embed_tokens(ids, softb, smap), torch.ones(1, T), N, keep_logits=True)
    tokb, lgb = tokb.cpu().view(-1).tolist(), lgb.cpu().float()[0]
    first = next((s for s in range(N) if tokb[s] != int(g["tokens"][s])), N)
    e_first = rel_err(lgb[0, cols], g["logits"][0])
    record_parity("split/config0_bf16_engine", err_pooled=eb_pool, err_soft_token=eb_soft, err_first_logits=e_first, tokens_equal_until=first, steps=N)
    assert eb_pool < 2e-2 and eb_soft < 2e-2 and e_first < 0.15, (eb_pool, eb_soft, e_first)
    if first < N:
        margin = float(g["top_vals"][first, 0] - g["top_vals"][first, 1])
        noise = float((lgb[first, cols] - g["logits"][first]).abs().max())
        assert margin <= 4 * noise, f"bf16 greedy token differs at step {first} outside the noise (margin {margin:.3e}, noise {noise:.3e})"


def test_esm_bulk_batch_replays_its_launch_chain_with_the_same_bits(monkeypatch):
    """Round 5: `pcy_esm_encode` replays the captured launch chain for BULK batches too (up to 40 000 tokens; the retrieval batch is 25 x
    1026), one graph launch per batch instead of ~230 kernel launches -- /root/reference/procyon/evaluate/framework/procyon.py:294-322 calls the
    encoder batch after batch.  Seven proteins of 1000 residues at ESM2-650M width (7 014 tokens: above the old 4 200-token bound, the
    256 x 256 GEMM path): pooled embeddings of the replayed calls (dispatch counter) must EQUAL the launch-by-launch run
    (PCY_DISABLE=esm_graph), with other tokens under the same shape, and the pooled forward may borrow the chain's persistent output
    buffer because a later call of the same shape does not disturb an earlier result."""
    from procyon_amd import _lib as L
    from procyon_amd import synth
    from procyon_amd.engine import Context, EsmConfig, EsmEngine
    ctx = Context.get()
    kw = dict(d=1280, n_layers=2, n_heads=20, ffn=5120)
    eng = EsmEngine(synth.esm_state_dict(**kw, device="cuda"), EsmConfig(**kw))
    cnt = lambda: int(ctx.lib.pcy_debug_dispatch_count(L.DISPATCH_ESM_GRAPH))
    ta, tb = synth.protein_tokens([1000] * 7, seed=11), synth.protein_tokens([1000] * 7, seed=12)
    monkeypatch.setenv("PCY_DISABLE", "esm_graph")
    ref_a, ref_b = eng.forward(ta).clone(), eng.forward(tb).clone()
    monkeypatch.delenv("PCY_DISABLE")
    n0 = cnt()
    o1 = eng.forward(ta)            # first sight of the shape: launch by launch
    o2 = eng.forward(ta)            # second: captured and replayed
    o3 = eng.forward(tb)            # replay with other tokens
    o4 = eng.forward(ta)
    ctx.sync()
    assert cnt() - n0 >= 3, cnt() - n0
    assert torch.equal(o1, ref_a) and torch.equal(o2, ref_a) and torch.equal(o3, ref_b) and torch.equal(o4, ref_a)
