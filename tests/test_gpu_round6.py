"""Round 6: the mid-batch decode step (procyon_amd/csrc/pcy_decode_mb.hip) -- every decoder layer of a 9..32-row decode step in ONE launch
(every caller of the reference generates with beam 10 or 20: /root/reference/scripts/caption_bulk.py:193-194,
/root/reference/procyon/evaluate/framework/procyon.py:72-76) -- and the beam search's one-prefill-per-prompt
(/root/reference/procyon/model/model_unified.py:751-768 replicates the prompt x beam BEFORE the prefill)."""
import pytest
import torch

from conftest import pcy_disable, rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
KW = dict(vocab=4096, d=4096, n_layers=2, n_heads=32, n_kv_heads=8, ffn=14336)


@pytest.fixture(scope="module")
def eng2():
    from procyon_amd import synth
    from procyon_amd.engine import LlamaConfig, LlamaEngine
    return LlamaEngine(synth.llama_state_dict(**KW), LlamaConfig(**KW, max_pos=2048))


@pytest.mark.parametrize("B,T,N", [(9, 300, 7), (10, 100, 8), (11, 77, 5), (12, 64, 5), (13, 50, 5), (15, 40, 5), (16, 90, 5), (10, 800, 5), (16, 1100, 4),
                                    (17, 300, 5), (20, 200, 6), (24, 33, 4), (32, 300, 5), (32, 1100, 3)])
def test_decode_mb_step_bit_identical(eng2, monkeypatch, B, T, N):
    """decode_step_mb_kernel (all layers of a B-row step in one launch: the batched path's work items -- skinny-MFMA GEMV tiles with their K
    splits, the decode attention, the K-split finish + RMSNorm -- as phases of one persistent kernel, flag hand-overs, weight rings that run
    ahead across the phases) against the launch-per-stage step (PCY_DISABLE=decode_mb_step: seven launches per layer): logits of every step,
    tokens and the appended K / V rows must be BIT-identical, eager and under hipGraph replay (the second replayed run starts at the flag /
    epoch state the first one left).  9..15 rows: attention units of 64 columns; 16: 128 columns, one batch tile; 17..32 (PCY_MB_MAX=32: by
    default these stay on the launches, which are faster there): two batch tiles.  No watchdog (Context.sync raises)."""
    from procyon_amd.engine import Context, GenState
    monkeypatch.setenv("PCY_MB_MAX", "32")
    torch.manual_seed(B * 1000 + T)
    emb = (torch.randn(B, T, 4096) * 0.02).to(BF).cuda()

    def run(step, use_graph):
        pcy_disable(monkeypatch, "" if step else "decode_mb_step")
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


def test_decode_mb_switches_inside_one_process(eng2, monkeypatch):
    """The fused step is opt-in (PCY_MB_MAX=<rows>; off by default: the launches with the rotated K order are as fast).  Switching it on and off
    and between batch sizes inside one process keeps the launches' bits: a flag holds the epoch of the step that raised it, the epoch word only
    counts fused steps, and a captured step is keyed on the mode."""
    from procyon_amd.engine import Context, GenState
    torch.manual_seed(6)

    def run(B, T, mb_max):
        if mb_max is None:
            monkeypatch.delenv("PCY_MB_MAX", raising=False)
        else:
            monkeypatch.setenv("PCY_MB_MAX", str(mb_max))
        pcy_disable(monkeypatch)
        emb = (torch.randn(B, T, 4096, generator=torch.Generator().manual_seed(B)) * 0.02).to(BF).cuda()
        cache = eng2.new_cache(B, T + 6)
        st = GenState(B, KW["vocab"], 6, "cuda")
        logits, _ = eng2.prefill(emb, None, cache, "last")
        st.logits.copy_(logits); st.pos.fill_(T)
        eng2.pick(cache, st, B, advance_pos=False)
        eng2.greedy_steps(cache, st, B, 3)
        Context.get().sync()
        return st.logits.cpu().clone()

    ref10, ref20, ref12 = run(10, 90, None), run(20, 60, None), run(12, 70, None)       # the launches
    for _ in range(2):
        assert torch.equal(run(10, 90, 16), ref10)      # fused
        assert torch.equal(run(20, 60, 16), ref20)      # launches (above the limit)
        assert torch.equal(run(20, 60, 32), ref20)      # fused, two batch tiles
        assert torch.equal(run(12, 70, 32), ref12)
        assert torch.equal(run(10, 90, None), ref10)


def test_decode_mb_beam10_loop_bit_identical(eng2, monkeypatch):
    """The reference's production call: diverse beam search, beam 10 in groups of 2 (scripts/caption_bulk.py:193-194, :130) = a 10-row
    decode step + pcy_beam_step + the K/V reorder of every step, as the replayed chain (pcy_llama_beam_steps).  One launch per step vs the
    seven launches per layer: tokens, running scores, the parent chain and the logits record equal."""
    from procyon_amd.engine import BeamState, Context, GenState
    torch.manual_seed(10)
    T, steps, beam, group = 120, 14, 10, 2
    emb = (torch.randn(1, T, 4096) * 0.02).to(BF).cuda().repeat(beam, 1, 1).contiguous()

    def run(step):
        pcy_disable(monkeypatch, "" if step else "decode_mb_step")
        cache = eng2.new_cache(beam, T + steps + 2)
        logits, _ = eng2.prefill(emb, None, cache, "last")
        bs = BeamState(1, beam, steps, 2, prompt_len=T, device="cuda")
        st = GenState(beam, KW["vocab"], 1, "cuda")
        st.pos, st.next_tok = bs.pos, bs.next_tok
        st.c.pos, st.c.next_tok = bs.pos.data_ptr(), bs.next_tok.data_ptr()
        rec = torch.zeros(steps, beam, KW["vocab"], dtype=BF, device="cuda")
        rec[0].copy_(logits)
        eng2.beam_step(logits.contiguous(), bs, group, 0.8)
        eng2.kv_reorder(cache, bs.src, T)
        eng2.beam_steps(cache, st, bs, group, 0.8, rec, steps - 1)
        out, n = bs.tokens()
        Context.get().sync()
        return out.cpu(), bs.cur.cpu().clone(), bs.anc[:n].cpu().clone(), rec[:n].cpu()

    ref, got = run(False), run(True)
    for x, y in zip(got, ref):
        assert torch.equal(x, y)


def test_beam_search_prefills_each_prompt_once(monkeypatch):
    """`_generate_beam_search` (model_unified.py:740-842) replicates every prompt x beam before the prefill (:751-752).  The mirror prefills
    each prompt once into row b of the B x beam-row cache, copies its K / V rows to the rows of its beams (pcy_kv_reorder with the source map
    r -> r // beam) and repeats its last-row logits.  Against the replicated prefill (PCY_DISABLE=beam_prefill_once), through the public
    `generate(method="beam")` on a two-prompt batch (ragged: the shorter prompt is left-padded): the step-0 logits of a beam are those of its
    prompt -- bit for bit when both prefills take the same GEMM kernels, to bf16 noise otherwise (a beam x larger token count may select other
    tiles) -- and then the whole search is the same search (tokens, scores, logits record)."""
    from procyon_amd import synth
    from procyon_amd import synthetic_model as SM
    from procyon_amd.engine import Context
    model = SM.build("small", device="cuda", max_new_tokens=32)
    prot = synth.protein_tokens([90, 41], seed=3)
    instr = ["w5 w6 <|protein|> w7 [ANSWER]", "w1 w2 <|protein|> and w3 w4 w8 w9 w2 w3 [ANSWER]"]
    inputs = lambda: {"data": {"seq": prot, "seq_idx": torch.arange(2), "text": [], "drug": None},
                      "input": {"seq": [[0], [1]], "text": [[], []], "drug": None},
                      "target": {"seq": None, "text": None, "drug": None}, "instructions": list(instr)}
    kw = dict(max_len=10, method="beam", beam_size=6, beam_group_size=2, diversity_penalty=0.8)

    def run(once):
        pcy_disable(monkeypatch, "" if once else "beam_prefill_once")
        tokens, scores, logits, _ = model.generate(inputs(), **kw)
        Context.get().sync()
        return tokens, scores, logits

    o1, c1, l1 = run(True)
    o0, c0, l0 = run(False)
    assert o1.shape == o0.shape and l1.shape == l0.shape and o1.shape[:2] == (2, 6)
    e0 = rel_err(l1[:, :, 0].float(), l0[:, :, 0].float())
    assert e0 < 2e-2, e0
    if torch.equal(l1[:, :, 0], l0[:, :, 0]):           # same prefill bits -> the same search
        assert torch.equal(o1, o0) and torch.equal(c1, c0) and torch.equal(l1, l0)
    else:                                               # other GEMM tiles in the beam x larger prefill: the same search up to ties inside the noise
        assert (o1[:, :, 0] == o0[:, :, 0]).all()
        assert rel_err(c1.float(), c0.float()) < 5e-2


def test_split_geometry_batched_decode_takes_the_mfma_path(monkeypatch):
    """ProCyon-Split's decoder (Llama-2-7B geometry: ffn 11008 = 86 x 128, 32 kv heads; /root/reference/README.md:50-51,
    /root/reference/procyon/training/training_args_IT.py:129-134) at beam-search batch sizes: until round 6 a K that is no multiple of 512 sent
    EVERY projection of a >= 4-row step to the streaming kernel in groups of 4 rows (three passes over the weights at 10 rows).  Now the
    skinny-MFMA GEMVs step in units of 128 (the down projection splits K in two halves of 43 steps).  A 10-row step against (a) the same rows
    decoded one at a time (the 1-row arithmetic: bf16 noise apart) and (b) the oracle on the CPU (2 layers)."""
    from oracle import llama_ref as LR
    from procyon_amd import synth
    from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
    kw = dict(vocab=2048, d=4096, n_layers=2, n_heads=32, n_kv_heads=32, ffn=11008)
    sd = synth.llama_state_dict(**kw)
    eng = LlamaEngine(sd, LlamaConfig(**kw, max_pos=512))
    pcy_disable(monkeypatch)
    torch.manual_seed(4)
    B, T, N = 10, 24, 3
    emb = (torch.randn(B, T, 4096) * 0.02).to(BF)

    def run(rows):
        n = len(rows)
        e = emb[rows].cuda().contiguous()
        cache = eng.new_cache(n, T + N + 2)
        st = GenState(n, kw["vocab"], N + 2, "cuda")
        logits, _ = eng.prefill(e, None, cache, "last")
        st.logits.copy_(logits); st.pos.fill_(T)
        eng.pick(cache, st, n, advance_pos=False)
        out = [logits.float().cpu()]
        for _ in range(N):
            eng.greedy_steps(cache, st, n, 1)
            out.append(st.logits.float().cpu())
        Context.get().sync()
        return torch.stack(out), st.tokens_out[:, :N + 1].cpu()

    lg10, tok10 = run(list(range(B)))
    assert torch.isfinite(lg10).all()
    for b in (0, 7):
        lg1, tok1 = run([b])
        same = bool((tok1[0] == tok10[b]).all())
        upto = N + 1 if same else int((tok1[0] != tok10[b]).nonzero()[0])     # (a different greedy token ends the comparable prefix)
        for s_ in range(min(upto + 1, N + 1)):
            assert rel_err(lg10[s_, b], lg1[s_, 0]) < 2e-2, (b, s_)
    # the oracle: prefill + the same greedy tokens teacher-forced
    geom = LR.LlamaGeom(**kw, max_pos=512)
    r = LR.llama_forward(sd, geom, inputs_embeds=emb, attn_mask=torch.ones(B, T), logits_rows="last")
    assert rel_err(lg10[0], r["logits"][:, -1].float()) < 2e-2
    past = r["past_kv"]
    for s_ in range(N):
        r = LR.llama_forward(sd, geom, input_ids=tok10[:, s_:s_ + 1].long(), attn_mask=None, past_kv=past, logits_rows="last")
        past = r["past_kv"]
        assert rel_err(lg10[s_ + 1], r["logits"][:, -1].float()) < 2e-2, s_


def test_beam_search_reorders_only_the_suffix(monkeypatch):
    """The beams of a prompt share its prefix rows, so the per-step K / V reorder moves slots [T, t) only (pcy_kv_reorder_range;
    PCY_DISABLE=beam_kv_suffix: every slot, what /root/reference/procyon/model/model_unified.py:830-832 does).  Both forms, through the replayed
    chain and through the four calls per step, on a two-prompt ragged batch: tokens, scores and the logits record EQUAL."""
    from procyon_amd import synth
    from procyon_amd import synthetic_model as SM
    from procyon_amd.engine import Context
    model = SM.build("small", device="cuda", max_new_tokens=48)
    prot = synth.protein_tokens([90, 41], seed=3)
    instr = ["w5 w6 <|protein|> w7 [ANSWER]", "w1 w2 <|protein|> and w3 w4 w8 w9 w2 w3 [ANSWER]"]
    inputs = lambda: {"data": {"seq": prot, "seq_idx": torch.arange(2), "text": [], "drug": None},
                      "input": {"seq": [[0], [1]], "text": [[], []], "drug": None},
                      "target": {"seq": None, "text": None, "drug": None}, "instructions": list(instr)}
    outs = []
    for beam, group, max_len in ((6, 2, 21), (10, 2, 12), (5, 5, 9)):
        kw = dict(max_len=max_len, method="beam", beam_size=beam, beam_group_size=group, diversity_penalty=0.8)
        res = []
        for off in ("", "beam_kv_suffix", "beam_graph", "beam_graph,beam_kv_suffix"):
            pcy_disable(monkeypatch, *[o for o in off.split(",") if o])
            res.append(model.generate(inputs(), **kw)[:3])
            Context.get().sync()
        for r in res[1:]:
            for x, y in zip(r, res[0]):
                assert torch.equal(x, y), (beam, group)


@pytest.mark.parametrize("T,N", [(40, 6), (300, 6), (764, 8), (1100, 4)])
def test_split_decode_step_bit_identical(monkeypatch, T, N):
    """ProCyon-Split's decoder (Llama-2-7B geometry: 32 kv heads, ffn 11008; /root/reference/README.md:50-51) at one row: every layer of a decode
    step in ONE launch (decode_step_mha_kernel, pcy_decode_mha.hip) against one launch per layer (PCY_DISABLE=decode_step) and against the
    launch-per-stage step (PCY_DISABLE=decode_step,decode_layer: streaming GEMVs in the rotated k order, 64-column attention workgroups), launched
    one by one and replayed: logits, tokens and the appended K / V rows EQUAL -- across the key-split threshold of the attention (768 keys)."""
    from procyon_amd import synth
    from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
    kw = dict(vocab=4096, d=4096, n_layers=2, n_heads=32, n_kv_heads=32, ffn=11008)
    eng = LlamaEngine(synth.llama_state_dict(**kw), LlamaConfig(**kw, max_pos=2048))
    torch.manual_seed(T)
    emb = (torch.randn(1, T, 4096) * 0.02).to(BF).cuda()

    def run(off, use_graph):
        pcy_disable(monkeypatch, *[o for o in off.split(",") if o])
        cache = eng.new_cache(1, T + N + 2)
        st = GenState(1, kw["vocab"], N + 2, "cuda")
        logits, _ = eng.prefill(emb, None, cache, "last")
        st.logits.copy_(logits); st.pos.fill_(T)
        eng.pick(cache, st, 1, advance_pos=False)
        out = []
        for _ in range(N):
            eng.greedy_steps(cache, st, 1, 1, use_graph=use_graph)
            out.append(st.logits.clone())
        Context.get().sync()
        return torch.stack(out).cpu(), st.tokens_out[:, :N + 1].cpu(), cache.k[:, :, :, T:T + N].cpu(), cache.v[:, :, :, T:T + N].cpu()

    ref = run("decode_step,decode_layer", False)
    assert torch.isfinite(ref[0].float()).all()
    for off, g in (("", False), ("", True), ("", True), ("decode_step", False), ("decode_step", True), ("decode_step,decode_layer", True)):
        got = run(off, g)
        for x, y in zip(got, ref):
            assert torch.equal(x, y), (off, g)


def test_split_decode_step_against_the_oracle(monkeypatch):
    """The one-launch Split step held to the oracle directly (2 layers, teacher-forced on the step's own greedy tokens), not only to its twin:
    /root/reference/procyon/model/pmc_llama.py:571-588 restated in oracle/llama_ref.py.  Bar: the bf16 pipeline's own noise (rel. error of the
    logits row < 2e-2, as the batched Split test above; full depth: f6, tests/test_gpu_fulldepth.py)."""
    from oracle import llama_ref as LR
    from procyon_amd import synth
    from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine
    kw = dict(vocab=2048, d=4096, n_layers=2, n_heads=32, n_kv_heads=32, ffn=11008)
    sd = synth.llama_state_dict(**kw)
    eng = LlamaEngine(sd, LlamaConfig(**kw, max_pos=512))
    pcy_disable(monkeypatch)
    torch.manual_seed(11)
    T, N = 37, 5
    emb = (torch.randn(1, T, 4096) * 0.02).to(BF)
    cache = eng.new_cache(1, T + N + 2)
    st = GenState(1, kw["vocab"], N + 2, "cuda")
    logits, _ = eng.prefill(emb.cuda(), None, cache, "last")
    st.logits.copy_(logits); st.pos.fill_(T)
    eng.pick(cache, st, 1, advance_pos=False)
    out = [logits.float().cpu()]
    for _ in range(N):
        eng.greedy_steps(cache, st, 1, 1)
        out.append(st.logits.float().cpu())
    Context.get().sync()
    tok = st.tokens_out[:, :N + 1].cpu()
    geom = LR.LlamaGeom(**kw, max_pos=512)
    r = LR.llama_forward(sd, geom, inputs_embeds=emb, attn_mask=torch.ones(1, T), logits_rows="last")
    assert rel_err(out[0], r["logits"][:, -1].float()) < 2e-2
    past = r["past_kv"]
    for s_ in range(N):
        r = LR.llama_forward(sd, geom, input_ids=tok[:, s_:s_ + 1].long(), attn_mask=None, past_kv=past, logits_rows="last")
        past = r["past_kv"]
        assert rel_err(out[s_ + 1], r["logits"][:, -1].float()) < 2e-2, s_
