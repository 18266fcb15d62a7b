"""GPU parity at FULL DEPTH against a higher-precision truth (BASELINE configs[1] geometry: Llama-3-8B, 32 layers; ESM2-650M,
33 layers, one 1024-residue protein).

The fixtures (tests/golden/f1_llama8b_T{64,512}.npz, f2_esm650m_1024.npz; made by tests/golden/make_fulldepth.py in the build
container) hold, for the SAME seeded weights this test regenerates, the oracle's bf16 result (the reference's arithmetic) and
an fp32 evaluation of the same weights (the truth both bf16 pipelines approximate).  north_star's 1e-3 bar against a
bf16-materialised reference is not attainable by ANY second implementation (the bf16 oracle itself sits 5e-3..1.3e-2 from the
fp32 truth after 32 layers), so the assertion is the one that is: the HIP path is no further from the truth than the
reference's own bf16 path,

    err(HIP, fp32) <= 1.25 x err(oracle_bf16, fp32),

on the prefill's last-row logits and on 64 teacher-forced cached decode steps (Llama; argmax agreement with the truth is
asserted as a rate: agree(HIP, fp32) >= agree(oracle_bf16, fp32) - 2 of 65) and on the pooled / shared / soft-token
embeddings (ESM + projectors).

Round 4: DAMPED fixtures (f3 / f4: residual branches x 0.25) were built to get a regime in which the bf16 oracle agrees with the truth on
nearly every argmax; measured, they do not (57 / 65, see test_llama8b_damped_full_depth and tools/diag_bf16_floor.py), so token agreement
is asserted where it CAN hold for two bf16 pipelines -- on every step whose top-2 margin clears 4 x the bf16 logit noise -- and as a rate
against the oracle; every number of every test here goes into the parity report (conftest.record_parity -> gpurun_out/parity_report.json -> profiles/r04_parity_report.json)
and one PARITY line per test survives `pytest -q`.
"""
import pytest
import torch

from conftest import record_parity, rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
LLAMA = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
ESM = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
SLACK = 1.25


def _llama_engine(damped):
    """Llama-3-8B geometry on the CPU-seeded weights of the fixtures (the other full-size tests seed on the device)."""
    from procyon_amd import synth
    from procyon_amd.engine import LlamaConfig, LlamaEngine
    sd = synth.llama_state_dict(**LLAMA, workers=16)
    if damped:
        synth.damp_residual_branches(sd, 0.25)
    return LlamaEngine(sd, LlamaConfig(**LLAMA, max_pos=4096), free_source=True)


@pytest.fixture(scope="module")
def llama():
    return _llama_engine(False)


@pytest.fixture(scope="module")
def llama_damped():
    return _llama_engine(True)


def _llama_steps(eng, g, T):
    """prefill + teacher-forced cached decode steps on the fixture's tokens -> logits [65, V] fp32 (CPU), prefill hidden row"""
    from procyon_amd.engine import GenState
    ids, toks = g["ids"].long(), g["tokens"].long()
    nstep = toks.numel()                          # prefill + 64 decode steps
    cache = eng.new_cache(1, T + nstep + 1)
    logits, hidden = eng.prefill(eng.embed_tokens(ids), None, cache, "last", want_hidden=True)
    got = [logits[0].cpu()]
    st = GenState(1, LLAMA["vocab"], nstep + 1, "cuda")
    for s in range(1, nstep):                     # teacher-forced on the bf16 oracle's greedy tokens
        st.pos.fill_(T + s - 1)
        st.next_tok.copy_(toks[s - 1:s].to(torch.int32))
        eng.decode(cache, st, 1)
        got.append(st.logits[0].cpu())
    return torch.stack(got).float(), hidden[0, -1].cpu().float()


def _llama_stats(got, g):
    """per-step errors and agreement counts of HIP logits [65, V] against the fixture's bf16 oracle and fp32 truth (column subset)"""
    cols = g["cols"].long()
    truth, ref = g["logits_fp32"], g["logits_bf16"].float()
    nstep = got.shape[0]
    out = dict(nstep=nstep, e_hip_truth=[], e_ref_truth=[], e_hip_ref=[], agree_hip_truth=0, agree_ref_truth=0, agree_hip_ref=0, rows=[])
    for s in range(nstep):
        out["e_hip_truth"].append(rel_err(got[s, cols], truth[s]))
        out["e_ref_truth"].append(rel_err(ref[s], truth[s]))
        out["e_hip_ref"].append(rel_err(got[s, cols], ref[s]))
        am, am_t, am_r = int(got[s].argmax()), int(g["top_ids_fp32"][s, 0]), int(g["top_ids_bf16"][s, 0])
        out["agree_hip_truth"] += am == am_t
        out["agree_ref_truth"] += am_r == am_t
        out["agree_hip_ref"] += am == am_r
        out["rows"].append((am, am_t, am_r, float(g["top_vals_fp32"][s, 0] - g["top_vals_fp32"][s, 1])))
    return out


@pytest.mark.parametrize("T", [64, 512])
def test_llama8b_full_depth_vs_fp32_truth(llama, golden, T):
    g = golden(f"f1_llama8b_T{T}")
    got, h_hip = _llama_steps(llama, g, T)
    st = _llama_stats(got, g)
    nstep = st["nstep"]
    worst = 0.0
    for s in range(nstep):
        e_hip, e_ref, e_hr = st["e_hip_truth"][s], st["e_ref_truth"][s], st["e_hip_ref"][s]
        worst = max(worst, e_hip / e_ref)
        am, am_t, am_r, margin = st["rows"][s]
        if s < 4 or s % 16 == 0 or am != am_t:
            print(f"T={T} step {s}: err(HIP,fp32) {e_hip:.3e}  err(oracle_bf16,fp32) {e_ref:.3e} (all {LLAMA['vocab']} columns: "
                  f"{float(g['err_bf16_full'][s]):.3e})  err(HIP,oracle_bf16) {e_hr:.3e}  argmax HIP {am} / fp32 {am_t} / oracle {am_r}  "
                  f"fp32 top-2 margin {margin:.3e}")
        assert e_hip <= SLACK * e_ref, (s, e_hip, e_ref)
    # argmax agreement as a RATE over the prefill + 64 teacher-forced decode steps: the HIP path must agree with the fp32 truth as
    # often as the reference's own bf16 arithmetic does, within two steps (round-2 review: the per-step near-tie assertion could
    # not fail -- every fp32 top-2 margin of an untrained model lies inside the bf16 logits noise)
    print(f"T={T}: worst err(HIP,fp32)/err(oracle_bf16,fp32) = {worst:.3f}; argmax agreement over {nstep} steps: "
          f"HIP vs fp32 {st['agree_hip_truth']}/{nstep}, oracle_bf16 vs fp32 {st['agree_ref_truth']}/{nstep}, HIP vs oracle_bf16 {st['agree_hip_ref']}/{nstep}")
    mean = lambda v: sum(v) / len(v)
    record_parity(f"fulldepth/llama8b_random_init_T{T}", steps=nstep, err_hip_fp32_mean=mean(st["e_hip_truth"]), err_oracle_fp32_mean=mean(st["e_ref_truth"]),
                  err_hip_oracle_mean=mean(st["e_hip_ref"]), err_hip_oracle_max=max(st["e_hip_ref"]), worst_ratio_hip_over_oracle=worst,
                  agree_hip_fp32=st["agree_hip_truth"], agree_oracle_fp32=st["agree_ref_truth"], agree_hip_oracle=st["agree_hip_ref"])
    assert nstep >= 65
    assert st["agree_hip_truth"] >= st["agree_ref_truth"] - 2, (st["agree_hip_truth"], st["agree_ref_truth"])
    # final-normed hidden row of the prefill: all 4096 entries are in the fixture
    e_hip, e_ref = rel_err(h_hip, g["hidden_fp32"][0]), rel_err(g["hidden_bf16"][0].float(), g["hidden_fp32"][0])
    print(f"T={T} prefill hidden row: err(HIP,fp32) {e_hip:.3e}  err(oracle_bf16,fp32) {e_ref:.3e}")
    assert e_hip <= SLACK * e_ref


def _margin_conditioned(st, g):
    """Steps whose fp32 top-2 margin exceeds 4 x the per-logit noise of the bf16 pipeline (rms over the fixture's columns of
    oracle_bf16 - fp32 at that step): there two bf16 implementations MUST pick the same token (a flip needs a 2.8-sigma event on the
    difference of two logit errors).  Returns (n_clear, n_clear_agree_hip_oracle, n_clear_agree_hip_fp32)."""
    truth, ref = g["logits_fp32"], g["logits_bf16"].float()
    clear = agree_o = agree_t = 0
    for s_, (am, am_t, am_r, margin) in enumerate(st["rows"]):
        noise = float((ref[s_] - truth[s_]).pow(2).mean().sqrt())
        if margin >= 4.0 * noise:
            clear += 1
            agree_o += am == am_r
            agree_t += am == am_t
    return clear, agree_o, agree_t


@pytest.mark.parametrize("T", [64, 512])
def test_llama8b_argmax_agrees_wherever_the_margin_clears_the_bf16_noise(llama, golden, T):
    """The falsifiable form of north_star's "bit-exact argmax ids" (/root/reference/procyon/model/model_unified.py:892-906).  Two
    exact-class bf16 pipelines are as far from each other as from the fp32 truth (tools/diag_bf16_floor.py: the oracle against itself with
    another accumulation order, 1.4e-2 vs 1.5e-2, at every weight scale), so on Gaussian logits a flat "63 of 65" cannot hold for ANY
    second implementation -- the oracle itself agrees with the truth on 51-57 of 65.  Asserted instead: on EVERY step whose fp32 top-2
    margin exceeds 4 x the bf16 logit noise, HIP argmax == oracle argmax == fp32 argmax; and over all steps the HIP path agrees with
    the ORACLE at least as often as the oracle agrees with the truth, minus 2."""
    g = golden(f"f1_llama8b_T{T}")
    got, _ = _llama_steps(llama, g, T)
    st = _llama_stats(got, g)
    clear, ao, at = _margin_conditioned(st, g)
    record_parity(f"fulldepth/llama8b_random_init_T{T}_margin_conditioned", steps=st["nstep"], clear_margin_steps=clear, clear_agree_hip_oracle=ao,
                  clear_agree_hip_fp32=at, agree_hip_oracle=st["agree_hip_ref"], agree_oracle_fp32=st["agree_ref_truth"])
    assert clear >= 8, clear
    assert ao == clear and at == clear, (clear, ao, at)
    assert st["agree_hip_ref"] >= st["agree_ref_truth"] - 2, (st["agree_hip_ref"], st["agree_ref_truth"])


def test_llama8b_full_depth_256_steps(llama, golden):
    """Round 5 (review: 10-19 clear-margin steps of 65 are a thin sample, and the headline generates 256 tokens): fixture
    f1_llama8b_T512_N256 -- the T = 512 prompt with 256 teacher-forced cached decode steps (cache 512 .. 768: across the key-split
    threshold of the decode attention), bf16 oracle + fp32 truth in one causal pass.  Same assertions as at 64 steps: truth distance per
    step, argmax on EVERY clear-margin step (>= 40 of them), agreement rates."""
    T = 512
    g = golden("f1_llama8b_T512_N256")
    got, _ = _llama_steps(llama, g, T)
    st = _llama_stats(got, g)
    nstep = st["nstep"]
    mean = lambda v: sum(v) / len(v)
    clear, ao, at = _margin_conditioned(st, g)
    worst = max(a / b for a, b in zip(st["e_hip_truth"], st["e_ref_truth"]))
    record_parity("fulldepth/llama8b_random_init_T512_256_steps", steps=nstep, err_hip_fp32_mean=mean(st["e_hip_truth"]), err_oracle_fp32_mean=mean(st["e_ref_truth"]),
                  err_hip_oracle_mean=mean(st["e_hip_ref"]), err_hip_oracle_max=max(st["e_hip_ref"]), worst_ratio_hip_over_oracle=worst,
                  agree_hip_fp32=st["agree_hip_truth"], agree_oracle_fp32=st["agree_ref_truth"], agree_hip_oracle=st["agree_hip_ref"],
                  clear_margin_steps=clear, clear_agree_hip_oracle=ao, clear_agree_hip_fp32=at)
    assert nstep == 257
    for s in range(nstep):
        assert st["e_hip_truth"][s] <= SLACK * st["e_ref_truth"][s], (s, st["e_hip_truth"][s], st["e_ref_truth"][s])
    assert clear >= 40, clear
    assert ao == clear and at == clear, (clear, ao, at)
    assert st["agree_hip_truth"] >= st["agree_ref_truth"] - 6 and st["agree_hip_ref"] >= st["agree_ref_truth"] - 6   # (2 of 65 scaled to 257)


def test_llama8b_full_depth_left_padded_rows_compat_mode(llama, golden):
    """Reference quirks Q1 / Q2 at FULL depth (review: checked at toy geometry only): two rows, the second left-padded by 21 of 64 slots;
    prefill with the attention mask and positions arange(T) for every row (/root/reference/procyon/model/pmc_llama.py:546-588), then 16
    teacher-forced cached steps with NO mask -- the pad slots' K / V rows are attended -- at position = cache length
    (/root/reference/procyon/model/model_unified.py:769, :887).  Fixture f5: the bf16 oracle driven the same way and the same procedure in
    fp32.  Per row and step: err(HIP, fp32) <= 1.25 x err(oracle, fp32); argmax on every clear-margin (row, step); the rows decode as a
    2-row batch (the small-batch step, pcy_decode_nb.hip)."""
    from procyon_amd.engine import GenState
    g = golden("f5_llama8b_leftpad_T64")
    ids, mask, toks = g["ids"].long(), g["mask"].float(), g["tokens"].long()
    T, nstep = ids.shape[1], toks.shape[0]
    cache = llama.new_cache(2, T + nstep + 1)
    logits, _ = llama.prefill(llama.embed_tokens(ids), mask, cache, "last")
    got = [logits.cpu()]
    st = GenState(2, LLAMA["vocab"], nstep + 1, "cuda")            # keep = None: compat mode
    for s in range(1, nstep):
        st.pos.fill_(T + s - 1)
        st.next_tok.copy_(toks[s - 1].to(torch.int32))
        llama.decode(cache, st, 2)
        got.append(st.logits.cpu())
    got = torch.stack(got).float()                                 # [17, 2, V]
    cols = g["cols"].long()
    truth, ref = g["logits_fp32"], g["logits_bf16"].float()
    e_ht, e_rt, e_hr, clear, clear_ok, agree_hr, agree_rt = [], [], [], 0, 0, 0, 0
    for s in range(nstep):
        for b in range(2):
            e_ht.append(rel_err(got[s, b, cols], truth[s, b])); e_rt.append(rel_err(ref[s, b], truth[s, b])); e_hr.append(rel_err(got[s, b, cols], ref[s, b]))
            am, am_t, am_r = int(got[s, b].argmax()), int(g["top_ids_fp32"][s, b, 0]), int(g["top_ids_bf16"][s, b, 0])
            agree_hr += am == am_r
            agree_rt += am_r == am_t
            noise = float((ref[s, b] - truth[s, b]).pow(2).mean().sqrt())
            if float(g["top_vals_fp32"][s, b, 0] - g["top_vals_fp32"][s, b, 1]) >= 4.0 * noise:
                clear += 1
                clear_ok += (am == am_r) and (am == am_t)
    mean = lambda v: sum(v) / len(v)
    record_parity("fulldepth/llama8b_leftpad_two_rows_compat", rows_x_steps=2 * nstep, err_hip_fp32_mean=mean(e_ht), err_oracle_fp32_mean=mean(e_rt),
                  err_hip_oracle_mean=mean(e_hr), err_hip_oracle_max=max(e_hr), worst_ratio_hip_over_oracle=max(a / b for a, b in zip(e_ht, e_rt)),
                  agree_hip_oracle=agree_hr, agree_oracle_fp32=agree_rt, clear_margin=clear, clear_agree=clear_ok)
    for a, b in zip(e_ht, e_rt):
        assert a <= SLACK * b, (a, b)
    assert clear_ok == clear, (clear, clear_ok)
    assert agree_hr >= agree_rt - 2, (agree_hr, agree_rt)
    assert max(e_hr) < 0.12        # a wrong position / a dropped pad slot gives O(1)


def _rows_compat_run(llama, ids, mask, toks, rows, teacher):
    """prefill (left-padded rows, compat mode) + teacher-forced cached steps of a B-row batch; rows in `teacher` are fed the fixture's tokens
    (toks [nstep, len(teacher)]), every other row its own argmax.  -> logits [nstep, B, V] fp32 (CPU)"""
    from procyon_amd.engine import GenState
    B, T = ids.shape
    nstep = toks.shape[0]
    cache = llama.new_cache(B, T + nstep + 1)
    logits, _ = llama.prefill(llama.embed_tokens(ids), mask, cache, "last")
    got = [logits.float().cpu()]
    st = GenState(B, LLAMA["vocab"], nstep + 1, "cuda")            # keep = None: compat mode (every cached slot is attended)
    tsel = torch.tensor(teacher)
    for s in range(1, nstep):
        nxt = got[-1].argmax(-1).to(torch.int32)
        nxt[tsel] = toks[s - 1].to(torch.int32)
        st.pos.fill_(T + s - 1)
        st.next_tok.copy_(nxt)
        llama.decode(cache, st, B)
        got.append(st.logits.float().cpu())
    return torch.stack(got)


def _rows_compat_check(name, got, g, sel, min_clear):
    """got [nstep, n, V] (the fixture's rows `sel` of the batch) against the fixture's bf16 oracle and fp32 truth: per (row, step) the truth
    distance bar, argmax on every clear-margin (row, step), agreement rates; one PARITY record."""
    cols = g["cols"].long()
    nstep = got.shape[0]
    e_ht, e_rt, e_hr, clear, clear_ok, agree_hr, agree_rt = [], [], [], 0, 0, 0, 0
    for s in range(nstep):
        for j, b in enumerate(sel):
            truth, ref = g["logits_fp32"][s, b], g["logits_bf16"][s, b].float()
            e_ht.append(rel_err(got[s, j, cols], truth)); e_rt.append(rel_err(ref, truth)); e_hr.append(rel_err(got[s, j, cols], ref))
            am, am_t, am_r = int(got[s, j].argmax()), int(g["top_ids_fp32"][s, b, 0]), int(g["top_ids_bf16"][s, b, 0])
            agree_hr += am == am_r
            agree_rt += am_r == am_t
            noise = float((ref - truth).pow(2).mean().sqrt())
            if float(g["top_vals_fp32"][s, b, 0] - g["top_vals_fp32"][s, b, 1]) >= 4.0 * noise:
                clear += 1
                clear_ok += (am == am_r) and (am == am_t)
    mean = lambda v: sum(v) / len(v)
    record_parity(name, rows_x_steps=len(e_ht), err_hip_fp32_mean=mean(e_ht), err_oracle_fp32_mean=mean(e_rt), err_hip_oracle_mean=mean(e_hr),
                  err_hip_oracle_max=max(e_hr), worst_ratio_hip_over_oracle=max(a / b for a, b in zip(e_ht, e_rt)), agree_hip_oracle=agree_hr,
                  agree_oracle_fp32=agree_rt, clear_margin=clear, clear_agree=clear_ok)
    for a, b in zip(e_ht, e_rt):
        assert a <= SLACK * b, (a, b)
    assert clear >= min_clear and clear_ok == clear, (clear, clear_ok)
    assert agree_hr >= agree_rt - max(2, len(e_ht) // 32), (agree_hr, agree_rt)
    assert max(e_hr) < 0.15        # a wrong position / a dropped pad slot / a stale hand-over gives O(1)


@pytest.mark.parametrize("nrows", [10, 8, 5])
def test_llama8b_full_depth_ten_ragged_rows_compat_mode(llama, golden, monkeypatch, nrows):
    """Round 6: the batch of the reference's beam-10 callers (/root/reference/scripts/caption_bulk.py:193-194,
    /root/reference/procyon/evaluate/framework/procyon.py:72-76) held to the ORACLE at full depth, not only to its launch-per-stage twin:
    fixture f7 -- ten ragged left-padded rows (0 .. 27 pad slots of 64), prefill with the mask and positions arange(T), 16 teacher-forced
    cached steps with no mask at position = cache length (Q1 / Q2), bf16 oracle + the same procedure in fp32.  10 rows decode on the
    mid-batch step (pcy_decode_mb.hip); the first 8 / 5 rows as batches of their own on the small-batch step (pcy_decode_nb.hip: so far held
    to the oracle at 2 rows only).  Per (row, step): err(HIP, fp32) <= 1.25 x err(oracle, fp32); argmax on every clear-margin (row, step)."""
    monkeypatch.delenv("PCY_DISABLE", raising=False)
    monkeypatch.setenv("PCY_NB_MAX", "8")        # (8 rows default to the batched launches since round 6; here the fused step is what is checked)
    g = golden("f7_llama8b_rows10_T64")
    ids, mask, toks = g["ids"].long()[:nrows], g["mask"].float()[:nrows], g["tokens"].long()[:, :nrows]
    got = _rows_compat_run(llama, ids, mask, toks, nrows, list(range(nrows)))
    _rows_compat_check(f"fulldepth/llama8b_ragged_rows_compat_B{nrows}", got, g, list(range(nrows)), min_clear=nrows)


def test_llama8b_full_depth_config3_ragged_batch32(llama, golden, monkeypatch):
    """BASELINE configs[3] 4b AT FULL SIZE (review: oracle-checked in miniature only): 32 ragged rows (T uniform in [128, 512], seed 7,
    left-padded to 512), prefill + 8 cached steps of the 32-row batch in compat mode; fixture f8 holds the bf16 oracle and the fp32 truth for
    THREE sampled rows (0, 13, 31; rows are independent of their batch mates), teacher-forced, the other rows follow their own argmax."""
    monkeypatch.delenv("PCY_DISABLE", raising=False)
    g = golden("f8_config3_rows_T512")
    ids, mask, toks, rows = g["ids"].long(), g["mask"].float(), g["tokens"].long(), g["rows"].long().tolist()
    got = _rows_compat_run(llama, ids, mask, toks, 32, rows)[:, rows]
    _rows_compat_check("fulldepth/llama8b_config3_ragged_batch32_rows_0_13_31", got, g, [0, 1, 2], min_clear=3)


def test_config4_pair_answer_row_full_depth(llama, golden):
    """BASELINE configs[4] at full geometry, bf16, ONE pair end to end (review: the pair-scoring path was oracle-checked in miniature only):
    fixture f9 -- six <|protein|> slots (receptor 805 residues + three peptides, slots [0, 1, 0, 2, 0, 3]), a 437-token prompt ending in
    [ANSWER]; ESM2-650M (33 layers) -> mean pool -> 3-layer token projector -> splice -> Llama-3-8B prefill (32 layers) -> the answer row.
    Pooled embeddings, soft tokens and the answer row's logits: err(HIP, fp32) <= 1.25 x err(oracle_bf16, fp32); the yes / no logits within
    4 x the bf16 logit noise of the truth; P(yes) / P(no) through pcy_qa_probs equal the softmax of the engine's own row."""
    from procyon_amd import synth
    from procyon_amd.engine import Context, EsmConfig, EsmEngine, MlpEngine
    g = golden("f9_config4_pair")
    prot, slots, ids = g["protein_tokens"].long(), g["slots"].long(), g["ids"].long()
    PROT, ANSWER, YES, NO = [int(v) for v in g["special"]]
    esm = EsmEngine(synth.esm_state_dict(**ESM, workers=8), EsmConfig(**ESM))
    token = MlpEngine([(w.cuda(), b.cuda()) for w, b in synth.mlp_layers(3, 1280, 4096, 2560, 0)])
    z = esm.forward(prot)
    soft = token(z[slots.cuda()].contiguous())
    m = (ids == PROT).view(-1)
    smap = torch.full((ids.numel(),), -1, dtype=torch.int32)
    smap[m] = torch.arange(int(m.sum()), dtype=torch.int32)
    emb = llama.embed_tokens(ids, soft, smap)
    T = ids.shape[1]
    assert int(ids[0, -1]) == ANSWER
    logits, _ = llama.prefill(emb, None, llama.new_cache(1, T), "last")
    lg = logits[0].float().cpu()
    cols = g["cols"].long()
    truth, ref = g["logits_fp32"], g["logits_bf16"].float()
    rec = {}
    for name, got, key in (("pooled", z.float().cpu(), "pooled"), ("soft_tokens", soft.float().cpu(), "soft"), ("answer_logits", lg[cols], "logits")):
        e_hip, e_ref = rel_err(got, g[key + "_fp32"]), rel_err(g[key + "_bf16"].float(), g[key + "_fp32"])
        rec.update({f"{name}_err_hip_fp32": e_hip, f"{name}_err_oracle_fp32": e_ref, f"{name}_err_hip_oracle": rel_err(got, g[key + "_bf16"].float())})
        assert e_hip <= SLACK * e_ref, (name, e_hip, e_ref)
    noise = float((ref - truth).pow(2).mean().sqrt())
    iy, in_ = int((cols == YES).nonzero()[0]), int((cols == NO).nonzero()[0])
    for tok, i in ((YES, iy), (NO, in_)):
        assert abs(float(lg[tok]) - float(truth[i])) <= 4.0 * noise, (tok, float(lg[tok]), float(truth[i]), noise)
    _, yn, _ = Context.get().qa_probs(logits, YES, NO, want_probs=False)
    p = logits.float().softmax(-1)[0]
    assert abs(float(yn[0, 0]) - float(p[YES])) <= 2.0 ** -8 * float(p[YES]) + 1e-12 and abs(float(yn[0, 1]) - float(p[NO])) <= 2.0 ** -8 * float(p[NO]) + 1e-12
    rec.update(p_yes_hip=float(yn[0, 0]), p_yes_oracle=float(g["p_yes_no_bf16"][0]), p_yes_fp32=float(g["p_yes_no_fp32"][0]),
               p_no_hip=float(yn[0, 1]), p_no_oracle=float(g["p_yes_no_bf16"][1]), p_no_fp32=float(g["p_yes_no_fp32"][1]), logit_noise_rms=noise)
    record_parity("fulldepth/config4_pair_six_slots_answer_row", **rec)


def test_llama8b_damped_full_depth(llama_damped, golden):
    """The same comparison on the DAMPED model (residual branches x 0.25, fixture f3): round 3's review asked for a trained-like regime in
    which the bf16 oracle agrees with fp32 on >= 63 / 65 steps.  Built and measured: damping does NOT produce that regime (the oracle
    agrees on 57 / 65; err(oracle, fp32) 8.7e-2 as undamped -- the noise is per-layer materialisation, not amplification), so the
    assertions are the margin-conditioned ones plus the truth-distance bar; every number goes into the parity report."""
    T = 64
    g = golden("f3_llama8b_damped_T64")
    got, _ = _llama_steps(llama_damped, g, T)
    st = _llama_stats(got, g)
    nstep = st["nstep"]
    mean = lambda v: sum(v) / len(v)
    clear, ao, at = _margin_conditioned(st, g)
    record_parity("fulldepth/llama8b_damped_T64", steps=nstep, err_hip_fp32_mean=mean(st["e_hip_truth"]), err_oracle_fp32_mean=mean(st["e_ref_truth"]),
                  err_hip_oracle_mean=mean(st["e_hip_ref"]), err_hip_oracle_max=max(st["e_hip_ref"]),
                  agree_hip_fp32=st["agree_hip_truth"], agree_oracle_fp32=st["agree_ref_truth"], agree_hip_oracle=st["agree_hip_ref"],
                  clear_margin_steps=clear, clear_agree_hip_oracle=ao, clear_agree_hip_fp32=at)
    for s in range(nstep):
        assert st["e_hip_truth"][s] <= SLACK * st["e_ref_truth"][s], (s, st["e_hip_truth"][s], st["e_ref_truth"][s])
    assert ao == clear and at == clear, (clear, ao, at)
    assert st["agree_hip_truth"] >= st["agree_ref_truth"] - 2 and st["agree_hip_ref"] >= st["agree_ref_truth"] - 2


def _esm_outputs(golden_name, damped, golden):
    from procyon_amd import synth
    from procyon_amd.engine import EsmConfig, EsmEngine, MlpEngine
    g = golden(golden_name)
    sd = synth.esm_state_dict(**ESM, workers=8)
    if damped:
        synth.damp_residual_branches(sd, 0.25)
    eng = EsmEngine(sd, EsmConfig(**ESM))
    mk = lambda layers: MlpEngine([(w.cuda(), b.cuda()) for w, b in layers])
    shared, token = mk(synth.mlp_layers(3, 1280, 1280, 2560, 20)), mk(synth.mlp_layers(3, 1280, 4096, 2560, 0))
    toks = g["tokens"].long()
    hid = eng.hidden_states(toks)[0]
    z = eng.forward(toks)
    return g, {"hidden_rows": hid[g["rows"].long().cuda()].cpu(), "pooled": z[0].cpu(), "shared": shared(z)[0].cpu(), "soft_token": token(z)[0].cpu()}


@pytest.mark.parametrize("attn", ["fast", "exact"])
@pytest.mark.parametrize("damped", [False, True])
def test_esm650m_full_depth_vs_fp32_truth(golden, monkeypatch, attn, damped):
    """33-layer ESM2-650M, one 1024-residue protein, both attention kernels (PCY_ESM_ATTN: the default single-pass kernel and the
    exact two-pass one that reproduces the reference's rounding points), random-init and damped weights: no further from the fp32
    truth than the bf16 oracle, and err(HIP, oracle_bf16) <= 2 x err(oracle_bf16, fp32) on every output."""
    monkeypatch.setenv("PCY_ESM_ATTN", attn)
    g, out = _esm_outputs("f4_esm650m_damped_1024" if damped else "f2_esm650m_1024", damped, golden)
    rec = {}
    for k, v in out.items():
        e_hip, e_ref = rel_err(v.float(), g[k + "_fp32"]), rel_err(g[k + "_bf16"].float(), g[k + "_fp32"])
        e_hr = rel_err(v.float(), g[k + "_bf16"].float())
        print(f"esm650m {'damped' if damped else 'random-init'} attention={attn} {k}: err(HIP,fp32) {e_hip:.3e}  err(oracle_bf16,fp32) {e_ref:.3e}  "
              f"err(HIP,oracle_bf16) {e_hr:.3e}")
        rec.update({f"{k}_err_hip_fp32": e_hip, f"{k}_err_oracle_fp32": e_ref, f"{k}_err_hip_oracle": e_hr})
        assert e_hip <= SLACK * max(e_ref, 1e-3 if damped else 0.0), k
        # two bf16 pipelines sit about as far from each other as each from the truth (tools/diag_bf16_floor.py): bounded by that
        assert e_hr <= 2.0 * max(e_ref, 1e-3), (k, e_hr, e_ref)
    record_parity(f"fulldepth/esm650m_{'damped' if damped else 'random_init'}_attention_{attn}", **rec)
