"""GPU: the fp8 (OCP e4m3) weight path of BASELINE.json configs[4] -- pcy_quant_rows_fp8, pcy_gemm_fp8 and the fp8 prefill --
against oracle/fp8_ref.py.  The reference has no fp8 path: kernel-level parity is against the oracle's definition (bit-exact
for the quantiser, the strict bf16 bar for the GEMM), model-level parity is agreement with the pinned bf16 path."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_bf16_close, pcy_disable, rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ctx():
    from procyon_amd.engine import Context
    return Context.get()


@pytest.mark.parametrize("rows,K,scale", [(5, 128, 1.0), (300, 4096, 0.02), (33, 14336, 3.0), (2, 2056, 1e-3)])
def test_quant_rows_bit_exact(rows, K, scale):
    from oracle import fp8_ref as F8
    torch.manual_seed(rows)
    x = (torch.randn(rows, K) * scale).to(BF)
    x[rows // 2] = 0
    x[0, :8] = torch.tensor([448.0, -448.0, 1e-8, 0.0, 3.0, -2.5, 0.0625, 240.0]).to(BF)
    q_ref, s_ref = F8.quant_rows(x)
    q, s = _ctx().quant_rows_fp8(x.cuda())
    assert torch.equal(s.cpu(), s_ref)
    assert torch.equal(q.cpu(), q_ref.view(torch.uint8))


@pytest.mark.parametrize("M,N,K,epi", [(300, 512, 256, "store"), (257, 384, 1280, "resid"), (1000, 1024, 4096, "store"),
                                       (64, 512, 14336, "resid"), (130, 1024, 512, "swiglu"), (5, 4096, 128, "store")])
def test_gemm_fp8_matches_oracle(M, N, K, epi):
    from oracle import fp8_ref as F8
    from procyon_amd import _lib as L
    from procyon_amd.engine import interleave_gate_up
    torch.manual_seed(M + N)
    x = (torch.randn(M, K) * 0.5).to(BF)
    w = (torch.randn(N, K) * 0.03).to(BF)
    ctx = _ctx()
    if epi == "swiglu":
        gate, up = w[:N // 2], w[N // 2:]
        g = F8.linear_fp8(x, gate)
        u = F8.linear_fp8(x, up)
        ref = F.silu(g) * u
        wp = interleave_gate_up(gate.cuda(), up.cuda())
        q8, sw = ctx.quant_rows_fp8(wp)
        a8, sa = ctx.quant_rows_fp8(x.cuda())
        out = ctx.gemm_fp8(a8, sa, q8, sw, epi=L.EPI_SWIGLU)
        # 1-ulp flips of gate and up compound through silu and the product; the MX MFMA does not add its 128 products as a
        # correctly rounded fp32 sum, so flips are more frequent than on the bf16 path (~0.7 % per GEMM output, 2 % here)
        assert_bf16_close(out.cpu(), ref, "swiglu", ulps=3, max_frac=0.05)
        return
    q8, sw = ctx.quant_rows_fp8(w.cuda())
    a8, sa = ctx.quant_rows_fp8(x.cuda())
    y = F8.linear_fp8(x, w)
    if epi == "resid":
        res = torch.randn(M, N).to(BF)
        out = ctx.gemm_fp8(a8, sa, q8, sw, resid=res.cuda(), epi=L.EPI_RESID)
        assert_bf16_close(out.cpu(), res + y, "resid", inter=y)
    else:
        out = ctx.gemm_fp8(a8, sa, q8, sw)
        assert_bf16_close(out.cpu(), y, "store")


def test_gemm_fp8_asymmetric_identity():
    """A = scaled identity rows, asymmetric W: catches a transposed / permuted output map that random data would hide."""
    ctx = _ctx()
    K = N = 256
    x = torch.zeros(256, K)
    x[torch.arange(256), torch.arange(256)] = 2.0
    # integers in [-7, 7], every row reaching 7: scale = 7/448 = 2^-6, so the codes are the integers times 64 -- exact in e4m3
    w = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 4) % 15 - 7.0
    a8, sa = ctx.quant_rows_fp8(x.to(BF).cuda())
    q8, sw = ctx.quant_rows_fp8(w.to(BF).cuda())
    out = ctx.gemm_fp8(a8, sa, q8, sw).cpu().float()
    assert torch.equal(out, (2.0 * w.T))


def _llama(kw, fp8=True):
    from oracle import llama_ref as LR
    from procyon_amd import synth
    from procyon_amd.engine import LlamaConfig, LlamaEngine
    sd = synth.llama_state_dict(**kw)
    eng = LlamaEngine({k: v.clone() for k, v in sd.items()}, LlamaConfig(**kw, max_pos=512), fp8_prefill=fp8)
    return sd, LR, eng


def test_llama_small_fp8_prefill_matches_oracle_and_decode_stays_bf16():
    kw = dict(vocab=320, d=256, n_layers=2, n_heads=4, n_kv_heads=2, ffn=512)
    sd, LR, eng = _llama(kw)
    torch.manual_seed(7)
    B, T = 3, 37
    emb = (torch.randn(B, T, 256) * 0.02).to(BF)
    mask = torch.ones(B, T)
    mask[1, :5] = 0
    ref = LR.llama_forward(sd, LR.LlamaGeom(**kw, weights="fp8"), inputs_embeds=emb, attn_mask=mask, logits_rows="last")
    ref16 = LR.llama_forward(sd, LR.LlamaGeom(**kw), inputs_embeds=emb, attn_mask=mask, logits_rows="last")
    cache = eng.new_cache(B, T + 4)
    logits, hidden = eng.prefill(emb.cuda(), mask, cache, "last", want_hidden=True)
    e8, e16 = rel_err(logits.cpu(), ref["logits"][:, -1]), rel_err(logits.cpu(), ref16["logits"][:, -1])
    eq = rel_err(ref["logits"][:, -1], ref16["logits"][:, -1])
    print(f"fp8 prefill: vs fp8 oracle {e8:.2e}, vs bf16 oracle {e16:.2e} (fp8 oracle vs bf16 oracle {eq:.2e})")
    # an activation one bf16 ulp apart can land in a neighbouring e4m3 bucket (a 6-12 % step; oracle/fp8_ref.py "Sensitivity"),
    # so the multi-layer bar is stated against the quantisation distance itself: the GPU must sit much closer to the fp8
    # oracle than fp8 sits to bf16
    assert e8 < 0.5 * eq and e8 < e16
    k_ref = torch.stack([kv[0] for kv in ref["past_kv"]])
    k16 = torch.stack([kv[0] for kv in ref16["past_kv"]])
    assert rel_err(cache.k[:, :B, :, :T].cpu(), k_ref) < 0.5 * rel_err(k_ref, k16)
    # switching the fp8 path off gives the pinned bf16 prefill again
    eng.set_fp8(False)
    logits16, _ = eng.prefill(emb.cuda(), mask, eng.new_cache(B, T + 4), "last")
    assert rel_err(logits16.cpu(), ref16["logits"][:, -1]) < 5e-3


def test_llama3_8b_layer_fp8_full_width():
    """One Llama-3-8B-geometry layer (d 4096, F 14336, GQA 32/8) on the fp8 path against the oracle: exercises the K = 4096 /
    14336 fp8 GEMMs with the SwiGLU and residual epilogues at full width."""
    kw = dict(vocab=512, d=4096, n_layers=1, n_heads=32, n_kv_heads=8, ffn=14336)
    sd, LR, eng = _llama(kw)
    torch.manual_seed(8)
    B, T = 2, 48
    emb = (torch.randn(B, T, 4096) * 0.02).to(BF)
    ref = LR.llama_forward(sd, LR.LlamaGeom(**kw, weights="fp8"), inputs_embeds=emb, attn_mask=torch.ones(B, T), want_hidden=True)
    cache = eng.new_cache(B, T)
    _, hidden = eng.prefill(emb.cuda(), None, cache, "last", want_hidden=True)
    ref16 = LR.llama_forward(sd, LR.LlamaGeom(**kw), inputs_embeds=emb, attn_mask=torch.ones(B, T))
    e, eq = rel_err(hidden.cpu(), ref["last_hidden"]), rel_err(ref["last_hidden"], ref16["last_hidden"])
    print(f"llama-3-8b layer fp8: gpu-vs-oracle {e:.2e} (fp8 oracle vs bf16 oracle {eq:.2e})")
    assert e < 0.5 * eq


def test_pair_scoring_fp8_agrees_with_bf16():
    """configs[4] in miniature: six-slot QA prompts scored through UnifiedProCyon.forward on the fp8 path vs the (pinned) bf16
    path of the same model: P(yes), P(no) at the last [ANSWER] agree closely and the yes/no decision is the same."""
    from procyon_amd import synth
    from procyon_amd import synthetic_model as SM
    model = SM.build("small", device="cuda", max_new_tokens=8)
    prot = synth.protein_tokens([80, 12, 80, 9, 80, 15, 11, 14], seed=5)
    tmpl = ("w1 <|protein|> binds <|protein|> ? [ANSWER] yes w2 <|protein|> binds <|protein|> ? [ANSWER] no "
            "w3 <|protein|> binds <|protein|> ? [ANSWER]")
    n = 8
    slots = [[0, 1, 2, 3, 4, 5 + (i % 3)] for i in range(n)]
    inp = lambda: {"data": {"seq": prot, "seq_idx": torch.arange(prot.shape[0]), "text": [], "drug": None},
                   "input": {"seq": slots, "text": [[] for _ in range(n)], "drug": None},
                   "target": {"seq": None, "text": None, "drug": None}, "instructions": [tmpl] * n}
    p16 = model.forward(inp(), retrieval=False)["outputs"].answer_logits[:, 0].float().softmax(-1).cpu()
    model.text_encoder.engine.quantize_fp8()
    p8 = model.forward(inp(), retrieval=False)["outputs"].answer_logits[:, 0].float().softmax(-1).cpu()
    model.text_encoder.engine.set_fp8(False)
    y16, n16, y8, n8 = p16[:, model.yes_token], p16[:, model.no_token], p8[:, model.yes_token], p8[:, model.no_token]
    print("abs dP(yes)", (y8 - y16).abs().max().item(), "P(yes) bf16", y16[:3].tolist())
    assert torch.allclose(y8, y16, rtol=0.2, atol=1e-6) and torch.allclose(n8, n16, rtol=0.2, atol=1e-6)
    assert torch.equal(y8 > n8, y16 > n16)
    assert not torch.equal(p8, p16)        # the fp8 path really ran


def test_fp8_prefill_fused_norm_quant_bit_identical(monkeypatch):
    """RMSNorm + per-token quantisation in one pass (default) vs the two launches (PCY_DISABLE=fp8_fused_norm): same statistic order,
    same rounding points, same scale rule -> bit-identical logits, hidden states and K/V at full width (d = 4096, one group of
    2048 elements per thread pass) and at the small geometry."""
    for kw, B, T in ((dict(vocab=512, d=4096, n_layers=1, n_heads=32, n_kv_heads=8, ffn=14336), 2, 40),
                     (dict(vocab=320, d=256, n_layers=2, n_heads=4, n_kv_heads=2, ffn=512), 3, 37)):
        sd, LR, eng = _llama(kw)
        torch.manual_seed(9)
        emb = (torch.randn(B, T, kw["d"]) * 0.02).to(BF).cuda()
        outs = []
        for fused in ("1", "0"):
            pcy_disable(monkeypatch, "" if fused == "1" else "fp8_fused_norm")
            cache = eng.new_cache(B, T)
            logits, hidden = eng.prefill(emb, None, cache, "last", want_hidden=True)
            outs.append((logits.cpu(), hidden.cpu(), cache.k.cpu().clone(), cache.v.cpu().clone()))
        for a, b in zip(*outs):
            assert torch.equal(a, b)


def test_fp8_error_growth_per_layer_is_a_random_walk():
    """Round-2 review item 6: the answer-row logits of the fp8 path sit 0.66 (relative) from the bf16 path after 32 random-init
    layers -- W8A8 noise or a kernel defect?  Walk the layers (tools/fp8_layer_walk.py): the distance of the residual stream must
    (a) start at the one-layer quantisation distance the CPU oracle's fp8 path shows for the same weights, (b) grow like
    e_1 * sqrt(l) -- independent, equally sized kicks into an undamped residual stream -- with no jump at any layer, and (c) the
    GPU's fp8 states must stay much closer to the oracle's fp8 states than fp8 is to bf16."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fp8_layer_walk import walk
    r = walk(n_layers=32, oracle_layers=3, T=48, verbose=False)
    rows = r["rows"]
    e = [x["gpu_fp8_vs_bf16"] for x in rows]
    print("err(fp8, bf16) after layers 1, 2, 4, 8, 16, 31:", [f"{e[i]:.3f}" for i in (0, 1, 3, 7, 15, 30)], "| sqrt model:",
          [f"{rows[i]['sqrt_model']:.3f}" for i in (0, 1, 3, 7, 15, 30)], "| last-row logits", f"{r['last_row_logits']:.3f}")
    for i in range(3):          # (a), (c): the first layers against the oracle
        assert 0.6 < e[i] / rows[i]["oracle_fp8_vs_bf16"] < 1.6, (i, e[i], rows[i]["oracle_fp8_vs_bf16"])
        assert rows[i]["gpu_fp8_vs_oracle_fp8"] < 0.6 * rows[i]["oracle_fp8_vs_bf16"], i
    for i in range(1, len(e)):  # (b): sqrt growth within a factor, and no jump
        assert 0.5 < e[i] / rows[i]["sqrt_model"] < 1.8, (i + 1, e[i], rows[i]["sqrt_model"])
        assert e[i] < 1.35 * e[i - 1] + 0.02, (i + 1, e[i - 1], e[i])
