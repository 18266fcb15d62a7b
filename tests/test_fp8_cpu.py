"""CPU: the fp8 oracle (oracle/fp8_ref.py).  The reference has no fp8 path, so these tests pin the oracle's DEFINITION
(OCP e4m3 via torch.float8_e4m3fn, per-row amax/448 scales, exact products) rather than a reference output."""
import torch
import torch.nn.functional as F

from oracle import fp8_ref as F8
from oracle import llama_ref as LR


def test_quant_rows_definition():
    torch.manual_seed(0)
    x = (torch.randn(7, 64) * 3).bfloat16()
    x[3] = 0                                            # all-zero row -> scale 1, codes 0
    q, s = F8.quant_rows(x)
    assert q.dtype == torch.float8_e4m3fn and s.dtype == torch.float32
    amax = x.float().abs().amax(-1)
    nz = [0, 1, 2, 4, 5, 6]
    assert torch.all(torch.log2(s) == torch.log2(s).round()) and s[3] == 1            # powers of two
    assert torch.all(amax[nz] / s[nz] <= 448.0) and torch.all(amax[nz] / s[nz] > 224.0)   # ... and the smallest that fit
    assert torch.all(q[3].float() == 0)
    big = torch.tensor([[448.0, 1.0], [449.0, 1.0], [2.0 ** -130, 0.0], [1.75, 1.0], [1.76, 0.5]])
    assert F8.quant_rows(big)[1].tolist() == [1.0, 2.0, 2.0 ** -126, 2.0 ** -8, 2.0 ** -7]
    # e4m3 keeps 3 mantissa bits: relative error of a normal value <= 2^-4, absolute <= half the smallest subnormal step
    d = F8.dequant(q, s)
    err = (d - x.float()).abs()
    assert torch.all(err <= x.float().abs() * 2.0 ** -4 + s[:, None] * 2.0 ** -10)


def test_e4m3_codes_are_ocp():
    # OCP e4m3fn landmarks (gfx950's format; MI300's fnuz variant differs): 448 = 0x7E, 1.0 = 0x38, 2^-9 = 0x01
    v = torch.tensor([448.0, 1.0, 2.0 ** -9, -0.5]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert v.tolist() == [0x7E, 0x38, 0x01, 0xB0]


def test_representable_values_round_trip_and_linear_is_exact_then():
    # inputs that are exactly scale * e4m3 values: quantisation is lossless and the fp8 linear equals the fp32 one
    torch.manual_seed(1)
    codes = torch.randint(0, 0x7F, (4, 128), dtype=torch.uint8)
    codes[:, 0] = 0x7E                                     # row maximum 448 * 2^-5 -> scale = 2^-5
    xa = (codes.view(torch.float8_e4m3fn).float() * 2.0 ** -5)
    wa = (torch.randint(0, 0x7F, (16, 128), dtype=torch.uint8))
    wa[:, 1] = 0x7E
    wa = wa.view(torch.float8_e4m3fn).float() * 2.0 ** -7
    q, s = F8.quant_rows(xa)
    assert torch.equal(F8.dequant(q, s), xa)
    y = F8.linear_fp8(xa, wa)
    assert torch.equal(y, F.linear(xa.double(), wa.double()).float())


def test_llama_oracle_fp8_switch_close_to_bf16_and_prefill_only():
    from procyon_amd import synth
    kw = dict(vocab=320, d=256, n_layers=2, n_heads=4, n_kv_heads=2, ffn=512)
    sd = synth.llama_state_dict(**kw)
    torch.manual_seed(2)
    emb = (torch.randn(2, 12, 256) * 0.02).bfloat16()
    ref = LR.llama_forward(sd, LR.LlamaGeom(**kw), inputs_embeds=emb, attn_mask=torch.ones(2, 12))
    f8 = LR.llama_forward(sd, LR.LlamaGeom(**kw, weights="fp8"), inputs_embeds=emb, attn_mask=torch.ones(2, 12))
    e = ((f8["logits"].float() - ref["logits"].float()).norm() / ref["logits"].float().norm()).item()
    assert 1e-3 < e < 0.15, e                              # a different (coarser) arithmetic, but the same function
    # cached decode steps stay bf16: same past -> identical logits under both switches
    tok = ref["logits"][:, -1].argmax(-1, keepdim=True)
    a = LR.llama_forward(sd, LR.LlamaGeom(**kw), input_ids=tok, past_kv=ref["past_kv"], logits_rows="last")
    b = LR.llama_forward(sd, LR.LlamaGeom(**kw, weights="fp8"), input_ids=tok, past_kv=ref["past_kv"], logits_rows="last")
    assert torch.equal(a["logits"], b["logits"])
