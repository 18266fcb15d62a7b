"""GPU: degenerate and boundary inputs of the path through the C ABI -- the shortest and longest sequences, single-token
prompts, one-token generations, capacity and argument errors (reported as error codes + text, never a crash), and the
consistency of the generation modes where they must coincide (beam 1 / group 1 == greedy)."""
import pytest
import torch

from conftest import parity_bar, rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
SM_LLAMA = dict(vocab=320, d=256, n_layers=2, n_heads=4, n_kv_heads=2, ffn=512)
SM_ESM = dict(d=128, n_layers=2, n_heads=2, ffn=256)


@pytest.mark.parametrize("attn", ["exact", "fast"])
def test_esm_shortest_and_longest_sequence_in_one_batch(attn, monkeypatch):
    from oracle import esm_ref as ER
    from oracle import procyon_ref as PR
    from procyon_amd import synth
    from procyon_amd.engine import EsmConfig, EsmEngine
    from conftest import assert_no_further_from_truth
    monkeypatch.setenv("PCY_ESM_ATTN", attn)
    sd = synth.esm_state_dict(**SM_ESM)
    eng = EsmEngine(sd, EsmConfig(**SM_ESM))
    toks = synth.protein_tokens([1, 1024, 2, 1025], seed=3)            # 1 residue ... 1025 residues (split into 1024 + 1)
    z_ref = PR.esm_plm_forward(sd, ER.EsmGeom(**SM_ESM), toks)
    z = eng.forward(toks).cpu()
    assert z.shape == z_ref.shape == (4, 128)
    one = synth.protein_tokens([1], seed=4)
    z1, z1_ref = eng.forward(one).cpu(), PR.esm_plm_forward(sd, ER.EsmGeom(**SM_ESM), one)
    if attn == "fast":      # single-pass attention: held to the fp32 evaluation (conftest.assert_no_further_from_truth)
        sd32 = {k: v.float() for k, v in sd.items()}
        assert_no_further_from_truth(z, z_ref, PR.esm_plm_forward(sd32, ER.EsmGeom(**SM_ESM), toks), "shortest + longest")
        assert_no_further_from_truth(z1, z1_ref, PR.esm_plm_forward(sd32, ER.EsmGeom(**SM_ESM), one), "one residue")
        return
    assert rel_err(z, z_ref) < 5e-3
    # a batch of one one-residue protein
    assert rel_err(z1, z1_ref) < 5e-3


def test_llama_single_token_prompt_and_single_token_generation():
    from oracle import llama_ref as LR
    from procyon_amd import synth
    from procyon_amd.engine import LlamaConfig, LlamaEngine
    sd = synth.llama_state_dict(**SM_LLAMA)
    eng = LlamaEngine({k: v.clone() for k, v in sd.items()}, LlamaConfig(**SM_LLAMA, max_pos=64))
    torch.manual_seed(0)
    emb = (torch.randn(2, 1, 256) * 0.02).to(BF)                       # T = 1
    for n in (1, 2, 5):
        tok_ref, lg_ref, _ = LR.greedy_generate(sd, LR.LlamaGeom(**SM_LLAMA), emb, torch.ones(2, 1), n)
        tok, lp, lg, _ = eng.generate_greedy(emb.cuda(), torch.ones(2, 1), n, keep_logits=True)
        assert tok.shape == (2, n) and lg.shape == (2, n, 320)
        assert rel_err(lg[:, 0].cpu(), lg_ref[:, 0]) < parity_bar(3e-3)
        assert torch.equal(tok[:, 0].cpu(), tok_ref[:, 0])


def test_errors_are_codes_with_text_not_crashes():
    from procyon_amd import synth
    from procyon_amd._lib import PcyError
    from procyon_amd.engine import Context, LlamaConfig, LlamaEngine
    ctx = Context.get()
    sd = synth.llama_state_dict(**SM_LLAMA)
    eng = LlamaEngine(sd, LlamaConfig(**SM_LLAMA, max_pos=32))
    emb = (torch.randn(1, 8, 256) * 0.02).to(BF).cuda()
    with pytest.raises(PcyError, match="exceed cache"):
        eng.prefill(emb, None, eng.new_cache(1, 4), "last")            # prompt longer than the cache
    with pytest.raises(PcyError, match="rope table"):
        eng.prefill((torch.randn(1, 40, 256) * 0.02).to(BF).cuda(), None, eng.new_cache(1, 64), "last")   # longer than max_pos
    with pytest.raises(PcyError, match="multiple of 128"):
        ctx.gemm_fp8(torch.zeros(4, 64, dtype=torch.uint8, device="cuda"), torch.ones(4, device="cuda"),
                     torch.zeros(8, 64, dtype=torch.uint8, device="cuda"), torch.ones(8, device="cuda"))
    # the context is still usable afterwards
    logits, _ = eng.prefill(emb, None, eng.new_cache(1, 16), "last")
    assert torch.isfinite(logits.float()).all()
    # zero rows: nothing launched, empty result
    out = ctx.gemm(torch.zeros(0, 64, dtype=BF, device="cuda"), torch.zeros(16, 64, dtype=BF, device="cuda"))
    assert out.shape == (0, 16)


def test_beam_of_one_equals_greedy():
    """beam_size = group_size = 1: the diverse beam search degenerates to arg-max of log-softmax = greedy; both device loops
    (graph-replayed decode + pcy_beam_step vs pcy_llama_greedy) must produce the same tokens."""
    from procyon_amd import synth
    from procyon_amd import synthetic_model as SM
    model = SM.build("small", device="cuda", max_new_tokens=16)
    prot = synth.protein_tokens([50], seed=2)
    inp = lambda: SM.caption_inputs(model, prot, n_prompt_words=20, n_slots=1, seed=3)
    tg, *_ = model.generate(inp(), max_len=12, method="greedy", truncate_on_eos=False)
    tb, sb, lb, _ = model.generate(inp(), max_len=12, method="beam", beam_size=1, beam_group_size=1, truncate_on_eos=False)
    assert torch.equal(tg.view(-1), tb.view(-1)), (tg.view(-1).tolist(), tb.view(-1).tolist())
    assert lb.shape[-1] == model.text_encoder.model.vocab_size
