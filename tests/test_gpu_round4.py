"""GPU tests added in round 4:
  * a row's logits from `pcy_llama_prefill` do not depend on how many rows were requested (64 vs 65: the lm_head GEMM is reserved for
    `pcy_llama_prefill_all`);
  * the single-pass ESM attention is launch-to-launch bit-stable (its softmax reads MFMA accumulators through inline asm; the
    chain now starts with a compiler-visible reader so the hazard recognizer pads it)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_prefill_logits_of_a_row_do_not_depend_on_the_number_of_rows_requested():
    """QA / pair scoring reads one row per prompt: 64 prompts or 65, the same prompt must get the same bits
    (/root/reference/procyon/data/inference_utils.py:582-604 reads `outputs.logits[i, answer_pos]`)."""
    from procyon_amd import synth
    from procyon_amd.engine import LlamaConfig, LlamaEngine
    kw = dict(vocab=331, d=256, n_layers=2, n_heads=4, n_kv_heads=2, ffn=512)
    eng = LlamaEngine(synth.llama_state_dict(**kw), LlamaConfig(**kw, max_pos=128), torch.device("cuda"))
    g = torch.Generator().manual_seed(5)
    B, T = 5, 24
    emb = (torch.randn(B, T, 256, generator=g) * 0.5).to(BF).cuda()
    rows65 = torch.arange(65, dtype=torch.int32)
    l65, _ = eng.prefill(emb, None, eng.new_cache(B, T), rows65)
    l64, _ = eng.prefill(emb, None, eng.new_cache(B, T), rows65[:64])
    l1, _ = eng.prefill(emb, None, eng.new_cache(B, T), rows65[7:8])
    assert torch.equal(l65[:64], l64) and torch.equal(l65[7:8], l1)
    # ... and the GEMM form of pcy_llama_prefill_all stays within rounding of it
    lall, _ = eng.prefill_all(emb, None, eng.new_cache(B, T), "all")
    assert rel_err(lall[:65].float().cpu(), l65.float().cpu()) < 4e-3


def test_single_pass_attention_is_launch_to_launch_bit_stable():
    """`attn_fast64_kernel` through `pcy_esm_encode`: 24 launches on the same input give the same bits (ragged lengths, incl. a
    sequence whose last wave holds a single query row)."""
    from procyon_amd import synth
    from procyon_amd.engine import EsmConfig, EsmEngine
    kw = dict(d=256, n_layers=2, n_heads=4, ffn=512)
    eng = EsmEngine(synth.esm_state_dict(**kw), EsmConfig(**kw))
    toks = synth.protein_tokens([255, 1, 64, 300, 127], seed=3)
    ref = eng.hidden_states(toks).clone()
    for _ in range(24):
        assert torch.equal(eng.hidden_states(toks), ref)
