"""GPU tests added in round 4:
  * a row's logits from `pcy_llama_prefill` do not depend on how many rows were requested (64 vs 65: the lm_head GEMM is reserved for
    `pcy_llama_prefill_all`);
  * the single-pass ESM attention is launch-to-launch bit-stable (its softmax reads MFMA accumulators through inline asm; the
    chain now starts with a compiler-visible reader so the hazard recognizer pads it)."""
import pytest
import torch

from conftest import pcy_disable, rel_err

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


# ---------------------------------------------------------------------------------------------- gemm_kernel_mid (pcy_gemm_mid.h)
@pytest.fixture(scope="module")
def ctx():
    from procyon_amd.engine import Context
    return Context.get()


def _rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(BF)


def _mid_count(ctx):
    from procyon_amd import _lib as L
    return int(ctx.lib.pcy_debug_dispatch_count(L.DISPATCH_GEMM_MID))


@pytest.mark.parametrize("epi", [0, 1, 3, 4])
@pytest.mark.parametrize("M,N,K", [(128, 1280, 1280), (1026, 1280, 5120), (1026, 3840, 1280), (2047, 1280, 1280), (512, 4096, 4096), (333, 992, 192)])
def test_mid_m_gemm_is_dispatched_and_bit_identical_to_the_small_tile_kernels(ctx, monkeypatch, M, N, K, epi):
    """128 <= M < 2048 below a full round of 256 x 256 tiles (one 1024-residue protein: /root/reference/procyon/model/esm.py:517-538 at
    batch 1; one 512-token prompt: /root/reference/procyon/model/pmc_llama.py:571-588): the launcher takes gemm_kernel_mid (asserted)
    and the result is BIT-identical to the 128 x 128 / 64 x 64 kernels it replaces (PCY_GEMM_MID=-1) -- same k order per element, so
    the encoder's packing invariance carries over.  (Those kernels are held to the oracle by test_gpu_kernels.py::test_gemm, which
    now reaches the mid kernel at four of its five shapes as well.)"""
    from procyon_amd.engine import interleave_gate_up
    A, W, b, r = _rnd(M, K, seed=1), _rnd(N, K, seed=2, std=0.05), _rnd(N, seed=3, std=0.1), _rnd(M, N, seed=4)
    if epi == 4:
        Wd, bias, res = interleave_gate_up(W[: N // 2].contiguous().cuda(), _rnd(N // 2, K, seed=5, std=0.05).cuda()), None, None
    else:
        Wd, bias, res = W.cuda(), b.cuda(), (r.cuda() if epi == 1 else None)
    monkeypatch.delenv("PCY_GEMM_MID", raising=False)
    n0 = _mid_count(ctx)
    out = ctx.gemm(A.cuda(), Wd, bias, res, epi)
    assert _mid_count(ctx) == n0 + 1
    monkeypatch.setenv("PCY_GEMM_MID", "-1")
    old = ctx.gemm(A.cuda(), Wd, bias, res, epi)
    assert _mid_count(ctx) == n0 + 1
    assert torch.equal(out, old)


@pytest.mark.parametrize("cfg", list(range(1, 15)))
def test_every_mid_m_configuration_is_bit_identical(ctx, monkeypatch, cfg):
    """all tile shapes (incl. the 128 x 96 ones: 28 staging pieces over 8 waves, the last piece repeated) / wave layouts / ring depths of gemm_kernel_mid on a ragged problem (M, N not multiples of any tile; 5 k-steps:
    shorter than the deepest ring), residual epilogue, against the 128 x 128 kernel"""
    M, N, K = 777, 1096, 320
    A, W, b, r = _rnd(M, K, seed=1).cuda(), _rnd(N, K, seed=2, std=0.05).cuda(), _rnd(N, seed=3, std=0.1).cuda(), _rnd(M, N, seed=4).cuda()
    monkeypatch.setenv("PCY_GEMM_MID", "-1")
    ref = ctx.gemm(A, W, b, r, 1)
    monkeypatch.setenv("PCY_GEMM_MID", str(cfg))
    n0 = _mid_count(ctx)
    out = ctx.gemm(A, W, b, r, 1)
    assert _mid_count(ctx) == n0 + 1
    assert torch.equal(out, ref)
    for _ in range(8):          # the counted-vmcnt ring: launch-to-launch stable
        assert torch.equal(ctx.gemm(A, W, b, r, 1), ref)


# ---------------------------------------------------------------------------------------------- replayed launch chain of the short-input encoder
def test_esm_encode_replays_a_captured_launch_chain_with_the_same_bits(ctx, monkeypatch):
    """`pcy_esm_encode` of a short input (one protein; /root/reference/procyon/model/esm.py:517-538 at batch 1) runs launch by launch the
    first time a shape is seen, captures its launch chain the second time and replays it from then on (asserted through the dispatch
    counter).  Same kernels, same arguments: the bits must equal the launch-by-launch run (PCY_DISABLE=esm_graph), also when two shapes
    alternate and when the tokens change under an unchanged shape."""
    from procyon_amd import _lib as L
    from procyon_amd import synth
    from procyon_amd.engine import EsmConfig, EsmEngine
    kw = dict(d=256, n_layers=3, n_heads=4, ffn=512)
    eng = EsmEngine(synth.esm_state_dict(**kw), EsmConfig(**kw))
    cnt = lambda: int(ctx.lib.pcy_debug_dispatch_count(L.DISPATCH_ESM_GRAPH))
    a1, a2, b1 = synth.protein_tokens([300], seed=1), synth.protein_tokens([300], seed=2), synth.protein_tokens([77, 130], seed=3)
    pcy_disable(monkeypatch, "esm_graph")
    ref = {k: eng.forward(t).clone() for k, t in (("a1", a1), ("a2", a2), ("b1", b1))}
    pcy_disable(monkeypatch)
    n0 = cnt()
    outs = [(k, eng.forward(t).clone()) for k, t in (("a1", a1), ("a1", a1), ("b1", b1), ("a2", a2), ("b1", b1), ("a1", a1), ("b1", b1))]
    assert cnt() - n0 >= 4, cnt() - n0            # a1 x2 (2nd, 4th, 6th of that shape: >= 3 replays incl. the capturing call) + b1 (2nd, 3rd)
    for k, o in outs:
        assert torch.equal(o, ref[k]), k


# ---------------------------------------------------------------------------------------------- the N > 1 path of bench.py on one GPU
def test_bench_under_a_process_group_splits_rows_and_gathers():
    """`bench.py` with PCY_BENCH_FORCE_DIST=1 (world 1, backend nccl = RCCL): the WHOLE distributed path of the driver's N > 1 runs --
    process-group set-up, the sharded retrieval leg with its RCCL all-gather, and the `configs` block with the 32 generation rows / 256
    pairs split across the ranks (contiguous chunks in rank order, /root/reference/procyon/data/samplers.py:154-196) and ONE gather of the
    token ids / probabilities through `pcy_allgather` -- so that the first real 8-GPU run cannot fail on plumbing."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, PCY_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--retrieval-proteins", "50"],
                         capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and "RCCL" in d["retrieval"]["collective"]
    c = d["configs"]
    for k in ("config3_4a_batch32_mixed_residues_T512", "config3_4b_batch32_ragged_prompts"):
        assert c[k]["rows"] == 32 and c[k]["ranks"] == 1 and c[k]["rows_per_rank"] == 32 and "all-gather" in c[k]["collective"] and c[k]["tokens_per_s"] > 0
    assert c["config4_pair_scoring_256"]["pairs"] == 256 and c["config4_pair_scoring_256"]["ranks"] == 1
    assert "batched_decode_roofline" not in d and "config4_fp8_accuracy_damped_model" not in c      # single-GPU-only blocks
