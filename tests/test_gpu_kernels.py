"""GPU parity: every primitive of libpcy.so (called through the C ABI) against the oracle's torch-CPU
restatement on the same seeded inputs.  Tolerance (stated once): bf16 outputs of two fp32-accumulating
implementations must agree except for rare 1-ulp flips -> `assert_bf16_close` (<= 2 bf16 ulp anywhere,
<= 2% of elements different) and norm-wise relative error <= 1e-3 (BASELINE.json north_star)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import assert_bf16_close, pcy_disable, rel_err

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ctx():
    from procyon_amd.engine import Context
    return Context.get()


def rnd(*shape, seed=0, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * std).to(BF)


def esm_gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def ref_linear(A, W, bias, resid, epi):
    y = F.linear(A, W, bias)
    if epi == 1:
        y = y + resid
    if epi == 2:
        y = F.gelu(y)
    if epi == 3:
        y = esm_gelu(y)
    return y


GEMM_SHAPES = [(200, 384, 256), (1026, 1280, 1280), (130, 1001, 128), (17, 256, 5120), (640, 3840, 1280)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm(ctx, M, N, K, epi):
    A, W, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, std=0.05), rnd(N, seed=3, std=0.1), rnd(M, N, seed=4)
    out = ctx.gemm(A.cuda(), W.cuda(), b.cuda(), r.cuda() if epi == 1 else None, epi).cpu()
    ref = ref_linear(A, W, b, r, epi)
    lin = F.linear(A, W, b)
    assert_bf16_close(out, ref, f"gemm {M}x{N}x{K} epi{epi}", inter=lin if epi else None, max_frac=0.05 if epi == 2 else 0.02)
    assert rel_err(out, ref) < 1e-3


@pytest.mark.parametrize("M,F_,K", [(200, 256, 256), (1026, 512, 128), (33, 1792, 512)])
def test_gemm_swiglu(ctx, M, F_, K):
    from procyon_amd.engine import interleave_gate_up
    A, g, u = rnd(M, K, seed=1), rnd(F_, K, seed=2, std=0.1), rnd(F_, K, seed=3, std=0.1)
    W = interleave_gate_up(g, u)
    out = ctx.gemm(A.cuda(), W.cuda(), None, None, 4).cpu()
    ref = F.silu(F.linear(A, g)) * F.linear(A, u)
    assert_bf16_close(out, ref, "gemm swiglu", ulps=2)


def test_gemm_transpose_detecting(ctx):
    """A = I with an asymmetric W (CDNA guide G9): catches a transposed or permuted output tile."""
    K = 128
    A = torch.eye(K).to(BF)
    W = ((torch.arange(256 * K).view(256, K) * 7) % 97 - 48).float().to(BF)
    out = ctx.gemm(A.cuda(), W.cuda()).cpu()
    assert torch.equal(out, W.t().contiguous())


@pytest.mark.parametrize("B", [1, 2, 3, 4, 6, 16, 20, 32, 45])   # B > 4 with K % 512 == 0 takes the skinny-MFMA kernel (32 rows per pass)
@pytest.mark.parametrize("N,K", [(4096, 4096), (512, 14336), (1000, 1280), (130, 136), (1001, 1024)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemv(ctx, B, N, K, epi):
    x, W, b, r = rnd(B, K, seed=1), rnd(N, K, seed=2, std=0.05), rnd(N, seed=3, std=0.1), rnd(B, N, seed=4)
    out = ctx.gemv(W.cuda(), x.cuda(), b.cuda(), r.cuda() if epi == 1 else None, epi).cpu()
    ref = ref_linear(x, W, b, r, epi)
    lin = F.linear(x, W, b)
    assert_bf16_close(out, ref, f"gemv B{B} {N}x{K} epi{epi}", inter=lin if epi else None, max_frac=0.05 if epi == 2 else 0.02)


@pytest.mark.parametrize("B", [1, 4])
@pytest.mark.parametrize("cast", [0, 1])
def test_gemv_fused_rms_and_swiglu(ctx, B, cast):
    from oracle.llama_ref import rms_norm
    from procyon_amd.engine import interleave_gate_up
    K, Fh = 4096, 1024
    x, w = rnd(B, K, seed=1), (1 + 0.02 * torch.randn(K)).to(BF)
    g, u = rnd(Fh, K, seed=2, std=0.03), rnd(Fh, K, seed=3, std=0.03)
    xn = rms_norm(x, w, 1e-5, "hf5" if cast == 0 else "hf431")
    out = ctx.gemv(interleave_gate_up(g, u).cuda(), x.cuda(), epi=4, rms_w=w.cuda(), rms_eps=1e-5, rms_cast=cast).cpu()
    ref = F.silu(F.linear(xn, g)) * F.linear(xn, u)
    assert_bf16_close(out, ref, "gemv rms+swiglu", ulps=2)
    Wq = rnd(768, K, seed=5, std=0.03)
    out = ctx.gemv(Wq.cuda(), x.cuda(), epi=0, rms_w=w.cuda(), rms_eps=1e-5, rms_cast=cast).cpu()
    assert_bf16_close(out, F.linear(xn, Wq), "gemv rms+store")


@pytest.mark.parametrize("d,ffn", [(4096, 14336), (2048, 4096)])
def test_decode_mlp_one_launch_vs_oracle(ctx, monkeypatch, d, ffn):
    """pcy_decode_mlp: one token through post_attention_layernorm -> LlamaMLP -> residual.  At the Llama-3-8B geometry this is
    the decode step's MLP chain kernel (gate/up + SwiGLU and down + residual in ONE launch, `act` handed between the stages as
    tagged words); other shapes run the two GEMV launches.  Checked against the oracle's arithmetic, and the one-launch path
    bit-for-bit against the two launches, over repeated calls on one slot (a stale tag would show)."""
    from oracle.llama_ref import rms_norm
    from procyon_amd.engine import interleave_gate_up
    w = (1 + rnd(d, seed=1, std=0.02).float()).to(BF)      # (seeded: the global RNG state depends on which tests ran before)
    g, u, dn = rnd(ffn, d, seed=2, std=0.03), rnd(ffn, d, seed=3, std=0.03), rnd(d, ffn, seed=4, std=0.02)
    wgu, wd, wc = interleave_gate_up(g, u).cuda(), dn.cuda(), w.cuda()
    for i in range(6):
        x = rnd(1, d, seed=10 + i)
        xn = rms_norm(x, w, 1e-5, "hf5")
        act = F.silu(F.linear(xn, g)) * F.linear(xn, u)
        ref = x + F.linear(act, dn)
        pcy_disable(monkeypatch)
        out = ctx.decode_mlp(x.cuda(), wc, wgu, wd).cpu()
        pcy_disable(monkeypatch, "mlp_chain")
        two = ctx.decode_mlp(x.cuda(), wc, wgu, wd).cpu()
        ctx.sync()
        assert torch.equal(out, two), i
        assert_bf16_close(out, ref, f"decode mlp d{d} ffn{ffn}", ulps=2, inter=x.float().abs() + F.linear(act, dn).float().abs())


@pytest.mark.parametrize("d", [4096, 1280, 136])
def test_norms(ctx, d):
    from oracle.llama_ref import rms_norm
    x, w, b = rnd(37, d, seed=1, std=2.0), (1 + 0.02 * torch.randn(d)).to(BF), (0.02 * torch.randn(d)).to(BF)
    for cast, nm in ((0, "hf5"), (1, "hf431")):
        assert_bf16_close(ctx.rmsnorm(x.cuda(), w.cuda(), 1e-5, cast).cpu(), rms_norm(x, w, 1e-5, nm), f"rms {nm}", max_frac=0.005)
    assert_bf16_close(ctx.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-5).cpu(), F.layer_norm(x, (d,), w, b, 1e-5), "ln", max_frac=0.005)


def test_embed_splice(ctx, golden):
    from oracle.procyon_ref import prepare_input_embeddings
    g = golden("g4_pad_splice")
    table = rnd(40, 64, seed=1)
    ids = g["ids"]
    soft = rnd(3, 64, seed=2)
    ref, _ = prepare_input_embeddings(table, ids, 31, soft)
    m = (ids == 31).view(-1)
    soft_map = torch.full((ids.numel(),), -1, dtype=torch.int32)
    soft_map[m] = torch.arange(int(m.sum()), dtype=torch.int32)
    out = ctx.embed_splice(table.cuda(), ids.to(torch.int32).view(-1).cuda(), soft.cuda(), soft_map.cuda()).cpu()
    assert torch.equal(out.view(3, 10, 64), ref)


def test_pool_matches_reference_pooler(ctx, golden):
    """ProteinPooler golden (the reference's own class): mean / corrected / max incl. multi-chunk proteins."""
    g = golden("g2_pool")
    z, keys, pad = g["z_bf16"], g["keys"], g["pad"]
    Bp, S, D = z.shape
    lens = (~pad).sum(1)
    cu = torch.zeros(Bp + 1, dtype=torch.int64)
    cu[1:] = lens.cumsum(0)
    packed = z[~pad]
    seg, rng = [0], []
    for i in range(int(keys.max()) + 1):
        for r in (keys == i).nonzero(as_tuple=True)[0].tolist():
            rng += [int(cu[r]), int(lens[r])]
        seg.append(len(rng) // 2)
    for mode, key in ((0, "mean_0"), (1, "mean_1"), (2, "max_0")):
        out = ctx.pool(packed.cuda().contiguous(), torch.tensor(seg, dtype=torch.int32).cuda(),
                       torch.tensor(rng, dtype=torch.int32).cuda(), 3, mode).cpu()
        assert_bf16_close(out, g[f"out_bf16_{key}"], f"pool {key}", max_frac=0.05)


def test_mlp(ctx):
    from oracle.procyon_ref import mlp_forward
    from procyon_amd import synth
    from procyon_amd.engine import MlpEngine
    for nl, M in ((3, 2), (3, 40), (1, 5)):
        layers = synth.mlp_layers(nl, 1280, 4096, 2560, seed_off=10)
        x = rnd(M, 1280, seed=7)
        eng = MlpEngine([(w.cuda(), None if b is None else b.cuda()) for w, b in layers])
        out, ref = eng(x.cuda()).cpu(), mlp_forward(x, layers)
        assert rel_err(out, ref) < 2e-3, (nl, M)   # 3 chained Linears with GELU tails: stack-level bar
        if nl == 1:
            assert_bf16_close(out, ref, f"mlp{nl} M{M}")


def _eager_attention(q, k, v, H, Hkv, dh, lens, causal, scale, keep=None):
    """HF eager attention (oracle/llama_ref.layer_forward, oracle/esm_ref.layer_forward) per packed sequence."""
    outs, t0 = [], 0
    for n in lens:
        qs = q[t0:t0 + n].view(n, H, dh).transpose(0, 1)
        ks = k[t0:t0 + n].view(n, Hkv, dh).transpose(0, 1).repeat_interleave(H // Hkv, 0)
        vs = v[t0:t0 + n].view(n, Hkv, dh).transpose(0, 1).repeat_interleave(H // Hkv, 0)
        s = torch.matmul(qs, ks.transpose(1, 2)) * scale
        allowed = torch.ones(n, n, dtype=torch.bool)
        if causal:
            allowed = torch.tril(allowed)
        if keep is not None:
            allowed = allowed & keep[t0:t0 + n].bool()[None, :]
        minv = torch.finfo(BF).min
        s = s + torch.where(allowed, torch.zeros((), dtype=BF), torch.full((), minv, dtype=BF))[None]
        p = F.softmax(s, dim=-1, dtype=torch.float32).to(BF)
        outs.append(torch.matmul(p, vs).transpose(0, 1).reshape(n, H * dh))
        t0 += n
    return torch.cat(outs)


@pytest.mark.parametrize("H,Hkv,dh,causal,lens", [(4, 4, 64, False, [70, 1, 129, 33]), (20, 20, 64, False, [300]),
                                                  (4, 4, 64, False, [1026, 191, 192, 193]), (2, 2, 128, True, [515]),
                                                  (8, 2, 128, True, [45, 200]), (4, 1, 64, True, [64, 65]), (8, 2, 128, True, [1200, 700]),
                                                  (2, 2, 32, True, [50])])
def test_attention_exact_rounding(ctx, H, Hkv, dh, causal, lens, monkeypatch):
    monkeypatch.setenv("PCY_ESM_ATTN", "exact")     # the two-pass kernels with the reference's rounding points (the single-pass ESM
    n = sum(lens)                                   # kernel has its own tests: tests/test_gpu_round3.py)
    q, k, v = rnd(n, H * dh, seed=1), rnd(n, Hkv * dh, seed=2), rnd(n, Hkv * dh, seed=3)
    scale = dh ** -0.5 if causal else 1.0
    if not causal:
        q = (q.float() * dh ** -0.5).to(BF)
    ref = _eager_attention(q, k, v, H, Hkv, dh, lens, causal, scale)
    out = ctx.attention(q.cuda(), k.cuda(), v.cuda(), lens, H, Hkv, dh, causal, scale).cpu()
    assert rel_err(out, ref) < 1e-3
    # a 1-ulp flip of one bf16 score S_j moves p_j by up to 0.8 % and O by 2^-7 * p_j * |v_j| (absolute); peaked rows
    # (or causal rows near the start) have p_j up to O(1): allow that on top of O's own ulp
    assert_bf16_close(out, ref, "attention", max_frac=0.03, inter=torch.full_like(ref, 0.5))


@pytest.mark.parametrize("T,pad_a,pad_b,dh", [(45, 5, 33, 64), (300, 0, 130, 128)])
def test_attention_left_pad_rows_uniform(ctx, T, pad_a, pad_b, dh):
    """fully masked (left-pad) query rows: the reference's finfo.min additive mask makes their softmax uniform over ALL
    keys (model_unified.py:769 quirk Q1 depends on the K/V these rows produce)."""
    H, Hkv = 4, 2
    q, k, v = rnd(2 * T, H * dh, seed=1), rnd(2 * T, Hkv * dh, seed=2), rnd(2 * T, Hkv * dh, seed=3)
    keep = torch.ones(2 * T, dtype=torch.uint8)
    keep[T:T + pad_b] = 0
    keep[:pad_a] = 0
    ref = _eager_attention(q, k, v, H, Hkv, dh, [T, T], True, dh ** -0.5, keep)
    out = ctx.attention(q.cuda(), k.cuda(), v.cuda(), [T, T], H, Hkv, dh, True, dh ** -0.5, keep.cuda()).cpu()
    assert rel_err(out, ref) < 1e-3
    assert_bf16_close(out, ref, "attention left-pad", max_frac=0.03, inter=torch.full_like(ref, 0.03 if T < 100 else 0.1))


@pytest.mark.parametrize("mode", [0, 1])
def test_rope(ctx, mode):
    from oracle import esm_ref as ER, llama_ref as LR
    from procyon_amd.engine import rope_tables
    nh, dh, n = 6, 64, 50
    x = rnd(n, nh * dh, seed=1)
    pos = torch.arange(n, dtype=torch.int32) % 37
    cos, sin = rope_tables(dh, 10000.0, 64, "cpu")
    xr = x.view(n, nh, dh).transpose(0, 1)[None]
    c, s_ = cos[pos.long()], sin[pos.long()]
    if mode == 0:
        ref, _ = LR.apply_rope(xr, xr, c[None], s_[None])
    else:
        ref, _ = ER.apply_rope((xr * dh ** -0.5), xr, c, s_, "fp32_once")
    ref = ref[0].transpose(0, 1).reshape(n, nh * dh)
    out = ctx.rope_(x.cuda().clone(), 0, nh, dh, pos.cuda(), cos.cuda(), sin.cuda(), mode, 0.0 if mode == 0 else dh ** -0.5).cpu()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("H,Hkv,dh,t,B", [(32, 8, 128, 600, 2), (8, 2, 64, 37, 2), (4, 4, 32, 5, 2), (8, 1, 128, 0, 2),
                                           (32, 8, 128, 333, 9), (32, 8, 128, 1030, 20), (16, 8, 128, 65, 16),
                                           (32, 8, 128, 0, 20), (32, 8, 128, 2500, 2)])
def test_attn_decode(ctx, H, Hkv, dh, t, B):
    """decode attention (rope + append + exact softmax) vs the oracle's eager formulas (llama_ref.layer_forward).
    B x Hkv >= 64 / >= 128 select the 64- / 128-column workgroup slices."""
    from oracle import llama_ref as LR
    from procyon_amd.engine import rope_tables
    Tmax = t + 3
    qkv = rnd(B, (H + 2 * Hkv) * dh, seed=1)
    kc, vc = rnd(B, Hkv, Tmax, dh, seed=2), rnd(B, Hkv, Tmax, dh, seed=3)
    cos, sin = rope_tables(dh, 10000.0, Tmax + 1, "cpu")
    q = qkv[:, :H * dh].view(B, 1, H, dh).transpose(1, 2)
    k = qkv[:, H * dh:(H + Hkv) * dh].view(B, 1, Hkv, dh).transpose(1, 2)
    v = qkv[:, (H + Hkv) * dh:].view(B, 1, Hkv, dh).transpose(1, 2)
    qr, kr = LR.apply_rope(q, k, cos[t][None, None].expand(B, 1, dh), sin[t][None, None].expand(B, 1, dh))
    K = torch.cat([kc[:, :, :t], kr], 2).repeat_interleave(H // Hkv, 1)
    V = torch.cat([vc[:, :, :t], v], 2).repeat_interleave(H // Hkv, 1)
    s = torch.matmul(qr, K.transpose(2, 3)) * dh ** -0.5
    p = F.softmax(s, dim=-1, dtype=torch.float32).to(BF)
    ref = torch.matmul(p, V).transpose(1, 2).reshape(B, H * dh)
    kcd, vcd = kc.cuda(), vc.cuda()
    out = ctx.attn_decode(qkv.cuda(), kcd, vcd, torch.tensor([t], dtype=torch.int32).cuda(), cos.cuda(), sin.cuda(), H, Hkv, dh).cpu()
    assert rel_err(out, ref) < 1e-3
    assert_bf16_close(out, ref, "attn decode", max_frac=0.03, inter=torch.full_like(ref, 0.05))
    assert torch.equal(kcd.cpu()[:, :, t], kr[:, :, 0]) and torch.equal(vcd.cpu()[:, :, t], v[:, :, 0])
    assert torch.equal(kcd.cpu()[:, :, :t], kc[:, :, :t])


@pytest.mark.parametrize("Q,N,D", [(3, 1000, 1280), (40, 513, 2560)])
def test_retrieval_scores(ctx, Q, N, D):
    from oracle.procyon_ref import retrieval_scores
    q, t = rnd(Q, D, seed=1), rnd(N, D, seed=2)
    ref = retrieval_scores(q, t)
    out = ctx.retrieval_scores(q.cuda(), t.cuda()).cpu()
    assert rel_err(out, ref) < 1e-3
    assert_bf16_close(out, ref.to(BF), "retrieval sims", inter=torch.full_like(ref, 0.02, dtype=torch.float32))
    # ranking of the top hits is what the callers consume (data/inference_utils.py:962-969)
    assert torch.equal(out.float().argmax(-1), ref.float().argmax(-1))


@pytest.mark.parametrize("B", [5, 20, 32, 40])
def test_gemv_mfma_swiglu_and_decode_layer(ctx, B):
    """batched decode path: SwiGLU on the skinny-MFMA kernel, then one full-width Llama-3-8B layer decoding B rows at once
    against the oracle (beam-20-shaped batch)."""
    from procyon_amd.engine import interleave_gate_up
    K, Fh = 4096, 1024 + 16
    x = rnd(B, K, seed=1)
    g, u = rnd(Fh, K, seed=2, std=0.03), rnd(Fh, K, seed=3, std=0.03)
    out = ctx.gemv(interleave_gate_up(g, u).cuda(), x.cuda(), epi=4).cpu()
    ref = F.silu(F.linear(x, g)) * F.linear(x, u)
    assert_bf16_close(out, ref, "mfma gemv swiglu", ulps=2)


@pytest.mark.parametrize("B", [5, 16, 20, 32, 45])
def test_gemv_mfma_lds_dma_variant_bit_identical(ctx, monkeypatch, B):
    """gemv_mfma4_kernel (default) / gemv_mfma3_kernel (PCY_DISABLE=gemv_mfma4): the batched decode GEMV with the weights through per-wave
    LDS-DMA rings (4 rows x 256 contiguous bytes per copy instruction) instead of register loads (16 rows x 64 bytes); the default
    stages x like the weights, in a ring of the same depth.  Same MFMA operands in the same order: all bit-identical to
    gemv_mfma2_kernel (PCY_DISABLE=gemv_lds) for every epilogue, incl. ragged N and the SwiGLU row pairing."""
    from procyon_amd.engine import interleave_gate_up
    x4, x14 = rnd(B, 4096, seed=1), rnd(B, 14336, seed=2)
    cases = [(rnd(4096, 4096, seed=3, std=0.03), x4, 1), (rnd(6144, 4096, seed=4, std=0.03), x4, 0), (rnd(1000, 4096, seed=5, std=0.03), x4, 2),
             (rnd(512, 14336, seed=6, std=0.02), x14, 1), (interleave_gate_up(rnd(1040, 4096, seed=7, std=0.03), rnd(1040, 4096, seed=8, std=0.03)), x4, 4)]
    for W, x, epi in cases:
        N = W.shape[0] // 2 if epi == 4 else W.shape[0]
        r = rnd(B, N, seed=9).cuda() if epi == 1 else None
        b = rnd(N, seed=10, std=0.1).cuda() if epi in (1, 2) else None
        outs = []
        for off in ("", "gemv_mfma4", "gemv_lds"):
            pcy_disable(monkeypatch, off)
            outs.append(ctx.gemv(W.cuda(), x.cuda(), b, r, epi).cpu())
        assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2]), (tuple(W.shape), epi)


def test_decode_batch20_full_width_layer():
    from oracle import llama_ref as LR
    from procyon_amd import synth
    from procyon_amd.engine import GenState, LlamaConfig, LlamaEngine
    kw = dict(vocab=512, d=4096, n_layers=1, n_heads=32, n_kv_heads=8, ffn=14336)
    sd = synth.llama_state_dict(**kw)
    geom = LR.LlamaGeom(**kw)
    eng = LlamaEngine(sd, LlamaConfig(**kw, max_pos=256))
    B, T = 20, 9
    g = torch.Generator().manual_seed(2)
    emb = (torch.randn(B, T, 4096, generator=g) * 0.02).to(BF)
    r = LR.llama_forward(sd, geom, inputs_embeds=emb, attn_mask=torch.ones(B, T))
    cache = eng.new_cache(B, 16)
    eng.prefill(emb.cuda(), None, cache, "last")
    tok = r["logits"][:, -1].argmax(-1)
    st = GenState(B, 512, 2, "cuda")
    st.pos.fill_(T)
    st.next_tok.copy_(tok.to(torch.int32))
    eng.decode(cache, st, B)
    ro = LR.llama_forward(sd, geom, input_ids=tok[:, None], attn_mask=None, past_kv=r["past_kv"])
    assert rel_err(st.logits.cpu(), ro["logits"][:, -1]) < 1e-2
    agree = (st.logits.cpu().float().argmax(-1) == ro["logits"][:, -1].float().argmax(-1)).float().mean()
    assert agree >= 0.9


def dispatch_counts(ctx):
    """launches per GEMM kernel family so far (pcy_debug_dispatch_count)"""
    return [int(ctx.lib.pcy_debug_dispatch_count(k)) for k in range(6)]


@pytest.mark.parametrize("M", [64, 2048])
def test_gemm_esm_gelu_epilogue_every_bf16_value(ctx, M):
    """The ESM GELU epilogue looks the five-rounding chain up in a table (2^-17 <= |x| < 2^7) and uses closed forms outside
    it: every one of the 65536 bf16 bit patterns through both kernel families -- 64x64 tiles at M = 64 and the persistent
    256x256 kernel at M = 2048 (N = 2048 >= 1280 output features; the dispatch is asserted) -- must give exactly what the
    reference's op-by-op bf16 chain x * 0.5 * (1.0 + erf(x / sqrt(2))) gives on the CPU."""
    import math
    from procyon_amd import _lib as L
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    vals = torch.cat([bits, bits]).view(BF)                # every bf16 value, twice
    W = vals.view(2048, 64).contiguous()                   # output[m][n] = W[n][m % 64] (A = identity rows, zero bias)
    A = torch.zeros(M, 64, dtype=BF)
    A[torch.arange(M), torch.arange(M) % 64] = 1.0
    before = dispatch_counts(ctx)
    out = ctx.gemm(A.cuda(), W.cuda(), torch.zeros(2048, dtype=BF).cuda(), None, 3).cpu()[:64]   # [64, 2048] = W^T
    after = dispatch_counts(ctx)
    which = L.DISPATCH_GEMM_BIG_PERSIST if M == 2048 else L.DISPATCH_GEMM_64
    assert [a - b for a, b in zip(after, before)] == [1 if k == which else 0 for k in range(6)]
    x = W.t().contiguous()
    # 0 * inf inside the MFMA turns the infinite / NaN patterns into NaN before the epilogue sees them, and -0 arrives as +0
    # (the accumulator starts at +0): compare every other pattern (65278 of them) bit for bit
    finite = torch.isfinite(x.float()) & (x.view(torch.int16) != -32768)
    ref = x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    assert torch.equal(out[finite].view(torch.int16), ref[finite].view(torch.int16))


@pytest.mark.parametrize("epi", [0, 1, 3, 4])
@pytest.mark.parametrize("M,N,K", [(2048, 1280, 1280), (2100, 1536, 2560), (2304, 5120, 1280), (2100, 1536, 4096), (2048, 1280, 4160)])
def test_gemm_256x256_kernels_vs_oracle(ctx, M, N, K, epi):
    """The 256x256 kernels -- gemm_kernel_big<STORE | RESID | SWIGLU> and gemm_kernel_big_persist<GELU_ESM>: the ESM retrieval leg
    and every batched prefill GEMM -- directly against the oracle's Linear (+ residual / ESM GELU / SwiGLU), with the dispatch
    asserted (M >= 2048, N >= 1280; M = 2100 leaves a ragged last row tile; K >= 4096 takes the software-pipelined k-loop, 4160 = an odd
    number of k-steps)."""
    from procyon_amd import _lib as L
    from procyon_amd.engine import interleave_gate_up
    A, W, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, std=0.05), rnd(N, seed=3, std=0.1), rnd(M, N, seed=4)
    before = dispatch_counts(ctx)
    if epi == 4:
        g, u = W[: N // 2].contiguous(), rnd(N // 2, K, seed=5, std=0.05)
        out = ctx.gemm(A.cuda(), interleave_gate_up(g.cuda(), u.cuda()), None, None, 4).cpu()
        gl, ul = F.linear(A, g), F.linear(A, u)
        ref = F.silu(gl) * ul
        # the two projections are bf16 tensors of their own: a 1-ulp flip of the gate moves silu(g) * u by <= 1.1 ulp(g) |u|, one
        # of silu(g) or of u by an ulp of the product
        lin = 1.1 * gl.float().abs() * ul.float().abs() + 2 * ref.float().abs()
    else:
        out = ctx.gemm(A.cuda(), W.cuda(), b.cuda(), r.cuda() if epi == 1 else None, epi).cpu()
        ref = ref_linear(A, W, b, r, epi)
        lin = F.linear(A, W, b) if epi else None
    after = dispatch_counts(ctx)
    which = L.DISPATCH_GEMM_BIG_PERSIST if epi == 3 else L.DISPATCH_GEMM_BIG
    assert [x - y for x, y in zip(after, before)] == [1 if k == which else 0 for k in range(6)], (before, after)
    assert_bf16_close(out, ref, f"big gemm {M}x{N}x{K} epi{epi}", inter=lin)
    assert rel_err(out, ref) < 1e-3


def test_esm_layer_256x256_fused_rotary_vs_oracle(ctx):
    """gemm_kernel_big<STORE + fused rotary> (the ESM qkv projection: q pre-scaled, q and k rotated in the epilogue) is only
    reachable through pcy_esm_encode: one full-width ESM2-650M layer over 2150 packed tokens against the oracle, asserting that
    all four GEMMs of the layer went to the 256x256 kernels (qkv + rotary, o + residual, fc2 + residual: big; fc1 + GELU:
    persistent) and none to the smaller tilings."""
    from oracle import esm_ref as ER
    from procyon_amd import _lib as L
    from procyon_amd import synth
    from procyon_amd.engine import EsmConfig, EsmEngine
    kw = dict(d=1280, n_layers=1, n_heads=20, ffn=5120)
    sd = synth.esm_state_dict(**kw)
    eng = EsmEngine(sd, EsmConfig(**kw))
    toks = synth.protein_tokens([1000, 700, 444], seed=21)
    before = dispatch_counts(ctx)
    out = eng.hidden_states(toks).cpu()
    d = [x - y for x, y in zip(dispatch_counts(ctx), before)]
    assert d[L.DISPATCH_GEMM_BIG] == 3 and d[L.DISPATCH_GEMM_BIG_PERSIST] == 1 and d[L.DISPATCH_GEMM_128] == 0 and d[L.DISPATCH_GEMM_64] == 0, d
    from conftest import alt_accumulation
    ref = ER.esm_forward(sd, ER.EsmGeom(**kw), toks)
    with alt_accumulation():     # the oracle with another fp32 accumulation order: the reproducibility floor of this layer
        twin = ER.esm_forward(sd, ER.EsmGeom(**kw), toks)
    keep = toks != 1
    err, floor = rel_err(out[keep], ref[keep]), rel_err(twin[keep], ref[keep])
    print(f"one ESM2-650M layer through the 256x256 kernels vs oracle: {err:.2e} (CPU-vs-CPU floor of this layer {floor:.2e})")
    assert err < max(2.5e-3, 2 * floor)


@pytest.mark.parametrize("B,beam,g,V,steps,eos", [(1, 4, 2, 300, 7, 299), (2, 6, 2, 481, 6, 5), (1, 5, 5, 300, 6, 299), (3, 8, 4, 260, 5, 7),
                                                    (1, 10, 2, 500, 5, 499), (2, 20, 4, 333, 4, 0), (1, 10, 2, 4099, 4, 17)])
def test_beam_step_kernel_matches_reference_bookkeeping(ctx, monkeypatch, B, beam, g, V, steps, eos):
    """pcy_beam_step (the whole per-step bookkeeping of the reference's diverse beam search in one launch) against the oracle's
    restatement of `_generate_beam_search` (oracle.llama_ref.beam_search, pinned to the reference's own loop by golden g7),
    both driven by the SAME synthetic "model": logits(step, previous token) from random tables.

    log_softmax: the kernel rounds the fp32 value (x - max) - log(sum) to bf16 once -- what torch's DEVICE kernel does, i.e.
    what the reference computes on its GPU (model_unified.py:782).  torch's CPU kernel for bf16 instead rounds log(sum) to bf16
    first (measured: -2.2500 where the exact -2.2587 rounds to -2.2656), so for this bit-exact comparison of the BOOKKEEPING
    the oracle's F.log_softmax is replaced by the correctly rounded one; end-to-end runs against the unpatched CPU oracle
    differ by such bf16 ulps in the scores (tests/test_gpu_unified.py allows 0.3)."""
    import torch.nn.functional as Fn
    from oracle import llama_ref as LR
    from procyon_amd.engine import BeamState, LlamaEngine

    def lsm_rounded_once(x, dim=-1):
        xd = x.double()
        m = xd.max(dim, keepdim=True).values
        return ((xd - m) - torch.log(torch.exp(xd - m).sum(dim, keepdim=True))).float().to(x.dtype)
    monkeypatch.setattr(LR.F, "log_softmax", lsm_rounded_once)
    BB, M = B * beam, 13
    # Exact ties among the best candidates are natural here (bf16 log-probabilities + fp32 running scores) and torch.topk leaves
    # their order unspecified (the kernel takes the lowest flat index): draw tables until the reference run meets no tie among
    # its g + 1 best candidates of any selection, so that the comparison is about the bookkeeping and not about tie order
    orig_topk = torch.Tensor.topk
    for seed in range(40):
        torch.manual_seed(1000 * seed + B * 100 + beam)
        tab = (torch.randn(steps, M, V) * 3.0).to(BF)                 # [steps, M, V]
        state = {"i": 0, "tie": False}

        def topk_spy(self, k, *a, **kw):
            v, _ = orig_topk(self, min(k + 1, self.numel()))
            state["tie"] = state["tie"] or bool((v[:-1] == v[1:]).any())
            return orig_topk(self, k, *a, **kw)

        def enc(input_embeds=None, input_ids=None, attn_masks=None, past_key_values=None):
            i = state["i"]
            state["i"] += 1
            if input_ids is None:
                lg = tab[0, 0][None].expand(BB, V)                   # identical rows after the prompt, as in the real model
                past = [[torch.zeros(BB, 1, 1, 1), torch.zeros(BB, 1, 1, 1)]]
            else:
                lg = tab[i][(input_ids.view(-1) * 7 + 3) % M]
                past = past_key_values
            return lg[:, None, :].clone(), past

        monkeypatch.setattr(torch.Tensor, "topk", topk_spy)
        t_ref, s_ref, lg_ref = LR.beam_search(enc, torch.zeros(B, 3, 8), torch.ones(B, 3), vocab_size=V, eos_id=eos, max_len=steps,
                                              beam_size=beam, beam_group_size=g, diversity_penalty=0.8)
        monkeypatch.setattr(torch.Tensor, "topk", orig_topk)
        if not state["tie"]:
            break
    else:
        pytest.skip("no tie-free table found")
    n_ref = state["i"]                                               # steps the reference ran before its EOS stop (if any)
    # ---- the kernel, driven the same way
    bs = BeamState(B, beam, steps, eos, prompt_len=3, device="cuda")
    tabd = tab.cuda()
    beam_step = LlamaEngine.beam_step.__get__(type("E", (), {"ctx": ctx})())   # the wrapper needs only .ctx
    rec = torch.zeros(steps, BB, V, dtype=BF, device="cuda")
    for i in range(steps):
        lg = tabd[0, 0][None].expand(BB, V).contiguous() if i == 0 else tabd[i][(bs.next_tok.long() * 7 + 3) % M].contiguous()
        rec[i] = lg
        beam_step(lg, bs, g, 0.8)
    tok, n = bs.tokens()
    assert n == n_ref, (n, n_ref)                                    # the device-side EOS stop froze the state at the same step
    assert torch.equal(tok.cpu().view(B, beam, n), t_ref[:, :, :n])
    assert torch.equal(bs.cur.cpu().view(B, beam), s_ref)
    assert int(bs.pos) == 3 + (n - 1)
    # logits record: per-slot rows + the parent chain == the reference's record, which is re-indexed by every step's parents
    # INCLUDING the step that produced the row (model_unified.py:785-787 append, then :827-829 reorder)
    anc = bs.anc[:n].cpu().long()
    slot = torch.arange(BB)
    for s_ in range(n - 1, -1, -1):
        slot = anc[s_][slot]
        assert torch.equal(rec[s_].cpu()[slot], lg_ref.view(BB, -1, V)[:, s_]), s_


@pytest.mark.parametrize("M", [64, 2048])
def test_gemm_swiglu_epilogue_every_bf16_gate_value(ctx, M):
    """The SwiGLU epilogue of both GEMM kernels (128x128 at M = 64, 256x256 at M = 2048): every bf16 pattern as the gate value,
    up = 1, must equal torch's F.silu on the bf16 tensor bit for bit.  (A table-lookup version of bf16(silu(g)) like the ESM
    GELU one passed this test too but bought nothing on the Llama gate/up GEMM -- 987 vs 991 TFLOP/s end to end -- and was
    dropped.)"""
    from procyon_amd import _lib as L
    from procyon_amd.engine import interleave_gate_up
    bits = torch.arange(65536, dtype=torch.int32).to(torch.int16)
    gate = torch.cat([bits, bits]).view(BF).view(2048, 64).contiguous()        # [F = 2048, K = 64]: every pattern twice
    up = torch.ones(2048, 64, dtype=BF)
    A = torch.zeros(M, 64, dtype=BF)
    A[torch.arange(M), torch.arange(M) % 64] = 1.0
    before = dispatch_counts(ctx)
    out = ctx.gemm(A.cuda(), interleave_gate_up(gate.cuda(), up.cuda()), None, None, L.EPI_SWIGLU).cpu()[:64]   # [64, F]: silu(gate[f][m])
    which = L.DISPATCH_GEMM_BIG if M == 2048 else L.DISPATCH_GEMM_128     # SwiGLU has no 64x64 variant
    assert [a - b for a, b in zip(dispatch_counts(ctx), before)] == [1 if k == which else 0 for k in range(6)]
    x = gate.t().contiguous()
    ok = torch.isfinite(x.float()) & (x.view(torch.int16) != -32768)           # inf / NaN become NaN in the MFMA, -0 arrives as +0
    ref = F.silu(x)
    assert torch.equal(out[ok].view(torch.int16), ref[ok].view(torch.int16))


@pytest.mark.parametrize("V", [500, 4099, 128263])
@pytest.mark.parametrize("mode", ["nucleus0.9", "nucleus0.5", "temp0.7", "temp1.0"])
def test_sample_step_kernel_vs_oracle(V, mode):
    """pcy_sample_pick (softmax, nucleus mask by histogram over the probabilities' bf16 values, inverse-CDF draw) against the
    oracle: the pre-sampling probability vector of `_generate_sampling` (oracle.llama_ref.sampling_probs, which sorts like the
    reference) and the draw for injected uniform variates (oracle.llama_ref.sample_token).  The vectors must be equal except
    for tokens of exactly the probability value at which the nucleus threshold falls (ties: the reference's torch.sort leaves
    their order open); the drawn token must be the oracle's unless u * total lies within fp32 rounding of a CDF step."""
    from oracle import llama_ref as LR
    from procyon_amd.engine import GenState, LlamaConfig, LlamaEngine
    from procyon_amd import synth
    kw = dict(vocab=V, d=64, n_layers=1, n_heads=2, n_kv_heads=1, ffn=128)
    eng = LlamaEngine(synth.llama_state_dict(**kw), LlamaConfig(**kw, max_pos=64))
    B, steps = 3, 5
    g = torch.Generator().manual_seed(V)
    cache = eng.new_cache(B, 16)
    st = GenState(B, V, steps, "cuda")
    u = torch.rand(steps * B, generator=g)
    u[0], u[1] = 0.0, 0.999999
    ud = u.cuda()
    nuc = float(mode[7:]) if mode.startswith("nucleus") else None
    temp = float(mode[4:]) if mode.startswith("temp") else 1.0
    lp_ref = torch.zeros(B)
    for s_ in range(steps):
        logits = (torch.randn(B, V, generator=g) * (1.0 + s_)).to(BF)
        st.logits.copy_(logits)
        probs = torch.empty(B, V, dtype=BF, device="cuda")
        eng.sample_pick(cache, st, B, False, ud, temp, nuc, probs)
        p_ref = LR.sampling_probs(logits, temperature=temp, nucleus_prob=nuc)
        p = probs.cpu()
        # the two softmax normalisers are fp32 sums in different orders: a probability may differ by one bf16 ulp; where the
        # MASK differs the token must sit at the threshold (its probability within a few ulps of the smallest kept one:
        # equal probabilities form a tie run whose order torch.sort leaves open, and a 1-ulp flip moves a token across it)
        full = logits.softmax(-1) if temp == 1.0 else (logits / temp).softmax(-1)
        for b in range(B):
            kept, kept_ref = p[b] > 0, p_ref[b] > 0
            both = kept & kept_ref
            assert ((p[b][both].float() - p_ref[b][both].float()).abs() <= 2.0 ** -7 * p_ref[b][both].float()).all()
            mism = kept != kept_ref
            if mism.any():
                assert nuc is not None
                edge = float(p_ref[b][kept_ref].float().min())
                assert ((full[b][mism].float() - edge).abs() <= 2.0 ** -5 * edge).all(), "mask differs away from the nucleus threshold"
                assert abs(int(kept.sum()) - int(kept_ref.sum())) <= max(4, V // 500)
        tok = st.next_tok.cpu().long()
        uu = u[s_ * B:(s_ + 1) * B]
        t_ref = LR.sample_token(p.float(), uu)     # the draw on the ENGINE's vector (equal to the oracle's up to tie order)
        cdf = p.double().cumsum(-1)
        for b in range(B):
            if tok[b] != t_ref[b]:
                target = float(uu[b]) * float(cdf[b, -1])
                near = min(abs(float(cdf[b, tok[b]]) - target), abs(float(cdf[b, t_ref[b]]) - target))
                assert near < 1e-5 * float(cdf[b, -1]), (s_, b, int(tok[b]), int(t_ref[b]))
            assert p[b, tok[b]] > 0
        lp_ref += torch.log_softmax(logits, -1)[torch.arange(B), tok].float()
        assert torch.equal(st.tokens_out[:, s_].cpu().long(), tok) and int(st.step) == s_ + 1
    assert torch.allclose(st.logprob.cpu(), lp_ref, atol=0.15)    # CPU bf16 log_softmax rounds log(sum) separately (see the beam test)
    assert int(eng.ctx.lib.pcy_ctx_sync(eng.ctx.h)) == 0
