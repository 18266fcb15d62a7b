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
embeddings (ESM + projectors).  err(HIP, oracle_bf16), argmax agreement and a near-tie histogram are printed.
"""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
LLAMA = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
ESM = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
SLACK = 1.25


@pytest.fixture(scope="module")
def llama():
    """Llama-3-8B geometry on the CPU-seeded weights of the fixture (the other full-size tests seed on the device)."""
    from procyon_amd import synth
    from procyon_amd.engine import LlamaConfig, LlamaEngine
    sd = synth.llama_state_dict(**LLAMA, workers=16)
    return LlamaEngine(sd, LlamaConfig(**LLAMA, max_pos=4096), free_source=True)


@pytest.mark.parametrize("T", [64, 512])
def test_llama8b_full_depth_vs_fp32_truth(llama, golden, T):
    from procyon_amd.engine import GenState
    g = golden(f"f1_llama8b_T{T}")
    ids, toks, cols = g["ids"].long(), g["tokens"].long(), g["cols"].long()
    nstep = toks.numel()                          # prefill + 64 decode steps
    cache = llama.new_cache(1, T + nstep + 1)
    logits, hidden = llama.prefill(llama.embed_tokens(ids), None, cache, "last", want_hidden=True)
    got = [logits[0].cpu()]
    st = GenState(1, LLAMA["vocab"], nstep + 1, "cuda")
    for s in range(1, nstep):                     # teacher-forced on the bf16 oracle's greedy tokens
        st.pos.fill_(T + s - 1)
        st.next_tok.copy_(toks[s - 1:s].to(torch.int32))
        llama.decode(cache, st, 1)
        got.append(st.logits[0].cpu())
    got = torch.stack(got).float()                # [65, V]
    truth, ref = g["logits_fp32"], g["logits_bf16"].float()
    worst = 0.0
    agree_hip_truth = agree_ref_truth = agree_hip_ref = 0
    for s in range(nstep):
        e_hip = rel_err(got[s, cols], truth[s])
        e_ref = rel_err(ref[s], truth[s])
        e_hr = rel_err(got[s, cols], ref[s])
        worst = max(worst, e_hip / e_ref)
        am = int(got[s].argmax())
        am_t, am_r = int(g["top_ids_fp32"][s, 0]), int(g["top_ids_bf16"][s, 0])
        agree_hip_truth += am == am_t
        agree_ref_truth += am_r == am_t
        agree_hip_ref += am == am_r
        margin = float(g["top_vals_fp32"][s, 0] - g["top_vals_fp32"][s, 1])
        if s < 4 or s % 16 == 0 or am != am_t:
            print(f"T={T} step {s}: err(HIP,fp32) {e_hip:.3e}  err(oracle_bf16,fp32) {e_ref:.3e} (all {LLAMA['vocab']} columns: "
                  f"{float(g['err_bf16_full'][s]):.3e})  err(HIP,oracle_bf16) {e_hr:.3e}  argmax HIP {am} / fp32 {am_t} / oracle {am_r}  "
                  f"fp32 top-2 margin {margin:.3e}")
        assert e_hip <= SLACK * e_ref, (s, e_hip, e_ref)
    # argmax agreement as a RATE over the prefill + 64 teacher-forced decode steps: the HIP path must agree with the fp32 truth as
    # often as the reference's own bf16 arithmetic does, within two steps (round-2 review: the per-step near-tie assertion could
    # not fail -- every fp32 top-2 margin of an untrained model lies inside the bf16 logits noise)
    print(f"T={T}: worst err(HIP,fp32)/err(oracle_bf16,fp32) = {worst:.3f}; argmax agreement over {nstep} steps: "
          f"HIP vs fp32 {agree_hip_truth}/{nstep}, oracle_bf16 vs fp32 {agree_ref_truth}/{nstep}, HIP vs oracle_bf16 {agree_hip_ref}/{nstep}")
    assert nstep >= 65
    assert agree_hip_truth >= agree_ref_truth - 2, (agree_hip_truth, agree_ref_truth)
    # final-normed hidden row of the prefill: all 4096 entries are in the fixture
    h_hip = hidden[0, -1].cpu().float()
    e_hip, e_ref = rel_err(h_hip, g["hidden_fp32"][0]), rel_err(g["hidden_bf16"][0].float(), g["hidden_fp32"][0])
    print(f"T={T} prefill hidden row: err(HIP,fp32) {e_hip:.3e}  err(oracle_bf16,fp32) {e_ref:.3e}")
    assert e_hip <= SLACK * e_ref


def test_esm650m_full_depth_vs_fp32_truth(golden):
    from procyon_amd import synth
    from procyon_amd.engine import EsmConfig, EsmEngine, MlpEngine
    g = golden("f2_esm650m_1024")
    eng = EsmEngine(synth.esm_state_dict(**ESM, workers=8), EsmConfig(**ESM))
    mk = lambda layers: MlpEngine([(w.cuda(), b.cuda()) for w, b in layers])
    shared, token = mk(synth.mlp_layers(3, 1280, 1280, 2560, 20)), mk(synth.mlp_layers(3, 1280, 4096, 2560, 0))
    toks = g["tokens"].long()
    hid = eng.hidden_states(toks)[0]
    z = eng.forward(toks)
    out = {"hidden_rows": hid[g["rows"].long().cuda()].cpu(), "pooled": z[0].cpu(), "shared": shared(z)[0].cpu(),
           "soft_token": token(z)[0].cpu()}
    for k, v in out.items():
        e_hip, e_ref = rel_err(v.float(), g[k + "_fp32"]), rel_err(g[k + "_bf16"].float(), g[k + "_fp32"])
        print(f"esm650m {k}: err(HIP,fp32) {e_hip:.3e}  err(oracle_bf16,fp32) {e_ref:.3e}  err(HIP,oracle_bf16) "
              f"{rel_err(v.float(), g[k + '_bf16'].float()):.3e}")
        assert e_hip <= SLACK * e_ref, k
