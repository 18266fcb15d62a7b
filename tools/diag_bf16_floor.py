"""Why "bit-exact argmax ids" between two bf16 implementations cannot be asserted on a Gaussian-logit model, whatever the weights' scale:
the oracle against ITSELF with another fp32 accumulation order inside its Linears (tests/conftest.py::alt_accumulation -- the same bf16
rounding points, sums added in a different order) on a 16-layer d = 1024 Llama, residual branches x 1 / 0.25 / 0.1.  Measured in the
build container: err(twin, oracle) 1.4-1.6e-2 ~= err(oracle, fp32) 1.5-1.7e-2 at EVERY scale -- one flipped bf16 ulp re-draws every
later rounding decision, so two exact-class bf16 pipelines are as far from each other as each is from the fp32 truth, and damping the
branches does not change it (the noise is ~0.4 % per layer of materialised tensors, not amplification).  Argmax agreement is then a
function of top-2 margin / noise only: the full-depth tests assert agreement on the steps whose margin exceeds 4 x the noise."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import llama_ref as LR
from procyon_amd import synth
from conftest import alt_accumulation
torch.set_num_threads(8)
kw = dict(vocab=4096, d=1024, n_layers=16, n_heads=8, n_kv_heads=2, ffn=3584)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for scale in (1.0, 0.25, 0.1):
    sd = synth.llama_state_dict(**kw)
    if scale != 1.0: synth.damp_residual_branches(sd, scale)
    geom = LR.LlamaGeom(**kw, max_pos=256)
    ids = torch.randint(0, 4000, (1, 64), generator=torch.Generator().manual_seed(1))
    rb = LR.llama_forward(sd, geom, input_ids=ids, attn_mask=torch.ones(1, 64))
    with alt_accumulation():
        rt = LR.llama_forward(sd, geom, input_ids=ids, attn_mask=torch.ones(1, 64))
    rf = LR.llama_forward({k: v.float() for k, v in sd.items()}, geom, input_ids=ids, attn_mask=torch.ones(1, 64))
    lb, lt, lf = rb["logits"][0].float(), rt["logits"][0].float(), rf["logits"][0]
    print(f"scale {scale}: err(twin, oracle) {rel(lt, lb):.4f}  err(oracle, fp32) {rel(lb, lf):.4f}  agree twin/oracle {float((lt.argmax(-1)==lb.argmax(-1)).float().mean()):.3f}  oracle/fp32 {float((lb.argmax(-1)==lf.argmax(-1)).float().mean()):.3f}")
