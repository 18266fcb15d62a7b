"""Small-batch decode step (pcy_decode_nb.hip): bit-identity against its launch-per-stage twin, then ms per step for each batch size.

  CHECK=1 (default)  2-layer model, B = 2..8: fused launch vs PCY_DISABLE=decode_nb_step, eager and replayed
  TIME=1 (default)   32-layer model: ms per decode step per batch size, new path / PCY_DISABLE=decode_nb (the round-4 path)
  T=<prompt tokens> BATCHES=1,2,..  PCY_MC_TRACE=1 (in-kernel stamps of one middle layer)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("PCY_NB_MAX", "8")   # the fused step at 8 rows is opt-in since round 6
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine

BF = torch.bfloat16


def set_disable(*names):
    names = [n for n in names if n]
    if names:
        os.environ["PCY_DISABLE"] = ",".join(names)
    else:
        os.environ.pop("PCY_DISABLE", None)


def check():
    kw = dict(vocab=4096, d=4096, n_layers=2, n_heads=32, n_kv_heads=8, ffn=14336)
    eng = LlamaEngine(synth.llama_state_dict(**kw), LlamaConfig(**kw, max_pos=2048))
    ok_all = True
    for B, T, N in [(2, 300, 10), (3, 100, 8), (4, 300, 10), (5, 64, 8), (6, 40, 6), (7, 90, 6), (8, 300, 8), (4, 800, 6), (2, 1100, 5)]:
        torch.manual_seed(B * 1000 + T)
        emb = (torch.randn(B, T, 4096) * 0.02).to(BF).cuda()

        def run(step, use_graph):
            set_disable("" if step else "decode_nb_step")
            cache = eng.new_cache(B, T + N + 2)
            st = GenState(B, kw["vocab"], N + 2, "cuda")
            logits, _ = eng.prefill(emb, None, cache, "last")
            st.logits.copy_(logits); st.pos.fill_(T)
            eng.pick(cache, st, B, advance_pos=False)
            out = []
            for _ in range(N):
                eng.greedy_steps(cache, st, B, 1, use_graph=use_graph)
                out.append(st.logits.clone())
            Context.get().sync()
            return torch.stack(out).cpu(), st.tokens_out[:, :N + 1].cpu(), cache.k[:, :, :, T:T + N].cpu(), cache.v[:, :, :, T:T + N].cpu()

        try:
            ref = run(False, False)
            res = []
            for use_graph in (False, True, True):
                got = run(True, use_graph)
                res.append(all(torch.equal(x, y) for x, y in zip(got, ref)))
            d = (got[0].float() - ref[0].float()).abs().max().item()
            print(f"check B={B} T={T}: eager/graph/graph identical = {res}  max|dlogit| = {d:.3g}  finite = {bool(torch.isfinite(got[0].float()).all())}", flush=True)
            ok_all = ok_all and all(res)
        except Exception as e:  # noqa: BLE001
            print(f"check B={B} T={T}: FAILED {e}", flush=True)
            ok_all = False
    set_disable()
    print("CHECK", "PASS" if ok_all else "FAIL", flush=True)
    del eng
    torch.cuda.empty_cache()


def timing():
    kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
    eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
    ctx = Context.get()
    T, N = int(os.environ.get("T", 700)), 80
    for B in [int(x) for x in os.environ.get("BATCHES", "1,2,3,4,5,8").split(",")]:
        emb = (torch.randn(B, T, 4096, device="cuda") * 0.02).bfloat16()
        row = []
        for mode in ("new", "old"):
            set_disable("" if mode == "new" else "decode_nb")
            cache = eng.new_cache(B, T + N)
            st = GenState(B, kw["vocab"], N, "cuda")
            logits, _ = eng.prefill(emb, None, cache, "last")
            st.logits.copy_(logits); st.pos.fill_(T)
            eng.pick(cache, st, B, advance_pos=False)
            if os.environ.get("PCY_MC_TRACE"): eng.greedy_steps(cache, st, B, 1, use_graph=False)
            eng.greedy_steps(cache, st, B, 4)
            ctx.timer_start(); eng.greedy_steps(cache, st, B, 64); ms = ctx.timer_stop() / 64
            ctx.sync()
            row.append(ms)
            if mode == "new" and os.environ.get("PCY_MC_TRACE") and B > 1:
                import numpy as np
                from procyon_amd import _lib as L
                n = 2 * 128 * 256 * 16
                buf = np.zeros(n, dtype=np.uint64)
                L.load().pcy_debug_mc_trace(buf.ctypes.data, n)
                tr = buf[n // 2:].reshape(128, 256, 16)[:32].astype(np.int64)
                lay = tr[5]
                t0 = lay[:, 0].min()
                names = {0: "start", 13: "x in LDS", 14: "RMSNorm done", 15: "qkv rows computed", 1: "qkv stored / q,k,v staged", 2: "attention done / ao seen", 3: "ao in LDS", 4: "o stored", 8: "xo in LDS",
                         9: "gate/up done", 5: "down rows requested", 6: "window 0 tags ok (wave 0)", 7: "window tags ok (wave 6: a late block)", 10: "window 0 in LDS", 11: "down rows done", 12: "end (rows published)"}
                for grp, sl in (("attn wgs", slice(0, 64)), ("proj wgs", slice(64, 192)), ("qkv-only wgs", slice(192, 256))):
                    print(f"  -- {grp}")
                    for i, nm in names.items():
                        col = (lay[sl, i] - t0) / 100.0
                        col = col[lay[sl, i] > 0]
                        if len(col): print(f"  {nm:28s} min {col.min():7.2f}  med {np.median(col):7.2f}  max {col.max():7.2f} us")
        hbm = (15.0099e9 + B * (T + N / 2) * 131072) / (row[0] * 1e-3) / 1e12
        print(f"B={B:2d} t~{T + N // 2}: new {row[0]:.3f} ms/step ({B * 1e3 / row[0]:.0f} tok/s, {hbm:.2f} TB/s = {hbm / 8:.3f} of peak)   old {row[1]:.3f} ms/step", flush=True)
    set_disable()


if __name__ == "__main__":
    if os.environ.get("CHECK", "1") != "0":
        check()
    if os.environ.get("TIME", "1") != "0":
        timing()
