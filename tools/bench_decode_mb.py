"""Mid-batch decode step (pcy_decode_mb.hip, 9..32 rows): bit-identity against its launch-per-stage twin, then ms per step per batch size.

  CHECK=1 (default)  2-layer model: fused launch vs PCY_DISABLE=decode_mb_step, eager and replayed
  TIME=1 (default)   32-layer model: ms per decode step per batch size, fused / launch per stage
  T=<prompt tokens> BATCHES=9,10,..  PCY_MC_TRACE=1 (in-kernel stamps of one middle layer)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine

BF = torch.bfloat16
os.environ.setdefault("PCY_MB_MAX", "32")   # the fused step is opt-in


def set_disable(*names):
    names = [n for n in names if n] + [n for n in os.environ.get("EXTRA_DISABLE", "").split(",") if n]   # EXTRA_DISABLE: switches held through the run (A/B)
    if names:
        os.environ["PCY_DISABLE"] = ",".join(names)
    else:
        os.environ.pop("PCY_DISABLE", None)


def check():
    kw = dict(vocab=4096, d=4096, n_layers=int(os.environ.get("CHECK_LAYERS", 2)), n_heads=32, n_kv_heads=8, ffn=14336)
    eng = LlamaEngine(synth.llama_state_dict(**kw), LlamaConfig(**kw, max_pos=2048))
    ok_all = True
    cases = [(9, 300, 6), (10, 100, 8), (12, 64, 5), (15, 40, 5), (16, 90, 5), (17, 300, 5), (20, 200, 8), (24, 33, 4), (32, 300, 6), (10, 800, 4), (32, 1100, 3)]
    if os.environ.get("CASES"):
        cases = [tuple(int(v) for v in c.split(":")) for c in os.environ["CASES"].split(",")]
    for B, T, N in cases:
        torch.manual_seed(B * 1000 + T)
        emb = (torch.randn(B, T, 4096) * 0.02).to(BF).cuda()

        def run(step, use_graph):
            set_disable("" if step else "decode_mb_step")
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
            break
    set_disable()
    print("CHECK", "PASS" if ok_all else "FAIL", flush=True)
    del eng
    torch.cuda.empty_cache()
    return ok_all


def timing():
    kw = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
    eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
    ctx = Context.get()
    T, N = int(os.environ.get("T", 728)), 80
    for B in [int(x) for x in os.environ.get("BATCHES", "10,16,20,32").split(",")]:
        emb = (torch.randn(B, T, 4096, device="cuda") * 0.02).bfloat16()
        row = []
        abls = [int(v) for v in os.environ.get("ABLS", "").split(",") if v]
        for abl in abls:       # timing ablations of the fused step (PCY_MB_ABL; eager launches: the value is a kernel argument)
            os.environ["PCY_MB_ABL"] = str(abl)
            set_disable("")
            cache = eng.new_cache(B, T + N)
            st = GenState(B, kw["vocab"], N, "cuda")
            logits, _ = eng.prefill(emb, None, cache, "last")
            st.logits.copy_(logits); st.pos.fill_(T)
            eng.pick(cache, st, B, advance_pos=False)
            eng.greedy_steps(cache, st, B, 4, use_graph=False)
            ctx.timer_start(); eng.greedy_steps(cache, st, B, 32, use_graph=False); ms = ctx.timer_stop() / 32
            ctx.sync()
            print(f"B={B:2d} ABL={abl:3d}: {ms:.3f} ms/step (eager)", flush=True)
        os.environ.pop("PCY_MB_ABL", None)
        if abls:
            continue
        for mode in ("new", "old"):
            set_disable("" if mode == "new" else "decode_mb_step")
            cache = eng.new_cache(B, T + N)
            st = GenState(B, kw["vocab"], N, "cuda")
            logits, _ = eng.prefill(emb, None, cache, "last")
            st.logits.copy_(logits); st.pos.fill_(T)
            eng.pick(cache, st, B, advance_pos=False)
            if os.environ.get("PCY_MC_TRACE"): eng.greedy_steps(cache, st, B, 1, use_graph=False)
            ug = os.environ.get("EAGER", "0") == "0"
            eng.greedy_steps(cache, st, B, 4, use_graph=ug)
            ctx.timer_start(); eng.greedy_steps(cache, st, B, 64, use_graph=ug); ms = ctx.timer_stop() / 64
            ctx.sync()
            row.append(ms)
            if mode == "new" and os.environ.get("PCY_MC_TRACE"):
                import numpy as np
                from procyon_amd import _lib as L
                n = 2 * 128 * 256 * 16
                buf = np.zeros(n, dtype=np.uint64)
                L.load().pcy_debug_mc_trace(buf.ctypes.data, n)
                tr = buf[n // 2:].reshape(128, 256, 16)[:32].astype(np.int64)
                lay = tr[5]
                t0 = lay[:, 0].min()
                names = {0: "layer start", 13: "Q: xn seen", 14: "Q: last step done", 1: "Q published", 2: "A: q/k/v partials seen", 3: "A published", 4: "O: ao seen", 12: "O: x(0) in LDS", 15: "O: last step done", 5: "O published", 6: "F1 done",
                         7: "G: xn seen", 8: "G published", 9: "D: act seen", 10: "D published", 11: "F2 done"}
                for i, nm in names.items():
                    col = (lay[:, i] - t0) / 100.0
                    col = col[lay[:, i] > 0]
                    if len(col): print(f"  {nm:26s} min {col.min():7.2f}  med {np.median(col):7.2f}  max {col.max():7.2f} us   (n={len(col)})")
                print(f"  next layer start           min {(tr[6][:, 0].min() - t0) / 100.0:7.2f}")
        alg = 15.0099e9 + B * (T + N / 2) * 131072
        print(f"B={B:2d} t~{T + N // 2}: fused {row[0]:.3f} ms/step ({B * 1e3 / row[0]:.0f} tok/s, {alg / (row[0] * 1e-3) / 1e12:.2f} TB/s = {alg / (row[0] * 1e-3) / 8e12:.3f} of peak)"
              f"   launch-per-stage {row[1]:.3f} ms/step ({alg / (row[1] * 1e-3) / 8e12:.3f})", flush=True)
    set_disable()


if __name__ == "__main__":
    ok = True
    if os.environ.get("CHECK", "1") != "0":
        ok = check()
    if ok and os.environ.get("TIME", "1") != "0":
        timing()
