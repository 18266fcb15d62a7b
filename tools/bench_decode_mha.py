"""One-row decode step of the ProCyon-Split geometry (Llama-2-7B: 32 kv heads, ffn 11008; pcy_decode_mha.hip): bit-identity of the one-launch
step against the launch-per-layer and launch-per-stage runs, then ms per token.

  CHECK=1 (default)  2-layer model: step / PCY_DISABLE=decode_step / PCY_DISABLE=decode_step,decode_layer, eager and replayed
  TIME=1 (default)   32-layer model: ms per token, fused step and launches;  T=<prompt tokens>  PCY_MC_TRACE=1 (in-kernel stamps, GRAPH=0)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from procyon_amd import synth
from procyon_amd.engine import Context, GenState, LlamaConfig, LlamaEngine

BF = torch.bfloat16
GEO = dict(d=4096, n_heads=32, n_kv_heads=32, ffn=11008)


def set_disable(names):
    if names:
        os.environ["PCY_DISABLE"] = names
    else:
        os.environ.pop("PCY_DISABLE", None)


def check():
    kw = dict(vocab=4096, n_layers=2, **GEO)
    eng = LlamaEngine(synth.llama_state_dict(**kw), LlamaConfig(**kw, max_pos=2048))
    ok_all = True
    for T, N in [(40, 8), (300, 8), (764, 8), (1100, 5)]:
        torch.manual_seed(T)
        emb = (torch.randn(1, T, 4096) * 0.02).to(BF).cuda()

        def run(disable, use_graph):
            set_disable(disable)
            cache = eng.new_cache(1, T + N + 2)
            st = GenState(1, kw["vocab"], N + 2, "cuda")
            logits, _ = eng.prefill(emb, None, cache, "last")
            st.logits.copy_(logits); st.pos.fill_(T)
            eng.pick(cache, st, 1, advance_pos=False)
            out = []
            for _ in range(N):
                eng.greedy_steps(cache, st, 1, 1, use_graph=use_graph)
                out.append(st.logits.clone())
            Context.get().sync()
            return torch.stack(out).cpu(), st.tokens_out[:, :N + 1].cpu(), cache.k[:, :, :, T:T + N].cpu(), cache.v[:, :, :, T:T + N].cpu()

        try:
            ref = run("decode_step,decode_layer", False)
            res = {}
            for name, dis, g in (("step eager", "", False), ("step graph", "", True), ("step graph 2", "", True), ("layer eager", "decode_step", False),
                                 ("layer graph", "decode_step", True)):
                got = run(dis, g)
                res[name] = all(torch.equal(x, y) for x, y in zip(got, ref))
            d = (got[0].float() - ref[0].float()).abs().max().item()
            print(f"check T={T}: identical to the launches = {res}  max|dlogit| = {d:.3g}  finite = {bool(torch.isfinite(got[0].float()).all())}", flush=True)
            ok_all = ok_all and all(res.values())
        except Exception as e:  # noqa: BLE001
            print(f"check T={T}: FAILED {e}", flush=True)
            ok_all = False
    set_disable("")
    print("CHECK", "PASS" if ok_all else "FAIL", flush=True)
    del eng
    torch.cuda.empty_cache()


def timing():
    kw = dict(vocab=32007, n_layers=32, **GEO)
    eng = LlamaEngine(synth.llama_state_dict(**kw, device="cuda"), LlamaConfig(**kw, max_pos=4096), free_source=True)
    ctx = Context.get()
    T, N = int(os.environ.get("T", 512)), 256
    graph = os.environ.get("GRAPH", "1") != "0"
    emb = (torch.randn(1, T, 4096, device="cuda") * 0.02).bfloat16()
    gb = (32 * (12288 * 4096 + 4096 * 4096 + 3 * 11008 * 4096) * 2 + 32007 * 4096 * 2) / 1e9
    for mode, dis in (("step", ""), ("layer", "decode_step"), ("launches", "decode_step,decode_layer")):
        if os.environ.get("MODES") and mode not in os.environ["MODES"].split(","): continue
        set_disable(dis)
        cache = eng.new_cache(1, T + N)
        st = GenState(1, kw["vocab"], N, "cuda")
        logits, _ = eng.prefill(emb, None, cache, "last")
        st.logits.copy_(logits); st.pos.fill_(T)
        eng.pick(cache, st, 1, advance_pos=False)
        eng.greedy_steps(cache, st, 1, 8, use_graph=graph)
        ctx.sync()
        best = 1e9
        for rep in range(3):
            ctx.timer_start(); eng.greedy_steps(cache, st, 1, 60, use_graph=graph); ms = ctx.timer_stop() / 60
            best = min(best, ms)
        ctx.sync()
        kvb = 32 * (T + 100) * 2 * 4096 * 2 / 1e9
        print(f"{mode:9s} {best:.3f} ms/token  {1e3 / best:.1f} tok/s  {(gb + kvb) / best:.2f} TB/s = {(gb + kvb) / best / 8:.3f} of peak", flush=True)
        if mode == "step" and os.environ.get("PCY_MC_TRACE"):
            import numpy as np
            from procyon_amd import _lib as L
            n = 2 * 128 * 256 * 16
            buf = np.zeros(n, dtype=np.uint64)
            L.load().pcy_debug_mc_trace(buf.ctypes.data, n)
            tr = buf[n // 2:].reshape(128, 256, 16)[:32].astype(np.int64)
            for l in (1, 16):
                lay = tr[l]
                t0 = lay[:, 0].min()
                names = {0: "start", 14: "RMSNorm(x) in LDS", 1: "qkv stored / q,k,v staged", 2: "attention done / ao seen", 3: "ao in LDS / xo in LDS", 4: "o stored",
                         13: "RMSNorm(xo) in LDS", 12: "gate/up round one done", 9: "gate/up done", 10: "act part 1 in LDS", 11: "act part 2 in LDS", 5: "layer end"}
                for grp, sl in (("attention wgs", slice(0, 64)), ("o-row wgs", slice(64, 192)), ("qkv-only wgs", slice(192, 256))):
                    print(f"  -- layer {l} {grp}")
                    for i, nm in names.items():
                        col = (lay[sl, i] - t0) / 100.0
                        col = col[lay[sl, i] > 0]
                        if len(col): print(f"  {nm:28s} min {col.min():7.2f}  med {np.median(col):7.2f}  max {col.max():7.2f} us")
                print("  next layer start (same clock): min %.2f med %.2f" % ((tr[l + 1][:, 0].min() - t0) / 100.0, (np.median(tr[l + 1][:, 0]) - t0) / 100.0))
    set_disable("")


if __name__ == "__main__":
    if os.environ.get("CHECK", "1") != "0":
        check()
    if os.environ.get("TIME", "1") != "0":
        timing()
