"""Synthetic workloads of BASELINE.json configs[3] and configs[4] through the model-level API (`UnifiedProCyon.generate` /
`.forward`), shared by bench.py's `configs` block, tools/run_configs.py and the full-size GPU tests.

config 4  batch 32, residue lengths uniform-int in [256, 2048] (seed 7; proteins > 1024 residues are split into chunks and
          pooled jointly), 512 greedy tokens; (4a) every prompt 512 tokens, (4b) ragged prompts, T uniform-int in [128, 512],
          left-padded (reference quirks Q1 / Q2 apply: SURVEY App. B)
config 5  256 (protein, peptide) pairs as six-slot QA prompts (one positive + one negative in-context pair + the query pair,
          T ~ 450): P(yes), P(no) at the last [ANSWER]; bf16 weights and the fp8 (e4m3, MX MFMA) weight path
"""
from __future__ import annotations

import time

import torch

from . import synth


def _sync():
    torch.cuda.synchronize()


def _words(n, salt):
    return [f"w{(salt + 31 * i) % 50000}" for i in range(n)]


def config4_inputs(ragged=False, rows=32, seed=7, subset=None):
    """The `rows` synthetic rows of config 4 (always drawn for the WHOLE batch, so a rank that runs `subset` of them sees exactly the
    rows the single-GPU run gives those indices); make() builds the input dict of the subset (default: all rows)."""
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(256, 2049, (rows,), generator=g).tolist()
    prot = synth.protein_tokens(lens, seed=seed)
    plen = torch.randint(128, 513, (rows,), generator=g).tolist() if ragged else [512] * rows
    instr = []
    for b in range(rows):
        n = plen[b] - 2                      # words + <|protein|> + [ANSWER]
        instr.append(" ".join(_words(n // 2, 17 * b) + ["<|protein|>"] + _words(n - n // 2, 13 * b + 5)) + " [ANSWER]")
    sel = list(range(rows)) if subset is None else list(subset)

    def make():
        return {"data": {"seq": prot[torch.tensor(sel)].clone(), "seq_idx": torch.arange(len(sel)), "text": [], "drug": None},
                "input": {"seq": [[b] for b in range(len(sel))], "text": [[] for _ in sel], "drug": None},
                "target": {"seq": None, "text": None, "drug": None}, "instructions": [instr[b] for b in sel]}
    return make, lens, plen


def run_config4(model, new_tokens=512, ragged=False, rows=32, group=None):
    """One GPU: the whole batch.  Under a process group (bench.py --gpus N): the `rows` rows split across the ranks
    (`distributed.run_sharded_rows`: contiguous chunks in rank order, /root/reference/procyon/data/samplers.py:154-196), every rank
    generates its rows, one all-gather of the int32 token ids; the time is the slowest rank's (barrier on both sides)."""
    import torch.distributed as td
    from .distributed import run_sharded_rows, shard_indices
    dist = td.is_available() and td.is_initialized()
    world = td.get_world_size(group) if dist else 1
    rank = td.get_rank(group) if dist else 0
    _, lens, plen = config4_inputs(ragged, rows)
    dev = model.text_encoder.engine.device

    def gen(idx):
        make, _, _ = config4_inputs(ragged, rows, subset=idx)
        toks, *_ = model.generate(make(), max_len=new_tokens, method="greedy")
        assert toks.shape == (len(idx), 1, new_tokens)
        t = toks.to(torch.int32)
        return t.to(dev) if dist and td.get_backend(group) == "nccl" else t

    def barrier():
        if dist:
            td.barrier(group)
        _sync()
    if dist and td.get_backend(group) == "nccl":      # communicator bootstrap outside the timed region
        from .distributed import _comm_for
        _comm_for(group)
    # warm-up = the same call: besides first-launch effects it makes the host allocator hold the pinned logits record of this shape
    # ([new_tokens, rows, vocab] bf16 = 4.2 GB at 32 x 512, written by the decode steps themselves because the reference returns the
    # per-step logits on the CPU); its first allocation costs ~0.4 s, every later call of a serving loop reuses it
    gen(shard_indices(rows, rank, world))
    barrier()
    t0 = time.perf_counter()
    toks = run_sharded_rows(gen, rows, rank, world, group)
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device=dev if td.get_backend(group) == "nccl" else "cpu")
        td.all_reduce(t, op=td.ReduceOp.MAX, group=group)
        dt = float(t)
    assert toks.shape == (rows, 1, new_tokens)
    out = {"rows": rows, "residues_min_max": [min(lens), max(lens)], "protein_chunks": int(sum((n + 1023) // 1024 for n in lens)),
           "prompt_tokens": "ragged %d-%d, left-padded" % (min(plen), max(plen)) if ragged else 512, "new_tokens": new_tokens,
           "seconds": round(dt, 3), "tokens_per_s": round(rows * new_tokens / dt, 1)}
    if dist:
        out.update(rows_per_rank=len(shard_indices(rows, rank, world)), ranks=world, scaling="strong (32 rows split across the ranks)",
                   collective="one all-gather of the int32 token ids" + (" (RCCL through pcy_allgather)" if td.get_backend(group) == "nccl" else ""),
                   token_checksum=int(toks.to(torch.int64).sum()))
    return out


def batched_decode_roofline(model, rows=32, prompt=512, new_tokens=512):
    """HBM roofline of the BATCHED decode step (BASELINE configs[3] shape: `rows` prompts of `prompt` tokens, `new_tokens` greedy
    steps): HIP-event time of the hipGraph-replayed steps, algorithmic bytes per step = every weight once
    (15,009,906,688 B for Llama-3-8B) + the K/V rows read, rows * t * 131,072 B at the mean cache length."""
    from .engine import Context, GenState
    eng = model.text_encoder.engine
    cfg = eng.cfg
    ctx = Context.get(eng.device)
    g = torch.Generator().manual_seed(3)
    emb = (torch.randn(rows, prompt, cfg.d, generator=g) * 0.02).to(torch.bfloat16).to(eng.device)
    cache = eng.new_cache(rows, prompt + new_tokens)
    st = GenState(rows, cfg.vocab, new_tokens, eng.device)
    logits, _ = eng.prefill(emb, None, cache, "last")
    st.logits.copy_(logits); st.pos.fill_(prompt)
    eng.pick(cache, st, rows, advance_pos=False)
    eng.greedy_steps(cache, st, rows, 8)
    n = new_tokens - 16
    ctx.timer_start()
    eng.greedy_steps(cache, st, rows, n)
    ms = ctx.timer_stop() / n
    ctx.sync()
    w_bytes = 2 * (cfg.n_layers * (cfg.d * (cfg.n_heads + 2 * cfg.n_kv_heads) * cfg.head_dim + cfg.n_heads * cfg.head_dim * cfg.d
                                   + 3 * cfg.d * cfg.ffn + 2 * cfg.d) + cfg.d + cfg.vocab * cfg.d)
    t_mean = prompt + 8 + n / 2.0
    kv_bytes = rows * t_mean * (2 * cfg.n_layers * cfg.n_kv_heads * cfg.head_dim * 2)
    gbps = (w_bytes + kv_bytes) / 1e9 / (ms / 1e3)
    import os
    mb_max = int(os.environ.get("PCY_MB_MAX", "0"))
    off = os.environ.get("PCY_DISABLE", "").split(",")
    nb_max = min(int(os.environ.get("PCY_NB_MAX", "6")), 8)
    kernel = ("small-batch decode step (hipGraph: embed, decode_step_nb_kernel = all layers in one launch, lm_head, pick)" if 2 <= rows <= nb_max else
              "mid-batch decode step (hipGraph: embed, norm, decode_step_mb_kernel = all layers in one launch, lm_head, pick)"
              if 9 <= rows <= min(mb_max, 32) and "decode_mb_step" not in off else
              "batched decode step (hipGraph: per layer qkv / attention (+ qkv finish) / o / finish+norm / gate-up / down / finish+norm launches, lm_head, pick)")
    return {"bound": "hbm", "kernel": kernel,
            "rows": rows, "mean_cache_len": round(t_mean, 1), "ms_per_step": round(ms, 4), "tokens_per_s_decode_only": round(rows * 1e3 / ms, 1),
            "bytes_per_step": int(w_bytes + kv_bytes), "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4)}


def decode_batch_curve(model, rows_list=(2, 4, 5, 8, 10, 16, 20, 32), prompt=696, new_tokens=144):
    """ms per decode step and fraction of the HBM peak for each batch size at a mean cache length of ~768 keys (beam search runs at
    batch = beam_size, /root/reference/procyon/model/model_unified.py:751-832; the N-GPU points of configs[3] at 32 / N rows per GPU)."""
    out = []
    for rows in rows_list:
        r = batched_decode_roofline(model, rows=rows, prompt=prompt, new_tokens=new_tokens)
        out.append({k: r[k] for k in ("rows", "mean_cache_len", "ms_per_step", "tokens_per_s_decode_only", "achieved", "frac", "kernel")})
    return out


def config3_projected_scaling(model, new_tokens=512, rows=32, ranks=(2, 4, 8), single_gpu=None):
    """The 1 -> 8 GPU curve of configs[3] from ONE GPU: generation shards as replicas (the rows split across the
    ranks, no traffic inside the loop, one final gather of <= 64 KB of token ids), so the N-GPU time is the time of the slowest rank's
    rows/N rows.  Every rank's chunk (`distributed.shard_indices`, the chunks the N-rank job would hand out) is generated here in turn
    and the slowest taken; tokens/s(N) = rows x new_tokens / that time.  `single_gpu`: the measured N = 1 result (run_config4)."""
    from .distributed import shard_indices

    def gen(idx):
        make, _, _ = config4_inputs(False, rows, subset=idx)
        toks, *_ = model.generate(make(), max_len=new_tokens, method="greedy")
        return toks

    pts = []
    if single_gpu is not None:
        pts.append({"gpus": 1, "rows_per_gpu": rows, "seconds": single_gpu["seconds"], "tokens_per_s": single_gpu["tokens_per_s"], "measured": "this GPU, all rows"})
    for n in ranks:
        worst = 0.0
        gen(shard_indices(rows, 0, n))       # warm-up of the shape (pinned logits record, graph capture)
        for r in range(n):
            _sync()
            t0 = time.perf_counter()
            gen(shard_indices(rows, r, n))
            _sync()
            worst = max(worst, time.perf_counter() - t0)
        pts.append({"gpus": n, "rows_per_gpu": rows // n, "seconds": round(worst, 3), "tokens_per_s": round(rows * new_tokens / worst, 1),
                    "measured": f"slowest of the {n} rank chunks, each run on this GPU"})
    base = pts[0]["tokens_per_s"]
    for p in pts:
        p["speedup_vs_first_point"] = round(p["tokens_per_s"] / base, 2)
    return {"what": "projected from single-GPU runs of every rank's chunk (no N > 1 hardware run): replicas with no traffic inside the loop; "
                    "the final gather of the token ids (<= 64 KB) is not included", "new_tokens": new_tokens, "points": pts}


def config5_inputs(pairs=256, chunk=64, seed=7):
    g = torch.Generator().manual_seed(seed)
    plen = [805] + [int(x) for x in torch.randint(8, 41, (pairs + 2,), generator=g)]   # receptor + peptides
    prot = synth.protein_tokens(plen, seed=11)
    tmpl = (" ".join(_words(140, 1)) + " <|protein|> binds <|protein|> ? [ANSWER] yes " + " ".join(_words(140, 2)) +
            " <|protein|> binds <|protein|> ? [ANSWER] no " + " ".join(_words(140, 3)) + " <|protein|> binds <|protein|> ? [ANSWER]")

    def make(lo, cnt=None):
        cnt = chunk if cnt is None else cnt
        return {"data": {"seq": prot.clone(), "seq_idx": torch.arange(prot.shape[0]), "text": [], "drug": None},
                "input": {"seq": [[0, 1, 0, 2, 0, 3 + lo + i] for i in range(cnt)], "text": [[] for _ in range(cnt)], "drug": None},
                "target": {"seq": None, "text": None, "drug": None}, "instructions": [tmpl] * cnt}
    return make, len(tmpl.split())


def score_pairs(model, pairs=256, chunk=64, first=0, count=None):
    """-> (P(yes), P(no)) [count, 2] fp32 and a 1/61 sample of every answer-row logit vector, for pairs [first, first + count) of the
    `pairs`-pair workload (default: all of them), `chunk` prompts per forward call"""
    make, _ = config5_inputs(pairs, chunk)
    count = pairs if count is None else count
    ys, ls = [], []
    for lo in range(first, first + count, chunk):
        lg = model.forward(make(lo, min(chunk, first + count - lo)), retrieval=False)["outputs"].answer_logits[:, 0]
        # the reference's read-out (data/inference_utils.py:582-604): softmax over the vocabulary in the model dtype, then the yes / no columns
        from .engine import Context
        ys.append(Context.get().qa_probs(lg, model.yes_token, model.no_token, want_probs=False)[1])
        ls.append(lg.float()[:, ::61].clone())
    return torch.cat(ys), torch.cat(ls)


def damp_residual_branches(model, scale):
    """Multiply the o and down projections of every decoder layer by `scale` IN PLACE (and drop cached fp8 copies).  With
    N(0, 0.02^2) weights every residual branch is as large as the stream it is added to and 32 layers amplify a 2^-9 rounding
    into ~8 % of the logits (tests/test_gpu_fulldepth.py) -- any two arithmetics then disagree on a yes/no coin flip.  Trained
    decoders keep their branches a fraction of the stream; scale ~ 0.25 puts the synthetic model into that regime, where the
    agreement between the fp8 and the bf16 weight path says something about the fp8 path."""
    eng = model.text_encoder.engine
    for ws in eng._keep:
        ws[1].mul_(scale)       # wo
        ws[3].mul_(scale)       # wdown
    eng.set_fp8(False)
    eng._arr8 = None            # the e4m3 copies are re-made from the scaled matrices on the next quantize_fp8()
    eng._keep8 = []


def run_config5(model, pairs=256, chunk=64, fp8=True, group=None):
    """pairs/s with bf16 weights and on the fp8 weight path, algorithmic TFLOP/s of the Llama prefill part (2 * params * tokens
    + attention), and how far the fp8 path's answers are from the bf16 path's.  Under a process group the pairs are split across the
    ranks (contiguous chunks, rank order) and the [pairs, 2] probabilities gathered once ("split pairs")."""
    import torch.distributed as td
    from .distributed import all_gather_rows, shard_indices
    dist = td.is_available() and td.is_initialized()
    world = td.get_world_size(group) if dist else 1
    rank = td.get_rank(group) if dist else 0
    assert pairs % world == 0, "pair scoring splits an equal number of pairs to every rank"
    mine = shard_indices(pairs, rank, world)
    first, count = mine[0], len(mine)
    eng = model.text_encoder.engine
    cfg = eng.cfg
    _, T = config5_inputs(pairs, chunk)
    out = {}
    if dist and td.get_backend(group) == "nccl":      # the communicator's bootstrap (broadcast of the id + ncclCommInit) is not part of a timed pass
        from .distributed import _comm_for
        _comm_for(group)
    for mode in (("bf16", "fp8") if fp8 else ("bf16",)):
        if mode == "fp8":
            eng.quantize_fp8()
        score_pairs(model, pairs, chunk, first, min(chunk, count))
        _sync()
        best = None
        for _ in range(2):          # two passes, the faster one is reported (a shared box now and then loses a third of a pass)
            if dist:
                td.barrier(group)
            _sync()
            t0 = time.perf_counter()
            y, lg = score_pairs(model, pairs, chunk, first, count)
            if dist:
                y, lg = all_gather_rows(y, world * count, group), all_gather_rows(lg, world * count, group)
                td.barrier(group)
            _sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        if dist:
            t = torch.tensor([best], device=y.device if td.get_backend(group) == "nccl" else "cpu")
            td.all_reduce(t, op=td.ReduceOp.MAX, group=group)
            best = float(t)
        out[mode] = (y.cpu()[:pairs], best, lg.cpu()[:pairs])
    if fp8:
        eng.set_fp8(False)
    per_tok = 2 * cfg.n_layers * (cfg.d * (cfg.n_heads + 2 * cfg.n_kv_heads) * cfg.head_dim + cfg.n_heads * cfg.head_dim * cfg.d + 3 * cfg.d * cfg.ffn)
    flop = pairs * (per_tok * T + 4 * cfg.n_layers * cfg.n_heads * cfg.head_dim * T * T / 2)
    res = {"pairs": pairs, "slots_per_prompt": 6, "prompt_tokens": T, "timing": "best of 2 passes", "bf16_pairs_per_s": round(pairs / out["bf16"][1], 1),
           "bf16_prefill_TFLOPs": round(flop / out["bf16"][1] / 1e12, 1), "bf16_mfma_frac_of_2.5PF": round(flop / out["bf16"][1] / 2.5e15, 3)}
    if fp8:
        y16, y8, l16, l8 = out["bf16"][0], out["fp8"][0], out["bf16"][2], out["fp8"][2]
        res.update({"fp8_pairs_per_s": round(pairs / out["fp8"][1], 1), "fp8_prefill_TFLOPs": round(flop / out["fp8"][1] / 1e12, 1),
                    "fp8_mfma_frac_of_5PF": round(flop / out["fp8"][1] / 5e15, 3),
                    "answer_logits_rel_err_fp8_vs_bf16": round(float((l8 - l16).norm() / l16.norm()), 4),
                    "yes_no_agreement_fp8_vs_bf16": round(float(((y16[:, 0] > y16[:, 1]) == (y8[:, 0] > y8[:, 1])).float().mean()), 4),
                    "mean_abs_dP_yes": float((y8[:, 0] - y16[:, 0]).abs().mean())})
    if dist:
        res.update(ranks=world, pairs_per_rank=count, scaling="strong (the pairs split across the ranks)",
                   collective="one all-gather of the [pairs, 2] probabilities")
    return res
