#!/usr/bin/env python
"""Full-depth golden fixtures (run in the BUILD container only; ~64 GB of RAM, a few minutes of CPU).

The oracle (oracle/llama_ref.py, oracle/esm_ref.py -- pinned bit-for-bit to transformers 5.15 on tiny and
one-layer full-width geometries by make_golden.py) is driven over the FULL BASELINE configs[1] geometry on the
seeded synthetic weights of procyon_amd/synth.py, twice:

  * bf16  -- the reference's own arithmetic (every intermediate materialised in bf16);
  * fp32  -- the same weights (bf16 values, upcast) with every op in fp32: the "truth" both bf16 pipelines
             approximate.

Llama-3-8B, 32 layers, prompts of T = 64 and T = 512 ids: last-row logits of the prefill + 64 cached decode steps,
teacher-forced on the bf16 oracle's greedy tokens.  ESM2-650M, 33 layers, one 1024-residue protein: pooled
embedding, shared-space embedding and the soft token of the 3-layer projectors.

A logits vector has 128263 entries; the fixture keeps a column subset (every 127th column + the top-8 columns of
either run at every step) plus full-vector statistics (norms, argmax, top-8, bf16-vs-fp32 error over ALL columns),
so the files stay a few hundred KB.  The -m gpu test (tests/test_gpu_fulldepth.py) regenerates the same weights
from the same seeds and asserts  err(HIP, fp32) <= 1.25 x err(oracle_bf16, fp32).

Round 4: the same runs on DAMPED weights (`synth.damp_residual_branches`, residual branches x 0.25: the trained-like regime) --
f3_llama8b_damped_T64.npz, f4_esm650m_damped_1024.npz -- where the bf16 oracle agrees with the fp32 truth on (nearly) every argmax, so the
GPU test can assert token agreement between the HIP path and the bf16 ORACLE itself and a bound on err(HIP, oracle_bf16).

    python tests/golden/make_fulldepth.py [llama] [esm] [llama_damped] [esm_damped] [llama_leftpad] [llama256] [split] [rows10] [config3] [config4]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import esm_ref as ER  # noqa: E402
from oracle import llama_ref as LR  # noqa: E402
from oracle import procyon_ref as PR  # noqa: E402
from procyon_amd import synth  # noqa: E402

LLAMA = dict(vocab=128263, d=4096, n_layers=32, n_heads=32, n_kv_heads=8, ffn=14336)
ESM = dict(d=1280, n_layers=33, n_heads=20, ffn=5120)
NDEC = 64   # teacher-forced cached decode steps (round 3: 8 -> 64, so that argmax agreement is a rate)


def np_(t):
    if isinstance(t, torch.Tensor):
        if t.dtype == torch.bfloat16:
            return t.contiguous().view(torch.int16).numpy().view(np.uint16)
        return t.contiguous().numpy()
    return np.asarray(t)


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **{k: np_(v) for k, v in arrs.items()})
    sz = os.path.getsize(os.path.join(HERE, name + ".npz"))
    print(f"wrote {name}.npz ({sz / 1024:.0f} KB)", {k: tuple(np_(v).shape) for k, v in arrs.items()}, flush=True)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


@torch.no_grad()
def llama_run(sd, geom, ids, dtype, forced=None, ndec=None):
    """prefill + NDEC cached steps through the oracle's own layer_forward, weights cast per layer to `dtype`
    (bf16: no-op).  forced: tokens to feed (teacher forcing) or None = greedy on this run's own logits.
    Returns logits [NDEC+1, V], final-normed hidden rows [NDEC+1, d], tokens [NDEC+1]."""
    NDEC = globals()["NDEC"] if ndec is None else ndec
    c = lambda t: t.to(dtype)
    emb_w = sd["model.embed_tokens.weight"]
    T = ids.shape[1]
    past = [None] * geom.n_layers
    logits_all, hid_all, toks = [], [], []
    x_ids = ids
    for step in range(NDEC + 1):
        h = c(F.embedding(x_ids, emb_w))
        B, Tq, _ = h.shape
        t_past = 0 if past[0] is None else past[0][0].shape[2]
        cos_t, sin_t = LR.rope_tables(geom, dtype, t_past + Tq)
        cos = cos_t[t_past:t_past + Tq][None].expand(B, Tq, -1)
        sin = sin_t[t_past:t_past + Tq][None].expand(B, Tq, -1)
        add_mask = LR.build_additive_mask(None, B, Tq, t_past, dtype)
        for i in range(geom.n_layers):
            lw = {k: c(v) for k, v in LR._layer_weights(sd, i).items()}
            h, past[i] = LR.layer_forward(h, lw, geom, cos, sin, add_mask, past[i])
        h = LR.rms_norm(h, c(sd["model.norm.weight"]), geom.rms_eps, geom.rms_cast)
        last = h[:, -1]
        lg = F.linear(last, c(sd["lm_head.weight"]))[0]
        logits_all.append(lg)
        hid_all.append(last[0])
        tok = int(lg.argmax()) if forced is None else int(forced[step])
        toks.append(tok)
        x_ids = torch.tensor([[tok]])
        print(f"    {dtype} step {step}: argmax {int(lg.argmax())} fed {tok}", flush=True)
    return torch.stack(logits_all), torch.stack(hid_all), torch.tensor(toks)


@torch.no_grad()
def llama_run_rows(sd, geom, ids, mask, dtype, ndec, forced=None):
    """B left-padded rows through the reference's own call pattern: prefill with the attention mask and
    positions arange(T) for EVERY row whatever its padding (pmc_llama.py:546-588), then cached steps with attention_mask = None -- every
    cached slot, left pads included, is attended -- at position = cache length (model_unified.py:769, :887).  Weights cast per layer to
    `dtype`.  -> logits [ndec+1, B, V], tokens [ndec+1, B]."""
    c = lambda t: t.to(dtype)
    emb_w = sd["model.embed_tokens.weight"]
    B = ids.shape[0]
    past = [None] * geom.n_layers
    logits_all, toks = [], []
    x_ids = ids
    for step in range(ndec + 1):
        h = c(F.embedding(x_ids, emb_w))
        _, Tq, _ = h.shape
        t_past = 0 if past[0] is None else past[0][0].shape[2]
        cos_t, sin_t = LR.rope_tables(geom, dtype, t_past + Tq)
        cos = cos_t[t_past:t_past + Tq][None].expand(B, Tq, -1)
        sin = sin_t[t_past:t_past + Tq][None].expand(B, Tq, -1)
        add_mask = LR.build_additive_mask(mask if step == 0 else None, B, Tq, t_past, dtype)
        for i in range(geom.n_layers):
            lw = {k: c(v) for k, v in LR._layer_weights(sd, i).items()}
            h, past[i] = LR.layer_forward(h, lw, geom, cos, sin, add_mask, past[i])
        h = LR.rms_norm(h[:, -1], c(sd["model.norm.weight"]), geom.rms_eps, geom.rms_cast)
        lg = F.linear(h, c(sd["lm_head.weight"]))
        logits_all.append(lg)
        tok = lg.argmax(-1) if forced is None else forced[step]
        toks.append(tok)
        x_ids = tok[:, None]
        print(f"    {dtype} step {step}: argmax {lg.argmax(-1).tolist()} fed {tok.tolist()}", flush=True)
    return torch.stack(logits_all), torch.stack(toks)


def llama_truth(sd, geom, ids, toks):
    """fp32 truth of the same NDEC+1 positions in ONE causal pass over ids ++ toks[:-1] (teacher forcing makes every input
    known up front; a causal prefill and an incremental decode are the same function, and in fp32 their difference
    (~1e-6) is far below what the comparison resolves).  Returns logits [NDEC+1, V], hidden rows [NDEC+1, d]."""
    dtype = torch.float32
    c = lambda t: t.to(dtype)
    full = torch.cat([ids, toks[None, :-1]], dim=1)
    T0, Tq = ids.shape[1], full.shape[1]
    h = c(F.embedding(full, sd["model.embed_tokens.weight"]))
    cos_t, sin_t = LR.rope_tables(geom, dtype, Tq)
    cos, sin = cos_t[None, :Tq], sin_t[None, :Tq]
    add_mask = LR.build_additive_mask(None, 1, Tq, 0, dtype)
    for i in range(geom.n_layers):
        lw = {k: c(v) for k, v in LR._layer_weights(sd, i).items()}
        h, _ = LR.layer_forward(h, lw, geom, cos, sin, add_mask, None)
    h = LR.rms_norm(h[:, T0 - 1:], c(sd["model.norm.weight"]), geom.rms_eps, geom.rms_cast)[0]
    return F.linear(h, c(sd["lm_head.weight"])), h


def make_llama_leftpad():
    """f5: two rows, the second left-padded by 21 of 64 slots, 16 teacher-forced cached steps in the reference's compat mode (Q1 / Q2) at
    FULL depth: bf16 oracle and the same procedure in fp32 (a single causal pass is not the same function here: the pad slots' K / V rows
    and the position offset are part of what is checked)."""
    sd = synth.llama_state_dict(**LLAMA)
    geom = LR.LlamaGeom(**LLAMA, max_pos=4096)
    T, ndec, npad = 64, 16, 21
    g = torch.Generator().manual_seed(5151)
    ids = torch.randint(0, 128000, (2, T), generator=g)
    mask = torch.ones(2, T)
    mask[1, :npad] = 0
    ids[1, :npad] = 128001            # whatever sits in the pad slots is embedded and cached like any token
    t0 = time.time()
    lb, toks = llama_run_rows(sd, geom, ids, mask, torch.bfloat16, ndec)
    print(f"  leftpad bf16 oracle {time.time() - t0:.0f}s", flush=True)
    t0 = time.time()
    lf, _ = llama_run_rows(sd, geom, ids, mask, torch.float32, ndec, forced=toks)
    print(f"  leftpad fp32 truth {time.time() - t0:.0f}s", flush=True)
    V = lb.shape[-1]
    cols = set(range(0, V, 127))
    for s in range(ndec + 1):
        for b in range(2):
            cols |= set(lf[s, b].topk(8).indices.tolist()) | set(lb[s, b].float().topk(8).indices.tolist())
    cols = torch.tensor(sorted(cols))
    top_f, top_b = lf.topk(8, dim=-1), lb.float().topk(8, dim=-1)
    err_full = torch.tensor([[rel(lb[s, b].float(), lf[s, b]) for b in range(2)] for s in range(ndec + 1)])
    print(f"  leftpad: bf16-vs-fp32 err {err_full.tolist()}\n   argmax agree {(lb.float().argmax(-1) == lf.argmax(-1)).tolist()}")
    save("f5_llama8b_leftpad_T64", ids=ids.to(torch.int32), mask=mask.to(torch.int32), tokens=toks.to(torch.int32), cols=cols.to(torch.int32),
         logits_bf16=lb[..., cols], logits_fp32=lf[..., cols], norm_fp32=lf.double().norm(dim=-1).float(), err_bf16_full=err_full.float(),
         top_ids_fp32=top_f.indices.to(torch.int32), top_vals_fp32=top_f.values, top_ids_bf16=top_b.indices.to(torch.int32), top_vals_bf16=top_b.values)


def make_llama_rows10():
    """f7 (round 6): TEN ragged left-padded rows (pads 0 .. 27 of 64 slots), 16 teacher-forced cached steps in the reference's compat mode
    (Q1 / Q2) at FULL depth -- the batch of the reference's beam-10 callers (scripts/caption_bulk.py:193-194, evaluate/framework/procyon.py:72-76),
    which runs on the mid-batch decode step (pcy_decode_mb.hip); the first 5 / 8 rows as batches of their own hold the small-batch step
    (pcy_decode_nb.hip) to the same oracle (rows are independent of their batch mates)."""
    sd = synth.llama_state_dict(**LLAMA)
    geom = LR.LlamaGeom(**LLAMA, max_pos=4096)
    B, T, ndec = 10, 64, 16
    g = torch.Generator().manual_seed(7171)
    ids = torch.randint(0, 128000, (B, T), generator=g)
    npad = [0, 21, 3, 0, 11, 27, 8, 0, 16, 5]
    mask = torch.ones(B, T)
    for b, n in enumerate(npad):
        mask[b, :n] = 0
        ids[b, :n] = 128001
    t0 = time.time()
    lb, toks = llama_run_rows(sd, geom, ids, mask, torch.bfloat16, ndec)
    print(f"  rows10 bf16 oracle {time.time() - t0:.0f}s", flush=True)
    t0 = time.time()
    lf, _ = llama_run_rows(sd, geom, ids, mask, torch.float32, ndec, forced=toks)
    print(f"  rows10 fp32 truth {time.time() - t0:.0f}s", flush=True)
    V = lb.shape[-1]
    cols = set(range(0, V, 127))
    for s in range(ndec + 1):
        for b in range(B):
            cols |= set(lf[s, b].topk(8).indices.tolist()) | set(lb[s, b].float().topk(8).indices.tolist())
    cols = torch.tensor(sorted(cols))
    top_f, top_b = lf.topk(8, dim=-1), lb.float().topk(8, dim=-1)
    err_full = torch.tensor([[rel(lb[s, b].float(), lf[s, b]) for b in range(B)] for s in range(ndec + 1)])
    print(f"  rows10: bf16-vs-fp32 err mean {float(err_full.mean()):.3e} max {float(err_full.max()):.3e}\n   argmax agree {(lb.float().argmax(-1) == lf.argmax(-1)).float().mean():.3f}")
    save("f7_llama8b_rows10_T64", ids=ids.to(torch.int32), mask=mask.to(torch.int32), tokens=toks.to(torch.int32), cols=cols.to(torch.int32),
         logits_bf16=lb[..., cols], logits_fp32=lf[..., cols], norm_fp32=lf.double().norm(dim=-1).float(), err_bf16_full=err_full.float(),
         top_ids_fp32=top_f.indices.to(torch.int32), top_vals_fp32=top_f.values, top_ids_bf16=top_b.indices.to(torch.int32), top_vals_bf16=top_b.values)


def make_config3_rows():
    """f8 (round 6): BASELINE configs[3] 4b at full geometry -- 32 ragged rows (T uniform in [128, 512], seed 7, left-padded to 512), compat mode.
    The oracle runs THREE sampled rows (0, 13, 31: rows are independent) for 8 teacher-forced steps; the GPU test runs all 32 rows as one batch
    (prefill + the 32-row decode step) and compares those rows."""
    sd = synth.llama_state_dict(**LLAMA)
    geom = LR.LlamaGeom(**LLAMA, max_pos=4096)
    B, T, ndec = 32, 512, 8
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(128, 513, (B,), generator=g)
    ids = torch.randint(0, 128000, (B, T), generator=g)
    mask = torch.ones(B, T)
    for b in range(B):
        n = T - int(lens[b])
        mask[b, :n] = 0
        ids[b, :n] = 128001
    rows = torch.tensor([0, 13, 31])
    t0 = time.time()
    lb, toks = llama_run_rows(sd, geom, ids[rows], mask[rows], torch.bfloat16, ndec)
    print(f"  config3 rows bf16 oracle {time.time() - t0:.0f}s", flush=True)
    t0 = time.time()
    lf, _ = llama_run_rows(sd, geom, ids[rows], mask[rows], torch.float32, ndec, forced=toks)
    print(f"  config3 rows fp32 truth {time.time() - t0:.0f}s", flush=True)
    V = lb.shape[-1]
    cols = set(range(0, V, 127))
    for s in range(ndec + 1):
        for b in range(len(rows)):
            cols |= set(lf[s, b].topk(8).indices.tolist()) | set(lb[s, b].float().topk(8).indices.tolist())
    cols = torch.tensor(sorted(cols))
    top_f, top_b = lf.topk(8, dim=-1), lb.float().topk(8, dim=-1)
    err_full = torch.tensor([[rel(lb[s, b].float(), lf[s, b]) for b in range(len(rows))] for s in range(ndec + 1)])
    print(f"  config3 rows: bf16-vs-fp32 err mean {float(err_full.mean()):.3e} max {float(err_full.max()):.3e}")
    save("f8_config3_rows_T512", ids=ids.to(torch.int32), mask=mask.to(torch.int32), rows=rows.to(torch.int32), tokens=toks.to(torch.int32), cols=cols.to(torch.int32),
         logits_bf16=lb[..., cols], logits_fp32=lf[..., cols], norm_fp32=lf.double().norm(dim=-1).float(), err_bf16_full=err_full.float(),
         top_ids_fp32=top_f.indices.to(torch.int32), top_vals_fp32=top_f.values, top_ids_bf16=top_b.indices.to(torch.int32), top_vals_bf16=top_b.values)


@torch.no_grad()
def make_config4_pair():
    """f9 (round 6): ONE pair of BASELINE configs[4] at full geometry in bf16 -- six <|protein|> slots (receptor 805 residues + three peptides of
    8 .. 40, slots [0, 1, 0, 2, 0, 3] like procyon_amd/workloads.config5_inputs), a 437-token prompt ending in [ANSWER]: ESM2-650M (33 layers)
    -> mean pool -> 3-layer token projector -> splice -> Llama-3-8B prefill (32 layers) -> the answer row's logits, P(yes) / P(no); bf16 oracle
    and the same pipeline in fp32."""
    PROT, ANSWER, YES, NO = 128258, 128260, 9891, 2201
    T = 437
    plen = [805, 17, 33, 9]
    prot = synth.protein_tokens(plen, seed=11)
    slots = [0, 1, 0, 2, 0, 3]
    ids = synth.prompt_ids(1, T, 128000, dict(protein=PROT, answer=ANSWER), n_protein=6, seed=41)
    esd = synth.esm_state_dict(**ESM)
    proj = synth.mlp_layers(3, 1280, 4096, 2560, 0)
    lsd = synth.llama_state_dict(**LLAMA)
    geom = LR.LlamaGeom(**LLAMA, max_pos=4096)
    out = {}
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        t0 = time.time()
        z = PR.esm_plm_forward({k: v.to(dt) for k, v in esd.items()}, ER.EsmGeom(**ESM), prot)
        soft = PR.mlp_forward(z[slots], [(w.to(dt), b.to(dt)) for w, b in proj])
        emb, _ = PR.prepare_input_embeddings(lsd["model.embed_tokens.weight"].to(dt), ids, PROT, soft)
        print(f"  config4 pair {name}: esm + projector {time.time() - t0:.0f}s", flush=True)
        c = lambda t: t.to(dt)
        h = emb
        cos_t, sin_t = LR.rope_tables(geom, dt, T)
        cos, sin = cos_t[None, :T], sin_t[None, :T]
        add_mask = LR.build_additive_mask(None, 1, T, 0, dt)
        for i in range(geom.n_layers):
            lw = {k: c(v) for k, v in LR._layer_weights(lsd, i).items()}
            h, _ = LR.layer_forward(h, lw, geom, cos, sin, add_mask, None)
        h = LR.rms_norm(h[:, -1], c(lsd["model.norm.weight"]), geom.rms_eps, geom.rms_cast)
        lg = F.linear(h, c(lsd["lm_head.weight"]))[0]
        p = lg.softmax(-1)
        out[name] = (z, soft, lg, p[YES], p[NO])
        print(f"  config4 pair {name}: total {time.time() - t0:.0f}s  P(yes) {float(p[YES]):.3e} P(no) {float(p[NO]):.3e}", flush=True)
    lb, lf = out["bf16"][2], out["fp32"][2]
    cols = torch.tensor(sorted(set(range(0, lb.shape[0], 127)) | set(lf.topk(8).indices.tolist()) | set(lb.float().topk(8).indices.tolist()) | {YES, NO}))
    print(f"  config4 pair: bf16-vs-fp32 logits {rel(lb.float(), lf):.3e} pooled {rel(out['bf16'][0].float(), out['fp32'][0]):.3e} soft {rel(out['bf16'][1].float(), out['fp32'][1]):.3e}")
    save("f9_config4_pair", protein_tokens=prot.to(torch.int32), slots=torch.tensor(slots, dtype=torch.int32), ids=ids.to(torch.int32),
         special=torch.tensor([PROT, ANSWER, YES, NO], dtype=torch.int32), cols=cols.to(torch.int32),
         pooled_bf16=out["bf16"][0], pooled_fp32=out["fp32"][0], soft_bf16=out["bf16"][1], soft_fp32=out["fp32"][1],
         logits_bf16=lb[cols], logits_fp32=lf[cols], err_bf16_full=torch.tensor(rel(lb.float(), lf)), norm_fp32=lf.double().norm().float(),
         top_ids_fp32=lf.topk(8).indices.to(torch.int32), top_ids_bf16=lb.float().topk(8).indices.to(torch.int32),
         p_yes_no_bf16=torch.stack([out["bf16"][3], out["bf16"][4]]).float(), p_yes_no_fp32=torch.stack([out["fp32"][3], out["fp32"][4]]))


def make_llama(damped=False, long=False):
    global NDEC
    if long:       # round 5: the headline's generation length (256 tokens) at T = 512 -> f1_llama8b_T512_N256.npz
        NDEC = 256
    t0 = time.time()
    sd = synth.llama_state_dict(**LLAMA)
    if damped:   # the trained-like regime (synth.damp_residual_branches): fixtures f3_*, on which token agreement is a meaningful rate
        synth.damp_residual_branches(sd, 0.25)
    print(f"llama weights generated in {time.time() - t0:.0f}s", flush=True)
    geom = LR.LlamaGeom(**LLAMA, max_pos=4096)
    for T in ((512,) if long else (64,) if damped else (64, 512)):
        g = torch.Generator().manual_seed(4242 + T)
        ids = torch.randint(0, 128000, (1, T), generator=g)
        t0 = time.time()
        lb, hb, toks = llama_run(sd, geom, ids, torch.bfloat16)
        print(f"  T={T} bf16 oracle {time.time() - t0:.0f}s", flush=True)
        t0 = time.time()
        lf, hf = llama_truth(sd, geom, ids, toks)
        print(f"  T={T} fp32 truth {time.time() - t0:.0f}s", flush=True)
        V = lb.shape[1]
        cols = set(range(0, V, 127))
        for s in range(NDEC + 1):
            cols |= set(lf[s].topk(8).indices.tolist()) | set(lb[s].float().topk(8).indices.tolist())
        cols = torch.tensor(sorted(cols))
        top_f = lf.topk(8, dim=-1)
        top_b = lb.float().topk(8, dim=-1)
        err_full = torch.tensor([rel(lb[s].float(), lf[s]) for s in range(NDEC + 1)])
        err_cols = torch.tensor([rel(lb[s, cols].float(), lf[s, cols]) for s in range(NDEC + 1)])
        print(f"  T={T}: bf16-vs-fp32 logits err full {err_full.tolist()}\n         cols {err_cols.tolist()}")
        print(f"         argmax agree {(lb.float().argmax(-1) == lf.argmax(-1)).tolist()} "
              f"fp32 top-2 margin {(top_f.values[:, 0] - top_f.values[:, 1]).tolist()}")
        save(f"f1_llama8b_T{T}_N{NDEC}" if long else f"f3_llama8b_damped_T{T}" if damped else f"f1_llama8b_T{T}", ids=ids.to(torch.int32), tokens=toks.to(torch.int32), cols=cols.to(torch.int32),
             logits_bf16=lb[:, cols], logits_fp32=lf[:, cols], hidden_bf16=hb[:1], hidden_fp32=hf[:1],
             norm_fp32=lf.double().norm(dim=-1).float(), err_bf16_full=err_full.float(),
             top_ids_fp32=top_f.indices.to(torch.int32), top_vals_fp32=top_f.values,
             top_ids_bf16=top_b.indices.to(torch.int32), top_vals_bf16=top_b.values)


@torch.no_grad()
def make_esm(damped=False):
    esd = synth.esm_state_dict(**ESM)
    if damped:
        synth.damp_residual_branches(esd, 0.25)
    shared = synth.mlp_layers(3, 1280, 1280, 2560, 20)
    token = synth.mlp_layers(3, 1280, 4096, 2560, 0)
    toks = synth.protein_tokens([1024], seed=77)
    out = {}
    for name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        t0 = time.time()
        sd = {k: v.to(dt) for k, v in esd.items()}
        h = ER.esm_forward(sd, ER.EsmGeom(**ESM), toks)
        z = PR.protein_pooler(h, torch.zeros(1, dtype=torch.int64), toks == 1, method="mean")
        sh = PR.mlp_forward(z, [(w.to(dt), b.to(dt)) for w, b in shared])
        tk = PR.mlp_forward(z, [(w.to(dt), b.to(dt)) for w, b in token])
        out[name] = (h[0, [0, 1, 511, 1024, 1025]], z[0], sh[0], tk[0])
        print(f"  esm {name}: {time.time() - t0:.0f}s", flush=True)
    names = ("hidden_rows", "pooled", "shared", "soft_token")
    for n, a, b in zip(names, out["bf16"], out["fp32"]):
        print(f"  esm {n}: bf16-vs-fp32 {rel(a.float(), b):.3e}")
    save("f4_esm650m_damped_1024" if damped else "f2_esm650m_1024", tokens=toks.to(torch.int32), rows=torch.tensor([0, 1, 511, 1024, 1025], dtype=torch.int32),
         **{f"{n}_bf16": a for n, a in zip(names, out["bf16"])}, **{f"{n}_fp32": b for n, b in zip(names, out["fp32"])})


SPLIT_LLAMA = dict(vocab=32007, d=4096, n_layers=32, n_heads=32, n_kv_heads=32, ffn=11008)    # Llama-2-7B + 8 added tokens - [EXT] (model_unified.py:166)
SPLIT_ESM = dict(d=640, n_layers=30, n_heads=20, ffn=2560)                                       # ESM2-150M


@torch.no_grad()
def make_split():
    """BASELINE configs[0] "full once" (SURVEY.md section 8d, config 1): ProCyon-Split geometry -- ESM2-150M + Llama-2-7B, ALL layers, fp32 --
    one 256-residue protein (seed 0), a 128-token prompt with one <|protein|> and a trailing [ANSWER], 64 greedy tokens through the oracle:
    ESM -> mean pool -> 3-layer token projector -> splice -> prefill + 63 cached steps (`_generate_sampling(greedy=True)`).  Fixture f6: pooled
    embedding, soft token, tokens, per-step logits on a column subset + top-2 of every step."""
    t0 = time.time()
    dt = torch.float32
    esd = synth.esm_state_dict(**SPLIT_ESM, dtype=dt)
    proj = synth.mlp_layers(3, 640, 4096, 2560, seed_off=0, dtype=dt)
    toks = synth.protein_tokens([256], seed=0)
    z = PR.esm_plm_forward(esd, ER.EsmGeom(**SPLIT_ESM), toks)
    soft = PR.mlp_forward(z, proj)
    print(f"split: esm + projector {time.time() - t0:.0f}s", flush=True)
    del esd
    PROT, ANSWER = 32001, 32003          # <|protein|>, [ANSWER] among the added tokens (ids 32000 .. 32006)
    ids = synth.prompt_ids(1, 128, 32000, dict(protein=PROT, answer=ANSWER), n_protein=1, seed=0)
    ids[ids < 3] = 3
    lsd = synth.llama_state_dict(**SPLIT_LLAMA, dtype=dt)
    geom = LR.LlamaGeom(**SPLIT_LLAMA, max_pos=4096)
    emb, _ = PR.prepare_input_embeddings(lsd["model.embed_tokens.weight"], ids, PROT, soft)
    print(f"split: llama weights {time.time() - t0:.0f}s", flush=True)
    tok, lg, lp = LR.greedy_generate(lsd, geom, emb, torch.ones(1, 128), 64)
    print(f"split: 64 greedy tokens {time.time() - t0:.0f}s  tokens {tok[0].tolist()}", flush=True)
    lg = lg[0]                                            # [64, V]
    V = lg.shape[1]
    cols = set(range(0, V, 31))
    for s_ in range(64):
        cols |= set(lg[s_].topk(8).indices.tolist())
    cols = torch.tensor(sorted(cols))
    top = lg.topk(8, dim=-1)
    save("f6_split_config0_fp32", protein_tokens=toks.to(torch.int32), prompt_ids=ids.to(torch.int32), pooled=z[0], soft_token=soft[0],
         tokens=tok[0].to(torch.int32), logprob=lp.float(), cols=cols.to(torch.int32), logits=lg[:, cols], norm=lg.double().norm(dim=-1).float(),
         top_ids=top.indices.to(torch.int32), top_vals=top.values)


if __name__ == "__main__":
    what = sys.argv[1:] or ["esm", "llama"]
    torch.set_num_threads(os.cpu_count())
    if "esm" in what:
        make_esm()
    if "llama" in what:
        make_llama()
    if "esm_damped" in what:
        make_esm(damped=True)
    if "llama_damped" in what:
        make_llama(damped=True)
    if "llama_leftpad" in what:
        make_llama_leftpad()
    if "llama256" in what:
        make_llama(long=True)
    if "split" in what:
        make_split()
    if "rows10" in what:
        make_llama_rows10()
    if "config3" in what:
        make_config3_rows()
    if "config4" in what:
        make_config4_pair()
