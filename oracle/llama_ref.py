"""Oracle: Llama decoder arithmetic as the reference reaches it (test infrastructure).

Follows `LlamaPostTokenization.forward` (/root/reference/procyon/model/pmc_llama.py:546-596)
-> HF `LlamaForCausalLM`; the layer math is cross-checked against the repo's own copy
of the 4.31 attention/model forward (pmc_llama.py:162-272, 285-412) and pinned
bit-for-bit against the container's transformers 5.15 eager implementation
(tests/golden/make_golden.py, fixture g6).

Every op is a plain torch-CPU op applied in the reference's order, so that in bf16
each rounding happens exactly where the reference rounds (after every Linear, every
residual add, the scaled scores, the softmax probabilities, each RoPE product).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class LlamaGeom:
    vocab: int
    d: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    ffn: int
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0  # pinned transformers ignores rope_theta
    max_pos: int = 8192
    # 'hf5'  : w * x_hat.to(dtype)       (transformers >= 4.32, container's 5.15)
    # 'hf431': (w * x_hat).to(dtype)     ([3P-INFERRED])
    rms_cast: str = "hf5"
    # rope tables: 'fp32' = cos/sin from fp32 inv_freq rounded once to model dtype
    # (4.31 cached tables); 'bf16_inv_freq' = 5.15 behaviour after model.bfloat16()
    rope_table: str = "fp32"
    # 'bf16' (the reference) or 'fp8': the q/k/v/o/gate/up/down projections of the PREFILL run through
    # oracle.fp8_ref.linear_fp8 (BASELINE configs[4]; no reference counterpart), cached decode steps stay bf16
    weights: str = "bf16"

    @property
    def dh(self) -> int:
        return self.d // self.n_heads


def rope_tables(geom: LlamaGeom, dtype: torch.dtype, n_pos: int | None = None):
    """cos/sin [n_pos, dh] in model dtype (HF LlamaRotaryEmbedding.forward)."""
    n_pos = n_pos or geom.max_pos
    dh = geom.dh
    inv_freq = 1.0 / (geom.rope_theta ** (torch.arange(0, dh, 2, dtype=torch.float) / dh))
    if geom.rope_table == "bf16_inv_freq" and dtype == torch.bfloat16:
        inv_freq = inv_freq.to(torch.bfloat16).float()
    pos = torch.arange(n_pos, dtype=torch.float)
    # HF: (inv_freq[None,:,None] @ pos[None,None,:]).transpose -> same products as outer()
    freqs = (inv_freq[:, None] @ pos[None, :]).transpose(0, 1)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float, cast: str = "hf5") -> torch.Tensor:
    dt = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    if cast == "hf5":
        return w * h.to(dt)
    return (w * h).to(dt)


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """q,k [B,H,T,dh]; cos/sin [B,T,dh] in model dtype. Op-by-op in model dtype."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def build_additive_mask(attn_mask_2d, B, T, t_past, dtype):
    """Additive [B,1,T,t_past+T] mask: 0 where (causal & key kept), finfo.min elsewhere.

    attn_mask_2d: [B, t_past+T] of 0/1 or None (= all ones, the decode quirk Q1:
    /root/reference/procyon/model/model_unified.py:769,887 pass no mask after step 0).
    """
    t_all = t_past + T
    q_pos = torch.arange(t_past, t_all)[:, None]
    k_pos = torch.arange(t_all)[None, :]
    allowed = (k_pos <= q_pos)[None, None].expand(B, 1, T, t_all)
    if attn_mask_2d is not None:
        keep = attn_mask_2d.to(torch.bool)[:, None, None, :]
        allowed = allowed & keep
    minv = torch.finfo(dtype).min
    return torch.where(allowed, torch.zeros((), dtype=dtype), torch.full((), minv, dtype=dtype))


def layer_forward(h, lw, geom: LlamaGeom, cos, sin, add_mask, past_kv):
    """One decoder layer.  h [B,T,d]; past_kv = (K,V) [B,Hkv,t,dh] or None."""
    B, T, _ = h.shape
    H, Hkv, dh = geom.n_heads, geom.n_kv_heads, geom.dh
    lin = F.linear
    if geom.weights == "fp8" and past_kv is None:
        from .fp8_ref import linear_fp8 as lin
    res = h
    x = rms_norm(h, lw["input_layernorm.weight"], geom.rms_eps, geom.rms_cast)
    q = lin(x, lw["self_attn.q_proj.weight"]).view(B, T, H, dh).transpose(1, 2)
    k = lin(x, lw["self_attn.k_proj.weight"]).view(B, T, Hkv, dh).transpose(1, 2)
    v = lin(x, lw["self_attn.v_proj.weight"]).view(B, T, Hkv, dh).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    if past_kv is not None:
        k = torch.cat([past_kv[0], k], dim=2)
        v = torch.cat([past_kv[1], v], dim=2)
    new_kv = (k, v)
    g = H // Hkv
    kk = k[:, :, None].expand(B, Hkv, g, k.shape[2], dh).reshape(B, H, k.shape[2], dh)
    vv = v[:, :, None].expand(B, Hkv, g, v.shape[2], dh).reshape(B, H, v.shape[2], dh)
    s = torch.matmul(q, kk.transpose(2, 3)) * (dh ** -0.5)
    if add_mask is not None:
        s = s + add_mask
    p = F.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(p, vv).transpose(1, 2).contiguous().reshape(B, T, H * dh)
    o = lin(o, lw["self_attn.o_proj.weight"])
    h = res + o
    res = h
    x = rms_norm(h, lw["post_attention_layernorm.weight"], geom.rms_eps, geom.rms_cast)
    gate = lin(x, lw["mlp.gate_proj.weight"])
    up = lin(x, lw["mlp.up_proj.weight"])
    m = lin(F.silu(gate) * up, lw["mlp.down_proj.weight"])
    return res + m, new_kv


def _layer_weights(sd, i):
    p = f"model.layers.{i}."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


@torch.no_grad()
def llama_forward(sd, geom: LlamaGeom, *, input_ids=None, inputs_embeds=None, attn_mask=None,
                  past_kv=None, logits_rows="all", want_hidden=False):
    """HF LlamaForCausalLM forward as `LlamaPostTokenization.forward` calls it.

    attn_mask: [B, t_past+T] or None.  Position ids are arange(t_past, t_past+T) for every
    row regardless of padding.
    Returns dict(logits, past_kv(list of (K,V)), hidden_states(list, L+1) if want_hidden).
    """
    assert (input_ids is None) != (inputs_embeds is None)
    emb = sd["model.embed_tokens.weight"]
    h = F.embedding(input_ids, emb) if inputs_embeds is None else inputs_embeds
    dtype = h.dtype
    B, T, _ = h.shape
    t_past = 0 if past_kv is None else past_kv[0][0].shape[2]
    cos_t, sin_t = rope_tables(geom, dtype, t_past + T)
    cos = cos_t[t_past:t_past + T][None].expand(B, T, -1)
    sin = sin_t[t_past:t_past + T][None].expand(B, T, -1)
    add_mask = build_additive_mask(attn_mask, B, T, t_past, dtype)
    hidden = []
    new_past = []
    for i in range(geom.n_layers):
        if want_hidden:
            hidden.append(h)
        h, kv = layer_forward(h, _layer_weights(sd, i), geom, cos, sin, add_mask,
                              None if past_kv is None else past_kv[i])
        new_past.append(kv)
    h = rms_norm(h, sd["model.norm.weight"], geom.rms_eps, geom.rms_cast)
    if want_hidden:
        hidden.append(h)
    if logits_rows == "last":
        logits = F.linear(h[:, -1:, :], sd["lm_head.weight"])
    else:
        logits = F.linear(h, sd["lm_head.weight"])
    return {"logits": logits, "past_kv": new_past, "hidden_states": hidden, "last_hidden": h}


@torch.no_grad()
def greedy_generate(sd, geom, inputs_embeds, attn_mask, max_len, clean_decode_mask=False):
    """`_generate_sampling(greedy=True)` (/root/reference/procyon/model/model_unified.py:861-921),
    called with keyword arguments (the positional call site at :998-1005 is broken, Q8).

    Returns tokens [B,max_len] int64, per-step logits [B,max_len,V] (model dtype),
    total log-prob [B] fp32.
    """
    B = inputs_embeds.shape[0]
    out = None
    past = None
    logits_all = []
    total_lp = torch.zeros(B)
    mask_so_far = attn_mask
    for i in range(max_len):
        if i == 0:
            r = llama_forward(sd, geom, inputs_embeds=inputs_embeds, attn_mask=attn_mask,
                              logits_rows="last")
        else:
            m = None
            if clean_decode_mask and attn_mask is not None:
                mask_so_far = torch.cat([mask_so_far, torch.ones(B, 1, dtype=mask_so_far.dtype)], 1)
                m = mask_so_far
            r = llama_forward(sd, geom, input_ids=out[:, -1:], attn_mask=m, past_kv=past,
                              logits_rows="last")
        past = r["past_kv"]
        logits = r["logits"][:, -1, :]
        logits_all.append(logits.clone())
        log_probs = F.log_softmax(logits, dim=-1)
        nxt = torch.argmax(logits, keepdim=True, dim=-1).long()
        total_lp += log_probs[torch.arange(B), nxt.squeeze(-1)].float()
        out = nxt if out is None else torch.cat([out, nxt], dim=-1)
    return out, torch.stack(logits_all, 1), total_lp


def nucleus_mask(probs, nucleus_prob):
    """`_get_nucleus_mask` (model_unified.py:844-858): ascending sort, keep where cumsum >= 1-p."""
    remove_prob = 1 - nucleus_prob
    sorted_vals, indices = probs.sort(dim=-1, descending=False)
    keep_vals = sorted_vals.cumsum(dim=-1) >= remove_prob
    keep_idxs = keep_vals.nonzero(as_tuple=True)
    keep_token_ids = indices[keep_idxs]
    mask = torch.zeros_like(probs)
    mask[keep_idxs[0], keep_token_ids] = 1
    return mask


def sampling_probs(logits, temperature=1.0, nucleus_prob=None):
    """Pre-sampling probability vector of `_generate_sampling` (:899-903)."""
    if nucleus_prob is not None:
        probs = logits.softmax(dim=-1)
        probs = probs * nucleus_mask(probs, nucleus_prob)
    else:
        probs = (logits / temperature).softmax(dim=-1)
    return probs


def sample_token(probs, u):
    """The draw of `torch.multinomial(probs, 1)` (model_unified.py:905) for GIVEN uniform variates u [B] in [0,1): inverse CDF in
    vocabulary order over the (not necessarily normalised) probabilities, token = first index whose cumulative probability
    exceeds u * total.  torch's own RNG stream is not part of the contract."""
    p = probs.double()
    cdf = p.cumsum(-1)
    target = u.double()[:, None] * cdf[:, -1:]
    return (cdf > target).float().argmax(-1)


@torch.no_grad()
def beam_search(text_encoder, input_embeds, attn_mask, *, vocab_size, eos_id, max_len=64,
                beam_size=5, beam_group_size=5, diversity_penalty=0.8, trace=None):
    """Diverse beam search of `_generate_beam_search`
    (/root/reference/procyon/model/model_unified.py:702-842).

    text_encoder(input_embeds=, input_ids=, attn_masks=, past_key_values=) -> (logits [B',T,V],
    past_key_values) where past_key_values is a list (layers) of [K, V] tensors
    [B', Hkv, t, dh] that this loop re-indexes in place, as the reference does (:830-832).

    Semantics kept: prompt replicated x beam before prefill (:751-752); step 0 takes the
    top-g of the group's first beam only (:788-795); the Hamming penalty is subtracted in
    place and therefore stays in the running score (:807-813,826; Q3); log-softmax in model
    dtype + fp32 running score (:782); no finished-beam handling, stop only when every
    beam holds an EOS somewhere in its (zero-initialised) row (:833; Q4).
    Returns tokens [B,beam,max_len] int64, scores [B,beam] fp32, logits [B,beam,steps,V].
    trace (test instrumentation, optional list): receives (step, b, group, top-(g+1) candidate scores) for every selection,
    i.e. the margins that decide whether a second implementation may legitimately pick differently.
    """
    B = input_embeds.shape[0]
    BB = B * beam_size
    if beam_size % beam_group_size != 0:
        raise ValueError("beam_group_size must evenly divide beam_size, got: "
                         f"{beam_size} % {beam_group_size} != 0")
    groups = beam_size // beam_group_size
    emb_rep = torch.repeat_interleave(input_embeds, repeats=beam_size, dim=0)
    mask_rep = torch.repeat_interleave(attn_mask, repeats=beam_size, dim=0)
    cur = torch.zeros((BB,))
    out = torch.zeros(BB, max_len, dtype=torch.int64)
    past = None
    out_logits = None
    for i in range(max_len):
        if i == 0:
            logits, past = text_encoder(input_embeds=emb_rep, attn_masks=mask_rep,
                                        past_key_values=None)
        else:
            logits, past = text_encoder(input_ids=out[:, i - 1].unsqueeze(-1),
                                        past_key_values=past)
        logits = logits[:, -1, :]
        it = logits.clone().unsqueeze(1)
        out_logits = it if out_logits is None else torch.cat([out_logits, it], dim=1)
        log_probs = F.log_softmax(logits, dim=-1) + cur[:, None]
        for b in range(B):
            beam_start = b * beam_size
            for k in range(groups):
                inc = 1 if i == 0 else beam_group_size
                gs = beam_start + k * beam_group_size
                ge = gs + beam_group_size
                lp = log_probs[gs:gs + inc]
                if k != 0:
                    prev = out[beam_start:gs, i]
                    lp -= diversity_penalty * torch.bincount(prev, minlength=vocab_size)
                top_v, top_i = lp.ravel().topk(beam_group_size)
                if trace is not None:
                    trace.append((i, b, k, lp.ravel().float().topk(beam_group_size + 1).values.clone()))
                sel = top_i % vocab_size
                orig = (top_i // vocab_size) + gs
                out[gs:ge] = out[orig]
                out[torch.arange(gs, ge), i] = sel
                cur[gs:ge] = top_v
                out_logits[gs:ge] = out_logits[orig]
                for layer in past:
                    layer[0][gs:ge] = layer[0][orig]
                    layer[1][gs:ge] = layer[1][orig]
        if torch.all((out == eos_id).any(dim=1)).item():
            break
    return (out.unflatten(0, (B, beam_size)), cur.unflatten(0, (B, beam_size)),
            out_logits.unflatten(0, (B, beam_size)))


def make_text_encoder(sd, geom):
    """Adapter with the `LlamaPostTokenization.forward` calling convention used by the
    generation loops (pmc_llama.py:546-596): returns (logits, list of [K,V])."""
    def enc(input_embeds=None, input_ids=None, attn_masks=None, past_key_values=None):
        pk = None if past_key_values is None else [(l[0], l[1]) for l in past_key_values]
        r = llama_forward(sd, geom, input_ids=input_ids, inputs_embeds=input_embeds,
                          attn_mask=attn_masks, past_kv=pk, logits_rows="all")
        return r["logits"], [[k, v] for k, v in r["past_kv"]]
    return enc
