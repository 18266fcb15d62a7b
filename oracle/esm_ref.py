"""Oracle: ESM2 encoder arithmetic as the reference reaches it (test infrastructure).

Follows `ESM_PLM.forward` (/root/reference/procyon/model/esm.py:504-558) -> fair-esm ESM2
(`esm.py:378-403`, third-party, absent here) or HF `EsmForMaskedLM` ("official" path,
`esm.py:404-420,532-534`).  The architecture is shared:

  embed * (1-0.12)/(1-mask_ratio), <mask> rows zeroed, pad rows zeroed
  L x [ LN -> q,k,v Linear(+bias) -> q*dh^-0.5 -> rotary(q,k) -> key-padding-masked
        softmax(fp32) -> .V -> out Linear(+bias) + residual ;
        LN -> fc1(+bias) -> erf-GELU -> fc2(+bias) + residual ]
  final LayerNorm   (= representations[L], hidden_states[L] of the HF path)

Pinned bit-for-bit against the container's transformers 5.15 EsmModel (fixture g5).
Switches for the two call conventions of the reference:
  mask_pads=True  : fair-esm path (pads masked as keys, zeroed in the embedding, excluded
                    from src_lengths) == HF path called WITH attention_mask
  mask_pads=False : the reference's HF-"official" call passes no attention_mask
                    (esm.py:533)
State-dict keys use the HF naming (`esm.embeddings.word_embeddings.weight`,
`esm.encoder.layer.N...`); fair-esm checkpoints map 1:1 (procyon_amd/weights.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

PAD_ID, CLS_ID, EOS_ID, MASK_ID = 1, 0, 2, 32


@dataclass
class EsmGeom:
    d: int
    n_layers: int
    n_heads: int
    ffn: int
    vocab: int = 33
    ln_eps: float = 1e-5
    rope_theta: float = 10000.0
    # 'fp32_once': q.float()*cos + rot(q.float())*sin rounded once (HF 5.15 EsmRotary)
    # 'model_dtype': three roundings in model dtype (fair-esm rotary, [3P-INFERRED])
    rope_math: str = "fp32_once"
    rope_table: str = "fp32"  # or 'bf16_inv_freq' (HF 5.15 after .bfloat16())

    @property
    def dh(self) -> int:
        return self.d // self.n_heads


def esm_gelu(x):
    """fair-esm / HF-ESM gelu, evaluated op by op in the tensor's dtype."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def rope_tables(geom: EsmGeom, dtype, n_pos):
    dh = geom.dh
    inv_freq = 1.0 / (geom.rope_theta ** (torch.arange(0, dh, 2, dtype=torch.float) / dh))
    if geom.rope_table == "bf16_inv_freq" and dtype == torch.bfloat16:
        inv_freq = inv_freq.to(torch.bfloat16).float()
    pos = torch.arange(n_pos, dtype=torch.float)
    freqs = (inv_freq[:, None] @ pos[None, :]).transpose(0, 1)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin, math_mode):
    cos = cos[None, None]
    sin = sin[None, None]
    if math_mode == "fp32_once":
        dt = q.dtype
        qe = (q.float() * cos) + (rotate_half(q.float()) * sin)
        ke = (k.float() * cos) + (rotate_half(k.float()) * sin)
        return qe.to(dt), ke.to(dt)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def embed(sd, tokens, dtype_like, mask_pads=True):
    """EsmEmbeddings.forward with token_dropout=True, rotary positions (no abs. pos)."""
    w = sd["esm.embeddings.word_embeddings.weight"]
    x = F.embedding(tokens, w)
    x = x.masked_fill((tokens == MASK_ID).unsqueeze(-1), 0.0)
    mask_ratio_train = 0.15 * 0.8
    if mask_pads:
        attn = (tokens != PAD_ID).to(torch.int64)
        src_len = attn.sum(-1)
    else:
        attn = None
        src_len = tokens.shape[1]
    mask_ratio_obs = (tokens == MASK_ID).sum(-1).float() / src_len
    x = (x * (1 - mask_ratio_train) / (1 - mask_ratio_obs)[:, None, None]).to(x.dtype)
    if attn is not None:
        x = (x * attn.unsqueeze(-1)).to(x.dtype)
    return x


def layer_forward(h, lw, geom: EsmGeom, cos, sin, add_mask):
    B, S, d = h.shape
    H, dh = geom.n_heads, geom.dh
    x = F.layer_norm(h, (d,), lw["attention.LayerNorm.weight"], lw["attention.LayerNorm.bias"], geom.ln_eps)
    q = F.linear(x, lw["attention.self.query.weight"], lw["attention.self.query.bias"]).view(B, S, H, dh).transpose(1, 2)
    k = F.linear(x, lw["attention.self.key.weight"], lw["attention.self.key.bias"]).view(B, S, H, dh).transpose(1, 2)
    v = F.linear(x, lw["attention.self.value.weight"], lw["attention.self.value.bias"]).view(B, S, H, dh).transpose(1, 2)
    q = q * dh ** -0.5
    q, k = apply_rope(q, k, cos, sin, geom.rope_math)
    s = torch.matmul(q, k.transpose(2, 3)) * 1.0
    if add_mask is not None:
        s = s + add_mask
    p = F.softmax(s, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).contiguous().reshape(B, S, d)
    o = F.linear(o, lw["attention.output.dense.weight"], lw["attention.output.dense.bias"])
    h = o + h
    x = F.layer_norm(h, (d,), lw["LayerNorm.weight"], lw["LayerNorm.bias"], geom.ln_eps)
    x = esm_gelu(F.linear(x, lw["intermediate.dense.weight"], lw["intermediate.dense.bias"]))
    x = F.linear(x, lw["output.dense.weight"], lw["output.dense.bias"])
    return x + h


def _layer_weights(sd, i):
    p = f"esm.encoder.layer.{i}."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


@torch.no_grad()
def esm_forward(sd, geom: EsmGeom, tokens, mask_pads=True, n_layers=None):
    """tokens int64 [B,S] -> final-LN hidden [B,S,d] (== representations[repr_layer])."""
    dtype = sd["esm.embeddings.word_embeddings.weight"].dtype
    B, S = tokens.shape
    h = embed(sd, tokens, dtype, mask_pads)
    cos, sin = rope_tables(geom, dtype, S)
    add_mask = None
    if mask_pads and (tokens == PAD_ID).any():
        keep = (tokens != PAD_ID)[:, None, None, :]
        minv = torch.finfo(dtype).min
        add_mask = torch.where(keep, torch.zeros((), dtype=dtype), torch.full((), minv, dtype=dtype))
    L = geom.n_layers if n_layers is None else n_layers
    for i in range(L):
        h = layer_forward(h, _layer_weights(sd, i), geom, cos, sin, add_mask)
    h = F.layer_norm(h, (geom.d,), sd["esm.encoder.emb_layer_norm_after.weight"],
                     sd["esm.encoder.emb_layer_norm_after.bias"], geom.ln_eps)
    return h
