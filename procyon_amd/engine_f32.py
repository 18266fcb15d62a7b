"""fp32 engines: the arithmetic of the reference's callers that never call `.bfloat16()`.

/root/reference/examples/paper_analyses/protpep_qa_scores.py:55-58 (`model.eval(); model.to(device)` -- the loop that defines BASELINE
configs[4]), /root/reference/scripts/qa_filter_captions.py:17-18 and /root/reference/scripts/caption_bulk.py:72-73 run the model as
loaded: fp32 weights, every torch op in fp32.  These engines string the fp32 operator family of libpcy.so (`pcy_f32_*`,
include/pcy.h; procyon_amd/csrc/pcy_f32.hip) together op for op like the reference's modules in that mode:

  EsmEngineF32    ESM2 encoder over packed sequences + ProteinPooler            (/root/reference/procyon/model/esm.py:131-173,504-538)
  MlpEngineF32    `create_mlp` projectors                                        (/root/reference/procyon/model/model_utils.py:13-41)
  LlamaEngineF32  token embedding + soft-token splice, decoder PREFILL, logits   (/root/reference/procyon/model/pmc_llama.py:546-596,
                  at chosen rows, final hidden states, the L+1-state sum          /root/reference/procyon/model/model_unified.py:556-581)

Cached decode is not built in fp32 (generation stays a bf16 path; the model mirror raises a clear error).  PyTorch holds the device
memory and does index bookkeeping only; every floating-point operation is a HIP kernel behind the C ABI.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .engine import Context, EsmConfig, EsmEngine, LlamaConfig, _h2d_many, batched_split_long_seq

F32 = torch.float32


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _chk(*ts):
    for t in ts:
        if t is not None and (t.dtype != F32 or not t.is_cuda or not t.is_contiguous()):
            raise TypeError(f"fp32 engine: expected a contiguous fp32 device tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


class F32Ops:
    """thin typed wrappers over the pcy_f32_* entry points"""

    def __init__(self, ctx=None, device=None):
        self.ctx = ctx or Context.get(device)
        self.lib, self.h, self.device = self.ctx.lib, self.ctx.h, self.ctx.device

    def linear(self, x, w, bias=None, resid=None, act=0, out=None):
        _chk(x, w, bias, resid)
        M, K = x.shape
        N = w.shape[0]
        out = torch.empty(M, N, dtype=F32, device=x.device) if out is None else out
        L.check(self.lib.pcy_f32_linear(self.h, _p(x), K, _p(w), _p(bias), _p(resid), 0 if resid is None else resid.shape[1], _p(out), N, M, N, K, act),
                "pcy_f32_linear")
        return out

    def layernorm(self, x, w, b, eps):
        _chk(x, w, b)
        y = torch.empty_like(x)
        L.check(self.lib.pcy_f32_layernorm(self.h, _p(x), _p(w), _p(b), _p(y), x.numel() // x.shape[-1], x.shape[-1], eps), "pcy_f32_layernorm")
        return y

    def rmsnorm(self, x, w, eps):
        _chk(x, w)
        y = torch.empty_like(x)
        L.check(self.lib.pcy_f32_rmsnorm(self.h, _p(x), _p(w), _p(y), x.numel() // x.shape[-1], x.shape[-1], eps), "pcy_f32_rmsnorm")
        return y

    def rope(self, buf, col0, nh, dh, pos, cos, sin, prescale=0.0):
        _chk(buf, cos, sin)
        L.check(self.lib.pcy_f32_rope(self.h, _p(buf), buf.shape[1], col0, nh, dh, _p(pos), _p(cos), _p(sin), buf.shape[0], prescale), "pcy_f32_rope")

    def attention(self, q, qcol0, k, kcol0, v, vcol0, cu, keep, nseq, max_len, H, Hkv, dh, causal, scale):
        _chk(q, k, v)
        o = torch.empty(q.shape[0], H * dh, dtype=F32, device=q.device)
        L.check(self.lib.pcy_f32_attention(self.h, _p(q), q.shape[1], qcol0, _p(k), k.shape[1], kcol0, _p(v), v.shape[1], vcol0, _p(o), H * dh,
                                           _p(cu), _p(keep), nseq, max_len, H, Hkv, dh, int(causal), scale), "pcy_f32_attention")
        return o

    def attn_decode(self, q, kcache, vcache, nkeys, H, Hkv, dh, scale):
        """q [B, H*dh] (a view with row stride q.stride(0) is fine) against slots [0, nkeys) of token-major caches [B, Tmax, Hkv*dh]"""
        _chk(kcache, vcache)
        B = q.shape[0]
        o = torch.empty(B, H * dh, dtype=F32, device=q.device)
        L.check(self.lib.pcy_f32_attn_decode(self.h, _p(q), q.stride(0), _p(kcache), _p(vcache), kcache.shape[2], kcache.shape[1], _p(o), H * dh, B,
                                             H, Hkv, dh, nkeys, scale), "pcy_f32_attn_decode")
        return o

    def embed(self, table, ids, soft=None, soft_map=None):
        _chk(table, soft)
        out = torch.empty(ids.numel(), table.shape[1], dtype=F32, device=table.device)
        L.check(self.lib.pcy_f32_embed(self.h, _p(table), _p(ids), _p(soft), _p(soft_map), _p(out), ids.numel(), table.shape[1]), "pcy_f32_embed")
        return out

    def silu_mul(self, g, u):
        _chk(g, u)
        o = torch.empty_like(g)
        L.check(self.lib.pcy_f32_silu_mul(self.h, _p(g), _p(u), _p(o), g.numel()), "pcy_f32_silu_mul")
        return o

    def acc_rows(self, dst, src, rows):
        _chk(dst, src)
        L.check(self.lib.pcy_f32_acc_rows(self.h, _p(src), src.shape[-1], _p(rows), _p(dst), rows.numel(), src.shape[-1]), "pcy_f32_acc_rows")


def _rope_tables_f32(dh, theta, n_pos, device):
    inv_freq = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    pos = torch.arange(n_pos, dtype=torch.float32)
    freqs = (inv_freq[:, None] @ pos[None, :]).transpose(0, 1)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(device).contiguous(), emb.sin().to(device).contiguous()


class MlpEngineF32:
    """`create_mlp` in eval mode: Linear(+bias) -> GELU ... -> Linear(+bias); one bias-free Linear when n_layers == 1"""

    def __init__(self, layers, ctx=None, device=None):
        self.ops = F32Ops(ctx, device)
        dev = self.ops.device
        self.layers = [(w.to(dev, F32).contiguous(), None if b is None else b.to(dev, F32).contiguous()) for w, b in layers]
        self.in_features, self.out_features = self.layers[0][0].shape[1], self.layers[-1][0].shape[0]

    def __call__(self, x):
        shape = x.shape
        x = x.reshape(-1, shape[-1]).to(self.ops.device, F32).contiguous()
        n = len(self.layers)
        for i, (w, b) in enumerate(self.layers):
            if x.shape[0]:
                x = self.ops.linear(x, w, b, act=1 if i < n - 1 else 0)
            else:
                x = torch.empty(0, w.shape[0], dtype=F32, device=x.device)
        return x.reshape(*shape[:-1], self.out_features)


class EsmEngineF32:
    """ESM2 (HF layout state dict, fp32) over packed varlen sequences + ProteinPooler; same call surface as `EsmEngine.forward`."""

    def __init__(self, sd, cfg: EsmConfig, device=None, ctx=None):
        self.ops = F32Ops(ctx, device)
        self.cfg, self.device = cfg, self.ops.device
        g = lambda k: sd[k].to(self.device, F32).contiguous()
        self.embed = g("esm.embeddings.word_embeddings.weight")
        self.fw, self.fb = g("esm.encoder.emb_layer_norm_after.weight"), g("esm.encoder.emb_layer_norm_after.bias")
        self.cos, self.sin = _rope_tables_f32(cfg.head_dim, cfg.rope_theta, cfg.max_len, self.device)
        self.layers = []
        for l in range(cfg.n_layers):
            p = f"esm.encoder.layer.{l}."
            self.layers.append(dict(
                wqkv=torch.cat([g(p + f"attention.self.{n}.weight") for n in ("query", "key", "value")], 0).contiguous(),
                bqkv=torch.cat([g(p + f"attention.self.{n}.bias") for n in ("query", "key", "value")], 0).contiguous(),
                wo=g(p + "attention.output.dense.weight"), bo=g(p + "attention.output.dense.bias"),
                ln1w=g(p + "attention.LayerNorm.weight"), ln1b=g(p + "attention.LayerNorm.bias"),
                w1=g(p + "intermediate.dense.weight"), b1=g(p + "intermediate.dense.bias"),
                w2=g(p + "output.dense.weight"), b2=g(p + "output.dense.bias"), ln2w=g(p + "LayerNorm.weight"), ln2b=g(p + "LayerNorm.bias")))

    def encode_packed(self, pk, mask_pads=True):
        ops, cfg = self.ops, self.cfg
        if pk["max_len"] > cfg.max_len:
            raise ValueError(f"sequence of {pk['max_len']} tokens exceeds the rotary table ({cfg.max_len})")
        tokens, pos, cu = _h2d_many([pk["tokens"], pk["pos"], pk["cu"]], self.device)
        d, H, dh = cfg.d, cfg.n_heads, cfg.head_dim
        x = torch.empty(pk["ntok"], d, dtype=F32, device=self.device)
        L.check(ops.lib.pcy_f32_esm_embed(ops.h, _p(self.embed), _p(tokens), _p(cu), pk["nseq"], pk["max_len"], _p(x), d, int(mask_pads)), "pcy_f32_esm_embed")
        for lw in self.layers:
            xn = ops.layernorm(x, lw["ln1w"], lw["ln1b"], cfg.ln_eps)
            qkv = ops.linear(xn, lw["wqkv"], lw["bqkv"])
            ops.rope(qkv, 0, H, dh, pos, self.cos, self.sin, prescale=dh ** -0.5)       # q * dh^-1/2, then rotated (esm2 attention)
            ops.rope(qkv, d, H, dh, pos, self.cos, self.sin)
            ao = ops.attention(qkv, 0, qkv, d, qkv, 2 * d, cu, None, pk["nseq"], pk["max_len"], H, H, dh, False, 1.0)
            x = ops.linear(ao, lw["wo"], lw["bo"], resid=x)
            xn = ops.layernorm(x, lw["ln2w"], lw["ln2b"], cfg.ln_eps)
            act = ops.linear(xn, lw["w1"], lw["b1"], act=1)
            x = ops.linear(act, lw["w2"], lw["b2"], resid=x)
        return ops.layernorm(x, self.fw, self.fb, cfg.ln_eps)

    def hidden_states(self, rows, mask_pads=True):
        """rows int64 [B',S] -> padded [B',S,d] representations (pad slots zero-filled; test helper)"""
        pk = EsmEngine.pack(rows.cpu(), mask_pads)
        h = self.encode_packed(pk, mask_pads)
        Bp, S = rows.shape
        out = torch.zeros(Bp, S, self.cfg.d, dtype=F32, device=self.device)
        out[(torch.arange(S)[None, :] < pk["lens"][:, None]).to(self.device)] = h
        return out

    def forward(self, tokens, pooling="mean", correction=False, mask_pads=True, max_protein_len=1024):
        """`ESM_PLM.forward(tokens, aggregate=True)` in fp32: split long proteins, encode, pool per original protein"""
        rows, keys = batched_split_long_seq(tokens.cpu().long(), max_protein_len=max_protein_len)
        pk = EsmEngine.pack(rows, mask_pads)
        h = self.encode_packed(pk, mask_pads)
        nprot = int(keys.max()) + 1
        seg, rng = [0], []
        for i in range(nprot):
            for r in (keys == i).nonzero(as_tuple=True)[0].tolist():
                rng += [int(pk["cu"][r]), int(pk["real"][r])]
            seg.append(len(rng) // 2)
        mode = {"mean": L.POOL_MEAN_CORRECTED if correction else L.POOL_MEAN, "max": L.POOL_MAX}[pooling]
        seg_t, rng_t = _h2d_many([torch.tensor(seg, dtype=torch.int32), torch.tensor(rng, dtype=torch.int32)], self.device)
        out = torch.empty(nprot, self.cfg.d, dtype=F32, device=self.device)
        L.check(self.ops.lib.pcy_f32_pool(self.ops.h, _p(h), self.cfg.d, _p(seg_t), _p(rng_t), nprot, mode, _p(out)), "pcy_f32_pool")
        return out


class KVCacheF32:
    """token-major fp32 K (roped) / V caches [L, B, Tmax, Hkv*dh]"""

    def __init__(self, cfg: LlamaConfig, B, Tmax, device):
        kw = cfg.n_kv_heads * cfg.head_dim
        self.k = torch.zeros(cfg.n_layers, B, Tmax, kw, dtype=F32, device=device)
        self.v = torch.zeros(cfg.n_layers, B, Tmax, kw, dtype=F32, device=device)
        self.B, self.Tmax = B, Tmax

    def reorder_(self, src_rows, t=None):
        """row b takes the cache of row src_rows[b] (beam search, model_unified.py:830-832) -- in place, only the rows that move and only the
        `t` slots written so far (all of them when t is None): a beam-10 x batch-8 search at 900 slots used to rebuild 2 x 9 GB per step"""
        idx = src_rows.to(self.k.device).long()
        moved = (idx != torch.arange(idx.numel(), device=idx.device)).nonzero(as_tuple=True)[0]
        if moved.numel() == 0:
            return
        t = self.Tmax if t is None else min(int(t), self.Tmax)
        for c in (self.k, self.v):
            c[:, moved, :t] = c[:, idx[moved], :t]       # (advanced indexing on the right gathers into a temporary first: sources are read before any write)


class LlamaEngineF32:
    """Llama decoder in fp32 (HF layout state dict): embedding + splice, L layers, final norm, logits at chosen rows; with a cache also
    the KV-cached decode step (one launch per operator: a compatibility path for callers that generate without .bfloat16(), not a fast one)."""

    def __init__(self, sd, cfg: LlamaConfig, device=None, ctx=None):
        self.ops = F32Ops(ctx, device)
        self.cfg, self.device = cfg, self.ops.device
        g = lambda k: sd[k].to(self.device, F32).contiguous()
        self.embed = g("model.embed_tokens.weight")
        self.final_norm = g("model.norm.weight")
        self.lm_head = g("lm_head.weight")
        self.cos, self.sin = _rope_tables_f32(cfg.head_dim, cfg.rope_theta, cfg.max_pos, self.device)
        self.layers = []
        for l in range(cfg.n_layers):
            p = f"model.layers.{l}."
            self.layers.append(dict(
                wqkv=torch.cat([g(p + f"self_attn.{n}_proj.weight") for n in "qkv"], 0).contiguous(), wo=g(p + "self_attn.o_proj.weight"),
                wg=g(p + "mlp.gate_proj.weight"), wu=g(p + "mlp.up_proj.weight"), wd=g(p + "mlp.down_proj.weight"),
                ln1=g(p + "input_layernorm.weight"), ln2=g(p + "post_attention_layernorm.weight")))

    def embed_tokens(self, ids, soft=None, soft_map=None):
        """[B,T] ids (+ soft tokens [n,d] fp32 and a flat row -> soft index map, -1 = a table row) -> [B,T,d] fp32"""
        B, T = ids.shape
        ids_d, = _h2d_many([ids.reshape(-1).to(torch.int32).cpu()], self.device)
        sm = None
        if soft_map is not None:
            sm, = _h2d_many([soft_map.reshape(-1).to(torch.int32).cpu()], self.device)
            soft = soft.to(self.device, F32).contiguous()
        return self.ops.embed(self.embed, ids_d, soft if soft_map is not None else None, sm).view(B, T, self.cfg.d)

    def new_cache(self, B, Tmax):
        return KVCacheF32(self.cfg, B, Tmax, self.device)

    def decode(self, cache: KVCacheF32, ids, t):
        """one new token per row (ids [B]) at cache length / rotary position t for EVERY row, no mask (reference quirks Q1 / Q2,
        model_unified.py:769, :887) -> logits [B, V] fp32; appends K / V at slot t"""
        ops, cfg = self.ops, self.cfg
        H, Hkv, dh = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim
        B = ids.numel()
        if t + 1 > cache.Tmax:
            raise ValueError(f"KV cache capacity {cache.Tmax} exhausted; raise max_new_tokens")
        ids_d, = _h2d_many([ids.reshape(-1).to(torch.int32).cpu()], self.device)
        x = ops.embed(self.embed, ids_d)
        pos = torch.full((B,), t, dtype=torch.int32, device=self.device)
        qw, kw = H * dh, Hkv * dh
        for l, lw in enumerate(self.layers):
            xn = ops.rmsnorm(x, lw["ln1"], cfg.rms_eps)
            qkv = ops.linear(xn, lw["wqkv"])
            ops.rope(qkv, 0, H, dh, pos, self.cos, self.sin)
            ops.rope(qkv, qw, Hkv, dh, pos, self.cos, self.sin)
            cache.k[l, :, t] = qkv[:, qw:qw + kw]
            cache.v[l, :, t] = qkv[:, qw + kw:]
            ao = ops.attn_decode(qkv, cache.k[l], cache.v[l], t + 1, H, Hkv, dh, dh ** -0.5)
            x = ops.linear(ao, lw["wo"], resid=x)
            xn = ops.rmsnorm(x, lw["ln2"], cfg.rms_eps)
            act = ops.silu_mul(ops.linear(xn, lw["wg"]), ops.linear(xn, lw["wu"]))
            x = ops.linear(act, lw["wd"], resid=x)
        return ops.linear(ops.rmsnorm(x, self.final_norm, cfg.rms_eps), self.lm_head)

    def prefill(self, embeds, attn_mask=None, logit_rows="last", want_hidden=False, sum_rows=None, cache=None):
        """embeds [B,T,d] fp32; attn_mask [B,T] 0/1 or None -> (logits [n,V] fp32, final-normed hidden [B,T,d] | None[, sum over the
        L+1 hidden states at `sum_rows` (flat b*T+t) [n,d]]) -- the return contract of `LlamaEngine.prefill`"""
        ops, cfg = self.ops, self.cfg
        B, T, d = embeds.shape
        if T > cfg.max_pos:
            raise ValueError(f"T={T} exceeds the rotary table ({cfg.max_pos})")
        H, Hkv, dh = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim
        x = embeds.to(self.device, F32).reshape(B * T, d).contiguous().clone()
        pos = torch.arange(T, dtype=torch.int32, device=self.device).repeat(B)
        cu = torch.arange(B + 1, dtype=torch.int32, device=self.device) * T
        keep = None
        if attn_mask is not None and not bool((attn_mask != 0).all()):
            keep = (attn_mask != 0).to(self.device, torch.uint8).reshape(-1).contiguous()
        srows = None if sum_rows is None else sum_rows.to(self.device, torch.int32).contiguous()
        hsum = None
        if srows is not None:
            hsum = torch.zeros(srows.numel(), d, dtype=F32, device=self.device)
        qw, kw = H * dh, Hkv * dh
        for li, lw in enumerate(self.layers):
            if hsum is not None:
                ops.acc_rows(hsum, x, srows)
            xn = ops.rmsnorm(x, lw["ln1"], cfg.rms_eps)
            qkv = ops.linear(xn, lw["wqkv"])
            ops.rope(qkv, 0, H, dh, pos, self.cos, self.sin)
            ops.rope(qkv, qw, Hkv, dh, pos, self.cos, self.sin)
            if cache is not None:
                cache.k[li, :B, :T] = qkv[:, qw:qw + kw].view(B, T, kw)
                cache.v[li, :B, :T] = qkv[:, qw + kw:].view(B, T, kw)
            ao = ops.attention(qkv, 0, qkv, qw, qkv, qw + kw, cu, keep, B, T, H, Hkv, dh, True, dh ** -0.5)
            x = ops.linear(ao, lw["wo"], resid=x)
            xn = ops.rmsnorm(x, lw["ln2"], cfg.rms_eps)
            act = ops.silu_mul(ops.linear(xn, lw["wg"]), ops.linear(xn, lw["wu"]))
            x = ops.linear(act, lw["wd"], resid=x)
        hn = ops.rmsnorm(x, self.final_norm, cfg.rms_eps)
        if hsum is not None:
            ops.acc_rows(hsum, hn, srows)
        if isinstance(logit_rows, str) and logit_rows == "last":
            rows = (torch.arange(B, dtype=torch.int64, device=self.device) + 1) * T - 1
        elif isinstance(logit_rows, str) and logit_rows == "all":
            rows = torch.arange(B * T, dtype=torch.int64, device=self.device)
        elif logit_rows is None:
            rows = torch.zeros(0, dtype=torch.int64, device=self.device)
        else:
            rows = logit_rows.to(self.device, torch.int64)
        logits = torch.empty(rows.numel(), cfg.vocab, dtype=F32, device=self.device)
        if rows.numel():
            ops.linear(hn.index_select(0, rows).contiguous(), self.lm_head, out=logits)
        hidden = hn.view(B, T, d) if want_hidden else None
        if sum_rows is not None:
            return logits, hidden, hsum
        return logits, hidden
