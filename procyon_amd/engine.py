"""Host wrappers over the C ABI (include/pcy.h): weight packing, descriptors, KV cache, packing of
varlen protein batches.  torch is plumbing only -- device memory, streams, index arithmetic; all
model arithmetic runs in libpcy.so's HIP kernels.  No fallback path exists.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from dataclasses import dataclass

import torch

from . import _lib as L

BF16 = torch.bfloat16


def _p(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def _chk_bf16(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == BF16 and t.is_contiguous(), (t.dtype, t.device, t.is_contiguous())


class Context:
    """One engine context per (device, stream).  Launches go to torch's current stream."""
    _cache = {}

    def __init__(self, device=None):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise L.PcyError("no HIP device visible: the ProCyon engine needs an MI355X (no CPU fallback)")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.device = dev
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
        h = C.c_void_p()
        L.check(self.lib.pcy_ctx_create(dev.index or 0, C.c_void_p(stream), C.byref(h)), "pcy_ctx_create")
        self.h = h

    @classmethod
    def get(cls, device=None):
        L.load()
        if not torch.cuda.is_available():
            raise L.PcyError("no HIP device visible: the ProCyon engine needs an MI355X (no CPU fallback)")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        key = (dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)
        if key not in cls._cache:
            cls._cache[key] = Context(dev)
        return cls._cache[key]

    def sync(self):
        L.check(self.lib.pcy_ctx_sync(self.h), "sync")

    def timer_start(self):
        L.check(self.lib.pcy_timer_start(self.h), "timer_start")

    def timer_stop(self):
        ms = C.c_float()
        L.check(self.lib.pcy_timer_stop(self.h, C.byref(ms)), "timer_stop")
        return ms.value

    # ---- primitive ops -------------------------------------------------------------------------
    def gemm(self, A, W, bias=None, resid=None, epi=L.EPI_STORE, out=None):
        _chk_bf16(A, W, bias, resid)
        M, K = A.shape
        N = W.shape[0]
        Nout = N // 2 if epi == L.EPI_SWIGLU else N
        out = torch.empty(M, Nout, dtype=BF16, device=A.device) if out is None else out
        L.check(self.lib.pcy_gemm(self.h, _p(A), K, _p(W), _p(bias), _p(resid), 0 if resid is None else resid.shape[1],
                                  _p(out), Nout, M, N, K, epi), "pcy_gemm")
        return out

    def gemv(self, W, x, bias=None, resid=None, epi=L.EPI_STORE, rms_w=None, rms_eps=1e-5, rms_cast=0, out=None):
        _chk_bf16(W, x, bias, resid, rms_w)
        B, K = x.shape
        N = W.shape[0] // 2 if epi == L.EPI_SWIGLU else W.shape[0]
        out = torch.empty(B, N, dtype=BF16, device=x.device) if out is None else out
        L.check(self.lib.pcy_gemv(self.h, _p(W), _p(x), K, _p(bias), _p(resid), _p(out), N, _p(rms_w), rms_eps, rms_cast,
                                  N, K, B, epi), "pcy_gemv")
        return out

    def decode_mlp(self, x, ln2, wgu, wdown, rms_eps=1e-5, rms_cast=0):
        """One token through a Llama MLP, in place: x[1, d] += down(SwiGLU(gate/up(RMSNorm(x) * ln2))); wgu packed like gemv(EPI_SWIGLU)."""
        _chk_bf16(x, ln2, wgu, wdown)
        d, ffn = x.shape[-1], wdown.shape[1]
        L.check(self.lib.pcy_decode_mlp(self.h, _p(x), _p(ln2), _p(wgu), _p(wdown), d, ffn, rms_eps, rms_cast), "pcy_decode_mlp")
        return x

    def quant_rows_fp8(self, x):
        """Per-row symmetric OCP e4m3 quantisation of a bf16 matrix [rows,K] -> (q uint8 [rows,K], scale fp32 [rows])."""
        _chk_bf16(x)
        rows, K = x.shape
        q = torch.empty(rows, K, dtype=torch.uint8, device=x.device)
        sc = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.check(self.lib.pcy_quant_rows_fp8(self.h, _p(x), x.stride(0), rows, K, _p(q), _p(sc)), "pcy_quant_rows_fp8")
        return q, sc

    def gemm_fp8(self, A8, sa, W8, sw, resid=None, epi=L.EPI_STORE, out=None):
        """epi(((A8 . W8^T) * sa[:,None]) * sw[None,:]) -> bf16; A8 [M,K], W8 [N,K] uint8 (e4m3 bits), sa [M], sw [N] fp32."""
        M, K = A8.shape
        N = W8.shape[0]
        Nout = N // 2 if epi == L.EPI_SWIGLU else N
        out = torch.empty(M, Nout, dtype=BF16, device=A8.device) if out is None else out
        L.check(self.lib.pcy_gemm_fp8(self.h, _p(A8), _p(sa), _p(W8), _p(sw), _p(resid), 0 if resid is None else resid.shape[1],
                                      _p(out), Nout, M, N, K, epi), "pcy_gemm_fp8")
        return out

    def rmsnorm(self, x, w, eps=1e-5, cast=0):
        _chk_bf16(x, w)
        y = torch.empty_like(x)
        L.check(self.lib.pcy_rmsnorm(self.h, _p(x), _p(w), _p(y), x.numel() // x.shape[-1], x.shape[-1], eps, cast), "rmsnorm")
        return y

    def layernorm(self, x, w, b, eps=1e-5):
        _chk_bf16(x, w, b)
        y = torch.empty_like(x)
        L.check(self.lib.pcy_layernorm(self.h, _p(x), _p(w), _p(b), _p(y), x.numel() // x.shape[-1], x.shape[-1], eps), "layernorm")
        return y

    def embed_splice(self, table, ids, soft=None, soft_map=None):
        """ids int32 [rows]; soft_map int32 [rows] (-1 = take table row)."""
        _chk_bf16(table, soft)
        rows, d = ids.numel(), table.shape[1]
        out = torch.empty(rows, d, dtype=BF16, device=table.device)
        L.check(self.lib.pcy_embed_splice(self.h, _p(table), _p(ids), _p(soft), _p(soft_map), _p(out), rows, d), "embed_splice")
        return out

    def rope_(self, buf, col0, nh, dh, pos, cos_t, sin_t, mode=0, prescale=0.0):
        """in place on a token-major [ntok, ld] buffer"""
        _chk_bf16(buf, cos_t, sin_t)
        L.check(self.lib.pcy_rope(self.h, _p(buf), buf.shape[1], col0, nh, dh, _p(pos), _p(cos_t), _p(sin_t), buf.shape[0],
                                  mode, prescale), "pcy_rope")
        return buf

    def attention(self, q, k, v, lens, H, Hkv, dh, causal, scale, keep=None):
        """q [ntok,H*dh], k,v [ntok,Hkv*dh] packed sequences of lengths `lens` (list)."""
        _chk_bf16(q, k, v)
        lens_t = torch.tensor(lens, dtype=torch.int64)
        cu = torch.zeros(len(lens) + 1, dtype=torch.int64)
        cu[1:] = lens_t.cumsum(0)
        vt_cu = torch.zeros(len(lens) + 1, dtype=torch.int64)
        vt_cu[1:] = ((lens_t + 31) // 32 * 32).cumsum(0)
        cu_d, vt_d = cu.to(q.device, torch.int32), vt_cu.to(q.device, torch.int32)
        o = torch.empty(q.shape[0], H * dh, dtype=BF16, device=q.device)
        L.check(self.lib.pcy_attention(self.h, _p(q), q.shape[1], 0, _p(k), k.shape[1], 0, _p(v), v.shape[1], 0, _p(o), H * dh,
                                       _p(cu_d), _p(vt_d), _p(keep), len(lens), int(lens_t.max()), int(vt_cu[-1]), H, Hkv, dh,
                                       int(causal), scale), "pcy_attention")
        return o

    def attn_decode(self, qkv, kcache, vcache, pos, cos_t, sin_t, H, Hkv, dh, keep=None, out=None):
        """qkv [B,(H+2Hkv)*dh]; kcache/vcache [B,Hkv,Tmax,dh]; pos int32 device scalar."""
        B, Tmax = kcache.shape[0], kcache.shape[2]
        o = torch.empty(B, H * dh, dtype=BF16, device=qkv.device) if out is None else out
        L.check(self.lib.pcy_attn_decode(self.h, _p(qkv), qkv.shape[1], _p(kcache), _p(vcache), _p(o), H * dh, _p(pos), _p(cos_t),
                                         _p(sin_t), _p(keep), B, H, Hkv, dh, Tmax), "pcy_attn_decode")
        return o

    def retrieval_scores(self, query, targets):
        """cosine similarities [Q,N] (bf16) as `ProcyonRetrievalEval.get_predictions` forms them (procyon.py:400-406)."""
        _chk_bf16(query, targets)
        Q, D = query.shape
        N = targets.shape[0]
        out = torch.empty(Q, N, dtype=BF16, device=query.device)
        L.check(self.lib.pcy_retrieval_scores(self.h, _p(query), Q, _p(targets), N, D, _p(out)), "pcy_retrieval_scores")
        return out

    def retrieval_topk(self, query, targets, k=None):
        """(indices int64 [Q,k], scores bf16 [Q,k]) of the k most similar targets per query, best first, ties by lower index
        (`get_proteins_from_embedding`, data/inference_utils.py:921-978); k=None ranks all N targets."""
        _chk_bf16(query, targets)
        Q, D = query.shape
        N = targets.shape[0]
        k = N if k is None else min(int(k), N)
        idx = torch.empty(Q, k, dtype=torch.int32, device=query.device)
        sc = torch.empty(Q, k, dtype=BF16, device=query.device)
        L.check(self.lib.pcy_retrieval_topk(self.h, _p(query), Q, _p(targets), N, D, k, _p(idx), _p(sc)), "pcy_retrieval_topk")
        return idx.long(), sc

    def qa_probs(self, logits, yes_id=None, no_id=None, want_probs=True, want_argmax=False):
        """The QA read-out on the device (pcy_qa_probs): `logits.softmax(dim=-1)` over the vocabulary for answer rows [rows, V] in the rows'
        dtype (bf16: fp32 statistics, one rounding; fp32), optionally the yes / no columns [rows, 2] (fp32 values of the stored
        probabilities) and the argmax of the stored probabilities -- data/inference_utils.py:582-604, training/train_utils.py:1048-1070.
        -> (probs | None, yes_no | None, argmax int64 | None)"""
        x = logits.to(self.device).contiguous()
        assert x.dim() == 2 and x.dtype in (BF16, torch.float32), (x.shape, x.dtype)
        rows, V = x.shape
        probs = torch.empty_like(x) if want_probs else None
        yn = torch.empty(rows, 2, dtype=torch.float32, device=x.device) if yes_id is not None else None
        am = torch.empty(rows, dtype=torch.int32, device=x.device) if want_argmax else None
        L.check(self.lib.pcy_qa_probs(self.h, _p(x), int(x.dtype == torch.float32), rows, V, int(yes_id or 0), int(no_id or 0), _p(probs), _p(yn),
                                      _p(am)), "pcy_qa_probs")
        return probs, yn, None if am is None else am.long()

    def _ret_f32_args(self, query, targets):
        q = query.to(self.device, torch.float32).contiguous()
        t = targets.to(self.device)
        if t.dtype != BF16:
            t = t.to(torch.float32)
        t = t.contiguous()
        assert q.dim() == 2 and t.dim() == 2 and q.shape[1] == t.shape[1], (q.shape, t.shape)
        return q, t, int(t.dtype == BF16)

    def retrieval_scores_f32(self, query, targets):
        """fp32 cosine similarities [Q,N] (fp32 accumulation, fp32 result); query any float dtype, targets fp32 or bf16 -- the
        scoring of `get_proteins_from_batched_embeddings` (data/inference_utils.py:981-999)."""
        q, t, tb = self._ret_f32_args(query, targets)
        out = torch.empty(q.shape[0], t.shape[0], dtype=torch.float32, device=self.device)
        L.check(self.lib.pcy_retrieval_scores_f32(self.h, _p(q), q.shape[0], _p(t), tb, t.shape[0], q.shape[1], _p(out)), "pcy_retrieval_scores_f32")
        return out

    def retrieval_topk_f32(self, query, targets, k=None):
        """(indices int64 [Q,k], scores fp32 [Q,k]) ranked on fp32 similarities, best first, ties by lower index
        (`get_proteins_from_embedding`, data/inference_utils.py:921-978); k=None ranks all N targets."""
        q, t, tb = self._ret_f32_args(query, targets)
        N = t.shape[0]
        k = N if k is None else min(int(k), N)
        idx = torch.empty(q.shape[0], k, dtype=torch.int32, device=self.device)
        sc = torch.empty(q.shape[0], k, dtype=torch.float32, device=self.device)
        L.check(self.lib.pcy_retrieval_topk_f32(self.h, _p(q), q.shape[0], _p(t), tb, N, q.shape[1], k, _p(idx), _p(sc)), "pcy_retrieval_topk_f32")
        return idx.long(), sc

    def pool(self, hidden, seg, rng, nprot, mode):
        _chk_bf16(hidden)
        d = hidden.shape[-1]
        out = torch.empty(nprot, d, dtype=BF16, device=hidden.device)
        L.check(self.lib.pcy_pool(self.h, _p(hidden), d, _p(seg), _p(rng), nprot, mode, _p(out)), "pool")
        return out


# ------------------------------------------------------------------------------------------------
class MlpEngine:
    """`create_mlp` projector (reference procyon/model/model_utils.py:13-41), eval mode."""

    def __init__(self, layers, ctx=None):
        self.ctx = ctx or Context.get()
        self.layers = [(w.contiguous(), None if b is None else b.contiguous()) for w, b in layers]
        n = len(self.layers)
        assert 1 <= n <= 8
        d = L.MlpDesc()
        d.n_layers = n
        d.dims[0] = self.layers[0][0].shape[1]
        for i, (w, b) in enumerate(self.layers):
            _chk_bf16(w, b)
            d.dims[i + 1] = w.shape[0]
            d.w[i] = w.data_ptr()
            d.b[i] = 0 if b is None else b.data_ptr()
        self.desc = d
        self.in_features, self.out_features = d.dims[0], d.dims[n]
        self.src_f32 = None      # fp32 (W, b) pairs of the same projector, set by the loader when the checkpoint is fp32
        self._f32 = None

    def f32(self):
        """the same projector in fp32 arithmetic (procyon_amd.engine_f32), for a model that was never cast to bf16"""
        if self._f32 is None:
            if self.src_f32 is None:
                raise RuntimeError("fp32 arithmetic was asked for, but this projector holds no fp32 weights")
            from .engine_f32 import MlpEngineF32
            self._f32 = MlpEngineF32(self.src_f32, self.ctx)
        return self._f32

    def drop_fp32(self):
        self.src_f32 = self._f32 = None

    def __call__(self, x):
        if x.dtype == torch.float32:
            return self.f32()(x)
        shape = x.shape
        x = x.reshape(-1, shape[-1]).contiguous()
        _chk_bf16(x)
        out = torch.empty(x.shape[0], self.out_features, dtype=BF16, device=x.device)
        if x.shape[0]:
            L.check(self.ctx.lib.pcy_mlp_forward(self.ctx.h, C.byref(self.desc), _p(x), x.shape[0], _p(out)), "pcy_mlp_forward")
        return out.reshape(*shape[:-1], self.out_features)


# ------------------------------------------------------------------------------------------------
def rope_tables(dh, theta, n_pos, device, inv_freq_bf16=False):
    """cos/sin [n_pos, dh] bf16.  Default: fp32 inv_freq, table rounded once (transformers 4.31's cached
    tables).  inv_freq_bf16=True reproduces transformers 5.x after `.bfloat16()`."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, dh, 2, dtype=torch.float32) / dh))
    if inv_freq_bf16:
        inv_freq = inv_freq.to(BF16).float()
    pos = torch.arange(n_pos, dtype=torch.float32)
    freqs = (inv_freq[:, None] @ pos[None, :]).transpose(0, 1)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(BF16).to(device).contiguous(), emb.sin().to(BF16).to(device).contiguous()


@dataclass
class LlamaConfig:
    vocab: int
    d: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    ffn: int
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0   # reference-compat default (App. B Q9); Llama-3's own value is 500000
    max_pos: int = 8192
    rms_cast: str = "hf5"         # 'hf5' (>=4.32) or 'hf431'
    rope_inv_freq_bf16: bool = False

    @property
    def head_dim(self):
        return self.d // self.n_heads


def interleave_gate_up(gate, up):
    """[F,d],[F,d] -> [2F,d] with 16-row gate/up interleave (layout the SwiGLU epilogues expect)."""
    F_, d = gate.shape
    assert F_ % 16 == 0
    return torch.stack([gate.view(F_ // 16, 16, d), up.view(F_ // 16, 16, d)], dim=1).reshape(2 * F_, d).contiguous()


class _Stager:
    """Host -> device upload of the small int32 index arrays of one engine call without stalling the stream and without a
    pinned allocation per call: a ring of persistent pinned staging buffers (grown on demand), ONE async copy per call into
    one device buffer, views handed out (a pageable source would make the copy wait for the queued GPU work, and the
    retrieval loop would alternate host packing and GPU encoding instead of overlapping them).  Measured equal to a
    `pin_memory()` copy per array (tools/archive/t_esm_percall.py: 22.1 ms per call at batch 8, 56.4 ms at batch 25, back to back);
    it replaces six pinned allocations and copies per call by one."""
    SLOTS = 3

    def __init__(self, device):
        self.device = torch.device(device)
        self.slots = [None] * self.SLOTS
        self.events = [None] * self.SLOTS
        self.i = 0

    def upload(self, arrays, into=None):
        """arrays: CPU int32 tensors -> list of device int32 tensors (views of one device buffer, 256-byte aligned).
        `into`: a persistent device uint8 buffer to copy into instead of a fresh one (same sizes -> the views keep their addresses
        from call to call: what a replayed launch chain needs)."""
        offs, total = [], 0
        for a in arrays:
            offs.append(total)
            total += (a.numel() * 4 + 255) // 256 * 256
        total = max(total, 256)
        k = self.i % len(self.slots)
        if self.events[k] is not None and not self.events[k].query():
            # the copy from this slot is still queued behind earlier kernels: never wait for the stream here (the host would
            # serialise with the GPU); take a fresh slot instead -- the ring grows to the number of calls in flight
            self.slots.insert(k, None)
            self.events.insert(k, None)
        if self.slots[k] is None or self.slots[k].numel() < total:
            self.slots[k] = torch.empty(max(total, 1 << 20) * 2, dtype=torch.uint8).pin_memory()
        host = self.slots[k]
        for a, o in zip(arrays, offs):
            n = a.numel() * 4
            host[o:o + n].view(torch.int32).copy_(a.contiguous().view(-1))
        devbuf = torch.empty(total, dtype=torch.uint8, device=self.device) if into is None else into[:total]
        devbuf.copy_(host[:total], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.events[k] = ev
        self.i = k + 1
        return [devbuf[o:o + a.numel() * 4].view(torch.int32).view(a.shape) for a, o in zip(arrays, offs)]


_stagers = {}


def _h2d_many(arrays, dev, into=None):
    dev = torch.device(dev)
    if dev.type != "cuda":
        return [a.to(dev) for a in arrays]
    if into is None and os.environ.get("PCY_STAGER", "1") == "0":      # A/B: a pinned copy per array and call (the previous behaviour)
        outs = []
        for a in arrays:
            src = a.contiguous().pin_memory()
            o = src.to(dev, non_blocking=True)
            o._pcy_src = src
            outs.append(o)
        return outs
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _stagers:
        _stagers[key] = _Stager(dev)
    return _stagers[key].upload(arrays, into)


class KVCache:
    """[L,B,Hkv,Tmax,dh] x2; `layer(l)` gives the reference's past_key_values[l] views."""

    def __init__(self, cfg: LlamaConfig, B, Tmax, device):
        self.k = torch.zeros(cfg.n_layers, B, cfg.n_kv_heads, Tmax, cfg.head_dim, dtype=BF16, device=device)
        self.v = torch.zeros_like(self.k)
        self.B, self.Tmax = B, Tmax
        self.c = L.KvCache(self.k.data_ptr(), self.v.data_ptr(), B, Tmax)

    def layer(self, l, t):
        return self.k[l, :, :, :t], self.v[l, :, :, :t]


class GenState:
    def __init__(self, B, vocab, max_steps, device, keep_logits=False, keep=None):
        self.pos = torch.zeros(1, dtype=torch.int32, device=device)
        self.step = torch.zeros(1, dtype=torch.int32, device=device)
        self.next_tok = torch.zeros(B, dtype=torch.int32, device=device)
        self.tokens_out = torch.zeros(B, max_steps, dtype=torch.int32, device=device)
        self.logprob = torch.zeros(B, dtype=torch.float32, device=device)
        self.logits = torch.empty(B, vocab, dtype=BF16, device=device)
        # the per-step logits record is what the reference returns on the CPU: pinned host memory written by the decode steps
        # themselves (rows padded to a multiple of 8 elements for 16-byte stores); PCY_LOGITS_HOST=0 keeps it on the device
        self.logits_all = self._logits_host = None
        ld = vocab
        if keep_logits:
            if os.environ.get("PCY_LOGITS_HOST", "1") != "0":
                ld = (vocab + 7) // 8 * 8
                self._logits_host = torch.empty(max_steps, B, ld, dtype=BF16, pin_memory=True)
                self.logits_all = self._logits_host[:, :, :vocab]
            else:
                self.logits_all = torch.empty(max_steps, B, vocab, dtype=BF16, device=device)
        self.keep = keep
        self.c = L.GenState(self.pos.data_ptr(), self.step.data_ptr(), self.next_tok.data_ptr(), self.tokens_out.data_ptr(),
                            self.logprob.data_ptr(), self.logits.data_ptr(),
                            0 if self.logits_all is None else self.logits_all.data_ptr(),
                            0 if keep is None else keep.data_ptr(), max_steps, ld)


class BeamState:
    """Device-side state of the diverse beam search (pcy_beam_step): token histories (double-buffered), running scores,
    parent slots per step, EOS flags, step / position / done scalars."""

    def __init__(self, B, beam, max_len, eos_id, prompt_len, device):
        BB = B * beam
        z = lambda *shape, dtype=torch.int32: torch.zeros(*shape, dtype=dtype, device=device)
        self.B, self.beam, self.BB, self.max_len = B, beam, BB, max_len
        self.out = z(2, BB, max_len)
        self.cur, self.cur_new = z(BB, dtype=torch.float32), z(BB, dtype=torch.float32)
        self.next_tok, self.src, self.anc = z(BB), z(BB), z(max_len, BB)
        self.has_eos = z(2, BB, dtype=torch.uint8)
        self.blk_eos, self.ticket = z(B), z(1)
        self.pos = torch.full((1,), prompt_len, dtype=torch.int32, device=device)
        self.step, self.done = z(1), z(1)
        self.c = L.BeamState(self.out.data_ptr(), max_len, self.cur.data_ptr(), self.cur_new.data_ptr(), self.next_tok.data_ptr(),
                             self.src.data_ptr(), self.anc.data_ptr(), self.has_eos.data_ptr(), self.blk_eos.data_ptr(),
                             self.ticket.data_ptr(), self.pos.data_ptr(), self.step.data_ptr(), self.done.data_ptr(), int(eos_id))

    def tokens(self):
        """[BB, steps] int64 histories after the steps run so far (syncs)."""
        steps = int(self.step)
        return self.out[steps & 1, :, :steps].long(), steps


class LlamaEngine:
    """HF-Llama-architecture decoder behind `LlamaPostTokenization.forward` (pmc_llama.py:546-596)."""

    def __init__(self, sd, cfg: LlamaConfig, device=None, ctx=None, free_source=False, fp8_prefill=False):
        """fp8_prefill: additionally keep per-output-channel e4m3 copies of the four projections of every layer and run
        the prefill's projections on the fp8 MFMA path (BASELINE configs[4]; `set_fp8(False)` switches back to bf16)."""
        self.ctx = ctx or Context.get(device)
        self.cfg = cfg
        dev = self.ctx.device
        self.device = dev
        g = lambda k: sd[k].to(dev, BF16).contiguous()
        self.embed = g("model.embed_tokens.weight")
        self.final_norm = g("model.norm.weight")
        self.lm_head = g("lm_head.weight")
        self.cos, self.sin = rope_tables(cfg.head_dim, cfg.rope_theta, cfg.max_pos, dev, cfg.rope_inv_freq_bf16)
        self._keep = []
        arr = (L.LlamaLayer * cfg.n_layers)()
        for l in range(cfg.n_layers):
            p = f"model.layers.{l}."
            wqkv = torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"),
                              g(p + "self_attn.v_proj.weight")], 0).contiguous()
            wo = g(p + "self_attn.o_proj.weight")
            wgu = interleave_gate_up(g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight"))
            wdown = g(p + "mlp.down_proj.weight")
            ln1, ln2 = g(p + "input_layernorm.weight"), g(p + "post_attention_layernorm.weight")
            self._keep.append((wqkv, wo, wgu, wdown, ln1, ln2))
            arr[l] = L.LlamaLayer(*[t.data_ptr() for t in (wqkv, wo, wgu, wdown, ln1, ln2)])
            if free_source:
                for k in [k for k in sd if k.startswith(p)]:
                    del sd[k]
        self._arr = arr
        self.desc = L.LlamaDesc(cfg.vocab, cfg.d, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.ffn,
                                cfg.max_pos, cfg.rms_eps, 0 if cfg.rms_cast == "hf5" else 1, self.embed.data_ptr(),
                                self.final_norm.data_ptr(), self.lm_head.data_ptr(), self.cos.data_ptr(), self.sin.data_ptr(),
                                C.cast(arr, C.POINTER(L.LlamaLayer)))
        self._arr8 = None
        if fp8_prefill:
            self.quantize_fp8()

    def quantize_fp8(self):
        """e4m3 copies + per-row scales of wqkv / wo / wgu / wdown (8 GB on top of the 16 GB of a Llama-3-8B), fp8 path on."""
        if self._arr8 is None:
            self._keep8 = []
            arr8 = (L.LlamaLayerFp8 * self.cfg.n_layers)()
            for l, ws in enumerate(self._keep):
                qs = [self.ctx.quant_rows_fp8(w) for w in ws[:4]]
                self._keep8.append(qs)
                arr8[l] = L.LlamaLayerFp8(*[q.data_ptr() for q, _ in qs], *[sc.data_ptr() for _, sc in qs])
            self._arr8 = arr8
        self.set_fp8(True)

    def set_fp8(self, on):
        if on and self._arr8 is None:
            raise ValueError("no fp8 weights: construct with fp8_prefill=True or call quantize_fp8()")
        self.desc.layers_fp8 = C.cast(self._arr8, C.POINTER(L.LlamaLayerFp8)) if on else C.POINTER(L.LlamaLayerFp8)()

    def new_cache(self, B, Tmax):
        return KVCache(self.cfg, B, Tmax, self.device)

    def embed_tokens(self, ids, soft=None, soft_map=None):
        """ids [B,T] int -> [B,T,d]; soft-token splice of `_prepare_input_embeddings` if soft_map given."""
        B, T = ids.shape
        i32 = ids.to(self.device, torch.int32).contiguous().view(-1)
        sm = None if soft_map is None else soft_map.to(self.device, torch.int32).contiguous().view(-1)
        return self.ctx.embed_splice(self.embed, i32, soft, sm).view(B, T, self.cfg.d)

    def prefill(self, embeds, attn_mask, cache: KVCache, logit_rows="last", want_hidden=False, sum_rows=None):
        """embeds [B,T,d] bf16; attn_mask [B,T] (0/1) or None.  Returns (logits [n,V], hidden [B,T,d]|None), and with
        `sum_rows` (flat token rows b*T+t) additionally the sum over all L+1 hidden states of those rows [n,d]
        (ret_token_access='all')."""
        B, T, d = embeds.shape
        embeds = embeds.contiguous()
        _chk_bf16(embeds)
        dev = self.device
        keep = None
        if attn_mask is not None and not bool((attn_mask != 0).all()):
            keep = (attn_mask != 0).to(dev, torch.uint8).contiguous()
        pos = torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
        Tp = (T + 31) // 32 * 32
        cu = torch.arange(B + 1, dtype=torch.int32, device=dev) * T
        vt_cu = torch.arange(B + 1, dtype=torch.int32, device=dev) * Tp
        if isinstance(logit_rows, str) and logit_rows == "last":
            rows = (torch.arange(B, dtype=torch.int32, device=dev) + 1) * T - 1
        elif logit_rows is None:
            rows = torch.zeros(0, dtype=torch.int32, device=dev)
        else:
            rows = logit_rows.to(dev, torch.int32).contiguous()
        n = rows.numel()
        logits = torch.empty(n, self.cfg.vocab, dtype=BF16, device=dev)
        hidden = torch.empty(B, T, d, dtype=BF16, device=dev) if want_hidden else None
        srows = None if sum_rows is None else sum_rows.to(dev, torch.int32).contiguous()
        ns = 0 if srows is None else srows.numel()
        hsum = torch.empty(ns, d, dtype=BF16, device=dev) if ns else None
        L.check(self.ctx.lib.pcy_llama_prefill(self.ctx.h, C.byref(self.desc), C.byref(cache.c), _p(embeds), _p(keep), _p(pos),
                                               _p(cu), _p(vt_cu), B, T, _p(rows), n, _p(logits), _p(hidden), _p(srows), ns, _p(hsum)),
                "pcy_llama_prefill")
        if sum_rows is not None:
            return logits, hidden, hsum
        return logits, hidden

    def prefill_all(self, embeds, attn_mask, cache: KVCache, logit_rows="all"):
        """The prefill with everything `LlamaPostTokenization.forward` returns (pmc_llama.py:575-596): (logits [n,V] for
        `logit_rows` ("all" = every token row, the reference's [B,T,V]; "last"; None; or flat rows), hidden_states [L+1,B,T,d] =
        embeddings, outputs of layers 0..L-2, final-normed output of layer L-1)."""
        B, T, d = embeds.shape
        embeds = embeds.contiguous()
        _chk_bf16(embeds)
        dev = self.device
        keep = None
        if attn_mask is not None and not bool((attn_mask != 0).all()):
            keep = (attn_mask != 0).to(dev, torch.uint8).contiguous()
        pos = torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
        Tp = (T + 31) // 32 * 32
        cu = torch.arange(B + 1, dtype=torch.int32, device=dev) * T
        vt_cu = torch.arange(B + 1, dtype=torch.int32, device=dev) * Tp
        if isinstance(logit_rows, str) and logit_rows == "all":
            rows = torch.arange(B * T, dtype=torch.int32, device=dev)
        elif isinstance(logit_rows, str) and logit_rows == "last":
            rows = (torch.arange(B, dtype=torch.int32, device=dev) + 1) * T - 1
        elif logit_rows is None:
            rows = torch.zeros(0, dtype=torch.int32, device=dev)
        else:
            rows = logit_rows.to(dev, torch.int32).contiguous()
        n = rows.numel()
        logits = torch.empty(n, self.cfg.vocab, dtype=BF16, device=dev)
        hidden_all = torch.empty(self.cfg.n_layers + 1, B, T, d, dtype=BF16, device=dev)
        L.check(self.ctx.lib.pcy_llama_prefill_all(self.ctx.h, C.byref(self.desc), C.byref(cache.c), _p(embeds), _p(keep), _p(pos),
                                                   _p(cu), _p(vt_cu), B, T, _p(rows), n, _p(logits), _p(hidden_all)),
                "pcy_llama_prefill_all")
        return logits, hidden_all

    def decode(self, cache: KVCache, st: GenState, B):
        L.check(self.ctx.lib.pcy_llama_decode(self.ctx.h, C.byref(self.desc), C.byref(cache.c), C.byref(st.c), B), "pcy_llama_decode")

    def decode_layers(self, cache: KVCache, st: GenState, B, reps=1):
        """measurement aid: `reps` passes over the decoder layers of a decode step (no embedding, no lm_head, no pick)."""
        L.check(self.ctx.lib.pcy_llama_decode_layers(self.ctx.h, C.byref(self.desc), C.byref(cache.c), C.byref(st.c), B, reps), "pcy_llama_decode_layers")

    def decode_graph(self, cache: KVCache, st: GenState, B):
        """decode step as one replayed hipGraph (~190 launches otherwise): loops that select between steps on the device."""
        L.check(self.ctx.lib.pcy_llama_decode_graph(self.ctx.h, C.byref(self.desc), C.byref(cache.c), C.byref(st.c), B), "pcy_llama_decode_graph")

    def beam_step(self, logits, bs: "BeamState", group_size, diversity_penalty):
        """one step of the reference's diverse-beam bookkeeping on the device (pcy_beam_step); logits [BB, vocab] bf16."""
        _chk_bf16(logits)
        L.check(self.ctx.lib.pcy_beam_step(self.ctx.h, _p(logits), logits.shape[1], bs.B, bs.beam, group_size, float(diversity_penalty),
                                           C.byref(bs.c)), "pcy_beam_step")

    def beam_steps(self, cache: KVCache, st: GenState, bs: "BeamState", group_size, diversity_penalty, logits_rec, n_steps, kv_t0=0):
        """n_steps iterations of decode -> record -> beam step -> KV reorder, each ONE replayed launch chain
        (pcy_llama_beam_steps); st.pos / st.next_tok are the beam state's arrays.  kv_t0: first cache slot the reorder moves (the prompt
        length when the beams of a prompt hold the same prefix rows)."""
        L.check(self.ctx.lib.pcy_llama_beam_steps(self.ctx.h, C.byref(self.desc), C.byref(cache.c), C.byref(st.c), bs.B, bs.beam, group_size,
                                                  float(diversity_penalty), C.byref(bs.c), _p(logits_rec), n_steps, int(kv_t0)), "pcy_llama_beam_steps")

    def pick(self, cache: KVCache, st: GenState, B, advance_pos):
        L.check(self.ctx.lib.pcy_greedy_pick(self.ctx.h, C.byref(self.desc), C.byref(cache.c), C.byref(st.c), B, int(advance_pos)),
                "pcy_greedy_pick")

    def greedy_steps(self, cache, st, B, n_steps, use_graph=True):
        L.check(self.ctx.lib.pcy_llama_greedy(self.ctx.h, C.byref(self.desc), C.byref(cache.c), C.byref(st.c), B, n_steps,
                                              int(use_graph)), "pcy_llama_greedy")

    def sample_pick(self, cache, st, B, advance_pos, uniforms, temperature=1.0, nucleus_prob=None, probs_out=None):
        """sampling / nucleus selection on st.logits with the uniform variates uniforms[step * B + b] (pcy_sample_pick)"""
        L.check(self.ctx.lib.pcy_sample_pick(self.ctx.h, C.byref(self.desc), C.byref(cache.c), C.byref(st.c), B, int(advance_pos),
                                             float(temperature), -1.0 if nucleus_prob is None else float(nucleus_prob), _p(uniforms), _p(probs_out)),
                "pcy_sample_pick")

    def sample_steps(self, cache, st, B, n_steps, uniforms, temperature=1.0, nucleus_prob=None):
        L.check(self.ctx.lib.pcy_llama_sample(self.ctx.h, C.byref(self.desc), C.byref(cache.c), C.byref(st.c), B, n_steps, float(temperature),
                                              -1.0 if nucleus_prob is None else float(nucleus_prob), _p(uniforms)), "pcy_llama_sample")

    def generate_sampling(self, embeds, attn_mask, max_len, temperature=1.0, nucleus_prob=None, keep_logits=False, uniforms=None):
        """`_generate_sampling(greedy=False)` (model_unified.py:861-921) entirely on the device: prefill, then per step the
        decode launches + the sampling kernels (pcy_llama_sample); the uniform variates of all steps are drawn up front from
        torch's device generator (one `torch.rand`).  Returns (tokens [B,max_len] int64, logprob [B], logits | None, state)."""
        B, T, _ = embeds.shape
        cache = self.new_cache(B, T + max_len)
        st = GenState(B, self.cfg.vocab, max_len, self.device, keep_logits, None)
        u = torch.rand(max_len * B, device=self.device, dtype=torch.float32) if uniforms is None else uniforms.to(self.device, torch.float32).contiguous()
        logits, _ = self.prefill(embeds, attn_mask, cache, "last")
        st.logits.copy_(logits)
        st.pos.fill_(T)
        self.sample_pick(cache, st, B, False, u, temperature, nucleus_prob)
        if max_len > 1:
            self.sample_steps(cache, st, B, max_len - 1, u, temperature, nucleus_prob)
        self.ctx.sync()       # the record may live in host memory; and a watchdog of the fused decode launches must fail THIS call
        la = None if st.logits_all is None else st.logits_all.transpose(0, 1)
        return st.tokens_out.long(), st.logprob, la, (st, cache, u)

    def kv_reorder(self, cache, src_rows, t, t0=0):
        """cache[:, b] = cache[:, src_rows[b]] over slots [t0, t) (t0 > 0: the rows are known to hold the same contents below t0)"""
        src = src_rows.to(self.device, torch.int32).contiguous()
        L.check(self.ctx.lib.pcy_kv_reorder_range(self.ctx.h, C.byref(self.desc), C.byref(cache.c), _p(src), src.numel(), int(t0), t), "pcy_kv_reorder")

    def generate_greedy(self, embeds, attn_mask, max_len, keep_logits=False, clean_decode_mask=False, use_graph=True):
        """`_generate_sampling(greedy=True)` (model_unified.py:861-921): prefill, then max_len-1 cached decode
        steps with no mask and position = cache length (Q1/Q2); no EOS stop.  Everything stays on the device;
        returns (tokens [B,max_len] int64, logprob [B] fp32, logits [B,max_len,V] bf16 | None, state)."""
        B, T, _ = embeds.shape
        cache = self.new_cache(B, T + max_len)
        keep = None
        if clean_decode_mask and attn_mask is not None:
            keep = torch.ones(B, T + max_len, dtype=torch.uint8, device=self.device)
            keep[:, :T] = (attn_mask != 0).to(self.device, torch.uint8)
        st = GenState(B, self.cfg.vocab, max_len, self.device, keep_logits, keep)
        logits, _ = self.prefill(embeds, attn_mask, cache, "last")
        st.logits.copy_(logits)
        st.pos.fill_(T)
        self.pick(cache, st, B, advance_pos=False)
        if max_len > 1:
            self.greedy_steps(cache, st, B, max_len - 1, use_graph)
        # complete before anything is handed out: the logits record may live in host memory, and a watchdog that fired inside a
        # fused decode launch (workgroups not all resident) must fail THIS call, not a later one (pcy_ctx_sync reads the sticky word)
        self.ctx.sync()
        la = None if st.logits_all is None else st.logits_all.transpose(0, 1)
        return st.tokens_out.long(), st.logprob, la, (st, cache)


# ------------------------------------------------------------------------------------------------
@dataclass
class EsmConfig:
    d: int
    n_layers: int
    n_heads: int
    ffn: int
    vocab: int = 33
    ln_eps: float = 1e-5
    rope_theta: float = 10000.0
    rope_math: str = "fp32_once"   # HF Esm; 'model_dtype' = fair-esm's three roundings
    rope_inv_freq_bf16: bool = False
    max_len: int = 1026

    @property
    def head_dim(self):
        return self.d // self.n_heads


PAD_ID, CLS_ID, EOS_ID, MASK_ID = 1, 0, 2, 32


def batched_split_long_seq(toks, padding_idx=1, eos_idx=2, max_protein_len=1024):
    """Host mirror of `batched_split_long_seq(..., "split")` (procyon/training/train_utils.py:1497-1571),
    non-mutating.  toks int64 CPU [B,W] -> (rows [B',max_protein_len+2], keys [B'])."""
    toks = toks.clone()
    W = toks.shape[1]
    cls_idx = int(toks[0, 0])
    add, keys = [], list(range(toks.shape[0]))
    for i in range(toks.shape[0]):
        e = int((toks[i] == eos_idx).nonzero(as_tuple=True)[0][0])
        if e <= max_protein_len + 1:
            continue
        n_add = e // (max_protein_len + 1)
        for j in range(n_add):
            bot = (j + 1) * max_protein_len + 1
            row = torch.full((1, W), padding_idx, dtype=torch.int64)
            tail = toks[i, bot:]
            row[0, 1:tail.shape[0] + 1] = tail
            row[0, 0] = cls_idx
            if j < n_add - 1:
                row[0, max_protein_len + 1] = eos_idx
                row[0, max_protein_len + 2:] = 1
            add.append(row)
            keys.append(i)
        toks[i, max_protein_len + 2:] = padding_idx
        toks[i, max_protein_len + 1] = eos_idx
    new = torch.cat([toks] + add, dim=0)[:, : max_protein_len + 2]
    return new, torch.tensor(keys, dtype=torch.int64)


class EsmEngine:
    """ESM2 encoder + ProteinPooler behind `ESM_PLM.forward` (procyon/model/esm.py:504-558)."""

    def __init__(self, sd, cfg: EsmConfig, device=None, ctx=None):
        self.ctx = ctx or Context.get(device)
        self.cfg = cfg
        dev = self.ctx.device
        self.device = dev
        g = lambda k: sd[k].to(dev, BF16).contiguous()
        self.embed = g("esm.embeddings.word_embeddings.weight")
        self.fw, self.fb = g("esm.encoder.emb_layer_norm_after.weight"), g("esm.encoder.emb_layer_norm_after.bias")
        self.cos, self.sin = rope_tables(cfg.head_dim, cfg.rope_theta, cfg.max_len, dev, cfg.rope_inv_freq_bf16)
        self._keep = []
        self._enc_slots = {}
        arr = (L.EsmLayer * cfg.n_layers)()
        for l in range(cfg.n_layers):
            p = f"esm.encoder.layer.{l}."
            wqkv = torch.cat([g(p + f"attention.self.{n}.weight") for n in ("query", "key", "value")], 0).contiguous()
            bqkv = torch.cat([g(p + f"attention.self.{n}.bias") for n in ("query", "key", "value")], 0).contiguous()
            ts = (wqkv, bqkv, g(p + "attention.output.dense.weight"), g(p + "attention.output.dense.bias"),
                  g(p + "attention.LayerNorm.weight"), g(p + "attention.LayerNorm.bias"),
                  g(p + "intermediate.dense.weight"), g(p + "intermediate.dense.bias"),
                  g(p + "output.dense.weight"), g(p + "output.dense.bias"), g(p + "LayerNorm.weight"), g(p + "LayerNorm.bias"))
            self._keep.append(ts)
            arr[l] = L.EsmLayer(*[t.data_ptr() for t in ts])
        self._arr = arr
        # masked-LM head (fair-esm RobertaLMHead / HF EsmLMHead: dense -> gelu -> layer_norm -> decoder tied to the embedding + bias),
        # only when the checkpoint carries it; consumed by ESM_PLM.forward(aggregate=False) (esm.py:547-558)
        self.lm_head = None
        if "lm_head.dense.weight" in sd:
            dec = sd["lm_head.decoder.weight"] if "lm_head.decoder.weight" in sd else sd["esm.embeddings.word_embeddings.weight"]
            self.lm_head = dict(wd=g("lm_head.dense.weight"), bd=g("lm_head.dense.bias"), lw=g("lm_head.layer_norm.weight"),
                                lb=g("lm_head.layer_norm.bias"), wo=dec.to(dev, BF16).contiguous(), bo=g("lm_head.bias"))
        self.desc = L.EsmDesc(cfg.d, cfg.n_layers, cfg.n_heads, cfg.ffn, cfg.vocab, cfg.ln_eps,
                              1 if cfg.rope_math == "fp32_once" else 0, self.embed.data_ptr(), self.fw.data_ptr(),
                              self.fb.data_ptr(), self.cos.data_ptr(), self.sin.data_ptr(), C.cast(arr, C.POINTER(L.EsmLayer)))

    # = the engine's own bound (pcy_esm_encode; PCY_DISABLE=esm_graph switches the replay off).  Round 5: raised from 4200 (one long protein) to
    # 40 000 tokens (the retrieval batch: 25 x 1026).  A bulk batch is ~230 launches of 25-300 us; on an idle host they are enqueued in 2.4 ms,
    # far ahead of the GPU, but a GPU box whose host cores were busy (other tenants) took ~50 ms per batch for them and the retrieval leg
    # dropped from 634 to 434 proteins/s with the encoder's own time unchanged (profiles/archive/r05_bench.json, history in DESIGN.md): one graph
    # launch per batch takes the host out of the loop.
    GRAPH_MAX_TOKENS = 40000

    def preferred_batch(self, tokens_per_protein, lo=16, hi=40, n_cu=256):
        """Proteins per engine call for retrieval-style bulk encoding ("batch size chosen by the engine", BASELINE configs[2]).
        The four GEMMs of a layer run in rounds of 256 (256 x 256 tiles) or 512 (128 x 128: the o projection) workgroups, so
        throughput follows how well B x S tokens fill the last round of each; beyond ~40 proteins the activations outgrow the
        256 MiB Infinity Cache and throughput declines.  Score = flop-weighted tile-round efficiency - 0.0015 B, fitted on
        ESM2-650M at S = 1026: batch 25 -> 455 proteins/s, 32 -> 419, 26 -> 392, 37 -> 448, 38 -> 434, 64 -> 411."""
        d, F, S = self.cfg.d, self.cfg.ffn, int(tokens_per_protein)
        best, best_score = lo, -1.0
        for B in range(lo, hi + 1):
            M, tot, cost = B * S, 0, 0
            for N, K, t in ((3 * d, d, 256), (d, d, 128 if d < 2560 else 256), (F, d, 256), (d, F, 256)):
                tiles = -(-M // t) * -(-N // t)
                slots = n_cu * (1 if t == 256 else 2)
                cost += -(-tiles // slots) * slots * t * t * K
                tot += M * N * K
            score = tot / cost - 0.0015 * B
            if score > best_score:
                best, best_score = B, score
        return best

    # ---- packing ---------------------------------------------------------------------------------
    @staticmethod
    def pack(rows, mask_pads=True):
        """rows int64 CPU [B',S] (right-padded) -> packed tokens + index arrays (all CPU int32).
        mask_pads=True packs only the non-pad prefix of each row; False keeps pads as tokens (Q12)."""
        Bp, S = rows.shape
        lens = (rows != PAD_ID).sum(1) if mask_pads else torch.full((Bp,), S, dtype=torch.int64)
        real = (rows != PAD_ID).sum(1)
        cu = torch.zeros(Bp + 1, dtype=torch.int64)
        cu[1:] = lens.cumsum(0)
        vt_cu = torch.zeros(Bp + 1, dtype=torch.int64)
        vt_cu[1:] = ((lens + 31) // 32 * 32).cumsum(0)
        idx = torch.arange(S)[None, :] < lens[:, None]
        tokens = rows[idx]
        pos = torch.arange(S)[None, :].expand(Bp, S)[idx]
        return dict(tokens=tokens.to(torch.int32), pos=pos.to(torch.int32), cu=cu.to(torch.int32), vt_cu=vt_cu.to(torch.int32),
                    lens=lens, real=real, ntok=int(cu[-1]), nseq=Bp, max_len=int(lens.max()), vt_total=int(vt_cu[-1]))

    def encode_packed(self, pk, mask_pads=True, borrow=False):
        """borrow=True: the caller consumes the result before the next call of this shape is enqueued (stream order) -- the persistent output
        buffer of a replayed launch chain is returned as it is instead of a copy."""
        dev = self.device
        if pk["max_len"] > self.cfg.max_len:
            raise ValueError(f"sequence of {pk['max_len']} tokens exceeds the rotary table ({self.cfg.max_len})")
        names = ("tokens", "pos", "cu", "vt_cu")
        # Short inputs (one protein: 230 launches of 5-35 us) are replayed from a captured launch chain inside pcy_esm_encode, which
        # needs the SAME device addresses from call to call: persistent index / output buffers per (tokens, sequences), a few shapes kept
        slot = None
        if dev.type == "cuda" and pk["ntok"] <= self.GRAPH_MAX_TOKENS:
            key = (pk["ntok"], pk["nseq"])
            slot = self._enc_slots.pop(key, None)
            if slot is None:
                nbytes = 2 * ((pk["ntok"] * 4 + 255) // 256 * 256) + 2 * (((pk["nseq"] + 1) * 4 + 255) // 256 * 256)
                slot = dict(idx=torch.empty(nbytes, dtype=torch.uint8, device=dev), hidden=torch.empty(pk["ntok"], self.cfg.d, dtype=BF16, device=dev))
            self._enc_slots[key] = slot                      # (most recently used last)
            while len(self._enc_slots) > 8:
                self._enc_slots.pop(next(iter(self._enc_slots)))
        t = dict(zip(names, _h2d_many([pk[k] for k in names], dev, into=None if slot is None else slot["idx"])))
        hidden = torch.empty(pk["ntok"], self.cfg.d, dtype=BF16, device=dev) if slot is None else slot["hidden"]
        L.check(self.ctx.lib.pcy_esm_encode(self.ctx.h, C.byref(self.desc), _p(t["tokens"]), _p(t["pos"]), _p(t["cu"]), _p(t["vt_cu"]),
                                            pk["ntok"], pk["nseq"], pk["max_len"], pk["vt_total"], int(mask_pads), _p(hidden)),
                "pcy_esm_encode")
        return hidden if (slot is None or borrow) else hidden.clone()    # (the persistent buffer is rewritten by the next call of this shape)

    def hidden_states(self, rows, mask_pads=True):
        """rows int64 [B',S] -> padded [B',S,d] representations (pad slots zero-filled; test helper)."""
        pk = self.pack(rows.cpu(), mask_pads)
        h = self.encode_packed(pk, mask_pads)
        Bp, S = rows.shape
        out = torch.zeros(Bp, S, self.cfg.d, dtype=BF16, device=self.device)
        idx = (torch.arange(S)[None, :] < pk["lens"][:, None]).to(self.device)
        out[idx] = h
        return out

    def lm_logits(self, hidden):
        """masked-LM logits [..., vocab] of final-layer states [..., d]: linear + bias -> ESM gelu (op by op, as fair-esm's
        `gelu`) -> LayerNorm -> tied decoder + bias; every Linear / norm rounds to bf16 like the eager reference"""
        if self.lm_head is None:
            raise ValueError("this ESM checkpoint carries no masked-LM head (lm_head.*): logits are not available")
        h = self.lm_head
        x = hidden.reshape(-1, hidden.shape[-1]).contiguous()
        y = self.ctx.gemm(x, h["wd"], bias=h["bd"], epi=L.EPI_GELU_ESM)
        y = self.ctx.layernorm(y, h["lw"], h["lb"], self.cfg.ln_eps)
        out = self.ctx.gemm(y, h["wo"], bias=h["bo"])
        return out.reshape(*hidden.shape[:-1], h["wo"].shape[0])

    def forward_tokens(self, tokens, mask_pads=True, max_protein_len=1024, long_protein_strategy="split", want_logits=True):
        """`ESM_PLM.forward(tokens, aggregate=False)` (esm.py:547-558): per-position final-layer states [B, max_eos + 1, d] (the
        chunks of a split protein laid end to end by `reverse_batched_split`) and, when the checkpoint has the head, the
        masked-LM logits of the same positions.  Pad positions are zero (the reference leaves whatever the padded forward
        computed there; they are masked by every consumer)."""
        from .sequences import reverse_batched_split, split_or_truncate_long_seq
        rows, keys, eos_loc = split_or_truncate_long_seq(tokens.cpu().long(), PAD_ID, EOS_ID, long_protein_strategy, max_protein_len)
        z = self.hidden_states(rows, mask_pads)
        logits = self.lm_logits(z) if (want_logits and self.lm_head is not None) else None
        if keys is not None:
            z = reverse_batched_split(z, keys, eos_loc)
            if logits is not None:
                logits = reverse_batched_split(logits, keys, eos_loc)
        return z, logits

    def forward(self, tokens, pooling="mean", correction=False, mask_pads=True, max_protein_len=1024):
        """`ESM_PLM.forward(tokens, aggregate=True)`: split long proteins, encode, pool per original protein."""
        rows, keys = batched_split_long_seq(tokens.cpu().long(), max_protein_len=max_protein_len)
        pk = self.pack(rows, mask_pads)
        h = self.encode_packed(pk, mask_pads, borrow=True)   # pooled below, on the same stream, before anything can overwrite it
        nprot = int(keys.max()) + 1
        seg = [0]
        rng = []
        for i in range(nprot):
            for r in (keys == i).nonzero(as_tuple=True)[0].tolist():  # chunk rows in batch order (esm.py:168-170)
                rng += [int(pk["cu"][r]), int(pk["real"][r])]          # pads never enter the pool (esm.py:171-173)
            seg.append(len(rng) // 2)
        mode = {"mean": L.POOL_MEAN_CORRECTED if correction else L.POOL_MEAN, "max": L.POOL_MAX}[pooling]
        seg_t, rng_t = _h2d_many([torch.tensor(seg, dtype=torch.int32), torch.tensor(rng, dtype=torch.int32)], self.device)
        return self.ctx.pool(h, seg_t, rng_t, nprot, mode)
