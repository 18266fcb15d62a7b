"""Mirror of `LlamaPostTokenization` (/root/reference/procyon/model/pmc_llama.py:415-596) over the HIP engine."""
from __future__ import annotations

from types import SimpleNamespace

import torch

from ..engine import GenState, KVCache, LlamaConfig, LlamaEngine


class _PastF32:
    """fp32 KV cache handle of an fp32 generation: `.cache` (engine_f32.KVCacheF32), `.t` keys held"""

    def __init__(self, cache, t):
        self.cache, self.t = cache, t


class _Past:
    """past_key_values: indexable [layer][0|1] -> [B,Hkv,t,dh] VIEWS of the engine cache, so the in-place row
    re-indexing the reference's beam search performs (model_unified.py:830-832) acts on the live cache.  The views are
    built on access (a decode loop that never looks at them does not pay 2L tensor slicings per step)."""

    def __init__(self, cache: KVCache, t: int):
        self.cache, self.t = cache, t

    def __len__(self):
        return self.cache.k.shape[0]

    def __getitem__(self, l):
        if isinstance(l, slice):
            return [self[i] for i in range(len(self))[l]]
        if l < 0:
            l += len(self)
        if not 0 <= l < len(self):
            raise IndexError(l)
        return [self.cache.k[l, :, :, :self.t], self.cache.v[l, :, :, :self.t]]

    def __iter__(self):
        return (self[l] for l in range(len(self)))


class _HiddenStates:
    """Lazy stand-in for `outputs.hidden_states`, handed out ONLY when the caller asks for it (`lazy_hidden=True`: the engine-backed
    `UnifiedProCyon`, which reads `hidden_states[-1]` alone -- ret_token_access='last', model_unified.py:556-559).  It keeps the last
    state; any other access (another index, iteration, len()) materialises the whole tuple by running the (deterministic) prefill once
    more with every layer's state written out.  It is NOT a tuple -- `torch.stack(obj, -1)` rejects it -- which is why a caller that
    passes the reference's arguments only (INTEGRATION.md route B: the reference's own `UnifiedProCyon` over this sub-module, whose
    ret_token_access='all' branch does `torch.stack(outputs.hidden_states, dim=-1)`, model_unified.py:563) gets a real tuple."""

    def __init__(self, n, last, materialise):
        self._n, self._last, self._mat, self._all = n, last, materialise, None

    def _full(self):
        if self._all is None:
            self._all = tuple(self._mat())
        return self._all

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, int) and (i == -1 or i == self._n - 1) and self._last is not None:
            return self._last
        return self._full()[i]

    def __iter__(self):
        return iter(self._full())


class LlamaPostTokenization:
    """forward(input_embeds|input_ids, attn_masks, full_labels, past_key_values, use_cache, output_attentions)
    -> object with .logits, .past_key_values, .hidden_states, .loss   (pmc_llama.py:546-596).

    Exactly one of input_embeds / input_ids (:562).  With the reference's arguments only, the returned object is the
    reference's: `.logits` [B,T,V] for every row and `.hidden_states` a real `tuple` of the L+1 [B,T,d] tensors (the wrapper always
    runs its model with `output_hidden_states=True`, pmc_llama.py:575,584) -- embeddings, outputs of layers 0..L-2, final-normed
    output of layer L-1.
    Extensions used by the engine-backed `UnifiedProCyon` (never required):
      * `lazy_hidden=True`: `.hidden_states` is a `_HiddenStates` (the last state only, the rest materialised on access).
      * `logit_positions`: LongTensor [B] -> logits only at those positions ([B,1,V]); the reference always materialises
        [B,T,V] (1 GB per 2048-token row).  None = all rows.
      * `want_hidden=False` skips the final hidden state, `hidden_sum_positions` asks for the sum over all L+1 states at
        given rows only (ret_token_access='all' without the tuple), `output_hidden_states=True` materialises the tuple eagerly.
      * full_labels / loss / output_attentions: training-side, not computed (loss=None).
      * max_new_tokens: KV capacity reserved beyond the prompt when use_cache=True.
    """

    def __init__(self, state_dict, cfg: LlamaConfig, device=None, max_new_tokens=256):
        # fp32 checkpoints keep their fp32 tensors (references, wherever they live) until the caller asks for bf16 -- as the reference's
        # module holds fp32 parameters until `.bfloat16()` -- so that a caller which never does gets fp32 arithmetic (engine_f32)
        self._src_f32 = dict(state_dict) if any(v.dtype == torch.float32 for v in state_dict.values()) else None
        self._engine_f32 = None
        self.engine = LlamaEngine(state_dict, cfg, device)
        self.cfg = cfg
        self.max_new_tokens = max_new_tokens
        self.model = SimpleNamespace(vocab_size=cfg.vocab, config=SimpleNamespace(hidden_size=cfg.d))
        self._state = None

    def eval(self):
        return self

    @property
    def engine_f32(self):
        """the fp32 prefill engine, built on first use from the retained fp32 weights"""
        if self._engine_f32 is None:
            if self._src_f32 is None:
                raise RuntimeError("fp32 arithmetic was asked for, but this text encoder holds no fp32 weights (built from bf16 tensors, or "
                                   ".bfloat16() was called before): load the checkpoint again and do not call .bfloat16()")
            from ..engine_f32 import LlamaEngineF32
            self._engine_f32 = LlamaEngineF32(self._src_f32, self.cfg, self.engine.device)
        return self._engine_f32

    def drop_fp32(self):
        self._src_f32 = self._engine_f32 = None

    def get_input_embeddings(self):
        return self.engine.embed

    def forward(self, input_embeds=None, input_ids=None, attn_masks=None, full_labels=None, past_key_values=None,
                use_cache=False, output_attentions=None, logit_positions=None, want_hidden=True, hidden_sum_positions=None,
                output_hidden_states=False, lazy_hidden=False):
        assert (input_embeds is not None) != (input_ids is not None), "Only one of input_embeds or input_ids can be provided"
        if isinstance(past_key_values, _PastF32):      # cached decode of an fp32 generation
            assert input_ids is not None and input_ids.shape[1] == 1, "cached decode takes input_ids [B,1]"
            cache, t = past_key_values.cache, past_key_values.t
            logits = self.engine_f32.decode(cache, input_ids.view(-1), t)
            return SimpleNamespace(logits=logits.view(input_ids.shape[0], 1, -1), past_key_values=_PastF32(cache, t + 1), hidden_states=None, loss=None)
        if input_embeds is not None and input_embeds.dtype == torch.float32:
            return self._forward_f32(input_embeds, attn_masks, past_key_values, use_cache, logit_positions, want_hidden, hidden_sum_positions)
        eng = self.engine
        if past_key_values is None:
            if input_embeds is None:
                input_embeds = eng.embed_tokens(input_ids)
            B, T, _ = input_embeds.shape
            cache = eng.new_cache(B, T + (self.max_new_tokens if use_cache else 0))
            if logit_positions is None:
                rows = torch.arange(B * T, dtype=torch.int32)
            else:
                rows = (torch.arange(B) * T + logit_positions.cpu().long()).to(torch.int32)
            hsum = None
            embeds_dev = input_embeds.to(eng.device)
            L1 = self.cfg.n_layers + 1

            def materialise():      # the whole tuple, from a second (deterministic) pass; the KV cache it fills is a scratch one
                _, hall = eng.prefill_all(embeds_dev, attn_masks, eng.new_cache(B, T), None)
                return [hall[i] for i in range(L1)]

            if output_hidden_states or not lazy_hidden:
                logits, hall = eng.prefill_all(embeds_dev, attn_masks, cache, rows)
                hs = tuple(hall[i] for i in range(L1))
                if hidden_sum_positions is not None:
                    flat = hall.view(L1, B * T, -1)[:, hidden_sum_positions.to(eng.device).long()]
                    hsum = flat.float().sum(0).to(hall.dtype)
            elif hidden_sum_positions is not None:
                # ret_token_access='all': sum of all L+1 hidden states, only at the requested flat token rows
                logits, hidden, hsum = eng.prefill(embeds_dev, attn_masks, cache, rows, want_hidden=want_hidden,
                                                   sum_rows=hidden_sum_positions)
                hs = _HiddenStates(L1, hidden, materialise)
            else:
                logits, hidden = eng.prefill(embeds_dev, attn_masks, cache, rows, want_hidden=want_hidden)
                hs = _HiddenStates(L1, hidden, materialise)
            logits = logits.view(B, -1, self.cfg.vocab)
            past = _Past(cache, T) if use_cache else None
            return SimpleNamespace(logits=logits, past_key_values=past, hidden_states=hs, hidden_state_sum_rows=hsum, loss=None)
        # cached decode: one new token per row, no mask, position = cache length (quirks Q1/Q2)
        assert input_ids is not None and input_ids.shape[1] == 1, "cached decode takes input_ids [B,1]"
        cache, t = past_key_values.cache, past_key_values.t
        B = input_ids.shape[0]
        if t + 1 > cache.Tmax:
            raise ValueError(f"KV cache capacity {cache.Tmax} exhausted; raise max_new_tokens")
        if self._state is None or self._state.next_tok.shape[0] != B:
            self._state = GenState(B, self.cfg.vocab, 1, eng.device)
        st = self._state
        st.pos.fill_(t)
        st.next_tok.copy_(input_ids.view(-1).to(torch.int32))
        eng.decode_graph(cache, st, B)
        return SimpleNamespace(logits=st.logits.clone().view(B, 1, -1), past_key_values=_Past(cache, t + 1),
                               hidden_states=None, loss=None)

    def _forward_f32(self, input_embeds, attn_masks, past_key_values, use_cache, logit_positions, want_hidden, hidden_sum_positions):
        """fp32 embeddings in -> the fp32 prefill (the callers that never call `.bfloat16()`); use_cache=True keeps the K / V rows in an fp32
        cache for the cached decode steps of an fp32 generation (/root/reference/scripts/caption_bulk.py:70-73, 123-132)"""
        if past_key_values is not None:
            raise RuntimeError("a prefill with input_embeds takes no past_key_values")
        eng = self.engine_f32
        B, T, _ = input_embeds.shape
        cache = eng.new_cache(B, T + self.max_new_tokens) if use_cache else None
        rows = "all" if logit_positions is None else (torch.arange(B) * T + logit_positions.cpu().long())
        hsum = None
        if hidden_sum_positions is not None:
            logits, hidden, hsum = eng.prefill(input_embeds, attn_masks, rows, want_hidden=True, sum_rows=hidden_sum_positions, cache=cache)
        else:
            logits, hidden = eng.prefill(input_embeds, attn_masks, rows, want_hidden=True, cache=cache)
        hs = _HiddenStates(self.cfg.n_layers + 1, hidden, lambda: (_ for _ in ()).throw(
            RuntimeError("the fp32 path keeps the final hidden state only (hidden_states[-1])")))
        return SimpleNamespace(logits=logits.view(B, -1, self.cfg.vocab), past_key_values=_PastF32(cache, T) if use_cache else None,
                               hidden_states=hs, hidden_state_sum_rows=hsum, loss=None)

    __call__ = forward
