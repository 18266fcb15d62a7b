"""Mirror of `ESM_PLM` (/root/reference/procyon/model/esm.py:318-558) over the HIP ESM2 engine."""
from __future__ import annotations

import torch

from ..engine import EsmConfig, EsmEngine

# layer counts / widths from esm.py:378-421; heads / FFN from the public ESM2 cards
ESM2_GEOMETRY = {
    "8m": dict(d=320, n_layers=6, n_heads=20, ffn=1280),
    "35m": dict(d=480, n_layers=12, n_heads=20, ffn=1920),
    "150m": dict(d=640, n_layers=30, n_heads=20, ffn=2560),
    "650m": dict(d=1280, n_layers=33, n_heads=20, ffn=5120),
    "3b": dict(d=2560, n_layers=36, n_heads=40, ffn=10240),
    "15b": dict(d=5120, n_layers=48, n_heads=40, ffn=20480),
}


class ESM_PLM:
    """forward(tokens, aggregate=True) -> (z [B,D], logits=None); forward(tokens, aggregate=False) -> (z [B,len,D], logits
    [B,len,33] or None when the checkpoint has no masked-LM head).

    pooling_method / protein_pooling_correction_option / long_protein_strategy / max_protein_len keep the
    reference's meaning (esm.py:318-376, training_args_IT.py:65-103).  `official=True` reproduces the
    HF-"official" call that passes no attention mask (esm.py:533, quirk Q12).  The masked-LM logits the
    reference also returns are never consumed on the inference path (model_unified.py:1060) and are not
    computed on the pooled path: `logits` is None there."""

    def __init__(self, state_dict, cfg: EsmConfig, pooling_method="max", protein_pooling_correction_option=False,
                 long_protein_strategy="split", max_protein_len=1024, official=False, device=None):
        if long_protein_strategy not in ("split", "truncate"):
            raise NotImplementedError(f"long_protein_strategy={long_protein_strategy!r} (train_utils.py:1497-1596 knows 'split' and 'truncate')")
        self.long_protein_strategy = long_protein_strategy
        # fp32 checkpoints keep their fp32 tensors until the caller asks for bf16 (see LlamaPostTokenization)
        self._src_f32 = dict(state_dict) if any(v.dtype == torch.float32 for v in state_dict.values()) else None
        self._engine_f32 = None
        self._cfg = cfg
        self.engine = EsmEngine(state_dict, cfg, device)
        self.embedding_size = cfg.d
        self.repr_layer = cfg.n_layers
        self.pooling_method = pooling_method.lower()
        if self.pooling_method not in ("mean", "max"):
            raise NotImplementedError(f"Protein pooling method {pooling_method} is not implemented")
        self.correction = protein_pooling_correction_option
        self.max_protein_len = max_protein_len
        self.official = official
        self.padding_idx, self.eos_idx = 1, 2

    def eval(self):
        return self

    @property
    def engine_f32(self):
        if self._engine_f32 is None:
            if self._src_f32 is None:
                raise RuntimeError("fp32 arithmetic was asked for, but this protein encoder holds no fp32 weights (built from bf16 tensors, or "
                                   ".bfloat16() was called before)")
            from ..engine_f32 import EsmEngineF32
            self._engine_f32 = EsmEngineF32(self._src_f32, self._cfg, self.engine.device)
        return self._engine_f32

    def drop_fp32(self):
        self._src_f32 = self._engine_f32 = None

    def forward_f32(self, tokens):
        """`forward(tokens, aggregate=True)` in fp32 arithmetic (the model was never cast to bf16) -> (z [B,D] fp32, None)"""
        if self.long_protein_strategy == "truncate":
            from ..sequences import split_or_truncate_long_seq
            tokens, _, _ = split_or_truncate_long_seq(tokens.cpu().long(), self.padding_idx, self.eos_idx, "truncate", self.max_protein_len)
        return self.engine_f32.forward(tokens, pooling=self.pooling_method, correction=self.correction, mask_pads=not self.official,
                                       max_protein_len=self.max_protein_len), None

    def forward(self, tokens, aggregate=True):
        if self.long_protein_strategy == "truncate":       # cut to max_protein_len residues, re-terminate (train_utils.py:1575-1588)
            from ..sequences import split_or_truncate_long_seq
            tokens, _, _ = split_or_truncate_long_seq(tokens.cpu().long(), self.padding_idx, self.eos_idx, "truncate", self.max_protein_len)
        if not aggregate:      # per-position states + masked-LM logits (esm.py:547-558)
            return self.engine.forward_tokens(tokens, mask_pads=not self.official, max_protein_len=self.max_protein_len)
        z = self.engine.forward(tokens, pooling=self.pooling_method, correction=self.correction,
                                mask_pads=not self.official, max_protein_len=self.max_protein_len)
        return z, None

    __call__ = forward
