"""Deterministic stand-in tokenizer for synthetic runs (no tokenizer files exist on the build/GPU boxes).

Implements the slice of the HuggingFace tokenizer interface that `UnifiedProCyon` uses
(/root/reference/procyon/model/model_unified.py:1088-1133,1193-1230,1007-1010): callable on a list of strings,
`encode`, `batch_decode`, `add_tokens`, `convert_tokens_to_ids`, `eos/pad/sep` attributes.  Words hash to ids in
[0, n_text); special tokens are registered in the reference's order so that, for the Llama-3 geometry, they land on
the ids of SURVEY App. A ([CLS] 128256 ... [EXT] 128263).  With a real Llama-3 `tokenizer.json` available, pass a
`transformers.AutoTokenizer` to `UnifiedProCyon` instead (after the same `add_tokens` sequence)."""
from __future__ import annotations

import re
import zlib

PROCYON_ADDED_TOKENS = ["[CLS]", "[PAD]", "<|protein|>", "[PROT]", "[ANSWER]", "<|struct|>", "<|drug|>", "[EXT]"]


class SyntheticTokenizer:
    def __init__(self, n_text=128000, base_vocab=128256, bos_token_id=128000, eos_token_id=128001):
        self.n_text = n_text
        self.bos_token_id, self.eos_token_id = bos_token_id, eos_token_id
        self.bos_token, self.eos_token = "<|begin_of_text|>", "<|end_of_text|>"
        self.padding_side = "right"
        self._added = {}
        self._next = base_vocab
        self.sep_token = self.pad_token = None
        self.sep_token_id = self.pad_token_id = None
        self.add_tokens("[CLS]")
        self.sep_token, self.sep_token_id = "[CLS]", self._added["[CLS]"]
        self.add_tokens("[PAD]")
        self.pad_token, self.pad_token_id = "[PAD]", self._added["[PAD]"]
        for t in PROCYON_ADDED_TOKENS[2:]:
            self.add_tokens(t)

    def __len__(self):
        return self._next

    def add_tokens(self, tok):
        if tok not in self._added:
            self._added[tok] = self._next
            self._next += 1
            self._pat = re.compile("(" + "|".join(re.escape(t) for t in sorted(self._added, key=len, reverse=True)) + ")")

    def convert_tokens_to_ids(self, tok):
        return self._added[tok]

    def _word_id(self, w):
        return zlib.crc32(w.encode()) % self.n_text

    def encode(self, text, add_special_tokens=True):
        ids = [self.bos_token_id] if add_special_tokens else []
        for piece in self._pat.split(text):
            if piece in self._added:
                ids.append(self._added[piece])
            else:
                # a leading space is part of the following word, as in byte-level BPE (" yes" != "yes")
                ids += [self._word_id(w) for w in re.findall(r"\s*\S+", piece)]
        return ids

    def __call__(self, texts, padding=False, truncation=False, add_special_tokens=True, max_length=None, **_):
        single = isinstance(texts, str)
        out = []
        for t in ([texts] if single else texts):
            ids = self.encode(t, add_special_tokens)
            if truncation and max_length is not None:
                ids = ids[:max_length]
            out.append(ids)
        from types import SimpleNamespace
        res = {"input_ids": out[0] if single else out}
        return _Enc(res)

    def batch_decode(self, ids):
        rev = {v: k for k, v in self._added.items()}
        rev[self.eos_token_id], rev[self.bos_token_id] = self.eos_token, self.bos_token
        rows = ids.tolist() if hasattr(ids, "tolist") else ids
        return [" ".join(rev.get(int(i), f"w{int(i)}") for i in row) for row in rows]


class _Enc(dict):
    @property
    def input_ids(self):
        return self["input_ids"]
