"""Row A0 of SURVEY.md section 8a in the product: residues -> ESM token ids, the way the reference reaches fair-esm's
`Alphabet` / `BatchConverter` through `convert_batch_protein` (procyon/data/data_utils.py:53-70; callers
data/it_collator.py:462-473, examples/paper_analyses/protpep_qa_scores.py:70-81), and the long-protein strategies of
`batched_split_long_seq` (training/train_utils.py:1497-1596).

"ESM-1b" alphabet (model_unified.py:216): <cls> <pad> <eos> <unk>, the 25 residue / ambiguity letters in fair-esm's order,
'.', '-', <null_1>, <mask>; a sequence becomes <cls> + one id per character (unknown letters -> <unk>) + <eos>, right-padded
with <pad> to the longest row of the batch."""
from __future__ import annotations

import torch

ESM_TOKENS = ("<cls>", "<pad>", "<eos>", "<unk>", "L", "A", "G", "V", "S", "E", "R", "T", "I", "D", "P", "K", "Q", "N", "F", "Y",
              "M", "H", "W", "C", "X", "B", "U", "Z", "O", ".", "-", "<null_1>", "<mask>")


class EsmAlphabet:
    """The attributes of fair-esm's `Alphabet` the reference touches (model_unified.py:216-222, data_utils.py:74-85)."""

    def __init__(self):
        self.all_toks = list(ESM_TOKENS)
        self.tok_to_idx = {t: i for i, t in enumerate(self.all_toks)}
        self.cls_idx, self.padding_idx, self.eos_idx, self.unk_idx = 0, 1, 2, 3
        self.mask_idx = self.tok_to_idx["<mask>"]
        self.prepend_bos = self.append_eos = True

    def __len__(self):
        return len(self.all_toks)

    def get_idx(self, tok):
        return self.tok_to_idx.get(tok, self.unk_idx)

    def get_tok(self, idx):
        return self.all_toks[idx]

    def encode(self, seq):
        """ids of one residue string (no <cls>/<eos>); special tokens written out in the text (e.g. "<mask>") are kept whole"""
        out, i = [], 0
        while i < len(seq):
            if seq[i] == "<":
                j = seq.find(">", i)
                if j > 0 and seq[i:j + 1] in self.tok_to_idx:
                    out.append(self.tok_to_idx[seq[i:j + 1]])
                    i = j + 1
                    continue
            out.append(self.get_idx(seq[i]))
            i += 1
        return out

    def get_batch_converter(self, truncation_seq_length=None):
        return EsmBatchConverter(self, truncation_seq_length)


ESM_ALPHABET = EsmAlphabet()


class EsmBatchConverter:
    """`alphabet.get_batch_converter()`: [(label, sequence), ...] -> (labels, sequences, int64 tokens [B, max_len + 2])."""

    def __init__(self, alphabet=ESM_ALPHABET, truncation_seq_length=None):
        self.alphabet = alphabet
        self.truncation_seq_length = truncation_seq_length

    def __call__(self, raw_batch):
        labels, seqs = zip(*raw_batch) if len(raw_batch) else ((), ())
        enc = [self.alphabet.encode(s) for s in seqs]
        if self.truncation_seq_length:
            enc = [e[:self.truncation_seq_length] for e in enc]
        a = self.alphabet
        width = max((len(e) for e in enc), default=0) + 2
        toks = torch.full((len(enc), width), a.padding_idx, dtype=torch.int64)
        for i, e in enumerate(enc):
            toks[i, 0] = a.cls_idx
            if e:
                toks[i, 1:len(e) + 1] = torch.tensor(e, dtype=torch.int64)
            toks[i, len(e) + 1] = a.eos_idx
        return list(labels), list(seqs), toks


def tokenize_proteins(sequences, truncation_seq_length=None):
    """list of residue strings -> ESM token matrix (what `ESM_PLM.forward` / `UnifiedProCyon.forward_sequences` consume)."""
    return EsmBatchConverter(ESM_ALPHABET, truncation_seq_length)([("", s) for s in sequences])[2]


def split_or_truncate_long_seq(toks, padding_idx=1, eos_idx=2, long_protein_strategy="split", max_protein_len=1024):
    """`batched_split_long_seq` (train_utils.py:1497-1596) -> (new_toks, batch_keys, eos_loc), non-mutating.

    'split': rows whose <eos> sits beyond column max_protein_len + 1 keep their first max_protein_len residues (+ <eos>); every
    further stretch of max_protein_len residues becomes a row <cls> + residues + <eos> appended AFTER all original rows;
    batch_keys[r] = original row of output row r; eos_loc = the original <eos> columns.
    'truncate' (:1575-1588): the matrix is cut to max_protein_len + 2 columns and a row that is still running at the cut gets
    <eos> in the last column; batch_keys and eos_loc are None.  (The reference's statement `new_toks[:, no_pad] = eos_idx` indexes
    COLUMNS with the row mask -- it raises or overwrites whole columns for any batch it applies to; the evident intent,
    `new_toks[no_pad, -1] = eos_idx`, is what is implemented.)"""
    if long_protein_strategy == "split":
        from .engine import batched_split_long_seq
        eos_loc = [int((toks[i] == eos_idx).nonzero(as_tuple=True)[0][0]) for i in range(toks.shape[0])]
        new, keys = batched_split_long_seq(toks, padding_idx=padding_idx, eos_idx=eos_idx, max_protein_len=max_protein_len)
        return new, keys, eos_loc
    if long_protein_strategy == "truncate":
        if toks.shape[1] > max_protein_len + 2:
            new = toks[:, :max_protein_len + 2].clone()
            no_pad = new[:, -1] != padding_idx
            new[no_pad, -1] = eos_idx
        else:
            new = toks
        return new, None, None
    raise NotImplementedError(f"long_protein_strategy={long_protein_strategy!r}")


def reverse_batched_split(protein_embeds, batch_keys, eos_locs):
    """`reverse_batched_split` (train_utils.py:1599-1649): per-position states of the chunk rows [B', S, d] -> one row per
    ORIGINAL protein [B, max(eos_locs) + 1, d].  The chunks of a protein are laid end to end in row order, dropping the last
    column of every chunk but the last (the <eos> a split chunk was re-terminated with) and the first column of every chunk but
    the first (its re-inserted <cls>); then padded with zeros / cut to max(eos_locs) + 1 positions."""
    d = protein_embeds.shape[-1]
    max_size = int(max(int(e) for e in eos_locs)) + 1
    out = []
    for i in range(int(batch_keys.max()) + 1):
        rows = (batch_keys == i).nonzero(as_tuple=True)[0].sort()[0]
        if rows.numel() == 0:
            continue
        parts = []
        for j, r in enumerate(rows.tolist()):
            lo = 1 if j > 0 else 0
            hi = protein_embeds.shape[1] - (1 if j < rows.numel() - 1 else 0)
            parts.append(protein_embeds[r, lo:hi])
        cat = torch.cat(parts, 0)
        if cat.shape[0] < max_size:
            cat = torch.cat([cat, torch.zeros(max_size - cat.shape[0], d, dtype=cat.dtype, device=cat.device)], 0)
        out.append(cat[:max_size])
    return torch.stack(out)
