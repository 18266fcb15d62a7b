"""`procyon.data.data_utils`: DATA_DIR / HOME_DIR and the residue tokeniser (reference: procyon/data/data_utils.py:19-26,53-91,
354-362).  The reference asserts DATA_DIR at import; here a missing variable only fails where a data file is opened."""
import os
from typing import List, Optional

import torch

from procyon_amd.sequences import ESM_ALPHABET, EsmBatchConverter, tokenize_proteins  # noqa: F401

DATA_DIR = os.getenv("DATA_DIR")
HOME_DIR = os.getenv("HOME_DIR") or ""
MODEL_DIR = os.path.join(DATA_DIR, "model_outputs/pretrain") if DATA_DIR else None


def require_data_dir():
    d = os.getenv("DATA_DIR") or DATA_DIR
    if not d:
        raise EnvironmentError("DATA_DIR must be set (ProCyon-Instruct data root)")
    return d


def convert_batch_protein(ids: List[int], is_protein_tokenized: bool, batch_converter, protein_sequences, protein_tokens,
                          protein_tokenizer, max_protein_len: Optional[int] = None) -> torch.Tensor:
    """`convert_batch_protein` (data_utils.py:53-91): protein ids -> ESM token matrix.  Untokenised: <cls> + residues + <eos>
    through the batch converter (right-padded with <pad>); tokenised: rows already carry <cls>/<eos>, optionally truncated."""
    if not is_protein_tokenized:
        _, _, batch_toks = batch_converter([("", protein_sequences[idx]) for idx in ids])
        return batch_toks
    seqs = [protein_tokens[idx] for idx in ids]
    assert seqs[0][0] == protein_tokenizer.cls_idx
    assert seqs[0][-1] == protein_tokenizer.eos_idx
    if max_protein_len is not None:
        seqs = [s[:max_protein_len] for s in seqs]
    out = torch.full((len(ids), max(len(s) for s in seqs)), protein_tokenizer.padding_idx, dtype=torch.long)
    for i, s in enumerate(seqs):
        out[i, :len(s)] = torch.as_tensor(s)
    return out


def get_text_sequences_compositions(text_type, text_info, column_subset=None):
    """`get_text_sequences_compositions` (data_utils.py:354-362): the description columns of one text dataset."""
    if text_type == "protein":
        return None
    from procyon.data.constants import entity_description_names
    cols = None if column_subset is None else column_subset[text_type]
    if cols is None or getattr(cols, "deferred", False):      # no column table at hand: every description column of the frame
        cols = entity_description_names(text_type, text_info)
    return text_info[cols]
