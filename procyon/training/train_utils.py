"""`procyon.training.train_utils`: the helpers the inference entry points import from here (reference:
procyon/training/train_utils.py:1048-1117,1335-1345,1497-1596)."""
import random

import numpy as np
import torch

from procyon.training.training_args_IT import DataArgs, ModelArgs, TrainArgs  # noqa: F401  (retrieval_utils.py:15 imports DataArgs from here)
from procyon_amd.engine import batched_split_long_seq as _split
from procyon_amd.evaluate import get_after_answer_tokens, get_final_tokens, get_qa_scores  # noqa: F401


def set_seed(seed):
    """`set_seed` (train_utils.py:1335-1345): python, numpy and torch (CPU + every device) generators."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def batched_split_long_seq(toks, padding_idx, eos_idx, long_protein_strategy="split", max_protein_len=1024):
    """`batched_split_long_seq` (train_utils.py:1497-1596) -> (new_toks, batch_keys, eos_loc).  'split' cuts long proteins into
    chunks appended after the originals; 'truncate' keeps the first max_protein_len residues and re-terminates the row."""
    from procyon_amd.engine import split_or_truncate_long_seq
    return split_or_truncate_long_seq(toks, padding_idx, eos_idx, long_protein_strategy, max_protein_len)
