"""Oracle: ProCyon's own glue arithmetic around the encoders (test infrastructure).

Restates rows A0, A1, A3, A4, A5, A10, A11 of SURVEY.md section 8a.  Each function cites the
reference lines it follows; fixtures g1-g4/g8 (tests/golden) were produced by running the
reference's own functions, AST-extracted from /root/reference, on the same seeded inputs.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import esm_ref

# ESM-1b alphabet order: <cls>0 <pad>1 <eos>2 <unk>3 then residues.
ESM_TOKENS = ["<cls>", "<pad>", "<eos>", "<unk>", "L", "A", "G", "V", "S", "E", "R", "T", "I", "D",
              "P", "K", "Q", "N", "F", "Y", "M", "H", "W", "C", "X", "B", "U", "Z", "O", ".", "-",
              "<null_1>", "<mask>"]
ESM_TOK2ID = {t: i for i, t in enumerate(ESM_TOKENS)}


def convert_batch_protein(seqs):
    """A0: fair-esm BatchConverter as `convert_batch_protein` drives it
    (/root/reference/procyon/data/data_utils.py:53-70): <cls> + residues + <eos>, right-pad <pad>."""
    maxlen = max(len(s) for s in seqs)
    toks = torch.full((len(seqs), maxlen + 2), ESM_TOK2ID["<pad>"], dtype=torch.int64)
    for i, s in enumerate(seqs):
        toks[i, 0] = ESM_TOK2ID["<cls>"]
        for j, ch in enumerate(s):
            toks[i, j + 1] = ESM_TOK2ID.get(ch, ESM_TOK2ID["<unk>"])
        toks[i, len(s) + 1] = ESM_TOK2ID["<eos>"]
    return toks


def batched_split_long_seq(toks, padding_idx=1, eos_idx=2, max_protein_len=1024):
    """A1: `batched_split_long_seq(..., long_protein_strategy="split")`
    (/root/reference/procyon/training/train_utils.py:1497-1571).

    Works on a copy (the reference mutates its argument in place, :1568-1569).  Returns
    new_toks [B', max_protein_len+2], batch_keys [B'] (extra rows appended after all originals),
    eos positions per original row.
    """
    toks = toks.clone()
    W = toks.shape[1]
    cls_idx = int(toks[0, 0])
    eos_loc = [int((toks[i] == eos_idx).nonzero(as_tuple=True)[0][0]) for i in range(toks.shape[0])]
    to_add = []
    keys = list(range(toks.shape[0]))
    for i, e in enumerate(eos_loc):
        if e <= max_protein_len + 1:
            continue
        n_add = e // (max_protein_len + 1)
        for j in range(n_add):
            bot = (j + 1) * max_protein_len + 1
            row = torch.full((1, W), padding_idx, dtype=torch.int64)
            tail = toks[i, bot:].clone()
            row[0, 1:tail.shape[0] + 1] = tail
            row[0, 0] = cls_idx
            if j < n_add - 1:
                row[0, max_protein_len + 1] = eos_idx
                row[0, max_protein_len + 2:] = 1
            to_add.append(row)
            keys.append(i)
        toks[i, max_protein_len + 2:] = padding_idx
        toks[i, max_protein_len + 1] = eos_idx
    new = torch.cat([toks] + to_add, dim=0)[:, : max_protein_len + 2]
    return new, torch.tensor(keys, dtype=torch.int64), eos_loc


def protein_pooler(z, batch_keys, padmask, method="mean", correction=False):
    """A3: `ProteinPooler.forward` (/root/reference/procyon/model/esm.py:131-173).

    z [B',S,D]; per original protein concat its chunk rows, drop pad rows (mean only), then
    nanmean over every remaining token incl. each chunk's CLS/EOS (:147) or x[1:-1] (:144-145);
    'max': pads set to -inf first (:156-157, in place in the reference -- copy here).
    """
    z = z.clone()
    if method == "max":
        z[padmask] = -float("inf")
    outs = []
    for i in range(int(batch_keys.max()) + 1):
        sel = batch_keys == i
        if sel.sum() == 0:
            continue
        x = z[sel].reshape(-1, z.shape[-1])
        if method == "mean":
            x = x[~padmask[sel].reshape(-1)]
            x = x[1:-1] if correction else x
            outs.append(x.nanmean(dim=-2))
        elif method == "max":
            outs.append(x.max(dim=-2)[0])
        else:
            raise NotImplementedError(method)
    return torch.stack(outs)


def mlp_forward(x, layers):
    """A4: `create_mlp` stacks in eval mode (/root/reference/procyon/model/model_utils.py:13-41):
    Linear(+bias) -> [Dropout: identity] -> nn.GELU() (erf) ... -> Linear(+bias); n_layers == 1 is
    one bias-free Linear.  layers = [(W, b_or_None), ...]."""
    n = len(layers)
    for i, (w, b) in enumerate(layers):
        x = F.linear(x, w, b)
        if i < n - 1:
            x = F.gelu(x)
    return x


def left_pad_tensors(tensors, pad_value=0):
    """`left_pad_tensors` (/root/reference/procyon/model/model_utils.py:151-170)."""
    L = max(t.size(0) for t in tensors)
    toks, masks = [], []
    for t in tensors:
        p = L - t.size(0)
        toks.append(torch.cat([torch.full((p,), pad_value), t]))
        masks.append(torch.cat([torch.zeros(p), torch.ones(t.size(0))]))
    return torch.stack(toks), torch.stack(masks)


def prepare_input_embeddings(embed_w, input_ids, prot_idx, prot_soft=None, struct_idx=None,
                             struct_soft=(), drug_idx=None, drug_soft=None, ret_idx=None, roll_num=0):
    """A5: `_prepare_input_embeddings` (/root/reference/procyon/model/model_unified.py:1135-1175)."""
    z = F.embedding(input_ids, embed_w).clone()
    if prot_soft is not None:
        m = input_ids == prot_idx
        assert int(m.sum()) == prot_soft.shape[0]
        z[m] = prot_soft
    if len(struct_soft) > 0:
        m = input_ids == struct_idx
        for i in range(z.shape[0]):
            if m[i].sum() > 0:
                assert int(m[i].sum()) == struct_soft[i].shape[0]
                z[i, m[i], :] = struct_soft[i]
    if drug_soft is not None:
        m = input_ids == drug_idx
        assert int(m.sum()) == drug_soft.shape[0]
        z[m] = drug_soft
    ret = input_ids == ret_idx if ret_idx is not None else None
    if ret is not None and roll_num != 0:
        ret = ret.roll(roll_num, 1)
    return z, ret


def get_after_answer_tokens(text_toks, answer_token):
    """`get_after_answer_tokens(get_final=True)` (/root/reference/procyon/training/train_utils.py:1104-1117)."""
    out = []
    for i in range(text_toks.shape[0]):
        idx = (text_toks[i] == answer_token).nonzero(as_tuple=True)[0]
        out.append(int(idx.max()) + 1)
    return torch.tensor(out)


def qa_yes_no_probs(logits, text_toks, answer_token, yes_token, no_token):
    """A10 QA tail: `get_qa_logits_inference` (/root/reference/procyon/data/inference_utils.py:582-604):
    softmax over the vocabulary, row (last [ANSWER] index + 1) - 1, yes/no columns."""
    preds = logits.softmax(dim=-1)
    inds = get_after_answer_tokens(text_toks, answer_token)
    rows = preds[torch.arange(preds.shape[0]), inds - 1]
    return rows[:, yes_token], rows[:, no_token], rows


def retrieval_text_embedding(hidden_states, ret_mask, lm_projector_layers, ret_token_access="last"):
    """A10 retrieval tail (/root/reference/procyon/model/model_unified.py:556-581)."""
    if ret_token_access == "all":
        pooled = torch.stack(hidden_states, dim=-1).sum(dim=-1)
    else:
        pooled = hidden_states[-1]
    return mlp_forward(pooled[ret_mask], lm_projector_layers)


@torch.no_grad()
def esm_plm_forward(esm_sd, geom, tokens, *, pooling="mean", correction=False, mask_pads=True,
                    max_protein_len=1024):
    """A1+A2+A3: `ESM_PLM.forward(tokens, aggregate=True)` (/root/reference/procyon/model/esm.py:504-546)."""
    new_toks, keys, _ = batched_split_long_seq(tokens, max_protein_len=max_protein_len)
    z = esm_ref.esm_forward(esm_sd, geom, new_toks, mask_pads=mask_pads)
    padmask = new_toks == esm_ref.PAD_ID
    return protein_pooler(z, keys, padmask, method=pooling, correction=correction)


@torch.no_grad()
def forward_sequences(esm_sd, geom, tokens, shared_layers, token_layers=None, **kw):
    """A11: `forward_sequences` (/root/reference/procyon/model/model_unified.py:1029-1086)."""
    z = esm_plm_forward(esm_sd, geom, tokens, **kw)
    out = {"original": z, "shared": mlp_forward(z, shared_layers), "token": None}
    if token_layers is not None:
        out["token"] = mlp_forward(z, token_layers)
    return out


def retrieval_scores(query, targets):
    """A12 scoring (/root/reference/procyon/evaluate/framework/procyon.py:400-406):
    L2-normalise both sides in the embeddings' own dtype, Q @ T^T, then to float64."""
    q = F.normalize(query)
    t = F.normalize(targets)
    return (q @ t.T).to(torch.float64)
