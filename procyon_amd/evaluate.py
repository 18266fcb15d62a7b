"""Eval-framework plugin level of the drop-in boundary: the three `get_predictions` loops of
`procyon/evaluate/framework/procyon.py` on top of the engine-backed `UnifiedProCyon`, and the QA reader they use.

The reference classes load a checkpoint in their constructors (`UnifiedProCyon.from_pretrained`, :55-76, :128-140,
:216-229); here the constructor takes the ready model object (build it with `procyon_amd.checkpoint.build_model` or
`procyon_amd.synthetic_model.build`) -- everything after that line of the reference constructors, and every
`get_predictions` body, follows the reference: same loader protocol (batches are the collator dictionaries with
`reference_indices`; `data_loader.dataset.aaseq_type`; `data_loader.collate_fn._get_input_contexts` /
`._convert_batch`), same return types.  Retrieval scoring runs on the device (`pcy_retrieval_scores`) and returns the
reference's float64 CPU matrix."""
from collections import defaultdict
from collections.abc import Mapping

import numpy as np
import torch

from .engine import Context


def move_inputs_to_device(data, device):
    """`procyon/evaluate/framework/utils.py:46-61`."""
    if isinstance(data, Mapping):
        return type(data)({k: move_inputs_to_device(v, device) for k, v in data.items()})
    if isinstance(data, (tuple, list)):
        return type(data)(move_inputs_to_device(v, device) for v in data)
    if isinstance(data, torch.Tensor):
        return data.to(device=device)
    return data


def get_after_answer_tokens(text_toks, answer_token, get_final=True):
    """`procyon/training/train_utils.py:1104-1117`."""
    where_answer = (text_toks == answer_token).nonzero()
    if get_final:
        found = [where_answer[where_answer[:, 0] == i, 1].max().item() for i in range(text_toks.shape[0])]
        return torch.tensor(found, device=text_toks.device) + 1
    return where_answer[:, 1] + 1


def get_final_tokens(text_toks, padding_token):
    """`procyon/training/train_utils.py:1094-1101`."""
    num_pads = (text_toks == padding_token).sum(dim=-1)
    return (torch.full_like(num_pads, text_toks.shape[1]) - num_pads) - 2


def get_qa_scores(model_out, padding_token=None, answer_token=None):
    """`procyon/training/train_utils.py:1048-1070`: predicted token at the position before the label (causal shift) and
    the label token.  The engine's `forward` returns the logits of exactly that row (`[B,1,V]`, position
    `answer_positions` = label index - 1) instead of `[B,T,V]`; full-width logits are read the reference's way."""
    y_tok_total = model_out["text_toks"].detach().clone().cpu()
    if padding_token is not None:
        y_inds = get_final_tokens(y_tok_total, padding_token=padding_token)
    elif answer_token is not None:
        y_inds = get_after_answer_tokens(y_tok_total, answer_token=answer_token)
    else:
        raise ValueError("One of padding_token or answer_token for get_qa_metrics must not be None")
    rows = torch.arange(y_tok_total.shape[0])
    y_toks = y_tok_total[rows, y_inds]
    logits = model_out["outputs"].logits
    preds = logits.softmax(dim=-1).detach().clone().cpu()
    pred_total = preds.argmax(dim=-1)
    if pred_total.shape[1] == 1:
        pos = model_out["answer_positions"].cpu()
        if not torch.equal(pos, y_inds - 1):
            raise ValueError("the engine returned logits at rows other than the label positions - 1")
        pred_toks = pred_total[:, 0]
    else:
        pred_toks = pred_total[rows, y_inds - 1]
    return pred_toks.detach().clone().cpu(), y_toks.detach().clone().cpu()


class ProcyonCaptionEval:
    """`ProcyonCaptionEval` (procyon.py:40-111): diverse-beam captions, the first beam of every group is kept."""

    def __init__(self, model, model_config, caption_max_len, device=None):
        self.model = model
        self.device = device or model.device
        self.max_len = caption_max_len
        self.method = model_config.get("generation_method", "beam")
        self.num_captions = model_config.get("num_captions", 5)
        self.beam_group_size = model_config.get("beam_group_size", 2)
        self.beam_size = model_config.get("beam_size", self.num_captions * self.beam_group_size)

    @torch.no_grad()
    def get_predictions(self, data_loader):
        import pandas as pd
        aaseq_indices, generated = [], []
        for model_inputs in data_loader:
            model_inputs = move_inputs_to_device(model_inputs, self.device)
            _, _, _, captions = self.model.generate(model_inputs, max_len=self.max_len, aaseq_type=data_loader.dataset.aaseq_type,
                                                    return_all_internals=False, method=self.method, beam_size=self.beam_size,
                                                    beam_group_size=self.beam_group_size, truncate_on_eos=True)
            for i, indices in enumerate(model_inputs["reference_indices"]["input"]["seq"]):
                for j in range(self.num_captions):
                    aaseq_indices.append(indices[-1])
                    generated.append(captions[i][j * self.beam_group_size])
        return pd.DataFrame({"seq_id": aaseq_indices, "generated_caption": generated})


class ProcyonQAEval:
    """`ProcyonQAEval` (procyon.py:114-205)."""

    def __init__(self, model, qa_num_samples=None, seed=42, device=None):
        self.model = model
        self.device = device or model.device
        self.num_samples = qa_num_samples
        self.rng = np.random.default_rng(seed=seed)
        self.yes_token = model.yes_token
        self.no_token = model.no_token

    @torch.no_grad()
    def get_predictions(self, data_loader, aaseq_type="protein"):
        results = defaultdict(list)
        samples_to_hit = None
        if self.num_samples is not None and self.num_samples < len(data_loader):
            samples_to_hit = set(self.rng.choice(np.arange(len(data_loader)), size=self.num_samples, replace=False))
        no_context_aug = data_loader.collate_fn._get_input_contexts([], []) is None
        query_text_idx = -1 if no_context_aug else -2
        for i, model_inputs in enumerate(data_loader):
            if samples_to_hit is not None and i not in samples_to_hit:
                continue
            out = self.model(move_inputs_to_device(model_inputs, self.device), return_mlm=False, retrieval=False,
                             get_full_labels=True, aaseq_type=aaseq_type, crop_off=True)
            seq_ids = [x[-1] for x in model_inputs["reference_indices"]["input"]["seq"]]
            text_ids = [x[query_text_idx] for x in model_inputs["reference_indices"]["input"]["text"]]
            y_toks = torch.LongTensor([(self.yes_token if y == "yes" else self.no_token) for y in model_inputs["target"]["text"]])
            pred_toks, _ = get_qa_scores(out, answer_token=self.model.answer_idx)
            results["seq_ids"].extend(seq_ids)
            results["text_ids"].extend(text_ids)
            results["pred"].append(pred_toks)
            results["y"].append(y_toks)
        results["pred"] = torch.cat(results["pred"])
        results["y"] = torch.cat(results["y"])
        return results


class ProcyonRetrievalEval:
    """`ProcyonRetrievalEval` (procyon.py:208-406) without the on-disk embedding cache (:324-376)."""

    def __init__(self, model, device=None, query_is_sequence=False):
        self.model = model
        self.device = device or model.device
        self.is_ppi = query_is_sequence     # AASeqDataset queries (PPI) keep their id under input.seq, text queries under input.text

    @torch.no_grad()
    def _get_query_embeddings(self, query_loader, query_order):
        embs, query_ids = [], []
        for model_inputs in query_loader:
            model_inputs["target"]["seq"] = None
            model_inputs = move_inputs_to_device(model_inputs, self.device)
            key = "seq" if self.is_ppi else "text"
            query_ids += [x[-1] for x in model_inputs["reference_indices"]["input"][key]]
            out = self.model(model_inputs, retrieval=True, aaseq_type=query_loader.dataset.aaseq_type)
            embs.append(out["contrastive_out"]["positive"]["text"].detach().clone())
        idx = {q: i for i, q in enumerate(query_ids)}        # the LAST occurrence of a query wins (procyon.py:280-288)
        return torch.cat(embs, dim=0)[[idx[q] for q in query_order]]

    @torch.no_grad()
    def _calculate_target_embeddings(self, target_loader, collate_fn, aaseq_type="protein"):
        embs, target_ids = [], []
        for protein_ids in target_loader:
            model_inputs = protein_ids if self.model.config.use_aaseq_embeddings else collate_fn._convert_batch("sequence", protein_ids)
            model_inputs = move_inputs_to_device(model_inputs, self.device)
            target_ids += protein_ids.tolist()
            embs.append(self.model.forward_sequences(model_inputs, aaseq_type=aaseq_type)["shared"].detach().clone())
        return torch.cat(embs, dim=0), target_ids

    @torch.no_grad()
    def get_predictions(self, query_loader, target_loader, query_order, target_order):
        q = self._get_query_embeddings(query_loader, query_order)
        t, target_ids = self._calculate_target_embeddings(target_loader, query_loader.collate_fn, query_loader.dataset.aaseq_type)
        idx = {tid: i for i, tid in enumerate(target_ids)}
        t = t[[idx[tid] for tid in target_order]]
        sims = Context.get().retrieval_scores(q.to(self.device).contiguous(), t.to(self.device).contiguous())
        return sims.detach().cpu().to(torch.float64)
