"""Bulk entry points of the reference's scripts/ directory over the engine:

  * `caption_bulk`        scripts/caption_bulk.py:62-148        one diverse-beam caption set per UniProt id, every-other beam kept,
                                                                a pickle of the running table every `save_frequency` proteins;
  * `qa_filter_captions`  scripts/qa_filter_captions.py:11-113  P(yes) / P(no) of "does this caption describe this protein" for
                                                                every (protein, response) pair.

Same inputs, outputs and file formats as the scripts.  What differs is the schedule: the scripts push ONE protein / ONE pair
through the model per call; here `batch_size` prompts share a call -- the prompts of one dataset have the same token length (a
protein is one soft token whatever its length), so a batch is a rectangular prefill with no padding, and beam search / the QA
read-out are per-row computations: the results are those of the one-by-one loop (tests/test_gpu_round3.py checks it).  A protein
that several prompts of a batch name (the in-context example) is embedded once.

The scripts never call `.bfloat16()` (they run the checkpoint's fp32 weights); the engine computes in bf16 only and refuses an
fp32 model, so these functions put the model in bf16 mode explicitly, as the evaluation framework and the service do."""
from __future__ import annotations

import math
import os
from typing import List, Optional

import numpy as np
import pandas as pd
import torch


def chunk_rows(n_rows: int, num_chunks: Optional[int], chunk_idx: Optional[int], script: str = "caption_bulk"):
    """[start, end) of this job's share of the table, as the scripts cut it (caption_bulk.py:80-92, qa_filter_captions.py:38-50).
    The two scripts close the LAST chunk by different tests -- caption_bulk by `chunk_idx >= ceil(n / num_chunks) - 1` (which,
    with more rows per chunk than chunks, never fires before the real last chunk only by accident), qa_filter_captions by
    `chunk_idx == num_chunks - 1`; both are reproduced."""
    if num_chunks is None or chunk_idx is None:
        return 0, n_rows
    per = math.ceil(n_rows / num_chunks)
    starts = np.arange(0, n_rows, per)
    start = int(starts[chunk_idx])
    last = chunk_idx >= (per - 1) if script == "caption_bulk" else chunk_idx == (num_chunks - 1)
    end = n_rows if last else int(starts[chunk_idx + 1])
    return start, end


def _merge_dedup(inputs: List[dict]) -> dict:
    """single-prompt model inputs -> one batch; a protein index named by several prompts is kept once in `data.seq`"""
    seqs = torch.cat([d["data"]["seq"].reshape(-1).cpu() for d in inputs])
    uniq, inv = torch.unique(seqs, sorted=False, return_inverse=True)
    # keep first-appearance order (torch.unique does not promise one)
    first = torch.full((uniq.numel(),), seqs.numel(), dtype=torch.long).scatter_reduce(0, inv, torch.arange(seqs.numel()), "amin")
    order = torch.argsort(first)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel())
    uniq, inv = uniq[order], rank[inv]
    out_seq, texts, in_text, p = [], [], [], 0
    for d in inputs:
        n = d["data"]["seq"].numel()
        out_seq.append([int(inv[p + j]) for j in d["input"]["seq"][0]])
        in_text.append([len(texts) + j for j in d["input"]["text"][0]])
        texts += list(d["data"]["text"])
        p += n
    return {"data": {"seq": uniq, "seq_idx": uniq, "text": texts, "drug": None},
            "input": {"seq": out_seq, "text": in_text, "drug": None},
            "target": {"seq": None, "text": None, "drug": None},
            "instructions": [d["instructions"][0] for d in inputs]}


@torch.no_grad()
def caption_bulk(model, model_args, data_args, uniprot_ids: List[str], prompt_dataset: str = "uniprot", prompt_relation: str = "all",
                 max_len: int = 200, beam_size: int = 10, diversity_penalty: float = 0.8, save_path: Optional[str] = None,
                 batch_size: int = 8, save_frequency: int = 5, device=None) -> pd.DataFrame:
    """-> DataFrame(uniprot_id, response0 .. response{beam_size // 2 - 1}); `save_path` receives the running table as a pickle
    whenever the one-by-one loop would have written it (after protein i for every i % save_frequency == 0, i > 0) and the final
    table as CSV (caption_bulk.py:99-148)."""
    from procyon.data.inference_utils import create_caption_input_simple, uniprot_id_to_index
    assert "drug" not in prompt_dataset, "DrugBank not supported in this script"
    model.eval()
    model.bfloat16()
    results = {"uniprot_id": []}
    results.update({f"response{i}": [] for i in range(beam_size // 2)})
    split_str = "<|end_of_text|>" if "llama-3" in getattr(model_args, "text_encoder_fname", "llama-3") else "</s>"
    n = len(uniprot_ids)
    for s in range(0, n, batch_size):
        ids = list(uniprot_ids[s:s + batch_size])
        inputs = [create_caption_input_simple(input_aaseq_ids=[uniprot_id_to_index(u)], data_args=data_args, input_description=None,
                                              drug_inputs=None, task_definition=None, instruction_source_dataset=prompt_dataset,
                                              instruction_source_relation=prompt_relation, aaseq_type="protein", task_type="caption",
                                              icl_example_number=1, device=device) for u in ids]
        _, _, _, out_text = model.generate(inputs=_merge_dedup(inputs), aaseq_type="protein", max_len=max_len, method="beam",
                                           return_all_internals=False, beam_size=beam_size, beam_group_size=2,
                                           diversity_penalty=diversity_penalty)
        for b, u in enumerate(ids):
            results["uniprot_id"].append(u)
            for j, t in enumerate(out_text[b]):
                if j % 2 == 1:      # beam groups of two: the first beam of every group
                    continue
                results[f"response{j // 2}"].append(t.split(split_str)[0])
            i = s + b
            if save_path is not None and i % save_frequency == 0 and i > 0:
                pd.DataFrame(results).to_pickle(save_path)
    df = pd.DataFrame(results)
    if save_path is not None:
        df.to_csv(save_path)
    return df


def load_caption_table(caption_fpath: Optional[str] = None, caption_dir: Optional[str] = None) -> pd.DataFrame:
    """the caption table(s) `qa_filter_captions.py` accepts (:22-36)"""
    if caption_fpath is not None:
        if caption_fpath.endswith("tsv.gz"):
            return pd.read_csv(caption_fpath, compression="gzip", sep="\t")
        if caption_fpath.endswith(".pickle") or caption_fpath.endswith(".pkl"):
            return pd.read_pickle(caption_fpath)
        if caption_fpath.endswith(".csv"):
            return pd.read_csv(caption_fpath)
        raise ValueError(f"unsupported caption file: {caption_fpath}")
    assert caption_dir is not None
    return pd.concat([pd.read_pickle(os.path.join(caption_dir, f)) for f in os.listdir(caption_dir)])


@torch.no_grad()
def qa_filter_captions(model, data_args, caption_df: pd.DataFrame, prompt_dataset: str = "uniprot", prompt_relation: str = "all",
                       save_path: Optional[str] = None, batch_size: int = 16, device=None) -> pd.DataFrame:
    """-> DataFrame(uniprot_id, response_num, caption_output, yes, no): the QA model's P(" yes") / P(" no") at the answer position
    for every response column of every row, in the scripts' order (rows outer, sorted response columns inner)."""
    from procyon.data.inference_utils import ProCyonQAInference, create_qa_input_simple, merge_model_input_dicts, uniprot_id_to_index
    model.eval()
    model.bfloat16()
    qa_model = ProCyonQAInference(model, device=device)
    pairs = []
    for i in range(caption_df.shape[0]):
        row = caption_df.iloc[i, :]
        for r in sorted(c for c in row.index if "response" in c):
            pairs.append((row["uniprot_id"], r, row[r]))
    scores = {"uniprot_id": [], "response_num": [], "caption_output": [], "yes": [], "no": []}
    for s in range(0, len(pairs), batch_size):
        chunk = pairs[s:s + batch_size]
        inputs = [create_qa_input_simple(input_aaseq_ids=[uniprot_id_to_index(u)], data_args=data_args, input_description=cap,
                                         drug_inputs=None, task_definition=None, instruction_source_dataset=prompt_dataset,
                                         instruction_source_relation=prompt_relation, aaseq_type="protein", icl_example_number=1,
                                         device=device) for u, _, cap in chunk]
        pred = qa_model(merge_model_input_dicts(inputs))["pred"]
        for b, (u, r, cap) in enumerate(chunk):
            scores["uniprot_id"].append(u)
            scores["response_num"].append(r)
            scores["caption_output"].append(cap)
            scores["yes"].append(pred[b, qa_model.yes_token].item())
            scores["no"].append(pred[b, qa_model.no_token].item())
    df = pd.DataFrame(scores)
    if save_path is not None:
        df.to_csv(save_path, index=False)
    return df
