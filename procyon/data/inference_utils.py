"""`procyon.data.inference_utils` (reference: procyon/data/inference_utils.py): the helpers the notebooks and scripts use to
turn (protein ids, free text, a task) into the nested `inputs` dict of `UnifiedProCyon.forward / generate`, to read QA
probabilities, and to rank proteins for a query embedding.

Written from the interface (argument names, defaults, returned structures, exceptions); one shared builder instead of the
reference's three near-identical functions.  Differences a caller can see:
  * nothing is read at import: `UNIPROT_IDS`, `functional_descriptions`, `DRUGMASK` are loaded from $DATA_DIR on first use
    (the reference reads three files while importing, :42-50,654-656);
  * `get_proteins_from_embedding` ranks on the embeddings' device through the engine when a GPU is present (normalise +
    GEMM + top-k in HIP kernels) instead of the hard-coded `.cuda()` matmul + full argsort (:955-978, quirk Q14);
  * column subsets come from `procyon.data.constants` (a JSON export of the reference's tables, or every description column).
"""
from __future__ import annotations

import json
import math
import os
from collections.abc import Callable
from typing import Dict, List, Optional

import numpy as np
import pandas as pd
import torch

from procyon.data.constants import CAPTION_SUBSETS, QA_SUBSETS, RETRIEVAL_SUBSETS
from procyon.data.data_utils import get_text_sequences_compositions, require_data_dir
from procyon.data.instruct_tune.instruct_constructor import get_prompt, get_prompt_open_def
from procyon.data.it_collator import construct_task_id
from procyon.evaluate.framework.utils import move_inputs_to_device
from procyon.training.train_utils import get_after_answer_tokens, get_final_tokens

# ---------------------------------------------------------------------------------------------- lazily loaded data tables
_LAZY = {}


def _protein_info():
    if "UNIPROT_IDS" not in _LAZY:
        _LAZY["UNIPROT_IDS"] = pd.read_pickle(os.path.join(require_data_dir(), "integrated_data/v1/protein/", "protein_info_filtered.pkl"))[
            ["index", "protein_id", "name"]]
    return _LAZY["UNIPROT_IDS"]


def _functional_descriptions():
    if "functional_descriptions" not in _LAZY:
        _LAZY["functional_descriptions"] = pd.read_pickle(os.path.join(
            require_data_dir(), "integrated_data/v1/protein/uniprot_functional_descriptions.pkl")).sort_values("index", axis=0)["function"]
    return _LAZY["functional_descriptions"]


def _drugmask():
    if "DRUGMASK" not in _LAZY:
        _LAZY["DRUGMASK"] = torch.load(os.path.join(require_data_dir(), "integrated_data/v1/drugbank/drugbank_mask.pt"), weights_only=False)
    return _LAZY["DRUGMASK"]


def __getattr__(name):   # `from procyon.data.inference_utils import UNIPROT_IDS` keeps working, loaded on demand
    if name == "UNIPROT_IDS":
        return _protein_info()
    if name == "functional_descriptions":
        return _functional_descriptions()
    if name == "DRUGMASK":
        return _drugmask()
    raise AttributeError(name)


def uniprot_id_to_index(uniprot_id: str) -> int:
    ids = _protein_info()
    hit = ids["protein_id"] == uniprot_id
    assert hit.sum() == 1, "ID {} not found in internal database".format(uniprot_id)
    return ids["index"].loc[hit].item()


def index_to_uniprot_id(i: int) -> str:
    ids = _protein_info()
    return ids["protein_id"].loc[ids["index"] == i].item()


# ---------------------------------------------------------------------------------------------- input construction
def _task_template(aaseq_type, dataset, relation, task_type):
    from procyon.data import data_utils
    home = os.getenv("HOME_DIR") or data_utils.HOME_DIR
    path = os.path.join(home, "procyon", "data", "instruct_tune", "tasks", construct_task_id(aaseq_type, dataset, relation, task_type) + ".json")
    with open(path, "r") as f:
        return json.load(f)


def _instruction(task_json, dataset, aaseq_type, icl_example_number, task_definition):
    kw = dict(task=task_json, num_examples=icl_example_number, is_special_definition=False, is_ppi=(dataset == "protein"),
              aaseq_type=aaseq_type)
    if task_definition is None:
        instruction, _, _, text_ids, aaseq_ids = get_prompt(**kw)
    else:
        instruction, _, _, _, text_ids, aaseq_ids = get_prompt_open_def(**kw)
        instruction = instruction.format(definition=task_definition)
    return instruction, text_ids, aaseq_ids


def _text_table(dataset, column_subset):
    df = pd.read_pickle(os.path.join(require_data_dir(), "integrated_data", "v1", dataset, f"{dataset}_info_filtered_composed.pkl"))
    return get_text_sequences_compositions(text_type=dataset, text_info=df, column_subset=column_subset)


def _first_present(table, row):
    """first non-missing description of a row (the reference's `.loc[notna()].tolist()[0]`, :350-355,752-757)"""
    r = table.iloc[row, :]
    return r.loc[r.notna()].tolist()[0]


def _model_input(aaseq_indices, descriptions, instruction, drug_inputs=None, input_drug=None):
    n = None if aaseq_indices is None else aaseq_indices.shape[0]
    return {
        "data": {"seq": aaseq_indices, "seq_idx": aaseq_indices, "text": descriptions, "drug": drug_inputs},
        "input": {"seq": None if n is None else [list(range(n))], "text": [list(range(len(descriptions)))], "drug": input_drug},
        "target": {"seq": None, "text": None, "drug": None},
        "instructions": [instruction],
    }


def _trim_ext(instruction):
    return instruction[:-5] if instruction[-5:] == "[EXT]" else instruction


def create_caption_input_simple(input_aaseq_ids: List[int], data_args, input_description: str = None, drug_inputs: List[int] = None,
                                task_definition: str = None, instruction_source_dataset: str = None,
                                instruction_source_relation: str = "all", aaseq_type: str = "protein", task_type: str = "caption",
                                icl_example_number: int = 1, device=None, disease_context_augmentation=False) -> Dict:
    """Inputs for phenotype generation about the proteins `input_aaseq_ids` (inference_utils.py:67-244): the dataset's caption
    template with `icl_example_number` worked examples; the query's trailing [EXT] is cut so that the prompt ends in
    "[ANSWER] " and the model continues from there."""
    assert drug_inputs is None
    if instruction_source_dataset is None:
        raise NotImplementedError
    dataset = instruction_source_dataset.lower()
    instruction, text_ids, aaseq_ids = _instruction(_task_template(aaseq_type, dataset, instruction_source_relation, "caption"),
                                                    dataset, aaseq_type, icl_example_number, task_definition)
    subset = None
    if task_type == "qa" and data_args.qa_subset_version is not None:
        subset = QA_SUBSETS[data_args.qa_subset_version]
    elif task_type == "caption" and data_args.caption_subset_version is not None:
        subset = CAPTION_SUBSETS[data_args.caption_subset_version]
    descriptions = _text_table(dataset, subset).iloc[text_ids, 0].tolist()
    indices = torch.LongTensor(aaseq_ids + input_aaseq_ids).to(device)
    if input_description is not None:
        descriptions = descriptions + [input_description]
    instruction = _trim_ext(instruction)
    if disease_context_augmentation:
        instruction = instruction.replace("[CONTEXT]", "[EXT]")
        contexts = _functional_descriptions().iloc[indices.cpu().numpy()].tolist()
        woven = []
        for i, c in enumerate(contexts):       # context of every protein, the examples' descriptions in between
            woven.append("Context: {}".format(c))
            if i != len(contexts) - 1:
                woven.append(descriptions[i])
        descriptions = woven
    else:
        instruction = instruction.replace("[CONTEXT]", "")
    return _model_input(indices, descriptions, instruction)


def create_qa_input_simple(input_aaseq_ids: List[int], data_args, input_description: str, drug_inputs: List[int] = None,
                           task_definition: str = None, instruction_source_dataset: str = None,
                           instruction_source_relation: str = "all", aaseq_type: str = "protein", icl_example_number: int = 1,
                           device=None, disease_context_augmentation=False) -> Dict:
    """Inputs for a yes/no question "does `input_description` apply to protein `input_aaseq_ids`" (inference_utils.py:247-420):
    positive and negative worked examples, the answer slot of the query filled with "null"."""
    assert drug_inputs is None
    if instruction_source_dataset is None:
        raise NotImplementedError
    dataset = instruction_source_dataset.lower()
    instruction, text_ids, aaseq_ids = _instruction(_task_template(aaseq_type, dataset, instruction_source_relation, "qa"),
                                                    dataset, aaseq_type, icl_example_number, task_definition)
    table = _text_table(dataset, QA_SUBSETS[data_args.qa_subset_version])
    descriptions = [_first_present(table, t) for t in text_ids]
    indices = torch.LongTensor(aaseq_ids + input_aaseq_ids).to(device)
    if input_description is not None:
        descriptions = descriptions + [input_description]
    instruction = _trim_ext(instruction)
    if disease_context_augmentation:
        instruction = instruction.replace("[CONTEXT]", "[EXT]")
        contexts = _functional_descriptions().iloc[indices.cpu().numpy()].tolist()
        woven = []
        for c, d in zip(contexts, descriptions):
            woven += [d, "Context: {}".format(c)]
        descriptions = woven
    else:
        instruction = instruction.replace("[CONTEXT]", "")
    instruction = instruction.format(answer="null")
    return _model_input(indices, descriptions, instruction)


def create_input_retrieval(input_description: str, data_args, instruction_source_dataset: str, drug_input_idx: List[int] = None,
                           task_definition: str = None, instruction_source_relation: str = "all", aaseq_type: str = "protein",
                           icl_example_number: int = 1) -> Dict:
    """Inputs for "which proteins match `input_description`" (inference_utils.py:663-843): the retrieval template, whose query
    ends in [PROT]; DrugBank descriptions gain a "<|drug|>" slot where a structure embedding exists."""
    dataset = instruction_source_dataset.lower()
    instruction, text_ids, aaseq_ids = _instruction(_task_template(aaseq_type, dataset, instruction_source_relation, "retrieval"),
                                                    dataset, aaseq_type, icl_example_number, task_definition)
    table = _text_table(dataset, RETRIEVAL_SUBSETS[data_args.retrieval_subset_version])
    descriptions = [_first_present(table, t) for t in text_ids]
    input_drug = drug_inputs = None
    if drug_input_idx is not None:
        mask = _drugmask()
        input_drug, drug_inputs = [], []
        if dataset == "drugbank":
            for i, t in enumerate(text_ids):
                if mask[t]:
                    descriptions[i] = descriptions[i] + "\nDrug: <|drug|>"
                    input_drug.append(i)
                    drug_inputs.append(t)
        if mask[drug_input_idx].item():
            input_description = input_description + "\nDrug: <|drug|>"
            input_drug.append(len(input_drug))
            drug_inputs.append(drug_input_idx)
            input_drug = torch.LongTensor(input_drug).unsqueeze(0).tolist()
            drug_inputs = torch.LongTensor(drug_inputs)
        else:
            print("WARNING: not inserting drug index because we don't have one in our database")
            input_drug = drug_inputs = None
    indices = torch.LongTensor(aaseq_ids) if len(aaseq_ids) > 0 else None
    instruction = _trim_ext(instruction).replace("[CONTEXT]", "")
    return _model_input(indices, descriptions + [input_description], instruction, drug_inputs=drug_inputs, input_drug=input_drug)


def merge_model_input_dicts(dict_list: List[Dict]) -> Dict:
    """Several single-prompt inputs -> one batch (inference_utils.py:847-882): `data` entries are concatenated, the slot
    lists of every further prompt are shifted behind what is already there."""
    merged = dict_list[0]
    for d in dict_list[1:]:
        for k, v in d["data"].items():
            if v is None:
                continue
            merged["data"][k] = merged["data"][k] + v if isinstance(v, list) else torch.cat([merged["data"][k], v])
        for k, v in d["input"].items():
            if v is None:
                continue
            base = int(np.max(merged["input"][k])) + 1
            merged["input"][k] += [[val + base for val in v[0]]]
        merged["instructions"] += d["instructions"]
    return merged


def create_batched_input_retrieval(input_descriptions: List[str], data_args, task_definitions: List[str] = None,
                                   instruction_source_dataset: str = None, instruction_source_relation: str = "all",
                                   aaseq_type: str = "protein", icl_example_number: int = 1) -> Dict:
    n = len(input_descriptions)
    if task_definitions is not None:
        assert len(task_definitions) == n
    else:
        task_definitions = [None] * n
    return merge_model_input_dicts([
        create_input_retrieval(input_description=input_descriptions[i], data_args=data_args, task_definition=task_definitions[i],
                               instruction_source_dataset=instruction_source_dataset,
                               instruction_source_relation=instruction_source_relation, aaseq_type=aaseq_type,
                               icl_example_number=icl_example_number) for i in range(n)])


# ---------------------------------------------------------------------------------------------- QA read-out
def get_qa_logits_inference(model_out, padding_token=None, answer_token=None):
    """Vocabulary distribution at the answer position and the label token there (inference_utils.py:582-604).  The answer
    position is (index after the last [ANSWER]) - 1, the causal shift; the engine's `forward` has computed the logits of exactly
    that row (`outputs.answer_logits` [B,1,V], `answer_positions`), a full [B,T,V] tensor is indexed the reference's way."""
    toks = model_out["text_toks"].detach().clone().cpu()
    if padding_token is not None:
        inds = get_final_tokens(toks, padding_token=padding_token)
    elif answer_token is not None:
        inds = get_after_answer_tokens(toks, answer_token=answer_token)
    else:
        raise ValueError("One of padding_token or answer_token for get_qa_metrics must not be None")
    rows = torch.arange(toks.shape[0])
    y_toks = toks[rows, inds]
    outputs = model_out["outputs"]
    ans = getattr(outputs, "answer_logits", None)
    if ans is not None and torch.equal(model_out["answer_positions"].cpu(), inds - 1):
        # the engine computed exactly the rows the reader wants: softmax over the vocabulary on the device (pcy_qa_probs)
        from procyon_amd.engine import Context
        pred_toks = Context.get().qa_probs(ans[:, 0])[0].cpu()
    else:
        preds = outputs.logits.softmax(dim=-1).detach().clone().cpu()
        pred_toks = preds[rows, inds - 1]
    return pred_toks.detach().clone().cpu(), y_toks.detach().clone().cpu()


class ProCyonQAInference:
    """Callable QA wrapper (inference_utils.py:607-651): `wrapper(model_inputs)["pred"]` = vocabulary probabilities at the
    answer position; `.yes_token` / `.no_token` index them."""

    def __init__(self, model, device=None):
        model.eval()
        self.device = device
        self.model = model.to(self.device)
        self.yes_token = model.yes_token
        self.no_token = model.no_token

    @torch.no_grad()
    def __call__(self, *args, **kwargs):
        return self.fwd_pass(*args, **kwargs)

    def fwd_pass(self, model_inputs, aaseq_type: str = "protein", output_attentions=None):
        out = self.model(move_inputs_to_device(model_inputs, self.device), return_mlm=False, retrieval=False, get_full_labels=True,
                         aaseq_type=aaseq_type, crop_off=True, output_attentions=output_attentions)
        pred, _ = get_qa_logits_inference(out, answer_token=self.model.answer_idx)
        return {"pred": pred, "y": None, "out": out}


# ---------------------------------------------------------------------------------------------- retrieval ranking
def _rank(query, targets, k):
    """(indices [Q,k], scores [Q,k]) of the k most similar rows of `targets` per query, best first; k None = all.  fp32 similarities
    (fp32 accumulation and result, `pcy_retrieval_topk_f32`) ranked on the device in a stable descending order -- the reference
    computes in the embeddings' dtype and its cached target matrix is fp32; bf16 cosines would tie by the hundred among 18k
    proteins.  No CPU path: `Context.get()` raises without a device, like the reference's hard-coded `.cuda()`."""
    from procyon_amd.engine import Context
    idx, sc = Context.get().retrieval_topk_f32(query, targets, k)
    return idx.cpu(), sc.cpu()


def get_proteins_from_embedding(protein_embeds: torch.Tensor, model_out: Optional[Dict] = None,
                                query_embeddings: Optional[torch.Tensor] = None, protein_ids: pd.DataFrame = None,
                                top_k: int = 20) -> pd.DataFrame:
    """Top retrieval hits as a data frame with columns uniprot_id, name, sim_score (inference_utils.py:921-999); top_k=None
    returns every protein, ranked.  `protein_ids` defaults to the ProCyon-Instruct protein table."""
    assert model_out is not None or query_embeddings is not None
    if model_out is None:
        assert query_embeddings is not None
    else:
        assert query_embeddings is None
        query_embeddings = model_out["contrastive_out"]["positive"]["text"][0, :].unsqueeze(0).detach().clone()
    if protein_ids is None:
        protein_ids = _protein_info()
    order, scores = _rank(query_embeddings.reshape(1, -1) if query_embeddings.dim() == 1 else query_embeddings[:1], protein_embeds, top_k)
    order, scores = order[0].tolist(), scores[0].tolist()
    return pd.DataFrame({"uniprot_id": protein_ids["protein_id"].iloc[order], "name": protein_ids["name"].iloc[order], "sim_score": scores})


@torch.no_grad()
def get_proteins_from_batched_embeddings(protein_embeds: torch.Tensor, query_embeddings: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Q, N] fp32 cosine similarities on the CPU (inference_utils.py:981-999)."""
    assert query_embeddings is not None
    from procyon_amd.engine import Context
    sims = Context.get().retrieval_scores_f32(query_embeddings if query_embeddings.dim() == 2 else query_embeddings[None], protein_embeds)
    return sims.squeeze().detach().cpu().float()


# ---------------------------------------------------------------------------------------------- description perturbation
def perturb_by_words(sentence: str, generator: np.random.Generator, perturbation_pct: float = 0.1) -> str:
    """drop a random `perturbation_pct` of the words, keeping the order of the rest (inference_utils.py:1001-1016)"""
    words = sentence.split()
    keep = set(generator.choice(np.arange(len(words)), size=math.floor((1 - perturbation_pct) * len(words)), replace=False))
    return " ".join(w for i, w in enumerate(words) if i in keep)


def desc_perturbation(desc: str, query_func: Callable, num_perturbations: int = 10, perturbation_pct: float = 0.1,
                      seed: Optional[float] = None) -> Dict:
    """`query_func` on `num_perturbations` word-dropped variants of `desc` -> {"perturb_i": result} (inference_utils.py:1019-1038)."""
    generator = np.random.default_rng(seed)
    return {f"perturb_{i}": query_func(perturb_by_words(desc, generator=generator, perturbation_pct=perturbation_pct))
            for i in range(num_perturbations)}
