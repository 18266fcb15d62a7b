"""Eval-framework plugin level of the drop-in boundary: the three plugins of
`procyon/evaluate/framework/procyon.py` on top of the engine-backed `UnifiedProCyon`, and the QA reader they use.

Constructors follow the registry contract -- `model_zoo[task][model_type](model_config, eval_args, model_args, device)`
(`procyon/evaluate/framework/core.py:210-216`): each loads its checkpoint through `UnifiedProCyon.from_pretrained(
checkpoint_dir=model_config["checkpoint_dir"])`, warns about `ModelArgs` mismatches, and puts the model in eval / bf16 mode
(`procyon.py:49-77,114-140,208-240`).  `from_model(...)` builds the same object around a ready model (tests, benches).
Every `get_predictions` body follows the reference: same loader protocol (batches are the collator dictionaries with
`reference_indices`; `data_loader.dataset.aaseq_type`; `data_loader.collate_fn._get_input_contexts` / `._convert_batch`),
same return types, incl. the on-disk target-embedding cache of the retrieval plugin (`procyon.py:324-376`).  Retrieval
scoring runs on the device (`pcy_retrieval_scores`) and returns the reference's float64 CPU matrix.

Engine-specific, optional `model_config` keys (absent in the reference's YAML, ignored by it): "tokenizer" (a tokenizer object
for boxes without Llama tokenizer files), "pretrained_weights_dir", "engine_kwargs" (dict forwarded to `from_pretrained`)."""
from collections import defaultdict
from collections.abc import Mapping

import dataclasses
import os

import numpy as np
import torch

from .engine import Context


def default_pretrained_weights_dir():
    """`DEFAULT_PRETRAINED_WEIGHTS_DIR` = f"{DATA_DIR}/model_weights/" (model_unified.py:33); None without $DATA_DIR (the
    reference asserts at import; here only a checkpoint that needs files from it fails, with the path in the message)."""
    d = os.getenv("DATA_DIR")
    return os.path.join(d, "model_weights/") if d else None


def compare_and_warn_model_args(model_args_a, model_args_b):
    """`compare_and_warn_model_args` (procyon/evaluate/framework/utils.py:103-141): print the fields in which the ModelArgs given
    for evaluation differ from the checkpoint's; `n_model_pieces`, `model_splitting` and every `*path` field are expected to
    differ.  Returns the mismatches (the reference returns None; callers ignore it)."""
    if model_args_a is None or model_args_b is None:
        return []
    ignore = {"n_model_pieces", "model_splitting"}
    names = [f.name for f in dataclasses.fields(model_args_a)] if dataclasses.is_dataclass(model_args_a) else sorted(vars(model_args_a))
    miss = object()
    mismatches = []
    for n in names:
        if n in ignore or n.endswith("path"):
            continue
        a, b = getattr(model_args_a, n, miss), getattr(model_args_b, n, miss)
        if a is miss or b is miss:      # a field one side does not declare (shell classes carry only what was pickled)
            continue
        if a != b:
            mismatches.append((n, a, b))
    if mismatches:
        print("Specified ModelArgs do not match those used in provided ProCyon checkpoint "
              "this may cause crashes or unexpected behavior. Use the EvalArgs.model_args_from_checkpoint "
              "command-line argument unless you're sure you want to do this, comment this out.\n"
              f"Mismatched fields: {' , '.join(f'{x[0]}: {x[1]} != {x[2]}' for x in mismatches)}")
    return mismatches


def _load_for_eval(model_config, model_args, device, **from_pretrained_kw):
    """The head of the three reference constructors (procyon.py:55-67,128-139,216-228): checkpoint -> model in eval / bf16
    mode on `device`."""
    from .model import UnifiedProCyon
    checkpoint_dir = model_config["checkpoint_dir"]
    kw = dict(model_config.get("engine_kwargs") or {})
    if model_config.get("tokenizer") is not None:
        kw["tokenizer"] = model_config["tokenizer"]
    kw.update(from_pretrained_kw)
    dev = torch.device(device)
    model, checkpoint_model_args = UnifiedProCyon.from_pretrained(
        pretrained_weights_dir=model_config.get("pretrained_weights_dir") or default_pretrained_weights_dir(),
        checkpoint_dir=checkpoint_dir, device=dev, **kw)
    compare_and_warn_model_args(model_args, checkpoint_model_args)
    model.eval()
    model.bfloat16()
    return model.to(dev), checkpoint_dir


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
    the label token.  The engine's `forward` has computed the logits of exactly that row (`outputs.answer_logits` [B,1,V], position
    `answer_positions` = label index - 1); full-width logits are read the reference's way."""
    y_tok_total = model_out["text_toks"].detach().clone().cpu()
    if padding_token is not None:
        y_inds = get_final_tokens(y_tok_total, padding_token=padding_token)
    elif answer_token is not None:
        y_inds = get_after_answer_tokens(y_tok_total, answer_token=answer_token)
    else:
        raise ValueError("One of padding_token or answer_token for get_qa_metrics must not be None")
    rows = torch.arange(y_tok_total.shape[0])
    y_toks = y_tok_total[rows, y_inds]
    outputs = model_out["outputs"]
    ans = getattr(outputs, "answer_logits", None)
    if ans is not None:
        pos = model_out["answer_positions"].cpu()
        if not torch.equal(pos, y_inds - 1):
            raise ValueError("the engine returned logits at rows other than the label positions - 1")
        # softmax over the vocabulary + argmax of the stored probabilities on the device (pcy_qa_probs)
        from .engine import Context
        pred_toks = Context.get().qa_probs(ans[:, 0], want_probs=False, want_argmax=True)[2].cpu()
    else:
        preds = outputs.logits.softmax(dim=-1).detach().clone().cpu()
        pred_toks = preds.argmax(dim=-1)[rows, y_inds - 1]
    return pred_toks.detach().clone().cpu(), y_toks.detach().clone().cpu()


class ProcyonCaptionEval:
    """`ProcyonCaptionEval` (procyon.py:40-111): diverse-beam captions, the first beam of every group is kept."""

    def __init__(self, model_config, eval_args, model_args, device):
        model, self.checkpoint_dir = _load_for_eval(model_config, model_args, device)
        self._setup(model, model_config, eval_args, model_args, device)

    @classmethod
    def from_model(cls, model, model_config, eval_args, model_args=None, device=None):
        """the same plugin around a ready model (no checkpoint directory)"""
        self = cls.__new__(cls)
        self.checkpoint_dir = model_config.get("checkpoint_dir")
        self._setup(model.eval().bfloat16(), model_config, eval_args, model_args, device or model.device)
        return self

    def _setup(self, model, model_config, eval_args, model_args, device):
        self.device = device
        self.model = model
        self.model_args = model_args
        self.max_len = eval_args.caption_max_len
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

    def __init__(self, model_config, eval_args, model_args, device):
        model, self.checkpoint_dir = _load_for_eval(model_config, model_args, device)
        self._setup(model, eval_args, model_args, device)

    @classmethod
    def from_model(cls, model, model_config, eval_args, model_args=None, device=None):
        self = cls.__new__(cls)
        self.checkpoint_dir = (model_config or {}).get("checkpoint_dir")
        self._setup(model.eval().bfloat16(), eval_args, model_args, device or model.device)
        return self

    def _setup(self, model, eval_args, model_args, device):
        self.device = device
        self.model = model
        self.model_args = model_args
        self.num_samples = eval_args.qa_num_samples
        self.rng = np.random.default_rng(seed=eval_args.seed)
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
    """`ProcyonRetrievalEval` (procyon.py:208-406), incl. the on-disk target-embedding cache (:324-376)."""

    def __init__(self, model_config, eval_args, model_args, device):
        # strict_load=False: "we don't store non-tuned weights" (procyon.py:219)
        model, self.checkpoint_dir = _load_for_eval(model_config, model_args, device, strict_load=False)
        self._setup(model, eval_args, model_args, device)

    @classmethod
    def from_model(cls, model, model_config, eval_args, model_args=None, device=None):
        self = cls.__new__(cls)
        self.checkpoint_dir = (model_config or {}).get("checkpoint_dir")
        self._setup(model.eval().bfloat16(), eval_args, model_args, device or model.device)
        return self

    def _setup(self, model, eval_args, model_args, device):
        self.device = device
        self.model = model
        self.model_args = model_args
        self.batch_size = eval_args.batch_size
        self.use_cached_target_embeddings = getattr(eval_args, "retrieval_use_cached_target_embeddings", False)

    @staticmethod
    def _query_is_sequence(dataset):
        """text queries keep their id under input.text, AASeqDataset (PPI) queries under input.seq (procyon.py:241-246,268-276).
        The dataset classes are the reference's (not importable here): they are told apart by walking the class's MRO by NAME --
        subclasses and wrappers that inherit from them are recognised as `isinstance` would -- AASeqTextUnifiedDataset first, as the
        reference checks it first.  A stand-in loader sets the `query_is_sequence` attribute instead.  Anything else raises the
        reference's ValueError."""
        if hasattr(dataset, "query_is_sequence"):
            return bool(dataset.query_is_sequence)
        names = [k.__name__ for k in type(dataset).__mro__]
        if "AASeqTextUnifiedDataset" in names:
            return False
        if "AASeqDataset" in names:
            return True
        raise ValueError(f"unexpected dataset type: {type(dataset)}")

    @torch.no_grad()
    def _get_query_embeddings(self, query_loader, query_order):
        is_ppi = self._query_is_sequence(query_loader.dataset)
        embs, query_ids = [], []
        for model_inputs in query_loader:
            model_inputs["target"]["seq"] = None
            model_inputs = move_inputs_to_device(model_inputs, self.device)
            key = "seq" if is_ppi else "text"
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
            embs.append(self.model.forward_sequences(model_inputs, aaseq_type=aaseq_type)["shared"].detach().clone().cpu())
        return torch.cat(embs, dim=0), target_ids

    def _all_target_ids(self, aaseq_type):
        """`get_retrieval_target_set(None, {}, EvalArgs(retrieval_eval_all_aaseqs=True), aaseq_type)` (retrieval.py:85-97): the index
        of the ProCyon-Instruct entity table."""
        import pandas as pd
        from procyon.data.data_utils import require_data_dir
        if aaseq_type not in ("protein", "domain"):
            raise ValueError(f"unknown aaseq type: {aaseq_type}")
        f = os.path.join(require_data_dir(), f"integrated_data/v1/{aaseq_type}/{aaseq_type}_info_filtered.pkl")
        return pd.read_pickle(f).index.to_series()

    def _get_cached_target_embeddings(self, collate_fn, aaseq_type):
        """`_get_cached_target_embeddings` (procyon.py:324-355): `<checkpoint_dir>/<aaseq_type>_target_embeddings.pkl` holds
        (embeddings [N, D] CPU, ids); computed over ALL entities of that type and written when missing."""
        print("loading cached target embeddings")
        path = os.path.join(self.checkpoint_dir, f"{aaseq_type}_target_embeddings.pkl")
        if not os.path.exists(path):
            print("retrieval_use_cached_target_embeddings is set to True but cached "
                  f"embeddings not found, calculating and writing to: {path}")
            all_targets = torch.as_tensor(self._all_target_ids(aaseq_type).to_numpy())
            loader = [all_targets[i:i + self.batch_size] for i in range(0, len(all_targets), self.batch_size)]   # ProteinEvalDataset + DataLoader(shuffle=False)
            emb, ids = self._calculate_target_embeddings(loader, collate_fn, aaseq_type=aaseq_type)
            with open(path, "wb") as fh:
                torch.save((emb, ids), fh)
            return emb, ids
        return torch.load(path, map_location="cpu", weights_only=False)

    @torch.no_grad()
    def _get_target_embeddings(self, target_loader, target_order, collate_fn, aaseq_type):
        """`_get_target_embeddings` (procyon.py:357-376): cached or computed, then rearranged / subset to `target_order`."""
        if self.use_cached_target_embeddings:
            emb, ids = self._get_cached_target_embeddings(collate_fn, aaseq_type)
        else:
            emb, ids = self._calculate_target_embeddings(target_loader, collate_fn, aaseq_type=aaseq_type)
        idx = {tid: i for i, tid in enumerate(ids)}
        return emb[[idx[tid] for tid in target_order]]

    @torch.no_grad()
    def get_predictions(self, query_loader, target_loader, query_order, target_order):
        q = self._get_query_embeddings(query_loader, query_order)
        t = self._get_target_embeddings(target_loader, target_order, query_loader.collate_fn, query_loader.dataset.aaseq_type)
        # F.normalize + matmul in the embeddings' dtype (bf16 for the bf16 model), then float64 on the CPU (procyon.py:400-406)
        sims = Context.get().retrieval_scores(q.to(self.device, torch.bfloat16).contiguous(), t.to(self.device, torch.bfloat16).contiguous())
        return sims.detach().cpu().to(torch.float64)
