"""Mirror of `UnifiedProCyon`'s inference API (/root/reference/procyon/model/model_unified.py) over the
MI355X engine: `forward` (QA / retrieval), `generate` (greedy / sampling / nucleus / diverse beam),
`forward_sequences`, plus the host-side text preparation that sits between them and the kernels.

Same method names, argument meaning, in-place side effects and exceptions as the reference, so the callers in
SURVEY.md section 8b (scripts/, evaluate/framework/procyon.py, inference/) can switch by import path
(INTEGRATION.md).  Training-only branches (contrastive loss, MLM, LoRA groups, freezing) are out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass
from itertools import chain
from types import SimpleNamespace
from typing import List, Optional

import os
import torch

from ..engine import BF16, MlpEngine
from .model_utils import left_pad_tensors


@dataclass
class ProCyonConfig:
    """The `ModelArgs` fields that shape the inference path (training_args_IT.py:27-651; shipped values
    configs/llama3-full.yml:29-68)."""
    text_encoder_fname: str = "llama-3-8b"
    use_aaseq_embeddings: bool = False
    protein_pooling_opt: str = "mean"
    protein_pooling_correction_option: bool = False
    long_protein_strategy: str = "split"
    max_protein_len: int = 1024
    max_text_len: int = 2048
    ret_token_access: str = "last"
    roll_num: int = 0
    use_protein_struct: bool = False
    protein_struct_dropout: float = 0.0
    use_drug_embeddings: bool = False
    protein_task_spc_lora: bool = False
    lora_specific_style: str = "none"


def mask_before(full_labels, answer_idx, before_last_answer=False):
    """`mask_before` (model_unified.py:39-60)."""
    answer_found = (full_labels == answer_idx).nonzero()
    if not before_last_answer:
        if torch.any((full_labels == answer_idx).sum(dim=1) > 1):
            raise ValueError('More than one {} token detected in an input'.format(answer_idx))
        found_map = answer_found
    else:
        found_map = torch.tensor([(i, int(answer_found[answer_found[:, 0] == i, 1].max()))
                                  for i in range(full_labels.shape[0])], device=full_labels.device)
    ar = torch.arange(full_labels.shape[1], device=full_labels.device).unsqueeze(0).repeat(full_labels.shape[0], 1)
    ind = found_map[:, 1].unsqueeze(1).repeat(1, full_labels.shape[1])
    return ind >= ar


def multi_replace_tokens(a, b, replace_token, eval=False):
    """`multi_replace_tokens` (model_unified.py:83-108): splice the token lists `b` over the occurrences of
    `replace_token` in `a` ([EXT] slots); eval=True leaves the last slot empty."""
    occ = [i for i, t in enumerate(a) if t == replace_token]
    if len(occ) != len(b):
        raise ValueError("Number of occurrences of replace_token does not match the length of b")
    if len(occ) == 0:
        return a
    result = a[:occ[0]]
    for i, o in enumerate(occ):
        if not (i == len(occ) - 1 and eval):
            result += b[i]
        result += a[o + 1:] if i == len(occ) - 1 else a[o + 1:occ[i + 1]]
    return result


def special_token_ids(tokenizer, text_encoder_fname):
    """The token ids `UnifiedProCyon.__init__` / `_init_tokenizer` keep (model_unified.py:1100-1133, 342-347), read from a
    tokenizer on which the eight ProCyon tokens are already registered (`procyon_amd.checkpoint.hf_tokenizer` does that in the
    reference's order; the reference reads them back with `tokenizer(tok, add_special_tokens=False).input_ids[0]`, which for an
    added token is its id).  Llama-3 names take the yes / no ids of " yes" / " no" (leading space), others of "yes" / "no"."""
    t = tokenizer
    ids = dict(prot_replacement_idx=t.convert_tokens_to_ids("<|protein|>"), prot_retrieval_idx=t.convert_tokens_to_ids("[PROT]"),
               answer_idx=t.convert_tokens_to_ids("[ANSWER]"), struct_idx=t.convert_tokens_to_ids("<|struct|>"),
               drug_idx=t.convert_tokens_to_ids("<|drug|>"), ext_idx=t.convert_tokens_to_ids("[EXT]"))
    if "llama-3" in text_encoder_fname.lower():  # model_unified.py:342-347
        ids["yes_token"] = t.encode(" yes", add_special_tokens=False)[0]
        ids["no_token"] = t.encode(" no", add_special_tokens=False)[0]
    else:
        ids["yes_token"] = t.encode("yes", add_special_tokens=False)[0]
        ids["no_token"] = t.encode("no", add_special_tokens=False)[0]
    return ids


class _LazyLogits:
    """`outputs.logits` of `UnifiedProCyon.forward` on the QA branch: behaves like the reference's [B, T, V] tensor
    (model_unified.py:548-554) for a caller that indexes it -- `logits[torch.arange(B), pos]`, `logits[:, pos]`, `logits.softmax(-1)`, any
    torch function -- but only the answer rows exist until something else is asked for: indexing exactly the answer positions returns them
    (the QA readers' access, data/inference_utils.py:582-604), anything else runs the prefill once more with every row's logits and caches
    the [B, T_real, V] tensor (columns beyond the last real token of the longest row are never computed; the reference pads to
    max_text_len).  `outputs.answer_logits` [B, 1, V] is the engine's own fast handle on the same rows."""

    def __init__(self, answer_rows, answer_pos, T, materialise):
        self._rows, self._pos, self._T, self._mk, self._full_t = answer_rows, answer_pos.long().cpu(), T, materialise, None

    def _full(self):
        if self._full_t is None:
            self._full_t = self._mk()
        return self._full_t

    @property
    def shape(self):
        return torch.Size((self._rows.shape[0], self._T, self._rows.shape[-1]))

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def dim(self):
        return 3

    @property
    def dtype(self):
        return self._rows.dtype

    @property
    def device(self):
        return self._rows.device

    def _is_answer_index(self, idx):
        if not (isinstance(idx, tuple) and len(idx) == 2):
            return False
        r, p = idx
        B = self._rows.shape[0]
        if isinstance(r, slice):
            if r != slice(None):
                return False
        else:
            r = torch.as_tensor(r).cpu()
            if r.dtype == torch.bool or r.shape != (B,) or not torch.equal(r.long(), torch.arange(B)):
                return False
        if isinstance(p, (int, slice)):
            return False
        p = torch.as_tensor(p).cpu()
        return p.dtype != torch.bool and p.shape == (B,) and torch.equal(p.long(), self._pos) and not isinstance(idx[0], slice)

    def __getitem__(self, idx):
        if self._full_t is None and self._is_answer_index(idx):
            return self._rows
        return self._full()[idx]

    def __getattr__(self, name):      # anything a tensor has (softmax, float, cpu, argmax, ...): on the materialised tensor
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._full(), name)

    def __len__(self):
        return self._rows.shape[0]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        un = lambda a: a._full() if isinstance(a, _LazyLogits) else a
        return func(*[un(a) for a in args], **{k: un(v) for k, v in (kwargs or {}).items()})


class UnifiedProCyon:
    def __init__(self, config: ProCyonConfig, text_encoder, tokenizer, protein_seq_encoder=None, token_projectors=None,
                 aaseq_shared_projector: Optional[MlpEngine] = None, aaseq_lm_projector: Optional[MlpEngine] = None,
                 protein_seq_embeddings=None, domain_embeddings=None, peptide_embeddings=None,
                 protein_struct_embeddings=None, drug_structure_embeddings=None):
        self.config = config
        self.text_encoder = text_encoder
        self.tokenizer = tokenizer
        self.protein_seq_encoder = protein_seq_encoder
        self.token_projectors = token_projectors or {}
        self.aaseq_shared_projector = aaseq_shared_projector
        self.aaseq_lm_projector = aaseq_lm_projector
        self.protein_seq_embeddings = protein_seq_embeddings
        self.domain_embeddings = domain_embeddings
        self.peptide_embeddings = peptide_embeddings
        self.protein_struct_embeddings = protein_struct_embeddings
        self.drug_structure_embeddings = drug_structure_embeddings
        self.input_embeddings = SimpleNamespace(weight=text_encoder.get_input_embeddings())
        self.device = text_encoder.engine.device
        self.training = False
        self.dtype = torch.float32     # until the caller asks for bf16 (see `bfloat16` below)
        self._tables_f32 = {}          # fp32 embedding tables of an fp32 checkpoint (checkpoint.build_model), kept until .bfloat16()
        self._tables_f32_dev = {}
        self.use_llama_tokenizer = True
        self.train_qa_full_lm = False
        self.struct_dropout_prob = config.protein_struct_dropout
        # special tokens, registered in the order of `_init_tokenizer` (model_unified.py:1100-1133)
        for k, v in special_token_ids(tokenizer, config.text_encoder_fname).items():
            setattr(self, k, v)

    # nn.Module protocol used by the callers (retrieval_utils.py:90-101, procyon.py:64-67).  The reference model is built in the
    # checkpoint's dtype (fp32) and every shipped entry point calls `.bfloat16()` before the first forward -- except
    # examples/paper_analyses/protpep_qa_scores.py:55-58, scripts/qa_filter_captions.py:17-18 and scripts/caption_bulk.py:72-73, which
    # run fp32.  The model tracks the dtype its caller has asked for and computes in it: bf16 on the bf16 engine; fp32 -- as long as the
    # fp32 weights of the checkpoint are still held, i.e. `.bfloat16()` was never called -- on the fp32 operator family
    # (procyon_amd/engine_f32.py: forward, forward_sequences AND generation, `_generate_*_f32`).  Never silently different arithmetic.
    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training is outside the engine's scope (inference / generation only)")
        return self.eval()

    def bfloat16(self):
        self.dtype = BF16
        # the reference's parameters ARE bf16 from here on: the fp32 copies go (they are 2x the bf16 engine's memory)
        for m in [self.text_encoder, self.protein_seq_encoder, self.aaseq_shared_projector, self.aaseq_lm_projector] + list(self.token_projectors.values()):
            if m is not None and hasattr(m, "drop_fp32"):
                m.drop_fp32()
        self._tables_f32, self._tables_f32_dev = {}, {}
        return self

    @property
    def _f32(self):
        return self.dtype == torch.float32

    def _check_fp32_sources(self):
        # (advisor finding, round 4: after .bfloat16() the fp32 weights are gone; a later .float() must fail HERE, not inside forward)
        enc = self.text_encoder
        if enc is not None and getattr(enc, "_src_f32", None) is None and getattr(enc, "_engine_f32", None) is None:
            raise RuntimeError("UnifiedProCyon.float(): the fp32 weights of this model were dropped by .bfloat16() (or the checkpoint was bf16); "
                               "reload the checkpoint to compute in fp32")

    def float(self):
        self._check_fp32_sources()
        self.dtype = torch.float32
        return self

    def half(self):
        self.dtype = torch.float16
        return self

    def to(self, *args, **kwargs):
        """`nn.Module.to`: a dtype argument is recorded (see `_require_bf16`); a device must be the engine's device."""
        cand = list(args) + [kwargs.get("dtype"), kwargs.get("device")]
        for a in cand:
            if isinstance(a, torch.dtype):
                if not a.is_floating_point:
                    raise TypeError(f"nn.Module.to only accepts floating point dtypes, but got desired dtype={a}")
                if a == torch.float32 and self.dtype != torch.float32:
                    self._check_fp32_sources()
                self.dtype = a
            elif isinstance(a, (str, torch.device, int)) and a is not None:
                dev = torch.device("cuda", a) if isinstance(a, int) else torch.device(a)
                mine = torch.device(self.device)
                if dev.type != mine.type or (dev.index is not None and mine.index is not None and dev.index != mine.index):
                    raise RuntimeError(f"the engine's weights live on {mine}; cannot move the model to {dev} "
                                       "(build it with device=... instead)")
        return self

    def _require_bf16_or_fp32(self, what):
        if self.dtype not in (BF16, torch.float32):
            raise RuntimeError(f"UnifiedProCyon.{what}: the engine computes in bfloat16 or float32, not {self.dtype}")

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    # ------------------------------------------------------------------------------------------
    def _embed_table(self, aaseq_type):
        tab = {"protein": self.protein_seq_embeddings, "domain": self.domain_embeddings,
               "peptide": self.peptide_embeddings}[aaseq_type]
        if tab is None:
            raise ValueError(f"no embedding table for aaseq_type={aaseq_type}")
        if self._f32:
            return self._table_f32({"protein": "protein_seq_embeddings", "domain": "domain_embeddings", "peptide": "peptide_embeddings"}[aaseq_type])
        return tab

    def _table_f32(self, name):
        if name not in self._tables_f32_dev:
            if name not in self._tables_f32:
                raise RuntimeError(f"fp32 arithmetic was asked for, but the {name} table is held in bf16 only")
            self._tables_f32_dev[name] = self._tables_f32[name].to(self.device, torch.float32)
        return self._tables_f32_dev[name]

    def _encode_aaseq(self, seq, aaseq_type):
        if self.config.use_aaseq_embeddings:
            return self._embed_table(aaseq_type)[seq.to(self.device).long()]
        if self._f32:
            return self.protein_seq_encoder.forward_f32(seq)[0]
        z, _ = self.protein_seq_encoder(seq, aggregate=True)
        return z

    def _preprocessing(self, inputs, aaseq_type='protein', exclude_protein_structure=False, crop_off=False,
                       no_pad=False, retrieval=False, left_pad=False):
        """`_preprocessing` (model_unified.py:352-481)."""
        aaseq_token_embeddings = aaseq_ret_embeddings = None
        if inputs["data"]["seq"] is not None:
            aaseq_token_embeddings = aaseq_ret_embeddings = self._encode_aaseq(inputs["data"]["seq"], aaseq_type)
        if inputs["input"]["seq"] is not None:
            full_index = list(chain.from_iterable(inputs["input"]["seq"]))
            pz_inputs = aaseq_token_embeddings[torch.tensor(full_index, dtype=torch.long, device=self.device)]
            protein_soft_tokens = self.token_projectors['aaseq'](pz_inputs)
        else:
            protein_soft_tokens = None
        if self.config.use_drug_embeddings and (inputs["data"]["drug"] is not None):
            full_index = list(chain.from_iterable(inputs["input"]["drug"]))
            drug_tab = self._table_f32("drug_structure_embeddings") if self._f32 else self.drug_structure_embeddings
            drug_z = drug_tab[inputs["data"]["drug"].to(self.device).long()][full_index]
            drug_soft_tokens = self.token_projectors["drug"](drug_z)
        else:
            drug_soft_tokens = None
        text_inputs = [[inputs["data"]["text"][i] for i in inp_list] for inp_list in inputs["input"]["text"]]
        instruction_list = inputs['instructions']
        protein_struct_tokens = []
        if (not exclude_protein_structure) and self.config.use_protein_struct and inputs["input"]["seq"]:
            # mutates inputs["instructions"] in place and draws from the global torch RNG, as the reference
            # does (model_unified.py:422,433-437; quirk Q6)
            include_mask = torch.bernoulli(torch.full((len(instruction_list),), 1 - self.struct_dropout_prob))
            all_row_indices = []
            for i in include_mask.nonzero(as_tuple=True)[0].tolist():
                instruction_list[i] = instruction_list[i].replace("<|protein|>", "<|protein|> <|struct|>")
                all_row_indices.append(torch.cat([inputs["data"]["seq_idx"][j].unsqueeze(0) for j in inputs["input"]["seq"][i]]))
            all_row_indices = torch.stack(all_row_indices, dim=0)
            ari_unique, ari_inverse = all_row_indices.unique(return_inverse=True)
            if aaseq_type == "protein":
                struct_z = (self._table_f32("protein_struct_embeddings") if self._f32 else self.protein_struct_embeddings)[ari_unique.to(self.device).long()]
            else:
                struct_z = torch.zeros(ari_unique.shape[0], self.protein_struct_embeddings.shape[1], dtype=self.dtype, device=self.device)
            token_z_expand = self.token_projectors["prot_structure"](struct_z)[ari_inverse.to(self.device)]
            for i, val in enumerate(include_mask):
                protein_struct_tokens.append(token_z_expand[i, ...] if val else [])
        input_ids, attn_masks = self._prepare_text_inputs_and_tokenize(
            instruction_list, text_inputs, crop_off=crop_off, retrieval=retrieval, no_pad=no_pad, left_pad=left_pad)
        input_embeds, ret_output_indices = self._prepare_input_embeddings(
            input_ids, protein_soft_tokens=protein_soft_tokens, protein_struct_tokens=protein_struct_tokens,
            drug_soft_tokens=drug_soft_tokens)
        return input_embeds, input_ids, attn_masks, ret_output_indices, aaseq_token_embeddings, aaseq_ret_embeddings

    def _prepare_text_inputs_and_tokenize(self, instructions: List[str], text_input_list: List[List[str]], crop_off=False,
                                          retrieval=False, no_pad=False, left_pad=False):
        """`_prepare_text_inputs_and_tokenize` (model_unified.py:1177-1293), eval path (no crop sampling)."""
        tk = self.tokenizer
        assert all([t[-1] != tk.sep_token for t in instructions])
        instruction_tokens = tk(instructions, padding=False, truncation=True, add_special_tokens=True,
                                max_length=self.config.max_text_len)['input_ids']
        max_len = max(len(l) for l in instruction_tokens)
        joint_tokens, attention_masks = [], []
        for i, text_input in enumerate(text_input_list):
            n_txt = len(text_input)
            if n_txt != 0:
                for j in range(n_txt):
                    if not isinstance(text_input[j], str):
                        text_input[j] = "null"
                toks = tk(text_input, padding=False, truncation=False, add_special_tokens=False)['input_ids']
                max_len_for_sample = (self.config.max_text_len - max_len) // n_txt
                for j in range(len(toks)):
                    drug_add = None
                    if self.drug_idx in toks[j]:
                        where_drug = toks[j].index(self.drug_idx) - 3
                        drug_add = toks[j][(where_drug - 3):]
                        toks[j] = toks[j][:(where_drug - 3)]
                    end_i = max_len_for_sample - (len(drug_add) if drug_add is not None else 0)
                    toks[j] = toks[j][0:end_i]
                    if drug_add is not None:
                        toks[j] = toks[j] + drug_add
            else:
                toks = []
            Lt = multi_replace_tokens(list(instruction_tokens[i]), toks, self.ext_idx, eval=False)
            if no_pad:
                Lt = torch.tensor(Lt)
            else:
                Lt = torch.tensor(Lt + [tk.eos_token_id] + [tk.pad_token_id] * max(self.config.max_text_len - len(Lt) - 1, 0))
            joint_tokens.append(Lt)
            attention_masks.append((Lt != tk.pad_token_id).int())
            assert not torch.any(Lt == self.ext_idx), 'ERROR [EXT] found in input'
        if left_pad:
            return left_pad_tensors(joint_tokens, pad_value=tk.pad_token_id)
        return torch.stack(joint_tokens, dim=0), torch.stack(attention_masks, dim=0)

    def _prepare_input_embeddings(self, input_ids, protein_soft_tokens=None, protein_struct_tokens=[], drug_soft_tokens=None):
        """`_prepare_input_embeddings` (model_unified.py:1135-1175): embedding lookup with the soft tokens written
        over the <|protein|> / <|struct|> / <|drug|> rows, in row-major order, by ONE gather kernel."""
        ids = input_ids.long()
        flat = ids.reshape(-1)
        soft_map = torch.full((flat.numel(),), -1, dtype=torch.int32)
        pieces, base = [], 0
        if protein_soft_tokens is not None:
            m = flat == self.prot_replacement_idx
            assert int(m.sum()) == protein_soft_tokens.shape[0]
            soft_map[m] = torch.arange(int(m.sum()), dtype=torch.int32) + base
            pieces.append(protein_soft_tokens)
            base += protein_soft_tokens.shape[0]
        if len(protein_struct_tokens) > 0:
            ms = ids == self.struct_idx
            for i in range(ids.shape[0]):
                n = int(ms[i].sum())
                if n > 0:
                    assert n == protein_struct_tokens[i].shape[0], f"expected: {n} got: {protein_struct_tokens[i].shape[0]}"
                    row = torch.zeros_like(ids, dtype=torch.bool)
                    row[i] = ms[i]
                    soft_map[row.reshape(-1)] = torch.arange(n, dtype=torch.int32) + base
                    pieces.append(protein_struct_tokens[i])
                    base += n
        if drug_soft_tokens is not None:
            m = flat == self.drug_idx
            assert int(m.sum()) == drug_soft_tokens.shape[0], f"expected: {int(m.sum())} got: {drug_soft_tokens.shape[0]}"
            soft_map[m] = torch.arange(int(m.sum()), dtype=torch.int32) + base
            pieces.append(drug_soft_tokens)
        soft = torch.cat(pieces, 0).contiguous() if pieces else None
        eng = self.text_encoder.engine_f32 if self._f32 else self.text_encoder.engine
        z = eng.embed_tokens(ids, soft, soft_map if pieces else None)
        ret = ids == self.prot_retrieval_idx
        if self.config.roll_num != 0:
            ret = ret.roll(self.config.roll_num, 1)
        return z, ret

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, inputs, return_mlm=False, retrieval=False, get_full_labels=False, aaseq_type='protein',
                exclude_protein_structure=False, crop_off=False, output_attentions=False, full_logits=False):
        """`forward` (model_unified.py:483-581), inference branches.  QA: `outputs.logits` reads like the reference's [B, T, V] tensor
        (:548-554) -- index it at the answer positions (`logits[torch.arange(B), pos]`, data/inference_utils.py:582-604) and the rows that
        were computed come back; any other access materialises every position once (_LazyLogits).  `outputs.answer_logits` [B, 1, V] and
        out["answer_positions"] are the engine's own handle on the answer rows.  Retrieval: contrastive_out["positive"]["text"] [B, D].
        full_logits=True (not in the reference's signature): outputs.logits is a plain [B, T_real, V] tensor from the start (the reference pads
        every row to max_text_len and materialises [B, 2048, V]; the trailing all-pad columns are not computed here)."""
        if return_mlm:
            raise NotImplementedError("return_mlm is a training path (model_unified.py:505-509)")
        self._require_bf16_or_fp32("forward")
        input_embeds, input_ids, attn_masks, ret_idx, tok_emb, ret_emb = self._preprocessing(
            inputs, aaseq_type=aaseq_type, crop_off=crop_off, retrieval=retrieval, exclude_protein_structure=False)
        full_labels = None
        B = input_ids.shape[0]
        # the reference right-pads every row to max_text_len (Q13); causal rows are unaffected by trailing pads,
        # so only the columns up to the last real token are run
        real = int(attn_masks.sum(1).max())
        emb = input_embeds[:, :real].contiguous()
        answer_pos = None
        if not retrieval:
            pad_id = self.tokenizer.pad_token_id
            full_labels = input_ids.clone()
            all_masks = (full_labels == pad_id) | (full_labels == self.prot_replacement_idx) | \
                (full_labels == self.prot_retrieval_idx) | (full_labels == self.drug_idx) | (full_labels == self.struct_idx)
            if self.use_llama_tokenizer:
                all_masks[:, -1] = True
            if not self.train_qa_full_lm:
                all_masks |= mask_before(full_labels, self.answer_idx, before_last_answer=True)
            full_labels = torch.where(all_masks, -100, full_labels)
            answer_pos = torch.tensor([int((input_ids[i] == self.answer_idx).nonzero()[:, 0].max()) for i in range(B)])
        if retrieval and self.config.ret_token_access not in ('last', 'all'):
            raise NotImplementedError("Invalid option {} for ret_token_access".format(self.config.ret_token_access))
        sum_all = retrieval and self.config.ret_token_access == 'all'
        ret_rows = ret_idx[:, :real].reshape(-1).nonzero()[:, 0] if sum_all else None   # flat b*T + t, row-major like boolean indexing
        outputs = self.text_encoder(input_embeds=emb, attn_masks=attn_masks[:, :real], full_labels=full_labels,
                                    logit_positions=None if full_logits else (answer_pos if not retrieval else torch.zeros(B, dtype=torch.long)),
                                    want_hidden=retrieval and not sum_all, hidden_sum_positions=ret_rows, lazy_hidden=True)
        if not retrieval and not full_logits:
            # the reference's `outputs.logits` is [B, T, V]; here the answer rows are computed and the rest on first access (_LazyLogits)
            enc, am = self.text_encoder, attn_masks[:, :real]
            outputs.answer_logits = outputs.logits                       # [B, 1, V]
            outputs.logits = _LazyLogits(outputs.answer_logits[:, 0], answer_pos, real,
                                         lambda: enc(input_embeds=emb, attn_masks=am, logit_positions=None, want_hidden=False, lazy_hidden=True).logits)
        out = {'outputs': outputs, 'text_toks': input_ids, 'full_labels': full_labels if get_full_labels else None,
               'contrastive_out': None, 'contrastive_loss': None, 'answer_positions': answer_pos}
        if retrieval:
            if sum_all:   # torch.stack(hidden_states, -1).sum(-1)[ret] (model_unified.py:560-563), computed at the [PROT] rows only
                extracted = outputs.hidden_state_sum_rows
            else:
                hidden = outputs.hidden_states[-1]
                extracted = hidden[ret_idx[:, :real].to(hidden.device)]
            shared_lm = self.aaseq_lm_projector(extracted)
            c = {"positive": {}, "negative": {}}
            if inputs["target"]["text"] is None:
                c["positive"]["text"] = shared_lm
            else:
                c["positive"]["text"] = shared_lm[inputs["target"]["text"]["positive"]]
                if inputs["target"]["text"]["negative"] is not None:
                    raise NotImplementedError
            if inputs["target"]["seq"] is not None:
                shared_plm = self.aaseq_shared_projector(ret_emb)
                c["positive"]["sequence"] = shared_plm[inputs["target"]["seq"]["positive"]]
                if inputs['target']["seq"]["negative"] is not None:
                    raise NotImplementedError
            out['contrastive_out'] = c
        return out

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def from_pretrained(**kw):
        """`UnifiedProCyon.from_pretrained` (model_unified.py:1296-1394); see procyon_amd.checkpoint.from_pretrained."""
        from ..checkpoint import from_pretrained
        return from_pretrained(**kw)

    @staticmethod
    def get_checkpoint_configs(resume_from_checkpoint):
        """(data_args, model_args, train_args) (model_unified.py:1396-1406)"""
        from ..checkpoint import get_checkpoint_configs
        return get_checkpoint_configs(resume_from_checkpoint)

    def forward_sequences(self, seq_input, get_soft_tokens=False, aaseq_type="protein"):
        """`forward_sequences` (model_unified.py:1029-1086)."""
        self._require_bf16_or_fp32("forward_sequences")
        if isinstance(seq_input, dict):
            seq_input = seq_input["data"]
        z = self._encode_aaseq(seq_input, aaseq_type)
        out = {"original": z, "shared": self.aaseq_shared_projector(z), "token": None}
        if get_soft_tokens:
            out["token"] = self.token_projectors['aaseq'](z)
        return out

    # ------------------------------------------------------------------------------------------
    def _get_nucleus_mask(self, probs, nucleus_prob):
        """`_get_nucleus_mask` (model_unified.py:844-858)."""
        sorted_vals, indices = probs.sort(dim=-1, descending=False)
        keep_idxs = (sorted_vals.cumsum(dim=-1) >= (1 - nucleus_prob)).nonzero(as_tuple=True)
        mask = torch.zeros_like(probs)
        mask[keep_idxs[0], indices[keep_idxs]] = 1
        return mask

    def _sampling_probs(self, logits, temperature=1.0, nucleus_prob=None):
        """Pre-sampling probability vector of `_generate_sampling` (model_unified.py:899-903): nucleus -> softmax(logits)
        times the un-renormalised nucleus mask; otherwise softmax(logits / temperature).  In the logits' dtype, like the
        reference."""
        if nucleus_prob is not None:
            probs = logits.softmax(dim=-1)
            return probs * self._get_nucleus_mask(probs, nucleus_prob)
        return (logits / temperature).softmax(dim=-1)

    @torch.no_grad()
    def _generate_sampling(self, input_embeds, attn_masks, max_len=64, num_text_per_instance=1, temperature=1.0,
                           greedy=False, nucleus_prob=None):
        """`_generate_sampling` (model_unified.py:861-921).  Greedy runs entirely on the device (hipGraph-replayed
        decode steps, fused argmax + log-prob), and so does sampling: `pcy_llama_sample` forms the reference's pre-sampling
        probability vector (`_sampling_probs` below is its torch restatement) and draws by inverse CDF with uniform variates
        taken from torch's device generator."""
        assert nucleus_prob is None or (0 < nucleus_prob < 1)
        eng = self.text_encoder.engine
        B = len(input_embeds)
        out_list, lp_list, logit_list = [], [], []
        for _ in range(num_text_per_instance):
            if greedy:
                tok, lp, lg, _ = eng.generate_greedy(input_embeds, attn_masks, max_len, keep_logits=True)
                out_list.append(tok.cpu())
                lp_list.append(lp.cpu().clone())
                logit_list.append(lg.cpu())
                continue
            # the whole loop on the device: decode launches + the sampling kernels (softmax, nucleus mask by histogram, inverse-CDF
            # draw with uniform variates from torch's device generator); the logits record goes to the host as it is produced
            tok, lp, lg, _ = eng.generate_sampling(input_embeds, attn_masks, max_len, temperature=temperature, nucleus_prob=nucleus_prob,
                                                   keep_logits=True)
            out_list.append(tok.cpu())
            lp_list.append(lp.cpu().clone())
            logit_list.append(lg.cpu())
        # one text per instance (every caller in the reference): a view instead of a host-to-host copy of the [B, max_len, V] record
        # (4.2 GB at batch 32 x 512 tokens)
        logits_out = logit_list[0].unsqueeze(1) if len(logit_list) == 1 else torch.stack(logit_list, dim=1)
        return torch.stack(out_list, dim=1), torch.stack(lp_list).T, logits_out

    @torch.no_grad()
    def _generate_beam_search(self, input_embeds, attn_mask, max_len=64, beam_size=5, beam_group_size=5,
                              diversity_penalty=0.8):
        """`_generate_beam_search` (model_unified.py:702-842): diverse beam search with the reference's exact
        bookkeeping (step-0 single-beam top-g, in-place Hamming penalty carried in the score, bf16 log-softmax +
        fp32 running score, EOS-anywhere stop).  The transformer steps run on the engine; the per-step
        O(B x groups) bookkeeping is `pcy_beam_step` (one launch; ties between equal candidate scores go to the lowest flat
        index, where torch.topk leaves the order unspecified)."""
        B = input_embeds.shape[0]
        BB = B * beam_size
        V = self.text_encoder.model.vocab_size
        if beam_size % beam_group_size != 0:
            raise ValueError("beam_group_size must evenly divide beam_size, got: "
                             f"{beam_size} % {beam_group_size} != 0")
        enc = self.text_encoder
        if beam_size > 32:
            raise ValueError("beam_size > 32 is not supported by the device-side beam step")
        T = input_embeds.shape[1]
        keep_new = enc.max_new_tokens
        enc.max_new_tokens = max(keep_new, max_len)
        try:
            return self._beam_search_body(input_embeds, attn_mask, B, BB, V, T, max_len, beam_size, beam_group_size, diversity_penalty)
        finally:
            enc.max_new_tokens = keep_new

    def _beam_search_body(self, input_embeds, attn_mask, B, BB, V, T, max_len, beam_size, beam_group_size, diversity_penalty):
        from ..engine import BeamState, GenState
        dev = self.device
        enc = self.text_encoder
        eng = enc.engine
        # Everything per step stays on the device and nothing synchronises: the decode step is ONE replayed hipGraph reading the
        # next tokens and the position from device memory, the reference's per-group bookkeeping is ONE launch (pcy_beam_step),
        # the KV reorder two (rows that keep their place are skipped).  The logits record is kept per SLOT and step and is
        # re-indexed once at the end along the parent chain (the reference re-indexes the whole history on the host in every
        # group of every step, model_unified.py:827-829).  The EOS stop (:833) is decided on the device; once it has fired the
        # queued steps change nothing, so the host looks at the flag only every few steps.
        # The reference replicates every prompt x beam BEFORE the prefill (model_unified.py:751-752): beam x the prefill on identical rows.
        # Here each prompt is prefilled ONCE into row b of a BB-row cache; its K / V rows are then copied to the rows of its beams (one
        # pcy_kv_reorder with the constant source map r -> r // beam) and its last-row logits repeated.  PCY_DISABLE=beam_prefill_once: the
        # reference's replicated prefill (a BB-row batch may take other GEMM tiles than a B-row one: equal to bf16 noise, tests).
        if beam_size > 1 and "beam_prefill_once" not in os.environ.get("PCY_DISABLE", "").split(","):
            cache = eng.new_cache(BB, T + enc.max_new_tokens)
            lg_b, _ = eng.prefill(input_embeds.to(eng.device), attn_mask, cache, "last")
            eng.kv_reorder(cache, torch.arange(BB, dtype=torch.int32) // beam_size, T)
            logits = lg_b.repeat_interleave(beam_size, dim=0).contiguous()
        else:
            emb_rep = torch.repeat_interleave(input_embeds, repeats=beam_size, dim=0)
            mask_rep = torch.repeat_interleave(attn_mask, repeats=beam_size, dim=0)
            o = enc(input_embeds=emb_rep, attn_masks=mask_rep, use_cache=True, past_key_values=None,
                    logit_positions=torch.full((BB,), T - 1), want_hidden=False)
            cache = o.past_key_values.cache
            logits = o.logits[:, -1, :].contiguous()
        # The beams of a prompt hold the SAME K / V rows in slots [0, T): the reference re-indexes the whole history of every row in every group
        # of every step (:830-832); moving equal bytes changes nothing, so the reorder starts at slot T (pcy_kv_reorder_range; at 10 beams and a
        # 512-token prompt the reorder was 0.4-0.6 ms of a 4.5 ms step).  PCY_DISABLE=beam_kv_suffix: every slot, as the reference (same result).
        kv_t0 = 0 if "beam_kv_suffix" in os.environ.get("PCY_DISABLE", "").split(",") else T
        if (kv_t0 * self.text_encoder.cfg.head_dim) % 8:
            kv_t0 = 0
        bs = BeamState(B, beam_size, max_len, self.tokenizer.eos_token_id, prompt_len=T, device=dev)
        st = GenState(BB, V, 1, dev)
        st.pos, st.next_tok = bs.pos, bs.next_tok                  # the decode graph reads what the beam step writes
        st.c.pos, st.c.next_tok = bs.pos.data_ptr(), bs.next_tok.data_ptr()
        rec = torch.empty(max_len, BB, V, dtype=logits.dtype, device=dev)
        if "beam_graph" in os.environ.get("PCY_DISABLE", "").split(","):   # the four calls per step (same kernels, same bits; tests)
            for i in range(max_len):
                if i > 0:
                    if T + i > cache.Tmax:
                        raise ValueError(f"KV cache capacity {cache.Tmax} exhausted; raise max_new_tokens")
                    eng.decode_graph(cache, st, BB)
                    logits = st.logits
                rec[i].copy_(logits)
                eng.beam_step(logits, bs, beam_group_size, diversity_penalty)
                eng.kv_reorder(cache, bs.src, T + i, t0=kv_t0)
                if (i & 7) == 7 and int(bs.done):
                    break
        else:
            # step 0 selects on the prefill's logits; every later step (decode -> record -> beam step -> KV reorder) is ONE replayed launch
            # chain (pcy_llama_beam_steps), enqueued up to the next multiple of 8 steps, where the host looks at the EOS flag
            rec[0].copy_(logits)
            eng.beam_step(logits, bs, beam_group_size, diversity_penalty)
            eng.kv_reorder(cache, bs.src, T, t0=kv_t0)
            i = 1
            while i < max_len:
                if T + i > cache.Tmax:
                    raise ValueError(f"KV cache capacity {cache.Tmax} exhausted; raise max_new_tokens")
                n = min(8 - (i & 7), max_len - i, cache.Tmax - T - i + 1)
                eng.beam_steps(cache, st, bs, beam_group_size, diversity_penalty, rec, n, kv_t0=kv_t0)
                i += n
                if (i & 7) == 0 and int(bs.done):
                    break
        out, steps = bs.tokens()                                   # synchronises
        anc = bs.anc[:steps].long()
        slot = torch.arange(BB, device=dev)
        idx = torch.empty(steps, BB, dtype=torch.long, device=dev)
        for s_ in range(steps - 1, -1, -1):                        # the record of step s is re-indexed by the parents of steps >= s
            slot = anc[s_][slot]
            idx[s_] = slot
        # [BB, steps, V] on the device, then ONE copy into pinned host memory (a pageable destination moves the 2.5 MB per step
        # and beam-10 record at a few GB/s: ~1 ms per generated token)
        out_logits_dev = rec[:steps].gather(1, idx[:, :, None].expand(steps, BB, V)).transpose(0, 1).contiguous()
        out_logits = torch.empty(out_logits_dev.shape, dtype=out_logits_dev.dtype, pin_memory=True)
        out_logits.copy_(out_logits_dev, non_blocking=True)
        full = torch.zeros(BB, max_len, dtype=torch.int64)
        full[:, :steps] = out.cpu()
        out, cur = full, bs.cur.cpu()
        eng.ctx.sync()      # stream complete + the sticky watchdog word of the fused launches checked (raises PcyError)
        return (out.unflatten(0, (B, beam_size)), cur.unflatten(0, (B, beam_size)), out_logits.unflatten(0, (B, beam_size)))

    # ---- fp32 generation (/root/reference/scripts/caption_bulk.py:70-73 never casts the model and calls generate(method="beam")): the
    # reference's own loops over the sub-module waist, every operator on the fp32 family (procyon_amd/engine_f32.py) -- a compatibility
    # path, one launch per operator and a host round trip per step, not a fast one
    @torch.no_grad()
    def _generate_sampling_f32(self, input_embeds, attn_masks, max_len=64, num_text_per_instance=1, temperature=1.0, greedy=False,
                               nucleus_prob=None):
        """`_generate_sampling` (model_unified.py:861-921) in fp32: prefill, then max_len - 1 cached steps; greedy argmax, or multinomial
        on softmax(logits / temperature) / the un-renormalised nucleus-masked softmax; chosen log-probabilities summed; no EOS stop."""
        assert nucleus_prob is None or (0 < nucleus_prob < 1)
        enc = self.text_encoder
        B = len(input_embeds)
        keep_new = enc.max_new_tokens
        enc.max_new_tokens = max(keep_new, max_len)
        try:
            outs, lps, lgs = [], [], []
            for _ in range(num_text_per_instance):
                toks, past, rec = None, None, []
                total = torch.zeros(B, device=self.device)
                for i in range(max_len):
                    o = enc(input_embeds=input_embeds, attn_masks=attn_masks, use_cache=True, logit_positions=torch.full((B,), input_embeds.shape[1] - 1)) \
                        if i == 0 else enc(input_ids=toks[:, -1:], use_cache=True, past_key_values=past)
                    past = o.past_key_values
                    logits = o.logits[:, -1, :]
                    rec.append(logits.cpu())
                    lsm = logits.log_softmax(-1)
                    nxt = logits.argmax(-1, keepdim=True) if greedy else torch.multinomial(self._sampling_probs(logits, temperature, nucleus_prob), 1)
                    total += lsm.gather(1, nxt)[:, 0]
                    toks = nxt if toks is None else torch.cat([toks, nxt], -1)
                outs.append(toks.cpu()); lps.append(total.cpu()); lgs.append(torch.stack(rec, 1))
        finally:
            enc.max_new_tokens = keep_new
        return torch.stack(outs, 1), torch.stack(lps).T, torch.stack(lgs, 1)

    @torch.no_grad()
    def _generate_beam_search_f32(self, input_embeds, attn_mask, max_len=64, beam_size=5, beam_group_size=5, diversity_penalty=0.8):
        """`_generate_beam_search` (model_unified.py:702-842) in fp32: the prompt repeated beam_size times before the prefill; step 0 extends
        beam 0 of every group only, later steps the group's g x V candidates; groups after the first lose penalty x (count of the tokens
        the earlier groups of the same input chose this step) IN the running score; outputs, scores, the logits record and the K / V rows
        follow the selected parents; stops when every beam holds an EOS."""
        if beam_size % beam_group_size != 0:
            raise ValueError(f"beam_group_size must evenly divide beam_size, got: {beam_size} % {beam_group_size} != 0")
        enc = self.text_encoder
        dev = self.device
        B = input_embeds.shape[0]
        BB, V, T = B * beam_size, enc.cfg.vocab, input_embeds.shape[1]
        groups = beam_size // beam_group_size
        emb = torch.repeat_interleave(input_embeds, beam_size, dim=0)
        mask = torch.repeat_interleave(attn_mask, beam_size, dim=0)
        cur = torch.zeros(BB, device=dev)
        out = torch.zeros(BB, max_len, dtype=torch.int64, device=dev)
        rec = None
        keep_new = enc.max_new_tokens
        enc.max_new_tokens = max(keep_new, max_len)
        try:
            past = None
            for i in range(max_len):
                o = enc(input_embeds=emb, attn_masks=mask, use_cache=True, logit_positions=torch.full((BB,), T - 1)) if i == 0 else \
                    enc(input_ids=out[:, i - 1:i], use_cache=True, past_key_values=past)
                past = o.past_key_values
                logits = o.logits[:, -1, :]
                step_rec = logits.cpu().unsqueeze(1)
                rec = step_rec if rec is None else torch.cat([rec, step_rec], 1)
                lp = logits.log_softmax(-1) + cur[:, None]
                src = torch.arange(BB, device=dev)
                for b in range(B):
                    b0 = b * beam_size
                    for gi in range(groups):
                        g0 = b0 + gi * beam_group_size
                        g1 = g0 + beam_group_size
                        cand = lp[g0:(g0 + 1 if i == 0 else g1)]
                        if gi != 0:
                            cand -= diversity_penalty * torch.bincount(out[b0:g0, i], minlength=V).to(dev)
                        top_v, top_i = cand.ravel().topk(beam_group_size)
                        parents = top_i // V + g0
                        out[g0:g1] = out[parents]
                        out[torch.arange(g0, g1), i] = top_i % V
                        cur[g0:g1] = top_v
                        rec[g0:g1] = rec[parents.cpu()]
                        src[g0:g1] = src[parents]        # (the K / V rows of every layer follow the parents, :830-832)
                past.cache.reorder_(src, past.t)
                if bool((out == self.tokenizer.eos_token_id).any(dim=1).all()):
                    break
        finally:
            enc.max_new_tokens = keep_new
        return (out.cpu().unflatten(0, (B, beam_size)), cur.cpu().unflatten(0, (B, beam_size)), rec.unflatten(0, (B, beam_size)))

    @torch.no_grad()
    def generate(self, inputs, max_len=64, aaseq_type='protein', method="sampling", temperature=1.0, greedy=False,
                 num_text_per_instance=1, return_all_internals=False, beam_size=5, beam_group_size=5,
                 diversity_penalty=0.8, exclude_protein_structure=False, nucleus_prob=0.9, truncate_on_eos=True):
        """`generate` (model_unified.py:924-1027).  The non-beam call site of the reference mis-binds its positional
        arguments (:998-1005, quirk Q8); the evident intent -- the same names bound by keyword -- is implemented, which
        means `nucleus_prob` (default 0.9) reaches `_generate_sampling` for every non-beam method, as written there;
        pass nucleus_prob=None for plain temperature sampling."""
        assert method in ["sampling", "temperature", "greedy", "beam", "nucleus"]
        self._require_bf16_or_fp32("generate")      # fp32 (a caller that never called .bfloat16()): the operator-by-operator fp32 loops below
        if method == "beam":
            num_text_per_instance = beam_size
        elif method == "greedy":
            greedy = True
        elif method in ["sampling", "nucleus"]:
            temperature = 1
        if temperature < 1e-8:
            greedy = True
        input_embeds, input_ids, attn_masks, _, _, _ = self._preprocessing(
            inputs, aaseq_type=aaseq_type, crop_off=True, no_pad=True,
            exclude_protein_structure=exclude_protein_structure, left_pad=True)
        whole_instructions = self.tokenizer.batch_decode(input_ids)
        gt_text = [inputs["data"]["text"][i] for i in inputs["target"]["text"]] if inputs["target"]["text"] is not None else None
        batch_size = input_embeds.shape[0]
        if method == "beam":
            tokens, log_probs, logits = (self._generate_beam_search_f32 if self._f32 else self._generate_beam_search)(
                input_embeds, attn_masks, max_len=max_len, beam_size=beam_size, diversity_penalty=diversity_penalty,
                beam_group_size=beam_group_size)
        else:
            tokens, log_probs, logits = (self._generate_sampling_f32 if self._f32 else self._generate_sampling)(
                input_embeds, attn_masks, max_len=max_len, num_text_per_instance=num_text_per_instance,
                temperature=temperature, greedy=greedy, nucleus_prob=nucleus_prob)
        flattened = torch.flatten(tokens, start_dim=0, end_dim=1)
        text = self.tokenizer.batch_decode(flattened)
        if truncate_on_eos:
            text = [x.split(self.tokenizer.eos_token)[0].strip() for x in text]
        text = [text[i * num_text_per_instance:(i + 1) * num_text_per_instance] for i in range(batch_size)]
        if return_all_internals:
            return {"out_tokens": tokens, "out_logits": logits, "out_log_probs": log_probs, "text": text,
                    "input_instructions": whole_instructions, "ground_truth_text": gt_text,
                    "text_references": inputs["reference_indices"]["target"]["text"],
                    "seq_references": [inputs["reference_indices"]["input"]["seq"][j][-1] for j, _ in enumerate(inputs["input"]["seq"])]}
        return tokens, log_probs, logits, text
