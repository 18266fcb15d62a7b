"""Checkpoint ingest: map a ProCyon state dict (`txllm_model_ckpt.pt` or the fp32 dict merged from
DeepSpeed ZeRO shards, /root/reference/procyon/model/model_unified.py:1371-1382) onto the engine.

Key layout of the reference's `UnifiedProCyon.state_dict()`:
  text_encoder.model.<HF LlamaForCausalLM keys>          (pmc_llama.py:478-487: self.model = LlamaForCausalLM)
  protein_seq_encoder.model.<fair-esm ESM2 keys | HF Esm keys>   (esm.py:378-420)
  token_projectors.{aaseq,prot_structure,drug}.<i>.{weight,bias}   create_mlp Sequential: Linear at 0, 3, 6, ...
  aaseq_shared_projector.<i>.*, aaseq_lm_projector.<i>.*
  {protein_seq,domain,peptide,protein_struct,drug_structure}_embeddings.weight
No checkpoint exists on the build / GPU boxes; the key mapping is unit-tested on synthetic dicts (tests/test_host_cpu.py).
"""
from __future__ import annotations

import re

import torch

_FAIR2HF = [
    (r"^embed_tokens\.weight$", "esm.embeddings.word_embeddings.weight"),
    (r"^emb_layer_norm_after\.(weight|bias)$", r"esm.encoder.emb_layer_norm_after.\1"),
    (r"^layers\.(\d+)\.self_attn\.q_proj\.(weight|bias)$", r"esm.encoder.layer.\1.attention.self.query.\2"),
    (r"^layers\.(\d+)\.self_attn\.k_proj\.(weight|bias)$", r"esm.encoder.layer.\1.attention.self.key.\2"),
    (r"^layers\.(\d+)\.self_attn\.v_proj\.(weight|bias)$", r"esm.encoder.layer.\1.attention.self.value.\2"),
    (r"^layers\.(\d+)\.self_attn\.out_proj\.(weight|bias)$", r"esm.encoder.layer.\1.attention.output.dense.\2"),
    (r"^layers\.(\d+)\.self_attn_layer_norm\.(weight|bias)$", r"esm.encoder.layer.\1.attention.LayerNorm.\2"),
    (r"^layers\.(\d+)\.fc1\.(weight|bias)$", r"esm.encoder.layer.\1.intermediate.dense.\2"),
    (r"^layers\.(\d+)\.fc2\.(weight|bias)$", r"esm.encoder.layer.\1.output.dense.\2"),
    (r"^layers\.(\d+)\.final_layer_norm\.(weight|bias)$", r"esm.encoder.layer.\1.LayerNorm.\2"),
    # masked-LM head (RobertaLMHead: dense -> gelu -> layer_norm -> tied decoder + bias), read by ESM_PLM.forward(aggregate=False)
    (r"^lm_head\.weight$", "lm_head.decoder.weight"),
]


def fair_esm_to_hf(sd):
    """fair-esm ESM2 parameter names -> the HF Esm names the engine consumes.  Unused tensors (contact head,
    rotary inv_freq buffers) are dropped; the masked-LM head is kept (ESM_PLM.forward(aggregate=False) returns its logits).
    Already-HF dicts pass through."""
    out = {}
    for k, v in sd.items():
        if k.startswith("esm.") or (k.startswith("lm_head.") and k != "lm_head.weight"):   # HF names (the masked-LM head keeps its names)
            out[k] = v
            continue
        for pat, rep in _FAIR2HF:
            nk, n = re.subn(pat, rep, k)
            if n:
                out[nk] = v
                break
    return out


def split_state_dict(sd):
    """ProCyon state dict -> dict(llama=HF-named, esm=HF-named or None, projectors={name: [(W,b|None),...]},
    tables={name: tensor})."""
    llama = {k[len("text_encoder.model."):]: v for k, v in sd.items() if k.startswith("text_encoder.model.")}
    esm_raw = {k[len("protein_seq_encoder.model."):]: v for k, v in sd.items() if k.startswith("protein_seq_encoder.model.")}
    esm = fair_esm_to_hf(esm_raw) if esm_raw else None
    esm_fair = any(not k.startswith("esm.") for k in esm_raw)      # fair-esm parameter names: the fair-esm ESM2 class produced them

    def mlp(prefix):
        idx = sorted({int(m.group(1)) for k in sd for m in [re.match(re.escape(prefix) + r"\.(\d+)\.weight$", k)] if m})
        return [(sd[f"{prefix}.{i}.weight"], sd.get(f"{prefix}.{i}.bias")) for i in idx]

    projectors = {}
    for name in ("aaseq", "prot_structure", "drug"):
        layers = mlp(f"token_projectors.{name}")
        if layers:
            projectors["token_" + name] = layers
    for name in ("aaseq_shared_projector", "aaseq_lm_projector"):
        layers = mlp(name)
        if layers:
            projectors[name] = layers
    tables = {k[:-len(".weight")]: v for k, v in sd.items() if re.match(r"^(protein_seq|domain|peptide|protein_struct|drug_structure)_embeddings\.weight$", k)}
    return dict(llama=llama, esm=esm, esm_fair=esm_fair, projectors=projectors, tables=tables)


TEXT_LORA_ALPHA = 8.0   # pmc_llama.py:430-431: `UnifiedProCyon` never forwards lora_r / lora_alpha to LlamaPostTokenization
                        # (model_unified.py:147-158), so every text adapter was trained and runs with r = 16, alpha = 8 (scale 0.5)


def lora_alpha_for(key_prefix, config=None):
    """LoRA alpha of the adapter whose base weight lives under `key_prefix` (PEFT scale = alpha / r, r read from the tensors).

    text_encoder.*        -> 8 (the constructor default above; the reference has no way to set another value)
    protein_seq_encoder.* -> config.aaseq_lora_alpha (model_unified.py:226-228; esm.py:436-437)
    anything else / missing config field -> ValueError: an unknown alpha must never silently become scale 1."""
    if key_prefix.startswith("text_encoder."):
        return TEXT_LORA_ALPHA
    if key_prefix.startswith("protein_seq_encoder."):
        a = getattr(config, "aaseq_lora_alpha", None) if config is not None else None
        if a is None:
            raise ValueError(f"LoRA adapter under {key_prefix}: the checkpoint's model_args carry no aaseq_lora_alpha; pass lora_alpha=")
        return float(a)
    raise ValueError(f"LoRA adapter under {key_prefix}: no rule for its alpha; pass lora_alpha=")


def merge_lora(sd, lora_alpha=None, config=None):
    """Checkpoints trained with `use_lora` carry PEFT names (peft 0.5.0): `<prefix>.base_model.model.<hf key>` for the frozen
    weights (a Linear that holds an adapter: `...<name>.weight` stays under that name in 0.5.0, `base_layer.weight` in later
    versions) and `...<name>.lora_A.<adapter>.weight` [r, in], `...lora_B.<adapter>.weight` [out, r].  The reference loads them
    into a PEFT-wrapped module; the engine packs plain matrices, so the delta is folded in: W += (alpha / r) * B @ A.
    alpha: `lora_alpha` when given (one value for every adapter), otherwise per adapter by `lora_alpha_for` (text adapters 8,
    protein-encoder adapters config.aaseq_lora_alpha); an adapter whose alpha cannot be determined raises."""
    if not any(".lora_A." in k or ".base_model.model." in k for k in sd):
        return sd
    out, lora = {}, {}
    for k, v in sd.items():
        k2 = k.replace(".base_model.model.", ".").replace(".base_layer.", ".")
        m = re.match(r"^(.*)\.lora_([AB])\.([^.]+)\.weight$", k2)
        if m:
            lora.setdefault((m.group(1), m.group(3)), {})[m.group(2)] = v
        elif ".lora_" not in k2:
            out[k2] = v
    for (base, _adapter), ab in lora.items():
        if "A" not in ab or "B" not in ab:
            raise KeyError(f"incomplete LoRA pair for {base}")
        wkey = base + ".weight"
        if wkey not in out:
            raise KeyError(f"LoRA adapter for {base} but no base weight {wkey} in the checkpoint")
        r = ab["A"].shape[0]
        alpha = float(lora_alpha) if lora_alpha is not None else lora_alpha_for(base, config)
        scale = alpha / r
        out[wkey] = (out[wkey].float() + scale * (ab["B"].float() @ ab["A"].float())).to(out[wkey].dtype)
    return out


def infer_llama_config(llama_sd, **over):
    if "model.embed_tokens.weight" not in llama_sd:
        raise KeyError("text_encoder.model.model.embed_tokens.weight is not in the checkpoint: frozen tensors are missing (the reference "
                       "fills them from the pretrained Llama under pretrained_weights_dir; merge them into the state dict first)")
    from .engine import LlamaConfig
    emb = llama_sd["model.embed_tokens.weight"]
    n_layers = 1 + max(int(m.group(1)) for k in llama_sd for m in [re.match(r"model\.layers\.(\d+)\.", k)] if m)
    d = emb.shape[1]
    kv = llama_sd["model.layers.0.self_attn.k_proj.weight"].shape[0]
    ffn = llama_sd["model.layers.0.mlp.gate_proj.weight"].shape[0]
    head_dim = over.pop("head_dim", 128)
    return LlamaConfig(vocab=emb.shape[0], d=d, n_layers=n_layers, n_heads=d // head_dim, n_kv_heads=kv // head_dim, ffn=ffn, **over)


def infer_esm_config(esm_sd, n_heads=None, **over):
    from .engine import EsmConfig
    emb = esm_sd["esm.embeddings.word_embeddings.weight"]
    n_layers = 1 + max(int(m.group(1)) for k in esm_sd for m in [re.match(r"esm\.encoder\.layer\.(\d+)\.", k)] if m)
    d = emb.shape[1]
    ffn = esm_sd["esm.encoder.layer.0.intermediate.dense.weight"].shape[0]
    if n_heads is None:
        n_heads = 40 if d >= 2560 else 20        # ESM2 family
    return EsmConfig(d=d, n_layers=n_layers, n_heads=n_heads, ffn=ffn, vocab=emb.shape[0], **over)


def build_model(sd, config, tokenizer, device="cuda", max_new_tokens=256, esm_heads=None, esm_rope_math=None, **llama_over):
    """state dict + `ProCyonConfig` + tokenizer -> engine-backed `UnifiedProCyon` (the tail of `from_pretrained`,
    model_unified.py:1370-1382 `load_state_dict(strict=False)`)."""
    from .engine import BF16, MlpEngine
    from .model import ESM_PLM, LlamaPostTokenization, UnifiedProCyon
    parts = split_state_dict(sd)
    dev = torch.device(device)
    text_encoder = LlamaPostTokenization(parts["llama"], infer_llama_config(parts["llama"], **llama_over), dev, max_new_tokens)
    plm = None
    if parts["esm"] and not config.use_aaseq_embeddings:
        # fair-esm's rotary embedding rounds every product in model dtype, HF's computes in fp32 and rounds once:
        # the encoder a checkpoint was trained with is recognisable by its parameter names
        rope_math = esm_rope_math or ("model_dtype" if parts["esm_fair"] else "fp32_once")
        plm = ESM_PLM(parts["esm"], infer_esm_config(parts["esm"], n_heads=esm_heads, rope_math=rope_math), pooling_method=config.protein_pooling_opt,
                      protein_pooling_correction_option=config.protein_pooling_correction_option,
                      max_protein_len=config.max_protein_len, device=dev)
    def mk(layers):
        m = MlpEngine([(w.to(dev, BF16), None if b is None else b.to(dev, BF16)) for w, b in layers])
        if any(w.dtype == torch.float32 for w, _ in layers):
            m.src_f32 = list(layers)     # kept until .bfloat16(): the fp32 callers' arithmetic (engine_f32)
        return m
    P = parts["projectors"]
    tok_proj = {n[len("token_"):]: mk(l) for n, l in P.items() if n.startswith("token_")}
    tabs = {k: v.to(dev, BF16) for k, v in parts["tables"].items()}
    tabs_f32 = {k: v for k, v in parts["tables"].items() if v.dtype == torch.float32}
    model = UnifiedProCyon(config, text_encoder, tokenizer, protein_seq_encoder=plm, token_projectors=tok_proj,
                          aaseq_shared_projector=mk(P["aaseq_shared_projector"]) if "aaseq_shared_projector" in P else None,
                          aaseq_lm_projector=mk(P["aaseq_lm_projector"]) if "aaseq_lm_projector" in P else None,
                          protein_seq_embeddings=tabs.get("protein_seq_embeddings"), domain_embeddings=tabs.get("domain_embeddings"),
                          peptide_embeddings=tabs.get("peptide_embeddings"), protein_struct_embeddings=tabs.get("protein_struct_embeddings"),
                          drug_structure_embeddings=tabs.get("drug_structure_embeddings"))
    model._tables_f32 = tabs_f32
    return model


# ---------------------------------------------------------------------------------------------------------------------
# `UnifiedProCyon.from_pretrained` / `get_checkpoint_configs` (model_unified.py:1296-1406)

class ArgsShell:
    """Stand-in for the reference's pickled dataclasses (`procyon.training.training_args_IT.ModelArgs` / `DataArgs` /
    `TrainArgs`): `model_args.pt` & co. are `torch.save`d dataclass INSTANCES, so unpickling them needs those classes
    importable.  The shell receives the instance dictionary; attribute access works as on the original."""

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{k}={v!r}' for k, v in sorted(vars(self).items()))})"


class _ShellPickle:
    """`pickle_module` for torch.load: any class under `procyon.*` (or otherwise not importable) becomes an ArgsShell
    subclass named after it."""
    import pickle as _pickle
    __name__ = "procyon_amd.checkpoint._ShellPickle"

    class Unpickler(_pickle.Unpickler):
        def find_class(self, module, name):
            if module.split(".")[0] != "procyon":
                try:
                    return super().find_class(module, name)
                except (ImportError, AttributeError):
                    pass
            return type(name, (ArgsShell,), {"__module__": module})

    @staticmethod
    def load(f, **kw):
        return _ShellPickle.Unpickler(f, **kw).load()

    Pickler = _pickle.Pickler
    dump, dumps, loads = _pickle.dump, _pickle.dumps, _pickle.loads


def load_args(path):
    """one of model_args.pt / data_args.pt / training_args.pt"""
    return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_ShellPickle)


def get_checkpoint_configs(resume_from_checkpoint):
    """`UnifiedProCyon.get_checkpoint_configs` (model_unified.py:1396-1406): (data_args, model_args, train_args)."""
    import os
    return tuple(load_args(os.path.join(resume_from_checkpoint, n)) for n in ("data_args.pt", "model_args.pt", "training_args.pt"))


def config_from_model_args(args):
    """ModelArgs (training_args_IT.py:27-651) -> the fields of it the inference path reads."""
    from dataclasses import fields
    from .model import ProCyonConfig
    kw = {f.name: getattr(args, f.name) for f in fields(ProCyonConfig) if hasattr(args, f.name)}
    return ProCyonConfig(**kw)


def hf_tokenizer(path):
    """`_init_tokenizer` (model_unified.py:1088-1133): Llama tokenizer files + the eight added tokens IN THIS ORDER
    ([EXT] last: the embedding table has len(tokenizer) - 1 rows, :166)."""
    import transformers
    tok = transformers.AutoTokenizer.from_pretrained(path)
    tok.padding_side = "right"
    if tok.sep_token is None:
        tok.add_tokens("[CLS]")
        tok.sep_token = "[CLS]"
    if tok.pad_token is None:
        tok.add_tokens("[PAD]")
        tok.pad_token = "[PAD]"
    for t in ("<|protein|>", "[PROT]", "[ANSWER]", "<|struct|>", "<|drug|>", "[EXT]"):
        tok.add_tokens(t)
    return tok


# ---------------------------------------------------------------------------------------------------------------------
# DeepSpeed ZeRO stage-1/2 checkpoint -> one fp32 state dict
def get_fp32_state_dict_from_zero_checkpoint(checkpoint_dir, tag=None):
    """What `deepspeed.utils.zero_to_fp32.get_fp32_state_dict_from_zero_checkpoint` returns for a ZeRO stage 1 / 2 checkpoint
    (the reference's fallback when `txllm_model_ckpt.pt` is absent, model_unified.py:1336,1380-1382; ProCyon-Full ships as
    ZeRO-2, world 32, `global_step59469`).  DeepSpeed itself (pyproject pin 0.12.4) is not a dependency of this engine; the
    published algorithm is restated:

      <dir>/latest                                       text file holding the tag (e.g. "global_step59469") unless `tag` given
      <dir>/<tag>/mp_rank_00_model_states.pt             'module' (16-bit weights + buffers), 'buffer_names', 'param_shapes'
                                                         (one ordered name -> shape map per optimizer param group),
                                                         'shared_params', optional 'frozen_param_shapes' / '_fragments'
      <dir>/<tag>/[bf16_]zero_pp_rank_<r>_mp_rank_00_optim_states.pt   'optimizer_state_dict': 'zero_stage', 'partition_count',
                                                         'single_partition_of_fp32_groups' (rank r's slice of every group)

    Stage <= 2 keeps, per param group, ONE flat fp32 buffer holding the group's parameters back to back in `param_shapes`
    order, padded to a multiple of 2 x world size and cut into `world` equal slices.  Merge = concatenate the slices in rank
    order and cut the parameters back out.  Frozen parameters are stored whole; buffers come from 'module'; tied parameters
    are re-tied through 'shared_params'."""
    import glob
    import math
    import os
    if tag is None:
        latest = os.path.join(checkpoint_dir, "latest")
        if not os.path.isfile(latest):
            raise ValueError(f"Unable to find 'latest' file at {latest}")
        tag = open(latest).read().strip()
    d = os.path.join(checkpoint_dir, tag)
    if not os.path.isdir(d):
        raise FileNotFoundError(f"Directory '{d}' doesn't exist")
    load = lambda f: torch.load(f, map_location="cpu", weights_only=False, pickle_module=_ShellPickle)
    rank_of = lambda f: int(re.search(r"zero_pp_rank_(\d+)_", os.path.basename(f)).group(1))
    optim_files = sorted(glob.glob(os.path.join(d, "*_optim_states.pt")), key=rank_of)
    if not optim_files:
        raise FileNotFoundError(f"can't find '*_optim_states.pt' files in directory '{d}'")
    model_files = sorted(glob.glob(os.path.join(d, "*_model_states.pt")))
    if not model_files:
        raise FileNotFoundError(f"can't find '*_model_states.pt' files in directory '{d}'")
    ms = load(model_files[0])
    flat_groups, stage, world = [], None, None
    for f in optim_files:
        osd = load(f)["optimizer_state_dict"]
        stage = osd["zero_stage"] if stage is None else stage
        pc = osd["partition_count"]
        world = max(pc) if isinstance(pc, (list, tuple)) else pc
        flat_groups.append(osd["single_partition_of_fp32_groups"])
    if stage > 2:
        raise NotImplementedError(f"ZeRO stage {stage} checkpoints are not supported (ProCyon checkpoints are stage 2)")
    if world != len(optim_files):
        raise ValueError(f"Expected {world} of '*_optim_states.pt' under '{d}' but found {len(optim_files)} files")
    out = {}
    for name in ms.get("buffer_names", []):                                   # buffers, as fp32
        out[name] = ms["module"][name].float()
    for name in (ms.get("frozen_param_shapes") or {}):                        # stage <= 2: frozen parameters are stored whole
        out[name] = ms["frozen_param_fragments"][name]
    align = 2 * world
    up = lambda n: align * math.ceil(n / align)
    param_shapes = ms["param_shapes"]
    if isinstance(param_shapes, dict):
        param_shapes = [param_shapes]
    for gi, shapes in enumerate(param_shapes):
        full = torch.cat([fg[gi] for fg in flat_groups], 0)
        offset = 0
        for name, shape in shapes.items():
            numel = int(math.prod(shape))
            out[name] = full.narrow(0, offset, numel).view(tuple(shape))
            offset += numel
        if up(offset) != up(full.numel()):
            raise ValueError(f"consumed {offset} numels out of {full.numel()} - something is wrong")
    for pair in ms.get("shared_params", []) or []:
        if pair[1] in out:
            out[pair[0]] = out[pair[1]]
    return out


def from_pretrained(*, pretrained_weights_dir=None, checkpoint_dir=None, model=None, config_only=False, config=None,
                    state_dict_relative_path="txllm_model_ckpt.pt", strict_load=False, load_plm_directly=False,
                    protein_pooling_correction_option=False, tokenizer=None, device="cuda", max_new_tokens=256, **engine_kw):
    """`UnifiedProCyon.from_pretrained` (model_unified.py:1296-1394) -> (model, config).

    Same keyword contract for what inference uses.  Differences: `model=` (update an existing module in place) is not
    supported -- the engine packs weights at construction; a checkpoint that only holds DeepSpeed ZeRO stage-2 shards is merged
    by `get_fp32_state_dict_from_zero_checkpoint` above; `tokenizer=` may supply
    the tokenizer object when the Llama tokenizer files are not at $LLAMA3_PATH / pretrained_weights_dir."""
    import os
    if model is not None:
        raise NotImplementedError("from_pretrained(model=...): in-place reload is not supported by the engine-backed model")
    config_checkpoint = load_args(os.path.join(checkpoint_dir, "model_args.pt"))
    data_args = load_args(os.path.join(checkpoint_dir, "data_args.pt"))   # noqa: F841 (read like the reference; paths are re-rooted there)
    if config is None:
        config = config_checkpoint
    if config_only:
        return None, config
    config.n_model_pieces = 1
    config.model_splitting = False
    if load_plm_directly and getattr(config, "use_aaseq_embeddings", False):
        # the checkpoint was trained on cached ESM embeddings: run the encoder they came from instead (:1343-1366)
        assert config.protein_seq_embeddings_path is not None
        aaseq_type, nparams_name, pooling_method = os.path.basename(config.protein_seq_embeddings_path).split(".")[0].split("_")
        assert pooling_method in ("max", "mean")
        if nparams_name == "esm2-3b":
            nparams = "3b"
        elif nparams_name == "esm-650m":
            nparams = "650m"
        else:
            raise NotImplementedError("Invalid number of parameters")
        config.use_aaseq_embeddings = False
        config.freeze_protein_encoder = "all"
        config.protein_encoder_num_params = nparams
        config.protein_pooling_opt = pooling_method
        config.long_protein_strategy = "split"
        config.max_protein_len = 1024
        config.protein_enc_batch_limit = None
        config.protein_pooling_correction_option = protein_pooling_correction_option
    sd_path = os.path.join(checkpoint_dir, state_dict_relative_path)
    if os.path.exists(sd_path):                                  # consolidated locally (model_unified.py:1376-1379)
        sd = torch.load(sd_path, map_location="cpu", weights_only=False, pickle_module=_ShellPickle)
    else:                                                        # DeepSpeed ZeRO shards (:1380-1382)
        sd = get_fp32_state_dict_from_zero_checkpoint(checkpoint_dir)
    sd = merge_lora(sd, config=config_checkpoint)   # text adapters: alpha 8; protein-encoder adapters: aaseq_lora_alpha / aaseq_lora_r
    if not any(k.startswith("protein_seq_encoder.model.") for k in sd) and not getattr(config, "use_aaseq_embeddings", False):
        # frozen encoder loaded from the fair-esm release file next to the other pretrained weights (esm.py:378-398)
        name = {"650m": "esm2_t33_650M_UR50D.pt", "3b": "esm2_t36_3B_UR50D.pt", "35m": "esm2_t12_35M_UR50D.pt",
                "8m": "esm2_t6_8M_UR50D.pt"}.get(str(getattr(config, "protein_encoder_num_params", "650m")).lower())
        path = None if name is None or pretrained_weights_dir is None else os.path.join(pretrained_weights_dir, name)
        if path is None or not os.path.exists(path):
            raise FileNotFoundError(f"protein encoder weights are neither in the checkpoint nor at {path}")
        esm_sd = torch.load(path, map_location="cpu", weights_only=False)["model"]
        esm_sd = {re.sub(r"^(encoder\.)?(sentence_encoder\.)?", "", k): v for k, v in esm_sd.items()}
        sd = dict(sd, **{"protein_seq_encoder.model." + k: v for k, v in esm_sd.items()})
    if tokenizer is None:
        fname = str(getattr(config, "text_encoder_fname", ""))
        path = os.getenv("LLAMA3_PATH") if "llama-3" in fname.lower() else os.path.join(pretrained_weights_dir or "", fname)
        if not path or not os.path.exists(path):
            raise FileNotFoundError(f"Llama tokenizer files not found at {path!r}; pass tokenizer=")
        tokenizer = hf_tokenizer(path)
    m = build_model(sd, config_from_model_args(config), tokenizer, device=device, max_new_tokens=max_new_tokens, **engine_kw)
    return m, config
