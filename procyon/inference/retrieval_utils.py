"""`procyon.inference.retrieval_utils` (reference: procyon/inference/retrieval_utils.py:21-201): start-up and one retrieval
query of the protein-retrieval service / CLI (procyon/app/main.py:9, scripts/protein_retrieval_disease_pheno.py:8).

Same signatures, same exceptions, same return values.  $CHECKPOINT_PATH, $HF_TOKEN etc. are read when a function runs, not
at import.  The ranking runs on the model's device through the engine (`get_proteins_from_embedding`); the reference's
`.cuda()` calls (quirk Q14) have no counterpart."""
import os
from pathlib import Path
from typing import Optional, Tuple

import pandas as pd
import torch

from procyon.data.inference_utils import create_input_retrieval, get_proteins_from_embedding
from procyon.evaluate.framework.utils import move_inputs_to_device
from procyon.inference.settings import logger
from procyon.model.model_unified import UnifiedProCyon
from procyon.training.train_utils import DataArgs


def _ckpt_dir():
    path = os.getenv("CHECKPOINT_PATH")
    if not path:
        raise EnvironmentError("CHECKPOINT_PATH environment variable not set")
    return os.path.expanduser(path)


def load_model_onto_device() -> Tuple[UnifiedProCyon, torch.device, DataArgs]:
    """(model, device, data_args) from $CHECKPOINT_PATH (retrieval_utils.py:76-106): bf16, eval mode, on the GPU."""
    logger.info("Now loading pretrained model")
    ckpt = _ckpt_dir()
    from procyon_amd.checkpoint import load_args
    data_args = load_args(os.path.join(ckpt, "data_args.pt"))
    model, _ = UnifiedProCyon.from_pretrained(checkpoint_dir=ckpt, pretrained_weights_dir=os.getenv("PRETRAINED_WEIGHTS_DIR"))
    model.bfloat16()
    model.eval()
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    model.to(device)
    logger.info("Done loading model and applying it to compute device")
    return model, device, data_args


def startup_retrieval(inference_bool: bool = True):
    """-> (model | None, device | None, data_args | None, all_protein_embeddings fp32 [N, D]) (retrieval_utils.py:21-73)."""
    logger.info("Now running startup functions for protein retrieval")
    if os.getenv("HF_TOKEN"):
        try:                                   # the hub login only matters when weights are fetched from the hub
            from huggingface_hub import login as hf_login
            hf_login(token=os.getenv("HF_TOKEN"))
        except Exception as e:                  # offline boxes: the checkpoint is local
            logger.debug(f"huggingface hub login skipped: {e}")
    if inference_bool:
        model, device, data_args = load_model_onto_device()
    else:
        model = device = data_args = None
    emb, _ids = torch.load(os.path.join(_ckpt_dir(), "protein_target_embeddings.pkl"), map_location="cpu", weights_only=False)
    emb = emb.float()
    logger.debug(f"shape of precalculated embeddings matrix: {emb.shape}")
    logger.info("Done running startup functions for protein retrieval")
    return model, device, data_args, emb


def _one_of(infile, text, what):
    if infile is not None:
        if text is not None:
            raise ValueError(f"Only one of {what}_infile and {what} can be provided.")
        with open(infile, "r") as f:
            text = f.read()
    elif text is None:
        raise ValueError(f"Either {what}_infile or {what} must be provided.")
    return text.replace("\n", " ")


def do_retrieval(model, data_args, device, instruction_source_dataset: str, all_protein_embeddings: torch.Tensor,
                 inference_bool: bool = True, task_desc_infile: Path = None, disease_desc_infile: Path = None,
                 task_desc: str = None, disease_desc: str = None) -> Optional[pd.DataFrame]:
    """Rank every protein of `all_protein_embeddings` for one disease description (retrieval_utils.py:109-201):
    data frame with columns uniprot_id, name, sim_score, best first; None when inference_bool is False."""
    logger.info("Now performing protein retrieval")
    if instruction_source_dataset not in ["disgenet", "omim"]:
        raise ValueError('instruction_source_dataset must be either "disgenet" or "omim"')
    task_desc = _one_of(task_desc_infile, task_desc, "task_desc")
    disease_desc = _one_of(disease_desc_infile, disease_desc, "disease_desc")
    if not inference_bool:
        return None
    inputs = create_input_retrieval(input_description=disease_desc, data_args=data_args, task_definition=task_desc,
                                    instruction_source_dataset=instruction_source_dataset, instruction_source_relation="all",
                                    aaseq_type="protein", icl_example_number=1)
    inputs = move_inputs_to_device(inputs, device=device)
    with torch.no_grad():
        model_out = model(inputs=inputs, retrieval=True, aaseq_type="protein")
    df = get_proteins_from_embedding(all_protein_embeddings, model_out, top_k=None)
    logger.info("Done performing protein retrieval")
    return df
