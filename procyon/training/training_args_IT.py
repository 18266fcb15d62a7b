"""`procyon.training.training_args_IT.{ModelArgs, DataArgs, TrainArgs}` (reference: procyon/training/training_args_IT.py).

Checkpoints hold pickled INSTANCES of these dataclasses (`model_args.pt`, `data_args.pt`, `training_args.pt`), so the classes
must be importable under exactly these paths for `torch.load` to rebuild them; unpickling restores the instance dictionary
without calling `__init__`, so every field the checkpoint was written with comes back whatever is declared here.  Declared
are the fields the inference path reads, with the reference's defaults (the `field(default=...)` values at the cited lines),
for callers that construct the objects themselves."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class ModelArgs:
    protein_encoder_num_params: str = "650m"          # :32
    max_protein_len: int = 1024                       # :65
    long_protein_strategy: str = "split"              # :77
    protein_pooling_opt: str = "max"                  # :103
    protein_enc_batch_limit: Optional[int] = None     # :115
    text_encoder_fname: str = "llama-3-8b"            # :129
    max_text_len: int = 2048                          # :148
    ret_token_access: str = "all"                     # :173
    attention_type: str = "vanilla"                   # :209
    use_aaseq_embeddings: bool = True                 # :335
    use_drug_embeddings: bool = False                 # :341
    use_protein_struct: bool = False                  # :347
    protein_struct_dropout: float = 0.0
    roll_num: int = 0                                 # :564
    protein_pooling_correction_option: bool = False   # :640
    protein_seq_embeddings_path: Optional[str] = None
    protein_task_spc_lora: bool = False
    lora_specific_style: str = "none"


@dataclass
class DataArgs:
    data_dir: Optional[str] = None
    qa_subset_version: Optional[int] = 1
    caption_subset_version: Optional[int] = 1
    retrieval_subset_version: Optional[int] = 1
    use_caption: bool = True
    use_qa: bool = True
    use_retrieval: bool = True


@dataclass
class TrainArgs:
    seed: int = 42
    resume_from_checkpoint: Optional[str] = None
    deepspeed_config: Optional[str] = None
