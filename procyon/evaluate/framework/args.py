"""`procyon.evaluate.framework.args.EvalArgs` (reference: procyon/evaluate/framework/args.py): the evaluation-run arguments the
plugin constructors receive (`model_zoo[task][model_type](model_config, eval_args, model_args, device)`, core.py:216).  Same
field names and defaults; only the fields are declared, the YAML / command-line plumbing of the reference's runner is not part
of the engine."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class EvalArgs:
    from_yaml: Optional[str] = None
    output_dir: Optional[str] = None
    models_config_yml: Optional[str] = None
    batch_size: int = 16
    num_workers: int = 0
    retrieval_eval_all_aaseqs: bool = True
    retrieval_use_cached_target_embeddings: int = True
    retrieval_balanced_metrics_num_samples: Optional[int] = None
    retrieval_balanced_metrics_neg_per_pos: int = 1
    retrieval_auroc_auprc_per_query: bool = True
    model_args_from_checkpoint: str = ""
    data_args_from_checkpoint: str = ""
    override_model_data_args_yml: Optional[str] = None
    filter_training_pairs: bool = True
    separate_splits: bool = True
    keep_splits_union: bool = True
    use_cached_results: bool = True
    seed: int = 42
    qa_num_samples: Optional[int] = None
    caption_max_len: int = 64
