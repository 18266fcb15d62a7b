"""`procyon.model.model_utils` (reference: procyon/model/model_utils.py:13-41,151-170)."""
from procyon_amd.model.model_utils import EngineMlp, create_mlp, create_mlp_from_weights, left_pad_tensors  # noqa: F401
