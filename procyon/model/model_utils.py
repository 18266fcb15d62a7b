"""`procyon.model.model_utils` (reference: procyon/model/model_utils.py:13-41,151-170)."""
from procyon_amd.model.model_utils import create_mlp, left_pad_tensors  # noqa: F401
