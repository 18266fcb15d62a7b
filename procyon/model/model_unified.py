"""`procyon.model.model_unified` (reference: procyon/model/model_unified.py): `UnifiedProCyon` and the two module-level helpers
callers import from here, backed by `procyon_amd.model`."""
from procyon_amd.model.model_unified import ProCyonConfig, UnifiedProCyon, mask_before, multi_replace_tokens  # noqa: F401
