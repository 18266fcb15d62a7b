"""`procyon.model.pmc_llama` (reference: procyon/model/pmc_llama.py:546-596)."""
from procyon_amd.model.pmc_llama import LlamaPostTokenization  # noqa: F401
