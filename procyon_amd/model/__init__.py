"""Host-side mirror of the reference's model API for the hot path:
`UnifiedProCyon`, `LlamaPostTokenization`, `ESM_PLM`, `create_mlp`/`left_pad_tensors` -- same names,
argument meaning and error behaviour as /root/reference/procyon/model/*, with every tensor op running in
libpcy.so's HIP kernels."""
from .esm import ESM_PLM  # noqa: F401
from .model_unified import ProCyonConfig, UnifiedProCyon  # noqa: F401
from .pmc_llama import LlamaPostTokenization  # noqa: F401
