"""`procyon.model.esm` (reference: procyon/model/esm.py:504-558): the ESM2 wrapper with the engine underneath."""
from procyon_amd.model.esm import ESM_PLM  # noqa: F401
