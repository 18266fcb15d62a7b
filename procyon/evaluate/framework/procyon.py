"""`procyon.evaluate.framework.procyon` (reference: procyon/evaluate/framework/procyon.py): the three evaluation plugins."""
from procyon_amd.evaluate import ProcyonCaptionEval, ProcyonQAEval, ProcyonRetrievalEval  # noqa: F401
