"""`procyon.evaluate.framework.utils.move_inputs_to_device` (reference: procyon/evaluate/framework/utils.py:46-61)."""
from procyon_amd.evaluate import move_inputs_to_device  # noqa: F401
