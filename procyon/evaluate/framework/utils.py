"""`procyon.evaluate.framework.utils.{move_inputs_to_device, compare_and_warn_model_args}` (reference: procyon/evaluate/framework/utils.py:46-61,103-141)."""
from procyon_amd.evaluate import compare_and_warn_model_args, move_inputs_to_device  # noqa: F401
