"""Mirror of the path-relevant helpers of /root/reference/procyon/model/model_utils.py."""
from __future__ import annotations

import torch

from ..engine import MlpEngine


def create_mlp(weights, ctx=None):
    """Engine-backed `create_mlp` stack (model_utils.py:13-41) from its Linear weights
    [(W, b|None), ...]: n == 1 is the bias-free single Linear (:26-27); otherwise
    Linear -> (Dropout = identity in eval) -> GELU(erf) ... -> Linear."""
    if len(weights) == 1 and weights[0][1] is not None:
        raise ValueError("a 1-layer create_mlp has no bias (model_utils.py:26-27)")
    return MlpEngine(weights, ctx)


def left_pad_tensors(tensors, pad_value=0):
    """`left_pad_tensors` (model_utils.py:151-170): pad 1-D id tensors on the left, float 0/1 masks."""
    max_length = max(t.size(0) for t in tensors)
    padded, masks = [], []
    for t in tensors:
        p = max_length - t.size(0)
        padded.append(torch.cat([torch.full((p,), pad_value), t]))
        masks.append(torch.cat([torch.zeros(p), torch.ones(t.size(0))]))
    return torch.stack(padded), torch.stack(masks)
