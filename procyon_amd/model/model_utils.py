"""Mirror of the path-relevant helpers of /root/reference/procyon/model/model_utils.py."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..engine import MlpEngine


def create_mlp_from_weights(weights, ctx=None):
    """Engine-backed `create_mlp` stack (model_utils.py:13-41) from its Linear weights
    [(W, b|None), ...]: n == 1 is the bias-free single Linear (:26-27); otherwise
    Linear -> (Dropout = identity in eval) -> GELU(erf) ... -> Linear."""
    if len(weights) == 1 and weights[0][1] is not None:
        raise ValueError("a 1-layer create_mlp has no bias (model_utils.py:26-27)")
    return MlpEngine(weights, ctx)


class EngineMlp(nn.Sequential):
    """What the reference's `create_mlp` returns -- an `nn.Sequential` of Linear / Dropout / GELU with the reference's module indices, so
    a checkpoint's `token_projectors.aaseq.{0,3,6}.{weight,bias}` keys load with `load_state_dict` -- whose forward runs the fused HIP
    projector (`pcy_mlp_forward`, or its fp32 twin for fp32 parameters) on the current parameters.  Eval mode on the GPU only: there is no
    CPU or training path behind it (it raises instead of falling back)."""

    def __init__(self, *mods):
        super().__init__(*mods)
        self._eng, self._eng_key = None, None

    def _engine(self):
        lin = [m for m in self if isinstance(m, nn.Linear)]
        key = tuple((m.weight.data_ptr(), m.weight._version, None if m.bias is None else (m.bias.data_ptr(), m.bias._version), m.weight.dtype)
                    for m in lin)
        if key != self._eng_key:
            w0 = lin[0].weight
            if not w0.is_cuda:
                raise RuntimeError("the engine-backed projector runs on the GPU: move the module with .to(device) first")
            layers = [(m.weight.detach().contiguous(), None if m.bias is None else m.bias.detach().contiguous()) for m in lin]
            if w0.dtype == torch.bfloat16:
                self._eng = MlpEngine(layers)
            elif w0.dtype == torch.float32:
                from ..engine_f32 import MlpEngineF32
                self._eng = MlpEngineF32(layers)
            else:
                raise RuntimeError(f"projector parameters of dtype {w0.dtype}: bfloat16 or float32 expected")
            self._eng_key = key
        return self._eng

    def forward(self, x):
        if self.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in self):
            raise RuntimeError("the engine-backed projector is an inference path: call .eval() (dropout is the identity there)")
        if not x.is_cuda:
            raise RuntimeError("the engine-backed projector runs on the GPU (no CPU fallback)")
        return self._engine()(x)


def create_mlp(n_layers, in_features, out_features, hidden_features=256, dropout_rate=0.25):
    """`create_mlp` with the reference's signature and module layout (/root/reference/procyon/model/model_utils.py:13-41): n_layers == 1 is
    one bias-free Linear; otherwise Linear(in, hidden) [-> Dropout] -> GELU ... -> Linear(hidden, out).  Returns an `EngineMlp`: the same
    state-dict keys as the reference's Sequential, the forward on the HIP projector.  (`create_mlp_from_weights` is the form that takes the
    Linear weights directly.)"""
    if n_layers == 1:
        return EngineMlp(nn.Linear(in_features, out_features, bias=False))
    layers = []
    for i in range(n_layers):
        in_size = hidden_features if i > 0 else in_features
        if i < n_layers - 1:
            layers.append(nn.Linear(in_size, hidden_features))
            if dropout_rate is not None:
                layers.append(nn.Dropout(dropout_rate))
            layers.append(nn.GELU())
        else:
            layers.append(nn.Linear(in_size, out_features))
    return EngineMlp(*layers)


def left_pad_tensors(tensors, pad_value=0):
    """`left_pad_tensors` (model_utils.py:151-170): pad 1-D id tensors on the left, float 0/1 masks."""
    max_length = max(t.size(0) for t in tensors)
    padded, masks = [], []
    for t in tensors:
        p = max_length - t.size(0)
        padded.append(torch.cat([torch.full((p,), pad_value), t]))
        masks.append(torch.cat([torch.zeros(p), torch.ones(t.size(0))]))
    return torch.stack(padded), torch.stack(masks)
