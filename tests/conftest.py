import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """npz -> dict of torch tensors; uint16 arrays are raw bf16 bit patterns."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        if a.dtype == np.uint16:
            out[k] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
        else:
            out[k] = torch.from_numpy(np.array(a))
    return out


def assert_bf16_close(a, b, what="", max_frac=0.02, ulps=2):
    """bf16 tensors produced by two fp32-accumulating implementations: identical except for
    rare 1-ulp rounding flips where the fp32 sums straddle a bf16 rounding boundary."""
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if torch.equal(a, b):
        return
    af, bf = a.float(), b.float()
    diff = (af - bf).abs()
    tol = ulps * 2.0 ** -8 * torch.maximum(af.abs(), bf.abs()) + 1e-30
    frac = (diff > 0).float().mean().item()
    assert bool((diff <= tol).all()), f"{what}: max diff {diff.max().item()} beyond {ulps} bf16 ulp"
    assert frac <= max_frac, f"{what}: {frac:.4f} of elements differ"


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
