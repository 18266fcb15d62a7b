"""CPU: the C-ABI shared library loads and exports every symbol include/pcy.h declares (no compute
calls without a GPU), and the product path fails loudly without a device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pcy.h")).read()
    return sorted(set(re.findall(r"\b(pcy_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from procyon_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, f"{n} declared in pcy.h but not bound"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.pcy_abi_version() == _lib.ABI_VERSION


def test_no_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from procyon_amd import _lib
    from procyon_amd.engine import Context
    with pytest.raises(_lib.PcyError):
        Context.get()


def test_product_never_imports_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "procyon_amd")):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", s, re.M), f
