"""The C-ABI library loads on a CPU-only box and exports every symbol include/gslnls_core.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "gslnls_core.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gslnls_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from gslnls_amd import _lib
    L = _lib.lib()
    names = _declared()
    assert len(names) >= 10
    for nm in names:
        assert hasattr(L, nm), nm
    assert sorted(_lib.symbols()) == names  # the Python binding covers the whole header


def test_no_compute_without_gpu_fails_loudly():
    """no silent CPU fallback: without a HIP device the entry points refuse to run"""
    import numpy as np
    import pytest
    from gslnls_amd import _lib, DenseProblem
    L = _lib.lib()
    if L.gslnls_device_count() > 0:
        pytest.skip("a GPU is present")
    x = np.linspace(0, 3, 64)
    with pytest.raises(_lib.GslnlsDeviceError):
        DenseProblem(1, 3, x, np.exp(-x))


def test_product_never_touches_the_oracle():
    """nothing under gslnls_amd/ or include/ may import, link or name the oracle"""
    bad = []
    for base in ("gslnls_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hpp", ".h", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"gslref|oracle/|libgslref", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
