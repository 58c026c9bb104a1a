"""CPU side of the executed shim test (tests/test_gpu_r_shim.py has the device runs): the reference-side shim compiled
against tests/r_mini/rmini.c and called with the twelve .Call arguments on a box WITHOUT a device -- the core answers
GSLNLS_E_NODEVICE and the shim must hand the call to the original C_nls untouched ("the GSL path still
exists"), without touching a byte it should not (p = 100: beyond every fixed-size scratch of the shim)."""
import ctypes as C

import numpy as np
import pytest

import test_gpu_r_shim as T
from test_gpu_function import gaussians


def test_without_a_device_the_shim_falls_through_to_c_nls():
    import gslnls_amd
    from gslnls_amd import _lib
    if _lib.lib().gslnls_device_count() >= 1:
        pytest.skip("a device is visible: tests/test_gpu_r_shim.py runs the shim on it")
    L = T.rshim.__wrapped__(gslnls_amd)
    x, y, model, jac, start, truth = gaussians(33, 300, 4233)
    names = ["th%d" % (k + 1) for k in range(len(start))]
    ans, seen, keep = T._call(L, gslnls_amd, model, jac, y, start, names)
    assert L.rm_fell_through() == 1 and L.rm_type(ans) == 0  # (the stand-in C_nls returns R_NilValue)
    assert not seen  # no closure was evaluated: the refusal came before any work
