import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def gslref():
    import gslref as G  # oracle/gslref.py (builds oracle/_build/libgslref.so with gcc on first use)
    G.lib()
    return G


@pytest.fixture(scope="session")
def readme():
    return load_golden("readme_traces.json")


@pytest.fixture(scope="session")
def nist():
    return {q["name"]: q for q in load_golden("nist_formula_problems.json")}


@pytest.fixture(scope="session")
def mgh():
    return {q["name"]: q for q in load_golden("mgh_function_problems.json")}


@pytest.fixture(scope="session")
def pins():
    return load_golden("unit_test_pins.json")


@pytest.fixture(scope="session")
def hostsim():
    import hostsim_py
    return hostsim_py


def c2_data(n, seed=20250927):
    """BASELINE config C2 inputs: x_i = 3(i-1)/(n-1), theta* = (5, 1.5, 1),
    noise 0.25 N(0,1) from numpy PCG64(seed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = 3.0 * np.arange(n, dtype=np.float64) / (n - 1)
    y = 5.0 * np.exp(-1.5 * x) + 1.0 + 0.25 * rng.standard_normal(n)
    return x, y
