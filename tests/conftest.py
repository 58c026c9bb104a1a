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


# ---- parity evidence: what each oracle comparison actually measured ------------------------------------------------------
# Tests call record_parity(label, value); the table is printed at the end of the session (always, not only under -rA), so a
# reader sees how much of each tolerance is used instead of only "passed".
PARITY = []


def record_parity(label, value, bar=None):
    PARITY.append((str(label), float(value), bar))


def rel_err(a, b):
    """max relative difference of two coefficient vectors; recorded under the running test's name"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    v = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
    label = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].replace("tests/", "")
    record_parity(label, v)
    return v


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not PARITY:
        return
    tr = terminalreporter
    tr.section("measured relative error vs the oracle (record_parity)")
    width = max(len(k) for k, _, _ in PARITY)
    for k, v, bar in PARITY:
        tr.write_line("%s  %.3e%s" % (k.ljust(width), v, "" if bar is None else "   (bar %.0e)" % bar))
    worst = max(PARITY, key=lambda t: t[1])
    tr.write_line("worst: %s %.3e over %d comparisons" % (worst[0], worst[1], len(PARITY)))


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
