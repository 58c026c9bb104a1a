"""GPU parity of multi-start (BASELINE config C4: NIST BoxBOD, thousands of Sobol starts) through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

TOL = float(np.finfo(float).eps ** 0.25)


@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1
    return gslnls_amd


def _boxbod(nist):
    q = nist["BoxBOD"]
    return q, np.array(q["data"]["x"]), np.array(q["data"]["y"]), np.array(list(q["target"].values()))


def _range_transform(u, l0, l1, kd):
    """the power-law map of a unit-cube point into [l0, l1] (src/nls_mstart.c:46-71)"""
    v = l0 + (l1 - l0) * u
    if l0 > 0.0:
        return (np.power(v - l0 + 1.0, kd) - 1.0) / kd + l0
    if l1 < 0.0:
        return -(np.power(-v + l1 + 1.0, kd) - 1.0) / kd + l1
    return np.where(v > 0.0, (np.power(np.abs(v) + 1.0, kd) - 1.0) / kd, -(np.power(np.abs(v) + 1.0, kd) - 1.0) / kd)


def test_c4_concentration_batch_matches_the_oracle_point_by_point(amd, gslref, hostsim, nist):
    """8192 Sobol starts in b1 in [1,500], b2 in [0.01,5], mstart_p = 5 LM iterations each.  Every
    per-point record of the GPU lanes is compared with the ORACLE: the start point (gslref.sobol + range transform),
    and -- for every 8th point -- the oracle's own single-start run of the concentration fit (driver2 with maxiter =
    mstart_p, gtol = 1e-3: src/nls_mstart.c:79-92): where it ended, ssr, iterations, status.  (The device headers
    compiled for the host, tests/hostsim, are checked against the same records as a by-product.)"""
    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control, pack_control
    q, x, y, tgt = _boxbod(nist)
    N = 8192
    ctrl = gsl_nls_control(solver="cholesky")
    ci, cd = pack_control(ctrl, "lm")
    prob = amd.DenseProblem(2, 2, x, y)
    ranges = np.array([1.0, 500.0, 0.01, 5.0])
    kd = np.array([0.75, 0.75])
    K = _lib.lib().gslnls_mstart_record_size(2)
    rec = np.zeros((N, K))
    ms = C.c_float(0)
    rc = _lib.lib().gslnls_mstart_batch(prob._h, 1, ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), 0, N,
                                        0, N, 5, 1e-6, ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), None,
                                        rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
    assert rc == 0
    prob.close()
    # ---- start points: oracle's sequential Sobol generator + the reference's range map ----
    pts = gslref.sobol(2, N)
    x0 = np.stack([_range_transform(pts[:, k], ranges[2 * k], ranges[2 * k + 1], kd[k]) for k in range(2)], axis=1)
    assert np.allclose(rec[:, 4:6], x0, rtol=1e-13)
    # ---- the fits: the oracle's single-start LM from the same point, 5 iterations, gtol 1e-3 ----
    octrl = gslref.control(solver="cholesky", maxiter=5, gtol=1e-3)
    rd = dict(model=gslref.MODEL_MISRA1A, x=x, y=y)
    checked = 0
    for i in range(0, N, 8):
        if not rec[i, 8] > 1e-6:
            continue   # det filter: no fit for this point (decisions are compared for all points below)
        o = gslref.nls(6, 2, x0[i], rowdata=rd, use_jac=True, ctrl=octrl)
        assert int(rec[i, 11]) == o["niter"] and int(rec[i, 12]) == o["conv"], (i, rec[i], o["niter"], o["conv"])
        if o["conv"] in (0, 11) and np.isfinite(o["ssr"]):
            assert np.allclose(rec[i, 0:2], o["par"], rtol=1e-7, atol=1e-10), (i, rec[i, 0:2], o["par"])
            assert abs(rec[i, 7] - o["ssr"]) <= 1e-9 * abs(o["ssr"]), (i, rec[i, 7], o["ssr"])
        checked += 1
    assert checked > 900
    # ---- det(J^T J) at the start point decides whether a point is fitted at all (src/nls_utils.c:23-73) ----
    e = np.exp(-np.outer(x0[:, 1], x))                      # N x 6
    J0, J1 = 1.0 - e, x0[:, :1] * x[None, :] * e
    det0 = np.sum(J0 * J0, axis=1) * np.sum(J1 * J1, axis=1) - np.sum(J0 * J1, axis=1) ** 2
    clear = np.abs(det0 - 1e-6) > 1e-9                      # away from the threshold
    assert np.array_equal((rec[:, 8] > 1e-6)[clear], (det0 > 1e-6)[clear])
    # ---- by-product: the host build of the device headers gives the same records ----
    ref = hostsim.mstart_batch_misra(x, y, ranges, kd, 0, N, 5, 1e-6, ci, cd, jac=1)
    assert np.array_equal(rec[:, 11], ref[:, 11]) and np.array_equal(rec[:, 12], ref[:, 12])
    # a visible fraction of the starts lands in the wrong basin (ssr ~ 9771.5), the rest near ssr 1168.009
    sel = (rec[:, 8] > 1e-6) & np.isfinite(rec[:, 7])
    good = np.sum(np.abs(rec[sel, 7] - 1168.0088766) < 1.0)
    assert good > 0.3 * N and np.sum(rec[sel, 7] > 5000) > 0


@pytest.mark.parametrize("model_id,p,ranges", [
    (1, 3, [0.5, 10.0, -2.0, 3.0, -4.0, -1.0]),                                      # ExpDecay: all three branches of the map
    (4, 8, [90, 110, 0.001, 0.1, 90, 110, 50, 80, 10, 40, 60, 80, 150, 200, 10, 30]),  # Gauss1 family
])
def test_start_points_for_p3_and_p8_follow_the_oracle_sobol(amd, gslref, model_id, p, ranges):
    """multi-start with p = 3 and p = 8 parameters draws dimensions 3..8 of the Sobol sequence (src/nls.c:277-280): the
    sampled points of a device batch equal gslref.sobol + the reference's range map, from a non-zero draw offset too
    (Bratley-Fox numbers of those dimensions are pinned structurally only: tests/test_qrng_pins.py)"""
    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control, pack_control
    n = 40
    rng = np.random.default_rng(5)
    x = np.linspace(1.0, 250.0, n)
    y = rng.uniform(1.0, 2.0, n)
    prob = amd.DenseProblem(model_id, p, x, y)
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
    rg = np.array(ranges, dtype=np.float64)
    kd = np.linspace(0.5, 1.0, p)
    K = _lib.lib().gslnls_mstart_record_size(p)
    for first, N in ((0, 1500), (70000, 300)):
        rec = np.zeros((N, K))
        ms = C.c_float(0)
        rc = _lib.lib().gslnls_mstart_batch(prob._h, 1, rg.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), first, N,
                                            0, N, 1, 1e300, ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), None,
                                            rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
        assert rc == 0
        pts = gslref.sobol(p, N, skip=first)
        x0 = np.stack([_range_transform(pts[:, k], rg[2 * k], rg[2 * k + 1], kd[k]) for k in range(p)], axis=1)
        # (device pow() and glibc's differ in the last ulp, and (pow(1 + d) - 1) / kd cancels for small d)
        assert np.allclose(rec[:, 2 * p:3 * p], x0, rtol=1e-12, atol=0), np.max(np.abs(rec[:, 2 * p:3 * p] / x0 - 1))
    prob.close()


CASES = [
    dict(start=dict(b1=[200, 250], b2=[0, 1])),                                       # 4.1.1
    dict(start=dict(b1=[200, 250], b2=1), weights=np.full(6, 10.0)),                  # 4.1.3
    dict(start=dict(b1=200, b2=np.nan), lower=dict(b2=0)),                            # 4.1.4
    dict(start=dict(b1=np.nan, b2=0.5), jac=True, upper=dict(b2=1)),                  # 4.1.5
    dict(start=dict(b1=[200, 250], b2=np.nan), jac=True, lower=dict(b2=0), upper=dict(b2=1)),  # 4.1.7
]


@pytest.mark.parametrize("case", CASES)
def test_unit_test_4_1_multistart(amd, gslref, nist, case):
    """unit_tests_gslnls.R:137-156: gsl_nls() with start ranges / missing starts reaches the certified BoxBOD
    optimum; the bookkeeping equals the oracle's sequential procedure"""
    q, x, y, tgt = _boxbod(nist)
    kw = dict(case)
    start = kw.pop("start")
    fit = amd.gsl_nls(q["formula"], data=q["data"], start=start,
                      control=dict(mstart_n=5, mstart_q=1, mstart_r=1.1, solver="cholesky"), **kw)
    assert fit["conv"] == 0
    assert np.all(np.abs(fit["par"] - tgt) <= TOL)
    assert fit["mstart"]["stop"] == 0 and fit["mstart"]["nsp"] >= 1


def test_c4_full_multistart_matches_oracle(amd, gslref, nist):
    """mstart_n = 8192 wide ranges: GPU batch + host commit vs the oracle's one-by-one loop"""
    q, x, y, tgt = _boxbod(nist)
    ctrlkw = dict(mstart_n=8192, mstart_q=819, solver="cholesky")
    fit = amd.gsl_nls(q["formula"], data=q["data"], start=dict(b1=[1, 500], b2=[0.01, 5]), jac=True, control=ctrlkw)
    o = gslref.nls(6, 2, [[1, 0.01], [500, 5]], rowdata=dict(model=gslref.MODEL_MISRA1A, x=x, y=y), use_jac=True,
                   ctrl=gslref.control(**ctrlkw))
    assert fit["conv"] == 0 and np.all(np.abs(fit["par"] - tgt) <= TOL)
    m, mo = fit["mstart"], o["mstart"]
    assert (m["nsp"], m["nwsp"], m["iters"], m["stop"]) == (mo["nsp"], mo["nwsp"], mo["iters"], mo["stop"])
    assert abs(m["ssropt"] - mo["ssropt"]) <= 1e-7 * mo["ssropt"]
    assert np.max(np.abs(fit["par"] - o["par"]) / np.abs(o["par"])) < 1e-6


def test_lanes_per_fit_do_not_change_the_records(amd, nist):
    """Small batches of tiny data sets run two or four lanes per fit (rows split over the lanes of a group, sums
    exchanged inside quads), large ones one lane per fit: the first 8192 draws fitted in a batch of 8192 (four lanes),
    of 20000 (two lanes) and of 40000 (one lane) must tell the same story -- same det-filter decisions, iteration
    counts and status codes, end points equal up to what the re-association of six-term sums grows to in five
    iterations."""
    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control, pack_control
    q, x, y, tgt = _boxbod(nist)
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
    prob = amd.DenseProblem(2, 2, x, y)
    ranges = np.array([1.0, 500.0, 0.01, 5.0])
    kd = np.array([0.75, 0.75])
    K = _lib.lib().gslnls_mstart_record_size(2)
    recs = []
    for N in (8192, 20000, 40000):
        rec = np.zeros((N, K))
        ms = C.c_float(0)
        for jac in (1, 0):
            rc = _lib.lib().gslnls_mstart_batch(prob._h, jac, ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), 0,
                                                N, 0, N, 5, 1e-6, ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP),
                                                None, rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
            assert rc == 0
            recs.append(rec[:8192].copy())
    prob.close()
    for jac in (0, 1):
        four, two, one = recs[jac], recs[2 + jac], recs[4 + jac]
        for other in (four, two):
            assert np.array_equal(other[:, 4:6], one[:, 4:6])                    # the sampled points themselves
            assert np.array_equal(other[:, 8] > 1e-6, one[:, 8] > 1e-6)          # det filter
            assert np.array_equal(other[:, 11], one[:, 11]) and np.array_equal(other[:, 12], one[:, 12])
            sel = (one[:, 8] > 1e-6) & np.isfinite(one[:, 7])
            if jac == 0:
                # recs[0::2] are the analytic-Jacobian runs: re-associated six-term sums, nothing else (measured 2e-12)
                assert np.allclose(other[sel, 0:2], one[sel, 0:2], rtol=1e-9, atol=1e-12)
                assert np.allclose(other[sel, 7], one[sel, 7], rtol=1e-9)
            else:
                # forward differences turn a last-bit change of a trial point into a ~1e-9 relative change of the
                # Jacobian (rounding noise eps f / delta is not smooth in the point), and five iterations from a random
                # start grow that to ~1e-4 for a few dozen of the 8192 points -- for any evaluation order, the
                # reference's included; iteration counts and status codes above are unaffected
                assert np.allclose(other[sel, 0:2], one[sel, 0:2], rtol=2e-3, atol=1e-9)
                assert np.allclose(other[sel, 7], one[sel, 7], rtol=2e-3)
                assert np.sum(np.abs(other[sel, 7] / one[sel, 7] - 1.0) > 1e-6) < 200


def test_lane_refill_gives_bit_identical_records(amd, nist, monkeypatch):
    """Large batches: a wavefront owns a slice of the batch and a lane that finishes takes the slice's next point
    (ms_fit_refill_kernel).  Which lane fits which point depends on how long the fits take -- the records do not: every
    record goes to its point's own place and a fit depends on nothing but its point.  Bitwise equal to the
    one-point-per-lane kernel, for the analytic Jacobian and forward differences, ragged slices, explicit starts."""
    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control, pack_control
    q, x, y, tgt = _boxbod(nist)
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
    prob = amd.DenseProblem(2, 2, x, y)
    ranges = np.array([1.0, 500.0, 0.01, 5.0])
    kd = np.array([0.75, 0.75])
    K = _lib.lib().gslnls_mstart_record_size(2)
    N = 70_001                                            # > 65536: one lane per fit; ragged last slice
    out = {}
    for mode, env in (("plain", {"GSLNLS_MS_REFILL": "0"}), ("refill", {"GSLNLS_MS_REFILL": "1", "GSLNLS_MS_WAVES": "97"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for jac in (1, 0):
            rec = np.full((N, K), np.nan)
            ms = C.c_float(0)
            rc = _lib.lib().gslnls_mstart_batch(prob._h, jac, ranges.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), 5,
                                                N, 0, N, 5, 1e-6, ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP),
                                                None, rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
            assert rc == 0
            out[mode, jac] = rec
    prob.close()
    for jac in (1, 0):
        assert not np.isnan(out["refill", jac][:, 0:6]).any()                    # every point was served
        assert np.array_equal(out["plain", jac], out["refill", jac], equal_nan=True)
    # the whole procedure (sampling, concentration, local searches -- explicit starts go through the same kernel)
    d = dict(x=x, y=y)
    kw = dict(data=d, start=dict(b1=[1.0, 500.0], b2=[0.01, 5.0]), jac=True,
              control=dict(mstart_n=20000, mstart_q=500, solver="cholesky"))
    monkeypatch.setenv("GSLNLS_MS_REFILL", "0")
    a = amd.gsl_nls("y ~ b1*(1-exp(-b2*x))", **kw)
    monkeypatch.setenv("GSLNLS_MS_REFILL", "1")
    monkeypatch.setenv("GSLNLS_MS_WAVES", "40")
    b = amd.gsl_nls("y ~ b1*(1-exp(-b2*x))", **kw)
    assert np.array_equal(a["par"], b["par"]) and a["ssr"] == b["ssr"] and a["mstart"] == b["mstart"]


def test_in_library_rccl_allgather_one_rank(amd, nist):
    """The collective of the multi-start path is issued by libgslnls_hip.so itself (csrc/rccl_comm.hpp): bind an RCCL
    communicator of ONE rank through the file bootstrap (the box has one GPU, and RCCL refuses two ranks on one device),
    force the sharded code path, and the whole procedure -- batch kernel -> ncclAllGather on the library's stream ->
    host commit -- must give bit for bit what the plain path gives."""
    import os
    import subprocess
    import sys
    import tempfile
    code = r"""
import json, os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import load_golden
import gslnls_amd as A
from gslnls_amd import _lib
L = _lib.lib()
q = {d["name"]: d for d in load_golden("nist_formula_problems.json")}["BoxBOD"]
d = dict(x=np.array(q["data"]["x"]), y=np.array(q["data"]["y"]))
kw = dict(data=d, start=dict(b1=[1.0, 500.0], b2=[0.01, 5.0]), jac=True, control=dict(mstart_n=1000, mstart_q=50, solver="cholesky"))
plain = A.gsl_nls("y ~ b1*(1-exp(-b2*x))", **kw)
rc = L.gslnls_comm_init_file(sys.argv[1].encode(), 0, 1, 30)
n0 = L.gslnls_comm_allgather_count()
coll = A.gsl_nls("y ~ b1*(1-exp(-b2*x))", **kw)
n1 = L.gslnls_comm_allgather_count()
L.gslnls_comm_destroy()
after = A.gsl_nls("y ~ b1*(1-exp(-b2*x))", **kw)
print(json.dumps(dict(rc=rc, err=L.gslnls_comm_last_error().decode(), collectives=n1 - n0,
                      plain=[plain["par"].tolist(), plain["ssr"], plain["mstart"]],
                      coll=[coll["par"].tolist(), coll["ssr"], coll["mstart"]],
                      after=[after["par"].tolist(), after["ssr"], after["mstart"]])))
""" % (ROOT, ROOT)
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, GSLNLS_COMM_FORCE_COLLECTIVE="1")
        out = subprocess.run([sys.executable, "-c", code, os.path.join(td, "nccl_id")], capture_output=True, text=True,
                             timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["rc"] == 0, r["err"]
    assert r["collectives"] >= 1                       # the library issued the all-gathers itself
    assert r["coll"] == r["plain"] == r["after"]        # and nothing changed, bit for bit
    tgt = np.array(list(nist["BoxBOD"]["target"].values()))
    assert np.all(np.abs(np.array(r["coll"][0]) - tgt) < 1.3e-4)


def test_rccl_path_fails_together_and_recovers(amd, nist):
    """Collective safety of the in-library all-gather (one rank, forced collective path): (a) a rank whose buffer growth
    fails (test hook GSLNLS_COMM_FAIL_ENSURE_RANK) makes the agreed growth fail -- an error, not a hang, and the next
    batch (which grows again) succeeds; (b) gslnls_mstart_batch(lo < 0, records = NULL) leaves the records in HBM and
    still reads the status words; (c) the bootstrap file is gone once every rank has joined; (d) the optional event
    pair reports the collective's own device time"""
    import os
    import subprocess
    import sys
    import tempfile
    code = r"""
import ctypes as C, json, os, sys, numpy as np
sys.path.insert(0, %r)
from gslnls_amd import _lib
from gslnls_amd.control import gsl_nls_control, pack_control
L = _lib.lib()
x = np.asfortranarray(np.array([1.0, 2.0, 3.0, 5.0, 7.0, 10.0]).reshape(6, 1)); y = np.array([109.0, 149.0, 149.0, 191.0, 213.0, 224.0])
model = _lib.Model(2, 2, 1, x.ctypes.data_as(C.c_void_p), 0); err = C.c_int(0)
h = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), 6, None, C.byref(err))
ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
rg = np.array([1.0, 500.0, 0.01, 5.0]); kd = np.array([0.75, 0.75]); ms = C.c_float(0)
DP, IP = _lib.DP, _lib.IP
def batch(count, rec):
    return L.gslnls_mstart_batch(h, 1, rg.ctypes.data_as(DP), kd.ctypes.data_as(DP), 0, count, -1, 0, 5, 1e-6, ci.ctypes.data_as(IP),
                                 cd.ctypes.data_as(DP), None, None if rec is None else rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
K = L.gslnls_mstart_record_size(2)
plain = np.zeros((512, K)); rc_plain = batch(512, plain)              # no communicator yet: the plain path
rc_init = L.gslnls_comm_init_file(sys.argv[1].encode(), 0, 1, 30)
file_left = os.path.exists(sys.argv[1])
rc_fail = batch(512, None)                                             # growth reported as failed -> agreed failure
msg = L.gslnls_comm_last_error().decode()
L.gslnls_comm_set_timing(1)
rc_none = batch(512, None)                                             # records stay in HBM
rec = np.zeros((512, K)); rc_rec = batch(512, rec)
nt = C.c_longlong(0); tot = L.gslnls_comm_allgather_ms(C.byref(nt))
n_coll = L.gslnls_comm_allgather_count()
L.gslnls_comm_destroy(); L.gslnls_dense_destroy(h)
print(json.dumps(dict(rc_plain=rc_plain, rc_init=rc_init, file_left=file_left, rc_fail=rc_fail, msg=msg, rc_none=rc_none, rc_rec=rc_rec,
                      same=bool(np.array_equal(rec, plain)), timed=nt.value, ms=tot, n_coll=n_coll)))
""" % (ROOT,)
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, GSLNLS_COMM_FORCE_COLLECTIVE="1", GSLNLS_COMM_FAIL_ENSURE_RANK="0", GSLNLS_COMM_NONCE="job-42")
        out = subprocess.run([sys.executable, "-c", code, os.path.join(td, "nccl_id")], capture_output=True, text=True,
                             timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["rc_plain"] == 0 and r["rc_init"] == 0 and not r["file_left"]
    assert r["rc_fail"] != 0 and "every rank" in r["msg"]
    assert r["rc_none"] == 0 and r["rc_rec"] == 0 and r["same"]
    assert r["timed"] == 2 and r["ms"] > 0.0 and r["n_coll"] >= 2


def _ms_rank_worker(rank, world, port, q):
    import ctypes as C
    import os
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from gslnls_amd import _lib, dist as gdist
    from gslnls_amd.control import gsl_nls_control, pack_control
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = _lib.lib()
    x = np.asfortranarray(np.array([1.0, 2.0, 3.0, 5.0, 7.0, 10.0]).reshape(6, 1))
    y = np.array([109.0, 149.0, 149.0, 191.0, 213.0, 224.0])
    model = _lib.Model(2, 2, 1, x.ctypes.data_as(C.c_void_p), 0)
    err = C.c_int(0)
    h = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), 6, None, C.byref(err))
    calls = gdist.init_multistart_comm(max_points=256, p=2)        # callback communicator over gloo (host buffers)
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
    rg, kd, ms = np.array([1.0, 500.0, 0.01, 5.0]), np.array([0.75, 0.75]), C.c_float(0)
    K = L.gslnls_mstart_record_size(2)
    DP, IP = _lib.DP, _lib.IP

    def batch(count, rec):
        return L.gslnls_mstart_batch(h, 1, rg.ctypes.data_as(DP), kd.ctypes.data_as(DP), 0, count, -1, 0, 5, 1e-6,
                                     ci.ctypes.data_as(IP), cd.ctypes.data_as(DP), None,
                                     None if rec is None else rec.ctypes.data_as(C.c_void_p), 0, C.byref(ms))
    rc_none = batch(101, None)                                     # records = NULL: was a write through a null vector
    rec = np.zeros((101, K))
    rc_rec = batch(101, rec)
    rc_tiny = batch(1, None)                                       # one point over two ranks: rank 1 owns an empty block
    q.put((rank, rc_none, rc_rec, rc_tiny, rec.tolist(), calls["n"]))
    dist.barrier()
    L.gslnls_dense_destroy(h)
    gdist.reset_comm()
    dist.destroy_process_group()


def test_sharded_batch_world2_without_host_records(amd):
    """two rank processes (gloo, sharing the test box's GPU): gslnls_mstart_batch(lo < 0, records = NULL) on the
    callback communicator completes on both ranks, and the same batch with records gives the single-process records"""
    import ctypes as C
    import torch.multiprocessing as mp
    from gslnls_amd import _lib
    from gslnls_amd.control import gsl_nls_control, pack_control
    L = _lib.lib()
    x = np.asfortranarray(np.array([1.0, 2.0, 3.0, 5.0, 7.0, 10.0]).reshape(6, 1))
    y = np.array([109.0, 149.0, 149.0, 191.0, 213.0, 224.0])
    model = _lib.Model(2, 2, 1, x.ctypes.data_as(C.c_void_p), 0)
    err = C.c_int(0)
    h = L.gslnls_dense_create(C.byref(model), y.ctypes.data_as(C.c_void_p), 6, None, C.byref(err))
    ci, cd = pack_control(gsl_nls_control(solver="cholesky"), "lm")
    rg, kd, ms = np.array([1.0, 500.0, 0.01, 5.0]), np.array([0.75, 0.75]), C.c_float(0)
    K = L.gslnls_mstart_record_size(2)
    solo = np.zeros((101, K))
    rc = L.gslnls_mstart_batch(h, 1, rg.ctypes.data_as(_lib.DP), kd.ctypes.data_as(_lib.DP), 0, 101, 0, 101, 5, 1e-6,
                               ci.ctypes.data_as(_lib.IP), cd.ctypes.data_as(_lib.DP), None, solo.ctypes.data_as(C.c_void_p),
                               0, C.byref(ms))
    L.gslnls_dense_destroy(h)
    assert rc == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + int(np.random.default_rng().integers(0, 2000))
    procs = [ctx.Process(target=_ms_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert r[1] == 0 and r[2] == 0 and r[3] == 0 and r[5] == 3          # three collectives, no failure
        assert np.array_equal(np.array(r[4]), solo)                          # bitwise the unsharded records
