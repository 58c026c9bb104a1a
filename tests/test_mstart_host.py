"""Multi-start host logic on CPU: the product's driver (gslnls_amd/csrc/mstart_driver.hpp: batched
concentration + speculative local searches + sequential commit) must reproduce the reference's
strictly sequential procedure as restated by the oracle (src/nls_mstart.c, src/nls.c:274-532).
The per-point fits here run through tests/hostsim (device headers on the host); the GPU versions
of the same checks are in tests/test_gpu_mstart.py."""
import numpy as np
import pytest

from conftest import load_golden

TOL = float(np.finfo(float).eps ** 0.25)


def test_product_sobol_matches_oracle_and_golden(gslref, hostsim):
    g = np.array(load_golden("sobol_d2.json")["points"])
    assert np.array_equal(hostsim.sobol(2, len(g)), g)          # index-addressed == golden (GSL order)
    for dim in (1, 3, 7, 40):
        assert np.array_equal(hostsim.sobol(dim, 300), gslref.sobol(dim, 300))  # == sequential Gray-code generator
    assert np.array_equal(hostsim.sobol(5, 50, first=1000), gslref.sobol(5, 50, skip=1000))  # skip-ahead
    assert np.allclose(hostsim.sobol(41, 20), gslref.halton(41, 20), rtol=0, atol=1e-15)  # p > 40 -> Halton


def _boxbod(nist):
    q = nist["BoxBOD"]
    return np.array(q["data"]["x"]), np.array(q["data"]["y"]), np.array(list(q["target"].values()))


CASES = [
    # (start ranges [lower; upper], has_start, lower, upper, weights, ctrl kw)
    dict(start=[[200, 0], [250, 1]]),                                                        # 4.1.1
    dict(start=[[200, 1], [250, 1]], weights=10.0),                                          # 4.1.3
    dict(start=[[200, -0.1], [200, 0.75]], has_start=[[1, 0], [1, 0]], lower=[-np.inf, 0]),  # 4.1.4
    dict(start=[[-0.1, 0.5], [0.75, 0.5]], has_start=[[0, 1], [0, 1]], upper=[np.inf, 1]),   # 4.1.5
    dict(start=[[-0.1, 0], [0.75, 1]], has_start=[[0, 1], [0, 1]], lower=[0, 0], upper=[250, 250]),  # 4.1.6
    dict(start=[[1, 0.01], [500, 5]], ctrl=dict(mstart_n=64, mstart_q=6)),                   # wide C4-style ranges
    dict(start=[[1, 0.01], [500, 5]], ctrl=dict()),                                          # default controls
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("jac", [1, 0])
def test_driver_replays_reference_sequence(gslref, hostsim, nist, case, jac):
    x, y, tgt = _boxbod(nist)
    kw = dict(mstart_n=5, mstart_q=1, mstart_r=1.1)
    kw.update(case.get("ctrl", {}))
    if "ctrl" in case and not case["ctrl"]:
        kw = {}
    ctrl = gslref.control(solver="cholesky", **kw)
    start = np.array(case["start"], dtype=float)
    hs = case.get("has_start")
    ci, cd = gslref.pack_control(ctrl, "lm", False, True, hs is not None and not np.all(hs))
    w = None if "weights" not in case else np.full(6, case["weights"])
    lo, up = case.get("lower"), case.get("upper")
    lu = None
    if lo is not None or up is not None:
        lo = np.full(2, -np.inf) if lo is None else np.asarray(lo, float)
        up = np.full(2, np.inf) if up is None else np.asarray(up, float)
        lu = np.ascontiguousarray(np.stack([lo, up], axis=1).reshape(-1))
    h = hostsim.mstart(2, 2, x, y, start, ci, cd, has_start=hs, jac=jac, sw=None if w is None else np.sqrt(w),
                       lupars=lu)
    o = gslref.nls(6, 2, start, rowdata=dict(model=gslref.MODEL_MISRA1A, x=x, y=y), use_jac=bool(jac), ctrl=ctrl,
                   weights=w, lower=None if lu is None else lu[0::2], upper=None if lu is None else lu[1::2],
                   has_start=hs)
    assert h["rc"] == 0
    # identical bookkeeping: stationary points found, major iterations, stop reason
    assert (h["nsp"], h["nwsp"], h["iters"], h["stop"]) == (o["mstart"]["nsp"], o["mstart"]["nwsp"],
                                                           o["mstart"]["iters"], o["mstart"]["stop"])
    assert abs(h["ssropt"] - o["mstart"]["ssropt"]) <= 1e-7 * abs(o["mstart"]["ssropt"])
    # the final single-start fit from the driver's optimum lands on the certified BoxBOD values
    f = hostsim.fit(2, 2, x, y, h["mpopt"], ci, cd, jac=jac, sw=None if w is None else np.sqrt(w), lupars=lu)
    assert np.all(np.abs(f["par"] - tgt) <= TOL)
    assert np.max(np.abs(f["par"] - o["par"]) / np.abs(o["par"])) < 1e-6


def _gather_worker(rank, world, port, q):
    import os
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import gslref
    import hostsim_py as H
    from conftest import load_golden
    dist.init_process_group("gloo", rank=rank, world_size=world)
    qd = {d["name"]: d for d in load_golden("nist_formula_problems.json")}["BoxBOD"]
    x, y = np.array(qd["data"]["x"]), np.array(qd["data"]["y"])
    ctrl = gslref.control(solver="cholesky", mstart_n=101, mstart_q=10)  # 101: uneven shards on purpose
    ci, cd = gslref.pack_control(ctrl, "lm")
    K = 3 * 2 + 8
    per = (101 + world - 1) // world
    shard = torch.zeros(per * K, dtype=torch.float64)
    allb = torch.zeros(world * per * K, dtype=torch.float64)
    calls = []

    def allgather(_ctx, per_points, k):
        calls.append((per_points, k))
        dist.all_gather_into_tensor(allb[:world * per_points * k], shard[:per_points * k])
        return 0
    h = H.mstart(2, 2, x, y, [[1, 0.01], [500, 5]], ci, cd, jac=1, rank=rank, world=world, allgather=allgather,
                 shard_buf=shard.numpy(), all_buf=allb.numpy(), cap_points=world * per)
    q.put((rank, h["rc"], h["mpopt"].tolist(), h["nsp"], h["nwsp"], h["iters"], h["ssropt"], h["total_fits"], len(calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_multistart_world2_gloo(gslref, hostsim, nist):
    """N > 1 path: two processes each fit half of every concentration batch, an all-gather (gloo here,
    RCCL on the GPUs) completes the records, both replay the same commit -> bit-identical state on every
    rank and identical to the single-process run."""
    import torch.multiprocessing as mp
    x, y, tgt = _boxbod(nist)
    ctrl = gslref.control(solver="cholesky", mstart_n=101, mstart_q=10)
    ci, cd = gslref.pack_control(ctrl, "lm")
    solo = hostsim.mstart(2, 2, x, y, [[1, 0.01], [500, 5]], ci, cd, jac=1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (np.random.default_rng().integers(0, 2000))
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, int(port), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    (r0, r1) = res
    assert r0[1] == 0 and r1[1] == 0
    assert r0[2:8] == r1[2:8]                                   # identical on both ranks
    assert r0[8] >= 1 and r0[8] == r1[8]                        # the collective ran, same number of times
    assert r0[2] == solo["mpopt"].tolist()                      # identical to the unsharded run, bit for bit
    assert (r0[3], r0[4], r0[5]) == (solo["nsp"], solo["nwsp"], solo["iters"]) and r0[6] == solo["ssropt"]
    f = hostsim.fit(2, 2, x, y, np.array(r0[2]), ci, cd, jac=1)
    assert np.all(np.abs(f["par"] - tgt) <= TOL)


def test_sharded_batch_without_host_records_reads_the_status_words(gslref, hostsim, nist):
    """gslnls_mstart_batch(lo < 0, records = NULL) on a callback communicator: the records of the batch are
    not wanted on the host, but the status words of the shards are still read from the gathered array -- no write
    through the unsized record vector, and a peer's failed shard still fails this rank"""
    x, y, _ = _boxbod(nist)
    ci, cd = gslref.pack_control(gslref.control(solver="cholesky"), "lm")
    rc, _, calls = hostsim.run_batch_comm(x, y, 101, 0, 2, ci, cd, want_records=False)
    assert rc == 0 and calls == [(51, 14)]
    rc1, rec1, _ = hostsim.run_batch_comm(x, y, 101, 1, 2, ci, cd, want_records=True)
    assert rc1 == 0 and np.all(rec1[:51] == 0.0) and np.any(rec1[51:] != 0.0)   # rank 1 computed the second block only
    rc2, _, calls2 = hostsim.run_batch_comm(x, y, 101, 0, 2, ci, cd, want_records=False, failing_peer=True)
    assert rc2 != 0 and len(calls2) == 1                                      # entered the collective, then failed
    # more ranks than points: the last ranks own an empty block and still take part
    rc3, _, calls3 = hostsim.run_batch_comm(x, y, 3, 3, 4, ci, cd, want_records=False)
    assert rc3 == 0 and calls3 == [(1, 14)]
