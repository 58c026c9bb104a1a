"""GPU parity of the workgroup-per-dataset batched robust fits (BASELINE config C5) through the C ABI:
every data set of the batch must come out exactly as the single-fit IRLS procedure of the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GAUSS1_TRUTH = np.array([98.778210871, 0.010497276517, 100.48990633, 67.481111276, 23.129773360, 71.994503004,
                         178.99805021, 18.389389025])          # NIST Gauss1 certified values, R/nls_test.R:303-312
GAUSS1_START = np.array([97.0, 0.009, 100.0, 65.0, 20.0, 70.0, 178.0, 16.5])  # NIST start 1, R/nls_test.R:302


def c5_data(B, n, seed0=20250929):
    """Gauss1-family model, x_i = 250 i/n, truth = Gauss1 target x (1 + 0.05 U(-1,1)),
    noise N(0, 2.5^2), 2 % of the points replaced by +50 outliers; seed PCG64(seed0 + dataset)"""
    x = 250.0 * np.arange(1, n + 1) / n
    X = np.tile(x, (B, 1))
    Y = np.zeros((B, n))
    TH = np.zeros((B, 8))
    for d in range(B):
        rng = np.random.Generator(np.random.PCG64(seed0 + d))
        th = GAUSS1_TRUTH * (1.0 + 0.05 * rng.uniform(-1, 1, 8))
        y = (th[0] * np.exp(-th[1] * x) + th[2] * np.exp(-(x - th[3]) ** 2 / th[4] ** 2)
             + th[5] * np.exp(-(x - th[6]) ** 2 / th[7] ** 2)) + 2.5 * rng.standard_normal(n)
        idx = rng.choice(n, n // 50, replace=False)
        y[idx] += 50.0
        Y[d], TH[d] = y, th
    return X, Y, TH


@pytest.fixture(scope="module")
def amd():
    import gslnls_amd
    from gslnls_amd import _lib
    assert _lib.lib().gslnls_device_count() >= 1
    return gslnls_amd


@pytest.mark.parametrize("loss", ["huber", "welsh", "hampel", "lqq"])
def test_batched_other_losses_match_single_fit_oracle(amd, gslref, loss):
    """the batched kernel with the other psi families (src/nls_irls.c:10-341; hampel and lqq carry three tuning constants,
    R/nls_rho.R:106-144) against the oracle's single fits"""
    B, n = 6, 640
    X, Y, TH = c5_data(B, n)
    prob = amd.BatchProblem(4, 8, X, Y)
    out = prob.irls(GAUSS1_START, loss=loss, jac=True, control=dict(solver="cholesky"))
    prob.close()
    for d in range(B):
        o = gslref.nls(n, 8, GAUSS1_START, rowdata=dict(model=gslref.MODEL_GAUSS1, x=X[d], y=Y[d]), use_jac=True,
                       ctrl=gslref.control(solver="cholesky"), loss=loss)
        assert out["conv"][d] == o["conv"] and out["irls_status"][d] == o["irls"]["irls_status"]
        assert out["irls_niter"][d] == o["irls"]["irls_niter"] and out["niter"][d] == o["niter"]
        assert np.allclose(out["par"][d], o["par"], rtol=1e-6)
        assert abs(out["sigma"][d] - o["irls"]["irls_sigma"]) <= 1e-6 * o["irls"]["irls_sigma"]


@pytest.mark.parametrize("n", [1000, 777])
def test_batched_bisquare_matches_single_fit_oracle(amd, gslref, n):
    B = 12
    X, Y, TH = c5_data(B, n)
    prob = amd.BatchProblem(4, 8, X, Y)
    out = prob.irls(GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
    prob.close()
    for d in range(B):
        o = gslref.nls(n, 8, GAUSS1_START, rowdata=dict(model=gslref.MODEL_GAUSS1, x=X[d], y=Y[d]), use_jac=True,
                       ctrl=gslref.control(solver="cholesky"), loss="bisquare")
        assert out["conv"][d] == o["conv"] == 0 and out["irls_status"][d] == o["irls"]["irls_status"] == 0
        assert out["irls_niter"][d] == o["irls"]["irls_niter"] and out["niter"][d] == o["niter"]
        assert np.allclose(out["par"][d], o["par"], rtol=1e-6)
        assert abs(out["sigma"][d] - o["irls"]["irls_sigma"]) <= 1e-6 * o["irls"]["irls_sigma"]
        assert abs(out["ssr"][d] - o["ssr"]) <= 1e-6 * o["ssr"]
        # robust: the 2 % outliers of +50 do not pull the fit away from the generating parameters
        assert np.max(np.abs(out["par"][d] / TH[d] - 1.0)) < 0.05


def test_batch_shards_are_independent(amd):
    """data sets are independent: fitting [lo, hi) shards separately == fitting the whole batch (bitwise)"""
    B, n = 10, 600
    X, Y, _ = c5_data(B, n)
    prob = amd.BatchProblem(4, 8, X, Y)
    full = prob.irls(GAUSS1_START, loss="huber", jac=True)
    a = prob.irls(GAUSS1_START, loss="huber", jac=True, lo=0, hi=4)
    b = prob.irls(GAUSS1_START, loss="huber", jac=True, lo=4, hi=10)
    prob.close()
    assert np.array_equal(np.vstack([a["par"], b["par"]]), full["par"])
    assert np.array_equal(np.concatenate([a["irls_niter"], b["irls_niter"]]), full["irls_niter"])


def test_c5_full_size_properties(amd):
    """BASELINE configs[4] at full size (4096 data sets x n = 1e4, p = 8, bisquare): properties that need no CPU
    reference -- every robust fit converges and recovers its generating parameters in spite of the 2 % outliers,
    a data set fitted inside the full batch comes out bitwise as in a small batch of its own, and a second run
    of the whole batch is bitwise identical."""
    B, n = 4096, 10000
    X, Y, TH = c5_data(B, n)
    prob = amd.BatchProblem(4, 8, X, Y)
    full = prob.irls(GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
    again = prob.irls(GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
    part = prob.irls(GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"), lo=1000, hi=1064)
    prob.close()
    assert int((full["conv"] == 0).sum()) == B and int((full["irls_status"] == 0).sum()) == B
    assert np.array_equal(full["par"], again["par"]) and np.array_equal(full["irls_niter"], again["irls_niter"])
    assert np.array_equal(part["par"], full["par"][1000:1064])
    assert np.array_equal(part["sigma"], full["sigma"][1000:1064])
    assert np.max(np.abs(full["par"] / TH - 1.0)) < 0.02
    # sigma = 1.4826 median|r| of N(0, 2.5^2) noise with 2 % gross outliers
    assert np.all(np.abs(full["sigma"] / 2.5 - 1.0) < 0.1)


def _irls_rank_worker(rank, world, port, B, n, q):
    import os
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import gslnls_amd
    from gslnls_amd import dist as gdist
    from test_gpu_batch import c5_data, GAUSS1_START
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, Y, _ = c5_data(B, n)
    per = (B + world - 1) // world
    lo, hi = min(B, rank * per), min(B, rank * per + per)
    calls = gdist.init_multistart_comm(max_points=B * 2, p=8)      # callback communicator over gloo (host buffers)
    prob = gslnls_amd.BatchProblem(4, 8, X[lo:hi], Y[lo:hi])      # this rank's block only
    out = prob.irls_gathered(B, GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
    prob.close()
    q.put((rank, out["par"].tolist(), out["sigma"].tolist(), out["irls_niter"].tolist(), out["conv"].tolist(), calls["n"]))
    dist.barrier()
    gdist.reset_comm()
    dist.destroy_process_group()


def test_batched_irls_world2_gathers_every_data_set(amd):
    """SURVEY.md 8(e), row 'batched IRLS': two rank processes (sharing the one GPU of the test box) each fit their
    contiguous block of an uneven split, ONE all-gather completes theta-hat / sigma-hat / status on both ranks --
    identical on both and bitwise equal to the single-process batch"""
    import torch.multiprocessing as mp
    B, n = 7, 500
    X, Y, _ = c5_data(B, n)
    prob = amd.BatchProblem(4, 8, X, Y)
    solo = prob.irls(GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
    one = prob.irls_gathered(B, GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
    prob.close()
    assert np.array_equal(one["par"], solo["par"])          # one rank: the gathered form is the plain one
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + int(np.random.default_rng().integers(0, 2000))
    procs = [ctx.Process(target=_irls_rank_worker, args=(r, 2, port, B, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    r0, r1 = res
    assert r0[1:5] == r1[1:5] and r0[5] == r1[5] == 1         # same on both ranks, exactly one collective
    assert np.array_equal(np.array(r0[1]), solo["par"]) and np.array_equal(np.array(r0[2]), solo["sigma"])
    assert r0[3] == solo["irls_niter"].tolist() and r0[4] == solo["conv"].tolist()


def _irls_empty_block_worker(rank, world, port, n, q):
    import os
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import gslnls_amd
    from gslnls_amd import dist as gdist
    from test_gpu_batch import c5_data, GAUSS1_START
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 1                                                            # one data set over two ranks: rank 1's block is empty
    X, Y, _ = c5_data(B, n)
    per = (B + world - 1) // world
    lo, hi = min(B, rank * per), min(B, rank * per + per)
    gdist.init_multistart_comm(max_points=64, p=8)
    prob = gslnls_amd.BatchProblem(4, 8, X[lo:hi].reshape(hi - lo, 1, n), Y[lo:hi])     # B = 0 on rank 1
    out = prob.irls_gathered(B, GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
    # a handle of the wrong size on ONE rank: carried through the collective, both ranks get an error, none hangs
    bad = gslnls_amd.BatchProblem(4, 8, X[0:1].reshape(1, 1, n), Y[0:1]) if rank == 1 else prob
    try:
        bad.irls_gathered(B, GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
        failed = False
    except Exception:  # noqa
        failed = True
    q.put((rank, out["par"].tolist(), out["conv"].tolist(), failed))
    dist.barrier()
    prob.close()
    gdist.reset_comm()
    dist.destroy_process_group()


def test_batched_irls_gather_with_an_empty_block_and_a_failing_rank(amd):
    """B_total smaller than the number of ranks (B_total = 9 on 8 ranks leaves ranks without data sets): the
    rank with the empty block joins the collective through a B = 0 handle; a rank-local argument error is carried
    through the all-gather and fails every rank instead of leaving the others blocked"""
    import torch.multiprocessing as mp
    n = 500
    X, Y, _ = c5_data(1, n)
    prob = amd.BatchProblem(4, 8, X, Y)
    solo = prob.irls(GAUSS1_START, loss="bisquare", jac=True, control=dict(solver="cholesky"))
    prob.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + int(np.random.default_rng().integers(0, 2000))
    procs = [ctx.Process(target=_irls_empty_block_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    for r in res:
        assert np.array_equal(np.array(r[1]), solo["par"]) and r[2] == solo["conv"].tolist() and r[3] is True
