"""The native multi-GPU half-iterations (include/cumf_dist_capi.h, csrc/als_dist.cpp) and the native `hugewiki` program.

The GPU box has one GPU: N ranks share cuda:0 and the exchange goes through `cumf_comm_create_custom` (gloo, host
staged) -- every offset, padding and stream-ordering decision of als_dist.cpp executes, RCCL itself at world 1
(tests/test_dist_gpu.py::test_rccl_backend_world1[native=1], test_hugewiki_binary_*).  The yardstick is the same engine
with torch.distributed collectives driven from Python (the path of rounds 1-5, itself checked against the oracle): the
same kernels on the same plans, so the factors must come out BIT-IDENTICAL."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ab_worker(rank, world, port, scheme, solver, d, m, n, f, lam, iters, theta_batch, theta0, q):
    """One rank: the engine twice on the same inputs -- torch.distributed collectives, then the native path."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CUMF_ALS_PIPE_CHUNKS"] = "4"   # both sides in four pieces (the default pipelines only matrices >= 32 MB)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cumf_als_amd import dist as cdist

        mat = cdist.HostMatrix(m, n, d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indptr"],
                               d["csc_indices"], d["csc_data"])
        out = []
        for native in (False, True):
            cdist.set_native(native)
            eng = cdist.DistALS(mat, f, lam, cdist.HipOps("cuda:0"), solver=solver, cg_iters=6, scheme=scheme,
                                theta_batch=theta_batch)
            assert (eng._ncomm is not None) == native
            if native:
                assert eng._ncomm.name == "custom"
                if scheme == "gather":
                    assert eng._nx.pieces == 4 and eng._nt.pieces == 4
            eng.init_factors(theta0)
            eng.iterate(iters)
            eng.update_x()
            sse = eng.update_theta(train_sse=True)
            torch.cuda.synchronize()
            out.append((eng.thetaT.cpu().numpy().copy(), eng.full_XT().cpu().numpy().copy(), sse))
            eng.close()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,scheme,solver,shape", [
    (2, "gather", "lu", (121, 91, 20, 3)),
    (2, "gather", "cg", (121, 91, 100, 3)),
    (2, "reduce", "lu", (121, 91, 20, 3)),
    (2, "reduce", "cg", (121, 91, 100, 3)),
    (3, "reduce", "lu", (130, 31, 20, 1)),     # 31 columns over three ranks: k = 11, the last share is short
    (4, "reduce", "cg", (90, 27, 20, 3)),      # Theta batches of 9 columns over four ranks: k = 3, rank 3's share is EMPTY
    (4, "gather", "lu", (7, 50, 20, 1)),       # 7 X rows over four ranks in four pieces: empty pieces and empty slabs
])
def test_native_equals_torch_collectives(oracle, alslib, world, scheme, solver, shape):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import torch.multiprocessing as mp

    from cumf_als_amd import datagen
    from tests.test_dist_cpu import _free_port

    m, n, f, theta_batch = shape
    lam, iters = 0.05, 2
    r = datagen.synth_ratings(m, n, max(m * n // 3, m + n), 100, seed=21, row_alpha=1.1)
    d = r.numpy()
    theta0 = (0.2 * np.random.RandomState(0).random_sample((n, f))).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ab_worker, args=(rk, world, port, scheme, solver, d, m, n, f, lam, iters, theta_batch, theta0, q))
             for rk in range(world)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=900) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    th_ref, x_ref = theta0.copy(), np.zeros((m, f), np.float32)
    oracle.do_als(d, th_ref, x_ref, m, n, f, lam, iters + 1, solver=solver)
    for rank, ((th_t, x_t, sse_t), (th_n, x_n, sse_n)) in outs:
        np.testing.assert_array_equal(th_n, th_t)
        np.testing.assert_array_equal(x_n, x_t)
        assert (sse_t is None) == (sse_n is None)
        if sse_t is not None:
            assert abs(sse_n - sse_t) <= 1e-9 * abs(sse_t)
        # ... and both are the factorisation (loose: the yardstick proper is the bit identity above)
        ok = np.isfinite(th_ref).all(1)
        tol = 5e-3 if solver == "lu" else 5e-2   # six CG steps on rows with fewer ratings than features: a sensitive iterate
        assert np.abs(th_n[ok] - th_ref[ok]).max() <= tol * np.abs(th_ref[ok]).max()
    for rank, (_, (th_n, x_n, _)) in outs[1:]:
        np.testing.assert_array_equal(th_n, outs[0][1][1][0])   # replicas agree
        np.testing.assert_array_equal(x_n, outs[0][1][1][1])


def _write_split(tmp_path, m, n, nnz, nnz_test, gpus, seed=13):
    from cumf_als_amd import convert, datagen

    r = datagen.synth_ratings(m, n, nnz, nnz_test, seed=seed, device="cpu")
    datagen.write_dataset(r, str(tmp_path / "d"))
    convert.split_dataset(str(tmp_path / "d"), str(tmp_path / "s"), gpus, m, n, r.nnz, r.nnz_test)
    return r


def _rmse_lines(text):
    import re

    tr = [float(v) for v in re.findall(r"Train RMSE in iter \d+: ([0-9.naninf-]+)", text)]
    te = [float(v) for v in re.findall(r"Test RMSE in iter \d+: ([0-9.naninf-]+)", text)]
    return np.array(list(zip(tr, te)))


@pytest.mark.parametrize("solver,rccl", [("lu", "0"), ("cg", "0"), ("lu", "1")])
def test_hugewiki_binary_single_gpu(oracle, alslib, tmp_path, solver, rccl):
    """./hugewiki (the compiled multi-GPU program, hugewiki.cu's main) on one GPU: its RMSE lines against the oracle's doALS
    to 1e-4 (exact test grid), its factors against the oracle's.  rccl=1: the same through a one-rank RCCL communicator
    (CUMF_DIST_FORCE_RCCL: ncclCommInitRank, reduce-scatter, all-gather and all-reduce execute for real)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from cumf_als_amd import lib

    m, n, f, lam, iters = 300, 120, 20, 0.05, 3
    r = _write_split(tmp_path, m, n, 15000, 1500, 1)
    d = r.numpy()
    th0, x0 = oracle.init_factors(m, n, f)
    _, log_o = oracle.do_als(d, th0, x0, m, n, f, lam, iters, solver=solver, test_grid_compat=False)
    os.makedirs(tmp_path / "model")
    env = dict(os.environ, CUMF_ALS_SOLVER=solver, CUMF_DIST_FORCE_RCCL=rccl, CUMF_ALS_DUMP_MODEL=str(tmp_path / "model"),
               CUMF_DIST_ID_FILE=str(tmp_path / "id"))
    out = subprocess.run([lib.HUGEWIKI_PATH, str(tmp_path / "s"), str(n), str(f), str(lam), str(iters), "3"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert ("transport rccl" in out.stdout) == (rccl == "1"), out.stdout
    log = _rmse_lines(out.stdout)
    assert log.shape == (iters, 2), out.stdout
    assert np.abs(log - np.asarray(log_o)).max() <= 1e-4, (log, log_o)
    th = np.fromfile(tmp_path / "model" / "thetaT.data", np.float32).reshape(n, f)
    x = np.fromfile(tmp_path / "model" / "XT.data0", np.float32).reshape(m, f)
    tol = 2e-4 if solver == "lu" else 3e-3
    assert np.abs(th - th0.reshape(n, f)).max() <= tol * np.abs(th0).max()
    assert np.abs(x - x0.reshape(m, f)).max() <= tol * np.abs(x0).max()


def test_hugewiki_binary_usage_and_split_mismatch(alslib, tmp_path):
    """Wrong argument count prints the usage (as main.cpp does); a split made for another GPU count is refused."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from cumf_als_amd import lib

    out = subprocess.run([lib.HUGEWIKI_PATH], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "Usage: ./hugewiki" in out.stdout
    _write_split(tmp_path, 60, 40, 900, 100, 2)
    out = subprocess.run([lib.HUGEWIKI_PATH, str(tmp_path / "s"), "40", "20", "0.05", "1", "1"], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "was split for 2 GPUs" in out.stderr
