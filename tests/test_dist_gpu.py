"""N > 1 path with the real HIP kernels: two processes share cuda:0 (the GPU box has one
GPU), collectives go through gloo with host staging.  Same expectations as the CPU tier."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# f = 32: f % 16 == 0, the in-kernel split with the packed rating block (kArithSplitPk) under both schemes
@pytest.mark.parametrize("scheme,solver,f", [("gather", "lu", 20), ("reduce", "lu", 20), ("reduce", "cg", 20),
                                             ("gather", "lu", 32), ("reduce", "cg", 32)])
def test_world2_hip_matches_single_process(oracle, alslib, scheme, solver, f):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import torch.multiprocessing as mp

    from cumf_als_amd import datagen
    from tests import dist_helpers
    from tests.test_dist_cpu import _free_port

    m, n, lam, iters = 120, 90, 0.05, 2
    r = datagen.synth_ratings(m, n, 6000, 600, seed=12, row_alpha=1.1)
    d = r.numpy()
    theta0 = (0.2 * np.random.RandomState(0).random_sample((n, f))).astype(np.float32)
    th_ref, x_ref = theta0.copy(), np.zeros((m, f), np.float32)
    oracle.do_als(d, th_ref, x_ref, m, n, f, lam, iters, solver=solver)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=dist_helpers.worker,
                         args=(rk, 2, port, scheme, solver, d, m, n, f, lam, iters, 2, theta0, q, "hip"))
             for rk in range(2)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    tol = 2e-4 if solver == "lu" else 3e-3
    for rank, th, x in outs:
        assert np.abs(th - th_ref).max() <= tol * np.abs(th_ref).max()
        assert np.abs(x - x_ref).max() <= tol * np.abs(x_ref).max()
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("solver", ["lu", "cg"])
def test_local_slab_reduce_single_rank(oracle, alslib, solver):
    """`DistALS.from_local_slab` (hugewiki path: slab-local CSC built on the device, partial Gram
    -> reduce-scatter -> solve -> all-gather) with one rank == plain ALS."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from cumf_als_amd import datagen
    from cumf_als_amd import dist as cdist

    m, n, f, lam, iters = 150, 60, 20, 0.05, 2
    r = datagen.synth_ratings(m, n, 4000, 300, seed=5, col_seed=7).to("cuda")
    d = r.numpy()
    theta0 = (0.2 * np.random.RandomState(0).random_sample((n, f))).astype(np.float32)
    th_ref, x_ref = theta0.copy(), np.zeros((m, f), np.float32)
    oracle.do_als(d, th_ref, x_ref, m, n, f, lam, iters, solver=solver)
    eng = cdist.DistALS.from_local_slab(m, n, [0, m], r.csr_indptr, r.csr_indices, r.csr_data, f, lam,
                                        cdist.HipOps("cuda"), solver=solver, theta_batch=3)
    eng.init_factors(theta0)
    eng.iterate(iters)
    torch.cuda.synchronize()
    tol = 2e-4 if solver == "lu" else 3e-3
    assert np.abs(eng.thetaT.cpu().numpy() - th_ref).max() <= tol * np.abs(th_ref).max()
    assert np.abs(eng.XT.cpu().numpy() - x_ref).max() <= tol * np.abs(x_ref).max()
    # the device-side CSC build agrees with scipy's ordering
    cp, ri, cv = cdist.local_csc_of_slab_torch(r.csr_indptr, r.csr_indices, r.csr_data, n)
    assert np.array_equal(cp, d["csc_indptr"]) and np.array_equal(ri.cpu().numpy(), d["csc_indices"])
    assert np.array_equal(cv.cpu().numpy(), d["csc_data"])


def test_hugewiki_runner_single_gpu(oracle, alslib, tmp_path):
    """convert split (1 slab) -> cumf_als_amd.hugewiki.run with the HIP kernels: the RMSE log of the
    reduce scheme agrees with the oracle's doALS (exact test grid) to 1e-4."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from cumf_als_amd import convert, datagen, hugewiki

    m, n, f, lam, iters = 300, 120, 20, 0.05, 3
    r = datagen.synth_ratings(m, n, 15000, 1500, seed=13, device="cpu")
    d = r.numpy()
    datagen.write_dataset(r, str(tmp_path / "d"))
    convert.split_dataset(str(tmp_path / "d"), str(tmp_path / "s"), 1, m, n, r.nnz, r.nnz_test)
    th0, x0 = oracle.init_factors(m, n, f)
    _, log_o = oracle.do_als(d, th0, x0, m, n, f, lam, iters, solver="lu", test_grid_compat=False)
    torch.cuda.set_device(0)
    eng, log = hugewiki.run(str(tmp_path / "s"), n, f, lam, iters, solver="lu", quiet=True)
    assert np.abs(np.array(log) - np.asarray(log_o)).max() <= 1e-4, (log, log_o)
    assert np.abs(eng.thetaT.cpu().numpy() - th0.reshape(n, f)).max() <= 2e-3 * np.abs(th0).max()


def _rccl_worker(port, scheme, solver, d, m, n, f, lam, iters, theta0, q, native="1"):
    """One rank, backend "nccl" (= RCCL): the device-side collectives of cumf_als_amd.dist --
    reduce_scatter_tensor on the packed Gram, all_gather_into_tensor on the factor slabs -- execute
    for real, which the gloo tests (host staging) never reach."""
    import os

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CUMF_DIST_NATIVE"] = native  # "1": cumf_dist_* (als_dist.cpp) over its own RCCL communicator; "0": torch.distributed
    os.environ["CUMF_ALS_PIPE_FORCE"] = "1"  # the pipelined (async) all-gathers, on one rank ...
    os.environ["CUMF_ALS_PIPE_CHUNKS"] = "4"  # ... and on both sides (the default pipelines only factor matrices >= 32 MB)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from cumf_als_amd import dist as cdist

        assert dist.get_backend() == "nccl"
        if scheme == "gather_device":  # what bench.py --gpus N runs: zero-copy slabs of a device-resident matrix
            from cumf_als_amd import datagen

            r = datagen.Ratings(m=m, n=n, **{k: torch.from_numpy(v).cuda() for k, v in d.items()})
            eng = cdist.DistALS.from_device_ratings(r, f, lam, cdist.HipOps("cuda:0"), solver=solver, cg_iters=6)
        else:
            mat = cdist.HostMatrix(m, n, d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indptr"],
                                   d["csc_indices"], d["csc_data"])
            eng = cdist.DistALS(mat, f, lam, cdist.HipOps("cuda:0"), solver=solver, cg_iters=6, scheme=scheme,
                                theta_batch=3)
        if eng._ncomm is not None:  # the native half-iterations own an RCCL communicator of their own
            assert eng._ncomm.name == "rccl"
            if scheme.startswith("gather"):
                assert eng._nx.pieces == 4 and eng._nt.pieces == 4 and eng._px is None and eng._pt is None
            else:
                assert eng._nr is not None
        elif scheme.startswith("gather"):
            assert eng._px is not None and eng._px.chunks == 4 and not eng._px.staged
            assert eng._pt is not None and eng._pt.chunks == 4 and not eng._pt.staged   # the 192 MB side is pipelined too
        eng.init_factors(theta0)
        eng.iterate(iters)
        torch.cuda.synchronize()
        q.put((eng.thetaT.cpu().numpy().copy(), eng.full_XT().cpu().numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("native", ["1", "0"])
@pytest.mark.parametrize("scheme,solver", [("reduce", "lu"), ("reduce", "cg"), ("gather", "lu"), ("gather_device", "lu")])
def test_rccl_backend_world1(oracle, alslib, scheme, solver, native):
    """the RCCL branches of dist.py (device tensors, no host staging, async
    reduce-scatter overlapped with the next Theta batch) run on the GPU box with world_size 1."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import torch.multiprocessing as mp

    from cumf_als_amd import datagen
    from tests.test_dist_cpu import _free_port

    m, n, f, lam, iters = 120, 90, 20, 0.05, 2
    r = datagen.synth_ratings(m, n, 6000, 600, seed=12, row_alpha=1.1)
    d = r.numpy()
    theta0 = (0.2 * np.random.RandomState(0).random_sample((n, f))).astype(np.float32)
    th_ref, x_ref = theta0.copy(), np.zeros((m, f), np.float32)
    oracle.do_als(d, th_ref, x_ref, m, n, f, lam, iters, solver=solver)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), scheme, solver, d, m, n, f, lam, iters, theta0, q, native))
    p.start()
    th, x = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    tol = 2e-4 if solver == "lu" else 3e-3
    assert np.abs(th - th_ref).max() <= tol * np.abs(th_ref).max()
    assert np.abs(x - x_ref).max() <= tol * np.abs(x_ref).max()


def test_bench_hugewiki_leg_over_rccl_world1(alslib):
    """The hugewiki leg of the N > 1 bench line over the backend the driver's 8-GPU run uses: RCCL
    ("nccl"), device tensors in every collective of the leg (MAX all-reduce of the elapsed time, all-gather of the per-rank
    diagnostics, the reduce-scatter / all-gather of the `reduce` scheme).  One rank is all a 1-GPU box allows; the two-rank
    form of the same code runs over gloo in test_bench_world2_branch_runs."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import json
    import os
    import subprocess
    import sys

    from tests.test_dist_cpu import _free_port

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import argparse, json, os, sys, torch\n"
        "import torch.distributed as dist\n"
        f"sys.path.insert(0, {root!r})\n"
        "import bench\n"
        "from cumf_als_amd import als, datagen\n"
        "torch.cuda.set_device(0)\n"
        "dev = torch.device('cuda', 0)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "a = argparse.Namespace(scale=0.05, seed=0, theta_batch=0, reference_solvers=False)\n"
        "out = bench.hugewiki_leg(a, als, datagen, dev, 1, 0, 'nccl', steps=2)\n"
        "print(json.dumps(out))\n"
        "dist.destroy_process_group()\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    hw = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert hw["scaling"] == "weak" and hw["scheme"] == "reduce" and hw["n_ranks_seen"] == 1
    assert hw["value"] > 0 and np.isfinite(hw["value"]) and hw["x_half_ms"] > 0 and hw["theta_half_ms"] > 0
    # round 6: the leg runs on torch.distributed collectives first, then on the native half-iterations (als_dist.cpp over
    # an RCCL communicator of their own), which take the object over; same factors either way
    # (whichever measured faster keeps the object's own numbers -- at this scale a step is < 1 ms and either may win --
    # the other set sits beside them)
    assert hw["native_equals_torch_collectives"] is True, hw
    if hw["collectives"].startswith("native"):
        other = hw["torch_collectives"]
    else:
        other = hw["native"]
        assert other["transport"] == "rccl" and other["ms_per_step"] > 1.02 * hw["ms_per_step"], hw
    assert other["value"] > 0 and other["x_half_ms"] > 0


def test_pack_unpack_upper(alslib):
    """cumf_pack_upper / cumf_unpack_upper: the packed upper triangle that the multi-GPU Theta phase
    reduce-scatters (half the bytes of hugewiki.cu:2703-2717's full f x f copies)."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from cumf_als_amd import als

    for f in (10, 64, 100, 200):
        a = torch.randn(37, f, f, device="cuda")
        a = a + a.transpose(1, 2)
        packed = als.pack_upper(a)
        iu = torch.triu_indices(f, f)
        assert torch.equal(packed.cpu(), a[:, iu[0], iu[1]].cpu())
        back = torch.zeros_like(a)
        als.unpack_upper(packed, back)
        assert torch.equal(back, a)


@pytest.mark.parametrize("launch", ["torchrun", "self"])
@pytest.mark.parametrize("args", [["--scheme", "gather"], ["--shape", "hugewiki", "--scheme", "reduce", "--solver", "cg"],
                                  ["--scheme", "reduce"]])
def test_bench_world2_branch_runs(alslib, args, launch):
    """bench.py's `world > 1` branch (process group, from_device_ratings / from_local_slab,
    barrier, MAX all-reduce of the elapsed time) executed with TWO ranks, launched exactly as the driver launches
    it (python -m torch.distributed.run ... bench.py --gpus 2) -- on the one GPU of this box, so gloo stands in
    for RCCL (CUMF_BENCH_BACKEND=gloo: collectives staged through the host) and both ranks share cuda:0.  The
    shape is shrunk (--scale): the line is checked for shape, not for speed.
    launch = "self" (VERDICT r03 next 2a): plain `python bench.py --gpus 2` without a torchrun environment must
    re-launch itself under torch.distributed.run and print the same single line, with the per-rank diagnostics."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import json
    import os
    import subprocess
    import sys

    from tests.test_dist_cpu import _free_port

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CUMF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    if launch == "self":
        if args != ["--scheme", "gather"]:
            pytest.skip("self-launch is exercised once")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, os.path.join(root, "bench.py")]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py")]
    cmd += ["--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.05", "--no-cpu-baseline"] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]     # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["value"] > 0 and np.isfinite(line["value"]) and line["ms_per_step"] > 0
    assert line["scaling"] == ("weak" if "hugewiki" in args else "strong")
    # per-rank diagnostics of the N > 1 line: two ranks seen, kernel time inside the half-iteration time
    rk = line["ranks"]
    assert rk["n_ranks_seen"] == 2 and rk["self_launched"] == (launch == "self")
    assert rk["scheme"] == ("reduce" if "reduce" in args else "gather")
    pr = rk["per_rank"]
    for side in ("x", "theta"):
        assert len(pr[f"{side}_half_ms"]) == 2 and len(pr[f"{side}_kernel_ms"]) == 2
        for h, k, n_l in zip(pr[f"{side}_half_ms"], pr[f"{side}_kernel_ms"], pr[f"{side}_launches"]):
            assert 0 < k <= h * 1.05 and n_l >= 1, (side, h, k, n_l)
    # the default N > 1 line (Netflix shape, strong scaling) also carries the configuration the 8-GPU
    # target is defined on -- the hugewiki slab per GPU, `reduce` scheme, weak scaling -- as `hugewiki`
    if "hugewiki" in args:
        assert "hugewiki" not in line
    else:
        # round 6: the line was measured on torch.distributed collectives first and then taken over by the native
        # half-iterations (cumf_dist_*: here over the custom transport, gloo); both sets of numbers are in it
        # (whichever measured faster keeps the headline; over the host-staged stand-in transport that may be either)
        assert line["native_equals_torch_collectives"] is True, line
        if line["collectives"].startswith("native"):
            assert rk["transport"] == "custom"
            tc = line["torch_collectives"]
            assert tc["value"] > 0 and tc["ms_per_step"] > 0 and len(tc["ranks"]["per_rank"]["x_half_ms"]) == 2
        else:
            assert line["native"]["value"] > 0 and line["native"]["ms_per_step"] > 1.02 * line["ms_per_step"]
            assert line["native"]["transport"] == "custom"
        hw = line["hugewiki"]
        other = hw["torch_collectives"] if hw["collectives"].startswith("native") else hw["native"]
        assert other["value"] > 0 and hw["native_equals_torch_collectives"] is True, hw
        assert hw["scaling"] == "weak" and hw["scheme"] == "reduce" and hw["n_ranks_seen"] == 2 and hw["theta_batch"] == 3
        assert hw["value"] > 0 and np.isfinite(hw["value"]) and hw["ms_per_step"] > 0
        assert hw["x_half_ms"] > 0 and hw["theta_half_ms"] > 0 and set(hw["non_kernel_ms"]) == {"x", "theta"}
        assert len(hw["per_rank"]["x_half_ms"]) == 2 and "configs[3]" in hw["workload"]


@pytest.mark.parametrize("fault", ["rank1_raises", "hang"])
def test_bench_line_survives_a_failing_hugewiki_leg(alslib, fault):
    """the driver gets one shot at N = 8, and the hugewiki leg runs behind the Netflix line in
    the same process.  Whatever happens to the leg, the ONE line must come out, with the Netflix numbers intact and
    `hugewiki: {"error": ...}`:
      rank1_raises  CUMF_BENCH_FAIL_LEG=1: rank 1 fails while preparing its slab -- the ranks agree on that with one all-reduce
                    before the leg's first collective and both skip it;
      hang          CUMF_BENCH_FAIL_LEG=hang1: rank 1 never arrives at that all-reduce -- rank 0's LineGuard prints the line
                    when its deadline (CUMF_BENCH_LEG_DEADLINE) passes and the ranks leave."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import json
    import os
    import subprocess
    import sys

    from tests.test_dist_cpu import _free_port

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CUMF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
               CUMF_BENCH_FAIL_LEG="1" if fault == "rank1_raises" else "hang1", CUMF_BENCH_LEG_DEADLINE="20")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.05", "--no-cpu-baseline", "--scheme", "gather"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and np.isfinite(line["value"]) and line["ranks"]["n_ranks_seen"] == 2
    assert "error" in line["hugewiki"] and "value" not in line["hugewiki"], line["hugewiki"]
    if fault == "rank1_raises":
        assert out.returncode == 0, out.stderr[-2000:]
        assert "rank 1" in line["hugewiki"]["error"] or "another rank" in line["hugewiki"]["error"]


@pytest.mark.parametrize("fault", ["rank1_raises", "hang"])
def test_bench_line_survives_a_failing_native_leg(alslib, fault):
    """Round 6: the native half-iterations are measured BEHIND the line that torch.distributed collectives produced; when
    the native leg fails (CUMF_BENCH_FAIL_NATIVE=1: agreed on by one all-reduce, both ranks skip it) the line keeps those
    numbers, says so, and the hugewiki leg still runs; when it hangs (hang1) the LineGuard prints the line at its deadline
    with `native: {"error": ...}`."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    import json
    import os
    import subprocess
    import sys

    from tests.test_dist_cpu import _free_port

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CUMF_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1",
               CUMF_BENCH_FAIL_NATIVE="1" if fault == "rank1_raises" else "hang1", CUMF_BENCH_LEG_DEADLINE="60")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--scale", "0.05", "--no-cpu-baseline", "--scheme", "gather"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and np.isfinite(line["value"]) and line["ranks"]["n_ranks_seen"] == 2
    assert "error" in line["native"] and "torch_collectives" not in line, line
    assert len(line["ranks"]["per_rank"]["x_half_ms"]) == 2   # the diagnostics of the path the numbers come from
    if fault == "rank1_raises":
        assert out.returncode == 0, out.stderr[-2000:]
        assert line["collectives"] == "torch.distributed"
        assert line["hugewiki"]["value"] > 0   # the next leg still ran


def test_quadratic_sse_terms_kernel(alslib):
    """cumf_quadratic_sse_terms: sum over a batch of 2 x.b - x^T A x + reg |x|^2 against numpy fp64; systems marked with reg < 0
    (a column without ratings, NaN solution) are skipped, reg == 0 (lambda = 0) is a system like any other."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from cumf_als_amd import als

    rng = np.random.RandomState(0)
    for f in (10, 64, 100, 200):
        batch = 37
        g = rng.standard_normal((batch, f, f + 5)).astype(np.float32)
        reg = (0.05 * rng.randint(1, 50, size=batch)).astype(np.float32)
        A = np.einsum("bik,bjk->bij", g, g).astype(np.float32) + reg[:, None, None] * np.eye(f, dtype=np.float32)
        b = rng.standard_normal((batch, f)).astype(np.float32)
        x = (0.1 * rng.standard_normal((batch, f))).astype(np.float32)
        reg[3] = -1.0
        x[3] = np.nan
        A[5] -= reg[5] * np.eye(f, dtype=np.float32)
        reg[5] = 0.0
        A64, b64, x64, r64 = (v.astype(np.float64) for v in (A, b, x, reg))
        q = 2.0 * (x64 * b64).sum(1) - np.einsum("bi,bij,bj->b", x64, A64, x64) + r64 * (x64 * x64).sum(1)
        want = q[reg >= 0].sum()
        got = float(als.quadratic_sse_terms(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(), torch.from_numpy(x).cuda(),
                                            torch.from_numpy(reg).cuda()).item())
        scale = np.abs(q[reg >= 0]).sum()
        assert abs(got - want) <= 2e-6 * scale, (f, got, want)


@pytest.mark.parametrize("scheme,solver", [("reduce", "lu"), ("reduce", "cg"), ("gather", "lu"), ("gather", "cg")])
def test_distals_train_sse_out_of_the_theta_update(alslib, scheme, solver):
    """DistALS.update_theta(train_sse=True) with the HIP ops (one rank, no process group): `reduce` -- the quadratic form of
    the materialised systems; `gather` -- the fused kernels' own SSE (cumf_als_update_fused_sse).  Against the RMSE kernel
    over the ratings."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    from cumf_als_amd import als, datagen
    from cumf_als_amd import dist as cdist

    f, lam = 100, 0.05
    r = datagen.synth_ratings(3000, 800, 200000, 500, seed=2, device="cuda")
    theta0 = (0.2 * np.random.RandomState(0).random_sample((r.n, f))).astype(np.float32)
    ops = cdist.HipOps(torch.device("cuda"))
    if scheme == "reduce":
        xb = np.array([0, r.m], dtype=np.int64)
        eng = cdist.DistALS.from_local_slab(r.m, r.n, xb, r.csr_indptr, r.csr_indices, r.csr_data, f, lam, ops, solver=solver,
                                            theta_batch=3)
    else:
        eng = cdist.DistALS.from_device_ratings(r, f, lam, ops, solver=solver)
    eng.init_factors(theta0)
    eng.iterate(1)
    eng.update_x()
    got = eng.update_theta(train_sse=True)
    assert got is not None
    torch.cuda.synchronize()
    want = float(als.sse(r.csr_data, r.coo_row, r.csr_indices, eng.thetaT, eng.full_XT()).item())
    print(f"DistALS {scheme} {solver}: train SSE from the Theta update {got:.4f}, RMSE kernel {want:.4f}, rel {abs(got - want) / want:.2e}")
    assert abs(got - want) <= 2e-5 * want, (got, want)
    eng.close()
