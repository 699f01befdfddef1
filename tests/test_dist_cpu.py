"""N > 1 path on CPU: world_size 2 (and 3, 4, 8: round 6), gloo, one process per rank (torch.multiprocessing spawn).

Checks that both partitionings of cumf_als_amd.dist reproduce the single-process result:
  * "gather": row slabs of X and Theta + one all-gather per half-iteration -- factors are
    bit-identical to the single-process oracle (each row's arithmetic is unchanged);
  * "reduce" (hugewiki scheme): partial Gram / RHS per X slab, reduce-scatter, solve,
    all-gather -- the Gram is summed in a different order, so factors agree to fp32
    rounding (and RMSE to 1e-4).
"""
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests import dist_helpers


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world, scheme, solver, d, m, n, f, lam, iters, theta_batch, theta0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=dist_helpers.worker,
                         args=(r, world, port, scheme, solver, d, m, n, f, lam, iters, theta_batch, theta0, q))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(outs, key=lambda t: t[0])


def test_balanced_slabs_and_local_csc():
    from cumf_als_amd import datagen
    from cumf_als_amd import dist as cdist

    r = datagen.synth_ratings(90, 70, 2500, 300, seed=3, row_alpha=1.2)
    d = r.numpy()
    for parts in (1, 2, 3, 8):
        b = cdist.balanced_slabs(d["csr_indptr"], parts)
        assert b[0] == 0 and b[-1] == r.m and (np.diff(b) >= 0).all() and len(b) == parts + 1
        nnz = np.diff(d["csr_indptr"][b])
        assert nnz.sum() == r.nnz and nnz.max() <= r.nnz / parts + np.diff(d["csr_indptr"]).max()
    # slab-local CSC: stacking the slabs' transposes reproduces the global CSC
    b = cdist.balanced_slabs(d["csr_indptr"], 3)
    cols = [[] for _ in range(r.n)]
    for g in range(3):
        rp, ci, va = cdist.slice_csr(d["csr_indptr"], d["csr_indices"], d["csr_data"], int(b[g]), int(b[g + 1]))
        cp, ri, cv = cdist.local_csc_of_slab(np.asarray(rp), np.asarray(ci), np.asarray(va), r.n)
        assert cp[-1] == len(ci) and (ri < (b[g + 1] - b[g])).all()
        for c in range(r.n):
            for k in range(cp[c], cp[c + 1]):
                cols[c].append((int(ri[k]) + int(b[g]), float(cv[k])))
    for c in range(r.n):
        s, e = d["csc_indptr"][c], d["csc_indptr"][c + 1]
        assert cols[c] == list(zip(d["csc_indices"][s:e].tolist(), d["csc_data"][s:e].tolist()))


@pytest.mark.parametrize("scheme", ["gather", "reduce"])
def test_world2_per_side_solvers(oracle, scheme):
    """The reference's hugewiki run solves X by CG and Theta by the batched LU (hugewiki.cu:2569, 2732): per-side solvers of
    DistALS, two ranks over gloo, against the oracle's half-iterations with the same solver per side."""
    from cumf_als_amd import datagen

    m, n, f, lam, iters = 60, 50, 10, 0.05, 2
    r = datagen.synth_ratings(m, n, 2400, 300, seed=11, row_alpha=1.1)
    d = {k: v for k, v in r.numpy().items()}
    theta0 = (0.2 * np.random.RandomState(0).random_sample((n, f))).astype(np.float32)
    th_ref, x_ref = theta0.copy(), np.zeros((m, f), np.float32)
    for _ in range(iters):
        oracle.half_iteration(d["csr_indptr"], d["csr_indices"], d["csr_data"], th_ref, x_ref, f, lam, solver="cg", cg_iters=4)
        oracle.half_iteration(d["csc_indptr"], d["csc_indices"], d["csc_data"], x_ref, th_ref, f, lam, solver="lu")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    side = dict(solver_x="cg", cg_iters_x=4, solver_theta="lu")
    procs = [ctx.Process(target=dist_helpers.worker,
                         args=(rk, 2, port, scheme, "cg", d, m, n, f, lam, iters, 2 if scheme == "reduce" else 1, theta0, q,
                               "oracle", side))
             for rk in range(2)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, th, x in outs:
        if scheme == "gather":   # every row's arithmetic is unchanged
            np.testing.assert_array_equal(th, th_ref)
            np.testing.assert_array_equal(x, x_ref)
        else:                    # the Gram is summed over two slabs in another order
            assert np.abs(th - th_ref).max() <= 1e-4 * np.abs(th_ref).max()
            assert np.abs(x - x_ref).max() <= 1e-3 * np.abs(x_ref).max()


def test_cost_balanced_slabs():
    """Slabs balanced by ratings + solve cost per row: on a side with many short rows the solves are half of the
    time, so equal-nnz slabs would leave the rank that holds the short rows with twice the work."""
    from cumf_als_amd import dist as cdist

    # 1000 heavy rows (500 ratings) followed by 100 000 light rows (5 ratings): equal nnz either half
    lens = np.concatenate([np.full(1000, 500), np.full(100000, 5)])
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    by_nnz = cdist.balanced_slabs(rowptr, 2)
    assert by_nnz[1] == 1000                                    # half of the ratings, 1 % of the rows
    w = cdist.solve_row_cost(100, "lu")
    by_cost = cdist.balanced_slabs(rowptr, 2, w)
    cost = lambda a, b: (rowptr[b] - rowptr[a]) + w * (b - a)
    c0, c1 = cost(0, by_cost[1]), cost(by_cost[1], len(lens))
    assert abs(c0 - c1) <= 0.01 * (c0 + c1) and by_cost[1] > 40000
    assert cdist.solve_row_cost(100, "lu") > cdist.solve_row_cost(100, "cg") > 0
    pb = cdist.pipeline_bounds(rowptr, by_cost, 4, w)
    assert pb.shape == (2, 5) and (np.diff(pb, axis=1) >= 0).all()
    assert pb[0, 0] == 0 and pb[0, -1] == by_cost[1] == pb[1, 0] and pb[1, -1] == len(lens)


def _matches_single_process(oracle, world, scheme, solver, m, n, theta_batch):
    from cumf_als_amd import datagen

    f, lam, iters = 10, 0.05, 3
    r = datagen.synth_ratings(m, n, 2400, 300, seed=11, row_alpha=1.1)
    d = {k: v for k, v in r.numpy().items()}
    theta0 = (0.2 * np.random.RandomState(0).random_sample((n, f))).astype(np.float32)
    th_ref, x_ref = theta0.copy(), np.zeros((m, f), np.float32)
    oracle.do_als(d, th_ref, x_ref, m, n, f, lam, iters, solver=solver)
    outs = _run(world, scheme, solver, d, m, n, f, lam, iters, theta_batch if scheme == "reduce" else 1, theta0)
    assert [o[0] for o in outs] == list(range(world))
    for rank, th, x in outs:
        if scheme == "gather":
            np.testing.assert_array_equal(th, th_ref)
            np.testing.assert_array_equal(x, x_ref)
        else:
            # LU: fp32 rounding of a re-ordered Gram sum.  CG: the truncated, threshold-stopped
            # iteration amplifies that rounding, so the bound is on the model
            # quality (train RMSE to 1e-4) plus a looser factor tolerance.
            tol = 2e-4 if solver == "lu" else 3e-3
            assert np.abs(th - th_ref).max() <= tol * np.abs(th_ref).max()
            assert np.abs(x - x_ref).max() <= tol * np.abs(x_ref).max()
            sse = oracle.sse(d["csr_data"], d["coo_row"], d["csr_indices"], th, x, r.nnz, f)
            sse_ref = oracle.sse(d["csr_data"], d["coo_row"], d["csr_indices"], th_ref, x_ref, r.nnz, f)
            assert abs((sse / r.nnz) ** 0.5 - (sse_ref / r.nnz) ** 0.5) <= 1e-4
    # every rank ends with the same replicas
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0][1], o[1])
        np.testing.assert_array_equal(outs[0][2], o[2])


@pytest.mark.parametrize("scheme,solver", [("gather", "cg"), ("gather", "lu"), ("reduce", "lu"), ("reduce", "cg")])
def test_world2_matches_single_process(oracle, scheme, solver):
    _matches_single_process(oracle, 2, scheme, solver, 60, 50, 2)


@pytest.mark.parametrize("world,scheme,solver,theta_batch", [(3, "gather", "lu", 1), (3, "reduce", "cg", 3), (4, "reduce", "lu", 2),
                                                             (8, "gather", "cg", 1), (8, "reduce", "lu", 3)])
def test_world_3_4_8_matches_single_process(oracle, world, scheme, solver, theta_batch):
    """the arithmetic that only appears above two ranks -- k = ceil(size / world) systems per rank and
    Theta batch with a short or EMPTY last share (n = 53 columns, THETA_BATCH = 3: batches of 17, 17, 19 over eight ranks
    give k = 3: the sixth rank holds 2 / 2 / 3 systems... the last ranks none), eight uneven slabs padded to the largest in
    SlabGather (m = 61 rows), slabs of a handful of rows -- on the CPU with the stand-in ops: `gather` bit-identical to the
    single-process oracle, `reduce` to rounding."""
    _matches_single_process(oracle, world, scheme, solver, 61, 53, theta_batch)


@pytest.mark.parametrize("solver", ["cg", "lu"])
def test_hugewiki_runner_from_split_files(tmp_path, solver):
    """convert split -> cumf_als_amd.hugewiki.run on 2 ranks (per-GPU slab files, no rank holds the
    whole matrix) reproduces the single-process oracle doALS: RMSE log to 1e-4."""
    from cumf_als_amd import convert, datagen
    from oracle import pyoracle

    m, n, f, lam, iters = 120, 60, 10, 0.05, 3
    r = datagen.synth_ratings(m, n, 3000, 400, seed=11, device="cpu")
    d = r.numpy()
    datagen.write_dataset(r, str(tmp_path / "d"))
    convert.split_dataset(str(tmp_path / "d"), str(tmp_path / "s"), 2, m, n, r.nnz, r.nnz_test)
    th0, x0 = pyoracle.init_factors(m, n, f)
    _, log_o = pyoracle.do_als(d, th0, x0, m, n, f, lam, iters, solver=solver, test_grid_compat=False)

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=dist_helpers.hugewiki_worker,
                         args=(rk, 2, port, str(tmp_path / "s"), n, f, lam, iters, solver, q)) for rk in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(outs[0][1], outs[1][1])  # Theta replicas identical
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    log = np.array(outs[0][3])
    assert np.abs(log - np.asarray(log_o)).max() <= 1e-4, (log, log_o)
    assert np.abs(outs[0][1] - th0.reshape(n, f)).max() <= 2e-3 * max(1.0, np.abs(th0).max())


@pytest.mark.parametrize("solver,theta_batch,world", [("lu", 1, 2), ("cg", 3, 2), ("lu", 3, 3), ("cg", 2, 8)])
def test_reduce_scheme_train_sse_from_the_reduced_systems(oracle, solver, theta_batch, world):
    """Round 4: in the `reduce` scheme the train SSE comes out of the Theta update -- sum r^2 (a constant of the data,
    all-reduced once) minus the all-reduced sum over every rank's systems of 2 t.b - t^T G t -- instead of a pass over
    every rank's ratings (hugewiki.cu:2750-2862).  Two ranks against the direct evaluation on the full factors."""
    from cumf_als_amd import datagen

    m, n, f, lam = 90, 40, 10, 0.05
    r = datagen.synth_ratings(m, n, 2000, 100, seed=4, row_alpha=1.1)
    d = {k: v for k, v in r.numpy().items()}
    theta0 = (0.2 * np.random.RandomState(1).random_sample((n, f))).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=dist_helpers.train_sse_worker,
                         args=(rk, world, port, solver, d, m, n, f, lam, theta_batch, theta0, q)) for rk in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[1] == outs[0][1] for o in outs)         # the all-reduced value is the same on every rank
    direct = oracle.sse(d["csr_data"], d["coo_row"], d["csr_indices"], outs[0][2], outs[0][3], r.nnz, f, dtype=np.float64)
    assert abs(outs[0][1] - direct) <= 2e-5 * direct, (outs[0][1], direct)


@pytest.mark.parametrize("unavailable_rank,world", [(-1, 2), (1, 2), (2, 4)])
def test_gather_scheme_train_sse_is_decided_collectively(oracle, unavailable_rank, world):
    """in the `gather` scheme the fused train SSE depends on each rank's own plans.  When only ONE
    rank's plans refuse it, every rank must fall back together (update_theta returns None everywhere) -- a rank that
    skipped the all-reduce of the bins alone would pair its next collective with the others' and hang or sum nonsense."""
    from cumf_als_amd import datagen

    m, n, f, lam = 90, 40, 10, 0.05
    r = datagen.synth_ratings(m, n, 2000, 100, seed=4, row_alpha=1.1)
    d = {k: v for k, v in r.numpy().items()}
    theta0 = (0.2 * np.random.RandomState(1).random_sample((n, f))).astype(np.float32)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=dist_helpers.gather_sse_worker,
                         args=(rk, world, port, "lu", d, m, n, f, lam, theta0, unavailable_rank, q)) for rk in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fused, direct = [o[1] for o in outs], [o[2] for o in outs]
    assert all(v == direct[0] for v in direct) and direct[0] > 0   # the fall-back's own all-reduce paired up
    if unavailable_rank < 0:
        assert all(v == fused[0] for v in fused) and abs(fused[0] - direct[0]) <= 1e-6 * direct[0], (fused, direct)
    else:
        assert all(v is None for v in fused), fused


def test_pipeline_bounds_and_row_map():
    """The pipelined all-gather of the X update (dist.PipelinedGather): every rank's slab in nnz-balanced
    pieces, computed identically everywhere; the row map sends every global row to exactly one slot of the
    padded receive buffer."""
    import numpy as np
    import torch

    from cumf_als_amd import dist as cdist

    rng = np.random.RandomState(0)
    lens = rng.randint(0, 50, size=1000)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for world, chunks in ((2, 4), (8, 4), (3, 1), (5, 7)):
        xb = cdist.balanced_slabs(rowptr, world)
        pb = cdist.pipeline_bounds(rowptr, xb, chunks)
        assert pb.shape == (world, chunks + 1)
        assert (pb[:, 0] == xb[:-1]).all() and (pb[:, -1] == xb[1:]).all()
        assert (np.diff(pb, axis=1) >= 0).all()
        g = cdist.PipelinedGather(pb, 1000, 3, torch.float32, torch.device("cpu"))
        idx = g.idx.numpy()
        assert len(np.unique(idx)) == 1000 and idx.max() < g.recv.shape[0]
        # emulate the collectives: piece c of rank r lands at off[c] + r * mx[c]
        full = torch.arange(3000, dtype=torch.float32).reshape(1000, 3)
        for c in range(chunks):
            for r in range(world):
                lo, hi = int(pb[r, c]), int(pb[r, c + 1])
                base = int(g.off[c]) + r * g.mx[c]
                g.recv[base: base + hi - lo] = full[lo:hi]
        out = torch.empty_like(full)
        g.finish(out)
        assert torch.equal(out, full)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_world2_pipelined_gather_equals_blocking(monkeypatch, world):
    """gather scheme, 2 / 3 / 8 ranks: the pipelined all-gathers (4 pieces per side: what the default picks for a factor
    matrix >= 32 MB) and the blocking ones (CUMF_ALS_PIPE_CHUNKS=1: the default below that) give bit-identical
    factors.  Eight ranks x four pieces of 70 / 50 rows: pieces of one or two rows, some empty on some ranks."""
    import numpy as np

    from cumf_als_amd import datagen

    m, n, f, lam, iters = 70, 50, 10, 0.05, 2
    r = datagen.synth_ratings(m, n, 2600, 300, seed=13, row_alpha=1.1)
    d = {k: v for k, v in r.numpy().items()}
    theta0 = (0.2 * np.random.RandomState(0).random_sample((n, f))).astype(np.float32)
    res = {}
    for chunks in ("4", "1"):
        monkeypatch.setenv("CUMF_ALS_PIPE_CHUNKS", chunks)  # inherited by the spawned ranks
        res[chunks] = _run(world, "gather", "lu", d, m, n, f, lam, iters, 1, theta0)
    for a, b in zip(res["4"], res["1"]):
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])


def test_default_pipeline_pieces_follow_the_gathered_bytes(monkeypatch):
    """Without CUMF_ALS_PIPE_CHUNKS a side is cut into 4 pieces only when its gathered factor matrix is >= 32 MB
    (Netflix f = 100: Theta 192 MB yes, X 7 MB no)."""
    import numpy as np

    from cumf_als_amd import dist as cdist

    monkeypatch.delenv("CUMF_ALS_PIPE_CHUNKS", raising=False)
    monkeypatch.setenv("CUMF_ALS_PIPE_FORCE", "1")

    class _Ops:
        def plan(self, rowptr, f, chunk=0, row_begin=0, row_end=None):
            return (row_begin, row_end)

    eng = cdist.DistALS.__new__(cdist.DistALS)
    eng.f, eng.world, eng.rank, eng.ops = 100, 1, 0, _Ops()
    monkeypatch.setattr(cdist.dist, "is_initialized", lambda: True)
    for rows, want in ((17770, None), (480189, 4)):
        rowptr = np.arange(rows + 1, dtype=np.int64) * 3
        pipe = eng._make_pipeline(rowptr, rowptr, np.array([0, rows]), 0)
        assert (pipe is None) if want is None else (pipe[0].shape == (1, want + 1)), rows


def test_train_sse_near_perfect_fit_is_not_reported():
    """the train SSE out of the Theta update is an fp32 identity per column; below 1e-3 of sum r^2 it is
    cancellation noise (possibly negative: a NaN RMSE in the caller).  DistALS then returns None -- as doALS re-evaluates
    with the RMSE kernel -- and the caller runs `slab_sse`; the yardstick sum r^2 is all-reduced once, the decision is taken
    on the all-reduced value: the same on every rank."""
    from cumf_als_amd import dist as cdist

    eng = cdist.DistALS.__new__(cdist.DistALS)
    eng.scheme, eng.world, eng.group = "gather", 1, None
    eng.t_val = torch.tensor([3.0, 4.0, 5.0, 1.0, 7.0])          # sum r^2 = 100
    eng.thetaT = torch.zeros(1)
    eng._sum_r2 = None
    assert eng._trusted_sse(5.0) == 5.0 and eng._sum_r2 == 100.0
    assert eng._trusted_sse(0.05) is None and eng._trusted_sse(-1e-3) is None and eng._trusted_sse(0.1) == 0.1
