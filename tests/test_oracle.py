"""Pin the CPU oracle (oracle/als_oracle.c) without a runnable reference.

The reference is CUDA-only and ships no golden vectors, so the oracle
is pinned by: an independent numpy fp64 formulation of the same normal equations, its own
fp64 build, algebraic properties, the libc known-answer of the factor initialisation, and
the committed fixtures under tests/golden/ (tests/test_golden.py).
"""
import numpy as np
import pytest


def _data(m=40, n=30, nnz=500, nnz_test=300, seed=0, **kw):
    from cumf_als_amd import datagen

    r = datagen.synth_ratings(m, n, nnz, nnz_test, seed=seed, **kw)
    return r, r.numpy()


def _theta(n, f, seed=1):
    return (0.2 * np.random.RandomState(seed).random_sample((n, f))).astype(np.float32)


def _numpy_normal_equations(d, theta, f, lam):
    """Independent formulation: dense fp64 einsum over each row's gathered factors."""
    ptr, idx, val = d["csr_indptr"], d["csr_indices"], d["csr_data"]
    rows = len(ptr) - 1
    A = np.zeros((rows, f, f))
    b = np.zeros((rows, f))
    for u in range(rows):
        cols = idx[ptr[u]:ptr[u + 1]]
        T = theta[cols].astype(np.float64)
        A[u] = T.T @ T + lam * len(cols) * np.eye(f)
        b[u] = T.T @ val[ptr[u]:ptr[u + 1]].astype(np.float64)
    return A, b


@pytest.mark.parametrize("f", [10, 20, 64, 100])
def test_gram_rhs_matches_numpy_fp64(oracle, f):
    r, d = _data()
    theta = _theta(r.n, f)
    lam = np.float32(0.05)
    A_np, b_np = _numpy_normal_equations(d, theta, f, float(lam))
    A64, b64 = oracle.gram_rhs(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, f, lam, dtype=np.float64)
    A32, b32 = oracle.gram_rhs(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, f, lam, dtype=np.float32)
    assert np.abs(A64 - A_np).max() <= 1e-12 * np.abs(A_np).max()
    assert np.abs(b64 - b_np).max() <= 1e-12 * np.abs(b_np).max()
    assert np.abs(A32 - A_np).max() <= 2e-6 * np.abs(A_np).max()
    assert np.abs(b32 - b_np).max() <= 2e-6 * np.abs(b_np).max()
    # properties: symmetric (bitwise, both triangles written from one value), PSD without the ridge
    assert np.array_equal(A32, A32.transpose(0, 2, 1))
    ptr = d["csr_indptr"]
    for u in range(0, len(ptr) - 1, 7):
        ridge = float(lam) * (ptr[u + 1] - ptr[u])
        w = np.linalg.eigvalsh(A64[u] - ridge * np.eye(f))
        assert w.min() >= -1e-9 * max(1.0, w.max())


def test_empty_row_is_all_zero(oracle):
    from cumf_als_amd import datagen

    r = datagen.from_coo(3, 3, [0, 0, 2], [0, 1, 2], [1, 2, 3], [0], [0], [1.0])
    d = r.numpy()
    A, b = oracle.gram_rhs(d["csr_indptr"], d["csr_indices"], d["csr_data"], _theta(3, 10), 10, 0.05)
    assert not A[1].any() and not b[1].any()  # als.cu:455: no loads, lambda * 0 on the diagonal


@pytest.mark.parametrize("f", [10, 40, 100])
def test_lu_and_cg_solve(oracle, f):
    r, d = _data(60, 50, 2500, 300, seed=2)
    theta = _theta(r.n, f)
    A64, b64 = oracle.gram_rhs(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, f, 0.05, dtype=np.float64)
    A32, b32 = A64.astype(np.float32), b64.astype(np.float32)
    x_np = np.linalg.solve(A64, b64[..., None])[..., 0]
    x64 = oracle.lu(A64, b64, f)
    x32 = oracle.lu(A32, b32, f)
    assert np.abs(x64 - x_np).max() <= 1e-9 * np.abs(x_np).max()
    assert np.abs(x32 - x_np).max() <= 5e-4 * np.abs(x_np).max()
    # CG run to convergence (f iterations, exact arithmetic => exact) equals the LU solution
    x0 = np.zeros_like(b64)
    xcg = oracle.cg(A64, x0, b64, f, 3 * f)
    # it stops once ||r||^2 < 1e-4 (cg.cu:31,195), so ||x - x*||_2 <= ||r||_2 / lambda_min <= 1e-2 / lambda_min
    lam_min = np.linalg.eigvalsh(A64)[:, 0]
    assert (np.linalg.norm(xcg - x_np, axis=1) <= 1.0001e-2 / lam_min).all()
    assert (np.linalg.norm(np.einsum("bij,bj->bi", A64, xcg) - b64, axis=1) ** 2 < 1e-4).all()
    # 6 warm-started iterations reduce the residual monotonically in the A-norm
    x6 = oracle.cg(A64, x0, b64, f, 6)
    e0 = np.einsum("bi,bij,bj->b", x_np, A64, x_np)
    e6 = np.einsum("bi,bij,bj->b", x6 - x_np, A64, x6 - x_np)
    assert (e6 <= e0 * (1 + 1e-12)).all()
    # zero iterations leave the warm start untouched (cg.cu:85 loop not entered, cg.cu:230 write-back)
    xw = _theta(len(b64), f, 5).astype(np.float64)
    assert np.array_equal(oracle.cg(A64, xw, b64, f, 0), xw)


def test_batches_do_not_change_results(oracle):
    """als.cu:768-777: rows are independent; X_BATCH / THETA_BATCH only slice them."""
    r, d = _data(50, 40, 1500, 300, seed=3)
    f = 20
    for solver in ("cg", "lu"):
        outs = []
        for xb, tb in ((1, 1), (3, 2), (7, 5)):
            th, x = oracle.init_factors(r.m, r.n, f)
            rm, log = oracle.do_als(d, th, x, r.m, r.n, f, 0.05, 3, xb, tb, solver=solver)
            outs.append((th.copy(), x.copy(), rm))
        for th, x, rm in outs[1:]:
            assert np.array_equal(th, outs[0][0]) and np.array_equal(x, outs[0][1]) and rm == outs[0][2]


def test_init_factors_known_answer(oracle):
    """main.cpp:72-78 with glibc: srand(0); 0.2*rand()/RAND_MAX."""
    th, x = oracle.init_factors(3, 5, 10)
    np.testing.assert_allclose(th.reshape(-1)[:4], [0.168037549, 0.0788765848, 0.156619847, 0.159688011], rtol=2e-7)
    assert not x.any()


def test_rmse_definitions(oracle):
    r, d = _data(60, 50, 2200, 700, seed=4)
    f = 10
    th = _theta(r.n, f, 1)
    x = _theta(r.m, f, 2)
    pred = np.einsum("ij,ij->i", x[d["coo_row"]].astype(np.float64), th[d["csr_indices"]].astype(np.float64))
    sse_np = ((d["csr_data"] - pred) ** 2).sum()
    sse = oracle.sse(d["csr_data"], d["coo_row"], d["csr_indices"], th, x, r.nnz, f)
    assert abs(sse - sse_np) <= 1e-5 * sse_np
    # truncated test grid of als.cu:1006: ((nnz_test-1)/256) blocks of 256
    cnt = ((r.nnz_test - 1) // 256) * 256
    assert cnt == 512
    full = oracle.sse(d["test_data"], d["test_row"], d["test_col"], th, x, r.nnz_test, f)
    trunc = oracle.sse(d["test_data"], d["test_row"], d["test_col"], th, x, cnt, f)
    assert trunc < full
    # SURPASS_NAN (als.cu:201-211): a NaN factor row stops that rating's dot product, e stays finite
    th2 = th.copy()
    th2[d["test_col"][0]] = np.nan
    assert np.isnan(oracle.sse(d["test_data"], d["test_row"], d["test_col"], th2, x, r.nnz_test, f))
    assert np.isfinite(oracle.sse(d["test_data"], d["test_row"], d["test_col"], th2, x, r.nnz_test, f, surpass_nan=True))


def test_als_converges_and_fp32_tracks_fp64(oracle):
    r, d = _data(200, 150, 20000, 2000, seed=5)
    f = 10
    th, x = oracle.init_factors(r.m, r.n, f)
    rm, log = oracle.do_als(d, th, x, r.m, r.n, f, 0.05, 6, solver="lu")
    assert log[-1, 0] < log[0, 0]  # train RMSE decreases
    th64, x64 = oracle.init_factors(r.m, r.n, f)
    rm64, log64 = oracle.do_als(d, th64, x64, r.m, r.n, f, 0.05, 6, solver="lu", dtype=np.float64)
    assert np.abs(log - log64).max() <= 1e-4
