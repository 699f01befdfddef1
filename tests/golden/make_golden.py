#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ with the CPU oracle.

The reference ships no golden vectors and cannot run here (CUDA only), so
these fixtures are produced by oracle/als_oracle.c (fp32 build) on small hand-shaped inputs:
they pin the oracle against accidental change and give the GPU tier a data-only target that
needs neither the oracle build nor /root/reference at run time.  Regenerate with
    python tests/golden/make_golden.py
Each fx_*.npz holds: the ten on-disk arrays, m/n/f/lambda, initial factors, Gram + RHS of the
first rows of both sides, LU and CG(6) solutions of those systems, factors after 1 and 2 full
ALS iterations (LU and CG), and the per-iteration RMSE log (reference-compatible truncated
test grid AND exact grid).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cumf_als_amd import datagen  # noqa: E402
from oracle import pyoracle  # noqa: E402

NROWS = 4  # systems kept per side


def shaped_ratings(m, n, lens, seed, nnz_test):
    """Rows with prescribed lengths (window boundaries 27/28/29 of SCAN_BATCH=28, als.cu:44;
    stage boundaries 31/32/33 of this repo; one empty row) + a random remainder."""
    rng = np.random.RandomState(seed)
    rows, cols = [], []
    for u, ln in enumerate(lens):
        c = rng.choice(n, size=min(ln, n), replace=False)
        rows += [u] * len(c)
        cols += sorted(c.tolist())
    rows, cols = np.array(rows), np.array(cols)
    vals = rng.randint(1, 6, size=len(rows)).astype(np.float32)
    tr = rng.randint(0, m, nnz_test)
    tc = rng.randint(0, n, nnz_test)
    tv = rng.randint(1, 6, size=nnz_test).astype(np.float32)
    return datagen.from_coo(m, n, rows, cols, vals, tr, tc, tv)


def build(name, m, n, f, lens, seed, lam=0.05, nnz_test=600):
    r = shaped_ratings(m, n, lens, seed, nnz_test)
    d = r.numpy()
    th0, x0 = pyoracle.init_factors(m, n, f)
    out = {k: v for k, v in d.items()}
    out.update(m=m, n=n, f=f, lam=np.float32(lam), theta0=th0, x0=x0)
    # X side systems from theta0; Theta side systems from a fixed pseudo-X
    xs = (0.2 * np.random.RandomState(seed + 1).random_sample((m, f))).astype(np.float32)
    for side, (ptr, idx, val, g) in {"x": (d["csr_indptr"], d["csr_indices"], d["csr_data"], th0),
                                     "t": (d["csc_indptr"], d["csc_indices"], d["csc_data"], xs)}.items():
        A, b = pyoracle.gram_rhs(ptr, idx, val, g, f, lam, 0, NROWS)
        out[f"{side}_gram"], out[f"{side}_rhs"] = A, b
        ok = np.diff(ptr[:NROWS + 1]) > 0
        out[f"{side}_lu"] = pyoracle.lu(A, b, f)
        out[f"{side}_cg6"] = pyoracle.cg(A, np.zeros_like(b), b, f, 6)
        out[f"{side}_nonempty"] = ok
    out["xs"] = xs
    for solver in ("lu", "cg"):
        for iters in (1, 2):
            th, x = th0.copy(), x0.copy()
            rm, log = pyoracle.do_als(d, th, x, m, n, f, lam, iters, solver=solver, test_grid_compat=True)
            out[f"theta_{solver}_{iters}"], out[f"x_{solver}_{iters}"] = th, x
            out[f"rmse_compat_{solver}_{iters}"] = log
            # fp64 arithmetic on the same fp32 data: its distance to the fp32 run is the
            # rounding-noise floor of this (possibly chaotic: truncated CG) trajectory
            th64, x64 = th0.copy(), x0.copy()
            pyoracle.do_als(d, th64, x64, m, n, f, lam, iters, solver=solver, test_grid_compat=True,
                            dtype=np.float64)
            out[f"noise_{solver}_{iters}"] = np.float32(max(np.nanmax(np.abs(th64 - th)), np.nanmax(np.abs(x64 - x))))
            th, x = th0.copy(), x0.copy()
            rm, log = pyoracle.do_als(d, th, x, m, n, f, lam, iters, solver=solver, test_grid_compat=False)
            out[f"rmse_exact_{solver}_{iters}"] = log
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "nnz", r.nnz, "bytes", os.path.getsize(os.path.join(HERE, name + ".npz")))


if __name__ == "__main__":
    pyoracle.build()
    # fx_tiny: m=7, n=5, one empty row (row 3), one full row (row 6)
    build("fx_tiny", 7, 5, 10, [2, 3, 1, 0, 4, 2, 5], seed=1, nnz_test=300)
    # window/stage boundary rows; n large enough for rows of 27..33 and 64, 65, 100
    lens = [27, 28, 29, 31, 32, 33, 1, 2, 3, 4, 5, 63, 64, 65, 100, 0, 12, 40, 56, 57, 7, 96, 97, 48]
    build("fx_f20", 24, 128, 20, lens, seed=2)
    build("fx_f64", 24, 128, 64, lens, seed=3)
    build("fx_f100", 24, 128, 100, lens, seed=4)
