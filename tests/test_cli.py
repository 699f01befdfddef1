"""./main -- the reference's command line (main.cpp:19-172) end to end on the GPU.

Writes a synthetic dataset in the on-disk format, runs the binary with the reference's ten
argv, scrapes the stdout lines print-test-result.sh:8-11 greps for, and compares the printed
RMSEs with the oracle started from the same libc-seeded factors (main.cpp:72-78)."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "cumf_als_amd", "csrc", "main")


def test_usage_without_gpu_needed():
    """argc != 10 prints the usage text and returns 0 (main.cpp:21-30) before touching the GPU."""
    if not os.path.exists(MAIN):
        pytest.skip("main not built")
    p = subprocess.run([MAIN], capture_output=True, text=True)
    assert p.returncode == 0 and "Usage: give M, N, F, NNZ, NNZ_TEST, lambda, X_BATCH, THETA_BATCH and DATA_DIR." in p.stdout
    p = subprocess.run([MAIN, "10", "10", "7", "10", "10", "0.05", "1", "1", "/tmp"], capture_output=True, text=True)
    assert "F has to be a multiple of 10" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["cg", "lu"])
def test_main_matches_oracle(oracle, alslib, tmp_path, solver):
    from cumf_als_amd import datagen

    m, n, f, lam = 400, 300, 20, 0.05
    r = datagen.synth_ratings(m, n, 40000, 3000, seed=21)
    datagen.write_dataset(r, str(tmp_path))
    env = dict(os.environ, CUMF_ALS_SOLVER=solver)
    p = subprocess.run([MAIN, str(m), str(n), str(f), str(r.nnz), str(r.nnz_test), str(lam), "1", "3", str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr
    out = p.stdout
    assert f"M = {m}, N = {n}, F = {f}, NNZ = {r.nnz}, NNZ_TEST = {r.nnz_test}" in out
    assert "*******parameters: m: 400, n:  300, f: 20, nnz: 40000 " in out
    assert ("\tCG solver with fp32." in out) == (solver == "cg")
    assert re.search(r"doALS takes seconds: \d+\.\d{3} for F = 20", out) and "ALS Done." in out
    train = [float(x) for x in re.findall(r"--------- Train RMSE in iter \d+: ([0-9.]+)", out)]
    test = [float(x) for x in re.findall(r"--------- Test RMSE in iter \d+: ([0-9.]+)", out)]
    assert len(train) == 10 and len(test) == 10  # ITERS = 10 (main.cpp:17)
    th, x = oracle.init_factors(m, n, f)
    rm, log = oracle.do_als(r.numpy(), th, x, m, n, f, lam, 10, 1, 3, solver=solver, test_grid_compat=True)
    tol = 2e-5 if solver == "lu" else 1e-4
    assert np.abs(np.array(train) - log[:, 0]).max() <= tol + 5e-7  # %f prints 6 decimals
    assert np.abs(np.array(test) - log[:, 1]).max() <= tol + 5e-7
