#!/usr/bin/env python3
"""BASELINE.json configs[0]: MovieLens-10M shape (71567 x 65133, 9 000 048 ratings, 1 000 006 test
ratings; main.cpp:26), f = 10, lambda = 0.05, X_BATCH = THETA_BATCH = 1, SINGLE-THREAD CPU ALS -- the
plumbing config, no GPU.  Runs the CPU oracle (oracle/, the line-cited restatement of the reference's
arithmetic; the reference has no CPU path of its own) with one OpenMP thread and records ratings/s
per half-iteration and the RMSE after one iteration.

    python tools/cpu_config1_record.py profiles/r02/cpu_ml10m_f10_1thread.json

Test infrastructure (imports oracle/); never part of the product path.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cumf_als_amd import datagen  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main(out_path, solver="cg"):
    pyoracle.build()
    shp = datagen.SHAPES["ml10m"]
    m, n, nnz, nnz_test, lam, f = shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], shp["lam"], 10
    t0 = time.time()
    r = datagen.synth_ratings(m, n, nnz, nnz_test, seed=0, device="cpu")
    d = r.numpy()
    t_gen = time.time() - t0
    pyoracle.set_num_threads(1)
    assert pyoracle.num_threads() == 1
    th, x = pyoracle.init_factors(m, n, f)  # main.cpp:72-78: srand(0), 0.2 * rand() / RAND_MAX; X = 0
    res = {"config": "BASELINE.json configs[0]: ML-10M shape f=10, X_BATCH=1 THETA_BATCH=1, single-thread CPU reference ALS",
           "shape": {"m": m, "n": n, "nnz": nnz, "nnz_test": nnz_test, "f": f, "lambda": lam},
           "data": "synthetic (cumf_als_amd.datagen.synth_ratings, seed 0)", "threads": pyoracle.num_threads(),
           "host": os.uname().nodename, "cpu_count": os.cpu_count(), "kind": "port (oracle/als_oracle.c)"}
    for solver in ("cg", "lu"):
        thx, xx = th.copy(), x.copy()
        t = {}
        t0 = time.time()
        xx = xx.reshape(m, f)
        t["x"] = pyoracle.time_half_iteration(d["csr_indptr"], d["csr_indices"], d["csr_data"], thx, xx, f, lam, solver=solver)
        thx = thx.reshape(n, f)
        t["theta"] = pyoracle.time_half_iteration(d["csc_indptr"], d["csc_indices"], d["csc_data"], xx, thx, f, lam, solver=solver)
        wall = time.time() - t0
        tr = pyoracle.sse(d["csr_data"], d["coo_row"], d["csr_indices"], thx, xx, nnz, f)
        te = pyoracle.sse(d["test_data"], d["test_row"], d["test_col"], thx, xx, nnz_test, f)
        res[solver] = {"x_half_iteration_s": t["x"], "theta_half_iteration_s": t["theta"],
                       "ratings_per_s_per_half_iteration": 2.0 * nnz / (t["x"] + t["theta"]), "wall_s": wall,
                       "rmse_train_after_1_iteration": float(np.sqrt(tr / nnz)),
                       "rmse_test_after_1_iteration": float(np.sqrt(te / nnz_test))}
    res["gen_seconds"] = t_gen
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02", "cpu_ml10m_f10_1thread.json"))
