"""BASELINE.json configs at full size on one GPU: timing + RMSE sanity, one JSON line each.
  python tools/sweep_configs.py [all|f200|quick]      (run on the GPU box; see profiles/r02/config_sweep.jsonl)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cumf_als_amd import als, datagen  # noqa: E402


def run(shape, f, solver, fused=True, x_batch=1, theta_batch=1, iters=2, gram="auto"):
    shp = datagen.SHAPES[shape]
    als.set_gram_mode(gram)
    r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device="cuda")
    eng = als.ALSEngine(r, f, shp["lam"], solver=solver, fused=fused, x_batch=x_batch, theta_batch=theta_batch)
    eng.init_factors()
    eng.iterate(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.iterate(iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    tr, te = eng.rmse()
    print(json.dumps({"shape": shape, "f": f, "solver": solver, "fused": fused, "gram": gram, "batches": [x_batch, theta_batch],
                      "ms_per_iteration": round(dt * 1e3, 2), "ratings_per_s_half_iter_G": round(2 * r.nnz / dt / 1e9, 3),
                      "rmse_train": round(tr, 5), "rmse_test": round(te, 5)}), flush=True)
    del eng, r
    torch.cuda.empty_cache()


what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "quick"):
    run("ml10m", 10, "cg")
    run("ml10m", 10, "lu")
    run("netflix", 64, "cg")
    run("netflix", 64, "lu")
    run("netflix", 100, "cg")
    run("netflix", 100, "lu")
    run("netflix", 100, "lu", gram="exact")
    run("netflix", 100, "lu", gram="fast")
    run("netflix", 100, "cg", gram="fast")
    run("netflix", 100, "lu", fused=False, theta_batch=3)
if what in ("all", "f200"):
    run("netflix", 128, "lu")
    run("netflix", 128, "cg")
    run("netflix", 160, "lu", iters=1)
    run("netflix", 160, "cg", iters=1)
    run("netflix", 200, "cg", iters=1)
    run("netflix", 200, "lu", iters=1)
    run("netflix", 200, "cg", iters=1, gram="fast")
    run("netflix", 200, "cg", fused=False, theta_batch=10, iters=1)
