#!/usr/bin/env python3
"""how many CG iterations the rows of a side actually run before ||r||^2 < 1e-4 ends the loop
(cg.cu:195), on real factors (after two full iterations) -- the profiling build's switch 65536.
  CUMF_ALS_LIB=cumf_als_amd/csrc/libALS_ablate.so python tools/cg_histogram.py [--shape hugewiki|netflix] [--cg-iters 6 100]
One JSON line per (side, cg_iters): rows by iterations run, mean iterations."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cumf_als_amd import als, datagen  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="hugewiki")
    ap.add_argument("--f", type=int, default=100)
    ap.add_argument("--cg-iters", type=int, nargs="*", default=[6, 100])
    a = ap.parse_args()
    shp = datagen.SHAPES[a.shape]
    if a.shape == "hugewiki":  # the 1/8 row slab one GPU holds (BASELINE.json configs[3])
        m, n, nnz = shp["m"] // 8, shp["n"], shp["nnz"] // 8
    else:
        m, n, nnz = shp["m"], shp["n"], shp["nnz"]
    r = datagen.synth_ratings(m, n, nnz, 4096, seed=0, device="cuda")
    eng = als.ALSEngine(r, a.f, shp["lam"], solver="cg", cg_iters=6)
    eng.init_factors()
    als.set_debug_switches(0)
    eng.iterate(2)
    torch.cuda.synchronize()
    keep_x, keep_t = eng.XT.clone(), eng.thetaT.clone()
    for iters in a.cg_iters:
        eng.cg_iters = iters
        for side, fn, keep, tgt in (("x", eng.update_x, keep_x, eng.XT), ("theta", eng.update_theta, keep_t, eng.thetaT)):
            als.set_debug_switches(65536)
            als.debug_cg_histogram(a.f)  # clear
            fn()
            torch.cuda.synchronize()
            hist = als.debug_cg_histogram(a.f)
            als.set_debug_switches(0)
            tgt.copy_(keep)
            rows = sum(hist)
            mean = sum(k * v for k, v in enumerate(hist)) / max(rows, 1)
            print(json.dumps({"shape": a.shape, "rows_of_side": m if side == "x" else n, "side": side, "f": a.f, "cg_iters": iters,
                              "rows_by_iterations_run (last bin: >= 15)": hist, "rows": rows, "mean_iterations": round(mean, 3)}), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
