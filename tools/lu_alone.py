#!/usr/bin/env python3
"""Where a fused half-iteration's time goes: the same launches with the profiling build's switches
0 (everything), 1 (no solve: the Gram pass alone), 2 (no Gram pass: the solve alone, on lambda n I).
  CUMF_ALS_LIB=cumf_als_amd/csrc/libALS_ablate.so python tools/lu_alone.py [--f 100] [--solver lu]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cumf_als_amd import als, datagen  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--f", type=int, default=100)
    ap.add_argument("--solver", default="lu")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--extra", type=int, nargs="*", default=[],
                    help="further solve-only runs with these LU ablation bits OR-ed to 2 (256 no panel preparation, 512 no fp32 "
                         "MFMAs, 1024 no trailing update, 2048 no back substitution)")
    ap.add_argument("--raw", type=int, nargs="*", default=[], help="further runs with exactly these switch values")
    ap.add_argument("--only", default="", help="run just this variant (full / gram_only / solve_only): for counter passes, whose "
                    "per-kernel means must not mix the variants; the warm-up iteration then runs with the other solver")
    ap.add_argument("--shape", default="netflix", help="netflix | hugewiki (the 1/8 row slab one GPU holds)")
    a = ap.parse_args()
    shp = datagen.SHAPES[a.shape]
    if a.shape == "hugewiki":
        r = datagen.synth_ratings(shp["m"] // 8, shp["n"], shp["nnz"] // 8, 4096, seed=0, device="cuda")
    else:
        r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device="cuda")
    eng = als.ALSEngine(r, a.f, shp["lam"], solver=a.solver)
    eng.init_factors()
    als.set_debug_switches(0)
    if a.only:
        eng.solver = "cg" if a.solver == "lu" else "lu"
    eng.iterate(1)
    eng.solver = a.solver
    keep_x, keep_t = eng.XT.clone(), eng.thetaT.clone()
    out = {"library": os.path.basename(os.environ.get("CUMF_ALS_LIB", "libALS.so")), "shape": a.shape, "f": a.f, "solver": a.solver}
    als.set_kernel_timing(True)
    for sw, name in [(0, "full"), (1, "gram_only"), (2, "solve_only")] + [(2 | e, f"solve_only+{e}") for e in a.extra] + [(v, f"switches_{v}") for v in a.raw]:
        if a.only and name != a.only:
            continue
        als.set_debug_switches(sw)
        xs, ts = [], []
        for _ in range(a.reps + 1):
            eng.update_x()
            xs.append(als.last_kernel_ms()[0])
            eng.XT.copy_(keep_x)
            eng.update_theta()
            ts.append(als.last_kernel_ms()[0])
            eng.thetaT.copy_(keep_t)
        out[name] = {"x_ms": round(sum(xs[1:]) / a.reps, 3), "theta_ms": round(sum(ts[1:]) / a.reps, 3)}
    als.set_debug_switches(0)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
