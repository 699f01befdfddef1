#!/usr/bin/env python3
"""Kernel times of the two half-iterations of one configuration with ONE build of the library (CUMF_ALS_LIB), from
real factors (one full iteration first, factors restored between launches).  Prints one JSON line.
  CUMF_ALS_LIB=variants/libALS_x.so python tools/time_halves.py [--f 100] [--solver lu] [--reps 5] [--check ref.pt]
--save / --check: store the factors of one full iteration / compare with stored ones (max relative difference)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cumf_als_amd import als, datagen  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="netflix")
    ap.add_argument("--f", type=int, default=100)
    ap.add_argument("--solver", default="lu")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--save", default="")
    ap.add_argument("--check", default="")
    a = ap.parse_args()
    shp = datagen.SHAPES[a.shape]
    r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device="cuda")
    eng = als.ALSEngine(r, a.f, shp["lam"], solver=a.solver)
    eng.init_factors()
    eng.iterate(1)
    torch.cuda.synchronize()
    keep_x, keep_t = eng.XT.clone(), eng.thetaT.clone()
    out = {"library": os.path.basename(os.environ.get("CUMF_ALS_LIB", "libALS.so")), "f": a.f, "solver": a.solver,
           "split_launch": os.environ.get("CUMF_ALS_SPLIT_LAUNCH", "")}
    if a.save:
        torch.save({"x": keep_x.cpu(), "t": keep_t.cpu()}, a.save)
    if a.check and os.path.exists(a.check):
        ref = torch.load(a.check)
        for k, v in (("x", keep_x), ("t", keep_t)):
            d = (v.cpu().double() - ref[k].double()).abs().max().item() / ref[k].double().abs().max().item()
            out[f"max_rel_diff_{k}"] = d
    als.set_kernel_timing(True)
    xs, ts = [], []
    for _ in range(a.reps + 1):
        eng.update_x()
        xs.append(sum(als.last_kernel_ms()))
        eng.XT.copy_(keep_x)
        eng.update_theta()
        ts.append(sum(als.last_kernel_ms()))
        eng.thetaT.copy_(keep_t)
    xs, ts = sorted(xs[1:]), sorted(ts[1:])
    out.update(x_ms=round(xs[len(xs) // 2], 3), theta_ms=round(ts[len(ts) // 2], 3), x_min=round(xs[0], 3), theta_min=round(ts[0], 3))
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
