#!/usr/bin/env python3
"""What the host side of a distributed half-iteration costs at RCCL world 1 (the one GPU of the box): the Netflix shape
through DistALS.from_device_ratings with the pipeline forced (four pieces per side, every piece followed by its RCCL
all-gather), and the hugewiki 1/8 slab through the `reduce` scheme (THETA_BATCH = 3: reduce-scatter, solve, all-gather per
batch) -- once with torch.distributed collectives driven from Python, once with the native half-iterations
(cumf_dist_*, als_dist.cpp).  half_ms = compute-stream time of the half-iteration, kernel_ms = the libALS launches in it,
non_kernel_ms = the rest.
  python tools/dist_overhead.py [--scale 1.0] [--steps 5]"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cumf_als_amd import als, datagen  # noqa: E402
from cumf_als_amd import dist as cdist  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--solver", default="lu")
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ["CUMF_ALS_PIPE_FORCE"] = "1"
    os.environ["CUMF_ALS_PIPE_CHUNKS"] = "4"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    shp = datagen.SHAPES["netflix"]
    s, f, lam = a.scale, 100, shp["lam"]
    m, n = int(shp["m"] * s), int(shp["n"] * s)
    r = datagen.synth_ratings(m, n, int(shp["nnz"] * s * s), 512, seed=0, device=dev)
    theta0 = (0.2 * np.random.RandomState(0).random_sample((n, f))).astype(np.float32)
    for native in (False, True, False, True):
        cdist.set_native(native)
        eng = cdist.DistALS.from_device_ratings(r, f, lam, cdist.HipOps(dev), solver=a.solver, cg_iters=6)
        eng.init_factors(theta0)
        eng.iterate(2)
        d = bench.rank_diagnostics(eng, als, dev, 1, "nccl", steps=a.steps)["max_over_ranks"]
        key = f"netflix_gather_{'native' if native else 'torch'}"
        out.setdefault(key, []).append({k: d[k] for k in ("x_half_ms", "x_kernel_ms", "x_non_kernel_ms", "theta_half_ms",
                                                          "theta_kernel_ms", "theta_non_kernel_ms")})
        eng.close()
    del r
    torch.cuda.empty_cache()
    ns = argparse.Namespace(scale=a.scale, seed=0, theta_batch=3, reference_solvers=False)
    hshp = datagen.SHAPES["hugewiki"]
    nh = int(hshp["n"] * s)
    theta0 = (0.2 * np.random.RandomState(0).random_sample((nh, f))).astype(np.float32)
    cdist.set_native(False)
    eng, rs, mh, nnzh = bench.make_slab_engine(ns, hshp, 1, 0, dev, f, hshp["lam"], theta0, "cg", 6, 3)
    for native in (False, True, False, True):
        if eng is None:
            cdist.set_native(native)
            eng = bench.slab_engine(ns, rs, mh, nh, 1, dev, f, hshp["lam"], theta0, "cg", 6, 3)
        eng.iterate(2)
        d = bench.rank_diagnostics(eng, als, dev, 1, "nccl", steps=a.steps)["max_over_ranks"]
        key = f"hugewiki_reduce_{'native' if native else 'torch'}"
        out.setdefault(key, []).append({k: d[k] for k in ("x_half_ms", "x_kernel_ms", "x_non_kernel_ms", "theta_half_ms",
                                                          "theta_kernel_ms", "theta_non_kernel_ms")})
        eng.close()
        eng = None
    cdist.set_native(None)
    print(json.dumps(out, indent=1))
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
