#!/usr/bin/env python3
"""The Gram pass alone: the half-iteration launches of bench.py's workload with the in-kernel solve switched off.

Runs against the PROFILING build of the library (cumf_als_amd/csrc/libALS_ablate.so, `make ablate`,
-DCUMF_ABLATE=1): only that build has the switch -- the product library compiles none of it.  bench.py starts
this script in a process of its own (CUMF_ALS_LIB=.../libALS_ablate.so) for its `roofline.gram_pass_alone` leg,
and tools/collect_profiles.sh runs it under rocprofv3 for profiles/<round>/gram_only/.

  CUMF_ALS_LIB=cumf_als_amd/csrc/libALS_ablate.so python tools/gram_pass_alone.py [--f 100] [--solver lu]

Prints one JSON line.  Both passes gather REAL factors (one full iteration with the switches off comes first, and
the factors are restored between launches): the matrix pipe's clock depends on the data (an all-NaN table runs
15 % faster).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cumf_als_amd import als, datagen  # noqa: E402

HBM_PEAK_GBS = 8000.0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="netflix")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--f", type=int, default=100)
    ap.add_argument("--solver", default="lu", choices=["lu", "cg"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--switches", type=int, default=0,
                    help="further ablation switches OR-ed to 1 (no solve): 8 = every gather hits row 0, 16 = no gather "
                         "DMA, 32 / 64 / 128 = gathers confined to the first 64 / 4096 / 65536 table rows")
    a = ap.parse_args()
    shp = datagen.SHAPES[a.shape]
    s, f, lam = a.scale, a.f, shp["lam"]
    m, n = max(2, int(shp["m"] * s)), max(2, int(shp["n"] * s))
    nnz, nnz_test = max(int(shp["nnz"] * s * s), m + n), max(int(shp["nnz_test"] * s * s), 512)
    r = datagen.synth_ratings(m, n, nnz, nnz_test, seed=a.seed, device="cuda")
    eng = als.ALSEngine(r, f, lam, solver=a.solver)
    g = torch.Generator(device="cpu")
    g.manual_seed(a.seed)
    eng.init_factors((0.2 * torch.rand((n, f), generator=g, dtype=torch.float32)).numpy())
    als.set_debug_switches(0)
    # one full iteration for real factors -- with the OTHER solver, so that its launches do not enter the per-kernel
    # averages of the kernel under study when this script runs under rocprofv3
    eng.solver = "cg" if a.solver == "lu" else "lu"
    eng.update_x()
    eng.update_theta()
    eng.solver = a.solver
    keep_x, keep_t = eng.XT.clone(), eng.thetaT.clone()
    g_ms = []
    als.set_debug_switches(1 | a.switches)  # no solve (+ the requested gather ablations)
    als.set_kernel_timing(True)
    names = {}
    for _ in range(a.reps + 1):
        eng.update_x()
        gx = als.last_kernel_ms()[0]
        names["x"] = als.last_kernel_name()
        eng.XT.copy_(keep_x)
        eng.update_theta()
        g_ms.append((gx, als.last_kernel_ms()[0]))
        names["theta"] = als.last_kernel_name()
        eng.thetaT.copy_(keep_t)
    als.set_kernel_timing(False)
    als.set_debug_switches(0)
    gx = sum(v[0] for v in g_ms[1:]) / len(g_ms[1:])
    gt = sum(v[1] for v in g_ms[1:]) / len(g_ms[1:])
    gb_x = 4.0 * f * nnz + 8.0 * nnz + 4.0 * (m + 1)  # Gram + RHS inputs only (no factor write)
    gb_t = 4.0 * f * nnz + 8.0 * nnz + 4.0 * (n + 1)
    print(json.dumps({
        "library": os.path.basename(os.environ.get("CUMF_ALS_LIB", "libALS.so")), "switches": 1 | a.switches,
        "kernel_x": names["x"],
        "kernel_theta": names["theta"], "x_side_ms": gx, "theta_side_ms": gt,
        "x_side_alg_bytes": gb_x, "theta_side_alg_bytes": gb_t,
        "x_side_frac_of_hbm_roof": gb_x / (gx * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "theta_side_frac_of_hbm_roof": gb_t / (gt * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "note": "the half-iteration kernels of the profiling build (libALS_ablate.so, -DCUMF_ABLATE=1) with the solve "
                "switched off; on the Netflix shape the Theta side gathers a 7 MB table that lives in L2, so its "
                "fraction is bytes-equivalent, not HBM traffic"}), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
