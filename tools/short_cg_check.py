#!/usr/bin/env python3
"""Round 6: the Gram-free CG of short rows (als_short.hip) against the Gram route (CUMF_ALS_SHORT_CG=0) and the fp64 oracle on
a small matrix with empty, short and long rows:  python tools/short_cg_check.py"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from cumf_als_amd import als, datagen
    r = datagen.synth_ratings(400, 150, 9000, 600, seed=7, row_alpha=1.1, ensure_nonempty=False)
    d = r.numpy()
    rg = r.to("cuda")
    f, lam = 100, 0.05
    rng = np.random.RandomState(5)
    theta = (0.2 * rng.random_sample((r.n, f))).astype(np.float32)
    x0 = (0.05 * rng.random_sample((r.m, f))).astype(np.float32)
    plan = als.Plan(d["csr_indptr"], f)
    x = torch.from_numpy(x0.copy()).cuda()
    bins = als.update_fused_sse(plan, rg.csr_indices, rg.csr_data, torch.from_numpy(theta).cuda(), x, lam, "cg", 6)
    torch.cuda.synchronize()
    np.save(sys.argv[2], x.cpu().numpy())
    print("kernel", als.last_kernel_name()[:70], "sse", float(bins.sum()))
    sys.exit(0)

from cumf_als_amd import datagen
from oracle import pyoracle
pyoracle.build()
out = {}
for mode in ("0", "1"):
    env = dict(os.environ, CUMF_ALS_SHORT_CG=mode)
    p = f"/tmp/short_cg_{mode}.npy"
    print(subprocess.run([sys.executable, __file__, "--child", p], env=env, capture_output=True, text=True).stdout.strip())
    out[mode] = np.load(p)
r = datagen.synth_ratings(400, 150, 9000, 600, seed=7, row_alpha=1.1, ensure_nonempty=False)
d = r.numpy()
f, lam = 100, 0.05
rng = np.random.RandomState(5)
theta = (0.2 * rng.random_sample((r.n, f))).astype(np.float32)
x0 = (0.05 * rng.random_sample((r.m, f))).astype(np.float32)
lens = np.diff(d["csr_indptr"])
x64 = pyoracle.half_iteration(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, x0.copy(), f, lam, solver="cg", dtype=np.float64)
x32 = pyoracle.half_iteration(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, x0.copy(), f, lam, solver="cg", dtype=np.float32)
print("rows", len(lens), "empty", int((lens == 0).sum()), "short (<= 64)", int((lens <= 64).sum()), "longest", int(lens.max()))
for name, x in (("gram route", out["0"]), ("short kernel", out["1"]), ("oracle fp32", x32)):
    nan_eq = np.array_equal(np.isnan(x), np.isnan(x64))
    fin = ~np.isnan(x64).any(1)
    err = np.abs(x[fin] - x64[fin]).max(1) / np.maximum(np.abs(x64[fin]).max(1), 1e-30)
    print(f"{name:14s} NaN pattern equal {nan_eq}; vs fp64 oracle: max rel {err.max():.3e} median {np.median(err):.3e}; "
          f"short rows max {err[lens[fin] <= 64].max():.3e}")
bad = np.nonzero(np.isnan(out["1"]).any(1) != np.isnan(x64).any(1))[0]
print("rows with differing NaN pattern:", bad[:20].tolist(), "lengths", lens[bad[:20]].tolist())
