#!/usr/bin/env python3
"""Fused LU half-iteration of the large systems (f >= 144) against the oracle on a small matrix, for a list of f:
   python tools/check_large_lu.py 144 160 176 192 200 206"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cumf_als_amd import als, datagen
from oracle import pyoracle

pyoracle.build()
r = datagen.synth_ratings(200, 150, 9000, 600, seed=7, row_alpha=1.1)
d = r.numpy()
rg = r.to("cuda")
for f in [int(v) for v in sys.argv[1:]]:
    rng = np.random.RandomState(5)
    theta = (0.2 * rng.random_sample((r.n, f))).astype(np.float32)
    x0 = np.zeros((r.m, f), np.float32)
    lam = 0.05
    x_o = pyoracle.half_iteration(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, x0.copy(), f, lam, solver="lu",
                                  dtype=np.float64)
    for chunk in (0, 64):
        plan = als.Plan(d["csr_indptr"], f, chunk=chunk)
        x = torch.from_numpy(x0.copy()).cuda()
        als.update_fused(plan, rg.csr_indices, rg.csr_data, torch.from_numpy(theta).cuda(), x, lam, "lu", 6)
        torch.cuda.synchronize()
        xh = x.cpu().numpy()
        err = np.abs(xh - x_o).max(1) / np.abs(x_o).max(1)
        bad = np.nonzero(err > 1e-3)[0]
        print(f"f={f} chunk={chunk} WG_LU={os.environ.get('CUMF_ALS_WG_LU', '')}: max rel err {err.max():.3e} median {np.median(err):.3e} "
              f"bad rows {len(bad)} of {len(err)}; kernel {als.last_kernel_name()[:60]}", flush=True)
        if len(bad):
            u = bad[0]
            e = np.abs(xh[u] - x_o[u]) / np.abs(x_o[u]).max()
            print("   first bad row", u, "rowlen", int(d["csr_indptr"][u + 1] - d["csr_indptr"][u]), "worst elements", np.argsort(-e)[:12].tolist(),
                  "err by block", [float(e[16 * b:16 * b + 16].max()) for b in range((f + 15) // 16)])
