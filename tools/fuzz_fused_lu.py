"""Randomised parity sweep of the fused LU (accumulator LU for f >= 96, thread-grid LU below) against the
CPU oracle's half-iteration: many f, row-length distributions and chunk sizes.  Prints the worst
relative error per f; exits non-zero above 5e-4 (tests/test_gpu_parity.py::test_fused_half_iteration's bound)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cumf_als_amd import als, datagen
from oracle import pyoracle

worst = 0.0
for f in (2, 10, 16, 30, 48, 64, 80, 94, 96, 98, 100, 102, 110, 112, 126, 128, 130, 144, 160, 176, 190, 200, 206):
    errs = []
    for seed in range(3):
        rng = np.random.RandomState(100 * f + seed)
        m, n = int(rng.randint(30, 120)), int(rng.randint(40, 300))
        nnz = int(rng.randint(m + n, min(m * n // 2, 20000)))
        r = datagen.synth_ratings(m, n, nnz, 64, seed=seed + f, row_alpha=float(rng.choice([0.0, 0.8, 1.3])), device="cpu")
        d = r.numpy()
        theta = (rng.random_sample((n, f)) * 0.4 - 0.1).astype(np.float32)
        x0 = np.zeros((m, f), np.float32)
        lam = float(rng.choice([0.01, 0.05, 0.5]))
        x_o = pyoracle.half_iteration(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, x0.copy(), f, lam, solver="lu")
        rg = r.to("cuda")
        for chunk in (0, 64, 300):
            plan = als.Plan(d["csr_indptr"], f, chunk=chunk)
            x = torch.zeros((m, f), device="cuda")
            als.update_fused(plan, rg.csr_indices, rg.csr_data, torch.from_numpy(theta).cuda(), x, lam, "lu", 6)
            torch.cuda.synchronize()
            xh = x.cpu().numpy()
            ok = np.isfinite(x_o).all(axis=1)  # rows without ratings are NaN on both sides
            assert (np.isfinite(xh).all(axis=1) == ok).all(), (f, seed, chunk)
            errs.append(np.abs(xh[ok] - x_o[ok]).max() / max(1.0, np.abs(x_o[ok]).max()))
    print(f"f={f:3d}  worst rel err {max(errs):.2e}")
    worst = max(worst, max(errs))
print("worst overall", worst)
sys.exit(0 if worst <= 5e-4 else 1)
