#!/usr/bin/env python3
"""Quick GPU check of the wave-per-item kernels (als_wave.hip) against the CPU oracle:
split-bf16 Gram vs the fp64 Gram (next to the exact fp32-MFMA path's own error), fused LU vs
the oracle's half-iteration, chunked rows, ragged/empty rows.  Test infrastructure."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cumf_als_amd import als, datagen  # noqa: E402
from oracle import pyoracle  # noqa: E402


def factors(rows, f, seed):
    rng = np.random.RandomState(seed)
    return (0.2 * rng.random_sample((rows, f))).astype(np.float32)


def main():
    pyoracle.build()
    bad = 0
    for f in (20, 30, 64, 100, 110):
        r = datagen.synth_ratings(96, 400, 9000, 300, seed=f, row_alpha=1.2)
        d = r.numpy()
        theta = factors(r.n, f, 1)
        lam = 0.05
        tt64, b64 = pyoracle.gram_rhs(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, f, lam, dtype=np.float64)
        rg = r.to("cuda")
        th = torch.from_numpy(theta).cuda()
        res = {}
        for mode in ("exact", "auto"):
            als.set_gram_mode(mode)
            for chunk in (0, 64):
                plan = als.Plan(d["csr_indptr"], f, chunk=chunk)
                tt, rhs = als.get_hermitian(plan, rg.csr_indices, rg.csr_data, th, lam)
                torch.cuda.synchronize()
                e = np.abs(tt.cpu().numpy() - tt64).max() / np.abs(tt64).max()
                eb = np.abs(rhs.cpu().numpy() - b64).max() / np.abs(b64).max()
                res[(mode, chunk)] = (e, eb)
        print(f"f={f} gram rel err vs fp64: " + " ".join(f"{k[0]}/c{k[1]}={v[0]:.2e},{v[1]:.2e}" for k, v in res.items()), flush=True)
        if res[("auto", 0)][0] > 4 * res[("exact", 0)][0] + 1e-7:
            bad += 1
        # fused LU
        x0 = np.zeros((r.m, f), np.float32)
        x_o = pyoracle.half_iteration(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, x0.copy(), f, lam, solver="lu")
        for mode in ("exact", "auto"):
            als.set_gram_mode(mode)
            for chunk in (0, 64):
                plan = als.Plan(d["csr_indptr"], f, chunk=chunk)
                x = torch.from_numpy(x0.copy()).cuda()
                als.update_fused(plan, rg.csr_indices, rg.csr_data, th, x, lam, "lu", 6)
                torch.cuda.synchronize()
                err = np.abs(x.cpu().numpy() - x_o).max() / max(1.0, np.abs(x_o).max())
                print(f"   fused LU {mode} chunk={chunk}: rel err vs oracle {err:.2e}", flush=True)
                if not (err < 5e-4):
                    bad += 1
    # ragged / tiny / empty rows
    rows = [0, 0, 2, 2, 2, 3] + [4] * 33 + [5] * 64
    cols = [0, 1, 0, 1, 2, 1] + list(range(33)) + list(range(64))
    vals = list(np.arange(len(rows)) % 5 + 1.0)
    r = datagen.from_coo(6, 70, rows, cols, vals, [0], [0], [1.0])
    d = r.numpy()
    for f in (10, 100):
        theta = factors(r.n, f, 2)
        als.set_gram_mode("auto")
        rg = r.to("cuda")
        plan = als.Plan(d["csr_indptr"], f)
        x = torch.zeros((r.m, f), device="cuda")
        als.update_fused(plan, rg.csr_indices, rg.csr_data, torch.from_numpy(theta).cuda(), x, 0.05, "lu", 6)
        xo = pyoracle.half_iteration(d["csr_indptr"], d["csr_indices"], d["csr_data"], theta, np.zeros((r.m, f), np.float32), f, 0.05, solver="lu")
        xh = x.cpu().numpy()
        fin = np.isfinite(xo)
        print(f"ragged f={f}: nan rows hip={np.isnan(xh).any(1).tolist()} oracle={np.isnan(xo).any(1).tolist()} "
              f"err={np.abs(xh[fin] - xo[fin]).max():.2e}", flush=True)
        if not np.array_equal(np.isnan(xh), np.isnan(xo)) or np.abs(xh[fin] - xo[fin]).max() > 5e-4 * max(1, np.abs(xo[fin]).max()):
            bad += 1
    print("WAVE_CHECK", "FAIL" if bad else "OK", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
