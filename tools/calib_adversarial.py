"""Calibration print-out for tests/test_gpu_parity.py::test_split_gram_adversarial_per_entry: per-row rho of the split
path and of the fmaf chain by row length, and global max / q99.9 / rms ratios."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cumf_als_amd import als
from oracle import pyoracle as oracle
from tests.test_gpu_parity import _adversarial_table

oracle.build()
for kind in ("mixed_sign", "wide_range", "cancel_pairs", "tiny", "near_subnormal"):
    for f in (30, 100, 200):
        rng = np.random.RandomState(1000 + f)
        n = 1600
        lens = np.array([1, 2, 31, 32, 33, 63, 64, 65, 100, 206, 400, 777, 1500] + list(rng.randint(1, 300, size=35)))
        indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        table = _adversarial_table(kind, n, f, rng)
        cols, vals = [], []
        for k in lens:
            c = np.sort(rng.choice(n, size=k, replace=False))
            cols.append(c)
            vals.append(100.0 * rng.random_sample(k))
        indices = np.concatenate(cols).astype(np.int32)
        data = np.concatenate(vals).astype(np.float32)
        tt64, b64 = oracle.gram_rhs(indptr, indices, data, table, f, 0.0, dtype=np.float64)
        ab64, abb64 = oracle.gram_rhs(indptr, indices, np.abs(data), np.abs(table), f, 0.0, dtype=np.float64)
        u = 2.0 ** -24
        floor = lens.astype(np.float64) * 2.0 ** -126
        dev = lambda a: torch.from_numpy(a).cuda()
        plan = als.Plan(indptr, f)
        rho = {}
        for mode in ("exact", "auto"):
            als.set_gram_mode(mode)
            tt, rhs = als.get_hermitian(plan, dev(indices), dev(data), dev(table), 0.0)
            torch.cuda.synchronize()
            rho[mode] = (np.abs(tt.cpu().numpy().astype(np.float64) - tt64) / (u * ab64 + floor[:, None, None])).reshape(len(lens), -1)
        als.set_gram_mode("auto")
        rows = [0, 1, 2, 3, 4, 6, 8, 9, 10, 11, 12]
        per = " ".join(f"n={lens[i]}:{rho['auto'][i].max():.1f}/{rho['exact'][i].max():.1f}" for i in rows)
        a, e = rho["auto"].ravel(), rho["exact"].ravel()
        print(f"{kind:14s} f={f:3d} max {a.max():5.1f}/{e.max():5.1f} q999 {np.quantile(a, .999):5.2f}/{np.quantile(e, .999):5.2f} "
              f"rms {np.sqrt((a * a).mean()):5.2f}/{np.sqrt((e * e).mean()):5.2f} | worst row ratio to n: "
              f"{(rho['auto'].max(1) / lens).max():.2f} | {per}", flush=True)
