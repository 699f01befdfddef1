"""Debug: Theta-side CG at the full Netflix shape, HIP vs oracle per row (by row length, gram mode, cg_iters)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cumf_als_amd import als, datagen
from oracle import pyoracle
from tests.test_gpu_fullsize import _oracle_rows, _sample_rows

F, LAM = 100, 0.048
shp = datagen.SHAPES["netflix"]
r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device="cuda")
theta0 = (0.2 * np.random.RandomState(0).random_sample((r.n, F))).astype(np.float32)
pyoracle.build()
for mode in ("auto", "exact"):
    als.set_gram_mode(mode)
    for iters in (1, 2, 6):
        eng = als.ALSEngine(r, F, LAM, solver="cg", cg_iters=6)
        eng.init_factors(theta0)
        eng.iterate(1)
        eng.update_x()
        eng.cg_iters = iters
        rng = np.random.RandomState(7)
        cols = _sample_rows(r.csc_indptr.cpu().numpy(), 2000, rng)
        warm = eng.thetaT.clone()
        eng.update_theta()
        torch.cuda.synchronize()
        ip = r.csc_indptr.cpu().numpy().astype(np.int64)
        lens = ip[cols + 1] - ip[cols]
        import ctypes
        # oracle with the same cg_iters
        sub_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        sel = torch.cat([torch.arange(int(ip[u]), int(ip[u + 1]), device="cuda") for u in cols])
        si, sv = r.csc_indices[sel].cpu().numpy(), r.csc_data[sel].cpu().numpy()
        x = np.ascontiguousarray(warm[torch.from_numpy(cols).cuda()].cpu().numpy())
        pyoracle.half_iteration(sub_ptr, si, sv, eng.XT.cpu().numpy(), x, F, LAM, solver="cg", cg_iters=iters)
        xh = eng.thetaT[torch.from_numpy(cols).cuda()].cpu().numpy()
        el = np.abs(xh - x).max(1) / np.maximum(1.0, np.abs(x).max(1))
        bins = [(1, 8), (8, 32), (32, 64), (64, 100), (100, 200), (200, 500), (500, 5000)]
        s = " ".join(f"[{a},{b}): n={int(((lens >= a) & (lens < b)).sum())} max={el[(lens >= a) & (lens < b)].max() if ((lens >= a) & (lens < b)).any() else 0:.1e}" for a, b in bins)
        print(f"mode={mode} cg_iters={iters}: within2e-4={float((el <= 2e-4).mean()):.3f} max={el.max():.2e} | {s}", flush=True)
als.set_gram_mode("auto")
