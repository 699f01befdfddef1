"""Debug: X-side CG(6) at the full Netflix shape, HIP vs fp32 oracle per row, by f and row length."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cumf_als_amd import als, datagen
from oracle import pyoracle
from tests.test_gpu_fullsize import _oracle_rows, _sample_rows

LAM = 0.048
shp = datagen.SHAPES["netflix"]
r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device="cuda")
pyoracle.build()
for f in [int(v) for v in (sys.argv[1:] or ["100", "96", "110", "80", "64"])]:
    theta0 = (0.2 * np.random.RandomState(0).random_sample((r.n, f))).astype(np.float32)
    eng = als.ALSEngine(r, f, LAM, solver="cg", cg_iters=6)
    eng.init_factors(theta0)
    eng.iterate(1)
    rng = np.random.RandomState(7)
    rows = _sample_rows(r.csr_indptr.cpu().numpy(), 1000, rng)
    ip = r.csr_indptr.cpu().numpy().astype(np.int64)
    lens = ip[rows + 1] - ip[rows]
    chunk = eng.x_plans[0].chunk
    for iters in (6, 3):
        eng.cg_iters = iters
        warm = eng.XT.clone()
        eng.update_x()
        torch.cuda.synchronize()
        x32, _ = _oracle_rows(pyoracle, r.csr_indptr, r.csr_indices, r.csr_data, eng.thetaT, warm, rows, f, LAM, "cg", cg_iters=iters)
        xh = eng.XT[torch.from_numpy(rows).cuda()].cpu().numpy()
        el = np.abs(xh - x32).max(1) / np.maximum(1.0, np.abs(x32).max(1))
        bad = el > 2e-4
        print(f"f={f} iters={iters} chunk={chunk}: bad {int(bad.sum())}/{len(rows)} max {el.max():.2e} | chunked rows: {int((lens > chunk).sum())} bad among chunked {int((bad & (lens > chunk)).sum())} "
              f"| bad lens: {sorted(lens[bad].tolist())[:12]} ... | worst rows lens {lens[np.argsort(el)[-5:]].tolist()} errs {np.sort(el)[-5:]}", flush=True)
        eng.XT.copy_(warm)
