import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cumf_als_amd import als, datagen
r = datagen.synth_ratings(400, 150, 9000, 600, seed=7, row_alpha=1.1, ensure_nonempty=False)
d = r.numpy(); rg = r.to("cuda")
f, lam = 100, 0.05
rng = np.random.RandomState(5)
theta = (0.2 * rng.random_sample((r.n, f))).astype(np.float32)
x0 = (0.05 * rng.random_sample((r.m, f))).astype(np.float32)
ptr, idx, val = d["csr_indptr"], d["csr_indices"], d["csr_data"]
lens = np.diff(ptr)
def cg(u, iters):
    T = theta[idx[ptr[u]:ptr[u+1]]].astype(np.float64); rv = val[ptr[u]:ptr[u+1]].astype(np.float64)
    A = T.T @ T + lam * len(rv) * np.eye(f); b = T.T @ rv
    x = x0[u].astype(np.float64); rr = b - A @ x; p = rr.copy(); rs = rr @ rr
    for _ in range(iters):
        ap = A @ p; al = rs / (p @ ap); x = x + al * p; rr = rr - al * ap; rn = rr @ rr
        if rn < 1e-4: break
        p = rr + (rn / rs) * p; rs = rn
    return x
plan = als.Plan(ptr, f)
for iters in (0, 1, 6):
    x = torch.from_numpy(x0.copy()).cuda()
    als.update_fused(plan, rg.csr_indices, rg.csr_data, torch.from_numpy(theta).cuda(), x, lam, "cg", iters)
    torch.cuda.synchronize()
    xh = x.cpu().numpy()
    for u in [int(np.argmin(np.abs(lens - t))) for t in (1, 3, 4, 5, 8, 20, 40, 64, 100)]:
        ref = cg(u, iters)
        print(f"iters {iters} row {u} len {lens[u]}: max |x - ref| / max|ref| = {np.abs(xh[u] - ref).max() / np.abs(ref).max():.3e}  x[:3] {xh[u][:3]} ref[:3] {ref[:3]} x[64:67] {xh[u][64:67]} ref {ref[64:67]}")
