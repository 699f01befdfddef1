"""Is the X-side item kernel bound by the gather traffic or by on-CU work?  Times update_x on
the Netflix-shape matrix with (a) real column indices, (b) all indices -> a 17770-row window
(L2-resident table), (c) all indices = 0 (one row, L1-resident)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cumf_als_amd import als, datagen
shp = datagen.SHAPES["netflix"]
r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device="cuda")
def timed(eng, what):
    als.set_kernel_timing(True)
    ts = []
    for _ in range(3):
        getattr(eng, what)(); ts.append(als.last_kernel_ms())
    als.set_kernel_timing(False)
    return min(t[0] for t in ts), min(t[1] for t in ts)
for solver in ("cg",):
    eng = als.ALSEngine(r, 100, 0.048, solver=solver)
    eng.init_factors()
    print("real indices       x:", timed(eng, "update_x"), " theta:", timed(eng, "update_theta"))
    orig = r.csr_indices.clone()
    r.csr_indices.copy_(orig % 17770)
    print("17770-row window   x:", timed(eng, "update_x"))
    r.csr_indices.zero_()
    print("single row         x:", timed(eng, "update_x"))
    r.csr_indices.copy_(orig)
    oc = r.csc_indices.clone()
    r.csc_indices.zero_()
    print("single row     theta:", timed(eng, "update_theta"))
    r.csc_indices.copy_(oc)
