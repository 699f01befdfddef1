#!/usr/bin/env python3
"""Run-to-run spread of the Theta-side launch: the same half-iteration timed repeatedly inside one process (same
allocations), and after re-creating the engine (new allocations).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cumf_als_amd import als, datagen

shp = datagen.SHAPES["netflix"]
r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device="cuda")
out = {"engines": []}
for e in range(3):
    eng = als.ALSEngine(r, 100, shp["lam"], solver="lu")
    eng.init_factors()
    eng.iterate(1)
    torch.cuda.synchronize()
    als.set_kernel_timing(True)
    xs, ts = [], []
    for rep in range(12):
        eng.update_x(); xs.append(round(sum(als.last_kernel_ms()), 3))
        eng.update_theta(); ts.append(round(sum(als.last_kernel_ms()), 3))
    out["engines"].append({"x": xs, "theta": ts, "thetaT_ptr": hex(eng.thetaT.data_ptr()), "XT_ptr": hex(eng.XT.data_ptr())})
    eng.close(); del eng
    torch.cuda.empty_cache()
print(json.dumps(out))
