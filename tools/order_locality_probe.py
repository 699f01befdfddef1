#!/usr/bin/env python3
"""Does the Theta-side launch care where its items' rating segments lie?  The plan dispatches items longest first, i.e. in an
order unrelated to the CSC arrays' own (column) order: every item start reads index / rating segments at a random place of
two 400 MB arrays.  Here the columns are RELABELLED by length (a permutation of the users: the same problem) so that the
dispatch order walks the arrays front to back, and the same fused kernel is timed on both layouts.
  python tools/order_locality_probe.py [--shape netflix] [--f 100] [--solver lu]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cumf_als_amd import als, datagen  # noqa: E402


def relabel_by_length(indptr, indices, data):
    lens = (indptr[1:] - indptr[:-1])
    perm = torch.argsort(lens, descending=True, stable=True)
    nl = lens[perm]
    new_ptr = torch.zeros_like(indptr)
    new_ptr[1:] = torch.cumsum(nl, 0)
    src = torch.arange(int(indptr[-1]), device=indptr.device, dtype=torch.int64)
    src += torch.repeat_interleave(indptr[:-1][perm] - new_ptr[:-1], nl)
    return new_ptr, indices[src].contiguous(), data[src].contiguous(), perm


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="netflix")
    ap.add_argument("--f", type=int, default=100)
    ap.add_argument("--solver", default="lu")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--side", default="theta")
    a = ap.parse_args()
    shp = datagen.SHAPES[a.shape]
    dev = torch.device("cuda", 0)
    r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device=dev)
    f, lam = a.f, shp["lam"]
    rng = np.random.RandomState(0)
    if a.side == "theta":
        ptr, idx, val, rows, trows = r.csc_indptr.to(torch.int64), r.csc_indices, r.csc_data, r.n, r.m
    else:
        ptr, idx, val, rows, trows = r.csr_indptr.to(torch.int64), r.csr_indices, r.csr_data, r.m, r.n
    table = torch.from_numpy((0.2 * rng.random_sample((trows, f))).astype(np.float32)).to(dev)
    upd0 = torch.from_numpy((0.2 * rng.random_sample((rows, f))).astype(np.float32)).to(dev)
    p2, i2, v2, perm = relabel_by_length(ptr, idx, val)
    als.set_kernel_timing(True)
    out = {"shape": a.shape, "f": f, "solver": a.solver, "side": a.side}
    res = {}
    plans = {"as_given": (als.Plan(ptr.cpu().numpy(), f), idx, val, upd0.clone()),
             "relabelled_by_length": (als.Plan(p2.cpu().numpy(), f), i2, v2, upd0[perm].clone())}
    for rep in range(a.reps + 1):
        for name, (plan, ci, va, u) in plans.items():
            als.kernel_ms_since_reset()
            als.update_fused(plan, ci, va, table, u, lam, a.solver, 6)
            torch.cuda.synchronize()
            k = als.kernel_ms_since_reset()
            if rep:
                res.setdefault(name, []).append(round(k[0] + k[1], 4))
    out["kernel_ms"] = res
    out["median_ms"] = {k: float(np.median(v)) for k, v in res.items()}
    # the same solution up to the relabelling (one fused update from the same start)
    a_, b_ = plans["as_given"][3], plans["relabelled_by_length"][3]
    out["same_factors"] = bool(torch.equal(a_[perm], b_))
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
