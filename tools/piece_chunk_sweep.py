#!/usr/bin/env python3
"""What one rank's pipeline pieces of the `gather` scheme cost at N ranks, measured on ONE GPU: the Netflix shape cut as
DistALS cuts it (cost-balanced slabs, `pieces` cost-balanced pieces per slab), the fused kernel of every piece of one rank
timed for several chunk lengths (the plan's item size: a piece of a 1/8 slab has a few thousand items for 2 048 wave slots,
so the chunk length decides how evenly the slots fill).
  python tools/piece_chunk_sweep.py [--world 8] [--pieces 4] [--rank 0] [--chunks 0,512,1024,2048,4096]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cumf_als_amd import als, datagen  # noqa: E402
from cumf_als_amd import dist as cdist  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--pieces", type=int, default=4)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--f", type=int, default=100)
    ap.add_argument("--solver", default="lu")
    ap.add_argument("--chunks", default="0,512,1024,2048,4096")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    shp = datagen.SHAPES["netflix"]
    dev = torch.device("cuda", 0)
    r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device=dev)
    f, lam = a.f, shp["lam"]
    rng = np.random.RandomState(0)
    theta = torch.from_numpy((0.2 * rng.random_sample((r.n, f))).astype(np.float32)).to(dev)
    x = torch.from_numpy((0.2 * rng.random_sample((r.m, f))).astype(np.float32)).to(dev)
    out = {"world": a.world, "pieces": a.pieces, "rank": a.rank, "f": f, "solver": a.solver, "sides": {}}
    als.set_kernel_timing(True)
    for side, indptr, indices, data, table, upd in (("x", r.csr_indptr, r.csr_indices, r.csr_data, theta, x),
                                                    ("theta", r.csc_indptr, r.csc_indices, r.csc_data, x, theta)):
        rp = indptr.cpu().numpy().astype(np.int64)
        cost = cdist.solve_row_cost(f, a.solver)
        sb = cdist.balanced_slabs(rp, a.world, cost)
        pb = cdist.pipeline_bounds(rp, sb, a.pieces, cost)[a.rank]
        s0, s1 = int(sb[a.rank]), int(sb[a.rank + 1])
        rl = rp[s0:s1 + 1] - rp[s0]
        ci, va = indices[rp[s0]:rp[s1]], data[rp[s0]:rp[s1]]
        res = {}
        for chunk in [int(c) for c in a.chunks.split(",")]:
            plans = [als.Plan(rl, f, int(pb[c] - s0), int(pb[c + 1] - s0), chunk) for c in range(a.pieces)]
            u = upd[s0:s1].clone()
            ms = []
            for rep in range(a.reps + 1):
                als.kernel_ms_since_reset()
                per = []
                for p in plans:
                    als.update_fused(p, ci, va, table, u, lam, a.solver, 6)
                    torch.cuda.synchronize()
                    k = als.kernel_ms_since_reset()
                    per.append(round(k[0] + k[1], 4))
                if rep:
                    ms.append(per)
            med = [float(np.median([m[c] for m in ms])) for c in range(a.pieces)]
            res[str(chunk)] = {"chunk_used": plans[0].chunk, "items": [p.n_items for p in plans],
                               "chunked_rows": [p.n_multi_rows for p in plans], "piece_ms": [round(v, 4) for v in med],
                               "sum_ms": round(sum(med), 4)}
            for p in plans:
                p.close()
        out["sides"][side] = res
    print(json.dumps(out, indent=1))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
