"""Multi-GPU ALS from pre-split per-GPU slab files -- the replacement of `hugewiki/hugewiki.cu`'s
`main()` (hugewiki.cu:2232-2870): X row-sharded and device-resident, Theta replicated, the Theta
update a partial Gram per X slab + reduce-scatter + solve + all-gather.

One process per GPU:

    python -m cumf_als_amd.convert split data_dir/ split_dir/ --gpus 8 --m M --n N --nnz NNZ --nnz-test T
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        --master-port 29500 -m cumf_als_amd.hugewiki split_dir/ --n N --f 100 --lambda 0.048 --iters 10

Rank g loads `R_train_csr.*.bin<g>` (row pointer rebased to 0) and `R_test_coo.*.bin<g>` (slab-local
row ids) written by the splitter; no rank ever holds the whole matrix.  Initialisation as the
reference (hugewiki.cu:2381-2393 = main.cpp:72-78): srand(0), thetaT[k] = 0.2 * rand()/RAND_MAX on
every rank (identical replicas), X = 0.  Prints the reference's RMSE lines from rank 0 and returns
the per-iteration (train, test) RMSE.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import convert, lib as _libmod
from .dist import DistALS, HipOps


def run(split_dir: str, n: int, f: int, lam: float, iters: int, solver: str = "cg", cg_iters: int = 6,
        theta_batch: int = 1, ops=None, quiet: bool = False, solver_x=None, solver_theta=None, cg_iters_x=None,
        cg_iters_theta=None):
    """Body of one rank.  `ops` defaults to the HIP kernels on this rank's GPU.  solver_x / solver_theta /
    cg_iters_x / cg_iters_theta override `solver` / `cg_iters` per side: the reference's own hugewiki run is
    solver_x="cg", cg_iters_x=100 (hugewiki.cu:2569), solver_theta="lu" (hugewiki.cu:2732) = `--reference-solvers`."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    bounds = np.array(open(os.path.join(split_dir, "slabs.txt")).read().split(), dtype=np.int64)
    if len(bounds) != world + 1:
        raise ValueError(f"{split_dir} was split for {len(bounds) - 1} GPUs, launched with {world}")
    rows = int(bounds[rank + 1] - bounds[rank])
    slab = convert.read_slab(split_dir, rank, rows, n)
    if ops is None:
        ops = HipOps(torch.device("cuda", torch.cuda.current_device()))
    to_dev = ops.to_device
    rowptr = torch.from_numpy(slab["csr_indptr"].astype(np.int64))
    colidx, val = to_dev(slab["csr_indices"]), to_dev(slab["csr_data"])
    eng = DistALS.from_local_slab(int(bounds[-1]), n, bounds, rowptr.to(colidx.device), colidx, val, f, lam, ops,
                                  solver=solver, cg_iters=cg_iters, theta_batch=theta_batch, solver_x=solver_x,
                                  solver_theta=solver_theta, cg_iters_x=cg_iters_x, cg_iters_theta=cg_iters_theta)
    thetaT = np.empty((n, f), np.float32)
    _libmod.load().cumf_rand_init(thetaT.ctypes.data_as(C.c_void_p), n * f, 0.2, 0)
    eng.thetaT.copy_(torch.from_numpy(thetaT))
    eng.XT.zero_()
    # train RMSE pairs the CSR entries with their (slab-local) row, als.cu:196-198
    train_row = to_dev(np.repeat(np.arange(rows, dtype=np.int32), np.diff(slab["csr_indptr"])))
    test = [to_dev(slab[k]) for k in ("test_data", "test_row", "test_col")]
    counts = torch.tensor([float(val.numel()), float(test[0].numel())], dtype=torch.float64)
    if world > 1:
        c = counts.to(colidx.device) if dist.get_backend() == "nccl" else counts
        dist.all_reduce(c)
        counts = c.cpu()
    nnz, nnz_test = float(counts[0]), float(counts[1])
    log = []
    t0 = time.time()
    for it in range(iters):
        eng.update_x()
        # the train SSE out of the Theta update itself (reduced systems: sum r^2 - (2 t.b - t^T G t));
        # the RMSE kernel over the slab's ratings (hugewiki.cu:2750-2862) only when the ops cannot deliver it
        sse = eng.update_theta(train_sse=True)
        if sse is None:
            sse = eng.slab_sse(val, train_row, colidx)
        tr = (max(sse, 0.0) / nnz) ** 0.5
        te = (eng.slab_sse(test[0], test[1], test[2]) / nnz_test) ** 0.5 if nnz_test else float("nan")
        log.append((tr, te))
        if rank == 0 and not quiet:
            print("--------- Train RMSE in iter %d: %f" % (it, tr))
            print("--------- Test RMSE in iter %d: %f" % (it, te), flush=True)
    if rank == 0 and not quiet:
        print("\ndoALS takes seconds: %.3f for F = %d on %d GPU(s)" % (time.time() - t0, f, world))
    return eng, log


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="multi-GPU ALS from per-GPU slab files (hugewiki.cu main)")
    ap.add_argument("split_dir")
    ap.add_argument("--n", type=int, required=True, help="number of columns (Theta rows)")
    ap.add_argument("--f", type=int, default=100)
    ap.add_argument("--lambda", dest="lam", type=float, default=0.048)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--solver", choices=["cg", "lu"], default="cg")
    ap.add_argument("--cg-iters", type=int, default=6)
    ap.add_argument("--theta-batch", type=int, default=1)
    ap.add_argument("--solver-x", choices=["cg", "lu"], default=None, help="solver of the X update (default: --solver)")
    ap.add_argument("--solver-theta", choices=["cg", "lu"], default=None,
                    help="solver of the Theta update on the reduced Gram (default: --solver)")
    ap.add_argument("--cg-iters-x", type=int, default=None)
    ap.add_argument("--cg-iters-theta", type=int, default=None)
    ap.add_argument("--reference-solvers", action="store_true",
                    help="what hugewiki.cu runs: X by CG with 100 iterations (hugewiki.cu:2569), Theta by the batched LU "
                         "(hugewiki.cu:2732); the same as --solver-x cg --cg-iters-x 100 --solver-theta lu")
    a = ap.parse_args(argv)
    if a.reference_solvers:
        a.solver_x, a.cg_iters_x, a.solver_theta = "cg", 100, "lu"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("cumf_als_amd.hugewiki needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    try:
        eng, _ = run(a.split_dir, a.n, a.f, a.lam, a.iters, a.solver, a.cg_iters, a.theta_batch, solver_x=a.solver_x,
                     solver_theta=a.solver_theta, cg_iters_x=a.cg_iters_x, cg_iters_theta=a.cg_iters_theta)
        eng.close()  # the library's pooled scratch of this device
    finally:
        if world > 1:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
