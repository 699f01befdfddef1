#!/usr/bin/env python3
"""bench.py -- ratings/sec per ALS half-iteration on synthetic Netflix-shaped ratings.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one full pass of the hot path over the dataset: update-X + update-Theta
(RHS + Gram + solve for every row of both sides), i.e. TWO half-iterations.  Inputs are
resident in HBM before the timed region.  Workload = BASELINE.json configs[1]:
Netflix shape (17770 x 480189, 99 072 112 ratings), f = 100, lambda = 0.048, LU solver.

Printed JSON (one line, rank 0):
  value       2 * nnz * K / t           ratings/s per half-iteration, whole job
  roofline    dominant kernel = the per-item Gram(+solve) kernel `als_item_kernel`;
              achieved = algorithmic bytes of a half-iteration (SURVEY.md 8d:
              4 f nnz + 8 nnz + 4 (rows+1) + 4 f rows) / its HIP-event duration, averaged
              over the X-side and Theta-side launches of the timed steps; peak = 8 TB/s HBM.
  cpu_baseline  the CPU oracle (oracle/, "port": the reference has no CPU path) timed on
              this host's cores on a row sample of the same matrix, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def alg_bytes(nnz: int, rows: int, f: int, cg: bool) -> float:
    """Algorithmic bytes of one half-iteration."""
    b = 4.0 * f * nnz + 4.0 * nnz + 4.0 * nnz + 4.0 * (rows + 1) + 4.0 * f * rows
    if cg:
        b += 4.0 * f * rows
    return b


def cpu_baseline(d, f, lam, solver, target_s=16.0):
    """Oracle half-iterations (all host cores, OpenMP over rows) on a row sample of the same
    matrix sized for ~target_s seconds of wall time: a 1/64 probe sets the rate, the sized
    sample (capped at the whole matrix) is what is reported."""
    from oracle import pyoracle

    pyoracle.build()
    rng = np.random.RandomState(0)
    sides = {
        "x": (d["csr_indptr"], d["csr_indices"], d["csr_data"], len(d["csc_indptr"]) - 1),
        "theta": (d["csc_indptr"], d["csc_indices"], d["csc_data"], len(d["csr_indptr"]) - 1),
    }
    gathers = {k: (0.2 * rng.random_sample((v[3], f))).astype(np.float32) for k, v in sides.items()}

    def run(frac):
        out, tot_nnz, tot_t = {}, 0, 0.0
        for side, (ptr, idx, val, _) in sides.items():
            rows = max(1, int(np.searchsorted(ptr, int(ptr[-1] * frac), side="left")))
            rows = min(rows, len(ptr) - 1)
            nn = int(ptr[rows])
            update = np.zeros((rows, f), np.float32)
            t = pyoracle.time_half_iteration(ptr[:rows + 1], idx[:nn], val[:nn], gathers[side], update, f, lam,
                                             solver=solver)
            out[side] = (rows, nn, t)
            tot_nnz += nn
            tot_t += t
        return out, tot_nnz, tot_t

    _, pn, pt = run(1.0 / 64)
    frac = min(1.0, max(1.0 / 64, (target_s / max(pt, 1e-6)) / 64))
    out, tot_nnz, tot_t = run(frac)
    return {
        "value": tot_nnz / tot_t,
        "unit": "ratings/s",
        "cores": pyoracle.num_threads(),
        "kind": "port",
        "sample": (f"one X and one Theta half-iteration of the CPU oracle (fp32, {solver.upper()} solver, OpenMP over "
                   f"rows, {pyoracle.num_threads()} threads) on the first {out['x'][0]} X rows ({out['x'][1]} ratings, "
                   f"{out['x'][2]:.2f} s) and the first {out['theta'][0]} Theta rows ({out['theta'][1]} ratings, "
                   f"{out['theta'][2]:.2f} s) of the same synthetic matrix ({100 * frac:.0f} % of its ratings per side)"),
    }


def measured_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (tools/collect_profiles.sh -> profiles/): (2 * FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950.  None when no profile is committed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", default="netflix")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the shape (debug only; the result is then invalid)")
    ap.add_argument("--f", type=int, default=100)
    ap.add_argument("--solver", default="lu", choices=["lu", "cg"])
    ap.add_argument("--cg-iters", type=int, default=6)
    ap.add_argument("--scheme", default="gather", choices=["gather", "reduce"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()

    from cumf_als_amd import als, datagen

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    shp = datagen.SHAPES[a.shape]
    s = a.scale
    lam, f = shp["lam"], a.f
    slab_mode = a.shape == "hugewiki"
    t0 = time.time()
    if slab_mode:
        # hugewiki scale (BASELINE.json configs[3]): WEAK scaling, every rank generates and keeps
        # one row slab of 1/8 of the hugewiki matrix (6.26 M x 39 780, 388 M ratings); the
        # "reduce" scheme never materialises the whole matrix anywhere.
        m_slab, n = max(2, int(shp["m"] * s) // 8), max(2, int(shp["n"] * s))
        nnz_slab = max(int(shp["nnz"] * s * s) // 8, m_slab + n)
        r = datagen.synth_ratings(m_slab, n, nnz_slab, 4096, seed=a.seed + 1000 * (rank + 1), device=dev,
                                  col_seed=a.seed)
        m, nnz = m_slab * world, nnz_slab * world
    else:
        m, n = max(2, int(shp["m"] * s)), max(2, int(shp["n"] * s))
        nnz, nnz_test = max(int(shp["nnz"] * s * s), m + n), max(int(shp["nnz_test"] * s * s), 512)
        r = datagen.synth_ratings(m, n, nnz, nnz_test, seed=a.seed, device=dev)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    g = torch.Generator(device="cpu")
    g.manual_seed(a.seed)
    theta0 = (0.2 * torch.rand((n, f), generator=g, dtype=torch.float32)).numpy()

    item_ms = []
    if slab_mode:
        from cumf_als_amd import dist as cdist

        xb = np.arange(world + 1, dtype=np.int64) * r.m
        eng = cdist.DistALS.from_local_slab(m, n, xb, r.csr_indptr, r.csr_indices, r.csr_data, f, lam,
                                            cdist.HipOps(dev), solver=a.solver, cg_iters=a.cg_iters)
        eng.init_factors(theta0)

        def step(timed):
            eng.update_x()
            if timed:
                item_ms.append(als.last_kernel_ms())
            eng.update_theta()
            if timed:
                item_ms.append(als.last_kernel_ms())

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                import torch.distributed as dist

                dist.barrier()
                torch.cuda.synchronize()
    elif world == 1:
        eng = als.ALSEngine(r, f, lam, solver=a.solver, cg_iters=a.cg_iters)
        eng.init_factors(theta0)

        def step(timed):
            eng.update_x()
            if timed:
                item_ms.append(als.last_kernel_ms())
            eng.update_theta()
            if timed:
                item_ms.append(als.last_kernel_ms())

        def barrier():
            torch.cuda.synchronize()
    else:
        import torch.distributed as dist

        from cumf_als_amd import dist as cdist

        d = r.numpy()
        mat = cdist.HostMatrix(m, n, d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indptr"],
                               d["csc_indices"], d["csc_data"])
        eng = cdist.DistALS(mat, f, lam, cdist.HipOps(dev), solver=a.solver, cg_iters=a.cg_iters, scheme=a.scheme)
        eng.init_factors(theta0)

        def step(timed):
            eng.update_x()
            eng.update_theta()

        def barrier():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(False)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        value = 2.0 * nnz * a.steps / elapsed
        out = {
            "metric": "ratings/sec per ALS half-iteration (Netflix f=100); RMSE vs reference",
            "value": value, "unit": "ratings/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True,
            "scaling": "weak" if slab_mode else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.shape}-shape synthetic ratings {m}x{n}, nnz={nnz}, f={f}, "
                                   f"lambda={lam}, solver={a.solver}"
                                   + (f"(cg_iters={a.cg_iters})" if a.solver == "cg" else "")
                                   + (f", row slab per GPU (1/8 hugewiki), reduce scheme over {world} GPU(s)" if slab_mode
                                      else ", X_BATCH=1 THETA_BATCH=1, fused Gram+solve" if world == 1
                                      else f", {a.scheme} scheme over {world} GPUs"),
                       "step": "update-X + update-Theta (two half-iterations)", "gen_seconds": round(t_gen, 2)},
        }

    if world == 1 and not slab_mode:
        # roofline leg: the same steps again with HIP events around each kernel launch
        als.set_kernel_timing(True)
        for _ in range(max(2, min(a.steps, 5))):
            step(True)
        torch.cuda.synchronize()
        als.set_kernel_timing(False)
        x_ms = [v[0] for v in item_ms[0::2]]
        t_ms = [v[0] for v in item_ms[1::2]]
        red_ms = [v[1] for v in item_ms]
        cg = a.solver == "cg"
        bx, bt = alg_bytes(nnz, m, f, cg), alg_bytes(nnz, n, f, cg)
        avg_ms = (sum(x_ms) + sum(t_ms)) / (len(x_ms) + len(t_ms))
        avg_bytes = 0.5 * (bx + bt)
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": "cumf::als_item_kernel<7, float4, LU>" if (f == 100 and not cg) else "cumf::als_item_kernel",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": (measured_traffic() or {}).get("bytes_per_launch"),
            "traffic_source": (measured_traffic() or {}).get("source"),
            "alg_bytes_per_launch": avg_bytes, "avg_launch_ms": avg_ms,
            "x_side_ms": sum(x_ms) / len(x_ms), "theta_side_ms": sum(t_ms) / len(t_ms),
            "reduce_kernel_ms_x_side": sum(red_ms[0::2]) / len(red_ms[0::2]),
            "gram_flops_per_launch": float(nnz) * f * (f + 1),
            "gram_tflops": float(nnz) * f * (f + 1) / (avg_ms * 1e-3) / 1e12,
            # the kernel is co-limited (25 flop/B vs a ridge of ~19.7): the same launch
            # against the fp32 MFMA roof (algorithmic flops of the symmetric Gram, FMA = 2)
            "mfma": {"bound": "mfma", "achieved": float(nnz) * f * (f + 1) / (avg_ms * 1e-3) / 1e12,
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": float(nnz) * f * (f + 1) / (avg_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS},
        }
        tr, te = eng.rmse()
        out["rmse"] = {"train": tr, "test": te, "after_iterations": a.warmup + a.steps + len(x_ms)}
        if not a.no_cpu_baseline:
            d = {k: v for k, v in r.numpy().items() if k.startswith("cs")}
            out["cpu_baseline"] = cpu_baseline(d, f, lam, a.solver)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
