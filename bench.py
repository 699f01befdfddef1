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
            out[side] = (rows, nn, t, update)
            tot_nnz += nn
            tot_t += t
        return out, tot_nnz, tot_t

    _, pn, pt = run(1.0 / 64)
    frac = min(1.0, max(1.0 / 64, (target_s / max(pt, 1e-6)) / 64))
    out, tot_nnz, tot_t = run(frac)
    oracle_out = {"gathers": gathers, "x": out["x"][3], "theta": out["theta"][3]}
    # fp64 evaluation of the same half-iterations where the fp32 oracle's own rounding matters: the X
    # side (rows of 10^4..10^5 ratings: a sequential fp32 chain that long is itself 1e-4 off) in full, a
    # 1/16 row sample of the Theta side.  Not timed into the baseline.
    t64 = time.time()
    for side, (ptr, idx, val, _) in sides.items():
        rows = out[side][0] if side == "x" else max(1, out[side][0] // 16)
        nn = int(ptr[rows])
        u64 = np.zeros((rows, f), np.float32)
        pyoracle.half_iteration(ptr[:rows + 1], idx[:nn], val[:nn], gathers[side], u64, f, lam, solver=solver,
                                dtype=np.float64)
        oracle_out[side + "64"] = u64
    oracle_out["fp64_seconds"] = time.time() - t64
    return oracle_out, {
        "value": tot_nnz / tot_t,
        "unit": "ratings/s",
        "cores": pyoracle.num_threads(),
        "kind": "port",
        "sample": (f"one X and one Theta half-iteration of the CPU oracle (fp32, {solver.upper()} solver, OpenMP over "
                   f"rows, {pyoracle.num_threads()} threads) on the first {out['x'][0]} X rows ({out['x'][1]} ratings, "
                   f"{out['x'][2]:.2f} s) and the first {out['theta'][0]} Theta rows ({out['theta'][1]} ratings, "
                   f"{out['theta'][2]:.2f} s) of the same synthetic matrix ({100 * frac:.0f} % of its ratings per side)"),
    }


def parity_at_scale(r, f, lam, solver, cg_iters, oracle_out, dev):
    """the oracle half-iterations the cpu_baseline leg has just computed (same
    matrix, same gather factors, zero warm start) against the HIP half-iterations from the SAME
    factors -- at BASELINE scale: rows of 10^5 ratings cut into dozens of chunks, the reduce kernel's
    slot order, 64-bit gather addresses.  Outside the timed region."""
    from cumf_als_amd import als

    eng = als.ALSEngine(r, f, lam, solver=solver, cg_iters=cg_iters)
    out = {"solver": solver, "gram_mode": als.get_gram_mode()}
    for side, gather_attr, step, plans, ptr in (("x", "thetaT", eng.update_x, eng.x_plans, r.csr_indptr),
                                                ("theta", "XT", eng.update_theta, eng.t_plans, r.csc_indptr)):
        ref = oracle_out[side]
        rows = ref.shape[0]
        getattr(eng, gather_attr).copy_(torch.from_numpy(oracle_out["gathers"][side]))
        upd = eng.XT if side == "x" else eng.thetaT
        upd.zero_()
        step()
        torch.cuda.synchronize()
        got = upd[:rows].cpu().numpy()
        fin = np.isfinite(ref)  # rows without ratings are NaN in both (cg.cu:128)
        lens = np.diff(ptr.cpu().numpy()[:rows + 1].astype(np.int64))
        chunk = plans[0].chunk
        out[f"{side}_side_max_rel"] = float(np.abs(got[fin] - ref[fin]).max() / np.abs(ref[fin]).max())
        # the same rows evaluated in fp64 by the oracle: the HIP path and the fp32 oracle each against it
        r64 = oracle_out[side + "64"]
        n64 = r64.shape[0]
        f64 = np.isfinite(r64)
        scale = np.abs(r64[f64]).max()
        out[f"{side}_side_rows_fp64"] = int(n64)
        out[f"{side}_side_hip_vs_fp64_max_rel"] = float(np.abs(got[:n64][f64] - r64[f64]).max() / scale)
        out[f"{side}_side_oracle32_vs_fp64_max_rel"] = float(np.abs(ref[:n64][f64] - r64[f64]).max() / scale)
        out[f"{side}_side_nan_pattern_equal"] = bool(np.array_equal(np.isnan(got), np.isnan(ref)))
        out[f"{side}_rows_compared"] = int(rows)
        out[f"{side}_rows_chunked"] = int((lens > chunk).sum())
        out[f"{side}_worst_row_len"] = int(lens.max())
        out[f"{side}_max_chunks_per_row"] = int(-(-int(lens.max()) // chunk))
    return out


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


def fast_leg(r, f, lam, a, theta0, eng_default):
    """Informational, outside the timed region of `value`: the same steps in the OPT-IN gram mode "fast"
    (pre-split f16x2 operands, 3 MFMA products per fp32 product, 22 significand bits),
    plus the distance of one X and one Theta half-iteration from the default arithmetic on the same inputs."""
    import torch

    from cumf_als_amd import als

    als.set_gram_mode("fast")
    try:
        eng = als.ALSEngine(r, f, lam, solver=a.solver, cg_iters=a.cg_iters)
        eng.init_factors(theta0)
        for _ in range(max(1, a.warmup)):
            eng.update_x()
            eng.update_theta()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            eng.update_x()
            eng.update_theta()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        flags = als.gram_fast_status()
        tr, te = eng.rmse()
        # same inputs, both arithmetics: the default engine's current factors
        eng.thetaT.copy_(eng_default.thetaT)
        eng.XT.copy_(eng_default.XT)
        eng.update_x()
        x_fast = eng.XT.clone()
        eng.XT.copy_(eng_default.XT)
        eng.update_theta()
        th_fast = eng.thetaT.clone()
        als.set_gram_mode("auto")
        keep_x, keep_t = eng_default.XT.clone(), eng_default.thetaT.clone()
        eng_default.update_x()
        dx = float((x_fast - eng_default.XT).abs().max() / eng_default.XT.abs().max())
        eng_default.XT.copy_(keep_x)
        eng_default.update_theta()
        dth = float((th_fast - eng_default.thetaT).abs().max() / eng_default.thetaT.abs().max())
        eng_default.thetaT.copy_(keep_t)
        return {"opt_in": "CUMF_ALS_GRAM=fast / cumf_set_gram_mode(CUMF_GRAM_FAST)", "ms_per_step": 1e3 * dt,
                "value": 2.0 * r.nnz / dt, "unit": "ratings/s", "range_flags": flags,
                "rmse": {"train": tr, "test": te, "after_iterations": max(1, a.warmup) + a.steps},
                "x_half_iteration_max_rel_vs_default": dx, "theta_half_iteration_max_rel_vs_default": dth}
    finally:
        als.set_gram_mode("auto")


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
    ap.add_argument("--no-fast-leg", action="store_true", help="skip the informational opt-in fast-mode leg")
    ap.add_argument("--no-gram-leg", action="store_true",
                    help="skip the Gram-pass-alone leg (profiling runs: its solve-less launches would skew per-kernel averages)")
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

        if a.scheme == "gather":
            # every rank generates the same matrix on its own GPU (same seed) and keeps zero-copy views
            # of its row / column slabs: nothing travels through host memory
            eng = cdist.DistALS.from_device_ratings(r, f, lam, cdist.HipOps(dev), solver=a.solver, cg_iters=a.cg_iters)
        else:
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
        mode = als.get_gram_mode()
        nb = f // 16 + 1
        wave = mode in ("auto", "fast") and 2 <= nb <= 13
        sv = "CG" if cg else "LU"
        kernel = ((f"cumf::als_wave_kernel<{nb}, {sv}, {100 if f == 100 else 0}>" if nb <= 7
                   else f"cumf::als_wave_multi_kernel<{nb}, 2, {sv}>") if wave
                  else f"cumf::als_item_kernel<{nb}, float4, {sv}>")
        traffic = measured_traffic() or {}
        # matrix-pipe work ISSUED per rating: upper-triangular 16x16 tiles x 2*16*16 flops, x6 bf16
        # products on the split path (als_wave.hip), x3 f16 products in the opt-in fast mode, x1 on the
        # fp32 MFMA path
        products = (3 if mode == "fast" else 6) if wave else 1
        issued = nb * (nb + 1) / 2 * 512.0 * products
        pipe_peak = 2500.0 if wave else MFMA_F32_PEAK_TFLOPS

        def side(ms, nbytes, key):
            ach = nbytes / (ms * 1e-3) / 1e9
            return {"ms": ms, "alg_bytes": nbytes, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                    "traffic": (traffic.get(key) or {}).get("bytes_per_launch"),
                    "gram_tflops_useful": float(nnz) * f * (f + 1) / (ms * 1e-3) / 1e12,
                    "matrix_pipe_frac_issued": float(nnz) * issued / (ms * 1e-3) / 1e12 / pipe_peak}

        xs, ts = sum(x_ms) / len(x_ms), sum(t_ms) / len(t_ms)
        # the Gram pass alone (north_star: ">= 70 % of the HBM roofline on get_hermitian"): the same launches
        # with the in-kernel solve switched off (ablation switch 1: the factors are wrong from here on, so this
        # is the last use of the engine's state before it is re-initialised for the RMSE below)
        gram_only = None
        if wave and nb <= 7 and not a.no_gram_leg:  # (the two-wave kernels of f >= 112 have no such switch)
            keep_x, keep_t = eng.XT.clone(), eng.thetaT.clone()
            g_ms = []
            try:
                als.set_debug_switches(1)
                als.set_kernel_timing(True)
                for _ in range(3):
                    eng.update_x()
                    gx = als.last_kernel_ms()[0]
                    eng.XT.copy_(keep_x)   # both passes gather REAL factors: the matrix pipe's clock depends on the data
                    eng.update_theta()     # (all-NaN tables run 15 % faster: power)
                    g_ms.append((gx, als.last_kernel_ms()[0]))
                    eng.thetaT.copy_(keep_t)
            finally:
                als.set_kernel_timing(False)
                als.set_debug_switches(0)
                eng.XT.copy_(keep_x)
                eng.thetaT.copy_(keep_t)
            gx = sum(v[0] for v in g_ms[1:]) / len(g_ms[1:])
            gt = sum(v[1] for v in g_ms[1:]) / len(g_ms[1:])
            gb_x = 4.0 * f * nnz + 8.0 * nnz + 4.0 * (m + 1)   # Gram + RHS inputs only (no factor write)
            gb_t = 4.0 * f * nnz + 8.0 * nnz + 4.0 * (n + 1)
            gram_only = {"x_side_ms": gx, "theta_side_ms": gt,
                         "x_side_frac_of_hbm_roof": gb_x / (gx * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "theta_side_frac_of_hbm_roof": gb_t / (gt * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "same kernel with the solve switched off (cumf_set_debug_switches(1)); the Theta "
                                 "side gathers a 7 MB table from L2, so its fraction is bytes-equivalent, not HBM traffic"}
        out["dtype"] = ("f32" if not wave else
                        "f32 (opt-in fast mode: pre-split f16x2 operands, 3 products, 22-bit significand, fp32 accumulate)"
                        if mode == "fast" else "f32 (bf16x3-split products on the bf16 matrix pipe, fp32 accumulate)")
        out["roofline"] = {
            "bound": "hbm", "kernel": kernel,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic.get("bytes_per_launch"), "traffic_source": traffic.get("source"),
            "alg_bytes_per_launch": avg_bytes, "avg_launch_ms": avg_ms,
            "x_side_ms": xs, "theta_side_ms": ts,
            "x_side": side(xs, bx, "x_side"), "theta_side": side(ts, bt, "theta_side"),
            "reduce_kernel_ms_x_side": sum(red_ms[0::2]) / len(red_ms[0::2]),
            "gram_mode": mode,
            "gram_pass_alone": gram_only,
            "gram_flops_per_launch": float(nnz) * f * (f + 1),
            "gram_tflops": float(nnz) * f * (f + 1) / (avg_ms * 1e-3) / 1e12,
            # the matrix pipe next to the HBM roof: flops issued (tile padding and, on the split path,
            # the six bf16 products per fp32 product included) against the pipe's dense peak
            "mfma": {"bound": "mfma", "pipe": (f"{'f16' if mode == 'fast' else 'bf16'} ({products} products per fp32 product)"
                                                if wave else "fp32"),
                     "achieved": float(nnz) * issued / (avg_ms * 1e-3) / 1e12, "peak": pipe_peak, "unit": "TFLOP/s",
                     "frac": float(nnz) * issued / (avg_ms * 1e-3) / 1e12 / pipe_peak},
            # what a perfect kernel of this design would take: the larger of the HBM time of the
            # algorithmic bytes and the matrix-pipe time of the issued flops
            "floor_ms": {"hbm": avg_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, "matrix_pipe": float(nnz) * issued / (pipe_peak * 1e12) * 1e3},
        }
        tr, te = eng.rmse()
        out["rmse"] = {"train": tr, "test": te, "after_iterations": a.warmup + a.steps + len(x_ms)}
        if mode == "auto" and wave and not a.no_fast_leg:
            out["gram_fast_mode"] = fast_leg(r, f, lam, a, theta0, eng)
        if not a.no_cpu_baseline:
            d = {k: v for k, v in r.numpy().items() if k.startswith("cs")}
            oracle_out, out["cpu_baseline"] = cpu_baseline(d, f, lam, a.solver)
            del eng
            out["parity_at_scale"] = parity_at_scale(r, f, lam, a.solver, a.cg_iters, oracle_out, dev)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
