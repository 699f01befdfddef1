#!/usr/bin/env python3
"""bench.py -- ratings/sec per ALS half-iteration on synthetic Netflix-shaped ratings.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one full pass of the hot path over the dataset: update-X + update-Theta
(RHS + Gram + solve for every row of both sides), i.e. TWO half-iterations.  Inputs are
resident in HBM before the timed region.  Workload = BASELINE.json configs[1]:
Netflix shape (17770 x 480189, 99 072 112 ratings), f = 100, lambda = 0.048, LU solver.

Printed JSON (one line, rank 0):
  value       2 * nnz * K / t           ratings/s per half-iteration, whole job
  roofline    dominant kernel = the per-item Gram(+solve) launch that takes the larger part of the step (the
              Theta-side instance at the headline shape; its name is read back from the library: the symbol
              that was dispatched); achieved = algorithmic bytes of that half-iteration (SURVEY.md 8d:
              4 f nnz + 8 nnz + 4 (rows+1) + 4 f rows) / its HIP-event duration; peak = 8 TB/s HBM.  Both
              launches are listed (x_side, theta_side) and their mean (step_mean).  Emitted
              for every --f, --solver and for --shape hugewiki (per-GPU slab).  `traffic` is replayed
              from the committed rocprofv3 --pmc passes (profiles/traffic.json), and only when that file
              was collected for the kernel that was dispatched here.
  cpu_baseline  the CPU oracle (oracle/, "port": the reference has no CPU path) timed on
              this host's cores on a row sample of the same matrix, N = 1 only.

N > 1 on a box with fewer GPUs than ranks (tests): CUMF_BENCH_BACKEND=gloo stands in for RCCL (collectives
staged through the host) and ranks share devices (local_rank % device_count).
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


def rmse_log_parity(r, f, lam, solver, cg_iters, iters=10):
    """the reference's own run length -- ITERS = 10 (main.cpp:17) full iterations of its
    loop (als.cu:727-1022) -- through doALS with the reference's batch setting X_BATCH = 1, THETA_BATCH = 3
    (test_als.sh:16) against oracle_doALS on the same matrix, the same srand(0) start (main.cpp:72-78) and the same
    truncated test grid: per-iteration train / test RMSE of both and their largest difference (north_star: RMSE to 1e-4).
    The same ten iterations once more with the RMSE kernel in place of the train SSE out of the Theta update
    (CUMF_ALS_RMSE=kernel; the factors are the same bit for bit): the difference of the two HIP logs is the fused SSE's
    share of any deviation.  Outside the timed region; ~13 s of oracle per iteration on 128 cores."""
    from cumf_als_amd import als
    from oracle import pyoracle

    d = r.numpy()
    th0, x0 = pyoracle.init_factors(r.m, r.n, f)

    def hip(rmse_mode):
        keep = os.environ.get("CUMF_ALS_RMSE")
        if rmse_mode:
            os.environ["CUMF_ALS_RMSE"] = rmse_mode
        try:
            t0 = time.time()
            th, x, rm, log = als.do_als(d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indices"], d["csc_indptr"],
                                        d["csc_data"], d["coo_row"], d["test_row"], d["test_col"], d["test_data"], r.m, r.n,
                                        f, r.nnz, r.nnz_test, lam, iters, 1, 3, torch.cuda.current_device(),
                                        thetat_init=th0, xt_init=x0, solver=solver, cg_iters=cg_iters, return_log=True)
            return th, x, rm, np.asarray(log, np.float64), time.time() - t0
        finally:
            if keep is None:
                os.environ.pop("CUMF_ALS_RMSE", None)
            else:
                os.environ["CUMF_ALS_RMSE"] = keep

    th_h, x_h, rm_h, log_h, t_hip = hip(None)
    th_k, x_k, _, log_k, _ = hip("kernel")
    th_o, x_o = th0.copy(), x0.copy()
    t0 = time.time()
    rm_o, log_o = pyoracle.do_als(d, th_o, x_o, r.m, r.n, f, lam, iters, x_batch=1, theta_batch=3, solver=solver,
                                  cg_iters=cg_iters)
    t_or = time.time() - t0
    # the final train RMSE re-evaluated in fp64 on the CPU from each side's own factors: what every reported value misses
    exact = lambda th, x: float(np.sqrt(pyoracle.sse(d["csr_data"], d["coo_row"], d["csr_indices"], th, x, r.nnz, f,
                                                     dtype=np.float64) / r.nnz))
    ex_h, ex_o = exact(th_h, x_h), exact(th_o, x_o)
    return {"iterations": iters, "x_batch": 1, "theta_batch": 3, "solver": solver,
            "hip": [[float(v) for v in row] for row in log_h], "oracle": [[float(v) for v in row] for row in log_o],
            "max_abs_diff": float(np.abs(log_h - log_o).max()),
            "max_abs_diff_per_iteration": [float(v) for v in np.abs(log_h - log_o).max(1)],
            "hip_rmse_kernel": {"log": [[float(v) for v in row] for row in log_k],
                                "factors_bit_identical_to_fused": bool(np.array_equal(th_h, th_k, equal_nan=True)
                                                                       and np.array_equal(x_h, x_k, equal_nan=True)),
                                "fused_minus_kernel_max_abs": float(np.abs(log_h - log_k).max()),
                                "kernel_vs_oracle_max_abs": float(np.abs(log_k - log_o).max())},
            "final_train_rmse_fp64_reevaluation": {"hip_factors": ex_h, "oracle_factors": ex_o,
                                                   "hip_fused_reported_minus": float(log_h[-1, 0] - ex_h),
                                                   "hip_kernel_reported_minus": float(log_k[-1, 0] - ex_h),
                                                   "oracle32_reported_minus": float(log_o[-1, 0] - ex_o)},
            "final_test_rmse": {"hip": float(rm_h), "oracle": float(rm_o)},
            "doALS_seconds_incl_upload": round(t_hip, 3), "oracle_seconds": round(t_or, 1)}


def measured_traffic(kernel: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (tools/collect_profiles.sh -> profiles/traffic.json): (2 * FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE
    doubled as MI355X_MICROARCH.md prescribes for gfx950.  The file is keyed by the kernel name rocprofv3
    printed; returns (entry or None, note): a profile of ANOTHER kernel is never replayed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            table = json.load(fh)
    except (OSError, ValueError):
        return None, "profiles/traffic.json is missing"
    for ent in table.get("kernels", []):
        if ent.get("kernel", "").replace(" ", "") == kernel.replace(" ", ""):
            return ent, None
    have = [e.get("kernel") for e in table.get("kernels", [])]
    return None, f"profiles/traffic.json holds no PMC pass for the dispatched kernel {kernel!r} (it has {have})"


def mode_leg(mode, r, f, lam, a, theta0, eng_default):
    """Informational, outside the timed region of `value`: the same steps in another gram mode -- "fast", the
    OPT-IN 22-bit arithmetic (pre-split f16x2 operands, 3 MFMA products per fp32 product), or
    "exact", the reference's own arithmetic (fp32 MFMA = the k-ordered fmaf chain of als.h:39-143, bit-exact
    Gram) -- plus the distance of one X and one Theta half-iteration from the default arithmetic on the same
    inputs."""
    return fast_leg(r, f, lam, a, theta0, eng_default, mode)


def fast_leg(r, f, lam, a, theta0, eng_default, mode="fast"):
    import torch

    from cumf_als_amd import als

    als.set_gram_mode(mode)
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
        flags = als.gram_fast_status() if mode == "fast" else 0
        kernel = als.last_kernel_name()
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
        return {"opt_in": f"CUMF_ALS_GRAM={mode} / cumf_set_gram_mode(CUMF_GRAM_{mode.upper()})", "kernel": kernel,
                "ms_per_step": 1e3 * dt, "value": 2.0 * r.nnz / dt, "unit": "ratings/s", "range_flags": flags,
                "rmse": {"train": tr, "test": te, "after_iterations": max(1, a.warmup) + a.steps},
                "x_half_iteration_max_rel_vs_default": dx, "theta_half_iteration_max_rel_vs_default": dth}
    finally:
        als.set_gram_mode("auto")


def gram_pass_alone(a):
    """The Gram pass alone (north_star: ">= 70 % of the HBM roofline on get_hermitian"): the same launches with
    the in-kernel solve compiled to a switch -- that switch exists only in the profiling build
    libALS_ablate.so, so the leg runs tools/gram_pass_alone.py in a process of its own with CUMF_ALS_LIB
    pointing at it.  None when that library was not built."""
    import subprocess

    from cumf_als_amd import lib as _lib

    if not os.path.exists(_lib.ABLATE_LIB_PATH):
        return None
    env = dict(os.environ, CUMF_ALS_LIB=_lib.ABLATE_LIB_PATH)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "gram_pass_alone.py"), "--shape", a.shape, "--f", str(a.f),
           "--solver", a.solver, "--seed", str(a.seed), "--scale", str(a.scale)]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # informational leg: never takes the bench line down
        return {"error": f"{type(e).__name__}: {e}"}


def rank_diagnostics(eng, als, dev, world, backend, steps=3):
    """N > 1: what every rank spends per half-iteration, so that a scaling curve can be read from
    the bench line alone.  Outside the timed region.  Per half-iteration, ranks aligned by a barrier first:
      half_ms    torch events on the compute stream around update_x / update_theta -- kernels, whatever the compute
                 stream waits for (collectives that are not hidden, the slowest rank) and launch gaps;
      kernel_ms  HIP events of libALS around its Gram(+solve) launches, summed over the launches of the half-iteration
                 (pipeline pieces, Theta batches): cumf_kernel_ms_since_reset.
    non_kernel_ms = half_ms - kernel_ms is the exposed part: collectives + waiting for other ranks + launch gaps (reduce
    scheme: also the batched solve and the unpack of the reduced Grams, which are separate small kernels).
    Returns {"per_rank": {...lists over ranks...}, "max_over_ranks": {...}} on every rank."""
    import torch.distributed as dist

    als.set_kernel_timing(True)
    rows = []
    for _ in range(steps):
        row = []
        for fn in (eng.update_x, eng.update_theta):
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            als.kernel_ms_since_reset()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            item, red, launches = als.kernel_ms_since_reset()
            row += [e0.elapsed_time(e1), item + red, float(launches)]
        rows.append(row)
    als.set_kernel_timing(False)
    mine = torch.tensor(rows, dtype=torch.float64).mean(0)            # [6]
    allr = [torch.zeros_like(mine) for _ in range(world)]
    if backend == "nccl":
        mine_d = mine.to(dev)
        allr = [torch.zeros_like(mine_d) for _ in range(world)]
        dist.all_gather(allr, mine_d)
    else:
        dist.all_gather(allr, mine)
    t = torch.stack([v.cpu() for v in allr])                           # [world, 6]
    names = ("x_half_ms", "x_kernel_ms", "x_launches", "theta_half_ms", "theta_kernel_ms", "theta_launches")
    per_rank = {k: [round(float(v), 4) for v in t[:, i]] for i, k in enumerate(names)}
    per_rank["x_non_kernel_ms"] = [round(h - k, 4) for h, k in zip(per_rank["x_half_ms"], per_rank["x_kernel_ms"])]
    per_rank["theta_non_kernel_ms"] = [round(h - k, 4) for h, k in zip(per_rank["theta_half_ms"], per_rank["theta_kernel_ms"])]
    mx = {k: max(v) for k, v in per_rank.items()}
    return {"per_rank": per_rank, "max_over_ranks": mx, "steps_averaged": steps}


def make_slab_engine(a, shp, world, rank, dev, f, lam, theta0, solver, cg_iters, theta_batch):
    """hugewiki scale (BASELINE.json configs[3]): WEAK scaling -- every rank generates and keeps one row slab of 1/8 of
    the hugewiki matrix (6.26 M x 39 780, 388 M ratings); the `reduce` scheme never materialises the whole matrix anywhere.
    Returns (engine, slab ratings, m of the whole job, nnz of the whole job)."""
    from cumf_als_amd import datagen
    from cumf_als_amd import dist as cdist

    s = a.scale
    m_slab, n = max(2, int(shp["m"] * s) // 8), max(2, int(shp["n"] * s))
    nnz_slab = max(int(shp["nnz"] * s * s) // 8, m_slab + n)
    r = datagen.synth_ratings(m_slab, n, nnz_slab, 4096, seed=a.seed + 1000 * (rank + 1), device=dev, col_seed=a.seed)
    m, nnz = m_slab * world, nnz_slab * world
    return slab_engine(a, r, m, n, world, dev, f, lam, theta0, solver, cg_iters, theta_batch), r, m, nnz


def slab_engine(a, r, m, n, world, dev, f, lam, theta0, solver, cg_iters, theta_batch):
    """The `reduce`-scheme engine over one rank's slab `r` (native half-iterations or torch.distributed collectives:
    whatever cumf_als_amd.dist.set_native / CUMF_DIST_NATIVE selects at this moment)."""
    from cumf_als_amd import dist as cdist

    xb = np.arange(world + 1, dtype=np.int64) * r.m
    # THETA_BATCH = 3 as the reference's hugewiki run (hugewiki.cu:27-41): with more than one rank the
    # reduce-scatter of batch b runs under the partial-Gram pass of batch b + 1
    # --reference-solvers: the per-side solvers of the reference's own hugewiki run -- X by CG with 100 iterations
    # (hugewiki.cu:2569), Theta by the batched LU on the reduced Grams (hugewiki.cu:2732)
    sides = dict(solver_x="cg", cg_iters_x=100, solver_theta="lu") if getattr(a, "reference_solvers", False) else {}
    eng = cdist.DistALS.from_local_slab(m, n, xb, r.csr_indptr, r.csr_indices, r.csr_data, f, lam, cdist.HipOps(dev),
                                        solver=solver, cg_iters=cg_iters,
                                        theta_batch=theta_batch if theta_batch > 0 else (3 if world > 1 else 1), **sides)
    eng.init_factors(theta0)
    return eng


def hugewiki_prepare(a, datagen, dev, world, rank):
    """The local part of the hugewiki leg: this rank's 1/8 slab and its engine (no collective)."""
    shp = datagen.SHAPES["hugewiki"]
    f, lam = 100, shp["lam"]
    n = max(2, int(shp["n"] * a.scale))
    g = torch.Generator(device="cpu")
    g.manual_seed(a.seed)
    theta0 = (0.2 * torch.rand((n, f), generator=g, dtype=torch.float32)).numpy()
    t0 = time.time()
    from cumf_als_amd import dist as cdist

    cdist.set_native(False)  # first with torch.distributed collectives; hugewiki_run then repeats it on the native path
    try:
        eng, r, m, nnz = make_slab_engine(a, shp, world, rank, dev, f, lam, theta0, "cg", 6, a.theta_batch)
    finally:
        cdist.set_native(None)
    torch.cuda.synchronize()
    return eng, r, m, nnz, n, f, lam, time.time() - t0, theta0


def timed_steps(eng, dev, backend, steps, warmup=1):
    """`steps` iterations of a distributed engine under the bench's timing rule (barrier + synchronize on both sides, MAX
    over ranks) -> seconds.  Every rank."""
    import torch.distributed as dist

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        eng.update_x()
        eng.update_theta()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.update_x()
        eng.update_theta()
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def promote_native(res: dict, nat: dict, keys) -> None:
    """The native path is the product's default: when its leg came through, its numbers become the object's own and the
    torch.distributed ones move under "torch_collectives"; a failed native leg stays as "native": {"error": ...} beside the
    torch.distributed numbers."""
    if "error" in nat:
        res["collectives"] = "torch.distributed"
        res["native"] = nat
        return
    if "ms_per_step" in nat and nat["ms_per_step"] > 1.02 * res["ms_per_step"]:
        # measured slower (e.g. the host-staged stand-in transport of the one-GPU tests): the object keeps the faster
        # configuration's numbers -- CUMF_DIST_NATIVE=0 selects it -- and carries the native ones beside them
        res["collectives"] = "torch.distributed (the native half-iterations measured slower here: see native)"
        res["native"] = nat
        res["native_equals_torch_collectives"] = nat.get("factors_equal")
        return
    res["torch_collectives"] = {k: res[k] for k in keys if k in res}
    res.update({k: nat[k] for k in keys if k in nat})
    res["collectives"] = "native (cumf_dist_*, als_dist.cpp: kernels and collectives enqueued from C++)"
    res["native_equals_torch_collectives"] = nat.get("factors_equal")


HW_KEYS = ("value", "ms_per_step", "x_half_ms", "theta_half_ms", "non_kernel_ms", "per_rank")


def hugewiki_run(a, als, state, dev, world, backend, steps=3, guard=None):
    """The collective part of the hugewiki leg (every rank: it holds collectives): the engine with torch.distributed
    collectives first (its numbers are in the guard's line before anything else is tried), then the same slab on the native
    half-iterations (cumf_dist_reduce_update_theta), which take the object over when they come through."""
    import torch.distributed as dist

    from cumf_als_amd import dist as cdist

    eng, r, m, nnz, n, f, lam, t_gen, theta0 = state

    def measure(e):
        elapsed = timed_steps(e, dev, backend, steps)
        diag = rank_diagnostics(e, als, dev, world, backend, steps=2)
        return {"value": 2.0 * nnz * steps / elapsed, "ms_per_step": 1e3 * elapsed / steps,
                "x_half_ms": diag["max_over_ranks"]["x_half_ms"], "theta_half_ms": diag["max_over_ranks"]["theta_half_ms"],
                "non_kernel_ms": {"x": diag["max_over_ranks"]["x_non_kernel_ms"],
                                  "theta": diag["max_over_ranks"]["theta_non_kernel_ms"]},
                "per_rank": diag["per_rank"]}

    theta_batches = len(eng.t_batches)
    res = {"unit": "ratings/s", "steps": steps, "warmup": 1, "scaling": "weak", "n_ranks_seen": dist.get_world_size(),
           "scheme": "reduce", "theta_batch": theta_batches, "solver": "cg(6)",
           "workload": f"hugewiki-shape synthetic ratings {m}x{n}, nnz={nnz}, f={f}, lambda={lam}: one 1/8 row slab "
                       f"({r.m} rows, {r.nnz} ratings) per GPU (BASELINE.json configs[3])",
           "gen_seconds": round(t_gen, 2), "collectives": "torch.distributed"}
    res.update(measure(eng))
    eng.init_factors(theta0)
    eng.iterate(1)
    th_t, x_t = eng.thetaT.clone(), eng.XT.clone()
    eng.close()
    del eng
    if guard is not None:
        guard.note(lambda line: line.__setitem__("hugewiki", dict(res)))
        guard.pending = ("hugewiki", "native")
    nat = native_leg(lambda: slab_engine(a, r, m, n, world, dev, f, lam, theta0, "cg", 6, a.theta_batch), measure,
                     theta0, (th_t, x_t), dev, world, dist.get_rank(), backend)
    promote_native(res, nat, HW_KEYS)
    return res


def native_leg(make_engine, measure, theta0, torch_factors, dev, world, rank, backend):
    """One engine on the native half-iterations (als_dist.cpp), measured by `measure`, and checked against the factors the
    torch.distributed engine produced in one iteration from the same start (same kernels, same plans: bit-identical).  A
    local failure is agreed on with ONE all-reduce before the leg's first collective; CUMF_BENCH_FAIL_NATIVE=<rank> /
    hang<rank> injects one (tests).  Returns the measured object or {"error": ...}; every rank."""
    import torch.distributed as dist

    from cumf_als_amd import dist as cdist

    err = None
    try:
        if os.environ.get("CUMF_BENCH_FAIL_NATIVE") == str(rank):
            raise RuntimeError(f"injected failure on rank {rank} (CUMF_BENCH_FAIL_NATIVE)")
        if os.environ.get("CUMF_BENCH_FAIL_NATIVE") == f"hang{rank}":
            time.sleep(1e6)
    except BaseException as e:  # noqa: BLE001
        err = f"rank {rank}: {type(e).__name__}: {e}"
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        return {"error": err or "another rank failed before the native leg"}
    eng = None
    try:
        cdist.set_native(True)
        eng = make_engine()
        if eng._ncomm is None:
            raise RuntimeError("the engine did not take the native path")
        nat = measure(eng)
        nat["transport"] = eng._ncomm.name
        eng.init_factors(theta0)
        eng.iterate(1)
        same = torch.equal(eng.thetaT, torch_factors[0]) and torch.equal(eng.XT, torch_factors[1])
        diff = max(float((eng.thetaT - torch_factors[0]).abs().nan_to_num(0.0).max().item()) if eng.thetaT.numel() else 0.0,
                   float((eng.XT - torch_factors[1]).abs().nan_to_num(0.0).max().item()) if eng.XT.numel() else 0.0)
        flag = torch.tensor([1.0 if same else 0.0, -diff], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        nat["factors_equal"] = bool(int(flag[0].item()))
        nat["factors_max_abs_diff"] = -float(flag[1].item())
        return nat
    except Exception as e:  # noqa: BLE001 -- symmetric failures (an RCCL error on every rank); a hang is the guard's
        return {"error": f"rank {rank}: {type(e).__name__}: {e}"}
    finally:
        cdist.set_native(None)
        if eng is not None:
            eng.close()


def hugewiki_leg(a, als, datagen, dev, world, rank, backend, steps=3):  # noqa: D401
    """N > 1: north_star's ">= 6 x at 8 GPUs" is stated on the hugewiki-scale synthetic shape
    (hugewiki.cu:27-41: weak scaling, one 1/8 row slab per GPU, X row-sharded, partial Grams reduce-scattered over RCCL),
    while the default N > 1 line is the Netflix shape (strong scaling).  This leg runs the slab configuration right behind
    the Netflix one -- same process group, `reduce` scheme, THETA_BATCH = 3, CG(6) -- with the same timing rule (barrier +
    synchronize on both sides, MAX over ranks), so that the one line the driver records at N = 2, 4, 8 carries both.
    Returns the object of `hugewiki` in the bench line (every rank must call it: it holds collectives).  bench.py itself
    goes through `hugewiki_leg_guarded`."""
    return hugewiki_run(a, als, hugewiki_prepare(a, datagen, dev, world, rank), dev, world, backend, steps)


class LineGuard:
    """N > 1: the ONE JSON line of the run must survive whatever happens to the hugewiki leg behind it (VERDICT r05 weak 3:
    the driver gets one shot at N = 8).  Rank 0 hands the finished Netflix line to the guard BEFORE the leg starts; the line
    is printed exactly once -- by `emit` when the leg returns, or by the guard's own thread, with `hugewiki: {"error": ...}`,
    when the leg has not returned by the deadline (a hang inside a collective) or the process is told to terminate
    (torchrun's SIGTERM after another rank died: delivered through signal.set_wakeup_fd, so it is seen even while the main
    thread sits in a C++ collective).  Every rank runs a guard: the ranks without a line just leave, so that the launcher
    returns."""

    def __init__(self, line, deadline_s: float):
        import signal
        import threading

        # the ranks without a line leave a little later than rank 0 prints, and quietly (exit code 0)
        self.line, self.deadline_s = line, deadline_s + (0.0 if line is not None else 10.0)
        self._lock = threading.Lock()
        self._done = False
        # where an error lands when the deadline passes: the key path of the leg that is running, e.g. ("native",),
        # ("hugewiki",), ("hugewiki", "native") -- what earlier legs left in the line (`note`) stays
        self.pending = ("hugewiki",)
        self._rd, self._wr = os.pipe()
        os.set_blocking(self._wr, False)
        try:
            signal.signal(signal.SIGTERM, lambda *_: None)  # a Python-level handler must exist for the wakeup fd to fire
            signal.set_wakeup_fd(self._wr, warn_on_full_buffer=False)
        except ValueError:  # not the main thread (tests that import bench): the deadline still holds
            pass
        self._thread = threading.Thread(target=self._watch, daemon=True)
        self._thread.start()

    def note(self, fn) -> None:
        """A finished leg records its result in the line (under the lock: the watcher may be printing)."""
        with self._lock:
            if not self._done and self.line is not None:
                fn(self.line)

    def _print(self, extra, why=None) -> bool:
        with self._lock:
            if self._done:
                return False
            self._done = True
            if self.line is not None:
                if why is not None:
                    d = self.line
                    for k in self.pending[:-1]:
                        d = d.setdefault(k, {})
                    d[self.pending[-1]] = {"error": why}
                elif extra is not None:
                    self.line["hugewiki"] = extra
                print(json.dumps(self.line), flush=True)
            return True

    def _watch(self):
        import select

        ready, _, _ = select.select([self._rd], [], [], self.deadline_s)
        why = ("terminated by the launcher (another rank failed)" if ready
               else f"no result after {self.deadline_s:.0f} s (collective hang?)")
        self._print(None, why)  # no-op when the line is already out
        sys.stdout.flush()
        os._exit(0)

    def emit(self, extra) -> None:
        self._print(extra)


def hugewiki_leg_guarded(a, als, datagen, dev, world, rank, backend, guard=None):
    """`hugewiki_leg` so that no rank's failure takes the line down: the local part (slab generation, engine construction:
    no collective) runs under try / except on every rank and the ranks agree on the outcome with ONE all-reduce (MIN) before
    the first collective of the leg; the collective part runs under try / except too (symmetric failures: an RCCL error)
    and under the LineGuard's deadline (asymmetric ones: a hang).  CUMF_BENCH_FAIL_LEG=<rank> / hang<rank> injects a failure /
    a hang on that rank (tests/test_dist_gpu.py)."""
    import torch.distributed as dist

    state, err = None, None
    try:
        if os.environ.get("CUMF_BENCH_FAIL_LEG") == str(rank):
            raise RuntimeError(f"injected failure on rank {rank} (CUMF_BENCH_FAIL_LEG)")
        if os.environ.get("CUMF_BENCH_FAIL_LEG") == f"hang{rank}":  # this rank never reaches the leg's collectives
            time.sleep(1e6)
        state = hugewiki_prepare(a, datagen, dev, world, rank)
    except BaseException as e:  # noqa: BLE001 -- OOM included: the line matters more than the leg
        err = f"rank {rank}: {type(e).__name__}: {e}"
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if state is not None:
            state[0].close()
        return {"error": err or "another rank failed while preparing its slab"}
    try:
        return hugewiki_run(a, als, state, dev, world, backend, guard=guard)
    except Exception as e:  # noqa: BLE001
        return {"error": f"rank {rank}: {type(e).__name__}: {e}"}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a torchrun environment (VERDICT r03 missing 2): re-run this very command
    line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per
    GPU over RCCL) and hand its exit code back.  Rank 0 of the child prints the JSON line on the inherited stdout."""
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    env["CUMF_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


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
    ap.add_argument("--theta-batch", type=int, default=0,
                    help="--shape hugewiki: Theta batches of the reduce scheme (0 = 3 when more than one rank, else 1)")
    ap.add_argument("--no-fast-leg", action="store_true",
                    help="skip the informational legs in the other gram modes (opt-in fast, reference-exact)")
    ap.add_argument("--allow-missing-traffic", action="store_true",
                    help="headline workload only: print traffic: null instead of failing when profiles/traffic.json "
                         "has no PMC pass of the dispatched kernel")
    ap.add_argument("--no-gram-leg", action="store_true",
                    help="skip the Gram-pass-alone leg (profiling runs: its solve-less launches would skew per-kernel averages)")
    ap.add_argument("--reference-solvers", action="store_true",
                    help="--shape hugewiki: X by CG(100), Theta by LU, as the reference's hugewiki run (hugewiki.cu:2569,2732)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-native-leg", action="store_true",
                    help="N > 1: keep the line on torch.distributed collectives (skip the native half-iterations behind it)")
    ap.add_argument("--no-hugewiki-leg", action="store_true",
                    help="N > 1: skip the hugewiki-slab leg (weak scaling, reduce scheme) behind the default Netflix one")
    ap.add_argument("--no-rmse-log", action="store_true",
                    help="skip the 10-iteration doALS-vs-oracle RMSE log of the parity_at_scale leg (~2 min of oracle)")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a.gpus)

    from cumf_als_amd import als, datagen

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE=1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    backend = os.environ.get("CUMF_BENCH_BACKEND", "nccl")  # "gloo": stand-in for RCCL when ranks share a GPU (tests)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        import datetime

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # an explicit collective timeout: a rank that never arrives becomes an error after pg_timeout seconds instead of a
        # silent hang (the LineGuard below prints the line first where it can)
        pg_timeout = datetime.timedelta(seconds=float(os.environ.get("CUMF_BENCH_PG_TIMEOUT", "900")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)

    shp = datagen.SHAPES[a.shape]
    s = a.scale
    lam, f = shp["lam"], a.f
    slab_mode = a.shape == "hugewiki"
    t0 = time.time()
    if slab_mode:
        n = max(2, int(shp["n"] * s))
    else:
        m, n = max(2, int(shp["m"] * s)), max(2, int(shp["n"] * s))
        nnz, nnz_test = max(int(shp["nnz"] * s * s), m + n), max(int(shp["nnz_test"] * s * s), 512)
        r = datagen.synth_ratings(m, n, nnz, nnz_test, seed=a.seed, device=dev)
    g = torch.Generator(device="cpu")
    g.manual_seed(a.seed)
    theta0 = (0.2 * torch.rand((n, f), generator=g, dtype=torch.float32)).numpy()

    item_ms = []
    kernels = {}
    if slab_mode:
        eng, r, m, nnz = make_slab_engine(a, shp, world, rank, dev, f, lam, theta0, a.solver, a.cg_iters, a.theta_batch)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    if slab_mode:

        def step(timed):
            eng.update_x()
            if timed:
                item_ms.append(als.kernel_ms_since_reset()[:2])  # summed over the launches of the half-iteration
                kernels["x"] = als.last_kernel_name()
            eng.update_theta()
            if timed:
                item_ms.append(als.kernel_ms_since_reset()[:2])  # summed over the launches of the half-iteration
                kernels["theta"] = als.last_kernel_name()

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
                item_ms.append(als.kernel_ms_since_reset()[:2])  # summed over the launches of the half-iteration
                kernels["x"] = als.last_kernel_name()
            eng.update_theta()
            if timed:
                item_ms.append(als.kernel_ms_since_reset()[:2])  # summed over the launches of the half-iteration
                kernels["theta"] = als.last_kernel_name()

        def barrier():
            torch.cuda.synchronize()
    else:
        import torch.distributed as dist

        from cumf_als_amd import dist as cdist

        def make_engine():
            if a.scheme == "gather":
                # every rank generates the same matrix on its own GPU (same seed) and keeps zero-copy views
                # of its row / column slabs: nothing travels through host memory
                e = cdist.DistALS.from_device_ratings(r, f, lam, cdist.HipOps(dev), solver=a.solver, cg_iters=a.cg_iters)
            else:
                # reduce scheme on a shape whose factors would fit one GPU: every rank keeps zero-copy views of ITS row slab
                # of the device-resident matrix (only the row pointer visits the host); the slab-local CSC is built on the
                # device (from_local_slab), as for the hugewiki slabs
                rp = r.csr_indptr.cpu().numpy().astype(np.int64)
                xb = cdist.balanced_slabs(rp, world, cdist.solve_row_cost(f, a.solver))
                x0, x1 = int(xb[rank]), int(xb[rank + 1])
                rowptr_l = torch.from_numpy(rp[x0:x1 + 1] - rp[x0]).to(dev)
                e = cdist.DistALS.from_local_slab(m, n, xb, rowptr_l, r.csr_indices[rp[x0]:rp[x1]], r.csr_data[rp[x0]:rp[x1]],
                                                  f, lam, cdist.HipOps(dev), solver=a.solver, cg_iters=a.cg_iters,
                                                  theta_batch=a.theta_batch if a.theta_batch > 0 else 3)
            e.init_factors(theta0)
            return e

        # The line is first measured with torch.distributed collectives driven from Python (rounds 1-5: the conservative
        # path) and handed to the LineGuard; the native half-iterations (als_dist.cpp, the product's default) are measured
        # behind it under the guard and take the line over when they come through.
        cdist.set_native(False)
        try:
            eng = make_engine()
        finally:
            cdist.set_native(None)

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

        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    diag = None
    if world > 1:
        diag = rank_diagnostics(eng, als, dev, world, backend)
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
                                   + (f", row slab per GPU (1/8 hugewiki), reduce scheme over {world} GPU(s)"
                                      + (", X by CG(100) / Theta by LU (hugewiki.cu:2569,2732)" if a.reference_solvers else "")
                                      if slab_mode
                                      else ", X_BATCH=1 THETA_BATCH=1, fused Gram+solve" if world == 1
                                      else f", {a.scheme} scheme over {world} GPUs"),
                       "step": "update-X + update-Theta (two half-iterations)", "gen_seconds": round(t_gen, 2)},
        }

    if rank == 0 and world > 1:
        out["ranks"] = {"n_ranks_seen": dist.get_world_size(), "backend": backend,
                        "devices_visible": torch.cuda.device_count(),
                        "scheme": "reduce" if slab_mode else a.scheme,
                        "self_launched": os.environ.get("CUMF_BENCH_SELF_LAUNCHED") == "1",
                        "note": "half_ms = compute-stream time of a half-iteration (ranks aligned by a barrier first); "
                                "kernel_ms = libALS Gram(+solve) launches inside it (HIP events, summed); non_kernel_ms = "
                                "exposed collectives + waiting for the slowest rank + launch gaps"
                                + (" + the batched solve / unpack kernels of the reduce scheme" if slab_mode or a.scheme == "reduce" else ""),
                        **diag}
    if world > 1:
        # The Netflix line is complete here.  The hugewiki leg (the configuration the 8-GPU target is defined on) runs behind
        # it under a guard that prints the line -- with hugewiki: {"error": ...} -- if the leg hangs or the launcher tears
        # the job down; a failure inside the leg comes back as the same object (every rank: the leg holds collectives).
        guard = LineGuard(out, float(os.environ.get("CUMF_BENCH_LEG_DEADLINE", "600")))
        if not slab_mode and not a.no_native_leg:
            # the same K steps on the native half-iterations; one iteration from theta0 on both engines for the factors
            guard.pending = ("native",)
            eng.init_factors(theta0)
            eng.iterate(1)
            factors_t = (eng.thetaT.clone(), eng.XT.clone())

            def measure(e):
                elapsed_n = timed_steps(e, dev, backend, a.steps, a.warmup)
                d = rank_diagnostics(e, als, dev, world, backend)
                return {"value": 2.0 * nnz * a.steps / elapsed_n, "ms_per_step": 1e3 * elapsed_n / a.steps,
                        "ranks": {"per_rank": d["per_rank"], "max_over_ranks": d["max_over_ranks"],
                                  "steps_averaged": d["steps_averaged"]}}

            nat = native_leg(make_engine, measure, theta0, factors_t, dev, world, rank, backend)

            def take_over(line):
                if "error" not in nat and nat["ms_per_step"] <= 1.02 * line["ms_per_step"]:
                    # rank 0's line: the diagnostics of the path whose numbers it carries
                    ranks_t = {k: line["ranks"].pop(k) for k in ("per_rank", "max_over_ranks", "steps_averaged")}
                    line["ranks"].update(nat.pop("ranks"))
                    line["ranks"]["transport"] = nat.pop("transport")
                    promote_native(line, nat, ("value", "ms_per_step"))
                    line["torch_collectives"]["ranks"] = ranks_t
                else:
                    promote_native(line, nat, ())

            guard.note(take_over)
            del factors_t
        hw = None
        if not slab_mode and not a.no_hugewiki_leg:
            guard.pending = ("hugewiki",)
            close = getattr(eng, "close", None)
            if close is not None:
                close()
            del eng, r
            torch.cuda.empty_cache()
            hw = hugewiki_leg_guarded(a, als, datagen, dev, world, rank, backend, guard=guard)
        guard.emit(hw)
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001 -- the line is out; a failed teardown must not turn the run into an error
            pass
        return 0
    if world == 1:
        # roofline leg: the same steps again with HIP events around each kernel launch
        als.set_kernel_timing(True)
        als.kernel_ms_since_reset()
        for _ in range(max(2, min(a.steps, 5))):
            step(True)
        torch.cuda.synchronize()
        als.set_kernel_timing(False)
        x_ms = [v[0] for v in item_ms[0::2]]
        t_ms = [v[0] for v in item_ms[1::2]]
        red_ms = [v[1] for v in item_ms]
        cg = a.solver == "cg"
        # rows of each side as this GPU sees them (slab mode: its X slab; the Theta side forms PARTIAL Grams of all
        # n columns over the slab's ratings and writes packed Grams instead of factors -- bytes_theta below)
        rows_x = r.m
        bx = alg_bytes(r.nnz, rows_x, f, cg)
        if slab_mode:
            bt = 4.0 * f * r.nnz + 8.0 * r.nnz + 4.0 * (n + 1) + 4.0 * n * (f * (f + 1) // 2 + f)
        else:
            bt = alg_bytes(r.nnz, n, f, cg)
        avg_ms = (sum(x_ms) + sum(t_ms)) / (len(x_ms) + len(t_ms))
        avg_bytes = 0.5 * (bx + bt)
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        mode = als.get_gram_mode()
        nb = f // 16 + 1
        kernel = kernels.get("x") or als.last_kernel_name()
        wave = "als_wave" in kernel
        traffic, traffic_note = measured_traffic(kernel)
        headline = a.shape == "netflix" and f == 100 and a.solver == "lu" and a.scale == 1.0 and mode == "auto"
        if traffic is None and headline and not a.allow_missing_traffic:
            # never fatal (round 6): a kernel instance newer than the committed PMC passes prints traffic: null and says why
            print("bench.py: " + traffic_note + "; re-collect with tools/collect_profiles.sh", file=sys.stderr)
        traffic = traffic or {}
        # matrix-pipe work ISSUED per rating: upper-triangular 16x16 tiles x 2*16*16 flops, x6 bf16
        # products on the split path (als_wave.hip), x3 f16 products in the opt-in fast mode, x1 on the
        # fp32 MFMA path
        products = (3 if mode == "fast" else 6) if wave else 1
        issued = nb * (nb + 1) / 2 * 512.0 * products
        split_one_wave = wave and mode in ("auto", "split") and nb <= 7  # kArithSplit3 / kArithPre on the one-wave kernels
        if split_one_wave:
            # round 5: the diagonal tiles of the one-wave kernels take four products (D + 2 S, restored per item): 2 nb fewer MFMAs
            issued -= 2 * nb * 512.0
        pipe_peak = 2500.0 if wave else MFMA_F32_PEAK_TFLOPS
        nnz_gpu = r.nnz

        def issued_by(name):
            # round 6: the instances with a PACKED last block (4th template argument 3 = kArithPrePk, 4 = kArithSplitPk) multiply
            # its column by three products instead of six (one instead of four on its diagonal tile): 3 nb fewer MFMAs
            args = (name or "").split("<")[-1].split(",")
            packed = "als_wave_kernel" in (name or "") and len(args) >= 4 and args[3].strip() in ("3", "4")
            return issued - (3 * nb * 512.0 if packed else 0.0)

        def side(ms, nbytes, key, name):
            ach = nbytes / (ms * 1e-3) / 1e9
            return {"kernel": name, "ms": ms, "alg_bytes": nbytes, "achieved": ach, "frac": ach / HBM_PEAK_GBS,
                    "traffic": (traffic.get(key) or {}).get("bytes_per_launch"),
                    "gram_tflops_useful": float(nnz_gpu) * f * (f + 1) / (ms * 1e-3) / 1e12,
                    "matrix_pipe_frac_issued": float(nnz_gpu) * issued_by(name) / (ms * 1e-3) / 1e12 / pipe_peak}

        xs, ts = sum(x_ms) / len(x_ms), sum(t_ms) / len(t_ms)
        gram_only = None
        if wave and not slab_mode and not a.no_gram_leg:
            gram_only = gram_pass_alone(a)
        out["dtype"] = ("f32" if not wave else
                        "f32 (opt-in fast mode: pre-split f16x2 operands, 3 products, 22-bit significand, fp32 accumulate)"
                        if mode == "fast" else "f32 (bf16x3-split products on the bf16 matrix pipe, fp32 accumulate)")
        sx_, st_ = side(xs, bx, "x_side", kernels.get("x")), side(ts, bt, "theta_side", kernels.get("theta"))
        dom_name, dom = ("theta_side", st_) if ts >= xs else ("x_side", sx_)
        out["roofline"] = {
            # the DOMINANT kernel = the launch (symbol) that takes the larger part of the step: its algorithmic bytes over
            # its own HIP-event duration (not the step mean under the other side's name)
            "bound": "hbm", "kernel": dom["kernel"], "dominant": dom_name,
            "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
            "traffic": dom["traffic"], "traffic_source": traffic.get("source") or traffic_note,
            "alg_bytes_per_launch": dom["alg_bytes"], "avg_launch_ms": dom["ms"],
            "step_mean": {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "alg_bytes_per_launch": avg_bytes,
                          "avg_launch_ms": avg_ms, "traffic": traffic.get("bytes_per_launch")},
            "x_side_ms": xs, "theta_side_ms": ts,
            "x_side": sx_,
            "theta_side": st_,
            "reduce_kernel_ms_x_side": sum(red_ms[0::2]) / len(red_ms[0::2]),
            "reduce_kernel_ms_theta_side": sum(red_ms[1::2]) / len(red_ms[1::2]),
            "gram_mode": mode,
            "gram_pass_alone": gram_only,
            "gram_flops_per_launch": float(nnz_gpu) * f * (f + 1),
            "gram_tflops": float(nnz_gpu) * f * (f + 1) / (avg_ms * 1e-3) / 1e12,
            # the matrix pipe next to the HBM roof: flops issued (tile padding and, on the split path,
            # the six bf16 products per fp32 product included) against the pipe's dense peak
            "mfma": {"bound": "mfma", "pipe": (f"{'f16' if mode == 'fast' else 'bf16'} ({products} products per fp32 product"
                                                + ("; 4 on the diagonal tiles)" if split_one_wave else ")")
                                                if wave else "fp32"),
                     "achieved": float(nnz_gpu) * issued / (avg_ms * 1e-3) / 1e12, "peak": pipe_peak, "unit": "TFLOP/s",
                     "frac": float(nnz_gpu) * issued / (avg_ms * 1e-3) / 1e12 / pipe_peak},
            # what a perfect kernel of this design would take: the larger of the HBM time of the
            # algorithmic bytes and the matrix-pipe time of the issued flops
            "floor_ms": {"hbm": avg_bytes / (HBM_PEAK_GBS * 1e9) * 1e3,
                         "matrix_pipe": float(nnz_gpu) * issued / (pipe_peak * 1e12) * 1e3},
        }
        if f >= 128:
            # SURVEY.md 8(d), "which roofline": HBM for f <= 64, HBM (north_star) at the co-limited f = 100, the fp32 MFMA
            # roof above -- 157.3 TFLOP/s on the USEFUL Gram flops nnz f (f + 1) (25.3 ms per Netflix half-iteration at
            # f = 200 against 9.9 ms of HBM time).  The headline fraction of these lines is taken against the roof that
            # binds; the HBM figures stay as the secondary block.
            rl = out["roofline"]
            gram_flops = float(nnz_gpu) * f * (f + 1)
            ach = gram_flops / (dom["ms"] * 1e-3) / 1e12
            rl["hbm"] = {"bound": "hbm", "achieved": rl["achieved"], "peak": rl["peak"], "unit": rl["unit"], "frac": rl["frac"]}
            rl.update({"bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": ach / MFMA_F32_PEAK_TFLOPS,
                       "roof": "fp32 MFMA roof of SURVEY.md 8(d) on the useful (symmetric) Gram flops nnz f (f + 1); the kernel "
                               "itself runs them as bf16x3-split products on the bf16 pipe (block `mfma`)"})
            for sd in (rl["x_side"], rl["theta_side"]):
                sd["frac_fp32_mfma_roof"] = sd["gram_tflops_useful"] / MFMA_F32_PEAK_TFLOPS
            rl["floor_ms"]["fp32_mfma_roof"] = gram_flops / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3
        if slab_mode:
            out["roofline"]["note"] = ("per-GPU slab: the X side is the fused Gram+solve kernel over the slab's rows; the "
                                       "Theta side is the partial-Gram kernel of all n columns over the slab's ratings "
                                       "(packed Grams written, then reduce-scatter + solve + all-gather outside it)")
    if world == 1 and not slab_mode:
        tr, te = eng.rmse()
        out["rmse"] = {"train": tr, "test": te, "after_iterations": a.warmup + a.steps + len(x_ms)}
        if mode == "auto" and wave and not a.no_fast_leg:
            out["gram_fast_mode"] = mode_leg("fast", r, f, lam, a, theta0, eng)
            out["gram_exact_mode"] = mode_leg("exact", r, f, lam, a, theta0, eng)
        if not a.no_cpu_baseline:
            d = {k: v for k, v in r.numpy().items() if k.startswith("cs")}
            oracle_out, out["cpu_baseline"] = cpu_baseline(d, f, lam, a.solver)
            del eng
            out["parity_at_scale"] = parity_at_scale(r, f, lam, a.solver, a.cg_iters, oracle_out, dev)
            if a.shape == "netflix" and a.scale == 1.0 and not a.no_rmse_log:
                out["parity_at_scale"]["rmse_log"] = rmse_log_parity(r, f, lam, a.solver, a.cg_iters)
    print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
