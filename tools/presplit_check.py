#!/usr/bin/env python3
"""Round 6: the pre-split gather table (kArithPre, als_wave.hip) against the in-kernel split.
  python tools/presplit_check.py            small shapes: planes vs a numpy restatement of the split; fused LU / CG
                                            half-iterations with the table pre-split and not -- must be BIT-IDENTICAL
  python tools/presplit_check.py --time     Netflix shape: kernel times of both half-iterations, alternating off / on
Prints one JSON line per case."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cumf_als_amd import als, datagen  # noqa: E402


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 (round to nearest even) as uint16 bit patterns, finite inputs."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return r.astype(np.uint16)


def bf16_val(h: np.ndarray) -> np.ndarray:
    return (h.astype(np.uint32) << 16).view(np.float32)


def split_planes(x: np.ndarray):
    h = bf16_rne(x)
    r1 = x - bf16_val(h)
    m = bf16_rne(r1)
    r2 = r1 - bf16_val(m)
    return h, m, bf16_rne(r2)


def check_planes(f: int, rows: int = 257) -> dict:
    rng = np.random.RandomState(f)
    t = (rng.standard_normal((rows, f)) * np.exp(rng.uniform(-8, 8, (rows, 1)))).astype(np.float32)
    t[3, :] = 0.0
    planes = als.presplit_table(torch.from_numpy(t).cuda()).cpu().numpy()
    fb, sf = f // 16, f % 16
    sw = 8 if f // 16 + 1 > 7 else 4   # halfwords per plane in the strip: the two-wave kernels' rows hold up to eight features
    halfs = planes.view(np.uint16).reshape(rows, -1)
    h, m, l = split_planes(t)
    ok = True
    for p, ref in enumerate((h, m, l)):
        ok &= bool(np.array_equal(halfs[:, 16 * fb * p: 16 * fb * (p + 1)], ref[:, :16 * fb]))
        if sf:
            ok &= bool(np.array_equal(halfs[:, 48 * fb + sw * p: 48 * fb + sw * p + sf], ref[:, 16 * fb:]))
            ok &= bool((halfs[:, 48 * fb + sw * p + sf: 48 * fb + sw * (p + 1)] == 0).all())
    if sf:
        ok &= bool((halfs[:, 48 * fb + 3 * sw: 48 * fb + 4 * sw] == 0).all())
    # the three terms add up to the value exactly
    s = (bf16_val(h).astype(np.float64) + bf16_val(m).astype(np.float64) + bf16_val(l).astype(np.float64))
    return {"case": "planes", "f": f, "rows": rows, "pitch": int(planes.shape[1]), "planes_equal_numpy_split": ok,
            "h_plus_m_plus_l_exact": bool(np.array_equal(s, t.astype(np.float64)))}


def check_fused(f: int, solver: str, seed: int = 0, modes=("off", "verify", "on")) -> dict:
    """Rows of every length class (0, 1, 31, 32, 33, 64, ~200, one chunked) through the fused call, once per mode.
    modes = ("off", "auto"): the six-product in-kernel split against what the library picks by itself -- at f % 16 == 0 on a
    table it does not pre-split, the in-kernel split with the packed rating block (kArithSplitPk)."""
    rng = np.random.RandomState(seed + f)
    n_cols = 3000
    lens = [0, 1, 2, 31, 32, 33, 63, 64, 65, 96, 100, 127, 128, 129, 200, 255, 256, 500, 9000] + list(rng.randint(1, 400, 150))
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(rowptr[-1])
    colidx = np.concatenate([rng.choice(n_cols, size=k, replace=False) if k <= n_cols else rng.randint(0, n_cols, k)
                             for k in lens]).astype(np.int32)
    val = rng.randint(1, 6, nnz).astype(np.float32) + rng.uniform(-0.3, 0.3, nnz).astype(np.float32)
    gather = (0.3 * rng.standard_normal((n_cols, f))).astype(np.float32)
    x0 = (0.1 * rng.standard_normal((len(lens), f))).astype(np.float32)
    plan = als.Plan(rowptr, f)
    ci, va, ga = (torch.from_numpy(v).cuda() for v in (colidx, val, gather))
    out = {}
    for mode in modes:
        als.set_presplit(mode)
        x = torch.from_numpy(x0.copy()).cuda()
        bins = als.update_fused_sse(plan, ci, va, ga, x, 0.05, solver, 6) if als.fused_sse_available(plan, solver) else None
        if bins is None:
            als.update_fused(plan, ci, va, ga, x, 0.05, solver, 6)
        torch.cuda.synchronize()
        out[mode] = (x.cpu().numpy(), None if bins is None else bins.cpu().numpy(), als.last_kernel_name())
    als.set_presplit("auto")
    first, last = modes[0], modes[-1]
    a, c = out[first][0], out[last][0]
    b = out["verify"][0] if "verify" in out else a
    same = bool(np.array_equal(a, b, equal_nan=True))
    fin = np.isfinite(a) & np.isfinite(c)
    sse = None
    if out[first][1] is not None:
        so, sp = float(out[first][1].sum()), float(out[last][1].sum())
        sse = abs(so - sp) / max(abs(so), 1e-30)
    res = {"kernel_" + m: out[m][2] for m in modes}
    res.update({"case": "fused", "f": f, "solver": solver, "rows": len(lens), "nnz": nnz, "chunked_rows": plan.n_multi_rows,
            # verification form (last block unpacked): the in-kernel split's bits
            "bit_identical": same if "verify" in out else None,
            "sse_bins_identical": None if (out[first][1] is None or "verify" not in out)
            else bool(np.array_equal(out[first][1], out["verify"][1])),
            "rows_differing": int((~np.all((a == b) | (np.isnan(a) & np.isnan(b)), axis=1)).sum()),
            # production form (last block packed): same error class, other bits in the last block column
            "packed_nan_pattern_equal": bool(np.array_equal(np.isnan(a), np.isnan(c))),
            "packed_max_rel_diff": float(np.abs(a[fin] - c[fin]).max() / np.abs(a[fin]).max()) if fin.any() else None,
            "packed_sse_rel_diff": sse})
    return res


def time_netflix(f: int, solver: str, reps: int) -> dict:
    shp = datagen.SHAPES["netflix"]
    r = datagen.synth_ratings(shp["m"], shp["n"], shp["nnz"], shp["nnz_test"], seed=0, device="cuda")
    eng = als.ALSEngine(r, f, shp["lam"], solver=solver)
    eng.init_factors()
    eng.iterate(1)
    torch.cuda.synchronize()
    keep_x, keep_t = eng.XT.clone(), eng.thetaT.clone()
    als.set_kernel_timing(True)
    res = {"off": ([], []), "verify": ([], []), "on": ([], [])}
    fac = {}
    for rep in range(reps + 1):
        for mode in ("off", "verify", "on"):
            als.set_presplit(mode)
            eng.update_x()
            tx = sum(als.last_kernel_ms())
            kx = als.last_kernel_name()
            fx = eng.XT.clone()
            eng.XT.copy_(keep_x)
            eng.update_theta()
            tt = sum(als.last_kernel_ms())
            kt = als.last_kernel_name()
            ft = eng.thetaT.clone()
            eng.thetaT.copy_(keep_t)
            if rep:
                res[mode][0].append(tx)
                res[mode][1].append(tt)
            fac[mode] = (fx, ft, kx, kt)
    als.set_presplit("auto")
    med = lambda v: round(sorted(v)[len(v) // 2], 3)
    return {"case": "time", "f": f, "solver": solver, "reps": reps,
            "x_ms": {m: med(res[m][0]) for m in res}, "theta_ms": {m: med(res[m][1]) for m in res},
            "theta_all": {m: [round(v, 3) for v in res[m][1]] for m in res},
            "kernels": {m: fac[m][2:] for m in fac},
            "x_bit_identical_verify": bool(torch.equal(fac["off"][0], fac["verify"][0])),
            "theta_bit_identical_verify": bool(torch.equal(fac["off"][1], fac["verify"][1])),
            "x_packed_max_rel": float((fac["off"][0] - fac["on"][0]).abs().max() / fac["off"][0].abs().max()),
            "theta_packed_max_rel": float((fac["off"][1] - fac["on"][1]).abs().max() / fac["off"][1].abs().max())}


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--f", type=int, nargs="*", default=None)
    ap.add_argument("--solver", nargs="*", default=["lu", "cg"])
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    bad = 0
    if a.time:
        for f in a.f or [100]:
            for s in a.solver:
                print(json.dumps(time_netflix(f, s, a.reps)), flush=True)
        return 0
    for f in a.f or [100, 64, 96, 68, 200, 128, 120, 180]:
        o = check_planes(f)
        bad += not (o["planes_equal_numpy_split"] and o["h_plus_m_plus_l_exact"])
        print(json.dumps(o), flush=True)
        for s in a.solver:
            o = check_fused(f, s)
            bad += not (o["bit_identical"] and o["packed_nan_pattern_equal"] and o["packed_max_rel_diff"] < 2e-4)
            print(json.dumps(o), flush=True)
    print("PRESPLIT CHECK", "FAILED" if bad else "OK")
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
