"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may
import this module (see the header of oracle/als_oracle.c).  The product
(`cumf_als_amd`, libALS.so) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile liboracle.so (and oracle/_ref when /root/reference is mounted)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, n) for n in ("als_oracle.c", "als_oracle_impl.h", "Makefile")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-s", "-C", _HERE], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        for suf, rp in (("_f32", _f32p), ("_f64", _f64p)):
            fn = getattr(_LIB, "oracle_gram_rhs" + suf)
            fn.restype = None
            fn.argtypes = [_i32p, _i32p, _f32p, _f32p, C.c_int, C.c_float, C.c_long, C.c_long,
                           C.c_void_p, C.c_void_p]
            fn = getattr(_LIB, "oracle_cg" + suf)
            fn.restype = None
            fn.argtypes = [rp, rp, rp, C.c_long, C.c_int, C.c_float]
            fn = getattr(_LIB, "oracle_lu" + suf)
            fn.restype = None
            fn.argtypes = [rp, rp, C.c_long, C.c_int]
            fn = getattr(_LIB, "oracle_sse" + suf)
            fn.restype = C.c_double
            fn.argtypes = [_f32p, _i32p, _i32p, rp, rp, C.c_long, C.c_int, C.c_int]
            fn = getattr(_LIB, "oracle_half_iteration" + suf)
            fn.restype = None
            fn.argtypes = [_i32p, _i32p, _f32p, _f32p, _f32p, C.c_long, C.c_int, C.c_float,
                           C.c_int, C.c_int, C.c_int]
            fn = getattr(_LIB, "oracle_doALS" + suf)
            fn.restype = C.c_float
            fn.argtypes = [_i32p, _i32p, _f32p, _i32p, _i32p, _f32p, _i32p, _f32p, _f32p,
                           _i32p, _i32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_long,
                           C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.c_void_p]
        _LIB.oracle_init_factors.restype = None
        _LIB.oracle_init_factors.argtypes = [_f32p, C.c_long, _f32p, C.c_long]
        _LIB.oracle_time_half_iteration_f32.restype = C.c_double
        _LIB.oracle_time_half_iteration_f32.argtypes = [_i32p, _i32p, _f32p, _f32p, _f32p, C.c_long,
                                                        C.c_int, C.c_float, C.c_int, C.c_int]
        _LIB.oracle_num_threads.restype = C.c_int
        _LIB.oracle_set_num_threads.argtypes = [C.c_int]
    return _LIB


def _suf(dtype):
    return "_f64" if np.dtype(dtype) == np.float64 else "_f32"


def gram_rhs(rowptr, colidx, val, factors, f, lam, row_begin=0, row_end=None, dtype=np.float32,
             want_gram=True, want_rhs=True):
    """Return (tt[rows,f,f], b[rows,f]) for rows [row_begin,row_end)."""
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    row_end = len(rowptr) - 1 if row_end is None else row_end
    rows = row_end - row_begin
    tt = np.zeros((rows, f, f), dtype) if want_gram else None
    b = np.zeros((rows, f), dtype) if want_rhs else None
    getattr(lib(), "oracle_gram_rhs" + _suf(dtype))(
        rowptr, np.ascontiguousarray(colidx, np.int32), np.ascontiguousarray(val, np.float32),
        np.ascontiguousarray(factors, np.float32), f, lam, row_begin, row_end,
        tt.ctypes.data if tt is not None else None, b.ctypes.data if b is not None else None)
    return tt, b


def cg(A, x, b, f, cg_iters):
    """Batched CG; returns the updated x (input x is the warm start, not modified)."""
    dtype = A.dtype
    A = np.ascontiguousarray(A)
    x = np.array(x, dtype=dtype, order="C", copy=True)
    b = np.ascontiguousarray(b, dtype)
    batch = x.size // f
    getattr(lib(), "oracle_cg" + _suf(dtype))(A.reshape(-1), x.reshape(-1), b.reshape(-1), batch, f,
                                              float(cg_iters))
    return x


def lu(A, b, f):
    """Batched unpivoted LU solve; returns x (A and b are not modified)."""
    dtype = A.dtype
    A = np.array(A, dtype=dtype, order="C", copy=True)
    b = np.array(b, dtype=dtype, order="C", copy=True)
    batch = b.size // f
    getattr(lib(), "oracle_lu" + _suf(dtype))(A.reshape(-1), b.reshape(-1), batch, f)
    return b


def sse(val, row, col, thetaT, XT, count, f, surpass_nan=False, dtype=np.float32):
    return getattr(lib(), "oracle_sse" + _suf(dtype))(
        np.ascontiguousarray(val, np.float32), np.ascontiguousarray(row, np.int32),
        np.ascontiguousarray(col, np.int32), np.ascontiguousarray(thetaT, dtype).reshape(-1),
        np.ascontiguousarray(XT, dtype).reshape(-1), count, f, int(surpass_nan))


def half_iteration(rowptr, colidx, val, gather, update, f, lam, nbatch=1, solver="cg", cg_iters=6,
                   dtype=np.float32):
    """Update `update` (rows x f, fp32) in place from `gather`; returns it."""
    rows = len(rowptr) - 1
    getattr(lib(), "oracle_half_iteration" + _suf(dtype))(
        np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(colidx, np.int32),
        np.ascontiguousarray(val, np.float32), np.ascontiguousarray(gather, np.float32).reshape(-1),
        update.reshape(-1), rows, f, lam, nbatch, 0 if solver == "cg" else 1, cg_iters)
    return update


def do_als(d, thetaT, XT, m, n, f, lam, iters, x_batch=1, theta_batch=1, solver="cg", cg_iters=6,
           test_grid_compat=True, surpass_nan=False, dtype=np.float32):
    """oracle_doALS over a dataset dict (keys of cumf_als_amd.datagen.FILES).

    thetaT / XT are updated in place.  Returns (final_test_rmse, rmse_log[iters,2]).
    """
    log = np.zeros((iters, 2), np.float64)
    nnz, nnz_test = len(d["csr_indices"]), len(d["test_row"])
    rmse = getattr(lib(), "oracle_doALS" + _suf(dtype))(
        d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indices"], d["csc_indptr"],
        d["csc_data"], d["coo_row"], thetaT.reshape(-1), XT.reshape(-1), d["test_row"],
        d["test_col"], d["test_data"], m, n, f, nnz, nnz_test, lam, iters, x_batch, theta_batch,
        0 if solver == "cg" else 1, cg_iters, int(test_grid_compat), int(surpass_nan),
        log.ctypes.data)
    return float(rmse), log


def init_factors(m, n, f):
    thetaT = np.zeros(n * f, np.float32)
    XT = np.zeros(m * f, np.float32)
    lib().oracle_init_factors(thetaT, n * f, XT, m * f)
    return thetaT.reshape(n, f), XT.reshape(m, f)


def time_half_iteration(rowptr, colidx, val, gather, update, f, lam, solver="lu", cg_iters=6):
    rows = len(rowptr) - 1
    return lib().oracle_time_half_iteration_f32(
        np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(colidx, np.int32),
        np.ascontiguousarray(val, np.float32), np.ascontiguousarray(gather, np.float32).reshape(-1),
        update.reshape(-1), rows, f, lam, 0 if solver == "cg" else 1, cg_iters)


def num_threads() -> int:
    return lib().oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(n)
