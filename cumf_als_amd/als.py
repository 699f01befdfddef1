"""Host-side mirror of the reference's operator surface, over the C ABI of libALS.so.

* `do_als(...)` has the argument list and outputs of the reference's TensorFlow op
  `DoAls` (`tensorflow/als_tf.cc:7-30,132-136`): numpy host arrays in, (thetaT, XT,
  rmse) out.  It forwards to `cumf_doALS_ex` exactly as the TF op forwards to `doALS`.
* `Plan`, `update_fused`, `get_hermitian`, `cg_solve`, `lu_solve`, `sse` are the
  device-pointer entry points (torch CUDA tensors are used only as device memory).
* `ALSEngine` keeps one dataset resident in HBM and steps half-iterations; it is what
  bench.py times and what the multi-GPU driver (`cumf_als_amd.dist`) builds on.

There is no CPU path here: every call lands in a HIP kernel of libALS.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _libmod

SOLVER_CG, SOLVER_LU = 0, 1
CUMF_ERR_FAST_RANGE = 10001  # include/cumf_als_capi.h


def _solver_id(solver) -> int:
    if solver in (SOLVER_CG, "cg", "CG"):
        return SOLVER_CG
    if solver in (SOLVER_LU, "lu", "LU"):
        return SOLVER_LU
    raise ValueError(f"unknown solver {solver!r}")


def _hostptr(a: np.ndarray, dtype) -> C.c_void_p:
    if a.dtype != dtype or not a.flags["C_CONTIGUOUS"]:
        raise TypeError(f"expected C-contiguous {np.dtype(dtype).name} array")
    return a.ctypes.data_as(C.c_void_p)


def do_als(csrrow, csrcol, csrval, cscrow, csccol, cscval, coorow, coorowtest, coocoltest, coovaltest,
           m, n, f, nnz, nnz_test, lambda_, iters, xbatch, thetabatch, deviceid=0, *,
           thetat_init=None, xt_init=None, solver="cg", cg_iters=6, fused=True,
           exact_test_grid=False, surpass_nan=False, quiet=True, return_log=False, tt_fp16=None):
    """`DoAls` (als_tf.cc): run `iters` ALS iterations on device `deviceid`.

    Argument names follow the TF op's inputs: csrrow = CSR indptr (m+1), csrcol = CSR
    indices, cscrow = CSC row ids (nnz), csccol = CSC indptr (n+1), coorow = row of each
    CSR entry.  Returns (thetaT[n,f], XT[m,f], rmse) (+ rmse_log[iters,2] if asked).

    Initial factors default to the CLI's initialisation (main.cpp:72-78 evaluated with
    numpy's generator is NOT the same stream as libc rand(); pass `thetat_init` to
    reproduce a particular start).  tt_fp16: True / False select the fp16 Gram storage of the CG solver
    (CUMF_TT_FP16, als.cu:25-33) for this call; None (default) leaves the process-wide setting
    (`cumf_set_tt_fp16` / environment CUMF_ALS_TT_FP16) alone.  Raises RuntimeError when the opt-in gram
    mode "fast" meets data outside its range (the C entry point returns NaN and sets cumf_last_error).
    """
    lib = _libmod.load()
    if thetat_init is None:
        rng = np.random.RandomState(0)
        thetat = (0.2 * rng.random_sample((n, f))).astype(np.float32)
    else:
        thetat = np.array(thetat_init, dtype=np.float32, order="C", copy=True).reshape(n, f)
    xt = (np.zeros((m, f), np.float32) if xt_init is None
          else np.array(xt_init, dtype=np.float32, order="C", copy=True).reshape(m, f))
    log = np.zeros((max(iters, 1), 2), np.float32)
    csrrow = np.ascontiguousarray(csrrow, np.int32)
    csccol = np.ascontiguousarray(csccol, np.int32)
    if len(csrrow) != m + 1 or len(csccol) != n + 1:
        raise ValueError("csrrow must hold m+1 and csccol n+1 row pointers")
    args = [
        _hostptr(csrrow, np.int32), _hostptr(np.ascontiguousarray(csrcol, np.int32), np.int32),
        _hostptr(np.ascontiguousarray(csrval, np.float32), np.float32),
        _hostptr(np.ascontiguousarray(cscrow, np.int32), np.int32), _hostptr(csccol, np.int32),
        _hostptr(np.ascontiguousarray(cscval, np.float32), np.float32),
        _hostptr(np.ascontiguousarray(coorow, np.int32), np.int32),
        _hostptr(thetat, np.float32), _hostptr(xt, np.float32),
        _hostptr(np.ascontiguousarray(coorowtest, np.int32), np.int32),
        _hostptr(np.ascontiguousarray(coocoltest, np.int32), np.int32),
        _hostptr(np.ascontiguousarray(coovaltest, np.float32), np.float32),
    ]
    prev_fp16 = lib.cumf_get_tt_fp16()  # (resolves the environment default on first use)
    if tt_fp16 is not None:
        lib.cumf_set_tt_fp16(int(bool(tt_fp16)))  # CUMF_TT_FP16 (als.cu:25-33): fp16 Gram storage for the CG solver
    try:
        lib.cumf_last_error()
        rmse = lib.cumf_doALS_ex(*args, int(m), int(n), int(f), int(nnz), int(nnz_test), float(lambda_),
                                 int(iters), int(xbatch), int(thetabatch), int(deviceid),
                                 _solver_id(solver), int(cg_iters), int(bool(fused)), int(bool(exact_test_grid)),
                                 int(bool(surpass_nan)), int(bool(quiet)), _hostptr(log, np.float32))
        err = lib.cumf_last_error()
    finally:
        lib.cumf_set_tt_fp16(prev_fp16)
    if err == CUMF_ERR_FAST_RANGE:
        raise RuntimeError("gram mode 'fast': a factor or a rating beyond the f16 range of the pre-split operands "
                           "(|value| >= 15.99): use the default gram mode")
    if err:
        raise RuntimeError(f"cumf_doALS_ex failed with error {err}")
    if return_log:
        return thetat, xt, float(rmse), log[:iters]
    return thetat, xt, float(rmse)


# ---------------------------------------------------------------------------------------
# device-pointer entry points (torch tensors as device memory)
# ---------------------------------------------------------------------------------------

def _dp(t, dtype=None):
    import torch

    if t is None:
        return None
    if not t.is_cuda or not t.is_contiguous():
        raise TypeError("expected a contiguous CUDA tensor")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def _stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Plan:
    """Work decomposition of one side (cumf_plan_create).  `rowptr` may be a numpy array
    or a tensor (copied to host); int32 or int64."""

    def __init__(self, rowptr, f: int, row_begin: int = 0, row_end: int | None = None, chunk: int = 0):
        lib = _libmod.load()
        if hasattr(rowptr, "detach"):
            rowptr = rowptr.detach().cpu().numpy()
        rowptr = np.ascontiguousarray(rowptr)
        if rowptr.dtype == np.int64:
            is64 = 1
        elif rowptr.dtype == np.int32:
            is64 = 0
        else:
            raise TypeError("rowptr must be int32 or int64")
        self.rows = len(rowptr) - 1
        self.row_begin = row_begin
        self.row_end = self.rows if row_end is None else row_end
        self.f = f
        self._h = C.c_void_p()
        _libmod.check(lib.cumf_plan_create(C.byref(self._h), rowptr.ctypes.data_as(C.c_void_p), is64, self.rows,
                                           self.row_begin, self.row_end, f, chunk), "cumf_plan_create")
        info = (C.c_long * 4)()
        _libmod.check(lib.cumf_plan_info(self._h, info), "cumf_plan_info")
        self.n_items, self.n_slots, self.n_multi_rows, self.chunk = (int(v) for v in info)

    @property
    def batch_rows(self) -> int:
        return self.row_end - self.row_begin

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _libmod.load().cumf_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def update_fused(plan: Plan, colidx, val, gather, update, lambda_: float, solver="cg", cg_iters: int = 6):
    """One fused half-iteration over the plan's rows (cumf_als_update_fused)."""
    import torch

    lib = _libmod.load()
    _libmod.check(lib.cumf_check_gather_table(gather.shape[0], plan.f, _solver_id(solver), 0), "cumf_check_gather_table")
    _libmod.check(lib.cumf_plan_set_gather_rows(plan._h, int(gather.shape[0])), "cumf_plan_set_gather_rows")
    _libmod.check(lib.cumf_als_update_fused(plan._h, _dp(colidx, torch.int32), _dp(val, torch.float32),
                                            _dp(gather, torch.float32), _dp(update, torch.float32), plan.f,
                                            float(lambda_), _solver_id(solver), int(cg_iters), _stream()),
                  "cumf_als_update_fused")
    return update


SSE_BINS = 1024  # CUMF_SSE_BINS


def fused_sse_available(plan: Plan, solver="cg") -> bool:
    return bool(_libmod.load().cumf_fused_sse_available(plan._h, _solver_id(solver)))


def update_fused_sse(plan: Plan, colidx, val, gather, update, lambda_: float, solver="cg", cg_iters: int = 6, bins=None):
    """`update_fused` + the train SSE of the plan's rows for free (cumf_als_update_fused_sse): returns the fp64 bins
    tensor [SSE_BINS] it ADDED to (zeroed here when not passed in); the SSE is `bins.sum()`."""
    import torch

    lib = _libmod.load()
    if bins is None:
        bins = torch.zeros(SSE_BINS, dtype=torch.float64, device=update.device)
    _libmod.check(lib.cumf_check_gather_table(gather.shape[0], plan.f, _solver_id(solver), 0), "cumf_check_gather_table")
    _libmod.check(lib.cumf_plan_set_gather_rows(plan._h, int(gather.shape[0])), "cumf_plan_set_gather_rows")
    _libmod.check(lib.cumf_als_update_fused_sse(plan._h, _dp(colidx, torch.int32), _dp(val, torch.float32),
                                                _dp(gather, torch.float32), _dp(update, torch.float32), plan.f,
                                                float(lambda_), _solver_id(solver), int(cg_iters),
                                                _dp(bins, torch.float64), _stream()), "cumf_als_update_fused_sse")
    return bins


def quadratic_sse_terms(A, b, x, reg, out=None):
    """sum over the batch of 2 x.b - x^T A x + reg |x|^2 as a 1-element fp64 tensor (ADDED to `out` when given):
    cumf_quadratic_sse_terms -- the train SSE of materialised systems is (sum r^2 of their ratings) minus it."""
    import torch

    lib = _libmod.load()
    f = b.shape[-1]
    batch = b.numel() // f
    if out is None:
        out = torch.zeros(1, dtype=torch.float64, device=b.device)
    _libmod.check(lib.cumf_quadratic_sse_terms(_dp(A, torch.float32), _dp(b, torch.float32), _dp(x, torch.float32),
                                               _dp(reg, torch.float32), batch, f, _dp(out, torch.float64), _stream()),
                  "cumf_quadratic_sse_terms")
    return out


def get_hermitian(plan: Plan, colidx, val, gather, lambda_: float, tt=None, rhs=None, want_rhs=True, half=False):
    """Materialise the Gram batch tt[rows,f,f] (+ rhs[rows,f]) of the plan's rows (cumf_get_hermitian).
    half=True (or a float16 `tt`): fp16 storage of the Gram, cumf_get_hermitian_fp16 (als.cu:335-441)."""
    import torch

    lib = _libmod.load()
    f, rows = plan.f, plan.batch_rows
    _libmod.check(lib.cumf_check_gather_table(gather.shape[0], f, SOLVER_LU, 1), "cumf_check_gather_table")
    half = half or (tt is not None and tt.dtype == torch.float16)
    if tt is None:
        tt = torch.empty((rows, f, f), dtype=torch.float16 if half else torch.float32, device=gather.device)
    if rhs is None and want_rhs:
        rhs = torch.empty((rows, f), dtype=torch.float32, device=gather.device)
    if half:
        _libmod.check(lib.cumf_get_hermitian_fp16(plan._h, _dp(colidx, torch.int32), _dp(val, torch.float32),
                                                  _dp(gather, torch.float32), _dp(tt, torch.float16),
                                                  _dp(rhs, torch.float32), f, float(lambda_), _stream()),
                      "cumf_get_hermitian_fp16")
        return tt, rhs
    _libmod.check(lib.cumf_get_hermitian(plan._h, _dp(colidx, torch.int32), _dp(val, torch.float32),
                                         _dp(gather, torch.float32), _dp(tt, torch.float32),
                                         _dp(rhs, torch.float32), f, float(lambda_), _stream()),
                  "cumf_get_hermitian")
    return tt, rhs


def get_hermitian_packed(plan: Plan, colidx, val, gather, lambda_: float, packed=None, rhs=None, want_rhs=True):
    """The Gram batch of the plan's rows as packed upper triangles packed[rows, f(f+1)/2] (+ rhs[rows,f]),
    written straight from the accumulators (cumf_get_hermitian_packed): the multi-GPU reduction payload."""
    import torch

    lib = _libmod.load()
    f, rows = plan.f, plan.batch_rows
    _libmod.check(lib.cumf_check_gather_table(gather.shape[0], f, SOLVER_LU, 1), "cumf_check_gather_table")
    if packed is None:
        packed = torch.empty((rows, f * (f + 1) // 2), dtype=torch.float32, device=gather.device)
    if rhs is None and want_rhs:
        rhs = torch.empty((rows, f), dtype=torch.float32, device=gather.device)
    _libmod.check(lib.cumf_get_hermitian_packed(plan._h, _dp(colidx, torch.int32), _dp(val, torch.float32),
                                                _dp(gather, torch.float32), _dp(packed, torch.float32),
                                                _dp(rhs, torch.float32), f, float(lambda_), _stream()),
                  "cumf_get_hermitian_packed")
    return packed, rhs


def cg_solve(A, x, b, cg_iters: int = 6):
    """Batched CG, x is the warm start and is overwritten (updateXWithCGHost, cg.h:30)."""
    import torch

    lib = _libmod.load()
    f = b.shape[-1]
    batch = b.numel() // f
    if A.dtype == torch.float16:  # updateXWithCGHost_tt_fp16 (cg.h:32)
        _libmod.check(lib.cumf_cg_solve_batched_fp16(_dp(A, torch.float16), _dp(x, torch.float32),
                                                     _dp(b, torch.float32), batch, f, int(cg_iters), _stream()),
                      "cumf_cg_solve_batched_fp16")
        return x
    _libmod.check(lib.cumf_cg_solve_batched(_dp(A, torch.float32), _dp(x, torch.float32), _dp(b, torch.float32),
                                            batch, f, int(cg_iters), _stream()), "cumf_cg_solve_batched")
    return x


def lu_solve(A, b, x=None):
    """Batched unpivoted LU solve (cublasSgetrfBatched + SgetrsBatched, als.cu:77,98)."""
    import torch

    lib = _libmod.load()
    f = b.shape[-1]
    batch = b.numel() // f
    if x is None:
        x = torch.empty_like(b)
    _libmod.check(lib.cumf_lu_solve_batched(_dp(A, torch.float32), _dp(b, torch.float32), _dp(x, torch.float32),
                                            batch, f, _stream()), "cumf_lu_solve_batched")
    return x


def pack_upper(full, packed=None):
    """full[batch, f, f] (symmetric) -> packed[batch, f(f+1)/2] upper triangles (cumf_pack_upper)."""
    import torch

    lib = _libmod.load()
    batch, f = full.shape[0], full.shape[-1]
    if packed is None:
        packed = torch.empty((batch, f * (f + 1) // 2), dtype=torch.float32, device=full.device)
    _libmod.check(lib.cumf_pack_upper(_dp(full, torch.float32), _dp(packed, torch.float32), batch, f, _stream()),
                  "cumf_pack_upper")
    return packed


def unpack_upper(packed, full):
    """packed[batch, f(f+1)/2] -> full[batch, f, f], both triangles (cumf_unpack_upper)."""
    import torch

    lib = _libmod.load()
    batch, f = full.shape[0], full.shape[-1]
    _libmod.check(lib.cumf_unpack_upper(_dp(packed, torch.float32), _dp(full, torch.float32), batch, f, _stream()),
                  "cumf_unpack_upper")
    return full


def sse(val, row, col, thetaT, XT, count: int | None = None, surpass_nan: bool = False, out=None):
    """Sum of squared errors over the first `count` ratings -> 1-element fp64 tensor (cumf_sse)."""
    import torch

    lib = _libmod.load()
    f = thetaT.shape[-1]
    if count is None:
        count = val.numel()
    if out is None:
        out = torch.zeros(1, dtype=torch.float64, device=val.device)
    _libmod.check(lib.cumf_sse(_dp(val, torch.float32), _dp(row, torch.int32), _dp(col, torch.int32),
                               _dp(thetaT, torch.float32), _dp(XT, torch.float32), int(count), f,
                               int(bool(surpass_nan)), _dp(out, torch.float64), _stream()), "cumf_sse")
    return out


GRAM_AUTO, GRAM_EXACT, GRAM_FAST = 0, 1, 2


def set_gram_mode(mode) -> None:
    """Arithmetic of the Gram pass (cumf_set_gram_mode): "auto"/"split" = fp32 via exact bf16x3
    splits on the bf16 matrix pipe where available, "exact" = fp32 MFMA (fmaf-chain bits), "fast"
    (opt-in) = pre-split f16 pairs, three products, 22 significand bits, |values| < 15.99."""
    m = {"auto": GRAM_AUTO, "split": GRAM_AUTO, "exact": GRAM_EXACT, "fast": GRAM_FAST}.get(mode, mode)
    _libmod.check(_libmod.load().cumf_set_gram_mode(int(m)), "cumf_set_gram_mode")


def get_gram_mode() -> str:
    return {GRAM_EXACT: "exact", GRAM_FAST: "fast"}.get(_libmod.load().cumf_get_gram_mode(), "auto")


PRESPLIT_AUTO, PRESPLIT_OFF, PRESPLIT_ON, PRESPLIT_VERIFY = -1, 0, 1, 2


def set_presplit(mode) -> None:
    """Pre-split gather tables (cumf_set_presplit): "auto" = where the planes of the table stay in the caches (default),
    "off" / "on" = never / whenever the shape allows, "verify" = on with the last feature block unpacked (bit-identical to
    "off"; the production form multiplies that block as one packed operand: same error class, other bits)."""
    m = {"auto": PRESPLIT_AUTO, "off": PRESPLIT_OFF, "on": PRESPLIT_ON, "verify": PRESPLIT_VERIFY}.get(mode, mode)
    _libmod.check(_libmod.load().cumf_set_presplit(int(m)), "cumf_set_presplit")


def get_presplit() -> str:
    return {PRESPLIT_OFF: "off", PRESPLIT_ON: "on", PRESPLIT_VERIFY: "verify"}.get(_libmod.load().cumf_get_presplit(), "auto")


def presplit_table(table: "torch.Tensor") -> "torch.Tensor":
    """The bf16 h | m | l planes of a rows x f fp32 table as the fused calls build them (cumf_presplit_table): a uint8
    tensor [rows, cumf_presplit_pitch(f)]."""
    import torch

    lib = _libmod.load()
    rows, f = int(table.shape[0]), int(table.shape[1])
    pitch = int(lib.cumf_presplit_pitch(f))
    if pitch == 0:
        raise ValueError(f"no pre-split kernels for f = {f}")
    out = torch.empty((rows, pitch), dtype=torch.uint8, device=table.device)
    _libmod.check(lib.cumf_presplit_table(_dp(table, torch.float32), out.data_ptr(), rows, f, _stream()), "cumf_presplit_table")
    return out


def gram_fast_status() -> int:
    """Range report of gram mode "fast" since the last call (waits for the device; cumf_gram_fast_status):
    bit 0 = a factor, bit 1 = a rating beyond the f16 range; 0 = clean."""
    import ctypes as C

    flags = C.c_int(0)
    _libmod.check(_libmod.load().cumf_gram_fast_status(C.byref(flags)), "cumf_gram_fast_status")
    return int(flags.value)


def check_gram_fast() -> None:
    """In gram mode "fast": raise if a factor or a rating left the f16 range since the last check (the
    affected rows are not finite); a no-op (no device wait) in every other mode."""
    if _libmod.load().cumf_get_gram_mode() != GRAM_FAST:
        return
    flags = gram_fast_status()
    if flags:
        what = " and ".join(w for b, w in ((1, "a factor"), (2, "a rating")) if flags & b)
        raise RuntimeError(f"gram mode 'fast': {what} beyond the f16 range of the pre-split operands "
                           "(|value| >= 15.99): use the default gram mode")


def set_debug_switches(switches: int) -> None:
    """Ablation switches (1 = no solve: the Gram pass alone, ...; the results are wrong on purpose).  They
    exist only in the profiling build libALS_ablate.so (`CUMF_ALS_LIB=.../libALS_ablate.so`, used by
    tools/gram_pass_alone.py); the product library has no such entry point and this raises."""
    lib = _libmod.load()
    if not hasattr(lib, "cumf_set_debug_switches"):
        raise RuntimeError("ablation switches exist only in libALS_ablate.so (load it through CUMF_ALS_LIB)")
    _libmod.check(lib.cumf_set_debug_switches(int(switches)), "cumf_set_debug_switches")


def debug_cg_histogram(f: int):
    """Profiling build only, after running with switch 65536: rows by the number of CG iterations they ran before
    ||r||^2 < 1e-4 ended the loop (cg.cu:195); read and cleared."""
    lib = _libmod.load()
    if not hasattr(lib, "cumf_debug_cg_histogram"):
        raise RuntimeError("the CG histogram exists only in libALS_ablate.so (load it through CUMF_ALS_LIB)")
    out = (C.c_ulonglong * 16)()
    _libmod.check(lib.cumf_debug_cg_histogram(int(f), out), "cumf_debug_cg_histogram")
    return [int(v) for v in out]


def last_kernel_name() -> str:
    """Name of the Gram(+solve) kernel the last half-iteration dispatched, as rocprofv3 prints it."""
    buf = C.create_string_buffer(256)
    _libmod.check(_libmod.load().cumf_last_kernel_name(buf, 256), "cumf_last_kernel_name")
    return buf.value.decode()


def set_kernel_timing(enable: bool) -> None:
    _libmod.check(_libmod.load().cumf_set_kernel_timing(int(bool(enable))), "cumf_set_kernel_timing")


def last_kernel_ms():
    """(item_kernel_ms, reduce_kernel_ms) of the last half-iteration (HIP events on its stream)."""
    a, b = C.c_float(), C.c_float()
    _libmod.check(_libmod.load().cumf_last_kernel_ms(C.byref(a), C.byref(b)), "cumf_last_kernel_ms")
    return a.value, b.value


def kernel_ms_since_reset():
    """(item_kernel_ms, reduce_kernel_ms, launches) summed over every half-iteration launch sequence since the
    previous call (cumf_kernel_ms_since_reset): a half-iteration made of several launches is read with one call."""
    a, b, n = C.c_float(), C.c_float(), C.c_int()
    _libmod.check(_libmod.load().cumf_kernel_ms_since_reset(C.byref(a), C.byref(b), C.byref(n)),
                  "cumf_kernel_ms_since_reset")
    return a.value, b.value, n.value


def release_scratch() -> None:
    """Free the pooled scratch of the CURRENT device (tile buffers of the f >= 144 LU path, pre-split tables of gram
    mode "fast"): cumf_release_scratch.  ALSEngine.close() / DistALS.close() call it."""
    _libmod.check(_libmod.load().cumf_release_scratch(), "cumf_release_scratch")


class ALSEngine:
    """A dataset resident in HBM + the two half-iteration plans (single GPU).

    `r` is a `cumf_als_amd.datagen.Ratings` already on the device.  Factors are the
    reference's `thetaT` (n x f) and `XT` (m x f), row-contiguous f-vectors.
    """

    def __init__(self, r, f: int, lambda_: float, solver="cg", cg_iters: int = 6, x_batch: int = 1,
                 theta_batch: int = 1, fused: bool = True, chunk: int = 0):
        import torch

        self.r, self.f, self.lam = r, f, float(lambda_)
        self.solver, self.cg_iters = solver, cg_iters
        # above the tile kernels' range (f > 207) the reference's unfused data flow runs (cumf_get_hermitian + batched solver)
        self.fused = bool(fused) and bool(_libmod.load().cumf_fused_available(int(f), _solver_id(solver)))
        self.m, self.n = r.m, r.n
        self.device = r.csr_indices.device
        self.x_plans = self._plans(r.csr_indptr, r.m, x_batch, chunk)
        self.t_plans = self._plans(r.csc_indptr, r.n, theta_batch, chunk)
        self.thetaT = torch.zeros((r.n, f), dtype=torch.float32, device=self.device)
        self.XT = torch.zeros((r.m, f), dtype=torch.float32, device=self.device)
        self._tt = None
        self._rhs = None

    def _plans(self, rowptr, rows, nbatch, chunk):
        rp = rowptr.detach().cpu().numpy()
        plans = []
        for b in range(nbatch):  # als.cu:768-777
            size = rows // nbatch if b != nbatch - 1 else rows - b * (rows // nbatch)
            off = b * (rows // nbatch)
            plans.append(Plan(rp, self.f, off, off + size, chunk))
        return plans

    def init_factors(self, thetaT=None, XT=None, seed: int = 0):
        import torch

        if thetaT is None:
            g = torch.Generator(device="cpu")
            g.manual_seed(seed)
            thetaT = 0.2 * torch.rand((self.n, self.f), generator=g, dtype=torch.float32)
        self.thetaT.copy_(torch.as_tensor(thetaT).reshape(self.n, self.f))
        if XT is None:
            self.XT.zero_()
        else:
            self.XT.copy_(torch.as_tensor(XT).reshape(self.m, self.f))

    def _half(self, plans, colidx, val, gather, update):
        import torch

        for p in plans:
            if self.fused:
                update_fused(p, colidx, val, gather, update, self.lam, self.solver, self.cg_iters)
            else:
                rows = p.batch_rows
                if self._tt is None or self._tt.shape[0] < rows:
                    self._tt = torch.empty((rows, self.f, self.f), dtype=torch.float32, device=self.device)
                    self._rhs = torch.empty((rows, self.f), dtype=torch.float32, device=self.device)
                tt, rhs = self._tt[:rows], self._rhs[:rows]
                get_hermitian(p, colidx, val, gather, self.lam, tt, rhs)
                xb = update[p.row_begin:p.row_end]
                if _solver_id(self.solver) == SOLVER_CG:
                    cg_solve(tt, xb, rhs, self.cg_iters)
                else:
                    lu_solve(tt, rhs, xb)

    def update_x(self):
        """update X from thetaT over the CSR rows (als.cu:727-855)."""
        self._half(self.x_plans, self.r.csr_indices, self.r.csr_data, self.thetaT, self.XT)

    def update_theta(self):
        """update Theta from XT over the CSC columns (als.cu:857-964)."""
        self._half(self.t_plans, self.r.csc_indices, self.r.csc_data, self.XT, self.thetaT)

    def update_theta_with_train_sse(self):
        """update Theta AND return the train SSE of the new factors (fp64 scalar tensor) from the same kernels
        (cumf_als_update_fused_sse); None -- after a plain update -- when the plans cannot deliver it."""
        import torch

        if not (self.fused and all(fused_sse_available(p, self.solver) for p in self.t_plans)):
            self.update_theta()
            return None
        bins = torch.zeros(SSE_BINS, dtype=torch.float64, device=self.device)
        for p in self.t_plans:
            update_fused_sse(p, self.r.csc_indices, self.r.csc_data, self.XT, self.thetaT, self.lam, self.solver,
                             self.cg_iters, bins)
        return bins.sum()

    def rmse(self, exact_test_grid: bool = True, surpass_nan: bool = False):
        """(train, test) RMSE as als.cu:966-1020."""
        r = self.r
        tr = sse(r.csr_data, r.coo_row, r.csr_indices, self.thetaT, self.XT, r.nnz, surpass_nan)
        cnt = r.nnz_test if exact_test_grid else max(0, ((r.nnz_test - 1) // 256) * 256)
        te = sse(r.test_data, r.test_row, r.test_col, self.thetaT, self.XT, cnt, surpass_nan)
        return (float(tr.item() / max(r.nnz, 1)) ** 0.5, float(te.item() / max(r.nnz_test, 1)) ** 0.5)

    def iterate(self, iters: int = 1):
        for _ in range(iters):
            self.update_x()
            self.update_theta()
        check_gram_fast()

    def close(self) -> None:
        """Destroy the plans and hand the library's pooled scratch of this device back (up to 48 GiB of tile buffer
        at f >= 144 that lives outside torch's caching allocator)."""
        for p in self.x_plans + self.t_plans:
            p.close()
        self.x_plans, self.t_plans = [], []
        release_scratch()
