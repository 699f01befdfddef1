"""Multi-GPU ALS: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

Replaces the reference's 4-GPU program `hugewiki/hugewiki.cu` (OpenMP thread per GPU,
`cudaMemcpy` peer copies into a staging buffer + `cublasSaxpy` on GPU 0, serial over
GPUs: hugewiki.cu:2703-2730; LU on GPU 0 only; three broadcasts of Theta:
hugewiki.cu:2744-2745; X round-tripping through host memory: hugewiki.cu:2571,2641).

Two partitionings, both with contiguous nnz-balanced row slabs fixed for the run:

* ``"gather"`` -- both factor matrices fit one GPU (Netflix on 288 GB).  Every rank
  holds full replicas of X and Theta, solves its slab of rows with the fused kernel and
  the slabs are exchanged with ONE all-gather per half-iteration.  No Gram leaves a GPU.
* ``"reduce"`` -- the hugewiki scheme: X is row-sharded and stays device-resident
  (hugewiki.cu:2273-2275 `csc_m[]` slabs), Theta is replicated.  update-X needs no
  communication.  update-Theta: each rank forms the PARTIAL Gram/RHS of every Theta row
  over its own X slab (slab-local CSC, lambda * n_local on the diagonal so that the
  partials sum to the full system: hugewiki.cu:1187-1687), one reduce-scatter sums them
  and leaves each rank 1/G of the systems, which it solves; one all-gather returns Theta.
  Bytes per GPU: (G-1)/G * batch*f*f*4 instead of the reference's serial (G-1) full copies
  into GPU 0.

Compute is injected through an "ops" object so that the partition + collective logic
can be exercised on CPU (gloo, world_size 2) with a stand-in; the product ops are
`HipOps` (libALS.so kernels), and nothing else is ever used outside tests.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
import torch.distributed as dist


# ----------------------------------------------------------------------------------------
# partitioning (pure functions of the row pointers; covered by CPU tests)
# ----------------------------------------------------------------------------------------

def solve_row_cost(f: int, solver) -> float:
    """What one row's SOLVE costs, in ratings of Gram work (measured on MI355X at f = 100: the in-kernel LU of a
    100 x 100 system takes as long as the Gram pass over ~180 ratings, CG(6) as ~70; LU ~ f^3 against f^2 per
    rating, CG ~ f^2 against f^2).  Used to balance slabs by cost instead of by ratings alone: on a side with
    many short rows (the Netflix Theta side: 206 ratings per row) the solves are half of the time."""
    return 1.8 * f if solver in ("lu", 1) else 70.0


def balanced_slabs(rowptr: np.ndarray, parts: int, row_cost: float = 0.0) -> np.ndarray:
    """Cut rows [0, R) into `parts` contiguous slabs of (nearly) equal cost = ratings + row_cost * rows
    (row_cost = 0: equal nnz).

    Returns `parts + 1` boundaries.  Replaces the hand-written `csc_m[]` /
    dynamic batch queue of hugewiki.cu:2273-2275, 2490-2496 with a static split.
    """
    rowptr = np.asarray(rowptr, dtype=np.int64)
    rows = len(rowptr) - 1
    cost = (rowptr - rowptr[0]).astype(np.float64) + float(row_cost) * np.arange(rows + 1, dtype=np.float64)
    targets = cost[-1] * np.arange(1, parts, dtype=np.float64) / parts
    cuts = np.searchsorted(cost, targets, side="left")
    bounds = np.concatenate([[0], np.clip(cuts, 0, rows), [rows]]).astype(np.int64)
    return np.maximum.accumulate(bounds)


def slice_csr(rowptr, colidx, val, r0: int, r1: int):
    """Rows [r0, r1) of a CSR matrix with the row pointer rebased to 0
    (the reference does this per batch with `zeroIndex`, hugewiki.cu:390-395, 2511)."""
    s, e = int(rowptr[r0]), int(rowptr[r1])
    return (rowptr[r0:r1 + 1] - rowptr[r0]), colidx[s:e], val[s:e]


def local_csc_of_slab(rowptr_l, colidx_l, val_l, n_cols: int):
    """CSC (as CSR of the transpose) of a row slab, with slab-LOCAL row ids -- the
    per-GPU `R_train_csc.*.bin{g}` files of hugewiki.cu:2332-2340, built here instead of
    pre-split on disk.  Inputs are numpy arrays of the rebased slab CSR."""
    rows = len(rowptr_l) - 1
    counts = np.diff(rowptr_l)
    row_of = np.repeat(np.arange(rows, dtype=np.int64), counts)
    order = np.argsort(colidx_l.astype(np.int64) * rows + row_of, kind="stable")
    colptr = np.zeros(n_cols + 1, dtype=np.int64)
    np.cumsum(np.bincount(colidx_l, minlength=n_cols), out=colptr[1:])
    return colptr.astype(rowptr_l.dtype), row_of[order].astype(np.int32), val_l[order]


def local_csc_of_slab_torch(rowptr_l: torch.Tensor, colidx_l: torch.Tensor, val_l: torch.Tensor, n_cols: int):
    """Same as `local_csc_of_slab`, on the device with torch ops (hugewiki-size slabs: 388 M
    ratings per GPU).  Returns (colptr int64 on host as numpy, rowidx int32, val) -- the last two
    on the device."""
    rows = rowptr_l.numel() - 1
    counts = (rowptr_l[1:] - rowptr_l[:-1]).to(torch.int64)
    row_of = torch.repeat_interleave(torch.arange(rows, device=colidx_l.device, dtype=torch.int64), counts)
    order = torch.argsort(colidx_l.to(torch.int64) * rows + row_of)
    colptr = torch.zeros(n_cols + 1, dtype=torch.int64, device=colidx_l.device)
    colptr[1:] = torch.cumsum(torch.bincount(colidx_l, minlength=n_cols), 0)
    return colptr.cpu().numpy(), row_of[order].to(torch.int32), val_l[order]


# ----------------------------------------------------------------------------------------
# compute ops (product = HIP kernels through the C ABI)
# ----------------------------------------------------------------------------------------

class HipOps:
    """Compute ops backed by libALS.so on the current CUDA device."""

    def __init__(self, device):
        from . import als

        self.als = als
        self.device = torch.device(device)

    def to_device(self, a: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def plan(self, rowptr: np.ndarray, f: int, chunk: int = 0, row_begin: int = 0, row_end=None):
        return self.als.Plan(np.ascontiguousarray(rowptr), f, row_begin, row_end, chunk)

    def update_fused(self, plan, colidx, val, gather, update, lam, solver, cg_iters):
        self.als.update_fused(plan, colidx, val, gather, update, lam, solver, cg_iters)

    def check(self) -> None:
        self.als.check_gram_fast()

    def release_scratch(self) -> None:
        self.als.release_scratch()

    def quad_terms(self, tt, rhs, x, reg, acc) -> None:
        """acc (1-element fp64 device tensor) += sum over the systems of 2 x.b - x^T A x + reg |x|^2
        (cumf_quadratic_sse_terms; reg < 0 marks a system without ratings).  No host synchronisation: the Theta pipeline
        keeps its Gram(b + 1) || reduce-scatter(b) overlap."""
        if rhs.shape[0] > 0:
            self.als.quadratic_sse_terms(tt, rhs, x, reg, out=acc)

    def fused_sse_available(self, plan, solver) -> bool:
        return self.als.fused_sse_available(plan, solver)

    def update_fused_sse(self, plan, colidx, val, gather, update, lam, solver, cg_iters, bins):
        self.als.update_fused_sse(plan, colidx, val, gather, update, lam, solver, cg_iters, bins)

    def get_hermitian(self, plan, colidx, val, gather, lam, tt, rhs):
        self.als.get_hermitian(plan, colidx, val, gather, lam, tt, rhs)

    def get_hermitian_packed(self, plan, colidx, val, gather, lam, packed, rhs):
        """Partial Gram batch as packed upper triangles, straight from the accumulators (no f x f batch)."""
        self.als.get_hermitian_packed(plan, colidx, val, gather, lam, packed, rhs)

    def solve(self, tt, rhs, x, solver, cg_iters):
        if solver in ("cg", 0):
            self.als.cg_solve(tt, x, rhs, cg_iters)
        else:
            self.als.lu_solve(tt, rhs, x)

    def pack_upper(self, full, packed):
        self.als.pack_upper(full, packed)

    def unpack_upper(self, packed, full):
        self.als.unpack_upper(packed, full)

    def sse(self, val, row, col, thetaT, XT) -> float:
        """sum (r - x_row . theta_col)^2 over the given ratings (als.cu:191-219)."""
        if val.numel() == 0:
            return 0.0
        return float(self.als.sse(val, row, col, thetaT, XT).item())


# ----------------------------------------------------------------------------------------
# collectives: on device tensors with nccl (RCCL), staged through the host with gloo
# ----------------------------------------------------------------------------------------

def _needs_host_staging(t: torch.Tensor) -> bool:
    return t.is_cuda and dist.get_backend() != "nccl"


class SlabGather:
    """out[bounds[g]:bounds[g+1]] <- rank g's slab, for all g (row slabs of unequal size), as ONE
    `all_gather_into_tensor` on slabs padded to the largest one (a single RCCL collective; on the
    fully connected xGMI mesh every link carries 1/(G-1) of it).  The padded send / receive buffers
    and the row map from the padded layout back to `out` are built once and reused: a call is one
    copy into the send buffer, the collective, and one `index_select` (no per-rank Python loop, no
    allocation)."""

    def __init__(self, bounds, cols: int, dtype, device, group=None):
        self.group = group
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.world = world
        self.sizes = [int(bounds[g + 1] - bounds[g]) for g in range(world)]
        self.mx = max(self.sizes) if self.sizes else 0
        self.staged = device.type == "cuda" and dist.is_initialized() and dist.get_backend() != "nccl"
        stage_dev = torch.device("cpu") if self.staged else device
        self.send = torch.zeros((self.mx, cols), dtype=dtype, device=stage_dev)
        self.recv = torch.empty((world * self.mx, cols), dtype=dtype, device=stage_dev)
        idx = np.concatenate([g * self.mx + np.arange(self.sizes[g], dtype=np.int64) for g in range(world)]) \
            if world else np.zeros(0, np.int64)
        self.idx = torch.from_numpy(idx).to(stage_dev)

    def __call__(self, out: torch.Tensor, mine: torch.Tensor) -> None:
        if self.world == 1 and not dist.is_initialized():
            if out.data_ptr() != mine.data_ptr():
                out[: mine.shape[0]].copy_(mine)
            return
        self.send[: mine.shape[0]].copy_(mine)
        dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        if self.staged:
            out.copy_(self.recv.index_select(0, self.idx))
        else:
            torch.index_select(self.recv, 0, self.idx, out=out)


def pipeline_bounds(rowptr: np.ndarray, slab_bounds: np.ndarray, chunks: int, row_cost: float = 0.0) -> np.ndarray:
    """Every rank's row slab cut into `chunks` contiguous cost-balanced pieces: [world, chunks + 1] global row
    ids, computed identically on every rank from the global row pointer."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    world = len(slab_bounds) - 1
    out = np.zeros((world, chunks + 1), dtype=np.int64)
    for g in range(world):
        a, b = int(slab_bounds[g]), int(slab_bounds[g + 1])
        out[g] = a + balanced_slabs(rowptr[a:b + 1], chunks, row_cost)
    return out


class PipelinedGather:
    """The all-gather of a row-sharded factor matrix, issued piece by piece while the rest is still being
    computed: rank g updates its slab in `chunks` pieces; after piece c its rows go into a padded send buffer
    and ONE `all_gather_into_tensor` of piece c of every rank is issued asynchronously (RCCL stream), so it
    runs under the kernel of piece c + 1 (compute stream).  `finish` waits for all pieces and places every
    row with one `index_select`.  At 8 GPUs on the Netflix shape that hides 3/4 of the 168 MB each GPU
    receives per X update behind the update itself.  Buffers and the row map are built once."""

    def __init__(self, pb: np.ndarray, rows_total: int, cols: int, dtype, device, group=None):
        self.group = group
        self.world, self.chunks = pb.shape[0], pb.shape[1] - 1
        self.pb = pb
        self.staged = device.type == "cuda" and dist.is_initialized() and dist.get_backend() != "nccl"
        stage_dev = torch.device("cpu") if self.staged else device
        sizes = pb[:, 1:] - pb[:, :-1]                      # [world, chunks]
        self.mx = [int(sizes[:, c].max()) for c in range(self.chunks)]
        self.send = [torch.zeros((self.mx[c], cols), dtype=dtype, device=stage_dev) for c in range(self.chunks)]
        total = sum(self.world * m for m in self.mx)
        self.recv = torch.empty((total, cols), dtype=dtype, device=stage_dev)
        self.off = np.concatenate([[0], np.cumsum([self.world * m for m in self.mx])]).astype(np.int64)
        idx = np.empty(rows_total, dtype=np.int64)
        for c in range(self.chunks):
            for g in range(self.world):
                lo, hi = int(pb[g, c]), int(pb[g, c + 1])
                idx[lo:hi] = self.off[c] + g * self.mx[c] + np.arange(hi - lo, dtype=np.int64)
        self.idx = torch.from_numpy(idx).to(stage_dev)
        self.works = []

    def issue(self, c: int, piece: torch.Tensor) -> None:
        """Piece c of this rank's slab is final: start its all-gather."""
        if self.mx[c] == 0:   # empty on every rank (identical decision everywhere: mx comes from the global row pointer)
            return
        self.send[c][: piece.shape[0]].copy_(piece)
        recv = self.recv[int(self.off[c]): int(self.off[c + 1])]
        self.works.append(dist.all_gather_into_tensor(recv, self.send[c], group=self.group, async_op=True))

    def finish(self, out: torch.Tensor) -> None:
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = []
        if self.staged:
            out.copy_(self.recv.index_select(0, self.idx))
        else:
            torch.index_select(self.recv, 0, self.idx, out=out)


def all_gather_rows(out: torch.Tensor, mine: torch.Tensor, bounds, group=None) -> None:
    """One-shot form of `SlabGather` (builds the buffers for this call only)."""
    SlabGather(bounds, out.shape[1], out.dtype, out.device, group)(out, mine)


def reduce_scatter_rows(full: torch.Tensor, group=None, out: torch.Tensor | None = None, async_op: bool = False):
    """Sum `full` ([world * k, ...]) over ranks; this rank's k rows land in `out` (allocated if None).
    Returns (out, work): `work` is the RCCL work handle when async_op (its .wait() orders the current
    stream behind the collective), else None."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    k = full.shape[0] // world
    if out is None:
        out = torch.empty((k,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    if not dist.is_initialized():
        out.copy_(full[:k])
        return out, None
    if _needs_host_staging(full) or not full.is_cuda:
        # gloo has no reduce_scatter: all_reduce + slice (test path only)
        buf = full.cpu() if full.is_cuda else full.clone()
        dist.all_reduce(buf, group=group)
        r = dist.get_rank(group)
        out.copy_(buf[r * k:(r + 1) * k])
        return out, None
    work = dist.reduce_scatter_tensor(out, full, group=group, async_op=async_op)
    return out, (work if async_op else None)


def all_gather_equal(out: torch.Tensor, mine: torch.Tensor, group=None) -> None:
    if not dist.is_initialized():
        out.copy_(mine)
        return
    if _needs_host_staging(out):
        recv = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(recv, mine.cpu(), group=group)
        out.copy_(recv)
    else:
        dist.all_gather_into_tensor(out, mine.contiguous(), group=group)


# ----------------------------------------------------------------------------------------
# the native half-iterations (include/cumf_dist_capi.h, csrc/als_dist.cpp): kernels and collectives enqueued back to
# back from C++ -- no Python between a kernel and its collective.  The classes below only hold the handles.
# ----------------------------------------------------------------------------------------

_NATIVE = None  # set_native(): True / False override the environment, None = CUMF_DIST_NATIVE (default on)


def set_native(flag) -> None:
    """Engines constructed from now on: True = the native half-iterations, False = torch.distributed collectives driven from
    Python (the A/B of bench.py and the tests), None = environment CUMF_DIST_NATIVE (default 1)."""
    global _NATIVE
    _NATIVE = flag


def native_wanted(ops) -> bool:
    """The native path is the product path of `HipOps`; stand-in ops (CPU tests) have no kernels to enqueue."""
    import os

    if not isinstance(ops, HipOps):
        return False
    return _NATIVE if _NATIVE is not None else os.environ.get("CUMF_DIST_NATIVE", "1") != "0"


class _TorchTransport:
    """`cumf_transport_t` over a torch.distributed group that cannot take device buffers (gloo: the 2-processes-on-one-GPU
    tests): the callbacks synchronise the communication stream, stage through the host, and return with the result in
    `recv`.  With the "nccl" backend the native layer owns an RCCL communicator instead (`NativeComm`)."""

    def __init__(self, group):
        import ctypes as C

        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.error = None
        ag_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
        rs_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
        ar_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

        class Transport(C.Structure):
            _fields_ = [("ctx", C.c_void_p), ("all_gather", ag_t), ("reduce_scatter_f32", rs_t), ("all_reduce_f64", ar_t)]

        self._cbs = (ag_t(self._guard(self._all_gather)), rs_t(self._guard(self._reduce_scatter)),
                     ar_t(self._guard(self._all_reduce)))  # kept alive with the object
        self.struct = Transport(None, *self._cbs)

    def _guard(self, fn):
        def wrapped(*args):
            try:
                fn(*args)
                return 0
            except Exception as e:  # an exception must not unwind through the C frames
                self.error = e
                return 1
        return wrapped

    def _d2h(self, ptr, nbytes, stream) -> torch.Tensor:
        host = torch.empty(nbytes, dtype=torch.uint8)
        if self.hip.hipStreamSynchronize(stream) or self.hip.hipMemcpy(host.data_ptr(), ptr, nbytes, 2):
            raise RuntimeError("hipMemcpy device -> host failed")
        return host

    def _h2d(self, ptr, host: torch.Tensor) -> None:
        if self.hip.hipMemcpy(ptr, host.data_ptr(), host.numel() * host.element_size(), 1):
            raise RuntimeError("hipMemcpy host -> device failed")

    def _all_gather(self, ctx, send, recv, nbytes, stream):
        mine = self._d2h(send, nbytes, stream)
        out = torch.empty(self.world * nbytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, mine, group=self.group)
        self._h2d(recv, out)

    def _reduce_scatter(self, ctx, send, recv, count, stream):
        full = self._d2h(send, self.world * count * 4, stream).view(torch.float32)
        dist.all_reduce(full, group=self.group)  # gloo has no reduce_scatter
        self._h2d(recv, full[self.rank * count:(self.rank + 1) * count].contiguous())

    def _all_reduce(self, ctx, buf, count, stream):
        v = self._d2h(buf, count * 8, stream).view(torch.float64)
        dist.all_reduce(v, group=self.group)
        self._h2d(buf, v)


class NativeComm:
    """`cumf_comm_t` of this rank: an RCCL communicator of its own when the group's backend is "nccl" (the id travels by
    one broadcast), device copies without a process group, the host-staged `_TorchTransport` otherwise."""

    def __init__(self, device, group=None):
        import ctypes as C

        from . import lib as _libmod

        self.lib = _libmod.load()
        self.check = _libmod.check
        self._h = C.c_void_p()
        self.transport = None
        torch.cuda.set_device(device)
        if not dist.is_initialized():
            self.check(self.lib.cumf_comm_create_local(C.byref(self._h)), "cumf_comm_create_local")
        elif dist.get_backend(group) == "nccl":
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            ident = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                self.check(self.lib.cumf_comm_unique_id(C.c_void_p(ident.data_ptr())), "cumf_comm_unique_id")
            ident = ident.to(device)
            dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = ident.cpu()
            self.check(self.lib.cumf_comm_create(C.byref(self._h), C.c_void_p(ident.data_ptr()), rank, world),
                       "cumf_comm_create")
        else:
            self.transport = _TorchTransport(group)
            self.check(self.lib.cumf_comm_create_custom(C.byref(self._h), C.byref(self.transport.struct),
                                                        self.transport.rank, self.transport.world),
                       "cumf_comm_create_custom")
        self.name = self.lib.cumf_comm_transport_name(self._h).decode()

    def checked(self, rc: int, what: str) -> None:
        if rc != 0 and self.transport is not None and self.transport.error is not None:
            err, self.transport.error = self.transport.error, None
            raise RuntimeError(f"{what}: transport callback failed") from err
        self.check(rc, what)

    def all_reduce_f64(self, t: torch.Tensor) -> torch.Tensor:
        """Sum a contiguous fp64 device tensor over the ranks, in place, stream-ordered."""
        import ctypes as C

        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        self.checked(self.lib.cumf_comm_all_reduce_f64(self._h, C.c_void_p(t.data_ptr()), t.numel(),
                                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                     "cumf_comm_all_reduce_f64")
        return t

    def close(self) -> None:
        if self._h:
            self.lib.cumf_comm_destroy(self._h)
            self._h = None


def _plan_array(plans):
    import ctypes as C

    return (C.c_void_p * len(plans))(*[p._h if p is not None else None for p in plans])


class NativeGather:
    """One side of the `gather` scheme (`cumf_dist_gather_*`): this rank's slab in pieces, the all-gather of piece c under
    the kernel of piece c + 1, all enqueued by one C call."""

    def __init__(self, comm: NativeComm, pb: np.ndarray, piece_plans, f: int, gather_rows: int):
        import ctypes as C

        self.comm, self.f = comm, f
        pb = np.ascontiguousarray(pb, dtype=np.int64)
        self.pieces = pb.shape[1] - 1
        self.plans = list(piece_plans)
        for p in self.plans:
            if p is not None:  # the pre-split gather table of cache-resident sides needs the table's row count
                comm.check(comm.lib.cumf_plan_set_gather_rows(p._h, int(gather_rows)), "cumf_plan_set_gather_rows")
        self._arr = _plan_array(self.plans)
        self._h = C.c_void_p()
        comm.check(comm.lib.cumf_dist_gather_create(C.byref(self._h), comm._h, pb.ctypes.data_as(C.c_void_p), self.pieces, f),
                   "cumf_dist_gather_create")

    def update(self, colidx, val, table, out, lam, solver, cg_iters, sse_bins=None) -> None:
        import ctypes as C

        from .als import _dp, _solver_id

        self.comm.checked(self.comm.lib.cumf_dist_gather_update(
            self._h, self._arr, _dp(colidx, torch.int32), _dp(val, torch.float32), _dp(table, torch.float32),
            _dp(out, torch.float32), float(lam), _solver_id(solver), int(cg_iters),
            _dp(sse_bins, torch.float64) if sse_bins is not None else None,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cumf_dist_gather_update")

    def close(self) -> None:
        if self._h:
            self.comm.lib.cumf_dist_gather_destroy(self._h)
            self._h = None


class NativeReduce:
    """The Theta update of the `reduce` scheme (`cumf_dist_reduce_*`): partial Grams -> reduce-scatter -> solve ->
    all-gather per Theta batch, pipelined over the batches, one C call per update."""

    def __init__(self, comm: NativeComm, n: int, f: int, batch_plans):
        import ctypes as C

        self.comm = comm
        self.plans = list(batch_plans)
        self._arr = _plan_array(self.plans)
        self._h = C.c_void_p()
        comm.check(comm.lib.cumf_dist_reduce_create(C.byref(self._h), comm._h, int(n), int(f), len(self.plans)),
                   "cumf_dist_reduce_create")

    def update_theta(self, lc_rowidx, lc_val, XT, thetaT, lam, solver, cg_iters, reg_all=None, terms=None) -> None:
        import ctypes as C

        from .als import _dp, _solver_id

        self.comm.checked(self.comm.lib.cumf_dist_reduce_update_theta(
            self._h, self._arr, _dp(lc_rowidx, torch.int32), _dp(lc_val, torch.float32), _dp(XT, torch.float32),
            _dp(thetaT, torch.float32), float(lam), _solver_id(solver), int(cg_iters),
            _dp(reg_all, torch.float32) if reg_all is not None else None,
            _dp(terms, torch.float64) if terms is not None else None,
            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cumf_dist_reduce_update_theta")

    def close(self) -> None:
        if self._h:
            self.comm.lib.cumf_dist_reduce_destroy(self._h)
            self._h = None


# ----------------------------------------------------------------------------------------
# the distributed engine
# ----------------------------------------------------------------------------------------

@dataclass
class HostMatrix:
    """Train matrix on the host: CSR over rows (m x n) and CSC over columns."""
    m: int
    n: int
    csr_indptr: np.ndarray
    csr_indices: np.ndarray
    csr_data: np.ndarray
    csc_indptr: np.ndarray
    csc_indices: np.ndarray
    csc_data: np.ndarray


class DistALS:
    """ALS over `world` ranks.  Every rank constructs it with the same host matrix (or, for
    `scheme="reduce"`, at least its own row slab -- see `from_local_slab`)."""

    def __init__(self, mat: HostMatrix, f: int, lam: float, ops, solver="cg", cg_iters: int = 6,
                 scheme: str = "gather", theta_batch: int = 1, group=None, chunk: int = 0, solver_x=None,
                 solver_theta=None, cg_iters_x=None, cg_iters_theta=None):
        self.f, self.lam, self.ops = f, float(lam), ops
        self.solver, self.cg_iters, self.scheme = solver, cg_iters, scheme
        self._set_side_solvers(solver_x, solver_theta, cg_iters_x, cg_iters_theta)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.m, self.n = mat.m, mat.n
        self.theta_batch = theta_batch
        dev = ops.to_device(np.zeros(1, np.float32)).device
        self.thetaT = torch.zeros((self.n, f), dtype=torch.float32, device=dev)

        # X side: contiguous nnz-balanced row slabs
        self.xb = balanced_slabs(mat.csr_indptr, self.world, solve_row_cost(f, self.solver_x))
        x0, x1 = int(self.xb[self.rank]), int(self.xb[self.rank + 1])
        rp, ci, va = slice_csr(mat.csr_indptr, mat.csr_indices, mat.csr_data, x0, x1)
        self.x_rows = x1 - x0
        self.x_plan = ops.plan(rp, f, chunk)
        self.x_colidx, self.x_val = ops.to_device(ci), ops.to_device(va)
        self._x_pipe = None
        if scheme == "gather":
            self._make_x_pipeline(mat.csr_indptr, np.asarray(rp), chunk)

        if scheme == "gather":
            self.XT = torch.zeros((self.m, f), dtype=torch.float32, device=dev)
            self.tb = balanced_slabs(mat.csc_indptr, self.world, solve_row_cost(f, self.solver_theta))
            t0, t1 = int(self.tb[self.rank]), int(self.tb[self.rank + 1])
            rp, ci, va = slice_csr(mat.csc_indptr, mat.csc_indices, mat.csc_data, t0, t1)
            self.t_rows = t1 - t0
            self.t_plan = ops.plan(rp, f, chunk)
            self.t_colidx, self.t_val = ops.to_device(ci), ops.to_device(va)
            # the Theta all-gather is the big one (n x f: 192 MB at the Netflix shape): pipelined like X
            self._t_pipe = self._make_pipeline(mat.csc_indptr, np.asarray(rp), self.tb, chunk,
                                               solve_row_cost(f, self.solver_theta))
        elif scheme == "reduce":
            # X slab only (device-resident for the whole run); slab-local CSC for the partial Grams
            self.XT = torch.zeros((self.x_rows, f), dtype=torch.float32, device=dev)
            cp, ri, cv = local_csc_of_slab(np.asarray(rp), np.asarray(ci), np.asarray(va), self.n)
            self.lc_rowidx, self.lc_val = ops.to_device(ri), ops.to_device(cv)
            self._lc_counts = np.diff(np.asarray(cp, dtype=np.int64))  # ratings of this slab per Theta column
            # Theta batches (als.cu:881-890), each padded to a multiple of world for the reduce-scatter
            self.t_batches = []
            for b in range(theta_batch):
                size = self.n // theta_batch if b != theta_batch - 1 else self.n - b * (self.n // theta_batch)
                off = b * (self.n // theta_batch)
                self.t_batches.append((off, size, ops.plan(cp, f, chunk, off, off + size)))
        else:
            raise ValueError(scheme)
        self._setup_comm()

    def _set_side_solvers(self, solver_x, solver_theta, cg_iters_x, cg_iters_theta) -> None:
        """Per-side solver: the reference's hugewiki run solves X with CG, 100 iterations (hugewiki.cu:2569), and
        Theta with the batched LU on the reduced Gram (hugewiki.cu:2732); `solver` / `cg_iters` are the defaults
        of both sides."""
        self.solver_x = self.solver if solver_x is None else solver_x
        self.solver_theta = self.solver if solver_theta is None else solver_theta
        self.cg_iters_x = self.cg_iters if cg_iters_x is None else cg_iters_x
        self.cg_iters_theta = self.cg_iters if cg_iters_theta is None else cg_iters_theta

    def _make_pipeline(self, rowptr_global, rowptr_local, bounds, chunk: int, row_cost: float = 0.0):
        """One side's update in pieces with the all-gather of each piece under the next one (`PipelinedGather`).
        Pieces: CUMF_ALS_PIPE_CHUNKS if set (1 = one kernel + one blocking all-gather); otherwise 4 when the
        gathered factor matrix is large enough for its all-gather to matter (>= 32 MB: the Netflix Theta side,
        192 MB) and 1 when it is not (the Netflix X side, 7 MB = tens of microseconds over xGMI: cutting a 1/8
        slab's few thousand work items into four launches would only leave wave slots empty in each of them).
        Returns (piece bounds [world, chunks + 1], [(lo, hi, plan)] of this rank) or None."""
        import os

        env = os.environ.get("CUMF_ALS_PIPE_CHUNKS")
        if env is not None:
            chunks = int(env)
        else:
            gathered_bytes = (len(rowptr_global) - 1) * self.f * 4
            chunks = 4 if gathered_bytes >= (32 << 20) else 1
        force = os.environ.get("CUMF_ALS_PIPE_FORCE") == "1"  # tests: the RCCL path with world_size 1
        if chunks <= 1 or not dist.is_initialized() or (self.world <= 1 and not force):
            return None
        pb = pipeline_bounds(rowptr_global, bounds, chunks, row_cost)
        r0 = int(bounds[self.rank])
        plans = []
        for c in range(chunks):
            lo, hi = int(pb[self.rank, c]) - r0, int(pb[self.rank, c + 1]) - r0
            plans.append((lo, hi, self.ops.plan(rowptr_local, self.f, chunk, lo, hi) if hi > lo else None))
        return (pb, plans)

    def _make_x_pipeline(self, rowptr_global, rowptr_local, chunk: int) -> None:
        self._x_pipe = self._make_pipeline(rowptr_global, rowptr_local, self.xb, chunk,
                                           solve_row_cost(self.f, self.solver_x))

    def _setup_comm(self) -> None:
        """Persistent communication buffers (nothing is allocated or zero-filled
        per half-iteration)."""
        dev, f, w = self.thetaT.device, self.f, self.world
        self._gx = self._gt = None
        self._sse_const = None  # (sum r^2 over all ranks, lambda * n_v per Theta column): built on first use
        self._sum_r2 = None     # gather scheme: sum r^2 over all ranks (near-perfect-fit rule of the fused train SSE)
        self._fused_sse_ok = {}  # solver -> every RANK's Theta plans deliver the fused train SSE (decided collectively, once)
        # the native half-iterations (als_dist.cpp): the product path of the HIP ops -- kernels and RCCL collectives enqueued
        # back to back by one C call per half-iteration, no torch.distributed call and no Python in between
        self._ncomm = self._nx = self._nt = self._nr = None
        self._px = self._pt = None
        if native_wanted(self.ops) and dev.type == "cuda":
            self._ncomm = NativeComm(dev, self.group)
            if self.scheme == "gather":
                def side(pipe, bounds, plan, gather_rows):
                    if pipe is not None:
                        return NativeGather(self._ncomm, pipe[0], [p for (_, _, p) in pipe[1]], f, gather_rows)
                    b = np.asarray(bounds, dtype=np.int64)
                    return NativeGather(self._ncomm, np.stack([b[:-1], b[1:]], axis=1), [plan], f, gather_rows)

                self._nx = side(getattr(self, "_x_pipe", None), self.xb, self.x_plan, self.n)
                self._nt = side(getattr(self, "_t_pipe", None), self.tb, self.t_plan, self.m)
            else:
                self._nr = NativeReduce(self._ncomm, self.n, f, [p for (_, _, p) in self.t_batches])
            return
        if self.scheme == "gather":
            self._gx = SlabGather(self.xb, f, torch.float32, dev, self.group)
            self._gt = SlabGather(self.tb, f, torch.float32, dev, self.group)
            self._px = PipelinedGather(self._x_pipe[0], self.m, f, torch.float32, dev, self.group) \
                if getattr(self, "_x_pipe", None) is not None else None
            self._pt = PipelinedGather(self._t_pipe[0], self.n, f, torch.float32, dev, self.group) \
                if getattr(self, "_t_pipe", None) is not None else None
            return
        # reduce scheme.  Per Theta batch: k = ceil(size / world) systems per rank.  Two sets of
        # buffers so that the reduce-scatter of batch b runs (RCCL stream) under the Gram pass of
        # batch b + 1 (compute stream).  Payload = packed upper triangles (f (f + 1) / 2 floats per
        # system instead of f * f: 0.80 GB instead of 1.59 GB per hugewiki Theta batch at f = 100),
        # written by the Gram kernels straight from their accumulators (cumf_get_hermitian_packed: no
        # f x f partial batch, no pack pass), + the RHS (f floats per system) as a second, small collective.
        self._kmax = max((size + w - 1) // w for (_, size, _) in self.t_batches)
        self._pk = f * (f + 1) // 2
        nb = 2 if len(self.t_batches) > 1 else 1
        self._nbuf = nb
        self._rhs = [torch.empty((w * self._kmax, f), dtype=torch.float32, device=dev) for _ in range(nb)]
        self._tri = [torch.zeros((w * self._kmax, self._pk), dtype=torch.float32, device=dev) for _ in range(nb)]
        self._mine = [torch.empty((self._kmax, self._pk), dtype=torch.float32, device=dev) for _ in range(nb)]
        self._mine_rhs = [torch.empty((self._kmax, f), dtype=torch.float32, device=dev) for _ in range(nb)]
        self._my_tt = torch.empty((self._kmax, f, f), dtype=torch.float32, device=dev)
        self._my_rhs = torch.empty((self._kmax, f), dtype=torch.float32, device=dev)
        self._x = torch.zeros((self._kmax, f), dtype=torch.float32, device=dev)
        self._gathered = torch.empty((w * self._kmax, f), dtype=torch.float32, device=dev)

    @classmethod
    def from_device_ratings(cls, r, f: int, lam: float, ops, solver="cg", cg_iters: int = 6, group=None,
                            chunk: int = 0, solver_x=None, solver_theta=None, cg_iters_x=None,
                            cg_iters_theta=None) -> "DistALS":
        """`gather` scheme from a `datagen.Ratings` that is already on this rank's device: the slabs are
        zero-copy views of its CSR / CSC arrays (no host round trip of the 1.6 GB matrix; only the two
        row-pointer arrays, a few MB, go to the host for the plans)."""
        self = cls.__new__(cls)
        self.f, self.lam, self.ops = f, float(lam), ops
        self.solver, self.cg_iters, self.scheme = solver, cg_iters, "gather"
        self._set_side_solvers(solver_x, solver_theta, cg_iters_x, cg_iters_theta)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.m, self.n = r.m, r.n
        self.theta_batch = 1
        dev = r.csr_indices.device
        self.thetaT = torch.zeros((self.n, f), dtype=torch.float32, device=dev)
        self.XT = torch.zeros((self.m, f), dtype=torch.float32, device=dev)
        rp = r.csr_indptr.cpu().numpy().astype(np.int64)
        cp = r.csc_indptr.cpu().numpy().astype(np.int64)
        self.xb = balanced_slabs(rp, self.world, solve_row_cost(f, self.solver_x))
        self.tb = balanced_slabs(cp, self.world, solve_row_cost(f, self.solver_theta))
        x0, x1 = int(self.xb[self.rank]), int(self.xb[self.rank + 1])
        t0, t1 = int(self.tb[self.rank]), int(self.tb[self.rank + 1])
        self.x_rows, self.t_rows = x1 - x0, t1 - t0
        self.x_plan = ops.plan(rp[x0:x1 + 1] - rp[x0], f, chunk)
        self.t_plan = ops.plan(cp[t0:t1 + 1] - cp[t0], f, chunk)
        self._make_x_pipeline(rp, rp[x0:x1 + 1] - rp[x0], chunk)
        self._t_pipe = self._make_pipeline(cp, cp[t0:t1 + 1] - cp[t0], self.tb, chunk,
                                           solve_row_cost(f, self.solver_theta))
        self.x_colidx, self.x_val = r.csr_indices[rp[x0]:rp[x1]], r.csr_data[rp[x0]:rp[x1]]
        self.t_colidx, self.t_val = r.csc_indices[cp[t0]:cp[t1]], r.csc_data[cp[t0]:cp[t1]]
        self._setup_comm()
        return self

    @classmethod
    def from_local_slab(cls, m_total: int, n: int, xb, rowptr_l: torch.Tensor, colidx_l: torch.Tensor,
                        val_l: torch.Tensor, f: int, lam: float, ops, solver="cg", cg_iters: int = 6,
                        theta_batch: int = 1, group=None, chunk: int = 0, solver_x=None, solver_theta=None,
                        cg_iters_x=None, cg_iters_theta=None) -> "DistALS":
        """`reduce` scheme from this rank's row slab only (hugewiki scale: no rank ever holds the
        whole matrix -- the reference pre-splits it into per-GPU files, hugewiki.cu:2332-2340).
        `xb`: the world+1 global slab boundaries; the three tensors are the slab's CSR with the
        row pointer rebased to 0, already on the device."""
        self = cls.__new__(cls)
        self.f, self.lam, self.ops = f, float(lam), ops
        self.solver, self.cg_iters, self.scheme = solver, cg_iters, "reduce"
        self._set_side_solvers(solver_x, solver_theta, cg_iters_x, cg_iters_theta)
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.m, self.n = m_total, n
        self.theta_batch = theta_batch
        self.xb = np.asarray(xb, dtype=np.int64)
        dev = colidx_l.device
        self.thetaT = torch.zeros((n, f), dtype=torch.float32, device=dev)
        self.x_rows = rowptr_l.numel() - 1
        self.x_plan = ops.plan(rowptr_l.cpu().numpy(), f, chunk)
        self.x_colidx, self.x_val = colidx_l, val_l
        self.XT = torch.zeros((self.x_rows, f), dtype=torch.float32, device=dev)
        cp, self.lc_rowidx, self.lc_val = local_csc_of_slab_torch(rowptr_l, colidx_l, val_l, n)
        self._lc_counts = np.diff(np.asarray(cp, dtype=np.int64))  # ratings of this slab per Theta column
        self.t_batches = []
        for b in range(theta_batch):
            size = n // theta_batch if b != theta_batch - 1 else n - b * (n // theta_batch)
            off = b * (n // theta_batch)
            self.t_batches.append((off, size, ops.plan(cp, f, chunk, off, off + size)))
        self._setup_comm()
        return self

    # -- factors ---------------------------------------------------------------------------
    def init_factors(self, thetaT: np.ndarray, XT: np.ndarray | None = None) -> None:
        self.thetaT.copy_(torch.from_numpy(np.ascontiguousarray(thetaT, np.float32)).reshape(self.n, self.f))
        if XT is None:
            self.XT.zero_()
        else:
            XT = np.ascontiguousarray(XT, np.float32).reshape(self.m, self.f)
            if self.scheme == "reduce":
                XT = XT[int(self.xb[self.rank]):int(self.xb[self.rank + 1])]
            self.XT.copy_(torch.from_numpy(XT))

    def full_XT(self) -> torch.Tensor:
        """X on every rank (gathers the slabs in the "reduce" scheme)."""
        if self.scheme == "gather":
            return self.XT
        out = torch.empty((self.m, self.f), dtype=torch.float32, device=self.XT.device)
        if dist.is_initialized():
            all_gather_rows(out, self.XT, self.xb, self.group)
        else:
            out.copy_(self.XT)
        return out

    # -- half-iterations -------------------------------------------------------------------
    def _update_gathered(self, pipe, pg, gather_all, plan, colidx, val, table, out, bounds, solver, cg_iters,
                         sse_bins=None) -> None:
        """`gather` scheme, one side: this rank's slab of `out` from the replicated `table`, then the slabs of
        all ranks exchanged -- piece by piece under the next piece's kernel when a pipeline exists."""
        native = self._nx if out is self.XT else self._nt
        if native is not None:
            native.update(colidx, val, table, out, self.lam, solver, cg_iters, sse_bins)
            return
        r0, r1 = int(bounds[self.rank]), int(bounds[self.rank + 1])
        mine = out[r0:r1]
        if pg is not None:
            for c, (lo, hi, piece_plan) in enumerate(pipe[1]):
                if piece_plan is not None:
                    if sse_bins is not None:
                        self.ops.update_fused_sse(piece_plan, colidx, val, table, mine, self.lam, solver, cg_iters, sse_bins)
                    else:
                        self.ops.update_fused(piece_plan, colidx, val, table, mine, self.lam, solver, cg_iters)
                pg.issue(c, mine[lo:hi])
            pg.finish(out)
            return
        if sse_bins is not None:
            self.ops.update_fused_sse(plan, colidx, val, table, mine, self.lam, solver, cg_iters, sse_bins)
        else:
            self.ops.update_fused(plan, colidx, val, table, mine, self.lam, solver, cg_iters)
        gather_all(out, mine)

    def update_x(self) -> None:
        if self.scheme == "gather":
            self._update_gathered(self._x_pipe, self._px, self._gx, self.x_plan, self.x_colidx, self.x_val,
                                  self.thetaT, self.XT, self.xb, self.solver_x, self.cg_iters_x)
        else:
            self.ops.update_fused(self.x_plan, self.x_colidx, self.x_val, self.thetaT, self.XT, self.lam,
                                  self.solver_x, self.cg_iters_x)

    # -- train SSE out of the Theta update (no pass over the ratings) ---------------------
    def _train_sse_constants(self):
        """(sum of r^2 over the ratings of ALL ranks, lambda * n_v for every Theta column as the reduced systems carry it
        on their diagonal): constants of the data, built once.  `reduce` scheme only."""
        if self._sse_const is None:
            val = self.lc_val
            s = torch.tensor([float((val.double() ** 2).sum().item())], dtype=torch.float64)
            cnt = torch.from_numpy(self._lc_counts).double()
            if dist.is_initialized() and self.world > 1:
                if dist.get_backend() == "nccl":
                    dev = self.XT.device
                    s, cnt = s.to(dev), cnt.to(dev)
                dist.all_reduce(s, group=self.group)
                dist.all_reduce(cnt, group=self.group)
            # lambda n_v; -1 marks a column without ratings anywhere (its solution is NaN and must not enter the sum) --
            # not "reg == 0", which is every column when lambda = 0
            reg = torch.where(cnt > 0, self.lam * cnt, torch.full_like(cnt, -1.0))
            self._sse_const = (float(s.item()), reg.float().to(self.XT.device))
        return self._sse_const

    def _sum_r2_all_ranks(self) -> float:
        """Sum of r^2 over the training ratings of ALL ranks (a constant of the data, all-reduced once): the yardstick of
        the near-perfect-fit rule below.  `reduce` scheme: part of `_train_sse_constants`; `gather` scheme: over this
        rank's Theta slab of the CSC."""
        if self.scheme == "reduce":
            return self._train_sse_constants()[0]
        if getattr(self, "_sum_r2", None) is None:
            s = torch.tensor([float((self.t_val.double() ** 2).sum().item())], dtype=torch.float64)
            if dist.is_initialized() and self.world > 1:
                if dist.get_backend() == "nccl":
                    s = s.to(self.thetaT.device)
                dist.all_reduce(s, group=self.group)
            self._sum_r2 = float(s.item())
        return self._sum_r2

    def _trusted_sse(self, sse: float):
        """The train SSE out of the Theta update is an fp32 identity per column (absolute error ~1e-7 of the column's
        sum r^2): on a near-perfect fit it is cancellation noise and may even come out negative.  The rule of
        doALS (als_driver.cpp): below 1e-3 of sum r^2 the value is not reported -- None, and the caller evaluates
        `slab_sse` directly.  `sse` is already all-reduced, so every rank decides alike."""
        return sse if sse >= 1e-3 * self._sum_r2_all_ranks() else None

    def _fused_sse_everywhere(self) -> bool:
        """`gather` scheme: can the fused kernels of EVERY rank deliver the train SSE of their Theta slab?  Availability
        depends on the plan (a slab with one chunked heavy column is refused for LU below f = 96 or CG at f = 112..128),
        so the ranks may differ -- and a rank that skipped the all-reduce of the bins while the others performed it would
        pair its next collective with theirs.  Decided once per solver by an all-reduce (MIN) of the flag."""
        key = self.solver_theta
        if key not in self._fused_sse_ok:
            ok = getattr(self.ops, "fused_sse_available", None)
            plans = [p for (_, _, p) in self._t_pipe[1] if p is not None] if self._t_pipe is not None else [self.t_plan]
            mine = ok is not None and all(ok(p, self.solver_theta) for p in plans)
            flag = torch.tensor([1 if mine else 0], dtype=torch.int32)
            if dist.is_initialized() and self.world > 1:
                if dist.get_backend() == "nccl":
                    flag = flag.to(self.thetaT.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            self._fused_sse_ok[key] = bool(int(flag.item()))
        return self._fused_sse_ok[key]

    def update_theta(self, train_sse: bool = False):
        """update Theta.  train_sse=True: also return sum over ALL ranks of (r - x_u . theta_v)^2 over the training ratings
        with the new Theta -- from the solved systems themselves, no pass over the ratings (None when the ops or the plans
        cannot deliver it, or when the fit is so close that the identity is cancellation noise -- `_trusted_sse`: the
        caller then runs `slab_sse`)."""
        if self.scheme == "gather":
            bins = None
            if train_sse:
                if not self._fused_sse_everywhere():  # the same answer on every rank: no collective is skipped one-sidedly
                    train_sse = False
                else:
                    bins = torch.zeros(1024, dtype=torch.float64, device=self.thetaT.device)
            self._update_gathered(self._t_pipe, self._pt, self._gt, self.t_plan, self.t_colidx, self.t_val,
                                  self.XT, self.thetaT, self.tb, self.solver_theta, self.cg_iters_theta, bins)
            if not train_sse:
                return None
            t = bins.sum().reshape(1)
            if self._ncomm is not None:
                return self._trusted_sse(float(self._ncomm.all_reduce_f64(t).item()))
            if dist.is_initialized() and self.world > 1:
                if dist.get_backend() != "nccl":
                    t = t.cpu()
                dist.all_reduce(t, group=self.group)
            return self._trusted_sse(float(t.item()))
        quad = getattr(self.ops, "quad_terms", None) if train_sse else None
        terms = None
        if quad is not None:
            s_total, reg_all = self._train_sse_constants()
            terms = torch.zeros(1, dtype=torch.float64, device=self.XT.device)  # accumulated on the device, read once
        if self._nr is not None:  # the whole Theta update in one C call (cumf_dist_reduce_update_theta)
            self._nr.update_theta(self.lc_rowidx, self.lc_val, self.XT, self.thetaT, self.lam, self.solver_theta,
                                  self.cg_iters_theta, reg_all if quad is not None else None, terms)
            if quad is None:
                return None
            return self._trusted_sse(s_total - float(self._ncomm.all_reduce_f64(terms).item()))
        # reduce scheme (replaces hugewiki.cu:2611-2745).  Pipeline over the Theta batches:
        #   packed Gram(b) -> reduce-scatter(b) [async, RCCL stream]   ||   packed Gram(b + 1) ...
        #   wait(b) -> unpack -> solve(b) -> all-gather(b)
        f, w = self.f, self.world
        pending = None  # (batch index, buffer set, work handle)

        def finish(bi, slot, works):
            off, size, _ = self.t_batches[bi]
            k = (size + w - 1) // w
            for wk in works:
                if wk is not None:
                    wk.wait()
            lo, hi = min(self.rank * k, size), min((self.rank + 1) * k, size)
            x = self._x[:k]
            if hi > lo:
                self.ops.unpack_upper(self._mine[slot][: hi - lo], self._my_tt[: hi - lo])
                x[: hi - lo].copy_(self.thetaT[off + lo: off + hi])          # CG warm start
                self.ops.solve(self._my_tt[: hi - lo], self._mine_rhs[slot][: hi - lo], x[: hi - lo],
                               self.solver_theta, self.cg_iters_theta)
                if quad is not None:  # 2 x.b - x^T A x + reg |x|^2 of this rank's systems (the solver left A, b intact)
                    quad(self._my_tt[: hi - lo], self._mine_rhs[slot][: hi - lo], x[: hi - lo],
                         reg_all[off + lo: off + hi].contiguous(), terms)
            gathered = self._gathered[: w * k]
            all_gather_equal(gathered, x, self.group)        # replaces hugewiki.cu:2744-2745
            self.thetaT[off: off + size].copy_(gathered[:size])

        overlap = len(self.t_batches) > 1
        for bi, (off, size, plan) in enumerate(self.t_batches):
            slot = bi % self._nbuf
            k = (size + w - 1) // w
            rhs, tri = self._rhs[slot][: w * k], self._tri[slot][: w * k]
            # partial Gram / RHS over this rank's X slab (hugewiki.cu:2668-2679), as packed upper triangles
            self.ops.get_hermitian_packed(plan, self.lc_rowidx, self.lc_val, self.XT, self.lam, tri[:size], rhs[:size])
            if size < w * k:  # the padding systems behind the last rank's share (a few rows, not the buffer)
                tri[size:].zero_()
                rhs[size:].zero_()
            if pending is not None:
                finish(*pending)
                pending = None
            _, w1 = reduce_scatter_rows(tri, self.group, out=self._mine[slot][:k], async_op=overlap)   # hugewiki.cu:2703-2717
            _, w2 = reduce_scatter_rows(rhs, self.group, out=self._mine_rhs[slot][:k], async_op=overlap)  # hugewiki.cu:2719-2730
            pending = (bi, slot, (w1, w2))
        if pending is not None:
            finish(*pending)
        if quad is None:
            return None
        t = terms
        if dist.is_initialized() and self.world > 1:
            if dist.get_backend() != "nccl":
                t = t.cpu()
            dist.all_reduce(t, group=self.group)
        return self._trusted_sse(s_total - float(t.item()))

    # -- RMSE (hugewiki.cu:2750-2862: per-GPU SSE over its slab, summed) -------------------------
    def slab_sse(self, val: torch.Tensor, row_local: torch.Tensor, col: torch.Tensor) -> float:
        """Sum over ALL ranks of the squared errors of each rank's ratings; `row_local` indexes the
        rank's own X slab (row ids rebased to the slab), `col` is global."""
        if self.scheme == "reduce":
            mine = self.XT
        else:
            mine = self.XT[int(self.xb[self.rank]):int(self.xb[self.rank + 1])]
        t = torch.tensor([self.ops.sse(val, row_local, col, self.thetaT, mine)], dtype=torch.float64)
        if dist.is_initialized() and self.world > 1:
            if dist.get_backend() == "nccl":
                t = t.to(self.XT.device)
            dist.all_reduce(t, group=self.group)
        return float(t.item())

    def iterate(self, iters: int = 1) -> None:
        for _ in range(iters):
            self.update_x()
            self.update_theta()
        check = getattr(self.ops, "check", None)  # gram mode "fast": range report of the HIP ops
        if check is not None:
            check()

    def close(self) -> None:
        """Destroy the plans (the side plans, the pipeline pieces', the Theta batches') and hand the library's pooled scratch
        of this device back (tile buffers of the f >= 144 LU path, pre-split tables of gram mode "fast": up to 48 GiB
        outside torch's caching allocator)."""
        for name in ("_nx", "_nt", "_nr", "_ncomm"):  # native objects before the plans they point at
            obj = getattr(self, name, None)
            if obj is not None:
                obj.close()
                setattr(self, name, None)
        plans = [getattr(self, "x_plan", None), getattr(self, "t_plan", None)]
        for pipe in (getattr(self, "_x_pipe", None), getattr(self, "_t_pipe", None)):
            if pipe is not None:
                plans += [p for (_, _, p) in pipe[1]]
        plans += [p for (_, _, p) in getattr(self, "t_batches", [])]
        for p in plans:
            if p is not None and hasattr(p, "close"):
                p.close()
        self.x_plan = self.t_plan = None
        self._x_pipe = self._t_pipe = None
        self.t_batches = []
        release = getattr(self.ops, "release_scratch", None)
        if release is not None:
            release()
