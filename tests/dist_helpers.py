"""Stand-in compute ops for the multi-process CPU tests of cumf_als_amd.dist.

TEST INFRASTRUCTURE: the product's ops are `cumf_als_amd.dist.HipOps` (HIP kernels).  There
is no GPU in the CPU test tier, so the partition + collective logic of DistALS is exercised
with the CPU oracle doing the per-rank arithmetic.  Never imported by the package.
"""
import numpy as np
import torch

from oracle import pyoracle


class _Plan:
    def __init__(self, rowptr, f, row_begin, row_end):
        self.rowptr = np.ascontiguousarray(rowptr).astype(np.int32)
        self.f = f
        self.row_begin = row_begin
        self.row_end = len(rowptr) - 1 if row_end is None else row_end


class OracleOps:
    device = torch.device("cpu")

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def plan(self, rowptr, f, chunk=0, row_begin=0, row_end=None):
        return _Plan(rowptr, f, row_begin, row_end)

    def update_fused(self, plan, colidx, val, gather, update, lam, solver, cg_iters):
        b, e = plan.row_begin, plan.row_end   # a plan may cover a sub-range of the rows (batches, pipeline pieces)
        rp = plan.rowptr[b:e + 1] - plan.rowptr[b]
        s0, s1 = int(plan.rowptr[b]), int(plan.rowptr[e])
        x = np.ascontiguousarray(update.numpy()[b:e], np.float32)
        pyoracle.half_iteration(np.ascontiguousarray(rp), colidx.numpy()[s0:s1], val.numpy()[s0:s1], gather.numpy(), x,
                                plan.f, lam, solver=solver, cg_iters=cg_iters)
        update[b:e].copy_(torch.from_numpy(x))

    def get_hermitian(self, plan, colidx, val, gather, lam, tt, rhs):
        A, b = pyoracle.gram_rhs(plan.rowptr, colidx.numpy(), val.numpy(), gather.numpy(), plan.f, lam,
                                 plan.row_begin, plan.row_end)
        tt.copy_(torch.from_numpy(A))
        rhs.copy_(torch.from_numpy(b))

    def get_hermitian_packed(self, plan, colidx, val, gather, lam, packed, rhs):
        A, b = pyoracle.gram_rhs(plan.rowptr, colidx.numpy(), val.numpy(), gather.numpy(), plan.f, lam,
                                 plan.row_begin, plan.row_end)
        iu = np.triu_indices(plan.f)
        packed.copy_(torch.from_numpy(np.ascontiguousarray(A[:, iu[0], iu[1]])))
        rhs.copy_(torch.from_numpy(b))

    def solve(self, tt, rhs, x, solver, cg_iters):
        f = rhs.shape[-1]
        if solver in ("cg", 0):
            out = pyoracle.cg(tt.numpy(), x.numpy(), rhs.numpy(), f, cg_iters)
        else:
            out = pyoracle.lu(tt.numpy(), rhs.numpy(), f)
        x.copy_(torch.from_numpy(np.ascontiguousarray(out, np.float32)))


    def quad_terms(self, tt, rhs, x, reg, acc):
        """acc += sum over the systems of 2 x.b - x^T A x + reg |x|^2 in fp64 (stand-in of cumf_quadratic_sse_terms;
        reg < 0 marks a system without ratings)."""
        A, b, t, rg = (v.numpy().astype(np.float64) for v in (tt, rhs, x, reg))
        keep = rg >= 0
        q = 2.0 * (t * b).sum(1) - np.einsum("bi,bij,bj->b", t, A, t) + rg * (t * t).sum(1)
        acc += float(q[keep].sum())

    def pack_upper(self, full, packed):
        f = full.shape[-1]
        iu = np.triu_indices(f)
        packed.copy_(torch.from_numpy(np.ascontiguousarray(full.numpy()[:, iu[0], iu[1]])))

    def unpack_upper(self, packed, full):
        f = full.shape[-1]
        iu = np.triu_indices(f)
        out = np.zeros((packed.shape[0], f, f), np.float32)
        out[:, iu[0], iu[1]] = packed.numpy()
        out[:, iu[1], iu[0]] = packed.numpy()
        full.copy_(torch.from_numpy(out))

    def sse(self, val, row, col, thetaT, XT):
        if val.numel() == 0:
            return 0.0
        return float(pyoracle.sse(val.numpy(), row.numpy(), col.numpy(), thetaT.numpy(), XT.numpy(),
                                  val.numel(), thetaT.shape[-1]))


class OracleSseOps(OracleOps):
    """OracleOps + the fused train SSE of the `gather` scheme (stand-in of cumf_als_update_fused_sse /
    cumf_fused_sse_available); `available` is this rank's answer for its own plans."""

    def __init__(self, available=True):
        self.available = available

    def fused_sse_available(self, plan, solver):
        return self.available

    def update_fused_sse(self, plan, colidx, val, gather, update, lam, solver, cg_iters, bins):
        assert self.available, "a rank whose plans cannot deliver the fused SSE must not be asked for it"
        self.update_fused(plan, colidx, val, gather, update, lam, solver, cg_iters)
        b, e = plan.row_begin, plan.row_end
        rp = plan.rowptr.astype(np.int64)
        rows = np.repeat(np.arange(b, e), np.diff(rp[b:e + 1]))
        s0, s1 = int(rp[b]), int(rp[e])
        pred = (update.numpy()[rows].astype(np.float64) * gather.numpy()[colidx.numpy()[s0:s1]].astype(np.float64)).sum(1)
        bins[0] += float(((val.numpy()[s0:s1].astype(np.float64) - pred) ** 2).sum())


def gather_sse_worker(rank, world, port, solver, d, m, n, f, lam, theta0, unavailable_rank, q):
    """One rank of the `gather` scheme asking for the train SSE out of the Theta update; rank `unavailable_rank`'s plans
    refuse it (-1: none does).  A refused request must come back as None on EVERY rank, and the fallback `slab_sse` (one
    all-reduce of its own) must then pair up across the ranks."""
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cumf_als_amd import dist as cdist

        mat = cdist.HostMatrix(m, n, d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indptr"],
                               d["csc_indices"], d["csc_data"])
        eng = cdist.DistALS(mat, f, lam, OracleSseOps(available=rank != unavailable_rank), solver=solver, cg_iters=6,
                            scheme="gather")
        eng.init_factors(theta0)
        eng.iterate(1)
        eng.update_x()
        fused = eng.update_theta(train_sse=True)
        x0, x1 = int(eng.xb[rank]), int(eng.xb[rank + 1])
        rp = np.asarray(d["csr_indptr"], np.int64)
        rows = np.repeat(np.arange(x1 - x0), np.diff(rp[x0:x1 + 1]))
        sl = slice(int(rp[x0]), int(rp[x1]))
        direct = eng.slab_sse(torch.from_numpy(np.ascontiguousarray(d["csr_data"][sl])), torch.from_numpy(rows),
                              torch.from_numpy(np.ascontiguousarray(d["csr_indices"][sl])))
        q.put((rank, fused, direct))
    finally:
        dist.destroy_process_group()


def worker(rank, world, port, scheme, solver, d, m, n, f, lam, iters, theta_batch, theta0, q, ops_kind="oracle",
           side_solvers=None):
    """Entry point of one rank (spawned): runs DistALS and returns full factors through `q`."""
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cumf_als_amd import dist as cdist

        if ops_kind == "hip":
            torch.cuda.set_device(0)
            ops = cdist.HipOps("cuda:0")
        else:
            ops = OracleOps()
        mat = cdist.HostMatrix(m, n, d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indptr"],
                               d["csc_indices"], d["csc_data"])
        eng = cdist.DistALS(mat, f, lam, ops, solver=solver, cg_iters=6, scheme=scheme, theta_batch=theta_batch,
                            **(side_solvers or {}))
        eng.init_factors(theta0)
        eng.iterate(iters)
        x = eng.full_XT().cpu().numpy().copy()
        th = eng.thetaT.cpu().numpy().copy()
        q.put((rank, th, x))
    finally:
        dist.destroy_process_group()


def hugewiki_worker(rank, world, port, split_dir, n, f, lam, iters, solver, q):
    """One rank of cumf_als_amd.hugewiki.run on CPU (gloo) with the oracle stand-in ops."""
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cumf_als_amd import hugewiki

        eng, log = hugewiki.run(split_dir, n, f, lam, iters, solver=solver, ops=OracleOps(), quiet=True)
        q.put((rank, eng.thetaT.numpy().copy(), eng.full_XT().numpy().copy(), log))
    finally:
        dist.destroy_process_group()


def train_sse_worker(rank, world, port, solver, d, m, n, f, lam, theta_batch, theta0, q):
    """One rank: two iterations of the `reduce` scheme, then one more Theta update that also returns the train SSE from the
    reduced systems (DistALS.update_theta(train_sse=True)); the full factors go back for the direct evaluation."""
    import os

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cumf_als_amd import dist as cdist

        mat = cdist.HostMatrix(m, n, d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indptr"],
                               d["csc_indices"], d["csc_data"])
        eng = cdist.DistALS(mat, f, lam, OracleOps(), solver=solver, cg_iters=6, scheme="reduce", theta_batch=theta_batch)
        eng.init_factors(theta0)
        eng.iterate(2)
        eng.update_x()
        sse = eng.update_theta(train_sse=True)
        q.put((rank, sse, eng.thetaT.numpy().copy(), eng.full_XT().numpy().copy()))
    finally:
        dist.destroy_process_group()
