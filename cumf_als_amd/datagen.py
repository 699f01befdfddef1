"""Synthetic ratings in the reference's on-disk format.

The reference consumes ten headerless little-endian 4-byte-element files
(`main.cpp:91-103`, read by `host_utilities.cpp:19-97`), produced upstream by
scipy `coo_matrix -> tocsr()/tocsc() -> ndarray.tofile`
(`data/netflix/prepare_netflix_data.py:84-105`):

    R_train_csr.{data,indptr,indices}.bin   f32[nnz] i32[m+1] i32[nnz]
    R_train_csc.{data,indices,indptr}.bin   f32[nnz] i32[nnz](row ids) i32[n+1]
    R_train_coo.row.bin                     i32[nnz]  (row of the i-th CSR entry)
    R_test_coo.{data,row,col}.bin           f32/i32/i32[nnz_test]

The real datasets (Netflix, MovieLens, hugewiki) are not available offline, so
this module generates ratings of the same shape: power-law row and column
popularity, no duplicate (row, col), integer ratings 1..5 stored as fp32 from a
planted low-rank model so that RMSE decreases meaningfully.  `R_train_coo.row`
is always emitted as the CSR row expansion, which is what `RMSE` in the
reference silently assumes (`als.cu:196-198`).

Generation runs in torch so the Netflix / hugewiki shapes can be produced on the
GPU in seconds; the result is deterministic for a given (seed, device type).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import torch

FILES = {
    "csr_data": ("R_train_csr.data.bin", np.float32),
    "csr_indptr": ("R_train_csr.indptr.bin", np.int32),
    "csr_indices": ("R_train_csr.indices.bin", np.int32),
    "csc_data": ("R_train_csc.data.bin", np.float32),
    "csc_indices": ("R_train_csc.indices.bin", np.int32),
    "csc_indptr": ("R_train_csc.indptr.bin", np.int32),
    "coo_row": ("R_train_coo.row.bin", np.int32),
    "test_data": ("R_test_coo.data.bin", np.float32),
    "test_row": ("R_test_coo.row.bin", np.int32),
    "test_col": ("R_test_coo.col.bin", np.int32),
}

# Shapes named in BASELINE.json / main.cpp:24-28 / hugewiki.cu:33-38.
SHAPES = {
    "ml10m": dict(m=71567, n=65133, nnz=9000048, nnz_test=1000006, lam=0.05),
    "netflix": dict(m=17770, n=480189, nnz=99072112, nnz_test=1408395, lam=0.048),
    "hugewiki": dict(m=50082603, n=39780, nnz=3101144313, nnz_test=344573330, lam=0.048),
}


@dataclass
class Ratings:
    """Train matrix as CSR + CSC (+ COO row expansion) and a COO test set.

    All tensors live on one device; indices are int32 (row pointers int32 when
    nnz < 2**31, else int64), values fp32.
    """

    m: int
    n: int
    csr_indptr: torch.Tensor
    csr_indices: torch.Tensor
    csr_data: torch.Tensor
    csc_indptr: torch.Tensor
    csc_indices: torch.Tensor
    csc_data: torch.Tensor
    coo_row: torch.Tensor
    test_row: torch.Tensor
    test_col: torch.Tensor
    test_data: torch.Tensor

    @property
    def nnz(self) -> int:
        return int(self.csr_indices.numel())

    @property
    def nnz_test(self) -> int:
        return int(self.test_row.numel())

    def to(self, device) -> "Ratings":
        kw = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.__dict__.items()}
        return Ratings(**kw)

    def numpy(self) -> dict:
        out = {}
        for k in FILES:
            out[k] = getattr(self, k).cpu().numpy()
        return out


def _powerlaw_cdf(count: int, alpha: float, gen: torch.Generator, device) -> torch.Tensor:
    """CDF of a shuffled Zipf-like popularity p_i ~ (i + offset)^-alpha."""
    ranks = torch.arange(1, count + 1, dtype=torch.float64, device=device)
    w = (ranks + 0.02 * count) ** (-alpha)
    perm = torch.randperm(count, generator=gen, device=device)
    w = w[perm]
    cdf = torch.cumsum(w, 0)
    return cdf / cdf[-1]


def _sample_pairs(m, n, count, cdf_r, cdf_c, gen, device):
    u = torch.rand(count, generator=gen, device=device, dtype=torch.float64)
    rows = torch.searchsorted(cdf_r, u).clamp_(max=m - 1)
    u = torch.rand(count, generator=gen, device=device, dtype=torch.float64)
    cols = torch.searchsorted(cdf_c, u).clamp_(max=n - 1)
    return rows * n + cols


def synth_ratings(m: int, n: int, nnz: int, nnz_test: int, seed: int = 0, *,
                  row_alpha: float = 0.9, col_alpha: float = 0.9, rank: int = 8,
                  noise: float = 0.4, ensure_nonempty: bool = True,
                  device: str | torch.device = "cpu", col_seed: int | None = None) -> Ratings:
    """Generate `nnz` train + `nnz_test` test ratings of an m x n matrix.

    Row/column degrees follow shuffled power laws (alpha = 0 gives the uniform
    control).  `ensure_nonempty` plants one train rating in every row and every
    column so that no normal equation is singular (an empty row makes the
    reference produce NaN factors, `cg.cu:128`).
    """
    device = torch.device(device)
    total = nnz + nnz_test
    if total > m * n:
        raise ValueError("more ratings requested than matrix cells")
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    cdf_r = _powerlaw_cdf(m, row_alpha, gen, device)
    if col_seed is None:
        cdf_c = _powerlaw_cdf(n, col_alpha, gen, device)
    else:  # row slabs of one matrix generated by different ranks share the column popularity
        cgen = torch.Generator(device=device)
        cgen.manual_seed(col_seed)
        cdf_c = _powerlaw_cdf(n, col_alpha, cgen, device)

    forced = torch.empty(0, dtype=torch.int64, device=device)
    if ensure_nonempty:
        if nnz < m + n:
            raise ValueError("nnz too small to make every row and column non-empty")
        r = torch.arange(m, device=device, dtype=torch.int64)
        c = torch.randint(0, n, (m,), generator=gen, device=device)
        f1 = r * n + c
        c = torch.arange(n, device=device, dtype=torch.int64)
        r = torch.randint(0, m, (n,), generator=gen, device=device)
        f2 = r * n + c
        forced = torch.unique(torch.cat([f1, f2]))

    keys = forced
    # draw, dedupe, top up until enough distinct cells exist
    need = total
    rounds = 0
    while keys.numel() < need:
        missing = need - keys.numel()
        draw = int(missing * 1.25) + 1024
        new = _sample_pairs(m, n, draw, cdf_r, cdf_c, gen, device)
        keys = torch.unique(torch.cat([keys, new]))
        rounds += 1
        if rounds > 200:
            raise RuntimeError("synthetic generator failed to reach the requested nnz")
    # choose which distinct cells are kept / go to the test set (forced cells stay in train)
    is_forced = torch.zeros(keys.numel(), dtype=torch.bool, device=device)
    if forced.numel():
        is_forced[torch.searchsorted(keys, forced)] = True
    score = torch.rand(keys.numel(), generator=gen, device=device)
    score[is_forced] = -1.0  # smallest scores are kept as train
    order = torch.argsort(score)
    train_sel = order[:nnz]
    test_sel = order[nnz:nnz + nnz_test]
    train_keys = torch.sort(keys[train_sel]).values  # row-major order == CSR order
    test_keys = keys[test_sel]  # random order, like a shuffled split

    # planted low-rank model -> ratings in {1..5}
    gx = torch.randn(m, rank, generator=gen, device=device) / rank ** 0.25
    if col_seed is None:
        gt = torch.randn(n, rank, generator=gen, device=device) / rank ** 0.25
    else:
        gt = torch.randn(n, rank, generator=cgen, device=device) / rank ** 0.25

    def rate(k):
        rr = torch.div(k, n, rounding_mode="floor")
        cc = k - rr * n
        out = torch.empty(k.numel(), dtype=torch.float32, device=device)
        step = 1 << 24
        for s in range(0, k.numel(), step):
            sl = slice(s, min(s + step, k.numel()))
            dot = (gx[rr[sl]] * gt[cc[sl]]).sum(1)
            eps = torch.randn(dot.numel(), generator=gen, device=device) * noise
            out[sl] = torch.clamp(torch.round(3.0 + 1.2 * dot + eps), 1.0, 5.0)
        return rr, cc, out

    tr_r, tr_c, tr_v = rate(train_keys)
    te_r, te_c, te_v = rate(test_keys)

    ptr_dtype = torch.int32 if nnz < 2 ** 31 else torch.int64
    csr_indptr = torch.zeros(m + 1, dtype=torch.int64, device=device)
    csr_indptr[1:] = torch.cumsum(torch.bincount(tr_r, minlength=m), 0)
    # CSC: stable sort by column keeps rows ascending inside a column (scipy tocsc order)
    csc_order = torch.argsort(tr_c * m + tr_r)
    csc_indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    csc_indptr[1:] = torch.cumsum(torch.bincount(tr_c, minlength=n), 0)
    return Ratings(
        m=m, n=n,
        csr_indptr=csr_indptr.to(ptr_dtype), csr_indices=tr_c.to(torch.int32), csr_data=tr_v,
        csc_indptr=csc_indptr.to(ptr_dtype), csc_indices=tr_r[csc_order].to(torch.int32),
        csc_data=tr_v[csc_order],
        coo_row=tr_r.to(torch.int32),
        test_row=te_r.to(torch.int32), test_col=te_c.to(torch.int32), test_data=te_v,
    )


def write_dataset(r: Ratings, data_dir: str) -> None:
    """Write the ten files exactly as the reference's producers do (`ndarray.tofile`)."""
    if r.nnz >= 2 ** 31:
        raise ValueError("on-disk indptr is 4-byte; shard the matrix (hugewiki.cu:2332-2340 style) first")
    os.makedirs(data_dir, exist_ok=True)
    arrays = r.numpy()
    for key, (name, dtype) in FILES.items():
        np.ascontiguousarray(arrays[key], dtype=dtype).tofile(os.path.join(data_dir, name))


def read_dataset(data_dir: str, m: int, n: int, nnz: int, nnz_test: int) -> dict:
    """Read the ten files back into numpy arrays (sizes are not stored in the files)."""
    counts = {
        "csr_data": nnz, "csr_indptr": m + 1, "csr_indices": nnz,
        "csc_data": nnz, "csc_indices": nnz, "csc_indptr": n + 1,
        "coo_row": nnz, "test_data": nnz_test, "test_row": nnz_test, "test_col": nnz_test,
    }
    out = {}
    for key, (name, dtype) in FILES.items():
        path = os.path.join(data_dir, name)
        arr = np.fromfile(path, dtype=dtype)
        if arr.size != counts[key]:
            raise ValueError(f"{path}: expected {counts[key]} elements, found {arr.size}")
        out[key] = arr
    return out


def from_coo(m: int, n: int, rows, cols, vals, test_rows, test_cols, test_vals) -> Ratings:
    """Build a `Ratings` from explicit COO triplets (used for hand-written fixtures)."""
    rows = torch.as_tensor(np.asarray(rows), dtype=torch.int64)
    cols = torch.as_tensor(np.asarray(cols), dtype=torch.int64)
    vals = torch.as_tensor(np.asarray(vals), dtype=torch.float32)
    order = torch.argsort(rows * n + cols)
    rows, cols, vals = rows[order], cols[order], vals[order]
    csr_indptr = torch.zeros(m + 1, dtype=torch.int64)
    csr_indptr[1:] = torch.cumsum(torch.bincount(rows, minlength=m), 0)
    co = torch.argsort(cols * m + rows)
    csc_indptr = torch.zeros(n + 1, dtype=torch.int64)
    csc_indptr[1:] = torch.cumsum(torch.bincount(cols, minlength=n), 0)
    return Ratings(
        m=m, n=n, csr_indptr=csr_indptr.to(torch.int32), csr_indices=cols.to(torch.int32),
        csr_data=vals, csc_indptr=csc_indptr.to(torch.int32), csc_indices=rows[co].to(torch.int32),
        csc_data=vals[co], coo_row=rows.to(torch.int32),
        test_row=torch.as_tensor(np.asarray(test_rows), dtype=torch.int32),
        test_col=torch.as_tensor(np.asarray(test_cols), dtype=torch.int32),
        test_data=torch.as_tensor(np.asarray(test_vals), dtype=torch.float32),
    )


def main(argv=None) -> int:
    import argparse

    ap = argparse.ArgumentParser(description="write a synthetic dataset in the cumf_als on-disk format")
    ap.add_argument("--shape", choices=sorted(SHAPES), help="named shape (overrides m/n/nnz/nnz_test)")
    ap.add_argument("--m", type=int)
    ap.add_argument("--n", type=int)
    ap.add_argument("--nnz", type=int)
    ap.add_argument("--nnz-test", type=int)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink a named shape (rows, cols, nnz) by this factor")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("out_dir")
    a = ap.parse_args(argv)
    if a.shape:
        s = SHAPES[a.shape]
        m, n = max(2, int(s["m"] * a.scale)), max(2, int(s["n"] * a.scale))
        nnz, nnz_test = int(s["nnz"] * a.scale ** 2), int(s["nnz_test"] * a.scale ** 2)
        nnz = max(nnz, m + n)
    else:
        m, n, nnz, nnz_test = a.m, a.n, a.nnz, a.nnz_test
    r = synth_ratings(m, n, nnz, nnz_test, a.seed, device=a.device)
    write_dataset(r, a.out_dir)
    print(f"wrote {a.out_dir}: M={m} N={n} NNZ={r.nnz} NNZ_TEST={r.nnz_test}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
