"""Data tooling around the cumf_als on-disk format.

Two commands:

``python -m cumf_als_amd.convert text TRAIN.txt TEST.txt OUT_DIR [--order col-row|row-col] [--base 1]``
    Coordinate text (``%``-comment lines, one ``dim dim nnz`` size line, then one triplet per
    line -- the layout of the GraphLab ``netflix_mm`` / ``netflix_mme`` files that
    prepare_netflix_data.py:39-58 consumes) -> the ten ``*.bin`` files of main.cpp:91-103.
    ``--order col-row`` (default) is the reference's reading of those files: the first field is
    the column (user) and the second the row (movie), so Netflix comes out with m = 17 770 rows
    (prepare_netflix_data.py:56-57, 84).  ``coo.row`` is written as the CSR row expansion (the
    order `RMSE` pairs it with the CSR arrays, als.cu:196-198), which is what the reference's
    script produces for row-sorted input and what it should have produced for ml10M.

``python -m cumf_als_amd.convert split DATA_DIR OUT_DIR --gpus G --m M --n N --nnz NNZ --nnz-test T``
    Cut the rows into G contiguous nnz-balanced slabs and write, per GPU g, the slab's CSR
    (row pointer rebased to 0) and the slab's CSC with slab-LOCAL row ids -- the pre-split
    ``..._R_train_csc.{data,indices,indptr}.bin<g>`` inputs hugewiki.cu:2332-2354 loads but
    the reference has no producer for -- plus ``slabs.txt`` with the row boundaries.  Row
    pointers of a slab fit 4 bytes as long as the slab has < 2^31 ratings (hugewiki on 8
    GPUs: 388 M per slab).
"""
from __future__ import annotations

import argparse
import os

import numpy as np

from . import datagen
from .dist import balanced_slabs, local_csc_of_slab, slice_csr


def read_coordinate_text(path: str, order: str = "col-row", base: int = 1):
    """Parse a coordinate text file -> (rows, cols, vals, declared (dim0, dim1, nnz) or None).

    Comment lines start with ``%``; the first remaining line is the size line when its third
    field equals the number of lines after it.  Fields are whitespace separated; values are
    stored as float32 (ratings may be integers or halves)."""
    try:
        import pandas as pd

        df = pd.read_csv(path, sep=r"\s+", header=None, comment="%", names=["a", "b", "v"],
                         dtype={"a": np.int64, "b": np.int64, "v": np.float64})
        a, b, v = df["a"].to_numpy(), df["b"].to_numpy(), df["v"].to_numpy()
    except ImportError:  # pragma: no cover - pandas is part of the image
        arr = np.loadtxt(path, comments="%", ndmin=2)
        a, b, v = arr[:, 0].astype(np.int64), arr[:, 1].astype(np.int64), arr[:, 2]
    header = None
    if len(v) and v[0] == len(v) - 1:  # "dim dim nnz" size line: its count matches the lines that follow
        header = (int(a[0]), int(b[0]), int(v[0]))
        a, b, v = a[1:], b[1:], v[1:]
    v = v.astype(np.float32)
    if order == "col-row":
        cols, rows = a - base, b - base
        dims = (header[1], header[0]) if header else None
    elif order == "row-col":
        rows, cols = a - base, b - base
        dims = (header[0], header[1]) if header else None
    else:
        raise ValueError(f"unknown field order {order!r}")
    if len(v) and (rows.min() < 0 or cols.min() < 0):
        raise ValueError(f"{path}: index below the base {base}")
    return rows, cols, v, (dims + (header[2],) if header else None)


def convert_text(train_path: str, test_path: str, out_dir: str, order: str = "col-row", base: int = 1,
                 m: int | None = None, n: int | None = None) -> datagen.Ratings:
    tr_r, tr_c, tr_v, tr_hdr = read_coordinate_text(train_path, order, base)
    te_r, te_c, te_v, _ = read_coordinate_text(test_path, order, base)
    if m is None:
        m = tr_hdr[0] if tr_hdr else int(max(tr_r.max(initial=-1), te_r.max(initial=-1))) + 1
    if n is None:
        n = tr_hdr[1] if tr_hdr else int(max(tr_c.max(initial=-1), te_c.max(initial=-1))) + 1
    for name, r, c in (("train", tr_r, tr_c), ("test", te_r, te_c)):
        if len(r) and (r.max() >= m or c.max() >= n):
            raise ValueError(f"{name} set has an index outside {m} x {n}")
    key = tr_r.astype(np.int64) * n + tr_c
    if len(np.unique(key)) != len(key):
        # scipy's coo->csr (prepare_netflix_data.py:90) would sum duplicates; refuse instead of guessing
        raise ValueError("training set holds duplicate (row, col) pairs")
    ratings = datagen.from_coo(m, n, tr_r, tr_c, tr_v, te_r, te_c, te_v)
    datagen.write_dataset(ratings, out_dir)
    return ratings


TEST_SLAB_FILES = {
    "test_data": ("R_test_coo.data.bin", np.float32),
    "test_row": ("R_test_coo.row.bin", np.int32),   # slab-local row ids
    "test_col": ("R_test_coo.col.bin", np.int32),
}

SLAB_FILES = {
    "csr_data": ("R_train_csr.data.bin", np.float32),
    "csr_indptr": ("R_train_csr.indptr.bin", np.int32),
    "csr_indices": ("R_train_csr.indices.bin", np.int32),
    "csc_data": ("R_train_csc.data.bin", np.float32),
    "csc_indices": ("R_train_csc.indices.bin", np.int32),
    "csc_indptr": ("R_train_csc.indptr.bin", np.int32),
}


def widen_indptr(indptr32: np.ndarray, nnz: int) -> np.ndarray:
    """4-byte on-disk row pointers -> int64.  With nnz >= 2^31 the file holds the values modulo 2^32
    (hugewiki.cu:1973,1984 reads them as unsigned; nnz = 3.1 G wraps once): reinterpret as uint32 and
    add 2^32 at every wrap (row pointers are non-decreasing).  Raises if the result does not end at nnz."""
    u = np.ascontiguousarray(indptr32).view(np.uint32).astype(np.int64)
    if nnz >= 2 ** 31:
        wraps = np.concatenate([[0], np.cumsum(np.diff(u) < 0)])
        u = u + (wraps.astype(np.int64) << 32)
    if u[0] != 0 or u[-1] != nnz or (np.diff(u) < 0).any():
        raise ValueError(f"row pointers do not describe {nnz} ratings (first {u[0]}, last {u[-1]}): "
                         "4-byte indptr overflow or wrong nnz")
    return u


def split_dataset(data_dir: str, out_dir: str, gpus: int, m: int, n: int, nnz: int, nnz_test: int):
    """Write per-GPU slab files `<name><g>` (g = 0..gpus-1) and `slabs.txt`; returns the bounds."""
    d = datagen.read_dataset(data_dir, m, n, nnz, nnz_test)
    rowptr = widen_indptr(d["csr_indptr"], nnz)
    bounds = balanced_slabs(rowptr, gpus)
    os.makedirs(out_dir, exist_ok=True)
    for g in range(gpus):
        r0, r1 = int(bounds[g]), int(bounds[g + 1])
        rp, ci, va = slice_csr(rowptr, d["csr_indices"], d["csr_data"], r0, r1)
        if rp[-1] >= 2 ** 31:
            raise ValueError(f"slab {g} has {rp[-1]} ratings: 4-byte row pointers overflow, use more GPUs")
        cp, ri, cv = local_csc_of_slab(rp, ci, va, n)
        arrays = {"csr_data": va, "csr_indptr": rp, "csr_indices": ci,
                  "csc_data": cv, "csc_indices": ri, "csc_indptr": cp}
        # the test ratings whose row falls in the slab, row ids rebased (the reference keeps the
        # test set as per-GPU CSC files, hugewiki.cu:2345-2354; COO is what the SSE kernel reads)
        sel = np.nonzero((d["test_row"] >= r0) & (d["test_row"] < r1))[0]
        arrays.update(test_data=d["test_data"][sel], test_row=d["test_row"][sel] - r0, test_col=d["test_col"][sel])
        for key, (name, dtype) in {**SLAB_FILES, **TEST_SLAB_FILES}.items():
            np.ascontiguousarray(arrays[key], dtype=dtype).tofile(os.path.join(out_dir, f"{name}{g}"))
    with open(os.path.join(out_dir, "slabs.txt"), "w") as fh:
        fh.write(" ".join(str(int(b)) for b in bounds) + "\n")
    return bounds


def read_slab(out_dir: str, g: int, rows: int, n: int):
    """Load slab g written by `split_dataset` (rows = its row count) -> dict of numpy arrays."""
    out = {}
    for key, (name, dtype) in {**SLAB_FILES, **TEST_SLAB_FILES}.items():
        out[key] = np.fromfile(os.path.join(out_dir, f"{name}{g}"), dtype=dtype)
    if out["csr_indptr"].size != rows + 1 or out["csc_indptr"].size != n + 1:
        raise ValueError(f"slab {g}: row pointer sizes do not match rows={rows}, n={n}")
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    t = sub.add_parser("text", help="coordinate text -> the ten .bin files")
    t.add_argument("train")
    t.add_argument("test")
    t.add_argument("out_dir")
    t.add_argument("--order", choices=["col-row", "row-col"], default="col-row")
    t.add_argument("--base", type=int, default=1)
    t.add_argument("--m", type=int)
    t.add_argument("--n", type=int)
    s = sub.add_parser("split", help="per-GPU row slabs with slab-local CSC")
    s.add_argument("data_dir")
    s.add_argument("out_dir")
    s.add_argument("--gpus", type=int, required=True)
    for k in ("m", "n", "nnz"):
        s.add_argument(f"--{k}", type=int, required=True)
    s.add_argument("--nnz-test", type=int, required=True)
    a = ap.parse_args(argv)
    if a.cmd == "text":
        r = convert_text(a.train, a.test, a.out_dir, a.order, a.base, a.m, a.n)
        print(f"wrote {a.out_dir}: M={r.m} N={r.n} NNZ={r.nnz} NNZ_TEST={r.nnz_test}")
    else:
        b = split_dataset(a.data_dir, a.out_dir, a.gpus, a.m, a.n, a.nnz, a.nnz_test)
        print(f"wrote {a.out_dir}: slab row boundaries {list(map(int, b))}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
