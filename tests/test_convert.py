"""Data tooling: coordinate text -> .bin files, per-GPU slab split."""
import numpy as np
import pytest

from cumf_als_amd import convert, datagen
from cumf_als_amd.dist import balanced_slabs


def _write_text(path, triplets, dims=None, comment=True):
    with open(path, "w") as fh:
        if comment:
            fh.write("% Generated for a test\n")
        if dims:
            fh.write("%d %d %d\n" % dims)
        for a, b, v in triplets:
            fh.write("%d %d  %g\n" % (a, b, v))


def test_text_to_bin_matches_scipy_producer(tmp_path):
    """Same arrays as prepare_netflix_data.py:56-105 issues through scipy on the same triplets
    (file order 'user movie rating', 1-based; rows = movies)."""
    from scipy.sparse import coo_matrix

    rng = np.random.default_rng(3)
    users, movies = 40, 13
    cells = rng.choice(users * movies, size=170, replace=False)
    cells.sort()  # the GraphLab files are sorted by movie then user: movie-major
    mv, us = cells // users, cells % users
    val = rng.integers(1, 6, size=cells.size)
    train = list(zip(us[:150] + 1, mv[:150] + 1, val[:150]))
    test = list(zip(us[150:] + 1, mv[150:] + 1, val[150:]))
    _write_text(tmp_path / "mm", train, dims=(users, movies, len(train)))
    _write_text(tmp_path / "mme", test, dims=(users, movies, len(test)))
    r = convert.convert_text(str(tmp_path / "mm"), str(tmp_path / "mme"), str(tmp_path / "out"))
    assert (r.m, r.n, r.nnz, r.nnz_test) == (movies, users, 150, 20)

    j, i, rating = np.array(train).T
    coo = coo_matrix((rating, (i - 1, j - 1)), shape=(movies, users))
    csr, csc = coo.tocsr(), coo.tocsc()
    d = datagen.read_dataset(str(tmp_path / "out"), movies, users, 150, 20)
    np.testing.assert_array_equal(d["csr_indptr"], csr.indptr)
    np.testing.assert_array_equal(d["csr_indices"], csr.indices)
    np.testing.assert_array_equal(d["csr_data"], csr.data.astype(np.float32))
    np.testing.assert_array_equal(d["csc_indptr"], csc.indptr)
    np.testing.assert_array_equal(d["csc_indices"], csc.indices)
    np.testing.assert_array_equal(d["csc_data"], csc.data.astype(np.float32))
    np.testing.assert_array_equal(d["coo_row"], coo.row)  # input was row-sorted: same as the CSR expansion
    tj, ti, tr = np.array(test).T
    np.testing.assert_array_equal(d["test_row"], ti - 1)
    np.testing.assert_array_equal(d["test_col"], tj - 1)
    np.testing.assert_array_equal(d["test_data"], tr.astype(np.float32))


def test_text_without_header_and_row_col_order(tmp_path):
    train = [(0, 1, 4.5), (2, 0, 3), (1, 1, 1)]
    test = [(2, 1, 5)]
    _write_text(tmp_path / "tr", train, comment=False)
    _write_text(tmp_path / "te", test, comment=False)
    r = convert.convert_text(str(tmp_path / "tr"), str(tmp_path / "te"), str(tmp_path / "o"),
                             order="row-col", base=0)
    assert (r.m, r.n, r.nnz) == (3, 2, 3)
    a = r.numpy()
    np.testing.assert_array_equal(a["csr_indptr"], [0, 1, 2, 3])
    np.testing.assert_array_equal(a["csr_indices"], [1, 1, 0])
    np.testing.assert_array_equal(a["csr_data"], np.float32([4.5, 1, 3]))
    np.testing.assert_array_equal(a["coo_row"], [0, 1, 2])


def test_text_rejects_duplicates_and_out_of_range(tmp_path):
    _write_text(tmp_path / "tr", [(1, 1, 3), (1, 1, 4)], comment=False)
    _write_text(tmp_path / "te", [(1, 1, 3)], comment=False)
    with pytest.raises(ValueError, match="duplicate"):
        convert.convert_text(str(tmp_path / "tr"), str(tmp_path / "te"), str(tmp_path / "o"))
    _write_text(tmp_path / "tr2", [(1, 1, 3), (2, 5, 4)], comment=False)
    with pytest.raises(ValueError, match="outside"):
        convert.convert_text(str(tmp_path / "tr2"), str(tmp_path / "te"), str(tmp_path / "o"), m=2, n=2)


def test_split_slabs_reassemble(tmp_path):
    """The slab files are a partition of the CSR, and each slab's CSC (slab-local row ids) is the
    transpose of that slab -- what hugewiki.cu:2332-2354 expects per GPU."""
    r = datagen.synth_ratings(57, 23, 600, 10, seed=5, device="cpu")
    datagen.write_dataset(r, str(tmp_path / "d"))
    bounds = convert.split_dataset(str(tmp_path / "d"), str(tmp_path / "s"), 3, r.m, r.n, r.nnz, r.nnz_test)
    a = r.numpy()
    np.testing.assert_array_equal(bounds, balanced_slabs(a["csr_indptr"], 3))
    assert open(tmp_path / "s" / "slabs.txt").read().split() == [str(int(b)) for b in bounds]
    dense = np.zeros((r.m, r.n), np.float32)
    rows = np.repeat(np.arange(r.m), np.diff(a["csr_indptr"]))
    dense[rows, a["csr_indices"]] = a["csr_data"]
    for g in range(3):
        r0, r1 = int(bounds[g]), int(bounds[g + 1])
        s = convert.read_slab(str(tmp_path / "s"), g, r1 - r0, r.n)
        assert s["csr_indptr"][0] == 0 and s["csr_indptr"][-1] == s["csr_data"].size
        slab = np.zeros((r1 - r0, r.n), np.float32)
        lr = np.repeat(np.arange(r1 - r0), np.diff(s["csr_indptr"]))
        slab[lr, s["csr_indices"]] = s["csr_data"]
        np.testing.assert_array_equal(slab, dense[r0:r1])
        slab_t = np.zeros_like(slab)
        lc = np.repeat(np.arange(r.n), np.diff(s["csc_indptr"]))
        slab_t[s["csc_indices"], lc] = s["csc_data"]
        np.testing.assert_array_equal(slab_t, slab)
        # rows inside a column are ascending (CSC of a row-sorted matrix)
        for c in range(r.n):
            seg = s["csc_indices"][s["csc_indptr"][c]:s["csc_indptr"][c + 1]]
            assert np.all(np.diff(seg) > 0)


def test_widen_indptr_above_2_31():
    """a global row pointer above 2^31 (hugewiki: 3.1 G ratings) is stored modulo 2^32 in
    the 4-byte file; the splitter must recover it instead of wrapping negative."""
    from cumf_als_amd import convert

    true = np.array([0, 2 ** 31 - 5, 2 ** 31 + 7, 3_101_144_313, 2 ** 32 - 1, 2 ** 32 + 10, 5_000_000_000],
                    dtype=np.int64)
    on_disk = (true % 2 ** 32).astype(np.uint32).view(np.int32)
    assert (on_disk < 0).any()
    got = convert.widen_indptr(on_disk, int(true[-1]))
    np.testing.assert_array_equal(got, true)
    with pytest.raises(ValueError):
        convert.widen_indptr(on_disk, int(true[-1]) + 1)
    small = np.array([0, 3, 3, 10], dtype=np.int32)
    np.testing.assert_array_equal(convert.widen_indptr(small, 10), small.astype(np.int64))
