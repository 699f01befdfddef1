"""On-disk format: byte-for-byte against the reference's own loaders.

oracle/_ref/libref_hostutil.so is compiled (oracle/Makefile) straight from the
reference's host_utilities.cpp -- the one file of the reference that builds without CUDA.
Files written by cumf_als_amd.datagen (same scipy-style `ndarray.tofile` layout as
prepare_netflix_data.py:84-105) must load identically through the reference's loaders,
through this repo's C++ loaders (csrc/host_utilities.cpp) and through datagen.read_dataset.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_hostutil.so")

MANGLED = {
    "csr": "_Z22loadCSRSparseMatrixBinPKcS0_S0_PfPiS2_il",
    "csc": "_Z22loadCSCSparseMatrixBinPKcS0_S0_PfPiS2_il",
    "coo": "_Z22loadCooSparseMatrixBinPKcS0_S0_PfPiS2_l",
    "coorow": "_Z28loadCooSparseMatrixRowPtrBinPKcPil",
}


def _load_with(lib, d, m, n, nnz, nnz_test):
    def p(name):
        return os.path.join(d, name).encode()

    out = {}
    f32 = lambda k: np.zeros(k, np.float32)
    i32 = lambda k: np.zeros(k, np.int32)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    data, row, col = f32(nnz), i32(m + 1), i32(nnz)
    fn = getattr(lib, MANGLED["csr"])
    fn.argtypes = [C.c_char_p] * 3 + [C.c_void_p] * 3 + [C.c_int, C.c_long]
    fn(p("R_train_csr.data.bin"), p("R_train_csr.indptr.bin"), p("R_train_csr.indices.bin"), vp(data), vp(row), vp(col), m, nnz)
    out.update(csr_data=data, csr_indptr=row, csr_indices=col)
    data, row, col = f32(nnz), i32(nnz), i32(n + 1)
    fn = getattr(lib, MANGLED["csc"])
    fn.argtypes = [C.c_char_p] * 3 + [C.c_void_p] * 3 + [C.c_int, C.c_long]
    fn(p("R_train_csc.data.bin"), p("R_train_csc.indices.bin"), p("R_train_csc.indptr.bin"), vp(data), vp(row), vp(col), n, nnz)
    out.update(csc_data=data, csc_indices=row, csc_indptr=col)
    data, row, col = f32(nnz_test), i32(nnz_test), i32(nnz_test)
    fn = getattr(lib, MANGLED["coo"])
    fn.argtypes = [C.c_char_p] * 3 + [C.c_void_p] * 3 + [C.c_long]
    fn(p("R_test_coo.data.bin"), p("R_test_coo.row.bin"), p("R_test_coo.col.bin"), vp(data), vp(row), vp(col), nnz_test)
    out.update(test_data=data, test_row=row, test_col=col)
    row = i32(nnz)
    fn = getattr(lib, MANGLED["coorow"])
    fn.argtypes = [C.c_char_p, C.c_void_p, C.c_long]
    fn(p("R_train_coo.row.bin"), vp(row), nnz)
    out.update(coo_row=row)
    return out


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    from cumf_als_amd import datagen

    r = datagen.synth_ratings(37, 23, 300, 60, seed=9)
    d = str(tmp_path_factory.mktemp("fmt"))
    datagen.write_dataset(r, d)
    return r, d


def test_files_and_sizes(dataset):
    from cumf_als_amd import datagen

    r, d = dataset
    sizes = {"csr_data": r.nnz, "csr_indptr": r.m + 1, "csr_indices": r.nnz, "csc_data": r.nnz, "csc_indices": r.nnz,
             "csc_indptr": r.n + 1, "coo_row": r.nnz, "test_data": r.nnz_test, "test_row": r.nnz_test,
             "test_col": r.nnz_test}
    for key, (name, dtype) in datagen.FILES.items():
        assert os.path.getsize(os.path.join(d, name)) == 4 * sizes[key]  # headerless 4-byte elements
    back = datagen.read_dataset(d, r.m, r.n, r.nnz, r.nnz_test)
    for k, v in r.numpy().items():
        assert np.array_equal(back[k], v)


def test_structure_matches_scipy(dataset):
    """Same arrays scipy's coo_matrix -> tocsr()/tocsc() produce (prepare_netflix_data.py:84-105)."""
    sp = pytest.importorskip("scipy.sparse")
    r, _ = dataset
    d = r.numpy()
    coo = sp.coo_matrix((d["csr_data"], (d["coo_row"], d["csr_indices"])), shape=(r.m, r.n))
    csr, csc = coo.tocsr(), coo.tocsc()
    assert np.array_equal(csr.indptr, d["csr_indptr"]) and np.array_equal(csr.indices, d["csr_indices"])
    assert np.array_equal(csr.data, d["csr_data"])
    assert np.array_equal(csc.indptr, d["csc_indptr"]) and np.array_equal(csc.indices, d["csc_indices"])
    assert np.array_equal(csc.data, d["csc_data"])
    # coo.row is the CSR row expansion (what RMSE assumes, als.cu:196-198)
    assert np.array_equal(d["coo_row"], np.repeat(np.arange(r.m), np.diff(d["csr_indptr"])))
    assert (np.diff(d["csr_indptr"]) > 0).all() and (np.diff(d["csc_indptr"]) > 0).all()


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (reference not mounted at build time)")
def test_reference_loaders_read_our_files(dataset):
    r, d = dataset
    ref = _load_with(C.CDLL(REF_SO), d, r.m, r.n, r.nnz, r.nnz_test)
    for k, v in r.numpy().items():
        assert np.array_equal(ref[k], v), k


def test_our_cpp_loaders_match(dataset, tmp_path):
    """csrc/host_utilities.cpp (plain C++, no HIP) against the same files."""
    r, d = dataset
    src = os.path.join(ROOT, "cumf_als_amd", "csrc", "host_utilities.cpp")
    so = str(tmp_path / "libhu.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), src, "-o", so], check=True)
    ours = _load_with(C.CDLL(so), d, r.m, r.n, r.nnz, r.nnz_test)
    for k, v in r.numpy().items():
        assert np.array_equal(ours[k], v), k


def test_loader_fails_loudly_on_missing_file(tmp_path):
    """Unlike host_utilities.cpp:27-31 (prints and returns), a bad DATA_DIR stops the process."""
    src = os.path.join(ROOT, "cumf_als_amd", "csrc", "host_utilities.cpp")
    so = str(tmp_path / "libhu.so")
    subprocess.run(["g++", "-O2", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"), src, "-o", so], check=True)
    code = (f"import ctypes as C; l=C.CDLL({so!r}); f=getattr(l,{MANGLED['coorow']!r}); "
            f"f.argtypes=[C.c_char_p,C.c_void_p,C.c_long]; b=(C.c_int*4)(); f(b'/nonexistent/x.bin', b, 4)")
    p = subprocess.run(["python", "-c", code], capture_output=True, text=True)
    assert p.returncode != 0 and "Unable to open file" in p.stderr
