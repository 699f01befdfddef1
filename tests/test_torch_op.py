"""torch.ops.cumf_als.do_als: the PyTorch counterpart of the reference's TF op (als_tf.cc)."""
import numpy as np
import pytest
import torch


def test_op_is_registered_with_the_tf_argument_list(alslib):
    import cumf_als_amd.torch_op as top

    op = torch.ops.cumf_als.do_als
    names = [a.name for a in op.default._schema.arguments]
    # als_tf.cc:7-27, minus the "_t" suffix of the scalar inputs
    assert names == ["csrrow", "csrcol", "csrval", "cscrow", "csccol", "cscval", "coorow", "coorowtest",
                     "coocoltest", "coovaltest", "m", "n", "f", "nnz", "nnz_test", "lambda_", "iters", "xbatch",
                     "thetabatch", "deviceid"]
    assert [r.name for r in op.default._schema.returns] == ["thetat", "xt", "rmse"]
    assert "do_als" in top.SCHEMA


def test_tf_style_init_is_libc_rand(alslib):
    """0.1 * rand()/RAND_MAX in k-order (als_tf.cc:121-123); glibc's first values after srand(1) --
    the state of a process that never called srand -- are 1804289383, 846930886, ..."""
    import ctypes as C

    from cumf_als_amd import lib as libmod

    lib = libmod.load()
    a = np.empty(4, np.float32)
    lib.cumf_rand_init(a.ctypes.data_as(C.c_void_p), 4, 0.1, 1)
    want = np.float32(0.1) * (np.float32([1804289383, 846930886, 1681692777, 1714636915]) / np.float32(2147483647))
    np.testing.assert_array_equal(a, want)
    # seed 0 + scale 0.2 is the CLI's init (main.cpp:72-76): SURVEY's known-answer values
    lib.cumf_rand_init(a.ctypes.data_as(C.c_void_p), 4, 0.2, 0)
    np.testing.assert_allclose(a, [0.168037549, 0.0788765848, 0.156619847, 0.159688011], rtol=1e-7)


@pytest.mark.gpu
def test_op_matches_do_als(alslib):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    import ctypes as C

    import cumf_als_amd.torch_op  # noqa: F401
    from cumf_als_amd import als, datagen, lib as libmod

    m, n, nnz, nnz_test, f = 300, 200, 20000, 1500, 20
    r = datagen.synth_ratings(m, n, nnz, nnz_test, seed=2, device="cpu")
    d = r.numpy()
    lib = libmod.load()
    lib.cumf_rand_init(np.empty(1, np.float32).ctypes.data_as(C.c_void_p), 0, 0.1, 7)  # pin the rand() stream
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}
    th, x, rmse = torch.ops.cumf_als.do_als(
        t["csr_indptr"].int(), t["csr_indices"], t["csr_data"], t["csc_indices"], t["csc_indptr"].int(),
        t["csc_data"], t["coo_row"], t["test_row"], t["test_col"], t["test_data"], m, n, f, r.nnz, r.nnz_test,
        0.05, 3, 1, 1, 0)
    assert th.shape == (n, f) and x.shape == (m, f) and rmse.shape == (1, 1)
    th0 = np.empty((n, f), np.float32)
    lib.cumf_rand_init(th0.ctypes.data_as(C.c_void_p), n * f, 0.1, 7)
    th2, x2, rmse2 = als.do_als(d["csr_indptr"], d["csr_indices"], d["csr_data"], d["csc_indices"],
                                d["csc_indptr"], d["csc_data"], d["coo_row"], d["test_row"], d["test_col"],
                                d["test_data"], m, n, f, r.nnz, r.nnz_test, 0.05, 3, 1, 1, 0, thetat_init=th0)
    np.testing.assert_array_equal(th.numpy(), th2)
    np.testing.assert_array_equal(x.numpy(), x2)
    assert float(rmse) == rmse2 and 0.0 < rmse2 < 5.0
