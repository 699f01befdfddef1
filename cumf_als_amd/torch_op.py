"""`torch.ops.cumf_als.do_als` -- the PyTorch counterpart of the reference's TensorFlow op
`DoAls` (tensorflow/als_tf.cc:7-30, 44-137).

Like the TF op it is a CPU op that drives the GPU: all inputs are host tensors, the op
initialises the factors the way the TF kernel does (thetaT[k] = 0.1 * rand()/RAND_MAX in
k-order with libc `rand()`, XT = 0: als_tf.cc:119-126), calls `doALS`, and returns
`(thetat, xt, rmse)`.  Differences, all deliberate:

* m, n, f, nnz, nnz_test, lambda, iters, xbatch, thetabatch, deviceid are scalars, not
  1-element tensors (als_tf.cc:17-26 reads element 0 of each);
* `thetat` / `xt` come back with the shape the data really has, (n, f) / (m, f)
  row-contiguous; the TF op labels the same flat buffers {f, n} / {f, m} (als_tf.cc:106-109);
* `rmse` keeps the TF shape (1, 1).

The solver follows `doALS`'s environment switches (`CUMF_ALS_SOLVER`).

    import cumf_als_amd.torch_op  # registers the op
    thetat, xt, rmse = torch.ops.cumf_als.do_als(csrRow, csrCol, csrVal, cscRow, cscCol, cscVal, cooRow,
                                                 cooRowTest, cooColTest, cooValTest, m, n, f, nnz, nnz_test,
                                                 llambda, iters, xbatch, thetabatch, 0)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import als, lib as _libmod

SCHEMA = ("do_als(Tensor csrrow, Tensor csrcol, Tensor csrval, Tensor cscrow, Tensor csccol, Tensor cscval, "
          "Tensor coorow, Tensor coorowtest, Tensor coocoltest, Tensor coovaltest, int m, int n, int f, int nnz, "
          "int nnz_test, float lambda_, int iters, int xbatch, int thetabatch, int deviceid) "
          "-> (Tensor thetat, Tensor xt, Tensor rmse)")

_library = torch.library.Library("cumf_als", "DEF")
_library.define(SCHEMA)


def tf_style_init(m: int, n: int, f: int):
    """thetaT as als_tf.cc:121-123 (continues the process's rand() stream, like the TF kernel), XT = 0."""
    lib = _libmod.load()
    thetat = np.empty((n, f), np.float32)
    lib.cumf_rand_init(thetat.ctypes.data_as(C.c_void_p), n * f, 0.1, -1)
    return thetat, np.zeros((m, f), np.float32)


def _host(t: torch.Tensor, dtype) -> np.ndarray:
    if t.is_cuda:
        raise TypeError("cumf_als.do_als takes host tensors (it is a CPU op that drives the GPU, like DoAls)")
    return np.ascontiguousarray(t.detach().numpy(), dtype)


def _do_als(csrrow, csrcol, csrval, cscrow, csccol, cscval, coorow, coorowtest, coocoltest, coovaltest,
            m, n, f, nnz, nnz_test, lambda_, iters, xbatch, thetabatch, deviceid):
    thetat0, xt0 = tf_style_init(m, n, f)
    solver = "lu" if os.environ.get("CUMF_ALS_SOLVER", "cg").lower() == "lu" else "cg"
    thetat, xt, rmse = als.do_als(
        _host(csrrow, np.int32), _host(csrcol, np.int32), _host(csrval, np.float32),
        _host(cscrow, np.int32), _host(csccol, np.int32), _host(cscval, np.float32),
        _host(coorow, np.int32), _host(coorowtest, np.int32), _host(coocoltest, np.int32),
        _host(coovaltest, np.float32), m, n, f, nnz, nnz_test, lambda_, iters, xbatch, thetabatch, deviceid,
        thetat_init=thetat0, xt_init=xt0, solver=solver,
        cg_iters=int(os.environ.get("CUMF_ALS_CG_ITERS", "6")),
        fused=os.environ.get("CUMF_ALS_PATH", "fused") != "unfused",
        exact_test_grid=os.environ.get("CUMF_ALS_EXACT_TEST_GRID", "0") not in ("", "0"),
        surpass_nan=os.environ.get("CUMF_ALS_SURPASS_NAN", "0") not in ("", "0"),
        quiet=os.environ.get("CUMF_ALS_QUIET", "0") not in ("", "0"))
    return torch.from_numpy(thetat), torch.from_numpy(xt), torch.tensor([[rmse]], dtype=torch.float32)


_library.impl("do_als", _do_als, "CPU")
