import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def alslib():
    """libALS.so loaded through ctypes; built if absent (hipcc cross-compiles without a GPU)."""
    from cumf_als_amd import lib

    if not os.path.exists(lib.LIB_PATH):
        lib.build()
    return lib.load()


@pytest.fixture
def gram_mode(request, alslib):
    """Select the Gram arithmetic for one test ("exact" = fp32 MFMA / fmaf-chain bits, "auto" =
    bf16x3 split where the wave kernels exist); always restored to the default afterwards."""
    from cumf_als_amd import als

    mode = getattr(request, "param", "auto")
    als.set_gram_mode(mode)
    yield mode
    als.set_gram_mode("auto")
