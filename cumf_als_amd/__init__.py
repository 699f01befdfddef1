"""MI355X-native ALS matrix-factorisation engine: drop-in for cuMF `cumf_als`'s `doALS` path.

The compute path is hand-written HIP for gfx950 behind a C ABI (`include/als.h`,
`include/cumf_als_capi.h`, built into `cumf_als_amd/csrc/libALS.so`).  This package
is the host-side mirror of the reference's operator surface (`tensorflow/als_tf.cc`):
it loads the library with ctypes and fails loudly when it is missing.
"""
__version__ = "0.1.0"
