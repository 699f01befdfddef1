"""ctypes loader of `cumf_als_amd/csrc/libALS.so` (the C ABI of include/cumf_als_capi.h).

The library is the product: there is no Python or CPU fallback.  `load()` raises
when the shared object is missing or lacks a declared symbol.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# CUMF_ALS_LIB: load another build of the library (kernel experiments; the profiling build libALS_ablate.so)
LIB_PATH = os.environ.get("CUMF_ALS_LIB") or os.path.join(CSRC, "libALS.so")
MAIN_PATH = os.path.join(CSRC, "main")
HUGEWIKI_PATH = os.path.join(CSRC, "hugewiki")  # the multi-GPU program (one process per GPU over RCCL)
ABLATE_LIB_PATH = os.path.join(CSRC, "libALS_ablate.so")  # -DCUMF_ABLATE=1 build (tools/gram_pass_alone.py)
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

# every extern "C" symbol declared in include/cumf_als_capi.h
C_SYMBOLS = [
    "cumf_doALS", "cumf_doALS_ex", "cumf_plan_create", "cumf_plan_destroy", "cumf_plan_info", "cumf_plan_set_gather_rows", "cumf_gram_fast_status",
    "cumf_fused_available", "cumf_als_update_fused", "cumf_fused_sse_available", "cumf_als_update_fused_sse", "cumf_quadratic_sse_terms", "cumf_get_hermitian", "cumf_get_hermitian_packed", "cumf_get_hermitian_fp16", "cumf_cg_solve_batched_fp16", "cumf_set_tt_fp16", "cumf_get_tt_fp16", "cumf_cg_solve_batched", "cumf_lu_solve_batched",
    "cumf_pack_upper", "cumf_unpack_upper", "cumf_sse", "cumf_set_gram_mode", "cumf_get_gram_mode", "cumf_set_presplit", "cumf_get_presplit", "cumf_presplit_pitch", "cumf_presplit_table", "cumf_check_gather_table", "cumf_set_kernel_timing", "cumf_last_kernel_ms", "cumf_kernel_ms_since_reset", "cumf_last_kernel_name", "cumf_last_error", "cumf_release_scratch", "cumf_rand_init", "cumf_widen_rowptr", "cumf_als_version", "cumf_als_arch",
]
# every extern "C" symbol declared in include/cumf_dist_capi.h (the multi-GPU half-iterations, als_dist.cpp)
DIST_SYMBOLS = [
    "cumf_comm_unique_id", "cumf_comm_create", "cumf_comm_create_local", "cumf_comm_create_custom", "cumf_comm_destroy",
    "cumf_comm_rank", "cumf_comm_world", "cumf_comm_transport_name", "cumf_comm_all_reduce_f64",
    "cumf_dist_gather_create", "cumf_dist_gather_update", "cumf_dist_gather_destroy",
    "cumf_dist_reduce_create", "cumf_dist_reduce_update_theta", "cumf_dist_reduce_destroy",
]
# C++-linkage drop-in symbols (include/als.h, include/cg.h) under the reference's mangled names
CXX_SYMBOLS = [
    "_Z5doALSPKiS0_PKfS0_S0_S2_S0_PfS3_S0_S0_S2_iiillfiiii",
    "_Z17updateXWithCGHostPfS_S_iif",
    "_Z25updateXWithCGHost_tt_fp16PfS_S_iif",
    "_Z23alsUpdateFeature100HostiPKiS0_fiiPKfPfS3_i",
]

_lib = None


def build(force: bool = False) -> str:
    """Compile libALS.so and ./main for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-s", "-C", CSRC, "clean"], check=True)
    subprocess.run(["make", "-s", f"-j{os.cpu_count() or 4}", "-C", CSRC, "build"], check=True)
    return LIB_PATH


def load():
    """Load libALS.so; raise (never fall back) if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  cumf_als_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    missing = [s for s in C_SYMBOLS + DIST_SYMBOLS + CXX_SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise RuntimeError(f"{LIB_PATH} lacks symbols declared in include/: {missing}")

    vp, ip, fp = C.c_void_p, C.c_void_p, C.c_void_p
    lib.cumf_plan_create.restype = C.c_int
    lib.cumf_plan_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_long, C.c_long, C.c_long,
                                     C.c_int, C.c_int]
    lib.cumf_plan_destroy.restype = C.c_int
    lib.cumf_plan_destroy.argtypes = [C.c_void_p]
    lib.cumf_plan_info.restype = C.c_int
    lib.cumf_plan_info.argtypes = [C.c_void_p, C.POINTER(C.c_long)]
    lib.cumf_plan_set_gather_rows.restype = C.c_int
    lib.cumf_plan_set_gather_rows.argtypes = [C.c_void_p, C.c_long]
    lib.cumf_gram_fast_status.restype = C.c_int
    lib.cumf_gram_fast_status.argtypes = [C.POINTER(C.c_int)]
    lib.cumf_fused_available.restype = C.c_int
    lib.cumf_fused_available.argtypes = [C.c_int, C.c_int]
    lib.cumf_als_update_fused.restype = C.c_int
    lib.cumf_als_update_fused.argtypes = [vp, ip, fp, fp, fp, C.c_int, C.c_float, C.c_int, C.c_int, vp]
    lib.cumf_fused_sse_available.restype = C.c_int
    lib.cumf_fused_sse_available.argtypes = [vp, C.c_int]
    lib.cumf_als_update_fused_sse.restype = C.c_int
    lib.cumf_als_update_fused_sse.argtypes = [vp, ip, fp, fp, fp, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp]
    lib.cumf_quadratic_sse_terms.restype = C.c_int
    lib.cumf_quadratic_sse_terms.argtypes = [fp, fp, fp, fp, C.c_long, C.c_int, vp, vp]
    lib.cumf_get_hermitian.restype = C.c_int
    lib.cumf_get_hermitian.argtypes = [vp, ip, fp, fp, fp, fp, C.c_int, C.c_float, vp]
    lib.cumf_get_hermitian_packed.restype = C.c_int
    lib.cumf_get_hermitian_packed.argtypes = [vp, ip, fp, fp, fp, fp, C.c_int, C.c_float, vp]
    lib.cumf_get_hermitian_fp16.restype = C.c_int
    lib.cumf_get_hermitian_fp16.argtypes = [vp, ip, fp, fp, vp, fp, C.c_int, C.c_float, vp]
    lib.cumf_cg_solve_batched_fp16.restype = C.c_int
    lib.cumf_cg_solve_batched_fp16.argtypes = [vp, fp, fp, C.c_long, C.c_int, C.c_int, vp]
    lib.cumf_set_tt_fp16.restype = C.c_int
    lib.cumf_set_tt_fp16.argtypes = [C.c_int]
    lib.cumf_get_tt_fp16.restype = C.c_int
    lib.cumf_cg_solve_batched.restype = C.c_int
    lib.cumf_cg_solve_batched.argtypes = [fp, fp, fp, C.c_long, C.c_int, C.c_int, vp]
    lib.cumf_lu_solve_batched.restype = C.c_int
    lib.cumf_lu_solve_batched.argtypes = [fp, fp, fp, C.c_long, C.c_int, vp]
    lib.cumf_sse.restype = C.c_int
    lib.cumf_sse.argtypes = [fp, ip, ip, fp, fp, C.c_long, C.c_int, C.c_int, vp, vp]
    lib.cumf_pack_upper.restype = C.c_int
    lib.cumf_pack_upper.argtypes = [fp, fp, C.c_long, C.c_int, vp]
    lib.cumf_unpack_upper.restype = C.c_int
    lib.cumf_unpack_upper.argtypes = [fp, fp, C.c_long, C.c_int, vp]
    lib.cumf_set_gram_mode.restype = C.c_int
    lib.cumf_set_gram_mode.argtypes = [C.c_int]
    lib.cumf_set_presplit.restype = C.c_int
    lib.cumf_set_presplit.argtypes = [C.c_int]
    lib.cumf_get_presplit.restype = C.c_int
    lib.cumf_get_presplit.argtypes = []
    lib.cumf_presplit_pitch.restype = C.c_long
    lib.cumf_presplit_pitch.argtypes = [C.c_int]
    lib.cumf_presplit_table.restype = C.c_int
    lib.cumf_presplit_table.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    lib.cumf_get_gram_mode.restype = C.c_int
    lib.cumf_check_gather_table.restype = C.c_int
    lib.cumf_check_gather_table.argtypes = [C.c_long, C.c_int, C.c_int, C.c_int]
    if hasattr(lib, "cumf_set_debug_switches"):  # the profiling build only (libALS_ablate.so through CUMF_ALS_LIB)
        lib.cumf_set_debug_switches.restype = C.c_int
        lib.cumf_set_debug_switches.argtypes = [C.c_int]
    lib.cumf_last_kernel_name.restype = C.c_int
    lib.cumf_last_kernel_name.argtypes = [C.c_char_p, C.c_int]
    lib.cumf_last_error.restype = C.c_int
    lib.cumf_release_scratch.restype = C.c_int
    lib.cumf_set_kernel_timing.restype = C.c_int
    lib.cumf_set_kernel_timing.argtypes = [C.c_int]
    lib.cumf_last_kernel_ms.restype = C.c_int
    lib.cumf_last_kernel_ms.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.cumf_kernel_ms_since_reset.restype = C.c_int
    lib.cumf_kernel_ms_since_reset.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.cumf_widen_rowptr.restype = C.c_int
    lib.cumf_widen_rowptr.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p]
    lib.cumf_rand_init.restype = None
    lib.cumf_rand_init.argtypes = [fp, C.c_long, C.c_float, C.c_long]
    lib.cumf_als_version.restype = C.c_int
    lib.cumf_als_arch.restype = C.c_char_p
    # include/cumf_dist_capi.h
    lib.cumf_comm_unique_id.restype = C.c_int
    lib.cumf_comm_unique_id.argtypes = [vp]
    lib.cumf_comm_create.restype = C.c_int
    lib.cumf_comm_create.argtypes = [C.POINTER(C.c_void_p), vp, C.c_int, C.c_int]
    lib.cumf_comm_create_local.restype = C.c_int
    lib.cumf_comm_create_local.argtypes = [C.POINTER(C.c_void_p)]
    lib.cumf_comm_create_custom.restype = C.c_int
    lib.cumf_comm_create_custom.argtypes = [C.POINTER(C.c_void_p), vp, C.c_int, C.c_int]
    lib.cumf_comm_destroy.restype = C.c_int
    lib.cumf_comm_destroy.argtypes = [vp]
    lib.cumf_comm_rank.restype = C.c_int
    lib.cumf_comm_rank.argtypes = [vp]
    lib.cumf_comm_world.restype = C.c_int
    lib.cumf_comm_world.argtypes = [vp]
    lib.cumf_comm_transport_name.restype = C.c_char_p
    lib.cumf_comm_transport_name.argtypes = [vp]
    lib.cumf_comm_all_reduce_f64.restype = C.c_int
    lib.cumf_comm_all_reduce_f64.argtypes = [vp, vp, C.c_long, vp]
    lib.cumf_dist_gather_create.restype = C.c_int
    lib.cumf_dist_gather_create.argtypes = [C.POINTER(C.c_void_p), vp, vp, C.c_int, C.c_int]
    lib.cumf_dist_gather_update.restype = C.c_int
    lib.cumf_dist_gather_update.argtypes = [vp, vp, ip, fp, fp, fp, C.c_float, C.c_int, C.c_int, vp, vp]
    lib.cumf_dist_gather_destroy.restype = C.c_int
    lib.cumf_dist_gather_destroy.argtypes = [vp]
    lib.cumf_dist_reduce_create.restype = C.c_int
    lib.cumf_dist_reduce_create.argtypes = [C.POINTER(C.c_void_p), vp, C.c_long, C.c_int, C.c_int]
    lib.cumf_dist_reduce_update_theta.restype = C.c_int
    lib.cumf_dist_reduce_update_theta.argtypes = [vp, vp, ip, fp, fp, fp, C.c_float, C.c_int, C.c_int, fp, vp, vp]
    lib.cumf_dist_reduce_destroy.restype = C.c_int
    lib.cumf_dist_reduce_destroy.argtypes = [vp]
    host_args = [vp] * 12 + [C.c_int, C.c_int, C.c_int, C.c_long, C.c_long, C.c_float, C.c_int, C.c_int, C.c_int,
                             C.c_int]
    lib.cumf_doALS.restype = C.c_float
    lib.cumf_doALS.argtypes = host_args
    lib.cumf_doALS_ex.restype = C.c_float
    lib.cumf_doALS_ex.argtypes = host_args + [C.c_int] * 6 + [vp]
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with HIP error {rc}")
