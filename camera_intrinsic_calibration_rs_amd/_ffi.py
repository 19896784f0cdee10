"""ctypes binding of the C ABI in include/ccal.h.

Fails loudly when the HIP library has not been built: there is no CPU fallback in the product.
"""
from __future__ import annotations

import ctypes as C
import os

PMAX = 10
MAX_CAMS = 8
KMAX = 128

OK, ERR_INVALID_ARG, ERR_HIP, ERR_NONFINITE, ERR_NOT_PD, ERR_NO_CONVERGENCE, ERR_UNSUPPORTED, ERR_NO_MEMORY = range(8)
STATUS_NAMES = ["CCAL_OK", "CCAL_ERR_INVALID_ARG", "CCAL_ERR_HIP", "CCAL_ERR_NONFINITE", "CCAL_ERR_NOT_PD",
                "CCAL_ERR_NO_CONVERGENCE", "CCAL_ERR_UNSUPPORTED", "CCAL_ERR_NO_MEMORY"]
METHOD_GN, METHOD_LM = 0, 1
TRANSPORT_NONE, TRANSPORT_RCCL, TRANSPORT_INPROC = 0, 1, 2
MULTI_MAX_DEVICES = 16

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


class ProblemDesc(C.Structure):
    _fields_ = [
        ("n_cams", C.c_int32), ("model", _ip), ("width", _dp), ("height", _dp),
        ("xy_same_focal", C.c_int32), ("n_slots", C.c_int32), ("n_obs", C.c_int32),
        ("obs_cam", _ip), ("obs_slot", _ip), ("obs_offsets", _lp),
        ("p3d_x", _fp), ("p3d_y", _fp), ("p3d_z", _fp), ("p2d_u", _fp), ("p2d_v", _fp),
        ("huber_delta", C.c_double),
    ]


class SolverOpts(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("max_iterations", C.c_int32),
        ("min_abs_error_decrease", C.c_double), ("min_rel_error_decrease", C.c_double), ("min_error", C.c_double),
        ("lm_initial_radius", C.c_double), ("lm_min_diagonal", C.c_double), ("lm_max_diagonal", C.c_double),
        ("verbose", C.c_int32), ("timeout_s", C.c_int32),
        ("error_metric", C.c_int32), ("reserved_", C.c_int32),
    ]


ERROR_SQUARED_NORM, ERROR_NORM = 0, 1          # ccal_error_metric (ccal_solver_opts.error_metric)


class Report(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("iterations", C.c_int32), ("lm_accepted", C.c_int32), ("lm_rejected", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("solve_ms", C.c_double),
        ("lm_spec_hits", C.c_int32), ("lm_spec_misses", C.c_int32),
    ]


class ModelConventions(C.Structure):
    _fields_ = [("kb4_small_radius", C.c_double), ("dist_lo", (C.c_double * 5) * 4), ("dist_hi", (C.c_double * 5) * 4),
                ("unproject_small_radius", C.c_double), ("ocv5_order", C.c_int32 * 5), ("reserved_", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

LIB_PATH = os.environ.get("CCAL_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libccal_hip.so")

# every symbol include/ccal.h declares: (name, restype, argtypes)
_vp = C.c_void_p
SYMBOLS = [
    ("ccal_ctx_create", C.c_int, [C.c_int, _vp, C.POINTER(_vp)]),
    ("ccal_ctx_destroy", None, [_vp]),
    ("ccal_last_error", C.c_char_p, [_vp]),
    ("ccal_version", C.c_char_p, []),
    ("ccal_model_num_params", C.c_int, [C.c_int]),
    ("ccal_problem_create", C.c_int, [_vp, C.POINTER(ProblemDesc), C.POINTER(_vp)]),
    ("ccal_problem_destroy", None, [_vp]),
    ("ccal_set_defaults", C.c_int, [C.POINTER(SolverOpts)]),
    ("ccal_set_bounds", C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double]),
    ("ccal_clear_bounds", C.c_int, [_vp, C.c_int, C.c_int]),
    ("ccal_fix_param", C.c_int, [_vp, C.c_int, C.c_int]),
    ("ccal_unfix_param", C.c_int, [_vp, C.c_int, C.c_int]),
    ("ccal_apply_reference_bounds", C.c_int, [_vp]),
    ("ccal_disable_distortions", C.c_int, [_vp, C.c_int, _dp]),
    ("ccal_set_allreduce", C.c_int, [_vp, ALLREDUCE_FN, _vp]),
    ("ccal_set_rccl_comm", C.c_int, [_vp, _vp]),
    ("ccal_rccl_available", C.c_int, []),
    ("ccal_rccl_version", C.c_int, []),
    ("ccal_rccl_unique_id", C.c_int, [_vp]),
    ("ccal_rccl_comm_create", C.c_int, [_vp, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    ("ccal_rccl_comm_destroy", C.c_int, [_vp]),
    ("ccal_rccl_comm_count", C.c_int, [_vp]),
    ("ccal_get_model_conventions", C.c_int, [_vp, C.POINTER(ModelConventions)]),
    ("ccal_set_model_conventions", C.c_int, [_vp, C.POINTER(ModelConventions)]),
    ("ccal_num_corners", C.c_int64, [_vp]),
    ("ccal_reduced_dim", C.c_int, [_vp]),
    ("ccal_block_dim", C.c_int, [_vp, C.c_int]),
    ("ccal_eff_num_params", C.c_int, [_vp, C.c_int]),
    ("ccal_jacobian_len", C.c_int64, [_vp]),
    ("ccal_eval", C.c_int, [_vp, _dp, _dp, _dp, C.c_int, _dp, _dp]),
    ("ccal_upload_params", C.c_int, [_vp, _dp, _dp, _dp]),
    ("ccal_download_params", C.c_int, [_vp, _dp, _dp, _dp]),
    ("ccal_eval_dev", C.c_int, [_vp, C.c_int, _vp, _vp]),
    ("ccal_sync", C.c_int, [_vp]),
    ("ccal_build_normal", C.c_int, [_vp, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp]),
    ("ccal_build_normal_dev", C.c_int, [_vp, C.c_double]),
    ("ccal_solve", C.c_int, [_vp, C.POINTER(SolverOpts), _dp, _dp, _dp, C.POINTER(Report)]),
    ("ccal_solve_dev", C.c_int, [_vp, C.POINTER(SolverOpts), C.POINTER(Report)]),
    ("ccal_solve_batch", C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(SolverOpts), C.POINTER(_dp), C.POINTER(_dp), C.POINTER(_dp),
                                   C.POINTER(Report)]),
    ("ccal_init_poses", C.c_int, [_vp, _dp, C.c_int, _dp, _ip]),
    ("ccal_pin_buffer", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("ccal_unpin_buffer", C.c_int, [C.c_void_p, C.c_void_p]),
    ("ccal_init_camera_extrinsic", C.c_int, [_dp, _dp, C.c_int, _dp, C.c_int, C.POINTER(Report)]),
    ("ccal_init_camera_extrinsic_opts", C.c_int, [_dp, _dp, C.c_int, _dp, C.c_int, C.POINTER(SolverOpts), C.POINTER(Report)]),
    ("ccal_se3_factor", C.c_int, [_dp, _dp, _dp, _dp, _dp]),
    # one process, several GPUs
    ("ccal_solve_sharded", C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(SolverOpts), _dp, C.POINTER(_dp), _dp, C.POINTER(Report)]),
    ("ccal_multi_create", C.c_int, [_ip, C.c_int, C.POINTER(_vp)]),
    ("ccal_multi_create_transport", C.c_int, [_ip, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("ccal_multi_rccl_ranks", C.c_int, [_vp]),
    ("ccal_create_last_error", C.c_char_p, []),
    ("ccal_partition_slots", C.c_int, [C.POINTER(ProblemDesc), C.c_int, _ip]),
    ("ccal_multi_destroy", None, [_vp]),
    ("ccal_multi_num_devices", C.c_int, [_vp]),
    ("ccal_multi_transport", C.c_int, [_vp]),
    ("ccal_multi_ctx", _vp, [_vp, C.c_int]),
    ("ccal_multi_last_error", C.c_char_p, [_vp]),
    ("ccal_multi_set_model_conventions", C.c_int, [_vp, C.POINTER(ModelConventions)]),
    ("ccal_multi_sync", C.c_int, [_vp]),
    ("ccal_multi_problem_create", C.c_int, [_vp, C.POINTER(ProblemDesc), C.POINTER(_vp)]),
    ("ccal_multi_problem_destroy", None, [_vp]),
    ("ccal_multi_problem_num_shards", C.c_int, [_vp]),
    ("ccal_multi_problem_shard", _vp, [_vp, C.c_int]),
    ("ccal_multi_problem_slot_range", C.c_int, [_vp, C.c_int, _ip, _ip]),
    ("ccal_multi_set_bounds", C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double]),
    ("ccal_multi_clear_bounds", C.c_int, [_vp, C.c_int, C.c_int]),
    ("ccal_multi_fix_param", C.c_int, [_vp, C.c_int, C.c_int]),
    ("ccal_multi_unfix_param", C.c_int, [_vp, C.c_int, C.c_int]),
    ("ccal_multi_apply_reference_bounds", C.c_int, [_vp]),
    ("ccal_multi_disable_distortions", C.c_int, [_vp, C.c_int, _dp]),
    ("ccal_multi_init_poses", C.c_int, [_vp, _dp, C.c_int, _dp, _ip]),
    ("ccal_multi_upload_params", C.c_int, [_vp, _dp, _dp, _dp]),
    ("ccal_multi_eval_dev", C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    ("ccal_multi_solve", C.c_int, [_vp, C.POINTER(SolverOpts), _dp, _dp, _dp, C.POINTER(Report)]),
    ("ccal_multi_validation", C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    ("ccal_multi_reprojection_errors", C.c_int, [_vp, _dp, _dp, _dp, _dp, _lp]),
    ("ccal_convert_model", C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _dp, C.c_double, C.c_double, C.c_int,
                                     C.POINTER(SolverOpts), C.POINTER(Report)]),
    ("ccal_reprojection_errors", C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    ("ccal_validation", C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp, _dp]),
]

LEGACY_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libccal_hip_legacy.so")
_lib = None
_legacy = None


class CcalLibraryMissing(RuntimeError):
    pass


def _bind(path, mode):
    lib = C.CDLL(path, mode=mode)
    # A/B TOOLS ONLY: an OLDER build named through CCAL_LIB (tools/ab_build.py against a previous round's library) may lack the newest
    # entry points; CCAL_LIB_ALLOW_MISSING=1 binds what is there.  The product path never sets it: a missing export raises.
    tolerant = os.environ.get("CCAL_LIB_ALLOW_MISSING") == "1" and "CCAL_LIB" in os.environ
    for name, res, args in SYMBOLS:
        if tolerant and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)   # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load libccal_hip.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CcalLibraryMissing(
            f"{LIB_PATH} not found: the HIP engine is not built (run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C camera_intrinsic_calibration_rs_amd/csrc`).  There is no CPU fallback.")
    _lib = _bind(LIB_PATH, C.RTLD_GLOBAL)
    return _lib


def load_for_switches():
    """TEST / A-B INFRASTRUCTURE: the second build when a developer switch (any CCAL_* variable the product ignores) is set in the
    environment, else the product library.  Child processes of the tests call this after setting their switch."""
    keep = {"CCAL_LIB", "CCAL_RCCL_LIB", "CCAL_MULTI_TRANSPORT"}
    dev = any(k.startswith("CCAL_") and k not in keep and not k.startswith("CCAL_BENCH") for k in os.environ)
    return load_legacy() if dev else load()


def load_legacy():
    """TEST INFRASTRUCTURE: the second build of the library that still carries the superseded matrix-core kernels
    (-DCCAL_LEGACY_KERNELS: k_gram1, k_gram, k_schur<false>; CCAL_GRAM=mfma / CCAL_GENERAL_GRAM=mfma select them there) - the
    independent second implementation some parity tests hold the product kernels against.  engine.Context(lib=load_legacy())."""
    global _legacy
    if _legacy is None:
        if not os.path.exists(LEGACY_LIB_PATH):
            raise CcalLibraryMissing(f"{LEGACY_LIB_PATH} not found (make -C camera_intrinsic_calibration_rs_amd/csrc)")
        _legacy = _bind(LEGACY_LIB_PATH, C.RTLD_LOCAL)
    return _legacy
