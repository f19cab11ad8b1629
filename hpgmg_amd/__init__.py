"""hpgmg_amd -- MI355X-native HPGMG-FV operator layer (HIP kernels behind a C ABI).

This package is only plumbing: it loads the two in-tree shared libraries

    libhpgmg_hip.so   hand-written gfx950 kernels + C-ABI launchers  (include/hpgmg_hip.h)
    libhpgmg_fv.so    host layer mirroring the reference's level.c / mg.c / solvers.c
                      plus the operator plugin operators_hip.c      (include/hpgmg_fv.h,
                                                                      include/hpgmg_operators.h)

and exposes them through ctypes.  There is NO Python or CPU implementation of any
operator here: if the libraries are missing, importing the binding raises.  The
CPU oracle (oracle/) is test infrastructure and is never imported from this package.

Reference interface mirrored: finite-volume/source/operators.h:14-50 (operator
plugin), mg.h:22-45 (cycles), hpgmg-fv.c:50-99,103-386 (benchmark protocol).
"""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_PKG)

# vector ids (reference defines.h:12-39)
VECTOR_TEMP, VECTOR_U, VECTOR_F, VECTOR_E, VECTOR_R, VECTOR_DINV = 0, 1, 2, 3, 4, 5
VECTOR_BETA_I, VECTOR_BETA_J, VECTOR_BETA_K, VECTOR_ALPHA, VECTOR_L1INV = 6, 7, 8, 9, 10
BC_PERIODIC, BC_DIRICHLET = 0, 1
STENCIL_SHAPE_BOX, STENCIL_SHAPE_STAR, STENCIL_SHAPE_NO_CORNERS = 0, 1, 2
RESTRICT_CELL, RESTRICT_FACE_I, RESTRICT_FACE_J, RESTRICT_FACE_K = 0, 1, 2, 3
OP_7PT, OP_27PT, OP_FV4, OP_FV2 = 0, 1, 2, 3
SMOOTH_CHEBY, SMOOTH_GSRB, SMOOTH_JACOBI = 0, 1, 2
(INFO_DIM, INFO_BOX_DIM, INFO_GHOSTS, INFO_JSTRIDE, INFO_KSTRIDE, INFO_VOLUME, INFO_NUM_MY_BOXES,
 INFO_NUM_VECTORS, INFO_BOXES_IN_I, INFO_MY_RANK, INFO_NUM_RANKS, INFO_NUM_MY_BLOCKS, INFO_ACTIVE, INFO_COUNT) = range(14)


class Config(ctypes.Structure):
    """hpgmg_config of include/hpgmg_operators.h."""
    _fields_ = [("op", ctypes.c_int), ("smoother", ctypes.c_int), ("helmholtz", ctypes.c_int), ("variable_coeff", ctypes.c_int)]


class HipLevel(ctypes.Structure):
    """hpgmg_hip_level of include/hpgmg_hip.h (kernel-side geometry record)."""
    _fields_ = [("box_base", ctypes.c_void_p), ("box_low", ctypes.c_void_p), ("num_boxes", ctypes.c_int),
                ("dim", ctypes.c_int), ("ghosts", ctypes.c_int), ("jStride", ctypes.c_int), ("kStride", ctypes.c_int),
                ("volume", ctypes.c_int), ("dim_i", ctypes.c_int), ("dim_j", ctypes.c_int), ("dim_k", ctypes.c_int),
                ("periodic", ctypes.c_int), ("box_nbr", ctypes.c_void_p), ("flags", ctypes.c_int), ("box_stride", ctypes.c_longlong)]


def _declare_driver_api(lib):
    """Argument/return types of the include/hpgmg_fv.h + hpgmg_operators.h entry points."""
    c_int, c_dbl, vp = ctypes.c_int, ctypes.c_double, ctypes.c_void_p
    P = ctypes.POINTER
    sig = {
        "hpgmg_configure": (c_int, [P(Config)]),
        "hpgmg_get_config": (None, [P(Config)]),
        "hpgmg_vectors_reserved": (c_int, []),
        "hpgmg_backend_name": (ctypes.c_char_p, []),
        "hpgmg_set_verbose": (None, [c_int]),
        "hpgmg_set_box_alignment": (None, [c_int, c_int, c_int, c_int]),
        "hpgmg_choose_boxes_in_i": (c_int, [c_int, c_int, c_int]),
        "hpgmg_solver_create": (vp, [c_int, c_int, c_int, c_int, c_int]),
        "hpgmg_solver_create_explicit": (vp, [c_int, c_int, c_int, c_int, c_int]),
        "hpgmg_solver_destroy": (None, [vp]),
        "hpgmg_solver_num_levels": (c_int, [vp]),
        "hpgmg_solver_level": (vp, [vp, c_int]),
        "hpgmg_solver_restrict_rhs": (None, [vp, c_int]),
        "hpgmg_solver_fmg": (c_dbl, [vp, c_int]),
        "hpgmg_solver_bench": (c_dbl, [vp, c_int, c_int, c_int]),
        "hpgmg_solver_richardson": (None, [vp, P(c_dbl)]),
        "hpgmg_level_info": (None, [vp, P(c_int)]),
        "hpgmg_level_h": (c_dbl, [vp]),
        "hpgmg_level_timers": (None, [vp, P(c_dbl)]),
        "hpgmg_set_timer_mode": (None, [c_int]),
        "hpgmg_get_timer_mode": (c_int, []),
        "hpgmg_level_eigenvalue": (c_dbl, [vp]),
        "hpgmg_level_set_eigenvalue": (None, [vp, c_dbl]),
        "hpgmg_level_box_low": (None, [vp, c_int, P(c_int)]),
        "hpgmg_level_list_counts": (c_int, [vp, c_int, c_int, P(c_int)]),
        "hpgmg_level_read_vector": (None, [vp, c_int, c_int, vp]),
        "hpgmg_level_write_vector": (None, [vp, c_int, c_int, vp]),
        "hpgmg_level_create": (vp, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_dbl]),
        "hpgmg_level_destroy": (None, [vp]),
        "hpgmg_mg_create": (vp, [vp, c_dbl, c_dbl, c_int]),
        "hpgmg_mg_destroy": (None, [vp]),
        "hpgmg_mg_level": (vp, [vp, c_int]),
        "hpgmg_mg_num_levels": (c_int, [vp]),
        "hpgmg_set_transport": (None, [vp]),
        # the reference's compile-time choices of mg.c / solvers.c as run-time setters (call before hpgmg_solver_create / MGBuild) and its other drivers
        "hpgmg_set_bottom_solver": (None, [c_int]), "hpgmg_get_bottom_solver": (c_int, []),      # 0 BiCGStab (-DUSE_BICGSTAB), 1 CG (-DUSE_CG)
        "hpgmg_set_ucycles": (None, [c_int]),                                                   # -DUSE_UCYCLES
        "hpgmg_set_fmg_vcycles": (None, [c_int]),                                               # -DUNLIMIT_FMG_ITERATIONS: 20
        "MGSolve": (None, [vp, c_int, c_int, c_int, c_dbl, c_dbl, c_dbl]),
        "FMGSolve": (None, [vp, c_int, c_int, c_int, c_dbl, c_dbl, c_dbl]),
        "MGPCG": (None, [vp, c_int, c_int, c_int, c_dbl, c_dbl, c_dbl]),
        "hpgmg_solver_mg": (vp, [vp]),
        # operators.h, same names as the reference
        "stencil_get_radius": (c_int, []), "stencil_get_shape": (c_int, []),
        "apply_op": (None, [vp, c_int, c_int, c_dbl, c_dbl]),
        "residual": (None, [vp, c_int, c_int, c_int, c_dbl, c_dbl]),
        "smooth": (None, [vp, c_int, c_int, c_dbl, c_dbl]),
        "rebuild_operator": (None, [vp, vp, c_dbl, c_dbl]),
        "restriction": (None, [vp, c_int, vp, c_int, c_int]),
        "interpolation_vcycle": (None, [vp, c_int, c_dbl, vp, c_int]),
        "interpolation_fcycle": (None, [vp, c_int, c_dbl, vp, c_int]),
        "exchange_boundary": (None, [vp, c_int, c_int]),
        "apply_BCs": (None, [vp, c_int, c_int]),
        "apply_BCs_p1": (None, [vp, c_int, c_int]),
        "dot": (c_dbl, [vp, c_int, c_int]), "norm": (c_dbl, [vp, c_int]), "mean": (c_dbl, [vp, c_int]),
        "error": (c_dbl, [vp, c_int, c_int]),
        "add_vectors": (None, [vp, c_int, c_dbl, c_int, c_dbl, c_int]),
        "scale_vector": (None, [vp, c_int, c_dbl, c_int]),
        "zero_vector": (None, [vp, c_int]),
        "shift_vector": (None, [vp, c_int, c_int, c_dbl]),
        "mul_vectors": (None, [vp, c_int, c_dbl, c_int, c_int]),
        "invert_vector": (None, [vp, c_int, c_dbl, c_int]),
        "init_vector": (None, [vp, c_int, c_dbl]),
        "color_vector": (None, [vp, c_int, c_int, c_int, c_int, c_int]),
        "random_vector": (None, [vp, c_int]),
        "initialize_problem": (None, [vp, c_dbl, c_dbl, c_dbl]),
        "IterativeSolver": (None, [vp, c_int, c_int, c_dbl, c_dbl, c_dbl]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def _declare_kernel_api(lib):
    """Argument/return types of the include/hpgmg_hip.h launchers used from Python."""
    c_int, c_dbl, vp = ctypes.c_int, ctypes.c_double, ctypes.c_void_p
    P = ctypes.POINTER
    L = P(HipLevel)
    sig = {
        "hpgmg_hip_device_count": (c_int, []), "hpgmg_hip_set_device": (c_int, [c_int]),
        "hpgmg_hip_set_stream": (None, [vp]), "hpgmg_hip_get_stream": (vp, []), "hpgmg_hip_sync": (c_int, []),
        "hpgmg_hip_malloc": (vp, [ctypes.c_size_t]), "hpgmg_hip_free": (None, [vp]),
        "hpgmg_hip_memcpy_h2d": (c_int, [vp, vp, ctypes.c_size_t]), "hpgmg_hip_memcpy_d2h": (c_int, [vp, vp, ctypes.c_size_t]),
        "hpgmg_hip_last_error": (ctypes.c_char_p, []),
        "hpgmg_hip_event_create": (vp, []), "hpgmg_hip_event_destroy": (None, [vp]),
        "hpgmg_hip_event_record": (c_int, [vp]), "hpgmg_hip_event_elapsed_ms": (c_dbl, [vp, vp]),
        "hpgmg_hip_profile_smoother": (None, [c_int]),
        "hpgmg_hip_profile_smoother_min_cells": (None, [ctypes.c_longlong]),
        "hpgmg_hip_profile_smoother_stride": (None, [c_int]),
        "hpgmg_hip_profile_smoother_read": (c_int, [P(c_dbl), P(ctypes.c_longlong), P(ctypes.c_longlong)]),
        "hpgmg_hip_smooth_cheby": (c_int, [L, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_dbl, c_dbl, c_dbl]),
        "hpgmg_hip_smooth_gsrb": (c_int, [L, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_dbl, c_int]),
        "hpgmg_hip_smooth_jacobi": (c_int, [L, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_dbl, c_dbl]),
        "hpgmg_hip_residual": (c_int, [L, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_dbl]),
        "hpgmg_hip_fill": (c_int, [L, c_int, c_dbl]),
        "hpgmg_hip_axpby": (c_int, [L, c_int, c_dbl, c_int, c_dbl, c_int]),
        "hpgmg_hip_scale": (c_int, [L, c_int, c_dbl, c_int]),
        "hpgmg_hip_norm_max": (c_int, [L, c_int, P(c_dbl)]),
        "hpgmg_hip_dot": (c_int, [L, c_int, c_int, P(c_dbl)]),
        "hpgmg_hip_sum": (c_int, [L, c_int, P(c_dbl)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def lib_paths():
    return os.path.join(_PKG, "libhpgmg_hip.so"), os.path.join(_PKG, "libhpgmg_fv.so")


_cache = {}


def load_kernels():
    """ctypes handle of libhpgmg_hip.so (the C-ABI kernel library).  Raises if it is not built."""
    if "hip" not in _cache:
        path = lib_paths()[0]
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `make -C hpgmg_amd/csrc` (or __graft_entry__.build()); "
                              "there is no CPU fallback for the operator kernels")
        _cache["hip"] = _declare_kernel_api(ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL))
    return _cache["hip"]


def load_driver():
    """ctypes handle of libhpgmg_fv.so (host layer + HIP operator plugin).  Raises if it is not built."""
    if "fv" not in _cache:
        load_kernels()
        path = lib_paths()[1]
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `make -C hpgmg_amd/csrc` (or __graft_entry__.build())")
        lib = _declare_driver_api(ctypes.CDLL(path))
        if lib.hpgmg_backend_name() != b"hip":
            raise ImportError("libhpgmg_fv.so is not linked against the HIP operator plugin")
        _cache["fv"] = lib
    return _cache["fv"]


def bind_driver_library(path):
    """Declare the same driver API on another build of the host layer (the test-suite binds its CPU checker build with this)."""
    return _declare_driver_api(ctypes.CDLL(path))
