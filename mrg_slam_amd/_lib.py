"""ctypes loader of libmrgfe.so (the C ABI of include/mrgfe.h).

There is NO CPU fallback: if the HIP library is missing it is built with hipcc (cross-compiles without a GPU), and if it
cannot be loaded, or no GPU is present when a context is created, an exception is raised.  Nothing under ``oracle/`` is
ever imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.environ.get("MRGFE_LIB") or os.path.join(_PKG, "libmrgfe.so")  # MRGFE_LIB: kernel-variant experiments only

MRGFE_OK, ERR_INVALID, ERR_HIP, ERR_OVERFLOW, ERR_EMPTY, ERR_STATE = 0, -1, -2, -3, -4, -5
NDT_HIP, GICP_HIP, SMALL_GICP_HIP, VGICP_HIP, ICP_HIP, PCL_GICP_HIP, PCL_GICP_OMP_HIP, PCL_NDT_HIP = 0, 1, 2, 3, 4, 5, 6, 7
SEARCH = {"KDTREE": 0, "DIRECT26": 1, "DIRECT7": 2, "DIRECT1": 3}


def layout(stride: int, xyz_offset: int = 0, intensity_offset: int = 12) -> int:
    """MRGFE_LAYOUT(stride, xyz_offset, intensity_offset) of include/mrgfe.h: the point-layout descriptor passed as ``stride_bytes``."""
    return int(stride) | ((int(intensity_offset) + 1) << 16) | (int(xyz_offset) << 24)


LAYOUT_PACKED = 16
LAYOUT_PCL_XYZI = layout(32, 0, 16)  # in-memory pcl::PointXYZI


class MrgfeError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libmrgfe error {status}: {message}")
        self.status = status


class PrefilterParams(C.Structure):
    """struct mrgfe_prefilter_params (the prefiltering_component ROS parameters, config/mrg_slam.yaml:41-64)."""

    _fields_ = [("enable_distance_filter", C.c_int), ("distance_near_thresh", C.c_double), ("distance_far_thresh", C.c_double), ("downsample_method", C.c_int),
                ("downsample_resolution", C.c_double), ("downsample_min_points_per_voxel", C.c_int), ("outlier_removal_method", C.c_int), ("radius_radius", C.c_double),
                ("radius_min_neighbors", C.c_int), ("statistical_mean_k", C.c_int), ("statistical_stddev", C.c_double)]


class InfParams(C.Structure):
    """struct mrgfe_inf_params (InformationMatrixCalculator's ROS parameters, config/mrg_slam.yaml:216-223,173)."""

    _fields_ = [("use_const_inf_matrix", C.c_int), ("const_stddev_x", C.c_double), ("const_stddev_q", C.c_double), ("var_gain_a", C.c_double), ("min_stddev_x", C.c_double),
                ("max_stddev_x", C.c_double), ("min_stddev_q", C.c_double), ("max_stddev_q", C.c_double), ("fitness_score_thresh", C.c_double)]


class RegParams(C.Structure):
    """struct mrgfe_reg_params (mirrors the reg_* ROS parameters of registrations.cpp:34-43)."""

    _fields_ = [
        ("method", C.c_int),
        ("num_threads", C.c_int),
        ("transformation_epsilon", C.c_double),
        ("maximum_iterations", C.c_int),
        ("max_correspondence_distance", C.c_double),
        ("max_optimizer_iterations", C.c_int),
        ("use_reciprocal_correspondences", C.c_int),
        ("correspondence_randomness", C.c_int),
        ("resolution", C.c_double),
        ("nn_search_method", C.c_int),
        ("step_size", C.c_double),
        ("outlier_ratio", C.c_double),
        ("rotation_epsilon", C.c_double),
    ]


class PairResult(C.Structure):
    """struct mrgfe_pair_result: the 384-byte record gathered across ranks."""

    _fields_ = [
        ("T", C.c_float * 16),
        ("H", C.c_double * 36),
        ("fitness", C.c_double),
        ("trans_probability", C.c_double),
        ("converged", C.c_int32),
        ("iterations", C.c_int32),
        ("evaluations", C.c_int32),
        ("pair_id", C.c_int32),
    ]


assert C.sizeof(PairResult) == 384

# every symbol include/mrgfe.h declares: name -> (restype, argtypes)
_vp, _fp, _dp, _ip, _u32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
_szp = C.POINTER(C.c_size_t)
SIGNATURES = {
    "mrgfe_last_error": (C.c_char_p, []),
    "mrgfe_version": (C.c_char_p, []),
    "mrgfe_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "mrgfe_ctx_create_priority": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "mrgfe_ctx_create_reserving": (C.c_int, [C.c_int, C.c_int, C.POINTER(_vp)]),
    "mrgfe_ctx_destroy": (None, [_vp]),
    "mrgfe_ctx_synchronize": (C.c_int, [_vp]),
    "mrgfe_pin_host_buffer": (C.c_int, [_vp, _vp, C.c_size_t]),
    "mrgfe_unpin_host_buffer": (C.c_int, [_vp, _vp]),
    "mrgfe_ctx_set_zero_copy_uploads": (C.c_int, [_vp, C.c_int]),
    "mrgfe_ctx_stream": (_vp, [_vp]),
    "mrgfe_ctx_fitness_stats": (C.c_int, [_vp, _dp]),
    "mrgfe_ctx_knn_stats": (C.c_int, [_vp, _dp]),
    "mrgfe_ingest_pointcloud2": (C.c_int, [_vp, C.POINTER(C.c_uint8), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, _fp, _vp]),
    "mrgfe_reg_default_params": (None, [C.c_int, C.POINTER(RegParams)]),
    "mrgfe_reg_create": (C.c_int, [_vp, C.POINTER(RegParams), C.POINTER(_vp)]),
    "mrgfe_reg_destroy": (None, [_vp]),
    "mrgfe_reg_set_target": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t]),
    "mrgfe_reg_set_source": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t]),
    "mrgfe_reg_set_target_device": (C.c_int, [_vp, _vp, C.c_size_t]),
    "mrgfe_reg_set_source_device": (C.c_int, [_vp, _vp, C.c_size_t]),
    "mrgfe_reg_source_becomes_target": (C.c_int, [_vp]),
    "mrgfe_reg_set_source_from_prefilter": (C.c_int, [_vp, _vp, C.c_size_t]),
    "mrgfe_reg_align": (C.c_int, [_vp, _fp, _fp]),
    "mrgfe_reg_has_converged": (C.c_int, [_vp]),
    "mrgfe_reg_final_transformation": (C.c_int, [_vp, _fp]),
    "mrgfe_reg_fitness": (C.c_int, [_vp, C.c_double, _dp]),
    "mrgfe_reg_nn1_target": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t, _ip, _fp]),
    "mrgfe_reg_iterations": (C.c_int, [_vp]),
    "mrgfe_reg_evaluations": (C.c_int, [_vp]),
    "mrgfe_reg_trans_probability": (C.c_double, [_vp]),
    "mrgfe_reg_hessian": (C.c_int, [_vp, _dp]),
    "mrgfe_ndt_evaluate": (C.c_int, [_vp, _fp, _dp, C.c_int, _dp, _dp, _dp]),
    "mrgfe_map_store_create": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "mrgfe_map_store_destroy": (None, [_vp]),
    "mrgfe_map_store_add": (C.c_int, [_vp, C.c_uint64, _fp, C.c_size_t, C.c_size_t]),
    "mrgfe_map_store_has": (C.c_int, [_vp, C.c_uint64, C.POINTER(C.c_size_t)]),
    "mrgfe_map_store_bytes": (C.c_size_t, [_vp]),
    "mrgfe_map_store_generate": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint64), _dp, C.POINTER(C.c_uint8), C.c_float, C.c_int, C.c_float, C.c_int, _fp, C.c_size_t,
                                           C.POINTER(C.c_size_t)]),
    "mrgfe_prefilter_default_params": (None, [C.POINTER(PrefilterParams)]),
    "mrgfe_prefilter": (C.c_int, [_vp, C.POINTER(PrefilterParams), _fp, C.c_size_t, C.c_size_t, _fp, C.POINTER(C.c_size_t)]),
    "mrgfe_prefilter_device": (C.c_int, [_vp, C.POINTER(PrefilterParams), _fp, C.c_size_t, C.c_size_t, _vp, C.POINTER(C.c_size_t)]),
    "mrgfe_knn": (C.c_int, [_vp, _fp, C.c_size_t, _fp, C.c_size_t, C.c_size_t, C.c_int, _ip, _fp]),
    "mrgfe_pclgicp_evaluate": (C.c_int, [_vp, _fp, _dp, _dp, _dp, C.POINTER(C.c_int)]),
    "mrgfe_gicp_linearize": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _ip]),
    "mrgfe_gicp_covariances": (C.c_int, [_vp, C.c_int, _dp]),
    "mrgfe_ndt_num_leaves": (C.c_int, [_vp]),
    "mrgfe_ndt_grid": (C.c_int, [_vp, _ip, _ip, _ip]),
    "mrgfe_ndt_leaves": (C.c_int, [_vp, _ip, _ip, _dp, _dp]),
    "mrgfe_ndt_mean_neighbours": (C.c_double, [_vp]),
    "mrgfe_distance_filter": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t, C.c_double, C.c_double, _fp, _szp]),
    "mrgfe_voxelgrid": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t, C.c_float, C.c_int, _fp, _szp, C.POINTER(C.c_int)]),
    "mrgfe_approx_voxelgrid": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t, C.c_float, _fp, _szp]),
    "mrgfe_radius_outlier": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t, C.c_double, C.c_int, _fp, _szp]),
    "mrgfe_statistical_outlier": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t, C.c_int, C.c_double, _fp, _szp]),
    "mrgfe_calc_fitness_score": (C.c_int, [_vp, _fp, C.c_size_t, _fp, C.c_size_t, C.c_size_t, _dp, C.c_double, _dp]),
    "mrgfe_inf_default_params": (None, [C.POINTER(InfParams)]),
    "mrgfe_inf_weight": (C.c_double, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "mrgfe_inf_matrix_from_fitness": (C.c_int, [C.POINTER(InfParams), C.c_double, _dp]),
    "mrgfe_calc_information_matrix": (C.c_int, [_vp, C.POINTER(InfParams), _fp, C.c_size_t, _fp, C.c_size_t, C.c_size_t, _dp, _dp, _dp]),
    "mrgfe_map_store_fitness": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _dp, C.c_double, _dp]),
    "mrgfe_map_store_information_matrix": (C.c_int, [_vp, C.POINTER(InfParams), C.c_uint64, C.c_uint64, _dp, _dp, _dp]),
    "mrgfe_map_cloud_generate": (C.c_int, [_vp, C.c_int, C.POINTER(_fp), _szp, C.c_size_t, _dp, C.POINTER(C.c_uint8), C.c_float, C.c_int, C.c_float, C.c_int, _fp,
                                           C.c_size_t, _szp]),
    "mrgfe_remove_points_near": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t, _fp, C.c_int, C.c_float, _fp, _szp, _fp, _szp]),
    "mrgfe_deskew": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t, _fp, C.c_double, _fp]),
    "mrgfe_transform_cloud": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t, _fp, _fp]),
    "mrgfe_batch_create": (C.c_int, [_vp, C.POINTER(RegParams), C.POINTER(_vp)]),
    "mrgfe_batch_destroy": (None, [_vp]),
    "mrgfe_batch_clear": (C.c_int, [_vp]),
    "mrgfe_batch_add_target": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t]),
    "mrgfe_batch_add_target_device": (C.c_int, [_vp, _vp, C.c_size_t]),
    "mrgfe_batch_add_pair": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t, C.c_size_t, _fp]),
    "mrgfe_batch_add_pair_device": (C.c_int, [_vp, C.c_int, _vp, C.c_size_t, _fp]),
    "mrgfe_batch_add_device": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_void_p), _szp, C.c_int, _ip, C.POINTER(C.c_void_p), _szp, _fp]),
    "mrgfe_batch_add_pair_keyed": (C.c_int, [_vp, C.c_int, C.c_uint64, _fp, C.c_size_t, C.c_size_t, _fp]),
    "mrgfe_batch_has_cloud": (C.c_int, [_vp, C.c_uint64, C.POINTER(C.c_size_t)]),
    "mrgfe_batch_store_bytes": (C.c_size_t, [_vp]),
    "mrgfe_batch_forget": (C.c_int, [_vp, C.c_uint64]),
    "mrgfe_batch_set_guess": (C.c_int, [_vp, C.c_int, _fp]),
    "mrgfe_batch_build_targets": (C.c_int, [_vp]),
    "mrgfe_batch_align": (C.c_int, [_vp, C.c_double, C.POINTER(PairResult)]),
    "mrgfe_batch_align_async": (C.c_int, [_vp, C.c_double, C.POINTER(PairResult)]),
    "mrgfe_batch_wait": (C.c_int, [_vp]),
    "mrgfe_batch_num_pairs": (C.c_int, [_vp]),
    "mrgfe_batch_fitness_stats": (C.c_int, [_vp, _dp]),
    "mrgfe_batch_kernel_stats": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(C.c_int64), _dp]),
    "mrgfe_reg_kernel_stats": (C.c_int, [_vp, C.c_int, _dp, C.POINTER(C.c_int64), _dp]),
    "mrgfe_batch_pair_counts": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "mrgfe_batch_largest_launch": (C.c_int, [_vp, _dp]),
    "mrgfe_node_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(RegParams), C.POINTER(_vp)]),
    "mrgfe_node_destroy": (None, [_vp]),
    "mrgfe_node_num_members": (C.c_int, [_vp]),
    "mrgfe_node_clear": (C.c_int, [_vp]),
    "mrgfe_node_add_target": (C.c_int, [_vp, _fp, C.c_size_t, C.c_size_t]),
    "mrgfe_node_add_target_keyed": (C.c_int, [_vp, C.c_uint64, _fp, C.c_size_t, C.c_size_t]),
    "mrgfe_node_add_pair": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t, C.c_size_t, _fp]),
    "mrgfe_node_add_pair_keyed": (C.c_int, [_vp, C.c_int, C.c_uint64, _fp, C.c_size_t, C.c_size_t, _fp]),
    "mrgfe_node_num_pairs": (C.c_int, [_vp]),
    "mrgfe_node_align": (C.c_int, [_vp, C.c_double, C.POINTER(PairResult)]),
    "mrgfe_node_shard": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mrgfe_node_last_gather": (C.c_int, [_vp]),
    "mrgfe_node_forget": (C.c_int, [_vp, C.c_uint64]),
    "mrgfe_node_store_bytes": (C.c_size_t, [_vp]),
    "mrgfe_node_select_best": (C.c_int, [C.POINTER(PairResult), C.c_int, _ip, _ip, _dp]),
    "mrgfe_batch_rounds": (C.c_int, [_vp]),
    "mrgfe_batch_timing": (C.c_int, [_vp, _dp]),
    "mrgfe_batch_timing_reset": (C.c_int, [_vp]),
}

# include/mrgfe_debug.h: diagnostic entry points (exported by the shipped library) ...
DEBUG_SIGNATURES = {
    "mrgfe_dbg_set_gicp_corr_passes": (C.c_int, [C.c_int]),
    "mrgfe_dbg_grid_set_query": (C.c_int, [_vp, C.POINTER(_fp), C.POINTER(C.c_size_t), C.c_int, _fp, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_int32), _fp]),
    "mrgfe_dbg_sort_pairs": (C.c_int, [_vp, _u32p, _u32p, C.c_size_t, C.c_int, _u32p, _u32p]),
    "mrgfe_dbg_wave_sums": (C.c_int, [_vp, C.c_int, _dp, C.c_int, _dp, _dp]),
    "mrgfe_dbg_exclusive_scan": (C.c_int, [_vp, _u32p, C.c_size_t, _u32p, _u32p]),
    "mrgfe_dbg_minmax": (C.c_int, [_vp, _fp, C.c_size_t, _fp, _fp, _u32p]),
    "mrgfe_dbg_set_host_control": (C.c_int, [C.c_int]),
    "mrgfe_dbg_set_fused_launch": (C.c_int, [C.c_int]),
    "mrgfe_dbg_set_ndt_reference_order": (C.c_int, [C.c_int]),
    "mrgfe_dbg_set_fit_sweep": (C.c_int, [C.c_int]),
    "mrgfe_dbg_set_prefilter_device_driven": (C.c_int, [C.c_int]),
    "mrgfe_dbg_set_pclgicp_reference_order": (C.c_int, [C.c_int]),
    "mrgfe_dbg_set_fit_stats": (C.c_int, [C.c_int]),
    "mrgfe_dbg_sincosf": (None, [_fp, C.c_size_t, _fp, _fp]),
    "mrgfe_dbg_exp": (C.c_int, [_vp, _dp, C.c_size_t, C.c_int, _dp]),
    "mrgfe_dbg_ctl_math": (C.c_int, [_vp, _dp, C.c_int, C.c_int, _fp, _dp, _dp]),
    "mrgfe_dbg_ctl_create": (C.c_int, [C.POINTER(RegParams), _fp, C.c_uint32, C.POINTER(_vp)]),
    "mrgfe_dbg_ctl_destroy": (None, [_vp]),
    "mrgfe_dbg_ctl_request": (C.c_int, [_vp, C.POINTER(C.c_int), _fp, _dp]),
    "mrgfe_dbg_ctl_result": (C.c_int, [_vp, C.c_double, _dp, _dp, C.c_double]),
    "mrgfe_dbg_ctl_final": (C.c_int, [_vp, _fp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}
# ... and its two fault injectors, which exist only in the -DMRGFE_TESTING build (libmrgfe_testing.so: tests/faultinject/ runs under MRGFE_LIB=that file)
TESTING_SIGNATURES = {
    "mrgfe_dbg_node_fail_member": (C.c_int, [_vp, C.c_int]),
    "mrgfe_dbg_fail_alloc_after": (C.c_long, [C.c_long]),
}
TESTING_LIB_PATH = os.path.join(_PKG, "libmrgfe_testing.so")


def build(force: bool = False) -> str:
    """Compile libmrgfe.so for gfx950 with hipcc (csrc/Makefile). No-op when it is newer than its sources."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".cpp", ".h")) or f == "Makefile"]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "mrgfe.h"))
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "mrgfe_debug.h"))
    default_lib = os.path.join(_PKG, "libmrgfe.so")
    outs = [default_lib, TESTING_LIB_PATH]  # (make builds both: the shipped library and the -DMRGFE_TESTING variant)
    stale = force or any(not os.path.exists(o) or any(os.path.getmtime(s) > os.path.getmtime(o) for s in srcs) for o in outs)
    if stale:
        subprocess.run(["make", "-C", _CSRC, "-j8", "-s"] + (["-B"] if force else []), check=True)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            try:
                build()
            except Exception as e:  # noqa: BLE001
                raise RuntimeError(f"libmrgfe.so is missing and could not be built with hipcc ({e}); there is no CPU fallback") from e
        # A process that uses both this library and PyTorch must bind them to ONE HIP runtime, and that only works out
        # when torch's bundled runtime (same soname) is loaded first: loaded after libmrgfe, torch reports "No HIP GPUs are
        # available".  So import torch here if it is installed (MRGFE_NO_TORCH=1 skips this for torch-free programs).
        if "torch" not in sys.modules and os.environ.get("MRGFE_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except Exception:  # noqa: BLE001
                pass
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if "torch" in sys.modules else C.DEFAULT_MODE)
        for name, (res, args) in {**SIGNATURES, **DEBUG_SIGNATURES}.items():
            if os.environ.get("MRGFE_LIB") and os.environ.get("MRGFE_LIB_ALLOW_MISSING") and not hasattr(L, name):
                continue  # an OLDER library file in an A/B measurement (MRGFE_LIB): symbols added since are simply not there
            f = getattr(L, name)  # AttributeError here == the library does not export a declared symbol
            f.restype, f.argtypes = res, args
        for name, (res, args) in TESTING_SIGNATURES.items():  # only the testing variant has them
            if hasattr(L, name):
                f = getattr(L, name)
                f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def last_error() -> str:
    return lib().mrgfe_last_error().decode("utf-8", "replace")


def check(status: int) -> int:
    if status < 0:
        raise MrgfeError(status, last_error())
    return status


class Context:
    """mrgfe_ctx: one per (process, GPU)."""

    def __init__(self, device: int = 0, high_priority: bool = False, reserve_cus: int = 0):
        """``high_priority``: the context's stream gets the device's highest stream priority (odometry next to loop-closure batches).
        ``reserve_cus`` > 0: the kernels of this context never occupy that many of the device's compute units (``mrgfe_ctx_create_reserving``:
        the loop-closure batches' context, so that the odometry contexts' small launches always find free compute units)."""
        self._h = _vp()
        if reserve_cus:
            check(lib().mrgfe_ctx_create_reserving(device, int(reserve_cus), C.byref(self._h)))
        else:
            check(lib().mrgfe_ctx_create_priority(device, int(bool(high_priority)), C.byref(self._h)))
        self.device = device

    def synchronize(self):
        check(lib().mrgfe_ctx_synchronize(self._h))

    def set_zero_copy_uploads(self, on: bool = True):
        """``mrgfe_ctx_set_zero_copy_uploads``: clouds in page-locked host memory go up by DMA from the caller's buffer — which must then stay
        unchanged until the consuming call (align / wait / synchronize) has returned."""
        check(lib().mrgfe_ctx_set_zero_copy_uploads(self._h, int(bool(on))))

    def fitness_stats(self) -> dict:
        """What the last getFitnessScore pass on this context did (``mrgfe_ctx_fitness_stats``)."""
        v = (C.c_double * 11)()
        check(lib().mrgfe_ctx_fitness_stats(self._h, v))
        keys = ("ms_block", "ms_sweep", "ms_far", "queries", "queued", "queued_far", "words", "boxes_tested", "cells", "points", "calls")
        return dict(zip(keys, [float(x) for x in v]))

    def knn_stats(self) -> dict:
        """The last exact k-NN launch on this context (``mrgfe_ctx_knn_stats``)."""
        v = (C.c_double * 5)()
        check(lib().mrgfe_ctx_knn_stats(self._h, v))
        return dict(zip(("ms", "queries", "k", "candidates", "launches"), [float(x) for x in v]))

    def close(self):
        if getattr(self, "_h", None):
            lib().mrgfe_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_default_ctx: dict[int, Context] = {}


def default_context(device: int | None = None) -> Context:
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("MRGFE_DEVICE") is None else int(os.environ["MRGFE_DEVICE"])
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
