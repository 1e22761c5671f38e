"""ctypes bindings of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (mrg_slam_amd) never does.  PARITY UNPINNED: the oracle restates un-vendored upstream code (see
oracle/quirks.h).  All 4x4 matrices cross as column-major float32/float64 (numpy: pass ``M.T.copy()`` /
``order='F'``); helpers here take and return ordinary row-major numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

SEARCH = {"KDTREE": 0, "DIRECT26": 1, "DIRECT7": 2, "DIRECT1": 3}
ORDER_STABLE, ORDER_STD_SORT = 0, 1


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h")) or f == "Makefile"]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, ip, dp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_ubyte)
        vp = C.c_void_p
        sig = {
            "orc_distance_filter": (C.c_int, [fp, C.c_int, C.c_double, C.c_double, fp]),
            "orc_voxelgrid": (C.c_int, [fp, C.c_int, C.c_float, C.c_int, C.c_int, fp, ip]),
            "orc_approx_voxelgrid": (C.c_int, [fp, C.c_int, C.c_float, fp]),
            "orc_radius_outlier": (C.c_int, [fp, C.c_int, C.c_double, C.c_int, fp, u8p]),
            "orc_statistical_outlier": (C.c_int, [fp, C.c_int, C.c_int, C.c_double, fp, u8p]),
            "orc_map_cloud_generate": (C.c_int, [C.c_int, C.POINTER(fp), ip, dp, u8p, C.c_float, C.c_int, C.c_float, C.c_int, fp, ip]),
            "orc_remove_points_near": (C.c_int, [fp, C.c_int, fp, C.c_int, C.c_float, fp, fp, ip]),
            "orc_deskew": (None, [fp, C.c_int, fp, C.c_double, fp]),
            "orc_knn": (None, [fp, C.c_int, fp, C.c_int, C.c_int, ip, fp]),
            "orc_nn1_brute": (None, [fp, C.c_int, fp, C.c_int, ip, fp]),
            "orc_calc_fitness_score": (C.c_double, [fp, C.c_int, fp, C.c_int, dp, C.c_double]),
            "orc_svd6_solve": (None, [dp, dp, dp, dp]),
            "orc_sym_eig3": (None, [dp, dp, dp]),
            "orc_euler_xyz": (None, [fp, fp]),
            "orc_pose_to_matrix": (None, [dp, fp]),
            "orc_transform_points": (None, [fp, fp, C.c_int, fp]),
            "orc_ndt_create": (vp, []),
            "orc_ndt_destroy": (None, [vp]),
            "orc_ndt_set_params": (None, [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]),
            "orc_inf_weight": (C.c_double, [C.c_double] * 5),
            "orc_inf_matrix": (None, [dp, C.c_double, dp]),
            "orc_ndt_set_fused": (None, [vp, C.c_int]),
            "orc_ndt_set_thread_sums": (None, [vp, C.c_int]),
            "orc_ndt_set_gpu_order": (None, [vp, C.c_int]),
            "orc_ndt_set_target": (C.c_int, [vp, fp, C.c_int]),
            "orc_ndt_set_source": (None, [vp, fp, C.c_int]),
            "orc_ndt_align": (None, [vp, fp, fp]),
            "orc_ndt_converged": (C.c_int, [vp]),
            "orc_ndt_iterations": (C.c_int, [vp]),
            "orc_ndt_evals": (C.c_int, [vp]),
            "orc_ndt_last_pose": (None, [vp, C.POINTER(C.c_double)]),
            "orc_ndt_mean_neighbours": (C.c_double, [vp]),
            "orc_ndt_trans_probability": (C.c_double, [vp]),
            "orc_ndt_final": (None, [vp, fp]),
            "orc_ndt_hessian": (None, [vp, dp]),
            "orc_ndt_fitness": (C.c_double, [vp, C.c_double]),
            "orc_ndt_evaluate": (C.c_double, [vp, fp, dp, C.c_int, dp, dp]),
            "orc_ndt_num_leaves": (C.c_int, [vp]),
            "orc_ndt_grid": (None, [vp, ip, ip, ip]),
            "orc_ndt_leaves": (None, [vp, ip, ip, dp, dp, dp]),
            "orc_pclndt_create": (vp, []),
            "orc_pclndt_destroy": (None, [vp]),
            "orc_pclndt_set_params": (None, [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]),
            "orc_pclndt_set_target": (C.c_int, [vp, fp, C.c_int]),
            "orc_pclndt_set_source": (None, [vp, fp, C.c_int]),
            "orc_pclndt_align": (None, [vp, fp, fp]),
            "orc_pclndt_converged": (C.c_int, [vp]),
            "orc_pclndt_iterations": (C.c_int, [vp]),
            "orc_pclndt_evals": (C.c_int, [vp]),
            "orc_pclndt_mean_neighbours": (C.c_double, [vp]),
            "orc_pclndt_trans_likelihood": (C.c_double, [vp]),
            "orc_pclndt_final": (None, [vp, fp]),
            "orc_pclndt_hessian": (None, [vp, dp]),
            "orc_pclndt_fitness": (C.c_double, [vp, C.c_double]),
            "orc_pclndt_evaluate": (C.c_double, [vp, fp, dp, C.c_int, dp, dp]),
            "orc_pclndt_num_leaves": (C.c_int, [vp]),
            "orc_pclndt_leaves": (None, [vp, ip, ip, ip, dp, dp, fp]),
            "orc_pclgicp_create": (vp, []),
            "orc_pclgicp_destroy": (None, [vp]),
            "orc_pclgicp_set_params": (None, [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]),
            "orc_pclgicp_set_gpu_order": (None, [vp, C.c_int]),
            "orc_pclgicp_set_sum_threads": (None, [vp, C.c_int]),
            "orc_pclgicp_set_target": (None, [vp, fp, C.c_int]),
            "orc_pclgicp_set_source": (None, [vp, fp, C.c_int]),
            "orc_pclgicp_align": (None, [vp, fp, fp]),
            "orc_pclgicp_converged": (C.c_int, [vp]),
            "orc_pclgicp_iterations": (C.c_int, [vp]),
            "orc_pclgicp_evaluations": (C.c_int, [vp]),
            "orc_pclgicp_inner_steps": (C.c_int, [vp]),
            "orc_pclgicp_final": (None, [vp, fp]),
            "orc_pclgicp_fitness": (C.c_double, [vp, C.c_double]),
            "orc_pclgicp_covariances": (None, [vp, C.c_int, dp]),
            "orc_pclgicp_evaluate": (C.c_double, [vp, fp, dp, dp, C.POINTER(C.c_int)]),
            "orc_gicp_create": (vp, []),
            "orc_gicp_destroy": (None, [vp]),
            "orc_gicp_set_params": (None, [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]),
            "orc_gicp_set_variant": (None, [vp, C.c_int]),
            "orc_gicp_set_double_search": (None, [vp, C.c_int]),
            "orc_gicp_set_reciprocal": (None, [vp, C.c_int]),
            "orc_gicp_set_resolution": (None, [vp, C.c_double]),
            "orc_gicp_num_voxels": (C.c_int, [vp]),
            "orc_gicp_set_target": (None, [vp, fp, C.c_int]),
            "orc_gicp_set_source": (None, [vp, fp, C.c_int]),
            "orc_gicp_align": (None, [vp, fp, fp]),
            "orc_gicp_converged": (C.c_int, [vp]),
            "orc_gicp_iterations": (C.c_int, [vp]),
            "orc_gicp_final": (None, [vp, fp]),
            "orc_gicp_hessian": (None, [vp, dp]),
            "orc_gicp_fitness": (C.c_double, [vp, C.c_double]),
            "orc_gicp_covariances": (None, [vp, C.c_int, dp]),
            "orc_gicp_linearize": (C.c_double, [vp, dp, dp, dp, ip]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _pf(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _pi(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _cloud(a):
    a = _f32(a)
    assert a.ndim == 2 and a.shape[1] == 4, "clouds are N x 4 float32 (x, y, z, intensity)"
    return a


def _colmajor(M, dtype=np.float32):
    return np.ascontiguousarray(np.asarray(M, dtype=dtype).T)  # row-major of M^T == column-major of M


# ---- filters ---------------------------------------------------------------------------------------------
def distance_filter(cloud, near=0.1, far=35.0):
    c = _cloud(cloud)
    out = np.empty_like(c)
    m = lib().orc_distance_filter(_pf(c), len(c), near, far, _pf(out))
    return out[:m].copy()


def approx_voxelgrid(cloud, leaf=0.1):
    """pcl::ApproximateVoxelGrid (downsample_method APPROX_VOXELGRID): N x 4 float32 in arrival order -> flushed centroids in flush order."""
    c = _cloud(cloud)
    out = np.empty((max(len(c), 1), 4), dtype=np.float32)
    m = lib().orc_approx_voxelgrid(_pf(c), len(c), leaf, _pf(out))
    return out[:m].copy()


def voxelgrid(cloud, leaf=0.1, min_points=1, order=ORDER_STABLE):
    c = _cloud(cloud)
    out = np.empty_like(c)
    m = C.c_int(0)
    status = lib().orc_voxelgrid(_pf(c), len(c), leaf, min_points, order, _pf(out), C.byref(m))
    return out[: m.value].copy(), status


def radius_outlier(cloud, radius=0.5, min_neighbors=2):
    c = _cloud(cloud)
    out = np.empty_like(c)
    keep = np.zeros(len(c), dtype=np.uint8)
    m = lib().orc_radius_outlier(_pf(c), len(c), radius, min_neighbors, _pf(out), keep.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out[:m].copy(), keep.astype(bool)


def statistical_outlier(cloud, mean_k=30, stddev_mul=1.2):
    c = _cloud(cloud)
    out = np.empty_like(c)
    keep = np.zeros(len(c), dtype=np.uint8)
    m = lib().orc_statistical_outlier(_pf(c), len(c), mean_k, stddev_mul, _pf(out), keep.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out[:m].copy(), keep.astype(bool)


def map_cloud_generate(clouds, poses, first_keyframe=None, resolution=0.1, min_points_per_voxel=1, distance_far_thresh=10000.0, skip_first_cloud=False):
    """MapCloudGenerator::generate; returns (cloud, status): status 0 ok, -1 no keyframes, -2 empty after processing."""
    cs = [_cloud(c) for c in clouds]
    K = len(cs)
    ptrs = (C.POINTER(C.c_float) * max(K, 1))(*[_pf(c) for c in cs])
    ns = np.array([len(c) for c in cs], dtype=np.int32)
    P = np.ascontiguousarray(np.stack([np.asarray(T, dtype=np.float64).T.reshape(16) for T in poses]) if K else np.zeros((0, 16)))
    fk = None if first_keyframe is None else np.ascontiguousarray(np.asarray(first_keyframe, dtype=np.uint8))
    out = np.empty((max(int(ns.sum()), 1), 4), dtype=np.float32)
    m = C.c_int(0)
    status = lib().orc_map_cloud_generate(K, ptrs, _pi(ns), _pd(P), None if fk is None else fk.ctypes.data_as(C.POINTER(C.c_ubyte)), resolution, min_points_per_voxel,
                                          distance_far_thresh, int(skip_first_cloud), _pf(out), C.byref(m))
    return out[: m.value].copy(), status


def remove_points_near(cloud, centres, radius):
    """other-robot point removal (mrg_slam_component.cpp:396-429); radius_sqr = float(radius * radius) like the reference.
    Returns (kept, removed)."""
    c = _cloud(cloud)
    ctr = np.ascontiguousarray(np.asarray(centres, dtype=np.float32).reshape(-1, 3))
    out, rem = np.empty_like(c), np.empty_like(c)
    nr = C.c_int(0)
    kept = lib().orc_remove_points_near(_pf(c), len(c), _pf(ctr), len(ctr), float(np.float32(float(radius) * float(radius))), _pf(out), _pf(rem), C.byref(nr))
    return out[:kept].copy(), rem[: nr.value].copy()


def deskew(cloud, ang_v, scan_period=0.1):
    c = _cloud(cloud)
    out = np.empty_like(c)
    av = np.ascontiguousarray(np.asarray(ang_v, dtype=np.float32))
    lib().orc_deskew(_pf(c), len(c), _pf(av), float(scan_period), _pf(out))
    return out


def knn(target, query, k):
    t, q = _cloud(target), _cloud(query)
    idx = np.empty((len(q), k), dtype=np.int32)
    sqd = np.empty((len(q), k), dtype=np.float32)
    lib().orc_knn(_pf(t), len(t), _pf(q), len(q), k, _pi(idx), _pf(sqd))
    return idx, sqd


def nn1_brute(target, query):
    t, q = _cloud(target), _cloud(query)
    idx = np.empty(len(q), dtype=np.int32)
    sqd = np.empty(len(q), dtype=np.float32)
    lib().orc_nn1_brute(_pf(t), len(t), _pf(q), len(q), _pi(idx), _pf(sqd))
    return idx, sqd


def calc_fitness_score(cloud1, cloud2, relpose, max_range=float("inf")):
    c1, c2 = _cloud(cloud1), _cloud(cloud2)
    T = _colmajor(relpose, np.float64)
    return lib().orc_calc_fitness_score(_pf(c1), len(c1), _pf(c2), len(c2), _pd(T), max_range)


INF_DEFAULTS = {"use_const_inf_matrix": False, "const_stddev_x": 0.5, "const_stddev_q": 0.1, "var_gain_a": 2.0, "min_stddev_x": 0.1, "max_stddev_x": 0.75,
                "min_stddev_q": 0.05, "max_stddev_q": 0.2, "fitness_score_thresh": 1.25}  # config/mrg_slam.yaml:216-223,173


def inf_weight(a, max_x, min_y, max_y, x):
    return lib().orc_inf_weight(a, max_x, min_y, max_y, x)


def calc_information_matrix(cloud1, cloud2, relpose, params=None):
    """InformationMatrixCalculator::calc_information_matrix (information_matrix_calculator.cpp:14-44): (6x6 matrix, fitness score)."""
    p = dict(INF_DEFAULTS)
    p.update(params or {})
    fit = 0.0 if p["use_const_inf_matrix"] else calc_fitness_score(cloud1, cloud2, relpose, np.finfo(np.float64).max)
    v = np.array([float(bool(p["use_const_inf_matrix"])), p["const_stddev_x"], p["const_stddev_q"], p["var_gain_a"], p["min_stddev_x"], p["max_stddev_x"], p["min_stddev_q"],
                  p["max_stddev_q"], p["fitness_score_thresh"]], dtype=np.float64)
    inf = np.empty((6, 6))
    lib().orc_inf_matrix(_pd(v), fit, _pd(inf))
    return inf, fit


# ---- linear algebra --------------------------------------------------------------------------------------
def svd6_solve(A, b):
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x, s = np.empty(6), np.empty(6)
    lib().orc_svd6_solve(_pd(A), _pd(b), _pd(x), _pd(s))
    return x, s


def sym_eig3(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    w, V = np.empty(3), np.empty((3, 3))
    lib().orc_sym_eig3(_pd(A), _pd(w), _pd(V))
    return w, V


def euler_xyz(T):
    Tc = _colmajor(T)
    out = np.empty(3, dtype=np.float32)
    lib().orc_euler_xyz(_pf(Tc), _pf(out))
    return out


def pose_to_matrix(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    Tc = np.empty((4, 4), dtype=np.float32)
    lib().orc_pose_to_matrix(_pd(p), _pf(Tc))
    return Tc.T.copy()


def transform_points(T, cloud):
    c = _cloud(cloud)
    out = np.empty_like(c)
    lib().orc_transform_points(_pf(_colmajor(T)), _pf(c), len(c), _pf(out))
    return out


# ---- NDT ---------------------------------------------------------------------------------------------------
class Ndt:
    """pclomp::NormalDistributionsTransform restated (oracle/ndt.cpp) behind the pcl::Registration call surface."""

    def __init__(self, resolution=1.0, step_size=0.1, outlier_ratio=0.55, transformation_epsilon=0.1, maximum_iterations=64, num_threads=1,
                 search="DIRECT7", fused=True, gpu_order_ppt=0, thread_sums=False):
        """gpu_order_ppt > 0 (diagnostic): the derivative sums are added in the HIP kernels' order (items of that many 256-point
        tiles, lane round-robin, shuffle tree, slice reduce) instead of the reference's point order, so that a GPU result can be
        reproduced bit for bit and summation-order noise told from an arithmetic difference (oracle/ndt.cpp)."""
        self._h = lib().orc_ndt_create()
        lib().orc_ndt_set_params(self._h, resolution, step_size, outlier_ratio, transformation_epsilon, maximum_iterations, num_threads, SEARCH[search])
        lib().orc_ndt_set_fused(self._h, int(fused))
        lib().orc_ndt_set_gpu_order(self._h, int(gpu_order_ppt))
        # thread_sums: ndt_omp's own accumulation (one accumulator per OpenMP thread, added in thread order: schedule-dependent in the last
        # bits) instead of the checker's per-point records added in point order — the mode bench.py times as cpu_baseline
        lib().orc_ndt_set_thread_sums(self._h, int(thread_sums))
        self._n_src = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_ndt_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def setInputTarget(self, cloud):
        c = _cloud(cloud)
        return lib().orc_ndt_set_target(self._h, _pf(c), len(c))

    def setInputSource(self, cloud):
        c = _cloud(cloud)
        self._n_src = len(c)
        lib().orc_ndt_set_source(self._h, _pf(c), len(c))

    def align(self, guess=None, want_aligned=False):
        g = _colmajor(np.eye(4) if guess is None else guess)
        out = np.empty((self._n_src, 4), dtype=np.float32) if want_aligned else None
        lib().orc_ndt_align(self._h, _pf(g), _pf(out) if want_aligned else None)
        return out

    def hasConverged(self):
        return bool(lib().orc_ndt_converged(self._h))

    def getFinalTransformation(self):
        Tc = np.empty((4, 4), dtype=np.float32)
        lib().orc_ndt_final(self._h, _pf(Tc))
        return Tc.T.copy()

    def getFitnessScore(self, max_range=float("inf")):
        return lib().orc_ndt_fitness(self._h, max_range)

    def getFinalNumIteration(self):
        return lib().orc_ndt_iterations(self._h)

    def getTransformationProbability(self):
        return lib().orc_ndt_trans_probability(self._h)

    def getHessian(self):
        H = np.empty((6, 6))
        lib().orc_ndt_hessian(self._h, _pd(H))
        return H

    @property
    def evals(self):
        return lib().orc_ndt_evals(self._h)

    def last_pose(self):
        """pose vector (double) of the last derivative evaluation: where the optimiser stood, before the cast to the float transformation"""
        p = np.empty(6)
        lib().orc_ndt_last_pose(self._h, _pd(p))
        return p

    @property
    def mean_neighbours(self):
        return lib().orc_ndt_mean_neighbours(self._h)

    def evaluate(self, T, p, mode=0):
        """One derivative evaluation: returns (score, grad[6], hess[6,6])."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        g, H = np.zeros(6), np.zeros((6, 6))
        s = lib().orc_ndt_evaluate(self._h, _pf(_colmajor(T)), _pd(p), mode, _pd(g), _pd(H))
        return s, g, H

    def leaves(self):
        n = lib().orc_ndt_num_leaves(self._h)
        keys, npts = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        mean, cov, icov = np.empty((n, 3)), np.empty((n, 3, 3)), np.empty((n, 3, 3))
        if n:
            lib().orc_ndt_leaves(self._h, _pi(keys), _pi(npts), _pd(mean), _pd(cov), _pd(icov))
        return keys, npts, mean, cov, icov

    def grid(self):
        a, b, c = (np.empty(3, dtype=np.int32) for _ in range(3))
        lib().orc_ndt_grid(self._h, _pi(a), _pi(b), _pi(c))
        return a, b, c


# ---- GICP --------------------------------------------------------------------------------------------------

class PclNdt:
    """pcl::NormalDistributionsTransform (PCL 1.12) restated (oracle/pcl_ndt.cpp) behind the pcl::Registration call surface: what the
    reference's factory returns for registration_method "NDT" and every unknown name (registrations.cpp:115-129)."""

    def __init__(self, resolution=1.0, step_size=0.1, outlier_ratio=0.55, transformation_epsilon=0.1, maximum_iterations=35, gpu_order=0, num_threads=1):
        """gpu_order > 0 (diagnostic): the per-point factorised sums of the HIP kernel, added in its tree (items of that many 256-point tiles)."""
        self._h = lib().orc_pclndt_create()
        lib().orc_pclndt_set_params(self._h, resolution, step_size, outlier_ratio, transformation_epsilon, maximum_iterations, int(gpu_order), int(num_threads))
        self._n_src = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_pclndt_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def setInputTarget(self, cloud):
        c = _cloud(cloud)
        return lib().orc_pclndt_set_target(self._h, _pf(c), len(c))

    def setInputSource(self, cloud):
        c = _cloud(cloud)
        self._n_src = len(c)
        lib().orc_pclndt_set_source(self._h, _pf(c), len(c))

    def align(self, guess=None, want_aligned=False):
        g = _colmajor(np.eye(4) if guess is None else guess)
        out = np.empty((self._n_src, 4), dtype=np.float32) if want_aligned else None
        lib().orc_pclndt_align(self._h, _pf(g), _pf(out) if want_aligned else None)
        return out

    def hasConverged(self):
        return bool(lib().orc_pclndt_converged(self._h))

    def getFinalTransformation(self):
        Tc = np.empty((4, 4), dtype=np.float32)
        lib().orc_pclndt_final(self._h, _pf(Tc))
        return Tc.T.copy()

    def getFitnessScore(self, max_range=float("inf")):
        return lib().orc_pclndt_fitness(self._h, max_range)

    def getFinalNumIteration(self):
        return lib().orc_pclndt_iterations(self._h)

    def getTransformationLikelihood(self):
        return lib().orc_pclndt_trans_likelihood(self._h)

    getTransformationProbability = getTransformationLikelihood  # the name PCL < 1.12 (and pclomp) uses

    def getHessian(self):
        H = np.empty((6, 6))
        lib().orc_pclndt_hessian(self._h, _pd(H))
        return H

    @property
    def evals(self):
        return lib().orc_pclndt_evals(self._h)

    @property
    def mean_neighbours(self):
        return lib().orc_pclndt_mean_neighbours(self._h)

    def evaluate(self, T, p, mode=0):
        p = np.ascontiguousarray(p, dtype=np.float64)
        g, H = np.zeros(6), np.zeros((6, 6))
        s = lib().orc_pclndt_evaluate(self._h, _pf(_colmajor(T)), _pd(p), mode, _pd(g), _pd(H))
        return s, g, H

    def leaves(self):
        n = lib().orc_pclndt_num_leaves(self._h)
        keys, npts, ins = (np.empty(n, dtype=np.int32) for _ in range(3))
        mean, icov, cent = np.empty((n, 3)), np.empty((n, 3, 3)), np.empty((n, 4), dtype=np.float32)
        if n:
            lib().orc_pclndt_leaves(self._h, _pi(keys), _pi(npts), _pi(ins), _pd(mean), _pd(icov), _pf(cent))
        return keys, npts, ins, mean, icov, cent

class FastGicp:
    """fast_gicp::FastGICP restated (oracle/gicp.cpp)."""

    VARIANT = 0

    def __init__(self, correspondence_randomness=20, max_correspondence_distance=2.0, transformation_epsilon=0.1, rotation_epsilon=2e-3,
                 maximum_iterations=64, num_threads=1):
        self._h = lib().orc_gicp_create()
        lib().orc_gicp_set_params(self._h, correspondence_randomness, max_correspondence_distance, transformation_epsilon, rotation_epsilon,
                                  maximum_iterations, num_threads)
        lib().orc_gicp_set_variant(self._h, self.VARIANT)
        self._n_src = self._n_tgt = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_gicp_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def setInputTarget(self, cloud):
        c = _cloud(cloud)
        self._n_tgt = len(c)
        lib().orc_gicp_set_target(self._h, _pf(c), len(c))

    def setInputSource(self, cloud):
        c = _cloud(cloud)
        self._n_src = len(c)
        lib().orc_gicp_set_source(self._h, _pf(c), len(c))

    def align(self, guess=None, want_aligned=False):
        g = _colmajor(np.eye(4) if guess is None else guess)
        out = np.empty((self._n_src, 4), dtype=np.float32) if want_aligned else None
        lib().orc_gicp_align(self._h, _pf(g), _pf(out) if want_aligned else None)
        return out

    def hasConverged(self):
        return bool(lib().orc_gicp_converged(self._h))

    def getFinalTransformation(self):
        Tc = np.empty((4, 4), dtype=np.float32)
        lib().orc_gicp_final(self._h, _pf(Tc))
        return Tc.T.copy()

    def getFitnessScore(self, max_range=float("inf")):
        return lib().orc_gicp_fitness(self._h, max_range)

    def getFinalNumIteration(self):
        return lib().orc_gicp_iterations(self._h)

    def getFinalHessian(self):
        H = np.empty((6, 6))
        lib().orc_gicp_hessian(self._h, _pd(H))
        return H

    def covariances(self, which="source"):
        n = self._n_src if which == "source" else self._n_tgt
        out = np.empty((n, 3, 3))
        lib().orc_gicp_covariances(self._h, 0 if which == "source" else 1, _pd(out))
        return out

    def linearize(self, T):
        T = np.ascontiguousarray(T, dtype=np.float64)
        H, b, n = np.empty((6, 6)), np.empty(6), C.c_int(0)
        e = lib().orc_gicp_linearize(self._h, _pd(T), _pd(H), _pd(b), C.byref(n))
        return e, H, b, n.value


class SmallGicp(FastGicp):
    """small_gicp::RegistrationPCL (GICP) restated: the reference's YAML default "SMALL_GICP" (oracle/gicp.h, variant 1)."""

    VARIANT = 1

    def __init__(self, *args, double_search=False, **kwargs):
        """``double_search`` (diagnostic): small_gicp's own correspondence search — double-precision transform and distances — instead of the
        float search the restatement shares with fast_gicp (oracle/gicp.h); tests/test_oracle_gicp.py measures what the difference amounts to."""
        super().__init__(*args, **kwargs)
        lib().orc_gicp_set_double_search(self._h, int(bool(double_search)))


class FastVgicp(FastGicp):
    """fast_gicp::FastVGICP restated (the FAST_VGICP / FAST_VGICP_CUDA algorithm, oracle/gicp.h, variant 2)."""

    VARIANT = 2

    def __init__(self, resolution=1.0, correspondence_randomness=20, transformation_epsilon=0.1, rotation_epsilon=2e-3, maximum_iterations=64, num_threads=1):
        super().__init__(correspondence_randomness, 2.0, transformation_epsilon, rotation_epsilon, maximum_iterations, num_threads)
        lib().orc_gicp_set_resolution(self._h, resolution)

    def numVoxels(self):
        return lib().orc_gicp_num_voxels(self._h)


class Icp(FastGicp):
    """pcl::IterativeClosestPoint restated (oracle/gicp.h, variant 3)."""

    VARIANT = 3

    def __init__(self, max_correspondence_distance=2.0, transformation_epsilon=0.01, maximum_iterations=64, use_reciprocal_correspondences=False):
        super().__init__(20, max_correspondence_distance, transformation_epsilon, 2e-3, maximum_iterations, 1)
        lib().orc_gicp_set_reciprocal(self._h, int(bool(use_reciprocal_correspondences)))


class PclGicp:
    """pcl::GeneralizedIterativeClosestPoint ("GICP") and pclomp::GeneralizedIterativeClosestPoint ("GICP_OMP", ``omp=True``: the older
    whole-gradient-norm stopping rule of the inner BFGS) restated (oracle/pcl_gicp.h), registrations.cpp:93-114."""

    def __init__(self, correspondence_randomness=20, max_correspondence_distance=2.0, transformation_epsilon=0.01, rotation_epsilon=2e-3, maximum_iterations=64,
                 max_optimizer_iterations=20, omp=False, num_threads=1, gpu_order=False, sum_threads=1):
        """``gpu_order``: diagnostic — the cost sums of every BFGS evaluation added in the order the HIP kernels add them.
        ``sum_threads`` T > 1: pclomp's accumulation for T OpenMP threads — per-thread partial sums over the static chunks of the correspondence
        list, added in thread order (what pclomp::GICP computes on a host with omp_get_max_threads() == T); 1: one chain (serial pcl::GICP)."""
        self._h = lib().orc_pclgicp_create()
        lib().orc_pclgicp_set_gpu_order(self._h, int(bool(gpu_order)))
        lib().orc_pclgicp_set_sum_threads(self._h, int(sum_threads))
        lib().orc_pclgicp_set_params(self._h, correspondence_randomness, max_correspondence_distance, transformation_epsilon, rotation_epsilon, maximum_iterations,
                                     max_optimizer_iterations, int(bool(omp)), num_threads)
        self._n_src = self._n_tgt = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().orc_pclgicp_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def setInputTarget(self, cloud):
        c = _cloud(cloud)
        self._n_tgt = len(c)
        lib().orc_pclgicp_set_target(self._h, _pf(c), len(c))

    def setInputSource(self, cloud):
        c = _cloud(cloud)
        self._n_src = len(c)
        lib().orc_pclgicp_set_source(self._h, _pf(c), len(c))

    def align(self, guess=None, want_aligned=False):
        g = _colmajor(np.eye(4) if guess is None else guess)
        out = np.empty((self._n_src, 4), dtype=np.float32) if want_aligned else None
        lib().orc_pclgicp_align(self._h, _pf(g), _pf(out) if want_aligned else None)
        return out

    def hasConverged(self):
        return bool(lib().orc_pclgicp_converged(self._h))

    def getFinalTransformation(self):
        T = np.empty((4, 4), dtype=np.float32)
        lib().orc_pclgicp_final(self._h, _pf(T))
        return T.T.copy()

    def getFitnessScore(self, max_range=float("inf")):
        return lib().orc_pclgicp_fitness(self._h, max_range)

    def getFinalNumIteration(self):
        return lib().orc_pclgicp_iterations(self._h)

    @property
    def evals(self):
        return lib().orc_pclgicp_evaluations(self._h)

    @property
    def inner_steps(self):
        return lib().orc_pclgicp_inner_steps(self._h)

    def covariances(self, which="source"):
        n = self._n_src if which == "source" else self._n_tgt
        out = np.empty((n, 3, 3))
        lib().orc_pclgicp_covariances(self._h, 0 if which == "source" else 1, _pd(out))
        return out

    def evaluate(self, T, x):
        """Cost of estimateRigidTransformationBFGS at x for the correspondences found at transformation T (guess identity): (f, g[6], correspondences)."""
        g, n = np.zeros(6), C.c_int(0)
        f = lib().orc_pclgicp_evaluate(self._h, _pf(_colmajor(T)), _pd(np.ascontiguousarray(x, dtype=np.float64)), _pd(g), C.byref(n))
        return f, g, n.value
