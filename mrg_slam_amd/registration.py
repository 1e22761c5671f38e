"""Host-side mirror of the reference's registration seam, bound to libmrgfe.so through ctypes.

The reference hands out ``pcl::Registration<PointXYZI,PointXYZI>::Ptr`` objects from
``select_registration_method(rclcpp::Node*)`` (/root/reference/include/mrg_slam/registrations.hpp:20,
src/mrg_slam/registrations.cpp:28-152) and its callers use exactly: setInputTarget, setInputSource, align(out, guess),
hasConverged, getFinalTransformation, getFitnessScore(max_range) and getSearchMethodTarget()->nearestKSearch
(SURVEY.md §8b).  The classes below keep those names, argument meanings and the "non-convergence is not an error"
behaviour, so the parity tests read like code written against the reference.

Clouds are N x 4 float32 arrays (x, y, z, intensity); matrices are ordinary row-major 4 x 4 numpy arrays (converted to
the C ABI's column-major layout here).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import GICP_HIP, ICP_HIP, NDT_HIP, PCL_GICP_HIP, PCL_GICP_OMP_HIP, PCL_NDT_HIP, SEARCH, SMALL_GICP_HIP, VGICP_HIP, Context, PairResult, RegParams, check, default_context, lib

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _cloud(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("clouds are N x 4 float32 arrays (x, y, z, intensity)")
    return a


def _colmajor(M) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(M, dtype=np.float32).T)


def default_params(method: int) -> RegParams:
    p = RegParams()
    lib().mrgfe_reg_default_params(method, C.byref(p))
    return p


class HipRegistration:
    """Common pcl::Registration call surface over one ``mrgfe_reg`` handle."""

    METHOD = None

    def __init__(self, params: RegParams, ctx: Context | None = None):
        self._ctx = ctx or default_context()
        self._params = params
        self._h = C.c_void_p()
        check(lib().mrgfe_reg_create(self._ctx._h, C.byref(params), C.byref(self._h)))
        self._n_src = self._n_tgt = 0

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mrgfe_reg_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass

    # -- pcl::Registration ------------------------------------------------------------------------------------
    def setInputTarget(self, cloud) -> int:
        """registration_->setInputTarget(cloud). Returns the library status (0 ok, ERR_OVERFLOW / ERR_EMPTY when PCL
        would abort the voxelisation): like PCL this does not raise for those, the registration simply has no target."""
        c = _cloud(cloud)
        self._n_tgt = len(c)
        st = lib().mrgfe_reg_set_target(self._h, c.ctypes.data_as(_fp), len(c), 16)
        if st < 0 and st not in (_lib.ERR_OVERFLOW, _lib.ERR_EMPTY):
            check(st)
        return st

    def setInputSource(self, cloud) -> None:
        c = _cloud(cloud)
        self._n_src = len(c)
        check(lib().mrgfe_reg_set_source(self._h, c.ctypes.data_as(_fp), len(c), 16))

    def setInputTargetDevice(self, dev_ptr: int, n: int) -> int:
        self._n_tgt = n
        st = lib().mrgfe_reg_set_target_device(self._h, C.c_void_p(dev_ptr), n)
        if st < 0 and st not in (_lib.ERR_OVERFLOW, _lib.ERR_EMPTY):
            check(st)
        return st

    def setInputSourceDevice(self, dev_ptr: int, n: int) -> None:
        self._n_src = n
        check(lib().mrgfe_reg_set_source_device(self._h, C.c_void_p(dev_ptr), n))

    def setInputSourceFromPrefilter(self, dev_ptr: int, n: int) -> None:
        """``setInputSourceDevice`` for the cloud ``prefilter_to_device`` has just left at ``dev_ptr`` on this registration's context (untouched since): the GICP
        family builds the source's search grid inside the box the prefilter chain already knows, without a bounding-box pass and stream wait of its own."""
        self._n_src = n
        check(lib().mrgfe_reg_set_source_from_prefilter(self._h, C.c_void_p(dev_ptr), n))

    def sourceBecomesTarget(self) -> int:
        """``registration_->setInputTarget(keyframe)`` where the keyframe is the cloud last given to ``setInputSource`` (the odometry's keyframe update,
        scan_matching_odometry_component.cpp:326-339): the GICP family keeps the covariances and the search grid it computed for the cloud as a source."""
        self._n_tgt = self._n_src
        st = lib().mrgfe_reg_source_becomes_target(self._h)
        if st < 0 and st not in (_lib.ERR_OVERFLOW, _lib.ERR_EMPTY):
            check(st)
        return st

    def align(self, guess=None, want_aligned: bool = False):
        """registration_->align(*aligned, guess); returns the aligned cloud when ``want_aligned``."""
        g = _colmajor(np.eye(4) if guess is None else guess)
        out = np.empty((self._n_src, 4), dtype=np.float32) if want_aligned else None
        check(lib().mrgfe_reg_align(self._h, g.ctypes.data_as(_fp), out.ctypes.data_as(_fp) if want_aligned else None))
        return out

    def hasConverged(self) -> bool:
        return bool(lib().mrgfe_reg_has_converged(self._h))

    def getFinalTransformation(self) -> np.ndarray:
        Tc = np.empty((4, 4), dtype=np.float32)
        check(lib().mrgfe_reg_final_transformation(self._h, Tc.ctypes.data_as(_fp)))
        return Tc.T.copy()

    def getFitnessScore(self, max_range: float = float("inf")) -> float:
        out = C.c_double(0)
        check(lib().mrgfe_reg_fitness(self._h, max_range, C.byref(out)))
        return out.value

    def nearestKSearch1(self, queries):
        """getSearchMethodTarget()->nearestKSearch(pt, 1, ...) for a whole cloud of queries: (indices, sq. distances)."""
        q = _cloud(queries)
        idx = np.empty(len(q), dtype=np.int32)
        sqd = np.empty(len(q), dtype=np.float32)
        check(lib().mrgfe_reg_nn1_target(self._h, q.ctypes.data_as(_fp), len(q), 16, idx.ctypes.data_as(_ip), sqd.ctypes.data_as(_fp)))
        return idx, sqd

    def getFinalNumIteration(self) -> int:
        return lib().mrgfe_reg_iterations(self._h)

    def getHessian(self) -> np.ndarray:
        H = np.empty((6, 6))
        check(lib().mrgfe_reg_hessian(self._h, H.ctypes.data_as(_dp)))
        return H

    @property
    def evals(self) -> int:
        return lib().mrgfe_reg_evaluations(self._h)

    def kernel_stats(self, mode: int = -1):
        """(device ms, launches, algorithmic bytes) of the dominant kernel in the last align(); NDT: per variant `mode`."""
        ms, n, b = C.c_double(0), C.c_int64(0), C.c_double(0)
        check(lib().mrgfe_reg_kernel_stats(self._h, mode, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value


class NdtHip(HipRegistration):
    """registration_method "NDT_HIP": drop-in for the NDT_OMP branch (registrations.cpp:130-148)."""

    METHOD = NDT_HIP

    def __init__(self, resolution=1.0, transformation_epsilon=0.01, maximum_iterations=64, search="DIRECT7", step_size=0.1, outlier_ratio=0.55,
                 num_threads=0, ctx: Context | None = None):
        p = default_params(NDT_HIP)
        p.resolution = resolution
        p.transformation_epsilon = transformation_epsilon
        p.maximum_iterations = maximum_iterations
        p.nn_search_method = SEARCH[search] if isinstance(search, str) else int(search)
        p.step_size = step_size
        p.outlier_ratio = outlier_ratio
        p.num_threads = num_threads
        super().__init__(p, ctx)

    def getTransformationProbability(self) -> float:
        return lib().mrgfe_reg_trans_probability(self._h)

    @property
    def mean_neighbours(self) -> float:
        return lib().mrgfe_ndt_mean_neighbours(self._h)

    def evaluate(self, T, p, mode=0):
        p = np.ascontiguousarray(p, dtype=np.float64)
        s, g, H = C.c_double(0), np.zeros(6), np.zeros((6, 6))
        check(lib().mrgfe_ndt_evaluate(self._h, _colmajor(T).ctypes.data_as(_fp), p.ctypes.data_as(_dp), mode, C.byref(s), g.ctypes.data_as(_dp), H.ctypes.data_as(_dp)))
        return s.value, g, H

    def leaves(self):
        n = lib().mrgfe_ndt_num_leaves(self._h)
        keys, npts = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        mean, icov = np.empty((n, 3)), np.empty((n, 3, 3))
        if n:
            check(lib().mrgfe_ndt_leaves(self._h, keys.ctypes.data_as(_ip), npts.ctypes.data_as(_ip), mean.ctypes.data_as(_dp), icov.ctypes.data_as(_dp)))
        return keys, npts, mean, icov

    def grid(self):
        a, b, c = (np.empty(3, dtype=np.int32) for _ in range(3))
        check(lib().mrgfe_ndt_grid(self._h, a.ctypes.data_as(_ip), b.ctypes.data_as(_ip), c.ctypes.data_as(_ip)))
        return a, b, c


class PclNdtHip(NdtHip):
    """registration_method "PCL_NDT_HIP": drop-in for the branch every name without "OMP" in it ends in — "NDT" and any unknown string
    (registrations.cpp:115-129): pcl::NormalDistributionsTransform of PCL 1.12.  Pair terms in f64, the radius search over the voxel
    centroids (its only neighbourhood), PCL's iteration test: the squared translation of the last step against the un-squared
    transformation_epsilon — with step_size 0.1 and any epsilon >= 0.01 that is ONE Newton iteration (oracle/quirks.h)."""

    METHOD = PCL_NDT_HIP

    def __init__(self, resolution=1.0, transformation_epsilon=0.1, maximum_iterations=35, step_size=0.1, outlier_ratio=0.55, ctx: Context | None = None):
        p = default_params(PCL_NDT_HIP)
        p.resolution = resolution
        p.transformation_epsilon = transformation_epsilon
        p.maximum_iterations = maximum_iterations
        p.step_size = step_size
        p.outlier_ratio = outlier_ratio
        HipRegistration.__init__(self, p, ctx)

    getTransformationLikelihood = NdtHip.getTransformationProbability  # PCL 1.12's name for score / N


class GicpHip(HipRegistration):
    """registration_method "GICP_HIP": drop-in for the FAST_GICP branch (registrations.cpp:55-63)."""

    METHOD = GICP_HIP

    def __init__(self, correspondence_randomness=20, max_correspondence_distance=2.0, transformation_epsilon=0.01, rotation_epsilon=2e-3,
                 maximum_iterations=64, num_threads=0, ctx: Context | None = None):
        p = default_params(GICP_HIP)
        p.correspondence_randomness = correspondence_randomness
        p.max_correspondence_distance = max_correspondence_distance
        p.transformation_epsilon = transformation_epsilon
        p.rotation_epsilon = rotation_epsilon
        p.maximum_iterations = maximum_iterations
        p.num_threads = num_threads
        super().__init__(p, ctx)

    def linearize(self, T):
        """update_correspondences + linearize at T (4x4, T_target_source): (H, b, sum of r^T M r, correspondences)."""
        Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float64).T)
        H, b, e, n = np.zeros((6, 6)), np.zeros(6), C.c_double(0), C.c_int(0)
        check(lib().mrgfe_gicp_linearize(self._h, Tc.ctypes.data_as(_dp), H.ctypes.data_as(_dp), b.ctypes.data_as(_dp), C.byref(e), C.byref(n)))
        return H, b, e.value, n.value

    def covariances(self, which="source"):
        n = self._n_src if which == "source" else self._n_tgt
        out = np.empty((n, 3, 3))
        check(lib().mrgfe_gicp_covariances(self._h, 0 if which == "source" else 1, out.ctypes.data_as(_dp)))
        return out


class SmallGicpHip(GicpHip):
    """registration_method "SMALL_GICP_HIP": drop-in for the SMALL_GICP branch (registrations.cpp:46-54), the default of
    config/mrg_slam.yaml: small_gicp's GICP factor (right perturbation) and Levenberg-Marquardt schedule."""

    METHOD = SMALL_GICP_HIP

    def __init__(self, correspondence_randomness=20, max_correspondence_distance=2.0, transformation_epsilon=0.01, rotation_epsilon=2e-3,
                 maximum_iterations=64, num_threads=0, ctx: Context | None = None):
        p = default_params(SMALL_GICP_HIP)
        p.correspondence_randomness = correspondence_randomness
        p.max_correspondence_distance = max_correspondence_distance
        p.transformation_epsilon = transformation_epsilon
        p.rotation_epsilon = rotation_epsilon
        p.maximum_iterations = maximum_iterations
        p.num_threads = num_threads
        HipRegistration.__init__(self, p, ctx)


class VgicpHip(GicpHip):
    """registration_method "VGICP_HIP": drop-in for the FAST_VGICP branch (registrations.cpp:76-84) and for the reference's own
    GPU slot FAST_VGICP_CUDA (:65-75): fast_gicp's voxelised GICP, the target as a Gaussian voxel map of edge ``resolution``."""

    METHOD = VGICP_HIP

    def __init__(self, resolution=1.0, correspondence_randomness=20, transformation_epsilon=0.01, rotation_epsilon=2e-3, maximum_iterations=64, num_threads=0,
                 ctx: Context | None = None):
        p = default_params(VGICP_HIP)
        p.resolution = resolution
        p.correspondence_randomness = correspondence_randomness
        p.transformation_epsilon = transformation_epsilon
        p.rotation_epsilon = rotation_epsilon
        p.maximum_iterations = maximum_iterations
        p.num_threads = num_threads
        HipRegistration.__init__(self, p, ctx)


class IcpHip(HipRegistration):
    """registration_method "ICP_HIP": drop-in for the ICP branch (registrations.cpp:85-92), pcl::IterativeClosestPoint with
    TransformationEstimationSVD and the default convergence criteria; ``use_reciprocal_correspondences`` as in :91."""

    METHOD = ICP_HIP

    def __init__(self, max_correspondence_distance=2.0, transformation_epsilon=0.01, maximum_iterations=64, use_reciprocal_correspondences=False, ctx: Context | None = None):
        p = default_params(ICP_HIP)
        p.max_correspondence_distance = max_correspondence_distance
        p.transformation_epsilon = transformation_epsilon
        p.maximum_iterations = maximum_iterations
        p.use_reciprocal_correspondences = int(bool(use_reciprocal_correspondences))
        super().__init__(p, ctx)


class PclGicpHip(HipRegistration):
    """registration_method "GICP" / "PCL_GICP_HIP" (pcl::GeneralizedIterativeClosestPoint, registrations.cpp:93-103) and, with ``omp=True``,
    "GICP_OMP" / "PCL_GICP_OMP_HIP" (pclomp::GeneralizedIterativeClosestPoint, :104-114): PCL's covariances and nearest-point correspondences on
    the GPU, the inner BFGS (``max_optimizer_iterations`` steps) on the host over 13 GPU sums per evaluation.  ``use_reciprocal_correspondences``
    is accepted and has no effect, as upstream: pcl::GICP's computeTransformation runs its own search loop."""

    METHOD = PCL_GICP_HIP

    def __init__(self, correspondence_randomness=20, max_correspondence_distance=2.0, transformation_epsilon=0.01, rotation_epsilon=2e-3, maximum_iterations=64,
                 max_optimizer_iterations=20, use_reciprocal_correspondences=False, omp=False, num_threads=0, ctx: Context | None = None):
        """``num_threads`` (``omp=True`` only): the number of OpenMP threads whose accumulation is reproduced — pclomp adds per-thread partial sums over
        static chunks of the correspondences, so its result depends on omp_get_max_threads() of the reference's host; 0 = 8 (the YAML's reg_num_threads)."""
        p = default_params(PCL_GICP_OMP_HIP if omp else PCL_GICP_HIP)
        p.num_threads = num_threads
        p.correspondence_randomness = correspondence_randomness
        p.max_correspondence_distance = max_correspondence_distance
        p.transformation_epsilon = transformation_epsilon
        p.rotation_epsilon = rotation_epsilon
        p.maximum_iterations = maximum_iterations
        p.max_optimizer_iterations = max_optimizer_iterations
        p.use_reciprocal_correspondences = int(bool(use_reciprocal_correspondences))
        super().__init__(p, ctx)

    def covariances(self, which="source"):
        n = self._n_src if which == "source" else self._n_tgt
        out = np.empty((n, 3, 3))
        check(lib().mrgfe_gicp_covariances(self._h, 0 if which == "source" else 1, out.ctypes.data_as(_dp)))
        return out

    def evaluate(self, T, x):
        """The correspondences pcl::GICP finds at transformation T (guess identity) and the cost estimateRigidTransformationBFGS minimises at x
        over them: (f, g[6], correspondences)."""
        Tc = _colmajor(T)
        xx = np.ascontiguousarray(x, dtype=np.float64)
        f, g, n = C.c_double(0), np.zeros(6), C.c_int(0)
        check(lib().mrgfe_pclgicp_evaluate(self._h, Tc.ctypes.data_as(_fp), xx.ctypes.data_as(_dp), C.byref(f), g.ctypes.data_as(_dp), C.byref(n)))
        return f.value, g, n.value


def select_registration_method(params: dict, ctx: Context | None = None) -> HipRegistration:
    """Python mirror of mrg_slam::select_registration_method (registrations.cpp:28-152) for the HIP back ends.

    ``params`` carries the reference's ROS parameter names (registration_method, reg_num_threads,
    reg_transformation_epsilon, reg_maximum_iterations, reg_max_correspondence_distance, reg_correspondence_randomness,
    reg_resolution, reg_nn_search_method).  "NDT_HIP" (and, to stay drop-in, "NDT_OMP"/"NDT") select :class:`NdtHip`;
    "GICP_HIP" / "FAST_GICP" select :class:`GicpHip`, "SMALL_GICP_HIP" / "SMALL_GICP" :class:`SmallGicpHip`, "VGICP_HIP" /
    "FAST_VGICP" / "FAST_VGICP_CUDA" :class:`VgicpHip`.  Like the reference, an unknown name falls through to NDT: names
    without "OMP" in them ("NDT", "PCL_NDT_HIP", or any unknown string) reach pcl::NormalDistributionsTransform there (:115-129) and
    select :class:`PclNdtHip`, its f64 formulation; "ICP" / "ICP_HIP"
    select :class:`IcpHip`; "GICP" / "GICP_OMP" (:93-114) select :class:`PclGicpHip` (pcl::GeneralizedIterativeClosestPoint / pclomp::GICP with
    the BFGS inner optimiser).
    """
    method = str(params.get("registration_method", "FAST_GICP"))
    eps = float(params.get("reg_transformation_epsilon", 0.01))
    iters = int(params.get("reg_maximum_iterations", 64))
    threads = int(params.get("reg_num_threads", 0))
    if method in ("VGICP_HIP", "FAST_VGICP", "FAST_VGICP_CUDA"):
        return VgicpHip(float(params.get("reg_resolution", 1.0)), int(params.get("reg_correspondence_randomness", 20)), eps, maximum_iterations=iters,
                        num_threads=threads, ctx=ctx)
    if method in ("SMALL_GICP_HIP", "SMALL_GICP"):
        return SmallGicpHip(int(params.get("reg_correspondence_randomness", 20)), float(params.get("reg_max_correspondence_distance", 2.0)), eps,
                            maximum_iterations=iters, num_threads=threads, ctx=ctx)
    if method in ("GICP_HIP", "FAST_GICP"):
        return GicpHip(int(params.get("reg_correspondence_randomness", 20)), float(params.get("reg_max_correspondence_distance", 2.0)), eps,
                       maximum_iterations=iters, num_threads=threads, ctx=ctx)
    if method in ("ICP", "ICP_HIP"):
        return IcpHip(float(params.get("reg_max_correspondence_distance", 2.0)), eps, iters, bool(params.get("reg_use_reciprocal_correspondences", False)), ctx=ctx)
    if "GICP" in method:
        # registrations.cpp:93-114: any other name with "GICP" in it is pcl::GeneralizedIterativeClosestPoint, with "OMP" in it as well pclomp's
        return PclGicpHip(int(params.get("reg_correspondence_randomness", 20)), float(params.get("reg_max_correspondence_distance", 2.0)), eps, maximum_iterations=iters,
                          max_optimizer_iterations=int(params.get("reg_max_optimizer_iterations", 20)),
                          use_reciprocal_correspondences=bool(params.get("reg_use_reciprocal_correspondences", False)), omp="OMP" in method,
                          num_threads=threads, ctx=ctx)  # (pclomp::GICP never sees reg_num_threads upstream: it sums over omp_get_max_threads() threads; that number is stated here)
    search = str(params.get("reg_nn_search_method", "DIRECT7"))
    if search not in ("KDTREE", "DIRECT1"):
        search = "DIRECT7"  # registrations.cpp:140-146: anything else means DIRECT7
    if "OMP" not in method and method != "NDT_HIP":
        # :115-129: every name without "OMP" in it — "NDT" and any unknown string alike — ends in pcl::NormalDistributionsTransform with
        # setTransformationEpsilon / setMaximumIterations / setResolution (:125-127); reg_nn_search_method and reg_num_threads are not read
        return PclNdtHip(float(params.get("reg_resolution", 1.0)), eps, iters, ctx=ctx)
    return NdtHip(float(params.get("reg_resolution", 1.0)), eps, iters, search, num_threads=threads, ctx=ctx)


class BatchMatcher:
    """Batched candidate matching: the candidate loop of LoopDetector::matching (src/mrg_slam/loop_detector.cpp:126-145)
    advanced for all candidates at once on one GPU (mrgfe_batch_*)."""

    def __init__(self, params: RegParams | None = None, ctx: Context | None = None, **ndt_kwargs):
        self._ctx = ctx or default_context()
        if params is None:
            params = default_params(NDT_HIP)
            params.transformation_epsilon = ndt_kwargs.get("transformation_epsilon", 0.01)
            params.maximum_iterations = ndt_kwargs.get("maximum_iterations", 64)
            params.resolution = ndt_kwargs.get("resolution", 1.0)
            params.nn_search_method = SEARCH[ndt_kwargs.get("search", "DIRECT7")]
        self._h = C.c_void_p()
        check(lib().mrgfe_batch_create(self._ctx._h, C.byref(params), C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mrgfe_batch_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass

    def clear(self):
        check(lib().mrgfe_batch_clear(self._h))

    def add_target(self, cloud) -> int:
        c = _cloud(cloud)
        return check(lib().mrgfe_batch_add_target(self._h, c.ctypes.data_as(_fp), len(c), 16))

    def add_target_records(self, records, n: int, layout: int) -> int:
        """A target held as strided point records (``layout`` = ``_lib.layout(...)``, e.g. ``LAYOUT_PCL_XYZI`` for the reference's
        in-memory pcl::PointXYZI): uploaded raw and gathered on the device."""
        buf = np.ascontiguousarray(records)
        return check(lib().mrgfe_batch_add_target(self._h, buf.ctypes.data_as(_fp), n, layout))

    def add_target_device(self, dev_ptr: int, n: int) -> int:
        return check(lib().mrgfe_batch_add_target_device(self._h, C.c_void_p(dev_ptr), n))

    def add_pair(self, target: int, source, guess=None, key: int = 0) -> int:
        """``key`` != 0 (the candidate's keyframe id): the cloud — and for the GICP methods its covariances — stay in the
        batch's HBM keyframe store across ``clear()``; once ``has_cloud(key)`` a later call may pass ``source=None``."""
        g = _colmajor(np.eye(4) if guess is None else guess)
        if key:
            if source is None:
                n = self.has_cloud(key)
                if n is None:
                    raise KeyError(f"keyframe {key} is not in the store")
                return check(lib().mrgfe_batch_add_pair_keyed(self._h, target, key, None, n, 16, g.ctypes.data_as(_fp)))
            c = _cloud(source)
            return check(lib().mrgfe_batch_add_pair_keyed(self._h, target, key, c.ctypes.data_as(_fp), len(c), 16, g.ctypes.data_as(_fp)))
        c = _cloud(source)
        return check(lib().mrgfe_batch_add_pair(self._h, target, c.ctypes.data_as(_fp), len(c), 16, g.ctypes.data_as(_fp)))

    def has_cloud(self, key: int):
        """Point count of the stored keyframe ``key``, or None."""
        n = C.c_size_t(0)
        return n.value if lib().mrgfe_batch_has_cloud(self._h, key, C.byref(n)) else None

    def store_bytes(self) -> int:
        return int(lib().mrgfe_batch_store_bytes(self._h))

    def timing(self, reset: bool = False) -> dict:
        """``mrgfe_batch_timing``: the reference's ``average_time_per_candidate_us`` (apps/mrg_slam_component.cpp:1032-1037) for this batch object."""
        v = (C.c_double * 4)()
        check(lib().mrgfe_batch_timing(self._h, v))
        if reset:
            check(lib().mrgfe_batch_timing_reset(self._h))
        return {"average_time_per_candidate_us": float(v[0]), "last_align_us": float(v[1]), "last_align_pairs": int(v[2]), "total_pairs": int(v[3])}

    def forget(self, key: int = 0) -> None:
        check(lib().mrgfe_batch_forget(self._h, key))

    def add_pair_device(self, target: int, dev_ptr: int, n: int, guess=None) -> int:
        g = _colmajor(np.eye(4) if guess is None else guess)
        return check(lib().mrgfe_batch_add_pair_device(self._h, target, C.c_void_p(dev_ptr), n, g.ctypes.data_as(_fp)))

    def add_device(self, target_ptrs, target_points, pair_target, source_ptrs, source_points, guesses) -> int:
        """Many device-resident targets and pairs in one call (``mrgfe_batch_add_device``): ``pair_target[i]`` indexes
        ``target_ptrs``; ``guesses`` is [n_pairs, 4, 4] row-major.  Returns the index of the first pair added."""
        nt, npair = len(target_ptrs), len(source_ptrs)
        tp = (C.c_void_p * max(nt, 1))(*[int(p) for p in target_ptrs])
        tn = (C.c_size_t * max(nt, 1))(*[int(n) for n in target_points])
        sp = (C.c_void_p * max(npair, 1))(*[int(p) for p in source_ptrs])
        sn = (C.c_size_t * max(npair, 1))(*[int(n) for n in source_points])
        pt = np.ascontiguousarray(pair_target, dtype=np.int32)
        g = np.ascontiguousarray(np.asarray(guesses, dtype=np.float32).reshape(npair, 4, 4).transpose(0, 2, 1))  # column-major per pair
        return check(lib().mrgfe_batch_add_device(self._h, nt, tp, tn, npair, pt.ctypes.data_as(_ip), sp, sn, g.ctypes.data_as(_fp)))

    def rounds(self) -> int:
        """Rounds (plan -> derivative launches -> reduce / controller step) of the last NDT align()."""
        return int(lib().mrgfe_batch_rounds(self._h))

    def set_guess(self, pair: int, guess) -> None:
        check(lib().mrgfe_batch_set_guess(self._h, pair, _colmajor(guess).ctypes.data_as(_fp)))

    def build_targets(self) -> None:
        check(lib().mrgfe_batch_build_targets(self._h))

    def align(self, fitness_max_range: float = -1.0):
        """Returns a structured numpy array of mrgfe_pair_result records (one per pair)."""
        n = lib().mrgfe_batch_num_pairs(self._h)
        res = (PairResult * max(n, 1))()
        check(lib().mrgfe_batch_align(self._h, fitness_max_range, res))
        return results_to_numpy(res, n)

    def align_async(self, fitness_max_range: float = -1.0):
        """``mrgfe_batch_align_async``: the align runs on the batch's worker thread; :meth:`wait` returns its records.  Keep two BatchMatchers on two
        contexts in flight to overlap one batch's build and straggler rounds with the other's derivative launches."""
        n = lib().mrgfe_batch_num_pairs(self._h)
        buf = (PairResult * max(n, 1))()
        check(lib().mrgfe_batch_align_async(self._h, fitness_max_range, buf))  # (a refused call must not replace the buffer a running align writes into)
        self._async = (buf, n)

    def wait(self):
        res, n = self._async
        try:
            check(lib().mrgfe_batch_wait(self._h))
        finally:
            self._async = None
        return results_to_numpy(res, n)

    def pair_counts(self, mode: int = -1):
        """(source points, valid point-voxel pairs) of the derivative evaluations the last align launched."""
        p, n = C.c_double(0), C.c_double(0)
        check(lib().mrgfe_batch_pair_counts(self._h, mode, C.byref(p), C.byref(n)))
        return p.value, n.value

    def largest_launch(self):
        """(device ms, [busy pairs per evaluation kind]) of the longest timed derivative launch of the last align()."""
        v = (C.c_double * 4)()
        check(lib().mrgfe_batch_largest_launch(self._h, v))
        return float(v[0]), [int(v[1]), int(v[2]), int(v[3])]

    def fitness_stats(self) -> dict:
        """getFitnessScore passes of the last align(), all launches added up (``mrgfe_batch_fitness_stats``)."""
        v = (C.c_double * 11)()
        check(lib().mrgfe_batch_fitness_stats(self._h, v))
        keys = ("ms_block", "ms_sweep", "ms_far", "queries", "queued", "queued_far", "words", "boxes_tested", "cells", "points", "launches")
        return dict(zip(keys, [float(x) for x in v]))

    def kernel_stats(self, mode: int = -1):
        """(device ms, launches, algorithmic bytes) of the derivative kernel variant `mode` (-1: all) in the last align()."""
        ms, n, b = C.c_double(0), C.c_int64(0), C.c_double(0)
        check(lib().mrgfe_batch_kernel_stats(self._h, mode, C.byref(ms), C.byref(n), C.byref(b)))
        return ms.value, n.value, b.value


class NodeMatcher:
    """``mrgfe_node_*``: the candidate batch of LoopDetector::matching (loop_detector.cpp:104,126-145) over several GPUs of one machine from ONE
    process — a member (context + batch + host thread) per entry of ``devices``; the same ordinal may appear more than once.  The pair list is cut
    into contiguous blocks, the 384-byte records come back in pair order (RCCL all-gather between distinct devices, host memory otherwise); they equal
    the records of one :class:`BatchMatcher` holding the whole list bit for bit."""

    def __init__(self, devices, params: RegParams | None = None, **ndt_kwargs):
        if params is None:
            params = default_params(NDT_HIP)
            params.transformation_epsilon = ndt_kwargs.get("transformation_epsilon", 0.01)
            params.maximum_iterations = ndt_kwargs.get("maximum_iterations", 64)
            params.resolution = ndt_kwargs.get("resolution", 1.0)
            params.nn_search_method = SEARCH[ndt_kwargs.get("search", "DIRECT7")]
        dev = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._h = C.c_void_p()
        check(lib().mrgfe_node_create(len(devices), dev, C.byref(params), C.byref(self._h)))
        self._keep = []  # the add calls only REFERENCE the clouds: keep them alive until align() returns

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().mrgfe_node_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass

    close = __del__

    @property
    def n_members(self) -> int:
        return lib().mrgfe_node_num_members(self._h)

    def clear(self):
        check(lib().mrgfe_node_clear(self._h))
        self._keep = []

    def add_target(self, cloud, key: int = 0, n_points: int | None = None) -> int:
        """``cloud=None``: a keyed target that is resident on the member that will need it (same key and point count as before)."""
        if cloud is None:
            return check(lib().mrgfe_node_add_target_keyed(self._h, key, None, int(n_points or 0), 16))
        c = _cloud(cloud)
        self._keep.append(c)
        return check(lib().mrgfe_node_add_target_keyed(self._h, key, c.ctypes.data_as(_fp), len(c), 16))

    def add_pair(self, target: int, source, guess=None, key: int = 0, n_points: int | None = None) -> int:
        g = _colmajor(np.eye(4) if guess is None else guess)
        if source is None:
            return check(lib().mrgfe_node_add_pair_keyed(self._h, target, key, None, int(n_points or 0), 16, g.ctypes.data_as(_fp)))
        c = _cloud(source)
        self._keep.append(c)
        return check(lib().mrgfe_node_add_pair_keyed(self._h, target, key, c.ctypes.data_as(_fp), len(c), 16, g.ctypes.data_as(_fp)))

    def align(self, fitness_max_range: float = -1.0):
        n = lib().mrgfe_node_num_pairs(self._h)
        res = (PairResult * max(n, 1))()
        try:
            check(lib().mrgfe_node_align(self._h, fitness_max_range, res))
        finally:
            self._keep = []
        return results_to_numpy(res, n)

    def has_cloud(self, key: int):
        """The BatchMatcher call surface (LoopDetector's matcher): a node cannot say which member will need a key before the pair list is complete, so the
        caller always hands the cloud over — the member that gets the pair uploads it only when the key is not resident there with the same point count."""
        return None

    def shard(self, member: int):
        a, b = C.c_int(0), C.c_int(0)
        check(lib().mrgfe_node_shard(self._h, member, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_gather(self) -> str:
        return "rccl" if lib().mrgfe_node_last_gather(self._h) == 1 else "host"

    def forget(self, key: int = 0):
        check(lib().mrgfe_node_forget(self._h, key))

    def store_bytes(self) -> int:
        return int(lib().mrgfe_node_store_bytes(self._h))

    def fail_member_once(self, member: int):
        if not hasattr(lib(), "mrgfe_dbg_node_fail_member"):
            raise RuntimeError("mrgfe_dbg_node_fail_member exists only in the -DMRGFE_TESTING build (MRGFE_LIB=mrg_slam_amd/libmrgfe_testing.so)")
        check(lib().mrgfe_dbg_node_fail_member(self._h, member))

    @staticmethod
    def select_best(records: np.ndarray, group_first):
        """``mrgfe_node_select_best``: [(position within the group or None, score)] for the groups ``group_first[g] .. group_first[g + 1]``."""
        gf = np.ascontiguousarray(group_first, dtype=np.int32)
        ng = len(gf) - 1
        rec = np.ascontiguousarray(records)
        best, score = np.empty(max(ng, 1), dtype=np.int32), np.empty(max(ng, 1))
        check(lib().mrgfe_node_select_best(C.cast(rec.ctypes.data, C.POINTER(PairResult)), ng, gf.ctypes.data_as(_ip), best.ctypes.data_as(_ip), score.ctypes.data_as(_dp)))
        return [(int(best[g]) if best[g] >= 0 else None, float(score[g])) for g in range(ng)]


RESULT_DTYPE = np.dtype([("T", np.float32, (16,)), ("H", np.float64, (36,)), ("fitness", np.float64), ("trans_probability", np.float64),
                         ("converged", np.int32), ("iterations", np.int32), ("evaluations", np.int32), ("pair_id", np.int32)])
assert RESULT_DTYPE.itemsize == 384


def results_to_numpy(res, n: int) -> np.ndarray:
    return np.frombuffer(bytes(res), dtype=RESULT_DTYPE, count=n).copy()


def result_matrix(rec) -> np.ndarray:
    """Row-major 4 x 4 transformation of one result record (stored column-major)."""
    return np.asarray(rec["T"], dtype=np.float32).reshape(4, 4).T.copy()
