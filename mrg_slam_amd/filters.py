"""Host-side mirror of the prefilter objects the reference's PrefilteringComponent holds
(/root/reference/apps/prefiltering_component.cpp:158-229): pcl::VoxelGrid, pcl::RadiusOutlierRemoval,
pcl::StatisticalOutlierRemoval plus its in-tree distance_filter, bound to libmrgfe.so.  The pcl::Filter call surface
(setters, setInputCloud, filter) is kept."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Context, check, default_context, lib

_fp = C.POINTER(C.c_float)


def _cloud(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("clouds are N x 4 float32 arrays (x, y, z, intensity)")
    return a


class _Filter:
    def __init__(self, ctx: Context | None = None):
        self._ctx = ctx or default_context()
        self._input = np.zeros((0, 4), dtype=np.float32)

    def setInputCloud(self, cloud):
        self._input = _cloud(cloud)

    def filter(self) -> np.ndarray:
        raise NotImplementedError


class VoxelGrid(_Filter):
    """pcl::VoxelGrid<PointXYZI> (prefiltering_component.cpp:168-171; scan_matching_odometry_component.cpp:176-179)."""

    def __init__(self, ctx=None):
        super().__init__(ctx)
        self._leaf, self._min_pts = 0.1, 1
        self.overflow = False

    def setLeafSize(self, lx, ly=None, lz=None):
        if (ly is not None and ly != lx) or (lz is not None and lz != lx):
            raise ValueError("the reference only uses cubic leaves (setLeafSize(r, r, r))")
        self._leaf = float(lx)

    def setMinimumPointsNumberPerVoxel(self, n):
        self._min_pts = int(n)

    def filter(self):
        c = self._input
        out = np.empty_like(c)
        m, ov = C.c_size_t(0), C.c_int(0)
        check(lib().mrgfe_voxelgrid(self._ctx._h, c.ctypes.data_as(_fp), len(c), 16, self._leaf, self._min_pts, out.ctypes.data_as(_fp), C.byref(m), C.byref(ov)))
        self.overflow = bool(ov.value)
        return out[: m.value].copy()


class ApproximateVoxelGrid(_Filter):
    """pcl::ApproximateVoxelGrid<PointXYZI> (downsample_method APPROX_VOXELGRID: prefiltering_component.cpp:172-175;
    scan_matching_odometry_component.cpp:180-183): order dependent, a cell may be emitted more than once, no minimum point count."""

    def __init__(self, ctx=None):
        super().__init__(ctx)
        self._leaf = 0.1

    def setLeafSize(self, lx, ly=None, lz=None):
        if (ly is not None and ly != lx) or (lz is not None and lz != lx):
            raise ValueError("the reference only uses cubic leaves (setLeafSize(r, r, r))")
        self._leaf = float(lx)

    def filter(self):
        c = self._input
        out = np.empty((max(len(c), 1), 4), dtype=np.float32)
        m = C.c_size_t(0)
        check(lib().mrgfe_approx_voxelgrid(self._ctx._h, c.ctypes.data_as(_fp), len(c), 16, self._leaf, out.ctypes.data_as(_fp), C.byref(m)))
        return out[: m.value].copy()


class RadiusOutlierRemoval(_Filter):
    """pcl::RadiusOutlierRemoval<PointXYZI> (prefiltering_component.cpp:195-198)."""

    def __init__(self, ctx=None):
        super().__init__(ctx)
        self._radius, self._min_neighbors = 0.5, 2

    def setRadiusSearch(self, r):
        self._radius = float(r)

    def setMinNeighborsInRadius(self, n):
        self._min_neighbors = int(n)

    def filter(self):
        c = self._input
        out = np.empty_like(c)
        m = C.c_size_t(0)
        check(lib().mrgfe_radius_outlier(self._ctx._h, c.ctypes.data_as(_fp), len(c), 16, self._radius, self._min_neighbors, out.ctypes.data_as(_fp), C.byref(m)))
        return out[: m.value].copy()


class StatisticalOutlierRemoval(_Filter):
    """pcl::StatisticalOutlierRemoval<PointXYZI> (prefiltering_component.cpp:189-192)."""

    def __init__(self, ctx=None):
        super().__init__(ctx)
        self._mean_k, self._stddev = 30, 1.2

    def setMeanK(self, k):
        self._mean_k = int(k)

    def setStddevMulThresh(self, s):
        self._stddev = float(s)

    def filter(self):
        c = self._input
        out = np.empty_like(c)
        m = C.c_size_t(0)
        check(lib().mrgfe_statistical_outlier(self._ctx._h, c.ctypes.data_as(_fp), len(c), 16, self._mean_k, self._stddev, out.ctypes.data_as(_fp), C.byref(m)))
        return out[: m.value].copy()


def distance_filter(cloud, near_thresh=0.1, far_thresh=35.0, ctx: Context | None = None) -> np.ndarray:
    """PrefilteringComponent::distance_filter (prefiltering_component.cpp:206-229)."""
    ctx = ctx or default_context()
    c = _cloud(cloud)
    out = np.empty_like(c)
    m = C.c_size_t(0)
    check(lib().mrgfe_distance_filter(ctx._h, c.ctypes.data_as(_fp), len(c), 16, near_thresh, far_thresh, out.ctypes.data_as(_fp), C.byref(m)))
    return out[: m.value].copy()


def calc_fitness_score(cloud1, cloud2, relpose, max_range=float("inf"), ctx: Context | None = None) -> float:
    """InformationMatrixCalculator::calc_fitness_score (src/mrg_slam/information_matrix_calculator.cpp:46-81)."""
    ctx = ctx or default_context()
    c1, c2 = _cloud(cloud1), _cloud(cloud2)
    T = np.ascontiguousarray(np.asarray(relpose, dtype=np.float64).T)
    out = C.c_double(0)
    check(lib().mrgfe_calc_fitness_score(ctx._h, c1.ctypes.data_as(_fp), len(c1), c2.ctypes.data_as(_fp), len(c2), 16, T.ctypes.data_as(C.POINTER(C.c_double)), max_range,
                                         C.byref(out)))
    return out.value


class InformationMatrixCalculator:
    """mrg_slam::InformationMatrixCalculator (src/mrg_slam/information_matrix_calculator.cpp): the 6x6 information matrix of a
    graph edge from the fitness score of its two keyframe clouds.  ``params`` carries the reference's ROS parameter names
    (use_const_inf_matrix, const_stddev_x/q, var_gain_a, min/max_stddev_x/q, fitness_score_thresh; YAML defaults otherwise)."""

    def __init__(self, params: dict | None = None, ctx: Context | None = None):
        self._ctx = ctx or default_context()
        self._p = _lib.InfParams()
        lib().mrgfe_inf_default_params(C.byref(self._p))
        for k, v in (params or {}).items():
            if not hasattr(self._p, k):
                raise KeyError(k)
            setattr(self._p, k, int(bool(v)) if k == "use_const_inf_matrix" else float(v))
        self.last_fitness_score = None

    calc_fitness_score = staticmethod(calc_fitness_score)

    @staticmethod
    def weight(a, max_x, min_y, max_y, x) -> float:
        return lib().mrgfe_inf_weight(a, max_x, min_y, max_y, x)

    def from_fitness(self, fitness_score: float) -> np.ndarray:
        inf = np.empty((6, 6))
        check(lib().mrgfe_inf_matrix_from_fitness(C.byref(self._p), fitness_score, inf.ctypes.data_as(C.POINTER(C.c_double))))
        return inf

    def calc_information_matrix(self, cloud1, cloud2, relpose) -> np.ndarray:
        c1, c2 = _cloud(cloud1), _cloud(cloud2)
        T = np.ascontiguousarray(np.asarray(relpose, dtype=np.float64).T)
        inf, fit = np.empty((6, 6)), C.c_double(0)
        dp = C.POINTER(C.c_double)
        check(lib().mrgfe_calc_information_matrix(self._ctx._h, C.byref(self._p), c1.ctypes.data_as(_fp), len(c1), c2.ctypes.data_as(_fp), len(c2), 16, T.ctypes.data_as(dp),
                                                  inf.ctypes.data_as(dp), C.byref(fit)))
        self.last_fitness_score = fit.value
        return inf

    def calc_information_matrix_keyed(self, store, key1: int, key2: int, relpose) -> np.ndarray:
        """The same for two keyframes of a :class:`MapCloudStore`: nothing is uploaded, key1's search grid is kept for its next edges."""
        T = np.ascontiguousarray(np.asarray(relpose, dtype=np.float64).T)
        inf, fit = np.empty((6, 6)), C.c_double(0)
        dp = C.POINTER(C.c_double)
        check(lib().mrgfe_map_store_information_matrix(store._h, C.byref(self._p), int(key1), int(key2), T.ctypes.data_as(dp), inf.ctypes.data_as(dp), C.byref(fit)))
        self.last_fitness_score = fit.value
        return inf


def knn(cloud, queries, k: int, ctx: Context | None = None):
    """pcl::search::KdTree::nearestKSearch(pt, k) for a cloud of queries: (indices [nq, k], squared distances [nq, k]),
    ascending by (distance, index), -1 where the cloud has fewer than k points."""
    ctx = ctx or default_context()
    c, q = _cloud(cloud), _cloud(queries)
    idx, sqd = np.empty((len(q), k), dtype=np.int32), np.empty((len(q), k), dtype=np.float32)
    check(lib().mrgfe_knn(ctx._h, c.ctypes.data_as(_fp), len(c), q.ctypes.data_as(_fp), len(q), 16, k, idx.ctypes.data_as(C.POINTER(C.c_int32)), sqd.ctypes.data_as(_fp)))
    return idx, sqd


def grid_set_query(clouds, queries, k: int, rounds: int = 1, ctx: Context | None = None):
    """Diagnostic (``mrgfe_dbg_grid_set_query``): search grids over several clouds built together, the queries answered against each.
    Returns (indices [len(clouds), nq, k], squared distances likewise); k == 1 is the exact 1-NN search, k > 1 the k-NN search."""
    ctx = ctx or default_context()
    cs = [_cloud(c) for c in clouds]
    q = _cloud(queries)
    ptrs = (_fp * len(cs))(*[c.ctypes.data_as(_fp) for c in cs])
    ns = (C.c_size_t * len(cs))(*[len(c) for c in cs])
    idx, sqd = np.empty((len(cs), len(q), k), dtype=np.int32), np.empty((len(cs), len(q), k), dtype=np.float32)
    check(lib().mrgfe_dbg_grid_set_query(ctx._h, ptrs, ns, len(cs), q.ctypes.data_as(_fp), len(q), k, rounds, idx.ctypes.data_as(C.POINTER(C.c_int32)), sqd.ctypes.data_as(_fp)))
    return idx, sqd


def _prefilter_params(p: dict) -> "_lib.PrefilterParams":
    q = _lib.PrefilterParams()
    lib().mrgfe_prefilter_default_params(C.byref(q))
    q.enable_distance_filter = int(bool(p["enable_distance_filter"]))
    q.distance_near_thresh, q.distance_far_thresh = p["distance_near_thresh"], p["distance_far_thresh"]
    q.downsample_method = {"NONE": 0, "VOXELGRID": 1, "APPROX_VOXELGRID": 2}[p["downsample_method"]]
    q.downsample_resolution, q.downsample_min_points_per_voxel = p["downsample_resolution"], p["downsample_min_points_per_voxel"]
    q.outlier_removal_method = {"NONE": 0, "RADIUS": 1, "STATISTICAL": 2}[p["outlier_removal_method"]]
    q.radius_radius, q.radius_min_neighbors = p["radius_radius"], p["radius_min_neighbors"]
    q.statistical_mean_k, q.statistical_stddev = p["statistical_mean_k"], p["statistical_stddev"]
    return q


_PREFILTER_DEFAULTS = {"enable_distance_filter": True, "distance_near_thresh": 0.1, "distance_far_thresh": 35.0, "downsample_method": "VOXELGRID",
                       "downsample_resolution": 0.1, "downsample_min_points_per_voxel": 1, "outlier_removal_method": "RADIUS", "radius_radius": 0.5,
                       "radius_min_neighbors": 2, "statistical_mean_k": 30, "statistical_stddev": 1.2}


def prefilter_to_device(cloud, dev_ptr: int, capacity: int, params: dict | None = None, ctx: Context | None = None) -> int:
    """The prefiltering chain with the filtered scan left in device memory at ``dev_ptr`` (packed float4, room for
    ``capacity`` >= len(cloud) points), ready for setInputSourceDevice / add_pair_device.  Returns the point count."""
    p = dict(_PREFILTER_DEFAULTS)
    p.update(params or {})
    c = _cloud(cloud)
    if capacity < len(c):
        raise ValueError("device buffer too small")
    ctx = ctx or default_context()
    q, m = _prefilter_params(p), C.c_size_t(0)
    check(lib().mrgfe_prefilter_device(ctx._h, C.byref(q), c.ctypes.data_as(_fp), len(c), 16, C.c_void_p(dev_ptr), C.byref(m)))
    return m.value


def prefilter(cloud, params: dict | None = None, ctx: Context | None = None) -> np.ndarray:
    """The chain of PrefilteringComponent::cloud_callback (:149-151) with the reference's parameter names and YAML
    defaults (config/mrg_slam.yaml:41-64): distance_filter -> downsample -> outlier_removal."""
    p = {"enable_distance_filter": True, "distance_near_thresh": 0.1, "distance_far_thresh": 35.0, "downsample_method": "VOXELGRID", "downsample_resolution": 0.1,
         "downsample_min_points_per_voxel": 1, "outlier_removal_method": "RADIUS", "radius_radius": 0.5, "radius_min_neighbors": 2, "statistical_mean_k": 30,
         "statistical_stddev": 1.2}
    p.update(params or {})
    c = _cloud(cloud)
    if p["downsample_method"] in ("VOXELGRID", "APPROX_VOXELGRID", "NONE") and p["outlier_removal_method"] in ("RADIUS", "STATISTICAL", "NONE"):
        # one call: the cloud stays in HBM between the passes (mrgfe_prefilter)
        ctx = ctx or default_context()
        q = _prefilter_params(p)
        out, m = np.empty((max(len(c), 1), 4), dtype=np.float32), C.c_size_t(0)
        check(lib().mrgfe_prefilter(ctx._h, C.byref(q), c.ctypes.data_as(_fp), len(c), 16, out.ctypes.data_as(_fp), C.byref(m)))
        return out[: m.value].copy()
    if p["enable_distance_filter"]:
        c = distance_filter(c, p["distance_near_thresh"], p["distance_far_thresh"], ctx)
    if p["downsample_method"] == "VOXELGRID":
        vg = VoxelGrid(ctx)
        vg.setLeafSize(p["downsample_resolution"])
        vg.setMinimumPointsNumberPerVoxel(p["downsample_min_points_per_voxel"])
        vg.setInputCloud(c)
        c = vg.filter()
    elif p["downsample_method"] == "APPROX_VOXELGRID":
        av = ApproximateVoxelGrid(ctx)
        av.setLeafSize(p["downsample_resolution"])
        av.setInputCloud(c)
        c = av.filter()
    elif p["downsample_method"] != "NONE":
        raise ValueError(f"unknown downsample_method {p['downsample_method']!r}")
    if p["outlier_removal_method"] == "RADIUS":
        ro = RadiusOutlierRemoval(ctx)
        ro.setRadiusSearch(p["radius_radius"])
        ro.setMinNeighborsInRadius(p["radius_min_neighbors"])
        ro.setInputCloud(c)
        c = ro.filter()
    elif p["outlier_removal_method"] == "STATISTICAL":
        so = StatisticalOutlierRemoval(ctx)
        so.setMeanK(p["statistical_mean_k"])
        so.setStddevMulThresh(p["statistical_stddev"])
        so.setInputCloud(c)
        c = so.filter()
    return c
