"""Host-side mirror of the per-point passes around the scan-matching path, bound to
libmrgfe.so:

* :class:`MapCloudGenerator` — mrg_slam::MapCloudGenerator (/root/reference/src/mrg_slam/map_cloud_generator.cpp:14-86,
  called from apps/mrg_slam_component.cpp:727,781,1097) with its pcl::ApproximateMeanVoxelGrid pass;
* :func:`remove_points_near` — the other-robot point removal of apps/mrg_slam_component.cpp:396-429;
* :func:`deskew` — PrefilteringComponent::deskewing (apps/prefiltering_component.cpp:231-292).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import Context, MrgfeError, check, default_context, lib
from .filters import _cloud

_fp = C.POINTER(C.c_float)
ERR_EMPTY = -4  # MRGFE_ERR_EMPTY (include/mrgfe.h)


@dataclass
class KeyFrameSnapshot:
    """The two members of mrg_slam::KeyFrameSnapshot the generator reads, plus the first_keyframe flag."""

    pose: np.ndarray  # 4 x 4, Eigen::Isometry3d
    cloud: np.ndarray  # N x 4 float32 (x, y, z, intensity)
    first_keyframe: bool = False


class MapCloudStore:
    """Keyframe clouds resident in HBM (mrgfe_map_store): the map is regenerated from all keyframes every time it is
    published, and between two calls only the poses change."""

    def __init__(self, ctx: Context | None = None):
        self._ctx = ctx or default_context()
        self._h = C.c_void_p()
        check(lib().mrgfe_map_store_create(self._ctx._h, C.byref(self._h)))

    def _release_out(self):
        if getattr(self, "_out", None) is not None and getattr(self, "_out_pinned", False):
            lib().mrgfe_unpin_host_buffer(self._ctx._h, self._out.ctypes.data_as(C.c_void_p))
        self._out, self._out_pinned = None, False

    def __del__(self):
        try:
            self._release_out()
            if getattr(self, "_h", None):
                lib().mrgfe_map_store_destroy(self._h)
                self._h = None
        except Exception:  # noqa: BLE001
            pass

    def add(self, key: int, cloud) -> None:
        c = _cloud(cloud)
        check(lib().mrgfe_map_store_add(self._h, key, c.ctypes.data_as(_fp), len(c), 16))

    def has(self, key: int):
        n = C.c_size_t(0)
        return n.value if lib().mrgfe_map_store_has(self._h, key, C.byref(n)) else None

    def bytes(self) -> int:
        return int(lib().mrgfe_map_store_bytes(self._h))

    def fitness(self, key1: int, key2: int, relpose, max_range: float = float("inf")) -> float:
        """InformationMatrixCalculator::calc_fitness_score(cloud1 = keyframe key1, cloud2 = keyframe key2, relpose) on the resident clouds."""
        T = np.ascontiguousarray(np.asarray(relpose, dtype=np.float64).T)
        out = C.c_double(0)
        check(lib().mrgfe_map_store_fitness(self._h, int(key1), int(key2), T.ctypes.data_as(C.POINTER(C.c_double)), max_range, C.byref(out)))
        return out.value

    def generate(self, keys, poses, first_keyframe=None, resolution: float = 0.1, min_points_per_voxel: int = 1, distance_far_thresh: float = 10000.0,
                 skip_first_cloud: bool = False, copy: bool = True):
        """MapCloudGenerator.generate over the stored keyframes ``keys`` with their current ``poses`` (4 x 4 each).
        The download lands in a buffer the store keeps between calls (a fresh 100 MB array per call costs more in page faults than the
        kernels and the copy together); ``copy=False`` returns a view of it that the next call overwrites."""
        K = len(keys)
        ks = np.ascontiguousarray(np.asarray(keys, dtype=np.uint64))
        P = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.float64).T.reshape(16) for p in poses])) if K else np.zeros((0, 16))
        first = np.ascontiguousarray(np.asarray(first_keyframe if first_keyframe is not None else np.zeros(K), dtype=np.uint8))
        cap = max(sum(self.has(int(k)) or 0 for k in keys), 1)
        if getattr(self, "_out", None) is None or len(self._out) < cap:
            self._release_out()
            self._out = np.empty((cap + cap // 4, 4), dtype=np.float32)
            self._out[:] = 0  # (touch the pages once, here)
            try:  # page-locked: the download is direct DMA
                check(lib().mrgfe_pin_host_buffer(self._ctx._h, self._out.ctypes.data_as(C.c_void_p), self._out.nbytes))
                self._out_pinned = True
            except MrgfeError:
                self._out_pinned = False
        out = self._out
        m = C.c_size_t(0)
        try:
            check(lib().mrgfe_map_store_generate(self._h, K, ks.ctypes.data_as(C.POINTER(C.c_uint64)), P.ctypes.data_as(C.POINTER(C.c_double)),
                                                 first.ctypes.data_as(C.POINTER(C.c_uint8)), float(resolution), int(min_points_per_voxel), float(distance_far_thresh),
                                                 int(bool(skip_first_cloud)), out.ctypes.data_as(_fp), cap, C.byref(m)))
        except MrgfeError as e:
            if e.status == ERR_EMPTY:
                return None
            raise
        return out[: m.value].copy() if copy else out[: m.value]


class MapCloudGenerator:
    def __init__(self, ctx: Context | None = None):
        self._ctx = ctx or default_context()

    def generate(self, keyframes, resolution: float, min_points_per_voxel: int = 1, distance_far_thresh: float = 10000.0, skip_first_cloud: bool = False):
        """Returns the map cloud (voxels in ascending voxel index; the reference's order is its hash map's), or None where
        the reference returns nullptr (no keyframes, or an empty cloud from more than one keyframe)."""
        kfs = list(keyframes)
        K = len(kfs)
        clouds = [_cloud(k.cloud) for k in kfs]
        ptrs = (_fp * max(K, 1))(*[c.ctypes.data_as(_fp) for c in clouds])
        ns = (C.c_size_t * max(K, 1))(*[len(c) for c in clouds])
        poses = np.ascontiguousarray(np.stack([np.asarray(k.pose, dtype=np.float64).T.reshape(16) for k in kfs])) if K else np.zeros((0, 16))
        first = np.ascontiguousarray(np.array([1 if k.first_keyframe else 0 for k in kfs], dtype=np.uint8))
        cap = max(int(sum(len(c) for c in clouds)), 1)
        out = np.empty((cap, 4), dtype=np.float32)
        m = C.c_size_t(0)
        try:
            check(lib().mrgfe_map_cloud_generate(self._ctx._h, K, ptrs, ns, 16, poses.ctypes.data_as(C.POINTER(C.c_double)), first.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                 float(resolution), int(min_points_per_voxel), float(distance_far_thresh), int(bool(skip_first_cloud)),
                                                 out.ctypes.data_as(_fp), cap, C.byref(m)))
        except MrgfeError as e:
            if e.status == ERR_EMPTY:
                return None
            raise
        return out[: m.value].copy()


def remove_points_near(cloud, centres_sensor, radius: float, ctx: Context | None = None):
    """Returns (kept, removed): every point closer than ``radius`` to one of ``centres_sensor`` (K x 3, sensor frame) goes to
    ``removed``.  robot_radius_sqr is float(radius * radius), as at mrg_slam_component.cpp:405-406."""
    ctx = ctx or default_context()
    c = _cloud(cloud)
    ctr = np.ascontiguousarray(np.asarray(centres_sensor, dtype=np.float32).reshape(-1, 3))
    kept, removed = np.empty_like(c), np.empty_like(c)
    nk, nr = C.c_size_t(0), C.c_size_t(0)
    check(lib().mrgfe_remove_points_near(ctx._h, c.ctypes.data_as(_fp), len(c), 16, ctr.ctypes.data_as(_fp), len(ctr), float(np.float32(float(radius) * float(radius))),
                                         kept.ctypes.data_as(_fp), C.byref(nk), removed.ctypes.data_as(_fp), C.byref(nr)))
    return kept[: nk.value].copy(), removed[: nr.value].copy()


def deskew(cloud, angular_velocity, scan_period: float = 0.1, ctx: Context | None = None) -> np.ndarray:
    ctx = ctx or default_context()
    c = _cloud(cloud)
    out = np.empty_like(c)
    av = np.ascontiguousarray(np.asarray(angular_velocity, dtype=np.float32).reshape(3))
    check(lib().mrgfe_deskew(ctx._h, c.ctypes.data_as(_fp), len(c), 16, av.ctypes.data_as(_fp), float(scan_period), out.ctypes.data_as(_fp)))
    return out


def transform_cloud(cloud, T, ctx: Context | None = None) -> np.ndarray:
    """pcl::transformPointCloud(cloud, out, Matrix4f(T)) (the base_link transform of PrefilteringComponent::cloud_callback,
    apps/prefiltering_component.cpp:141): float arithmetic, non-finite points unchanged, intensity copied."""
    ctx = ctx or default_context()
    c = _cloud(cloud)
    out = np.empty_like(c)
    Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float32).T)
    check(lib().mrgfe_transform_cloud(ctx._h, c.ctypes.data_as(_fp), len(c), 16, Tc.ctypes.data_as(_fp), out.ctypes.data_as(_fp)))
    return out
