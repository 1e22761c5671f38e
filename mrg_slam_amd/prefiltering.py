"""Host mirror of ``PrefilteringComponent`` (/root/reference/apps/prefiltering_component.cpp:114-292): the caller of the prefilter rows of
the hot path — ``cloud_callback`` = ``deskewing`` (:231-292, with the IMU queue of ``imu_callback`` :114) -> transform into
``base_link_frame`` (:126-146, ``pcl_ros::transformPointCloud``) -> ``distance_filter`` (:207-229) -> ``downsample`` (:151-174) ->
``outlier_removal`` (:176-205).  ``ops`` supplies the point operations: by default the HIP path (``mrg_slam_amd``), for the parity tests the
CPU oracle — the SAME control flow runs over both.

What the ROS side does around it (``pcl::fromROSMsg`` / ``toROSMsg``, tf lookup, publishers) stays with the caller: ``cloud_callback``
takes the cloud as an ``[n, 4]`` float array (``mrg_slam_amd.io.ingest_pointcloud2`` makes one from a PointCloud2 payload on the GPU) and
``lookup_transform(base_link_frame, frame_id)`` stands for ``tf_buffer_->lookupTransform`` (returns a 4 x 4 matrix, or raises)."""
from __future__ import annotations

import numpy as np

DEFAULTS = {  # config/mrg_slam.yaml:43-68 (the component's declare_parameter defaults differ: STATISTICAL 20 / 1.0, radius 0.8, near 1.0; :98-111)
    "base_link_frame": "base_link",
    "downsample_method": "VOXELGRID",
    "downsample_resolution": 0.1,
    "downsample_min_points_per_voxel": 1,
    "outlier_removal_method": "RADIUS",
    "statistical_mean_k": 30,
    "statistical_stddev": 1.2,
    "radius_radius": 0.5,
    "radius_min_neighbors": 2,
    "enable_distance_filter": True,
    "distance_near_thresh": 0.1,
    "distance_far_thresh": 35.0,
    "enable_deskewing": False,
    "scan_period": 0.1,
}


class HipOps:
    """The point operations on the GPU (``libmrgfe``): the three filters go through one fused call (one upload, one download)."""

    def __init__(self, ctx=None):
        self.ctx = ctx

    def deskew(self, cloud, ang_v, scan_period):
        from .map_cloud import deskew

        return deskew(cloud, ang_v, scan_period, ctx=self.ctx)

    def transform(self, cloud, T):
        from .map_cloud import transform_cloud

        return transform_cloud(cloud, T, ctx=self.ctx)

    def filters(self, cloud, p):
        from .filters import prefilter

        return prefilter(cloud, p, ctx=self.ctx)


class OracleOps:
    """The same operations by the CPU oracle, one after the other as the reference calls them (tests only)."""

    def __init__(self, orc):
        self.orc = orc

    def deskew(self, cloud, ang_v, scan_period):
        return self.orc.deskew(cloud, ang_v, scan_period)

    def transform(self, cloud, T):
        return self.orc.transform_points(np.asarray(T, dtype=np.float32), cloud)

    def filters(self, cloud, p):
        c = cloud
        if p["enable_distance_filter"]:
            c = self.orc.distance_filter(c, p["distance_near_thresh"], p["distance_far_thresh"])
        if p["downsample_method"] == "VOXELGRID":
            c = self.orc.voxelgrid(c, p["downsample_resolution"], p["downsample_min_points_per_voxel"])[0]
        elif p["downsample_method"] == "APPROX_VOXELGRID":
            c = self.orc.approx_voxelgrid(c, p["downsample_resolution"])
        if p["outlier_removal_method"] == "RADIUS":
            c = self.orc.radius_outlier(c, p["radius_radius"], p["radius_min_neighbors"])[0]
        elif p["outlier_removal_method"] == "STATISTICAL":
            c = self.orc.statistical_outlier(c, p["statistical_mean_k"], p["statistical_stddev"])[0]
        return c


class PrefilteringComponent:
    def __init__(self, params: dict | None = None, ops=None, lookup_transform=None):
        self.p = dict(DEFAULTS)
        self.p.update(params or {})
        if self.p["downsample_method"] not in ("VOXELGRID", "APPROX_VOXELGRID", "NONE"):
            raise ValueError(f"unknown downsample_method {self.p['downsample_method']!r}")
        if self.p["outlier_removal_method"] not in ("RADIUS", "STATISTICAL", "NONE"):
            raise ValueError(f"unknown outlier_removal_method {self.p['outlier_removal_method']!r}")
        self.ops = ops or HipOps()
        self.lookup_transform = lookup_transform
        self.imu_queue: list[tuple[float, np.ndarray]] = []  # (stamp, angular velocity): imu_queue_

    def imu_callback(self, stamp: float, angular_velocity) -> None:
        """:114 — the subscription exists only with enable_deskewing (:61-64)."""
        if self.p["enable_deskewing"]:
            self.imu_queue.append((float(stamp), np.asarray(angular_velocity, dtype=np.float32).reshape(3)))

    def deskewing(self, cloud: np.ndarray, stamp: float) -> np.ndarray:
        """:231-292.  The IMU message used is the first one newer than the scan, or the last one of the queue when none is; everything
        before it leaves the queue (:262-270)."""
        if not self.imu_queue:
            return cloud
        loc = 0
        ang_v = self.imu_queue[0][1]
        while loc < len(self.imu_queue):
            ang_v = self.imu_queue[loc][1]
            if self.imu_queue[loc][0] > stamp:
                break
            loc += 1
        del self.imu_queue[:loc]
        return self.ops.deskew(cloud, ang_v, self.p["scan_period"])

    def cloud_callback(self, cloud, stamp: float = 0.0, frame_id: str = ""):
        """:116-149.  Returns the filtered cloud (what ``points_pub_`` publishes), or None where the reference returns early (empty input,
        no transform into base_link_frame)."""
        src = np.ascontiguousarray(np.asarray(cloud, dtype=np.float32).reshape(-1, 4))
        if len(src) == 0:
            return None
        src = self.deskewing(src, stamp)
        if self.p["base_link_frame"] and self.lookup_transform is not None:
            try:
                T = self.lookup_transform(self.p["base_link_frame"], frame_id)
            except Exception:  # noqa: BLE001 - tf2::TransformException: warn and return early (:133-138)
                return None
            src = self.ops.transform(src, T)
        return self.ops.filters(src, self.p)
