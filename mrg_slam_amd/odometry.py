"""Host mirror of ``ScanMatchingOdometryComponent::matching`` (/root/reference/apps/scan_matching_odometry_component.cpp:195-350): the
caller of the hot path in the odometry component — first frame becomes the keyframe (``setInputTarget``), every later frame is
``setInputSource`` + ``align(prev_trans * msf_delta)`` against it, the keyframe is replaced when the motion since it exceeds
``keyframe_delta_translation`` / ``keyframe_delta_angle`` / ``keyframe_delta_time`` (:326-339), optional transform thresholding with
``max_consecutive_rejections`` (:278-318).  ``registration`` is anything with the ``pcl::Registration`` call surface the reference uses
(``setInputTarget`` / ``setInputSource`` / ``align(guess)`` / ``hasConverged`` / ``getFinalTransformation``): a HIP registration
(``mrg_slam_amd.registration``) or the CPU oracle's classes — the parity tests run the SAME loop over both.

Arithmetic as in the reference: poses are float 4 x 4 matrices (``Eigen::Matrix4f``), products in float; the rotation angle is
``acos(Quaternionf(R).w())`` (Eigen's matrix -> quaternion conversion, restated in ``quat_w``)."""
from __future__ import annotations

import numpy as np

DEFAULTS = {  # config/mrg_slam.yaml:77-86 (the component's own declare_parameter defaults differ: 0.25 / 0.15 / 1.0, :103-105)
    "keyframe_delta_translation": 1.0,
    "keyframe_delta_angle": 0.5236,
    "keyframe_delta_time": 10000.0,
    "enable_transform_thresholding": False,
    "max_acceptable_translation": 1.0,
    "max_acceptable_angle": 1.0,
    "max_consecutive_rejections": 5,
}


def quat_w(R) -> np.float32:
    """w of Eigen::Quaternionf(R) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl for a 3 x 3 matrix), in float."""
    R = np.asarray(R, dtype=np.float32)
    t = np.float32(R[0, 0] + R[1, 1] + R[2, 2])
    if t > np.float32(0):
        return np.float32(0.5) * np.sqrt(t + np.float32(1.0), dtype=np.float32)
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    tt = np.sqrt(np.float32(R[i, i] - R[j, j] - R[k, k] + np.float32(1.0)), dtype=np.float32)
    return np.float32((R[k, j] - R[j, k]) * (np.float32(0.5) / tt))


def rotation_angle(T) -> float:
    """``std::acos(Eigen::Quaternionf(T.block<3,3>(0,0)).w())`` (:281,327), promoted to double like the reference's ``double da``."""
    w = float(quat_w(np.asarray(T)[:3, :3]))
    return float(np.arccos(np.float32(min(1.0, max(-1.0, w)))))


class ScanMatchingOdometry:
    """State and decisions of ``ScanMatchingOdometryComponent::matching``; ``downsample`` is the component's own downsample step
    (default NONE, config/mrg_slam.yaml:95: the cloud passes through)."""

    def __init__(self, registration, params: dict | None = None, downsample=None, set_target=None, set_source=None):
        """``set_target`` / ``set_source`` override how a cloud is handed to the registration (e.g. device-resident clouds through
        ``setInputTargetDevice``); by default ``registration.setInputTarget(cloud)`` / ``setInputSource(cloud)``."""
        self.reg = registration
        self.p = dict(DEFAULTS)
        self.p.update(params or {})
        self._downsample = downsample or (lambda c: c)
        self._promote = set_target is None and set_source is None and hasattr(registration, "sourceBecomesTarget") and self.p.get("promote_source_to_keyframe", True)
        self._set_target = set_target or (lambda c: registration.setInputTarget(c))
        self._set_source = set_source or (lambda c: registration.setInputSource(c))
        self.keyframe_cloud = None
        self.keyframe_pose = np.eye(4, dtype=np.float32)
        self.keyframe_stamp = 0.0
        self.prev_time = None
        self.prev_trans = np.eye(4, dtype=np.float32)
        self.consecutive_rejections = 0
        self.keyframes = 0  # how many times a cloud became the keyframe
        self.last_converged = True

    def matching(self, stamp: float, cloud, msf_delta=None) -> np.ndarray:
        """Returns the odometry pose (float 4 x 4) of this frame."""
        if self.keyframe_cloud is None:  # :197-205
            self.prev_time = None
            self.prev_trans = np.eye(4, dtype=np.float32)
            self.keyframe_pose = np.eye(4, dtype=np.float32)
            self.keyframe_stamp = stamp
            self.keyframe_cloud = self._downsample(cloud)
            self._set_target(self.keyframe_cloud)
            self.keyframes += 1
            return np.eye(4, dtype=np.float32)
        filtered = self._downsample(cloud)
        self._set_source(filtered)
        delta = np.eye(4, dtype=np.float32) if msf_delta is None else np.asarray(msf_delta, dtype=np.float32)
        self.reg.align((self.prev_trans @ delta).astype(np.float32))  # :265-266
        self.last_converged = bool(self.reg.hasConverged())
        if not self.last_converged:  # :270-273
            return (self.keyframe_pose @ self.prev_trans).astype(np.float32)
        trans = np.asarray(self.reg.getFinalTransformation(), dtype=np.float32)
        odom = (self.keyframe_pose @ trans).astype(np.float32)
        if self.p["enable_transform_thresholding"]:  # :278-318
            d = (np.linalg.inv(self.prev_trans).astype(np.float32) @ trans).astype(np.float32)
            dx = float(np.linalg.norm(d[:3, 3].astype(np.float32)))
            da = rotation_angle(d)
            if dx > self.p["max_acceptable_translation"] or da > self.p["max_acceptable_angle"]:
                self.consecutive_rejections += 1
                if self.consecutive_rejections >= self.p["max_consecutive_rejections"]:
                    self._new_keyframe(filtered, odom, stamp)
                    self.consecutive_rejections = 0
                    return self.keyframe_pose
                self.prev_time = stamp
                return (self.keyframe_pose @ self.prev_trans).astype(np.float32)
            self.consecutive_rejections = 0
        self.prev_time = stamp
        self.prev_trans = trans
        delta_translation = float(np.linalg.norm(trans[:3, 3]))
        delta_angle = rotation_angle(trans)
        delta_time = stamp - self.keyframe_stamp
        if (delta_translation > self.p["keyframe_delta_translation"] or delta_angle > self.p["keyframe_delta_angle"]
                or delta_time > self.p["keyframe_delta_time"]):  # :326-339
            self._new_keyframe(filtered, odom, stamp)
        return odom

    def _new_keyframe(self, filtered, odom, stamp):
        # (`filtered` is the cloud that has just been aligned as the source, :208 -> :333: a HIP registration takes it over with what it computed for it)
        self.keyframe_cloud = filtered
        if self._promote:
            self.reg.sourceBecomesTarget()
        else:
            self._set_target(filtered)
        self.keyframe_pose = odom
        self.keyframe_stamp = stamp
        self.prev_time = stamp
        self.prev_trans = np.eye(4, dtype=np.float32)
        self.keyframes += 1
