"""PrefilteringComponent mirror (mrg_slam_amd/prefiltering.py = apps/prefiltering_component.cpp:114-292): the IMU queue and early returns on the
CPU with the oracle's point operations, and the same call sequence through the HIP operations against the oracle's, bit for bit."""
import numpy as np
import pytest


def _scan(n=6000, seed=3):
    rng = np.random.default_rng(seed)
    c = np.zeros((n, 4), dtype=np.float32)
    c[:, :3] = rng.normal(0, 8.0, (n, 3)) * [1.0, 1.0, 0.15]
    c[:, 3] = rng.uniform(0, 255, n)
    return c


def _base_link():
    from mrg_slam_amd import synth

    return synth.make_pose([0.3, -0.1, 0.45], synth.rot_xyz(0.01, -0.02, 1.2)).astype(np.float32)


def test_imu_queue_and_early_returns_follow_the_reference():
    from mrg_slam_amd.prefiltering import OracleOps, PrefilteringComponent
    from oracle import oracle as orc

    used = []

    class Ops(OracleOps):
        def deskew(self, cloud, ang_v, scan_period):
            used.append(np.asarray(ang_v).copy())
            return super().deskew(cloud, ang_v, scan_period)

    c = PrefilteringComponent({"enable_deskewing": True}, ops=Ops(orc))
    scan = _scan(500)
    assert c.cloud_callback(np.zeros((0, 4), np.float32)) is None  # empty input: return (:121-123)
    out = c.cloud_callback(scan, stamp=1.0)  # no IMU message yet: the cloud is not deskewed (:234-236)
    assert used == [] and out is not None
    for k in range(6):
        c.imu_callback(0.5 + 0.2 * k, [0.1 * k, 0.0, 0.3])  # stamps 0.5, 0.7, 0.9, 1.1, 1.3, 1.5
    c.cloud_callback(scan, stamp=1.0)  # first message newer than the scan: stamp 1.1 (k = 3); the three before it leave the queue
    assert used[-1][0] == pytest.approx(0.3) and [s for s, _ in c.imu_queue] == pytest.approx([1.1, 1.3, 1.5])
    c.cloud_callback(scan, stamp=9.0)  # none newer: the LAST message is used and the queue is emptied (:262-270)
    assert used[-1][0] == pytest.approx(0.5) and c.imu_queue == []
    # enable_deskewing off: no subscription, nothing queued (:61-64)
    d = PrefilteringComponent({"enable_deskewing": False}, ops=Ops(orc))
    d.imu_callback(0.0, [1, 1, 1])
    assert d.imu_queue == []
    # no transform into base_link: warn and return early (:133-138)

    def no_tf(target, source):
        raise RuntimeError("no transform")

    assert PrefilteringComponent(ops=Ops(orc), lookup_transform=no_tf).cloud_callback(scan, frame_id="velodyne") is None
    # round 4: APPROX_VOXELGRID is served (it stayed on the CPU in rounds 1 - 3); only an unknown name is refused
    PrefilteringComponent({"downsample_method": "APPROX_VOXELGRID"}, ops=Ops(orc))
    with pytest.raises(ValueError):
        PrefilteringComponent({"downsample_method": "OCTREE"}, ops=Ops(orc))


def test_oracle_chain_is_the_composition_of_its_steps():
    from mrg_slam_amd.prefiltering import OracleOps, PrefilteringComponent
    from oracle import oracle as orc

    scan, T = _scan(), _base_link()
    c = PrefilteringComponent({"enable_deskewing": True, "outlier_removal_method": "STATISTICAL"}, ops=OracleOps(orc), lookup_transform=lambda a, b: T)
    c.imu_callback(5.0, [0.2, -0.1, 0.7])
    out = c.cloud_callback(scan, stamp=1.0, frame_id="velodyne")
    e = orc.transform_points(T, orc.deskew(scan, [0.2, -0.1, 0.7], 0.1))
    e = orc.statistical_outlier(orc.voxelgrid(orc.distance_filter(e, 0.1, 35.0), 0.1, 1)[0], 30, 1.2)[0]
    np.testing.assert_array_equal(out, e)


@pytest.mark.gpu
@pytest.mark.parametrize("outlier", ["RADIUS", "STATISTICAL", "NONE"])
@pytest.mark.parametrize("deskew", [False, True])
def test_hip_component_equals_the_oracle_component(street_pair_vlp16, outlier, deskew):
    """The same callback sequence (IMU messages, three scans, transform into base_link) through the HIP operations and the oracle's: the
    published clouds are identical."""
    from mrg_slam_amd.prefiltering import HipOps, OracleOps, PrefilteringComponent
    from oracle import oracle as orc

    T = _base_link()
    scans = [street_pair_vlp16[0], street_pair_vlp16[1], _scan(9000, 8)]
    scans[1] = scans[1].copy()
    scans[1][::97, 1] = np.nan  # fromROSMsg keeps NaN returns: they pass the transform untouched and the distance filter drops them
    params = {"enable_deskewing": deskew, "outlier_removal_method": outlier, "downsample_resolution": 0.2}
    outs = []
    for ops in (HipOps(), OracleOps(orc)):
        c = PrefilteringComponent(params, ops=ops, lookup_transform=lambda a, b: T)
        got = []
        for k, s in enumerate(scans):
            c.imu_callback(0.05 + 0.1 * k, [0.3 * (k + 1), -0.2, 0.9])
            c.imu_callback(0.15 + 0.1 * k, [0.1, 0.4 * k, -0.6])
            got.append(c.cloud_callback(s, stamp=0.1 * (k + 1), frame_id="velodyne"))
        outs.append(got)
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
        assert len(a) > 100


@pytest.mark.gpu
def test_transform_cloud_matches_oracle():
    from mrg_slam_amd import transform_cloud
    from oracle import oracle as orc

    scan, T = _scan(20000, 5), _base_link()
    scan[::41, 0] = np.inf
    out = transform_cloud(scan, T)
    fin = np.isfinite(scan[:, :3]).all(1)
    np.testing.assert_array_equal(out[fin], orc.transform_points(T, scan[fin]))
    np.testing.assert_array_equal(out[~fin], scan[~fin])
    assert transform_cloud(scan[:0], T).shape == (0, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["NDT", "SMALL_GICP"])
def test_prefiltering_into_odometry_hip_chain_equals_oracle_chain(method):
    """The two components composed as the reference composes them (the prefiltering component's output feeds
    ScanMatchingOdometryComponent::matching): a short VLP-16 drive with IMU messages and a base_link offset, the HIP chain against the
    oracle chain — identical filtered clouds, odometry poses within the bar, the same keyframe switches."""
    from mrg_slam_amd import NdtHip, PrefilteringComponent, ScanMatchingOdometry, SmallGicpHip, synth
    from mrg_slam_amd.prefiltering import HipOps, OracleOps
    from oracle import oracle as orc

    scene = synth.street_scene()
    frames = 7
    poses = [synth.make_pose([0.8 * k, 0.05 * k, 0.0], synth.rot_xyz(0.0, 0.0, np.deg2rad(2.0 * k))) for k in range(frames)]
    T_bl = synth.make_pose([0.2, 0.0, 0.3], synth.rot_xyz(0.0, 0.0, 0.0)).astype(np.float32)  # lidar -> base_link
    raw = [synth.synth_lidar(scene, poses[k], "VLP16", seed=4000 + k) for k in range(frames)]
    deltas = [np.eye(4)] + [synth.warm_guess(np.linalg.inv(poses[k - 1]) @ poses[k], 50 + k) for k in range(1, frames)]
    params = {"enable_deskewing": True, "scan_period": 0.1}
    odo_params = {"keyframe_delta_translation": 1.5}

    def run(ops, reg):
        pre = PrefilteringComponent(params, ops=ops, lookup_transform=lambda a, b: T_bl)
        odo = ScanMatchingOdometry(reg, odo_params)
        clouds, odoms, kfs = [], [], []
        for k in range(frames):
            pre.imu_callback(0.1 * k + 0.05, [0.0, 0.0, 0.02])  # a slow yaw rate
            f = pre.cloud_callback(raw[k], stamp=0.1 * k, frame_id="velodyne")
            clouds.append(f)
            odoms.append(np.array(odo.matching(0.1 * k, f, deltas[k]), dtype=np.float64))
            kfs.append(float(odo.keyframe_stamp))
        return clouds, odoms, kfs

    if method == "NDT":
        g, o = NdtHip(resolution=1.0, transformation_epsilon=0.1), orc.Ndt(resolution=1.0, transformation_epsilon=0.1, num_threads=8)
    else:
        g, o = SmallGicpHip(transformation_epsilon=0.1), orc.SmallGicp(transformation_epsilon=0.1, num_threads=8)
    gc, go, gk = run(HipOps(), g)
    oc, oo, ok = run(OracleOps(orc), o)
    for a, b in zip(gc, oc):
        np.testing.assert_array_equal(a, b)
    assert gk == ok and len(set(gk)) >= 2  # the keyframe was switched at least once, at the same frames
    for a, b in zip(go, oo):
        assert np.linalg.norm(a[:3, 3] - b[:3, 3]) <= 1e-4 and synth.rotation_angle(a, b) <= 1e-4


def test_odometry_keyframe_update_takes_the_source_over_when_the_registration_can():
    """ScanMatchingOdometry._new_keyframe (scan_matching_odometry_component.cpp:326-339): a registration with ``sourceBecomesTarget`` is told to keep what
    it made for the scan it has just aligned; one without it (the CPU oracle), or a caller that routes the clouds itself, gets ``setInputTarget`` as before."""
    import numpy as np

    from mrg_slam_amd.odometry import ScanMatchingOdometry

    class Reg:
        def __init__(self, can_promote):
            self.calls = []
            if can_promote:
                self.sourceBecomesTarget = lambda: self.calls.append("promote") or 0

        def setInputTarget(self, c):
            self.calls.append("target")

        def setInputSource(self, c):
            self.calls.append("source")

        def align(self, guess=None, want_aligned=False):
            self.calls.append("align")

        def hasConverged(self):
            return True

        def getFinalTransformation(self):
            T = np.eye(4, dtype=np.float32)
            T[0, 3] = 1.5  # farther than keyframe_delta_translation: every frame becomes a keyframe
            return T

    cloud = np.zeros((10, 4), dtype=np.float32)
    for can, routed, want in ((True, False, "promote"), (False, False, "target"), (True, True, "routed")):
        reg = Reg(can)
        routed_calls = []
        kw = {"set_target": lambda c: routed_calls.append("routed"), "set_source": lambda c: None} if routed else {}
        odo = ScanMatchingOdometry(reg, **kw)
        odo.matching(0.0, cloud)
        odo.matching(0.1, cloud)
        odo.matching(0.2, cloud)
        assert odo.keyframes == 3
        updates = (routed_calls if routed else reg.calls)
        # (the first keyframe is always handed over: there is no source yet)
        assert updates.count(want) == (2 if want == "promote" else 3), (can, routed, reg.calls, routed_calls)
        if want == "promote":
            assert reg.calls.count("target") == 1
