"""GPU: prefilter chain and exact nearest-neighbour queries against the CPU oracle (bit-exact: these are index /
membership computations plus float sums in a defined order)."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


def test_distance_filter_exact():
    from mrg_slam_amd import distance_filter
    from oracle import oracle as orc

    c = small_cloud(20000, 3, extent=(45, 45, 3))
    c[:5, :3] = 0.01
    np.testing.assert_array_equal(distance_filter(c, 0.1, 35.0), orc.distance_filter(c, 0.1, 35.0))
    assert len(distance_filter(np.zeros((0, 4), np.float32))) == 0
    assert len(distance_filter(c, 100.0, 200.0)) == 0


@pytest.mark.parametrize("leaf,min_pts,n", [(0.1, 1, 30000), (0.5, 1, 30000), (0.5, 3, 8000), (2.0, 1, 8000), (0.05, 1, 3000)])
def test_voxelgrid_exact(leaf, min_pts, n):
    from mrg_slam_amd import VoxelGrid
    from oracle import oracle as orc

    c = small_cloud(n, 5)
    vg = VoxelGrid()
    vg.setLeafSize(leaf, leaf, leaf)
    vg.setMinimumPointsNumberPerVoxel(min_pts)
    vg.setInputCloud(c)
    out = vg.filter()
    exp, status = orc.voxelgrid(c, leaf, min_pts, orc.ORDER_STABLE)
    assert status == 0 and not vg.overflow
    np.testing.assert_array_equal(out, exp)
    # PCL's own (std::sort) in-voxel order gives the same voxels and centroids to float rounding
    exp2, _ = orc.voxelgrid(c, leaf, min_pts, orc.ORDER_STD_SORT)
    assert exp2.shape == out.shape
    np.testing.assert_allclose(out, exp2, rtol=0, atol=2e-5)


def test_voxelgrid_edge_cases():
    from mrg_slam_amd import VoxelGrid
    from oracle import oracle as orc

    vg = VoxelGrid()
    vg.setLeafSize(0.1)
    vg.setInputCloud(np.zeros((0, 4), np.float32))
    assert len(vg.filter()) == 0
    c = small_cloud(500)
    c[0, :3] = [3e5, -3e5, 3e5]  # index overflow: PCL warns and passes the cloud through
    vg.setInputCloud(c)
    out = vg.filter()
    assert vg.overflow
    np.testing.assert_array_equal(out, c)
    c = small_cloud(500)
    c[7, 1] = np.nan  # non-finite points are dropped
    vg.setInputCloud(c)
    exp, _ = orc.voxelgrid(c, 0.1, 1)
    np.testing.assert_array_equal(vg.filter(), exp)


@pytest.mark.parametrize("radius,min_nb", [(0.5, 2), (0.3, 1), (1.0, 8)])
def test_radius_outlier_exact(radius, min_nb):
    from mrg_slam_amd import RadiusOutlierRemoval
    from oracle import oracle as orc

    c = small_cloud(20000, 7)
    c[:20, :3] += 100.0
    ro = RadiusOutlierRemoval()
    ro.setRadiusSearch(radius)
    ro.setMinNeighborsInRadius(min_nb)
    ro.setInputCloud(c)
    out = ro.filter()
    exp, keep = orc.radius_outlier(c, radius, min_nb)
    assert 0 < len(exp) < len(c)
    np.testing.assert_array_equal(out, exp)
    # lattice with spacing exactly r: the compare is inclusive (sqdist <= r*r)
    g = np.zeros((27, 4), np.float32)
    g[:, :3] = np.stack(np.meshgrid(*[np.arange(3) * 0.5] * 3, indexing="ij"), -1).reshape(-1, 3)
    ro2 = RadiusOutlierRemoval()
    ro2.setRadiusSearch(0.5)
    ro2.setMinNeighborsInRadius(3)
    ro2.setInputCloud(g)
    np.testing.assert_array_equal(ro2.filter(), g)


def test_statistical_outlier_exact():
    from mrg_slam_amd import StatisticalOutlierRemoval
    from oracle import oracle as orc

    c = small_cloud(6000, 13)
    c[:10, :3] += 60.0
    for k, s in ((30, 1.2), (10, 0.5)):
        so = StatisticalOutlierRemoval()
        so.setMeanK(k)
        so.setStddevMulThresh(s)
        so.setInputCloud(c)
        exp, _ = orc.statistical_outlier(c, k, s)
        np.testing.assert_array_equal(so.filter(), exp)


def test_prefilter_chain_on_street_scan(street_pair_vlp16):
    from mrg_slam_amd import prefilter
    from oracle import oracle as orc

    tgt, _, _ = street_pair_vlp16
    exp = orc.distance_filter(tgt, 0.1, 35.0)
    exp, _ = orc.voxelgrid(exp, 0.1, 1)
    exp, _ = orc.radius_outlier(exp, 0.5, 2)
    np.testing.assert_array_equal(prefilter(tgt), exp)


def test_nearest_neighbour_and_fitness_exact():
    from mrg_slam_amd import NdtHip, calc_fitness_score, synth
    from oracle import oracle as orc

    t, q = small_cloud(20000, 11), small_cloud(3000, 12, extent=(30, 20, 5))
    q[:3, :3] += 500.0  # far outside the target's bounding box
    reg = NdtHip()
    reg.setInputTarget(t)
    idx, sqd = reg.nearestKSearch1(q)
    bi, bd = orc.nn1_brute(t, q)
    np.testing.assert_array_equal(sqd, bd)
    np.testing.assert_array_equal(idx, bi)
    rel = synth.make_pose([0.05, -0.02, 0.01], synth.rot_xyz(0.001, 0.002, -0.003))
    for max_range in (float("inf"), 0.01, 1e-9):
        got = calc_fitness_score(t, q, rel, max_range)
        exp = orc.calc_fitness_score(t, q, rel, max_range)
        assert got == pytest.approx(exp, rel=1e-12)
    assert calc_fitness_score(t, np.zeros((0, 4), np.float32), rel) == np.finfo(np.float64).max


def _adversarial_target(rng, kind):
    if kind == "clusters":  # dense blobs metres apart: most of the bounding box is empty, far queries cross many empty bricks
        centres = rng.uniform(-40, 40, (12, 3)) * [1, 1, 0.1]
        pts = np.concatenate([c + rng.normal(0, 0.15, (3000, 3)) for c in centres])
    elif kind == "plane_and_specks":  # one dense sheet plus isolated points far above it
        sheet = np.c_[rng.uniform(-30, 30, (40000, 2)), rng.normal(0, 0.01, 40000)]
        specks = rng.uniform(-30, 30, (40, 3)) + [0, 0, 45]
        pts = np.concatenate([sheet, specks])
    elif kind == "line":  # degenerate extent along two axes
        pts = np.c_[rng.uniform(-50, 50, 20000), np.zeros(20000), np.zeros(20000)]
    elif kind == "duplicates":  # every point four times: ties must resolve to the lowest index
        base = rng.uniform(-10, 10, (2500, 3))
        pts = np.concatenate([base, base, base, base])
    else:  # "tiny"
        pts = rng.uniform(-1, 1, (3, 3))
    out = np.zeros((len(pts), 4), np.float32)
    out[:, :3] = pts
    return out


@pytest.mark.parametrize("kind", ["clusters", "plane_and_specks", "line", "duplicates", "tiny"])
def test_nearest_neighbour_exact_on_adversarial_clouds(kind):
    """The occupancy-pyramid search is exhaustive: indices and squared distances equal the brute-force scan (ties to the
    lowest index) for queries next to, between, far from and outside the points, and the bounded fitness sums agree."""
    from mrg_slam_amd import NdtHip, calc_fitness_score
    from oracle import oracle as orc

    rng = np.random.default_rng({"clusters": 1, "plane_and_specks": 2, "line": 3, "duplicates": 4, "tiny": 5}[kind])
    t = _adversarial_target(rng, kind)
    lo, hi = t[:, :3].min(0), t[:, :3].max(0)
    q = np.zeros((4000, 4), np.float32)
    q[:1000, :3] = t[rng.integers(0, len(t), 1000), :3] + rng.normal(0, 0.02, (1000, 3))      # next to points
    q[1000:2500, :3] = rng.uniform(lo - 1, hi + 1, (1500, 3))                                  # anywhere in the box
    q[2500:3500, :3] = rng.uniform(lo - 60, hi + 60, (1000, 3))                                # mostly outside
    q[3500:3990, :3] = t[rng.integers(0, len(t), 490), :3]                                     # exactly on points
    q[3990:, :3] = [[1e4, 0, 0], [0, -1e4, 0], [0, 0, 1e4], [1e4, 1e4, 1e4], [-3e3, 2e3, 5e2], [np.nan, 0, 0], [0, np.inf, 0], [7, 7, 7], [0, 0, 0], [-0.0, 0.0, -0.0]]
    reg = NdtHip()
    reg.setInputTarget(t)
    idx, sqd = reg.nearestKSearch1(q)
    bi, bd = orc.nn1_brute(t, q)
    finite = np.isfinite(q[:, :3]).all(1)
    np.testing.assert_array_equal(sqd[finite], bd[finite])
    np.testing.assert_array_equal(idx[finite], bi[finite])
    assert (idx[~finite] == -1).all()
    qf = q[finite]
    for max_range in (float("inf"), 25.0, 1.0, 0.01):
        assert calc_fitness_score(t, qf, np.eye(4), max_range) == pytest.approx(orc.calc_fitness_score(t, qf, np.eye(4), max_range), rel=1e-12)


@pytest.mark.parametrize("kind,k", [("clusters", 20), ("plane_and_specks", 20), ("line", 7), ("duplicates", 30), ("tiny", 5)])
def test_knn_exact_on_adversarial_clouds(kind, k):
    """nearestKSearch(k) (GICP covariances, StatisticalOutlierRemoval): equal to a full sort of all squared distances by
    (distance, index), also for queries in empty regions (level climbing) and for clouds smaller than k."""
    from mrg_slam_amd import knn

    rng = np.random.default_rng({"clusters": 11, "plane_and_specks": 12, "line": 13, "duplicates": 14, "tiny": 15}[kind])
    t = _adversarial_target(rng, kind)
    lo, hi = t[:, :3].min(0), t[:, :3].max(0)
    q = np.zeros((600, 4), np.float32)
    q[:300, :3] = t[rng.integers(0, len(t), 300), :3]                # the cloud's own points (self k-NN)
    q[300:500, :3] = rng.uniform(lo - 1, hi + 1, (200, 3))
    q[500:, :3] = rng.uniform(lo - 40, hi + 40, (100, 3))
    idx, sqd = knn(t, q, k)
    tx, ty, tz = (t[:, a].astype(np.float32) for a in range(3))
    for i in range(len(q)):
        # the library's float expression: ((dx*dx + dy*dy) + dz*dz) in float32
        dx, dy, dz = tx - q[i, 0], ty - q[i, 1], tz - q[i, 2]
        d = (dx * dx + dy * dy) + dz * dz
        order = np.lexsort((np.arange(len(t)), d))[:k]
        m = len(order)
        np.testing.assert_array_equal(idx[i, :m], order.astype(np.int32), err_msg=f"query {i}")
        np.testing.assert_array_equal(sqd[i, :m], d[order])
        assert (idx[i, m:] == -1).all()


@pytest.mark.parametrize("kind,k", [("clusters", 20), ("plane_and_specks", 20), ("line", 7), ("duplicates", 30), ("tiny", 5), ("scan", 20), ("scan", 31), ("scan_nan", 10)])
def test_knn_of_a_clouds_own_points(street_pair_vlp16, kind, k):
    """The cloud's own points as queries (GICP covariances, StatisticalOutlierRemoval), isolated points included (they climb rings and levels):
    a sample of rows equals the full sort by (distance, index); non-finite points get empty rows."""
    from mrg_slam_amd import knn

    rng = np.random.default_rng(21 + k)
    if kind.startswith("scan"):
        t = street_pair_vlp16[0].copy()
        t[:40, :3] += rng.uniform(-300, 300, (40, 3)).astype(np.float32)  # specks far from everything
        if kind == "scan_nan":
            t[::13, 2] = np.nan
            t[7::101, 0] = np.inf
    else:
        t = _adversarial_target(rng, kind)
    idx, sqd = knn(t, t, k)
    fin = np.isfinite(t[:, :3]).all(1)
    assert (idx[~fin] == -1).all() and (sqd[~fin] == -1.0).all()
    ids = np.nonzero(fin)[0]
    tx, ty, tz = (t[ids, a].astype(np.float32) for a in range(3))
    sample = np.unique(np.concatenate([rng.choice(ids, min(300, len(ids)), replace=False), ids[ids < 40]]))
    for i in sample:
        dx, dy, dz = tx - t[i, 0], ty - t[i, 1], tz - t[i, 2]
        d = (dx * dx + dy * dy) + dz * dz
        order = np.lexsort((ids, d))[:k]
        m = len(order)
        np.testing.assert_array_equal(idx[i, :m], ids[order].astype(np.int32), err_msg=f"query {i}")
        np.testing.assert_array_equal(sqd[i, :m], d[order])
        assert (idx[i, m:] == -1).all()


@pytest.mark.parametrize("outlier", ["RADIUS", "STATISTICAL", "NONE"])
@pytest.mark.parametrize("downsample", ["VOXELGRID", "NONE"])
def test_fused_prefilter_equals_the_three_calls(street_pair_vlp16, outlier, downsample):
    """mrgfe_prefilter (one upload, the passes back to back in HBM, one download) == distance_filter -> VoxelGrid ->
    outlier removal called one after the other == the oracle chain."""
    from mrg_slam_amd import RadiusOutlierRemoval, StatisticalOutlierRemoval, VoxelGrid, distance_filter, prefilter
    from oracle import oracle as orc

    raw = street_pair_vlp16[0]
    params = {"downsample_method": downsample, "outlier_removal_method": outlier, "downsample_resolution": 0.2, "distance_far_thresh": 30.0}
    got = prefilter(raw, params)
    c = distance_filter(raw, 0.1, 30.0)
    e = orc.distance_filter(raw, 0.1, 30.0)
    if downsample == "VOXELGRID":
        vg = VoxelGrid()
        vg.setLeafSize(0.2)
        vg.setInputCloud(c)
        c = vg.filter()
        e, _ = orc.voxelgrid(e, 0.2, 1)
    if outlier == "RADIUS":
        ro = RadiusOutlierRemoval()
        ro.setRadiusSearch(0.5)
        ro.setMinNeighborsInRadius(2)
        ro.setInputCloud(c)
        c = ro.filter()
        e, _ = orc.radius_outlier(e, 0.5, 2)
    elif outlier == "STATISTICAL":
        so = StatisticalOutlierRemoval()
        so.setMeanK(30)
        so.setStddevMulThresh(1.2)
        so.setInputCloud(c)
        c = so.filter()
        e, _ = orc.statistical_outlier(e, 30, 1.2)
    np.testing.assert_array_equal(got, c)
    np.testing.assert_array_equal(got, e)
    assert len(prefilter(np.zeros((0, 4), np.float32), params)) == 0


def test_prefilter_to_device_feeds_the_scan_matcher_without_leaving_hbm(street_pair_vlp16):
    """mrgfe_prefilter_device leaves the filtered scan in device memory; registering from there gives the result of the
    host round trip (prefiltering_component -> scan_matching_odometry_component as two ROS nodes)."""
    import torch

    from mrg_slam_amd import NdtHip, prefilter, prefilter_to_device

    tgt_raw, src_raw, _ = street_pair_vlp16
    tgt, src = prefilter(tgt_raw), prefilter(src_raw)
    buf = torch.empty((len(src_raw), 4), dtype=torch.float32, device="cuda:0")
    m = prefilter_to_device(src_raw, buf.data_ptr(), len(src_raw))
    assert m == len(src)
    np.testing.assert_array_equal(buf[:m].cpu().numpy(), src)
    a, b = NdtHip(transformation_epsilon=0.01), NdtHip(transformation_epsilon=0.01)
    for r in (a, b):
        r.setInputTarget(tgt)
    a.setInputSource(src)
    b.setInputSourceDevice(buf.data_ptr(), m)
    a.align(np.eye(4))
    b.align(np.eye(4))
    np.testing.assert_array_equal(a.getFinalTransformation(), b.getFinalTransformation())
    with pytest.raises(ValueError):
        prefilter_to_device(src_raw, buf.data_ptr(), 10)


def test_information_matrix_of_graph_edges_host_clouds_and_store():
    """calc_information_matrix for an odometry edge and a loop edge (graph_database.cpp:139-142, 579-581): host clouds and the
    store-resident keyed variant give the oracle's fitness score (f64 rounding) and therefore its 6x6 matrix; the keyed variant
    reuses key1's search grid across edges."""
    from mrg_slam_amd import InformationMatrixCalculator, MapCloudStore, prefilter, synth
    from oracle import oracle as orc

    scene = synth.street_scene()
    poses = synth.arc_trajectory(4)
    clouds = [prefilter(synth.synth_lidar(scene, poses[k], "VLP16", 900 + k)) for k in range(4)]
    calc = InformationMatrixCalculator()
    store = MapCloudStore()
    for k, c in enumerate(clouds):
        store.add(k + 1, c)
    rng = np.random.default_rng(1)
    for (a, b) in [(1, 0), (2, 1), (3, 2), (3, 0), (3, 1), (0, 3)]:
        rel = synth.perturb_pose(np.linalg.inv(poses[a]) @ poses[b], rng, (0.05, 0.05, 0.02), (0.2, 0.2, 0.5))
        o_inf, o_fit = orc.calc_information_matrix(clouds[a], clouds[b], rel)
        inf = calc.calc_information_matrix(clouds[a], clouds[b], rel)
        assert calc.last_fitness_score == pytest.approx(o_fit, rel=1e-9)
        np.testing.assert_allclose(inf, o_inf, rtol=1e-9, atol=0)
        inf_k = calc.calc_information_matrix_keyed(store, a + 1, b + 1, rel)
        assert calc.last_fitness_score == pytest.approx(o_fit, rel=1e-9)
        np.testing.assert_array_equal(inf_k, inf)
        assert store.fitness(a + 1, b + 1, rel, 0.04) == pytest.approx(orc.calc_fitness_score(clouds[a], clouds[b], rel, 0.04), rel=1e-9)
    const = InformationMatrixCalculator({"use_const_inf_matrix": True, "const_stddev_x": 0.25})
    np.testing.assert_array_equal(const.calc_information_matrix(clouds[0], clouds[1], np.eye(4)), np.diag([4.0] * 3 + [10.0] * 3))
    with pytest.raises(Exception):
        store.fitness(1, 99, np.eye(4))


def test_device_driven_prefilter_chain_equals_the_host_driven_one_and_the_oracle():
    """Round 4: mrgfe_prefilter keeps the stages' point counts on the device and waits once (csrc/filters.hip: PfState).  Same bits as the host-driven
    stages of round 3 and as the oracle chain — on ordinary scans and on the ones the device-driven chain hands back: nothing survives the distance
    filter, non-finite points, PCL's 'leaf size is too small' pass-through, a radius grid beyond its table."""
    from mrg_slam_amd import prefilter, synth
    from mrg_slam_amd._lib import lib
    from oracle import oracle as orc

    def oracle_chain(c, p):
        c = orc.distance_filter(c, p.get("distance_near_thresh", 0.1), p.get("distance_far_thresh", 35.0))
        c, _ = orc.voxelgrid(c, p.get("downsample_resolution", 0.1), p.get("downsample_min_points_per_voxel", 1))
        c, _ = orc.radius_outlier(c, p.get("radius_radius", 0.5), p.get("radius_min_neighbors", 2))
        return c

    sc = synth.street_scene()
    scan = synth.synth_lidar(sc, np.eye(4), "VLP16", 4711)
    rng = np.random.default_rng(5)
    with_nan = scan.copy()
    with_nan[rng.choice(len(scan), 200, replace=False), rng.integers(0, 3, 200)] = np.nan
    with_nan[7, 0] = np.inf
    spread = small_cloud(3000, 3, extent=(30000.0, 30000.0, 30000.0))  # 0.01 m leaves over 60 km: PCL passes the cloud through
    cases = [("street scan", scan, {}), ("coarser leaf, more neighbours", scan, {"downsample_resolution": 0.3, "radius_min_neighbors": 4, "radius_radius": 0.8}),
             ("min points per voxel 2", scan, {"downsample_min_points_per_voxel": 2}), ("nothing survives the distance filter", scan, {"distance_near_thresh": 500.0, "distance_far_thresh": 600.0}),
             ("non-finite points", with_nan, {}), ("leaf size too small", spread, {"downsample_resolution": 0.01, "distance_far_thresh": 1e9}),
             ("radius grid beyond its table", small_cloud(4000, 9, extent=(400.0, 400.0, 40.0)), {"distance_far_thresh": 1e9, "radius_radius": 0.4}),
             ("three points", scan[:3], {}), ("all non-finite", np.full((50, 4), np.nan, dtype=np.float32), {})]
    try:
        for name, cloud, p in cases:
            assert lib().mrgfe_dbg_set_prefilter_device_driven(1) == 1
            fast = prefilter(cloud, p)
            assert lib().mrgfe_dbg_set_prefilter_device_driven(0) == 0
            slow = prefilter(cloud, p)
            np.testing.assert_array_equal(fast, slow, err_msg=name)
            np.testing.assert_array_equal(fast, oracle_chain(cloud, p), err_msg=name)
    finally:
        lib().mrgfe_dbg_set_prefilter_device_driven(1)


def test_device_driven_prefilter_chain_at_tile_boundaries_and_random_parameters():
    """The chain's compaction kernels add up tile counts, rank by ballots and merge per-tile boundary boxes (csrc/filters.hip: pf_compact_kernel):
    sizes around the 2048-point tiles and the 256-point rounds, survivors in the last tile only / the first tile only / none, random leaves, radii and
    neighbour counts — always the host-driven stages' and the oracle's output, bit for bit."""
    from mrg_slam_amd import prefilter
    from mrg_slam_amd._lib import lib
    from oracle import oracle as orc

    rng = np.random.default_rng(20261004)
    sizes = [1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 6145, 10000]
    try:
        for trial, n in enumerate(sizes + [int(v) for v in rng.integers(300, 30000, 14)]):
            cloud = small_cloud(n, 1000 + trial, extent=(rng.uniform(3, 40), rng.uniform(3, 40), rng.uniform(0.5, 6)))
            kind = trial % 5
            if kind == 1:    # only the last points survive the distance filter
                cloud[: max(0, n - 7), :3] *= np.float32(1e-4)
            elif kind == 2:  # only the first points do
                cloud[min(n, 5):, :3] = np.float32(900.0)
            elif kind == 3 and n > 10:
                cloud[rng.choice(n, max(1, n // 50), replace=False), rng.integers(0, 3)] = np.nan
            p = {"distance_near_thresh": 0.1, "distance_far_thresh": float(rng.choice([8.0, 35.0, 1e4])), "downsample_resolution": float(rng.choice([0.05, 0.1, 0.3, 1.0])),
                 "downsample_min_points_per_voxel": int(rng.choice([1, 1, 2])), "radius_radius": float(rng.choice([0.3, 0.5, 1.2])), "radius_min_neighbors": int(rng.choice([1, 2, 5]))}
            assert lib().mrgfe_dbg_set_prefilter_device_driven(1) == 1
            fast = prefilter(cloud, p)
            assert lib().mrgfe_dbg_set_prefilter_device_driven(0) == 0
            slow = prefilter(cloud, p)
            name = f"trial {trial}: n={n} kind={kind} {p}"
            np.testing.assert_array_equal(fast, slow, err_msg=name)
            c = orc.distance_filter(cloud, p["distance_near_thresh"], p["distance_far_thresh"])
            c, _ = orc.voxelgrid(c, p["downsample_resolution"], p["downsample_min_points_per_voxel"])
            c, _ = orc.radius_outlier(c, p["radius_radius"], p["radius_min_neighbors"])
            np.testing.assert_array_equal(fast, c, err_msg=name)
    finally:
        lib().mrgfe_dbg_set_prefilter_device_driven(1)


@pytest.mark.parametrize("leaf", [0.05, 0.1, 0.5, 5.0, 80.0])
def test_approx_voxelgrid_exact(street_pair_vlp16, leaf):
    """pcl::ApproximateVoxelGrid (downsample_method APPROX_VOXELGRID) — a sequential loop over a 512-entry history in the reference, 512 independent
    sequences on the GPU (csrc/filters.hip): the same points in the same ORDER as the oracle's loop, bit for bit — scans, shuffled clouds, clouds of one
    cell, duplicates, non-finite and out-of-int-range coordinates, one and zero points."""
    from mrg_slam_amd import ApproximateVoxelGrid
    from oracle import oracle as orc

    def hip(c):
        f = ApproximateVoxelGrid()
        f.setLeafSize(leaf, leaf, leaf)
        f.setInputCloud(c)
        return f.filter()

    rng = np.random.default_rng(17)
    scan = street_pair_vlp16[0]
    odd = small_cloud(6000, 5, extent=(40.0, 40.0, 5.0))
    odd[rng.choice(6000, 60, replace=False), rng.integers(0, 3, 60)] = np.nan
    odd[11, 1] = np.inf
    odd[12, 2] = -np.inf
    odd[13, 0] = 3e12   # beyond the int range after the division by the leaf
    odd[14, 0] = -3e12
    odd[100:140] = odd[100]  # duplicates
    for name, c in (("scan", scan), ("shuffled scan", scan[rng.permutation(len(scan))]), ("odd values", odd), ("one point", scan[:1]), ("two points", scan[:2]),
                    ("empty", np.zeros((0, 4), np.float32)), ("one cell", (small_cloud(500, 2) * np.float32(1e-3)).astype(np.float32))):
        got, exp = hip(c), orc.approx_voxelgrid(c, leaf)
        assert got.shape == exp.shape, (name, got.shape, exp.shape)
        np.testing.assert_array_equal(got, exp, err_msg=name)  # (NaN centroids of cells with a non-finite coordinate compare equal)


def test_prefilter_chain_with_approx_voxelgrid(street_pair_vlp16):
    from mrg_slam_amd import prefilter
    from mrg_slam_amd.prefiltering import HipOps, OracleOps, PrefilteringComponent
    from oracle import oracle as orc

    raw = street_pair_vlp16[1]
    p = {"downsample_method": "APPROX_VOXELGRID", "downsample_resolution": 0.2}
    e = orc.distance_filter(raw, 0.1, 35.0)
    e = orc.approx_voxelgrid(e, 0.2)
    e, _ = orc.radius_outlier(e, 0.5, 2)
    np.testing.assert_array_equal(prefilter(raw, p), e)
    a = PrefilteringComponent(p, ops=HipOps()).cloud_callback(raw, 0.0, "base_link")
    b = PrefilteringComponent(p, ops=OracleOps(orc)).cloud_callback(raw, 0.0, "base_link")
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, e)
