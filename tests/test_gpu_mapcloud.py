"""GPU: SURVEY.md §8(f) rows 2 and 4 — map-cloud generation (MapCloudGenerator + ApproximateMeanVoxelGrid), other-robot
point removal and deskewing through the C ABI against the CPU oracle.  Bit-exact: membership tests and float expressions
in one documented order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _keyframes(K, n, seed, spread=6.0):
    from mrg_slam_amd import KeyFrameSnapshot, synth

    rng = np.random.default_rng(seed)
    kfs = []
    for k in range(K):
        c = rng.normal(0, spread, (n + 41 * k, 4)).astype(np.float32)
        c[:, 3] = rng.uniform(0, 1, len(c)).astype(np.float32)
        T = synth.make_pose([2.5 * k, -1.0 * k, 0.1 * k], synth.rot_z(0.15 * k))
        kfs.append(KeyFrameSnapshot(T, c, first_keyframe=(k == 0)))
    return kfs


@pytest.mark.parametrize("K,n,resolution,min_pts,far,skip", [(5, 4000, 0.5, 1, 10000.0, False), (5, 4000, 0.25, 2, 9.0, False), (3, 20000, 0.1, 1, -1.0, True),
                                                              (4, 3000, 0.0, 1, 6.0, False), (1, 2000, 1.0, 1, 10000.0, False)])
def test_map_cloud_matches_oracle(K, n, resolution, min_pts, far, skip):
    from mrg_slam_amd import MapCloudGenerator
    from oracle import oracle as orc

    kfs = _keyframes(K, n, 11 + K)
    got = MapCloudGenerator().generate(kfs, resolution, min_pts, far, skip)
    want, status = orc.map_cloud_generate([k.cloud for k in kfs], [k.pose for k in kfs], [k.first_keyframe for k in kfs], resolution, min_pts, far, skip)
    assert status == 0 and got is not None
    np.testing.assert_array_equal(got, want)
    assert len(got) > 0


def test_map_cloud_street_keyframes_match_oracle():
    """keyframes of the synthetic street (prefiltered VLP-16 scans along the arc), the reference's default parameters
    (config/mrg_slam.yaml:235-237)"""
    from mrg_slam_amd import KeyFrameSnapshot, MapCloudGenerator, prefilter, synth
    from oracle import oracle as orc

    scene = synth.street_scene()
    poses = synth.arc_trajectory(6, step=2.0)
    kfs = [KeyFrameSnapshot(poses[k], prefilter(synth.synth_lidar(scene, poses[k], "VLP16", synth.BASE_SEED + k)), k == 0) for k in range(6)]
    got = MapCloudGenerator().generate(kfs, 0.1, 1, 10000.0, False)
    want, status = orc.map_cloud_generate([k.cloud for k in kfs], [k.pose for k in kfs], [k.first_keyframe for k in kfs], 0.1, 1, 10000.0, False)
    assert status == 0
    np.testing.assert_array_equal(got, want)
    # overlapping scans of a static street: the union is much smaller than the sum
    assert len(got) < 0.9 * sum(len(k.cloud) for k in kfs)


def test_map_cloud_reference_nullptr_cases():
    from mrg_slam_amd import MapCloudGenerator

    gen = MapCloudGenerator()
    kfs = _keyframes(2, 500, 3)
    assert gen.generate([], 0.5) is None                                    # no keyframes
    assert gen.generate(kfs, 0.5, 1, 1e-3, False) is None                   # every point cut, more than one keyframe
    one = gen.generate(kfs[:1], 0.5, 1, 1e-3, False)                        # one keyframe: an empty cloud, not a failure
    assert one is not None and len(one) == 0
    none_left = gen.generate(kfs, 0.5, 10**6, 10000.0, False)               # the count threshold removes every voxel
    assert none_left is not None and len(none_left) == 0
    skipped = gen.generate(kfs, 0.5, 1, 10000.0, True)                      # first keyframe skipped
    assert skipped is not None and len(skipped) > 0


@pytest.mark.parametrize("n,centres,radius", [(30000, [[1.0, 0.5, 0.0], [-3.0, 2.0, 0.2]], 1.5), (5000, [[0.0, 0.0, 0.0]], 100.0), (5000, [[50.0, 50.0, 50.0]], 0.5), (0, [[0.0, 0.0, 0.0]], 1.0)])
def test_remove_points_near_matches_oracle(n, centres, radius):
    from mrg_slam_amd import remove_points_near
    from oracle import oracle as orc

    c = np.random.default_rng(n + 1).normal(0, 4, (n, 4)).astype(np.float32)
    kept, removed = remove_points_near(c, centres, radius)
    ekept, eremoved = orc.remove_points_near(c, centres, radius)
    np.testing.assert_array_equal(kept, ekept)
    np.testing.assert_array_equal(removed, eremoved)
    assert len(kept) + len(removed) == n


def test_remove_points_near_without_centres_keeps_everything():
    from mrg_slam_amd import remove_points_near

    c = np.random.default_rng(2).normal(0, 4, (1000, 4)).astype(np.float32)
    kept, removed = remove_points_near(c, np.zeros((0, 3)), 2.0)
    np.testing.assert_array_equal(kept, c)
    assert len(removed) == 0


@pytest.mark.parametrize("n,w,period", [(50000, [0.3, -0.2, 0.8], 0.1), (1234, [0.0, 0.0, 0.0], 0.1), (7, [5.0, 1.0, -2.0], 0.05)])
def test_deskew_matches_oracle(n, w, period):
    from mrg_slam_amd import deskew
    from oracle import oracle as orc

    c = np.random.default_rng(n).normal(0, 15, (n, 4)).astype(np.float32)
    np.testing.assert_array_equal(deskew(c, w, period), orc.deskew(c, w, period))


def test_map_store_equals_the_host_cloud_generator():
    """Keyframe clouds resident in HBM (mrgfe_map_store): same map as MapCloudGenerator.generate over host clouds for the same
    keyframes in the same order, for changing poses, subsets, reordering, the far cut and skip_first_cloud."""
    from mrg_slam_amd import KeyFrameSnapshot, MapCloudGenerator, MapCloudStore, synth

    rng = np.random.default_rng(71)
    clouds = {k: np.c_[rng.normal(0, 6, (1500 + 100 * k, 3)), rng.uniform(0, 1, 1500 + 100 * k)].astype(np.float32) for k in range(1, 7)}
    clouds[7] = np.zeros((0, 4), np.float32)  # an empty keyframe
    store, gen = MapCloudStore(), MapCloudGenerator()
    assert store.bytes() == 0 and store.has(1) is None
    for k, c in clouds.items():
        store.add(k, c)
        store.add(k, c)  # again: no-op
    assert store.bytes() == 16 * sum(len(c) for c in clouds.values()) and store.has(3) == len(clouds[3])
    with pytest.raises(RuntimeError):
        store.add(3, clouds[4])  # same key, other size

    def poses_for(keys, seed):
        r = np.random.default_rng(seed)
        return [synth.make_pose(r.normal(0, 4, 3), synth.rot_xyz(*r.normal(0, 0.3, 3))) for _ in keys]

    cases = [([1, 2, 3, 4, 5, 6, 7], 0.5, 1, 10000.0, False), ([6, 2, 4], 0.25, 2, 9.0, False), ([1, 2, 3], 0.0, 1, 7.0, False), ([3, 1, 2, 5], 0.5, 1, 10000.0, True)]
    for it, (keys, res, minp, far, skip) in enumerate(cases * 2):  # second round: same keyframes, new poses (after optimisation)
        poses = poses_for(keys, 100 + it)
        first = [k == keys[0] for k in keys]
        got = store.generate(keys, poses, first, res, minp, far, skip)
        exp = gen.generate([KeyFrameSnapshot(p, clouds[k], f) for k, p, f in zip(keys, poses, first)], res, minp, far, skip)
        np.testing.assert_array_equal(got, exp)
    assert store.generate([], [], None) is None                                      # no keyframes: nullptr in the reference
    assert store.generate([7, 7], poses_for([7, 7], 1), None) is None                # nothing left from more than one keyframe
    with pytest.raises(RuntimeError):
        store.generate([1, 99], poses_for([1, 99], 2))                               # unknown keyframe


@pytest.mark.parametrize("extent_xy,extent_z,resolution", [(400.0, 40.0, 0.1), (3000.0, 60.0, 0.1), (60000.0, 300.0, 0.05), (5.0, 2.0, 0.001)])
def test_map_cloud_of_any_extent(extent_xy, extent_z, resolution):
    """The reference's ApproximateMeanVoxelGrid keys a hash map on the integer cell and has no extent limit
    (ApproximateMeanVoxelGrid.hpp:85-91); a KITTI-sized map at the default 0.1 m has far more than 2^31 cells in its bounding box
    (round 1 returned MRGFE_ERR_OVERFLOW from 200 x 200 x 30 m on).  Keys are the cells themselves, bit-packed: 32 bits or fewer
    are one radix sort, more are two (3 km at 0.1 m: 15 + 15 + 10 bits; 60 km at 0.05 m: 21 + 21 + 13).  Bit-exact against the oracle."""
    from mrg_slam_amd import KeyFrameSnapshot, MapCloudGenerator, synth
    from oracle import oracle as orc

    rng = np.random.default_rng(int(extent_xy))
    kfs = []
    for k in range(6):
        c = rng.normal(0, 4.0, (6000, 4)).astype(np.float32)
        c[:, 2] *= 0.3
        c[:, 3] = rng.uniform(0, 1, len(c)).astype(np.float32)
        centre = [rng.uniform(-extent_xy / 2, extent_xy / 2), rng.uniform(-extent_xy / 2, extent_xy / 2), rng.uniform(-extent_z / 2, extent_z / 2)]
        kfs.append(KeyFrameSnapshot(synth.make_pose(centre, synth.rot_z(rng.uniform(0, 6.28))), c, first_keyframe=(k == 0)))
    kfs[1] = KeyFrameSnapshot(kfs[0].pose, kfs[1].cloud, False)  # two keyframes overlap: shared voxels
    got = MapCloudGenerator().generate(kfs, resolution, 1, 10000.0, False)
    want, status = orc.map_cloud_generate([k.cloud for k in kfs], [k.pose for k in kfs], [k.first_keyframe for k in kfs], resolution, 1, 10000.0, False)
    assert status == 0 and got is not None
    np.testing.assert_array_equal(got, want)
    got2 = MapCloudGenerator().generate(kfs, resolution, 2, 10000.0, False)
    want2, _ = orc.map_cloud_generate([k.cloud for k in kfs], [k.pose for k in kfs], [k.first_keyframe for k in kfs], resolution, 2, 10000.0, False)
    np.testing.assert_array_equal(got2, want2)
    assert len(got2) < len(got)
