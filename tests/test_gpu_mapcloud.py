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
