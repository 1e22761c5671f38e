"""CPU oracle for SURVEY.md §8(f) rows 2 and 4 (map cloud, other-robot removal, deskewing) against independent numpy
restatements of /root/reference/src/mrg_slam/map_cloud_generator.cpp:14-86,
include/pcl/filters/ApproximateMeanVoxelGrid.hpp:63-126, apps/mrg_slam_component.cpp:396-429 and
apps/prefiltering_component.cpp:231-292."""
import numpy as np
import pytest

from mrg_slam_amd import synth
from oracle import oracle as orc


def _keyframes(K=4, n=1500, seed=5):
    rng = np.random.default_rng(seed)
    clouds, poses = [], []
    for k in range(K):
        c = rng.normal(0, 6, (n + 37 * k, 4)).astype(np.float32)
        c[:, 3] = rng.uniform(0, 1, len(c)).astype(np.float32)
        clouds.append(c)
        poses.append(synth.make_pose([2.0 * k, -1.0 * k, 0.1 * k], synth.rot_z(0.2 * k)))
    return clouds, poses


def _numpy_map_cloud(clouds, poses, first, resolution, min_pts, far, skip_first):
    """dict-of-voxels restatement: float32 arithmetic in the documented order, voxels sorted by (iz, iy, ix)."""
    pts = []
    far_sq = np.float32(far) * np.float32(far)
    for c, T, f in zip(clouds, poses, first):
        if f and skip_first:
            continue
        P = np.asarray(T, dtype=np.float64).astype(np.float32)
        x, y, z = c[:, 0], c[:, 1], c[:, 2]
        keep = np.ones(len(c), dtype=bool)
        if far > 0:
            keep = ~(((x * x + y * y) + z * z) > far_sq)
        q = np.empty((len(c), 4), dtype=np.float32)
        for r in range(3):
            q[:, r] = ((P[r, 0] * x + P[r, 1] * y) + P[r, 2] * z) + P[r, 3] * np.float32(1.0)
        q[:, 3] = c[:, 3]
        pts.append(q[keep])
    cloud = np.concatenate(pts) if pts else np.zeros((0, 4), np.float32)
    if resolution <= 0:
        return cloud
    inv = np.float32(1.0) / np.float32(resolution)
    ijk = np.floor(cloud[:, :3] * inv).astype(np.int64)
    vox = {}
    for i in range(len(cloud)):
        key = (ijk[i, 2], ijk[i, 1], ijk[i, 0])
        s = vox.setdefault(key, [0, np.zeros(4, dtype=np.float32)])
        s[0] += 1
        s[1] = (s[1] + cloud[i]).astype(np.float32)
    out = [s[1] / np.float32(s[0]) for key, s in sorted(vox.items()) if s[0] >= min_pts]
    return np.array(out, dtype=np.float32).reshape(-1, 4)


@pytest.mark.parametrize("resolution,min_pts,far,skip", [(0.5, 1, 10000.0, False), (1.0, 3, 8.0, False), (0.5, 1, -1.0, True), (0.0, 1, 7.0, False)])
def test_map_cloud_matches_numpy(resolution, min_pts, far, skip):
    clouds, poses = _keyframes()
    first = [True, False, False, False]
    got, status = orc.map_cloud_generate(clouds, poses, first, resolution, min_pts, far, skip)
    want = _numpy_map_cloud(clouds, poses, first, resolution, min_pts, far, skip)
    assert status == 0
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_map_cloud_reference_failure_cases():
    clouds, poses = _keyframes(K=2)
    assert orc.map_cloud_generate([], [], None)[1] == -1                                       # no keyframes -> nullptr
    got, status = orc.map_cloud_generate(clouds, poses, None, 0.5, 1, 1e-3, False)               # everything cut, K > 1 -> nullptr
    assert status == -2 and len(got) == 0
    got, status = orc.map_cloud_generate(clouds[:1], poses[:1], None, 0.5, 1, 1e-3, False)       # K == 1: empty cloud, not a failure
    assert status == 0 and len(got) == 0
    got, status = orc.map_cloud_generate(clouds, poses, None, 0.5, 10**6, 10000.0, False)        # threshold removes every voxel: empty, ok
    assert status == 0 and len(got) == 0


def test_remove_points_near_matches_numpy():
    rng = np.random.default_rng(3)
    c = rng.normal(0, 4, (5000, 4)).astype(np.float32)
    centres = np.array([[1.0, 0.5, 0.0], [-3.0, 2.0, 0.2]], dtype=np.float32)
    kept, removed = orc.remove_points_near(c, centres, 1.5)
    r2 = np.float32(1.5 * 1.5)
    gone = np.zeros(len(c), dtype=bool)
    for ctr in centres:
        d = c[:, :3] - ctr
        gone |= ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]) < r2
    assert np.array_equal(kept, c[~gone]) and np.array_equal(removed, c[gone])
    assert 0 < gone.sum() < len(c)
    k0, r0 = orc.remove_points_near(c, np.zeros((0, 3)), 1.5)
    assert np.array_equal(k0, c) and len(r0) == 0


def test_deskew_is_the_small_angle_rotation():
    rng = np.random.default_rng(4)
    c = rng.normal(0, 10, (4000, 4)).astype(np.float32)
    w = np.array([0.3, -0.2, 0.8], dtype=np.float32)
    period = 0.1
    got = orc.deskew(c, w, period)
    assert np.array_equal(got[:, 3], c[:, 3])
    assert np.array_equal(got[0, :3], c[0, :3])  # delta_t = 0 for the first point
    # float64 restatement: q = (1, dt/2 * -w), p' = q^-1 p q (q is not normalised; the inverse divides by |q|^2)
    i = np.arange(len(c), dtype=np.float64)
    dt = period * i / len(c)
    v = (dt[:, None] / 2.0) * (-w.astype(np.float64))[None, :]
    n2 = 1.0 + (v * v).sum(1)
    iv, iw = -v / n2[:, None], 1.0 / n2
    p = c[:, :3].astype(np.float64)
    uv = 2.0 * np.cross(iv, p)
    want = p + iw[:, None] * uv + np.cross(iv, uv)
    assert np.abs(got[:, :3] - want).max() < 5e-6 * 40
    # and it is a rotation by about |w| * dt about -w (up to the 1/|q|^2 scale the formula carries)
    ang = np.linalg.norm(w) * dt
    assert np.allclose(np.linalg.norm(got[:, :3], axis=1), np.linalg.norm(p, axis=1), rtol=2e-3)
    assert ang.max() < 0.1
