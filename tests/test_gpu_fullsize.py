"""GPU: BASELINE.json's full sizes (VLP-64, ~130k points per scan).  The CPU oracle handles one such pair in well under
a second, so parity is checked directly against it, plus size-independent properties of the hot path (sortedness /
permutation of the radix sort at batch scale, source-order invariance of align, prefilter idempotence)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vlp64():
    from mrg_slam_amd import synth

    scene = synth.street_scene()
    tgt, src, rel = synth.scan_pair(0, "VLP64", scene)
    return tgt, src, rel


def test_prefilter_chain_exact_at_full_size(vlp64):
    from mrg_slam_amd import RadiusOutlierRemoval, VoxelGrid, distance_filter
    from oracle import oracle as orc

    tgt, _, _ = vlp64
    assert len(tgt) > 120000
    d = distance_filter(tgt, 0.1, 35.0)
    np.testing.assert_array_equal(d, orc.distance_filter(tgt, 0.1, 35.0))
    vg = VoxelGrid()
    vg.setLeafSize(0.1)
    vg.setInputCloud(d)
    v = vg.filter()
    np.testing.assert_array_equal(v, orc.voxelgrid(d, 0.1, 1)[0])
    ro = RadiusOutlierRemoval()
    ro.setInputCloud(v)
    r = ro.filter()
    np.testing.assert_array_equal(r, orc.radius_outlier(v, 0.5, 2)[0])
    # idempotence: every survivor still has its neighbours among the survivors' superset -> filtering the raw voxel cloud
    # twice with the same parameters removes nothing new only if no removed point was a needed neighbour; what does hold
    # exactly is that the distance filter and the voxel grid are idempotent on their own output
    np.testing.assert_array_equal(distance_filter(d, 0.1, 35.0), d)


@pytest.mark.parametrize("eps", [0.1, 0.01])
def test_ndt_align_full_size_matches_oracle_and_is_order_invariant(vlp64, eps):
    from mrg_slam_amd import NdtHip, distance_filter, synth
    from oracle import oracle as orc

    tgt, src, rel = vlp64
    ft, fs = distance_filter(tgt), distance_filter(src)
    guess = synth.warm_guess(rel, 0)
    g = NdtHip(transformation_epsilon=eps)
    o = orc.Ndt(transformation_epsilon=eps, num_threads=8)
    assert g.setInputTarget(ft) == 0 and o.setInputTarget(ft) == 0
    g.setInputSource(fs)
    o.setInputSource(fs)
    g.align(guess)
    o.align(guess)
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4 and synth.rotation_angle(Tg, To) <= 1e-4
    assert (g.hasConverged(), g.getFinalNumIteration(), g.evals) == (o.hasConverged(), o.getFinalNumIteration(), o.evals)
    assert g.getFitnessScore() == pytest.approx(o.getFitnessScore(), rel=1e-9)
    gk, gn, gm, gi = g.leaves()
    ok, on, om, _, oi = o.leaves()
    np.testing.assert_array_equal(gk, ok)
    np.testing.assert_array_equal(gn, on)
    np.testing.assert_array_equal(gm, om)
    np.testing.assert_array_equal(gi[on >= 6], oi[on >= 6])
    # property: the order of the source points only changes the order of f64 additions
    perm = np.random.default_rng(1).permutation(len(fs))
    g.setInputSource(fs[perm])
    g.align(guess)
    Tp = g.getFinalTransformation()
    assert np.linalg.norm(Tp[:3, 3].astype(np.float64) - Tg[:3, 3]) <= 1e-6 and synth.rotation_angle(Tp, Tg) <= 1e-6


def test_radix_sort_properties_at_batch_scale():
    from mrg_slam_amd import default_context
    from mrg_slam_amd._lib import check, lib

    n = 4_200_000  # 32 scans of ~130k keys
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 1 << 17, n, dtype=np.uint32)  # ~71 x 71 x 25 voxels
    vals = np.arange(n, dtype=np.uint32)
    ok, ov = np.empty_like(keys), np.empty_like(vals)
    p = C.POINTER(C.c_uint32)
    check(lib().mrgfe_dbg_sort_pairs(default_context()._h, keys.ctypes.data_as(p), vals.ctypes.data_as(p), n, 17, ok.ctypes.data_as(p), ov.ctypes.data_as(p)))
    assert (np.diff(ok.astype(np.int64)) >= 0).all()  # sorted
    np.testing.assert_array_equal(ok, keys[ov])  # a permutation carrying its keys
    same = ok[1:] == ok[:-1]
    assert (ov[1:][same] > ov[:-1][same]).all()  # stable
    assert np.bincount(ov, minlength=n).max() == 1


@pytest.mark.parametrize("method", ["GICP", "SMALL_GICP", "VGICP", "PCL_GICP", "ICP_RECIPROCAL"])
def test_gicp_family_full_size_matches_oracle(vlp64, method):
    """BASELINE config[2] at its stated size: scan-to-keyframe GICP on ~130k-point clouds (k = 20 covariances over both clouds, 1-NN
    correspondences within 2 m).  The final transform must be the oracle's (bar 1e-4 m / 1e-4 rad; it is bit-identical on this
    pair), with the same convergence flag and iteration count."""
    from mrg_slam_amd import GicpHip, IcpHip, PclGicpHip, SmallGicpHip, VgicpHip, distance_filter, synth
    from oracle import oracle as orc

    tgt, src, rel = vlp64
    ft, fs = distance_filter(tgt), distance_filter(src)
    assert len(ft) > 120000 and len(fs) > 120000
    guess = synth.warm_guess(rel, 1)
    g, o = {"GICP": (GicpHip(transformation_epsilon=0.01), orc.FastGicp(transformation_epsilon=0.01, num_threads=32)),
            "SMALL_GICP": (SmallGicpHip(transformation_epsilon=0.01), orc.SmallGicp(transformation_epsilon=0.01, num_threads=32)),
            "VGICP": (VgicpHip(resolution=1.0, transformation_epsilon=0.01), orc.FastVgicp(resolution=1.0, transformation_epsilon=0.01, num_threads=32)),
            "PCL_GICP": (PclGicpHip(transformation_epsilon=0.01), orc.PclGicp(transformation_epsilon=0.01, num_threads=32)),
            "ICP_RECIPROCAL": (IcpHip(transformation_epsilon=0.01, use_reciprocal_correspondences=True),
                               orc.Icp(transformation_epsilon=0.01, use_reciprocal_correspondences=True))}[method]
    g.setInputTarget(ft)
    o.setInputTarget(ft)
    g.setInputSource(fs)
    o.setInputSource(fs)
    g.align(guess)
    o.align(guess)
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4 and synth.rotation_angle(Tg, To) <= 1e-4
    assert g.hasConverged() == o.hasConverged() and g.getFinalNumIteration() == o.getFinalNumIteration()
    assert np.linalg.norm(Tg[:3, 3] - rel[:3, 3]) < 0.1  # and it found the motion


def test_small_gicp_batch_full_size_takes_the_correspondence_passes_and_equals_single_registrations(vlp64):
    """Five 130k-point candidates against one 130k-point keyframe (650k queries a round: the batch takes nn_nearest_batch's passes by default,
    single registrations the lane-group search): every record equals the single registration's bit for bit, one is held against the oracle,
    and the fitness grids of the batch (built together) give the single registration's score."""
    from mrg_slam_amd import BatchMatcher, SmallGicpHip, synth
    from mrg_slam_amd._lib import SMALL_GICP_HIP
    from mrg_slam_amd.registration import default_params, result_matrix
    from oracle import oracle as orc

    tgt, src, rel = vlp64
    prm = default_params(SMALL_GICP_HIP)
    prm.transformation_epsilon = 0.01
    rng = np.random.default_rng(17)
    guesses = [synth.perturb_pose(rel, rng) for _ in range(4)] + [np.eye(4)]
    sources = [src, src[:-999], src[::2], src[3000:], src]
    bm = BatchMatcher(prm)
    t = bm.add_target(tgt)
    for s_, g_ in zip(sources, guesses):
        bm.add_pair(t, s_, g_)
    res = bm.align(fitness_max_range=float("inf"))
    for k, (s_, g_) in enumerate(zip(sources, guesses)):
        reg = SmallGicpHip(transformation_epsilon=0.01)
        reg.setInputTarget(tgt)
        reg.setInputSource(s_)
        reg.align(g_)
        np.testing.assert_array_equal(result_matrix(res[k]), reg.getFinalTransformation(), err_msg=f"pair {k}")
        assert res[k]["converged"] == int(reg.hasConverged()) and res[k]["iterations"] == reg.getFinalNumIteration()
        assert res[k]["fitness"] == pytest.approx(reg.getFitnessScore(), rel=1e-12)
    o = orc.SmallGicp(transformation_epsilon=0.01, num_threads=16)
    o.setInputTarget(tgt)
    o.setInputSource(sources[2])
    o.align(guesses[2])
    To = o.getFinalTransformation()
    Tg = result_matrix(res[2])
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4 and synth.rotation_angle(Tg, To) <= 1e-4
    assert res[2]["iterations"] == o.getFinalNumIteration()


def test_registration_far_above_the_stated_size():
    """Fifteen times BASELINE's scan size: a 2-million-point NDT registration (31k tiles of source points, a radix sort of ~1000 tiles per pass, an exact-NN grid
    for getFitnessScore over 2 M targets) and a 600k-point SMALL_GICP one (k-NN covariances of both clouds) — the indices, tile tables and byte offsets of the path at
    a size where a 32-bit slip would show; both against the oracle, bit for bit."""
    import os

    from conftest import small_cloud
    from mrg_slam_amd import NdtHip, SmallGicpHip, synth
    from oracle import oracle as orc

    threads = max(2, min(32, os.cpu_count() or 2))
    n = 2_000_000
    tgt = small_cloud(n, 31, extent=(120.0, 90.0, 12.0))
    rel = synth.make_pose([0.4, -0.2, 0.05], synth.rot_xyz(0.004, -0.003, 0.02))
    src = orc.transform_points(np.linalg.inv(rel), tgt[: n - 12345])
    src[:, :3] += np.random.default_rng(3).normal(0, 0.01, (len(src), 3)).astype(np.float32)
    r = NdtHip(resolution=1.0, transformation_epsilon=0.01)
    r.setInputTarget(tgt)
    r.setInputSource(src)
    r.align(np.eye(4))
    o = orc.Ndt(resolution=1.0, transformation_epsilon=0.01, num_threads=threads)
    o.setInputTarget(tgt)
    o.setInputSource(src)
    o.align(np.eye(4))
    assert r.getFinalNumIteration() == o.getFinalNumIteration() and bool(r.hasConverged()) == bool(o.hasConverged())
    np.testing.assert_array_equal(r.getFinalTransformation(), o.getFinalTransformation())
    assert r.getFitnessScore(float("inf")) == pytest.approx(o.getFitnessScore(float("inf")), rel=1e-12)
    m = 600_000
    g = SmallGicpHip(transformation_epsilon=0.01)
    g.setInputTarget(tgt[:m])
    g.setInputSource(src[:m])
    g.align(np.eye(4))
    og = orc.SmallGicp(transformation_epsilon=0.01, num_threads=threads)
    og.setInputTarget(tgt[:m])
    og.setInputSource(src[:m])
    og.align(np.eye(4))
    assert g.getFinalNumIteration() == og.getFinalNumIteration()
    np.testing.assert_array_equal(g.getFinalTransformation(), og.getFinalTransformation())
