"""GPU: GICP_HIP / SMALL_GICP_HIP against the CPU oracle (restated fast_gicp::FastGICP / small_gicp::RegistrationPCL) through
the C ABI."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


def _pair(n=3000, seed=0):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    tgt = small_cloud(n, seed)
    rel = synth.make_pose([0.2, -0.1, 0.03], synth.rot_xyz(0.01, -0.008, 0.03))
    src = orc.transform_points(np.linalg.inv(rel), small_cloud(n, seed + 50))
    return tgt, src, rel


def _rot_angle(Ra, Rb):
    from mrg_slam_amd import synth

    return synth.rotation_angle(Ra, Rb)


@pytest.mark.parametrize("eps", [0.1, 0.01])
def test_gicp_align_matches_oracle(eps):
    from mrg_slam_amd import GicpHip
    from oracle import oracle as orc

    tgt, src, rel = _pair()
    g = GicpHip(transformation_epsilon=eps)
    o = orc.FastGicp(transformation_epsilon=eps, num_threads=4)
    g.setInputTarget(tgt)
    o.setInputTarget(tgt)
    g.setInputSource(src)
    o.setInputSource(src)
    aligned = g.align(np.eye(4), want_aligned=True)
    o.align(np.eye(4))
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert g.hasConverged() == o.hasConverged()
    assert g.getFinalNumIteration() == o.getFinalNumIteration()
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4
    assert _rot_angle(Tg[:3, :3], To[:3, :3]) <= 1e-4
    np.testing.assert_allclose(g.getHessian(), o.getFinalHessian(), rtol=0, atol=1e-6 * np.abs(o.getFinalHessian()).max())
    np.testing.assert_array_equal(aligned, orc.transform_points(Tg, src))
    assert g.getFitnessScore() == pytest.approx(o.getFitnessScore(), rel=1e-3)
    assert np.linalg.norm(Tg[:3, 3] - rel[:3, 3]) < 0.1


@pytest.mark.parametrize("eps", [0.1, 0.01, 1e-4])
@pytest.mark.parametrize("guess_seed", [None, 3])
def test_small_gicp_align_matches_oracle(eps, guess_seed):
    """SMALL_GICP_HIP (the reference's YAML default, registrations.cpp:46-54): same decisions and result as the restated
    small_gicp optimiser; bar 1e-4 m / 1e-4 rad."""
    from mrg_slam_amd import SmallGicpHip, synth
    from oracle import oracle as orc

    tgt, src, rel = _pair()
    guess = np.eye(4) if guess_seed is None else synth.perturb_pose(rel, np.random.default_rng(guess_seed))
    g = SmallGicpHip(transformation_epsilon=eps)
    o = orc.SmallGicp(transformation_epsilon=eps, num_threads=1)
    g.setInputTarget(tgt)
    o.setInputTarget(tgt)
    g.setInputSource(src)
    o.setInputSource(src)
    aligned = g.align(guess, want_aligned=True)
    o.align(guess)
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert g.hasConverged() == o.hasConverged()
    assert g.getFinalNumIteration() == o.getFinalNumIteration()
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4
    assert _rot_angle(Tg[:3, :3], To[:3, :3]) <= 1e-4
    np.testing.assert_allclose(g.getHessian(), o.getFinalHessian(), rtol=0, atol=1e-6 * np.abs(o.getFinalHessian()).max())
    np.testing.assert_array_equal(aligned, orc.transform_points(Tg, src))
    assert np.linalg.norm(Tg[:3, 3] - rel[:3, 3]) < 0.1


@pytest.mark.parametrize("variant", ["fast", "small"])
def test_gicp_covariances_and_linearize_match_oracle(variant):
    """Kernel-level parity: regularised k-NN covariances and one update_correspondences + linearize pass."""
    from mrg_slam_amd import GicpHip, SmallGicpHip, synth
    from oracle import oracle as orc

    tgt, src, rel = _pair(2500, 7)
    g = (GicpHip if variant == "fast" else SmallGicpHip)(transformation_epsilon=0.01)
    o = (orc.FastGicp if variant == "fast" else orc.SmallGicp)(transformation_epsilon=0.01, num_threads=1)
    for r in (g, o):
        r.setInputTarget(tgt)
        r.setInputSource(src)
    T = synth.perturb_pose(rel, np.random.default_rng(2)).astype(np.float64)
    H, b, e, n = g.linearize(T)
    eo, Ho, bo, no = o.linearize(T)
    assert n == no and n > 0.5 * len(src)
    assert e == pytest.approx(eo, rel=1e-12)
    np.testing.assert_allclose(H, Ho, rtol=0, atol=1e-12 * np.abs(Ho).max())
    np.testing.assert_allclose(b, bo, rtol=0, atol=1e-12 * np.abs(bo).max() + 1e-9)
    for which in ("source", "target"):
        np.testing.assert_allclose(g.covariances(which), o.covariances(which), rtol=0, atol=1e-12)


def test_small_gicp_linearize_is_the_right_perturbation_of_the_left_one():
    """The two variants linearise the same cost: H_right = Ad^T H_left Ad, b_right = Ad^T b_left with the adjoint of T
    (rotation block first), and the error is the same number."""
    from mrg_slam_amd import GicpHip, SmallGicpHip, synth

    tgt, src, rel = _pair(2000, 4)
    T = synth.perturb_pose(rel, np.random.default_rng(1)).astype(np.float64)
    out = []
    for cls in (GicpHip, SmallGicpHip):
        g = cls(transformation_epsilon=0.01)
        g.setInputTarget(tgt)
        g.setInputSource(src)
        out.append(g.linearize(T))
    (Hl, bl, el, nl), (Hr, br, er, nr) = out
    R, t = T[:3, :3], T[:3, 3]
    skew = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ad = np.block([[R, np.zeros((3, 3))], [skew @ R, R]])
    assert nl == nr and el == er
    np.testing.assert_allclose(Hr, Ad.T @ Hl @ Ad, rtol=0, atol=1e-9 * np.abs(Hl).max())
    np.testing.assert_allclose(br, Ad.T @ bl, rtol=0, atol=1e-9 * np.abs(bl).max())


def test_small_gicp_batch_equals_single_registrations():
    from mrg_slam_amd import BatchMatcher, SmallGicpHip, synth
    from mrg_slam_amd._lib import SMALL_GICP_HIP
    from mrg_slam_amd.registration import default_params, result_matrix
    from oracle import oracle as orc

    tgt = small_cloud(4000, 210)
    rng = np.random.default_rng(10)
    prm = default_params(SMALL_GICP_HIP)
    prm.transformation_epsilon = 0.01
    bm = BatchMatcher(prm)
    t = bm.add_target(tgt)
    pairs = []
    for k in range(4):
        rel = synth.make_pose(rng.normal(0, 0.15, 3), synth.rot_xyz(*rng.normal(0, 0.015, 3)))
        src = orc.transform_points(np.linalg.inv(rel), tgt[: 2500 + 300 * k])
        guess = synth.perturb_pose(np.eye(4), rng)
        pairs.append((src, guess))
        bm.add_pair(t, src, guess)
    res = bm.align(fitness_max_range=float("inf"))
    for k, (src, guess) in enumerate(pairs):
        reg = SmallGicpHip(transformation_epsilon=0.01)
        reg.setInputTarget(tgt)
        reg.setInputSource(src)
        reg.align(guess)
        np.testing.assert_array_equal(result_matrix(res[k]), reg.getFinalTransformation())
        assert res[k]["converged"] == int(reg.hasConverged()) and res[k]["iterations"] == reg.getFinalNumIteration()
        assert res[k]["fitness"] == pytest.approx(reg.getFitnessScore(), rel=1e-12)


@pytest.mark.parametrize("eps,res", [(0.1, 1.0), (0.01, 1.0), (0.01, 0.5), (1e-4, 2.0)])
@pytest.mark.parametrize("guess_seed", [None, 5])
def test_vgicp_align_matches_oracle(eps, res, guess_seed):
    """VGICP_HIP (FAST_VGICP, registrations.cpp:76-84, and the reference's GPU slot FAST_VGICP_CUDA :65-75): same decisions and
    result as the restated fast_gicp::FastVGICP; bar 1e-4 m / 1e-4 rad."""
    from mrg_slam_amd import VgicpHip, synth
    from oracle import oracle as orc

    tgt, src, rel = _pair()
    guess = np.eye(4) if guess_seed is None else synth.perturb_pose(rel, np.random.default_rng(guess_seed))
    g = VgicpHip(resolution=res, transformation_epsilon=eps)
    o = orc.FastVgicp(resolution=res, transformation_epsilon=eps, num_threads=1)
    for r in (g, o):
        r.setInputTarget(tgt)
        r.setInputSource(src)
    aligned = g.align(guess, want_aligned=True)
    o.align(guess)
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert g.hasConverged() == o.hasConverged()
    assert g.getFinalNumIteration() == o.getFinalNumIteration()
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4
    assert _rot_angle(Tg[:3, :3], To[:3, :3]) <= 1e-4
    np.testing.assert_allclose(g.getHessian(), o.getFinalHessian(), rtol=0, atol=1e-6 * np.abs(o.getFinalHessian()).max())
    np.testing.assert_array_equal(aligned, orc.transform_points(Tg, src))
    # 3000 random points in 1 m voxels are a coarse target: only the easy start with a tight epsilon must also reach the truth
    if eps <= 0.01 and guess_seed is None and res <= 1.0:
        assert np.linalg.norm(Tg[:3, 3] - rel[:3, 3]) < 0.15


def test_vgicp_linearize_matches_oracle():
    """One update_correspondences + linearize pass of the voxelised cost: same correspondences (voxel of the transformed point,
    floor(x / res - 0.5)), weights sqrt(points per voxel), H / b / error to f64 rounding."""
    from mrg_slam_amd import VgicpHip, synth
    from oracle import oracle as orc

    tgt, src, rel = _pair(2500, 9)
    src = np.concatenate([src, np.float32([[500, 0, 0, 0], [np.nan, 0, 0, 0]])])  # a point without a voxel, a non-finite one
    for res in (1.0, 0.37):
        g = VgicpHip(resolution=res, transformation_epsilon=0.01)
        o = orc.FastVgicp(resolution=res, transformation_epsilon=0.01, num_threads=1)
        for r in (g, o):
            r.setInputTarget(tgt)
            r.setInputSource(src)
        T = synth.perturb_pose(rel, np.random.default_rng(3)).astype(np.float64)
        H, b, e, n = g.linearize(T)
        eo, Ho, bo, no = o.linearize(T)
        assert n == no and 0 < n < len(src)
        assert e == pytest.approx(eo, rel=1e-12)
        np.testing.assert_allclose(H, Ho, rtol=0, atol=1e-12 * np.abs(Ho).max())
        np.testing.assert_allclose(b, bo, rtol=0, atol=1e-12 * np.abs(bo).max() + 1e-9)


def test_vgicp_batch_equals_single_registrations():
    from mrg_slam_amd import BatchMatcher, VgicpHip, synth
    from mrg_slam_amd._lib import VGICP_HIP
    from mrg_slam_amd.registration import default_params, result_matrix
    from oracle import oracle as orc

    targets = [small_cloud(4000, 220), small_cloud(3500, 221)]
    rng = np.random.default_rng(12)
    prm = default_params(VGICP_HIP)
    prm.transformation_epsilon = 0.01
    prm.resolution = 1.0
    bm = BatchMatcher(prm)
    tids = [bm.add_target(t) for t in targets]
    pairs = []
    for k in range(5):
        ti = k % 2
        rel = synth.make_pose(rng.normal(0, 0.1, 3), synth.rot_xyz(*rng.normal(0, 0.01, 3)))
        src = orc.transform_points(np.linalg.inv(rel), targets[ti][: 2500 + 300 * k])
        guess = synth.perturb_pose(np.eye(4), rng)
        pairs.append((ti, src, guess))
        bm.add_pair(tids[ti], src, guess, key=900 + k)  # through the keyframe store as well
    res = bm.align(fitness_max_range=float("inf"))
    again = bm.align(fitness_max_range=float("inf"))
    for k, (ti, src, guess) in enumerate(pairs):
        reg = VgicpHip(resolution=1.0, transformation_epsilon=0.01)
        reg.setInputTarget(targets[ti])
        reg.setInputSource(src)
        reg.align(guess)
        np.testing.assert_array_equal(result_matrix(res[k]), reg.getFinalTransformation())
        np.testing.assert_array_equal(result_matrix(again[k]), reg.getFinalTransformation())
        assert res[k]["converged"] == int(reg.hasConverged()) and res[k]["iterations"] == reg.getFinalNumIteration()
        assert res[k]["fitness"] == pytest.approx(reg.getFitnessScore(), rel=1e-12)


@pytest.mark.parametrize("cls_name", ["GicpHip", "SmallGicpHip", "VgicpHip"])
def test_gicp_family_degenerate_inputs(cls_name):
    """Empty and non-finite clouds must not fault the GPU; results mirror the oracle's decisions."""
    import mrg_slam_amd
    from oracle import oracle as orc

    cls = getattr(mrg_slam_amd, cls_name)
    ocls = {"GicpHip": orc.FastGicp, "SmallGicpHip": orc.SmallGicp, "VgicpHip": orc.FastVgicp}[cls_name]
    tgt, src, _ = _pair(1500, 21)
    empty = np.zeros((0, 4), np.float32)
    nan_src = src.copy()
    nan_src[::3, 0] = np.nan
    far_src = src + np.float32([1000, 0, 0, 0])  # no correspondences at all
    for t, s_ in ((tgt, empty), (empty, src), (tgt, nan_src), (tgt, far_src), (tgt[:3], src[:5])):
        g, o = cls(transformation_epsilon=0.01), ocls(transformation_epsilon=0.01, num_threads=1)
        for r in (g, o):
            r.setInputTarget(t)
            r.setInputSource(s_)
            r.align(np.eye(4))
        Tg, To = g.getFinalTransformation().astype(np.float64), o.getFinalTransformation().astype(np.float64)
        assert g.hasConverged() == o.hasConverged(), (cls_name, len(t), len(s_))
        if np.all(np.isfinite(To)):
            assert np.linalg.norm(Tg[:3, 3] - To[:3, 3]) <= 1e-4 and _rot_angle(Tg[:3, :3], To[:3, :3]) <= 1e-4
        else:
            assert not np.all(np.isfinite(Tg))


def test_vgicp_rejects_a_voxel_map_that_does_not_fit():
    from mrg_slam_amd import VgicpHip
    from mrg_slam_amd._lib import MrgfeError

    tgt, src, _ = _pair(1500, 22)
    g = VgicpHip(resolution=1e-3)  # 20 m / 1 mm per axis: far more than 2^24 voxels
    g.setInputTarget(tgt)
    g.setInputSource(src)
    with pytest.raises(MrgfeError):
        g.align(np.eye(4))
    with pytest.raises(MrgfeError):
        VgicpHip(resolution=0.0)


@pytest.mark.parametrize("eps", [0.01, 1e-4, 1e-8])
@pytest.mark.parametrize("guess_seed", [None, 7])
def test_icp_align_matches_oracle(eps, guess_seed):
    """ICP_HIP (pcl::IterativeClosestPoint, registrations.cpp:85-92): same iteration count, convergence decision and result
    as the restated algorithm; bar 1e-4 m / 1e-4 rad."""
    from mrg_slam_amd import IcpHip, synth
    from oracle import oracle as orc

    tgt, src, rel = _pair()
    guess = np.eye(4) if guess_seed is None else synth.perturb_pose(rel, np.random.default_rng(guess_seed))
    g, o = IcpHip(transformation_epsilon=eps), orc.Icp(transformation_epsilon=eps)
    for r in (g, o):
        r.setInputTarget(tgt)
        r.setInputSource(src)
    aligned = g.align(guess, want_aligned=True)
    o.align(guess)
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert g.hasConverged() == o.hasConverged()
    assert g.getFinalNumIteration() == o.getFinalNumIteration()
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4
    assert _rot_angle(Tg[:3, :3], To[:3, :3]) <= 1e-4
    np.testing.assert_array_equal(aligned, orc.transform_points(Tg, src))
    assert g.getFitnessScore() == pytest.approx(o.getFitnessScore(), rel=1e-3, abs=1e-9)


@pytest.mark.parametrize("guess_seed", [None, 7])
def test_icp_with_reciprocal_correspondences_matches_oracle(guess_seed):
    """setUseReciprocalCorrespondences(true) (registrations.cpp:91): a pair counts only if the target point's nearest (transformed) source
    point is the query again — fewer correspondences, another trajectory; the same as the restated algorithm, and not the one-way result."""
    from mrg_slam_amd import IcpHip, select_registration_method, synth
    from oracle import oracle as orc

    tgt, src, rel = _pair()
    src = src[: len(src) * 2 // 3]  # different sizes: the reciprocal test prunes
    guess = np.eye(4) if guess_seed is None else synth.perturb_pose(rel, np.random.default_rng(guess_seed))
    g = select_registration_method({"registration_method": "ICP", "reg_use_reciprocal_correspondences": True, "reg_transformation_epsilon": 1e-4})
    assert type(g) is IcpHip
    o, plain = orc.Icp(transformation_epsilon=1e-4, use_reciprocal_correspondences=True), IcpHip(transformation_epsilon=1e-4)
    for r in (g, o, plain):
        r.setInputTarget(tgt)
        r.setInputSource(src)
        r.align(guess)
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert g.hasConverged() == o.hasConverged() and g.getFinalNumIteration() == o.getFinalNumIteration()
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4 and _rot_angle(Tg[:3, :3], To[:3, :3]) <= 1e-4
    assert not np.array_equal(Tg, plain.getFinalTransformation())


def test_icp_exact_copy_limits_and_degenerate_inputs():
    from mrg_slam_amd import IcpHip, synth
    from oracle import oracle as orc

    tgt = small_cloud(3000, 31)
    rel = synth.make_pose([0.15, -0.1, 0.02], synth.rot_xyz(0.01, -0.005, 0.02))
    src = orc.transform_points(np.linalg.inv(rel), tgt)  # an exact rigid copy: ICP must land on the motion
    g = IcpHip(transformation_epsilon=1e-8)
    g.setInputTarget(tgt)
    g.setInputSource(src)
    g.align(np.eye(4))
    T = g.getFinalTransformation().astype(np.float64)
    assert g.hasConverged() and np.linalg.norm(T[:3, 3] - rel[:3, 3]) < 1e-5 and synth.rotation_angle(T, rel) < 1e-5
    one = IcpHip(transformation_epsilon=1e-12, maximum_iterations=1)  # the iteration limit counts as converged (failure_after_max_iter_ = false)
    one.setInputTarget(tgt)
    one.setInputSource(src)
    one.align(np.eye(4))
    assert one.hasConverged() and one.getFinalNumIteration() == 1
    for t, s_ in ((tgt, np.zeros((0, 4), np.float32)), (np.zeros((0, 4), np.float32), src), (tgt, src + np.float32([1000, 0, 0, 0]))):
        r, o = IcpHip(), orc.Icp()
        for x in (r, o):
            x.setInputTarget(t)
            x.setInputSource(s_)
            x.align(np.eye(4))
        assert r.hasConverged() == o.hasConverged() and r.getFinalNumIteration() == o.getFinalNumIteration()
        np.testing.assert_allclose(r.getFinalTransformation(), o.getFinalTransformation(), atol=1e-5)


@pytest.mark.parametrize("variant", ["fast", "small"])
@pytest.mark.parametrize("max_corr", [0.3, 2.0, 1e3])
def test_correspondence_passes_equal_the_lane_group_search(street_pair_vlp16, variant, max_corr):
    """update_correspondences as the batched passes of getFitnessScore carrying the index (nn_nearest_batch: block / seed / sweep / pyramid walk)
    against one lane group per query (gicp_corr_kernel) and against the oracle: the same correspondences, so the same linearisation to the
    last bit, at poses near the truth and far from it (most queries then leave the block pass open), with a tight, the default and a huge
    max_correspondence_distance; ties at equal distance go to the lowest index either way (a cloud matched against itself with duplicates)."""
    from mrg_slam_amd import GicpHip, SmallGicpHip, synth
    from mrg_slam_amd._lib import lib
    from oracle import oracle as orc

    tgt, src, rel = street_pair_vlp16
    tgt = np.concatenate([tgt, tgt[::7]])  # duplicates: equal distances, different indices
    cls, ocls = (GicpHip, orc.FastGicp) if variant == "fast" else (SmallGicpHip, orc.SmallGicp)
    g = cls(transformation_epsilon=0.01, max_correspondence_distance=max_corr)
    o = ocls(transformation_epsilon=0.01, num_threads=8, max_correspondence_distance=max_corr)
    for r in (g, o):
        r.setInputTarget(tgt)
        r.setInputSource(src)
    rng = np.random.default_rng(5)
    poses = [rel, synth.perturb_pose(rel, rng), np.eye(4), synth.make_pose([6.0, -4.0, 0.5], synth.rot_xyz(0.02, 0.01, 0.4)), synth.make_pose([40.0, 25.0, 3.0], synth.rot_xyz(0, 0, 1.0))]
    try:
        for T in poses:
            T = np.asarray(T, dtype=np.float64)
            lib().mrgfe_dbg_set_gicp_corr_passes(0)
            H0, b0, e0, n0 = g.linearize(T)
            lib().mrgfe_dbg_set_gicp_corr_passes(2)
            H1, b1, e1, n1 = g.linearize(T)
            assert n1 == n0 and e1 == e0
            np.testing.assert_array_equal(H1, H0)
            np.testing.assert_array_equal(b1, b0)
            eo, Ho, bo, no = o.linearize(T)
            assert n1 == no
            assert e1 == pytest.approx(eo, rel=1e-12)
    finally:
        lib().mrgfe_dbg_set_gicp_corr_passes(1)
    # and a registration from start to end
    lib().mrgfe_dbg_set_gicp_corr_passes(0)
    try:
        g.align(synth.perturb_pose(rel, np.random.default_rng(9)))
        Ta, ia = g.getFinalTransformation().copy(), g.getFinalNumIteration()
    finally:
        lib().mrgfe_dbg_set_gicp_corr_passes(2)
    try:
        g.align(synth.perturb_pose(rel, np.random.default_rng(9)))
    finally:
        lib().mrgfe_dbg_set_gicp_corr_passes(1)
    np.testing.assert_array_equal(g.getFinalTransformation(), Ta)
    assert g.getFinalNumIteration() == ia


def test_batch_correspondence_passes_equal_the_lane_group_search(street_pair_vlp16):
    """The batched LM rounds with the correspondence passes forced on (mode 2; by default batches of >= 400k queries take them) and off: the same
    records, bit for bit, for candidates near their answer and one far from it."""
    from mrg_slam_amd import BatchMatcher, synth
    from mrg_slam_amd._lib import SMALL_GICP_HIP, lib
    from mrg_slam_amd.registration import default_params

    tgt, src, rel = street_pair_vlp16
    prm = default_params(SMALL_GICP_HIP)
    prm.transformation_epsilon = 0.01
    rng = np.random.default_rng(3)
    guesses = [synth.perturb_pose(rel, rng) for _ in range(5)] + [np.eye(4), synth.make_pose([5.0, 3.0, 0.2], synth.rot_xyz(0, 0, 0.3))]
    out = []
    try:
        for mode in (0, 2):
            lib().mrgfe_dbg_set_gicp_corr_passes(mode)
            bm = BatchMatcher(prm)
            t = bm.add_target(tgt)
            for k, gss in enumerate(guesses):
                bm.add_pair(t, src[: len(src) - 100 * k], gss)
            out.append(bm.align(fitness_max_range=float("inf")).copy())
    finally:
        lib().mrgfe_dbg_set_gicp_corr_passes(1)
    for f in ("T", "H", "fitness", "converged", "iterations"):
        np.testing.assert_array_equal(out[0][f], out[1][f], err_msg=f)


def test_batch_correspondence_passes_with_empty_clouds():
    """Empty source clouds and an empty target go through the correspondence passes like through the lane-group search: the same records."""
    from mrg_slam_amd import BatchMatcher, synth
    from mrg_slam_amd._lib import SMALL_GICP_HIP, lib
    from mrg_slam_amd.registration import default_params

    tgt, src, rel = _pair(1500, 31)
    empty = np.zeros((0, 4), dtype=np.float32)
    prm = default_params(SMALL_GICP_HIP)
    out = []
    try:
        for mode in (0, 2):
            lib().mrgfe_dbg_set_gicp_corr_passes(mode)
            bm = BatchMatcher(prm)
            t0, t1 = bm.add_target(tgt), bm.add_target(empty)
            bm.add_pair(t0, src, rel)
            bm.add_pair(t0, empty, np.eye(4))
            bm.add_pair(t1, src, np.eye(4))
            bm.add_pair(t0, src[:700], synth.perturb_pose(rel, np.random.default_rng(1)))
            out.append(bm.align(fitness_max_range=-1.0).copy())
    finally:
        lib().mrgfe_dbg_set_gicp_corr_passes(1)
    for f in ("T", "H", "converged", "iterations"):
        np.testing.assert_array_equal(out[0][f], out[1][f], err_msg=f)


@pytest.mark.parametrize("cls_name", ["GicpHip", "SmallGicpHip", "VgicpHip", "PclGicpHip", "NdtHip"])
@pytest.mark.parametrize("resident", [False, True])
def test_source_becomes_target_equals_set_input_target(street_pair_vlp16, cls_name, resident):
    """The odometry's keyframe update (scan_matching_odometry_component.cpp:326-339 -> :333: keyframe = the scan just aligned) through
    mrgfe_reg_source_becomes_target: the registration that took its source over as the target — with the covariances and the grid it made for it as a source —
    aligns the next frame bit for bit like one that was handed the same cloud through setInputTarget; also when no align came in between (nothing to take
    over), and with clouds it uploaded itself as well as with clouds resident in HBM."""
    import torch

    import mrg_slam_amd as M
    from mrg_slam_amd import synth

    cls = getattr(M, cls_name)
    a, b = street_pair_vlp16[0], street_pair_vlp16[1]
    c = (b + np.float32(0.01) * np.random.default_rng(3).normal(size=b.shape).astype(np.float32))[: len(b) - 17]
    guess = synth.make_pose([0.3, 0.1, 0.0], synth.rot_xyz(0.0, 0.0, 0.01))
    dev = [torch.from_numpy(x).to("cuda:0") for x in (a, b, c)] if resident else None

    def tgt(r, k, x):
        return r.setInputTargetDevice(dev[k].data_ptr(), len(x)) if resident else r.setInputTarget(x)

    def src(r, k, x):
        return r.setInputSourceDevice(dev[k].data_ptr(), len(x)) if resident else r.setInputSource(x)

    for align_first in (True, False):
        plain, promo = cls(transformation_epsilon=0.01), cls(transformation_epsilon=0.01)
        for r in (plain, promo):
            tgt(r, 0, a)
            src(r, 1, b)
            if align_first:
                r.align(guess)
        assert tgt(plain, 1, b) == 0
        assert promo.sourceBecomesTarget() == 0
        for r in (plain, promo):
            src(r, 2, c)
            r.align(guess)
        np.testing.assert_array_equal(promo.getFinalTransformation(), plain.getFinalTransformation())
        assert promo.hasConverged() == plain.hasConverged() and promo.getFinalNumIteration() == plain.getFinalNumIteration()
        assert promo.getFitnessScore() == plain.getFitnessScore()
        # and the promoted registration goes on like any other: a third keyframe the ordinary way
        for r in (plain, promo):
            tgt(r, 0, a)
            src(r, 1, b)
            r.align(guess)
        np.testing.assert_array_equal(promo.getFinalTransformation(), plain.getFinalTransformation())


def test_source_becomes_target_needs_a_source():
    from mrg_slam_amd import SmallGicpHip
    from mrg_slam_amd._lib import MrgfeError

    with pytest.raises(MrgfeError):
        SmallGicpHip().sourceBecomesTarget()


@pytest.mark.parametrize("cls_name", ["SmallGicpHip", "GicpHip", "NdtHip"])
def test_source_from_prefilter_equals_set_input_source_device(cls_name):
    """mrgfe_reg_set_source_from_prefilter: the source's search grid built inside the box the prefilter chain knows to enclose its output (no bounding-box pass of
    its own) answers like the grid over the tight box — same transformation, iterations and fitness as setInputSourceDevice of the same buffer; a pointer or
    count the prefilter did not produce, and a prefilter that ran on another context, fall back to the ordinary path."""
    import torch

    import mrg_slam_amd as M
    from mrg_slam_amd import Context, prefilter, prefilter_to_device, synth

    ctx = Context()
    cls = getattr(M, cls_name)
    scene = synth.street_scene()
    poses = synth.arc_trajectory(3)
    raw = [synth.synth_lidar(scene, poses[k], "VLP16", synth.BASE_SEED + k) for k in range(3)]
    kf = prefilter(raw[0], ctx=ctx)
    guess = synth.warm_guess(synth.rel_pose(poses[0], poses[1]), 3)
    buf = torch.empty((max(len(r) for r in raw) + 16, 4), dtype=torch.float32, device="cuda:0")
    other = torch.empty_like(buf)
    res = {}
    for how in ("plain", "from_prefilter", "stale_pointer", "other_context"):
        reg = cls(transformation_epsilon=0.01, ctx=ctx)
        reg.setInputTarget(kf)
        pf_ctx = Context() if how == "other_context" else ctx
        m = prefilter_to_device(raw[1], buf.data_ptr(), buf.shape[0], ctx=pf_ctx)
        pf_ctx.synchronize()
        if how == "plain":
            reg.setInputSourceDevice(buf.data_ptr(), m)
        elif how == "stale_pointer":  # the same points somewhere else: not what the prefilter left
            other[:m] = buf[:m]
            torch.cuda.synchronize()
            reg.setInputSourceFromPrefilter(other.data_ptr(), m)
        else:
            reg.setInputSourceFromPrefilter(buf.data_ptr(), m)
        reg.align(guess)
        res[how] = (reg.getFinalTransformation().copy(), reg.hasConverged(), reg.getFinalNumIteration(), reg.getFitnessScore())
        if how == "from_prefilter" and cls_name != "NdtHip":  # ... and the keyframe update takes the boxed grid over
            assert reg.sourceBecomesTarget() == 0
            m2 = prefilter_to_device(raw[2], other.data_ptr(), other.shape[0], ctx=ctx)
            reg.setInputSourceFromPrefilter(other.data_ptr(), m2)
            reg.align(synth.warm_guess(synth.rel_pose(poses[1], poses[2]), 4))
            chained = reg.getFinalTransformation().copy()
            ref = cls(transformation_epsilon=0.01, ctx=ctx)
            ref.setInputTargetDevice(buf.data_ptr(), m)
            ref.setInputSourceDevice(other.data_ptr(), m2)
            ref.align(synth.warm_guess(synth.rel_pose(poses[1], poses[2]), 4))
            np.testing.assert_array_equal(chained, ref.getFinalTransformation())
    for how in ("from_prefilter", "stale_pointer", "other_context"):
        np.testing.assert_array_equal(res[how][0], res["plain"][0], err_msg=how)
        assert res[how][1:] == res["plain"][1:], how
