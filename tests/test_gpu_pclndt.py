"""GPU: PCL_NDT_HIP — pcl::NormalDistributionsTransform (PCL 1.12), the class the reference's factory returns for registration_method "NDT"
and for every name it does not know (registrations.cpp:115-129) — against the CPU oracle (oracle/pcl_ndt.cpp) through the C ABI.

Bars: final transformation within 1e-4 m / 1e-4 rad of the oracle (north_star); per-evaluation sums to f64 rounding — every pair term is f64
on both sides, only the association (per-point factorisation) and the order of the additions differ; against the oracle's GPU-order mode
(same association, same tree) only the two exp implementations differ, by an ulp here and there."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu

TOL_T, TOL_R = 1e-4, 1e-4


def _pair(n=6000, seed=0, noise=0.01, m=None):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    tgt = small_cloud(n, seed)
    rel = synth.make_pose([0.25, -0.1, 0.03], synth.rot_xyz(0.01, -0.008, 0.03))
    src = orc.transform_points(np.linalg.inv(rel), tgt[: m or n])
    src[:, :3] += np.random.default_rng(seed + 1).normal(0, noise, (len(src), 3)).astype(np.float32)
    return tgt, src, rel


def _both(tgt, src, gpu_order=0, **kw):
    from mrg_slam_amd import PclNdtHip
    from oracle import oracle as orc

    g = PclNdtHip(**kw)
    o = orc.PclNdt(gpu_order=gpu_order, num_threads=4, **kw)
    assert g.setInputTarget(tgt) == 0 and o.setInputTarget(tgt) == 0
    g.setInputSource(src)
    o.setInputSource(src)
    return g, o


@pytest.mark.parametrize("res", [1.0, 0.6, 2.0])
@pytest.mark.parametrize("force_hash", ["0", "1"])
def test_single_evaluation_matches_oracle(res, force_hash, monkeypatch):
    from oracle import oracle as orc

    monkeypatch.setenv("MRGFE_FORCE_HASH", force_hash)
    tgt, src, _ = _pair(12000, seed=3, m=5000)
    g, o = _both(tgt, src, resolution=res)
    _, o2 = _both(tgt, src, gpu_order=1, resolution=res)
    p = np.array([0.2, -0.05, 0.01, 0.012, -0.006, 0.025])
    T = orc.pose_to_matrix(p)
    for mode in (0, 1, 2):
        gs, gg, gH = g.evaluate(T, p, mode)
        os_, og, oH = o.evaluate(T, p, mode)
        ts, tg, tH = o2.evaluate(T, p, mode)
        if mode != 2:
            assert abs(gs) > 1.0
            assert gs == pytest.approx(os_, rel=1e-12)
            np.testing.assert_allclose(gg, og, rtol=0, atol=5e-11 * np.abs(og).max())
            assert gs == pytest.approx(ts, rel=1e-14)
            np.testing.assert_allclose(gg, tg, rtol=0, atol=5e-11 * np.abs(tg).max())  # (the device's exp and glibc's differ by an ulp here and there, and
            # thin voxels — inverse covariance eigenvalues of 1e4 / m^2 at resolution 0.6 — make the gradient a difference of terms 1e4 times its size)
        if mode != 1:
            np.testing.assert_allclose(gH, oH, rtol=0, atol=5e-11 * np.abs(oH).max())
            np.testing.assert_allclose(gH, tH, rtol=0, atol=5e-11 * np.abs(tH).max())
            np.testing.assert_array_equal(gH, gH.T)  # the f64 items fill the upper triangle and mirror it


@pytest.mark.parametrize("eps", [0.1, 1e-4, 1e-6])
@pytest.mark.parametrize("guess_kind", ["identity", "warm", "far"])
def test_align_matches_oracle(eps, guess_kind):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    tgt, src, rel = _pair(9000, seed=5)
    g, o = _both(tgt, src, transformation_epsilon=eps, maximum_iterations=40)
    guess = {"identity": np.eye(4), "warm": synth.warm_guess(rel, 3), "far": synth.make_pose([0.8, 0.5, 0.1], synth.rot_xyz(0.02, 0.01, -0.08)) @ rel}[guess_kind]
    aligned = g.align(guess, want_aligned=True)
    o.align(guess)
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert (g.hasConverged(), g.getFinalNumIteration(), g.evals) == (o.hasConverged(), o.getFinalNumIteration(), o.evals)
    if eps == 0.1:
        assert g.getFinalNumIteration() == 1  # PCL's rule: |t_step|^2 <= 0.1 holds for any step the line search can return (quirks.h)
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= TOL_T and synth.rotation_angle(Tg, To) <= TOL_R
    # in fact: f64 terms on both sides, so the float pose matrices agree to the last bit unless a sum's rounding moved a line-search decision
    assert np.abs(Tg.astype(np.float64) - To).max() <= 1e-6
    assert g.getTransformationLikelihood() == pytest.approx(o.getTransformationLikelihood(), rel=1e-9)
    np.testing.assert_allclose(g.getHessian(), o.getHessian(), rtol=0, atol=1e-9 * np.abs(o.getHessian()).max())
    assert g.mean_neighbours == pytest.approx(o.mean_neighbours, rel=1e-12)
    np.testing.assert_array_equal(aligned, orc.transform_points(Tg, src))
    assert g.getFitnessScore() == pytest.approx(o.getFitnessScore(), rel=1e-9)


def test_host_and_device_control_and_launch_layouts_agree():
    from mrg_slam_amd import synth
    from mrg_slam_amd._lib import lib

    tgt, src, rel = _pair(8000, seed=11)
    guess = synth.warm_guess(rel, 1)
    out = []
    try:
        for host, fused in ((1, 1), (0, 1), (1, 0), (0, 0)):
            lib().mrgfe_dbg_set_host_control(host)
            lib().mrgfe_dbg_set_fused_launch(fused)
            g, _ = _both(tgt, src, transformation_epsilon=1e-5, maximum_iterations=30)
            g.align(guess)
            out.append((g.getFinalTransformation(), g.getHessian(), g.getFinalNumIteration(), g.evals))
    finally:
        lib().mrgfe_dbg_set_host_control(-1)
        lib().mrgfe_dbg_set_fused_launch(1)
    for T, H, it, ev in out[1:]:
        np.testing.assert_array_equal(T, out[0][0])
        np.testing.assert_array_equal(H, out[0][1])
        assert (it, ev) == out[0][2:]


def test_batch_equals_sequential_registrations_and_oracle():
    from mrg_slam_amd import BatchMatcher, PclNdtHip, synth
    from mrg_slam_amd._lib import PCL_NDT_HIP
    from mrg_slam_amd.registration import default_params, result_matrix
    from oracle import oracle as orc

    targets = [small_cloud(7000, 100), small_cloud(5000, 101)]
    rng = np.random.default_rng(5)
    pairs = []
    for k in range(7):
        ti = k % 2
        rel = synth.make_pose(rng.normal(0, 0.2, 3), synth.rot_xyz(*rng.normal(0, 0.02, 3)))
        src = orc.transform_points(np.linalg.inv(rel), targets[ti][: 2500 + 300 * k])
        pairs.append((ti, src, synth.perturb_pose(np.eye(4), rng)))
    prm = default_params(PCL_NDT_HIP)
    prm.transformation_epsilon, prm.maximum_iterations = 1e-5, 30
    bm = BatchMatcher(prm)
    tids = [bm.add_target(t) for t in targets]
    for ti, src, guess in pairs:
        bm.add_pair(tids[ti], src, guess)
    res = bm.align(fitness_max_range=float("inf"))
    for k, (ti, src, guess) in enumerate(pairs):
        reg = PclNdtHip(transformation_epsilon=1e-5, maximum_iterations=30)
        reg.setInputTarget(targets[ti])
        reg.setInputSource(src)
        reg.align(guess)
        np.testing.assert_array_equal(result_matrix(res[k]), reg.getFinalTransformation())
        assert (res[k]["converged"], res[k]["iterations"], res[k]["evaluations"]) == (int(reg.hasConverged()), reg.getFinalNumIteration(), reg.evals)
        np.testing.assert_array_equal(res[k]["H"].reshape(6, 6), reg.getHessian())
        assert res[k]["fitness"] == pytest.approx(reg.getFitnessScore(), rel=1e-12)
        o = orc.PclNdt(transformation_epsilon=1e-5, maximum_iterations=30)
        o.setInputTarget(targets[ti])
        o.setInputSource(src)
        o.align(guess)
        To = o.getFinalTransformation()
        assert np.linalg.norm(result_matrix(res[k])[:3, 3].astype(np.float64) - To[:3, 3]) <= TOL_T and synth.rotation_angle(result_matrix(res[k]), To) <= TOL_R
        assert res[k]["iterations"] == o.getFinalNumIteration()
    pts, nbrs = bm.pair_counts()
    ms, launches, nbytes = bm.kernel_stats()
    assert ms > 0 and launches > 0 and nbytes == pytest.approx(pts * (16 + 27 * 8) + nbrs * 112, rel=1e-12)


def test_rejected_leaf_answers_the_radius_search():
    """a voxel of eight identical points: rejected by the eigenvalue check, but its centroid is in the kd-tree (oracle/quirks.h) — the pair scores
    -d1 and pulls nowhere; NDT_HIP's KDTREE search inherits it, its DIRECT searches do not"""
    from mrg_slam_amd import NdtHip, PclNdtHip
    from oracle import oracle as orc

    tgt = np.concatenate([small_cloud(2500, 4), np.tile(np.array([[30.25, 30.25, 10.25, 0.5]], np.float32), (8, 1))])
    src = np.array([[30.3, 30.2, 10.3, 0.0]], np.float32)
    expect = orc.PclNdt()
    expect.setInputTarget(tgt)
    expect.setInputSource(src)
    s0, g0, _ = expect.evaluate(np.eye(4), np.zeros(6), 0)
    assert s0 > 1.0 and (g0 == 0).all()
    for reg, want in ((PclNdtHip(), s0), (NdtHip(search="KDTREE"), s0), (NdtHip(search="DIRECT7"), 0.0)):
        assert reg.setInputTarget(tgt) == 0
        reg.setInputSource(src)
        s, g, _ = reg.evaluate(np.eye(4), np.zeros(6), 0)
        assert s == pytest.approx(want, rel=1e-6, abs=0) and (g == 0).all()
        npts = reg.leaves()[1]
        assert (npts == -1).sum() >= 1  # the read-out still says "rejected"


def test_exactly_planar_voxels_follow_pcl_112s_eigenvalue_tolerance():
    """pcl::VoxelGridCovariance of PCL >= 1.11 invalidates a leaf only when one of its two smaller eigenvalues is below -dummy_precision() (1e-12);
    ndt_omp's fork tests `< 0`.  A voxel of exactly coplanar points has a smallest eigenvalue of ~ -1e-18 .. +1e-18 by rounding: PCL's NDT inflates and keeps every
    such voxel, pclomp's drops the negative half.  A cloud of exactly planar patches at many offsets: PCL_NDT_HIP's leaves equal the oracle's PclNdt leaves (counts incl.
    the -1 marks, inverse covariances), no planar voxel is marked invalid there, and NDT_HIP on the same cloud — the pclomp rule, equal to ITS oracle — marks some."""
    from mrg_slam_amd import NdtHip, PclNdtHip
    from oracle import oracle as orc

    rng = np.random.default_rng(12)
    pts = []
    for k in range(400):  # patches of 12 points, each exactly in a plane z = const, x = const or y = const of its own voxel
        c = np.floor(rng.uniform(-20, 20, 3)) + 0.5
        p = c + rng.uniform(-0.45, 0.45, (12, 3))
        p[:, k % 3] = np.float32(c[k % 3] + rng.uniform(-0.3, 0.3))
        pts.append(p)
    tgt = np.concatenate(pts).astype(np.float32)
    tgt = np.concatenate([tgt, np.zeros((len(tgt), 1), np.float32)], axis=1)
    g, o = PclNdtHip(), orc.PclNdt()
    assert g.setInputTarget(tgt) == 0 and o.setInputTarget(tgt) == 0
    kg, ng, mg, ig = g.leaves()
    ko, no, _, mo, io, _ = o.leaves()
    np.testing.assert_array_equal(kg, ko)
    np.testing.assert_array_equal(ng, no)
    np.testing.assert_array_equal(mg, mo)
    np.testing.assert_array_equal(ig, io)
    assert (ng >= 6).sum() >= 300 and (ng == -1).sum() == 0  # every planar voxel is kept (inflated to 1 % of its largest eigenvalue)
    g2, o2 = NdtHip(), orc.Ndt()
    assert g2.setInputTarget(tgt) == 0 and o2.setInputTarget(tgt) == 0
    n2, no2 = g2.leaves()[1], o2.leaves()[1]
    np.testing.assert_array_equal(n2, no2)
    assert (n2 == -1).sum() >= 10  # pclomp's `< 0`: the patches whose smallest eigenvalue rounded to the negative side are dropped


def test_degenerate_inputs_behave_like_the_reference():
    from mrg_slam_amd import PclNdtHip, _lib, synth

    g = PclNdtHip()
    assert g.setInputTarget(np.zeros((0, 4), np.float32)) == _lib.ERR_EMPTY
    g.setInputSource(small_cloud(100))
    g.align(np.eye(4))
    assert not g.hasConverged()
    # no voxel reaches 6 points: zero gradient, zero step -> PCL >= 1.11.1: converged_ = (delta_norm == 0) = true, 0 iterations, final = guess
    sparse = small_cloud(40, extent=(200, 200, 50))
    g2 = PclNdtHip()
    assert g2.setInputTarget(sparse) == 0
    g2.setInputSource(sparse)
    guess = synth.make_pose([0.3, 0, 0], np.eye(3))
    g2.align(guess)
    assert g2.hasConverged() and g2.getFinalNumIteration() == 0
    np.testing.assert_array_equal(g2.getFinalTransformation(), guess.astype(np.float32))
    far = small_cloud(100)
    far[0, 0] = 1e6
    assert PclNdtHip(resolution=0.01).setInputTarget(far) == _lib.ERR_OVERFLOW


def test_factory_name_ndt_runs_pcls_class(street_pair_vlp16):
    """select_registration_method({"registration_method": "NDT"}) with the YAML's epsilon 0.1 == the oracle's pcl::NormalDistributionsTransform on a
    street scan pair: one Newton iteration, the same pose"""
    from mrg_slam_amd import PclNdtHip, prefilter, select_registration_method, synth
    from oracle import oracle as orc

    tgt, src, rel = street_pair_vlp16
    ft, fs = prefilter(tgt), prefilter(src)
    for name in ("NDT", "some unknown method"):
        reg = select_registration_method({"registration_method": name, "reg_transformation_epsilon": 0.1, "reg_maximum_iterations": 64, "reg_resolution": 1.0,
                                          "reg_nn_search_method": "DIRECT7", "reg_num_threads": 8})
        assert type(reg) is PclNdtHip
        o = orc.PclNdt(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64)
        assert reg.setInputTarget(ft) == 0 and o.setInputTarget(ft) == 0
        reg.setInputSource(fs)
        o.setInputSource(fs)
        guess = synth.warm_guess(rel, 0)
        reg.align(guess)
        o.align(guess)
        Tg, To = reg.getFinalTransformation(), o.getFinalTransformation()
        assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= TOL_T and synth.rotation_angle(Tg, To) <= TOL_R
        assert (reg.hasConverged(), reg.getFinalNumIteration(), reg.evals) == (True, 1, o.evals)


@pytest.mark.parametrize("eps", [0.1, 1e-5])
def test_full_size_pair_matches_oracle(eps):
    """BASELINE config[1]'s size: ~130k points per scan"""
    from mrg_slam_amd import PclNdtHip, distance_filter, synth
    from oracle import oracle as orc

    scene = synth.street_scene()
    tgt, src, rel = synth.scan_pair(0, "VLP64", scene)
    ft, fs = distance_filter(tgt), distance_filter(src)
    assert len(fs) > 120000
    guess = synth.warm_guess(rel, 0)
    g = PclNdtHip(transformation_epsilon=eps, maximum_iterations=30)
    o = orc.PclNdt(transformation_epsilon=eps, maximum_iterations=30)
    assert g.setInputTarget(ft) == 0 and o.setInputTarget(ft) == 0
    g.setInputSource(fs)
    o.setInputSource(fs)
    g.align(guess)
    o.align(guess)
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert (g.hasConverged(), g.getFinalNumIteration(), g.evals) == (o.hasConverged(), o.getFinalNumIteration(), o.evals)
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3]) <= TOL_T and synth.rotation_angle(Tg, To) <= TOL_R
    assert g.getFitnessScore() == pytest.approx(o.getFitnessScore(), rel=1e-9)
    perm = np.random.default_rng(1).permutation(len(fs))  # property: the order of the source only reorders f64 additions
    g.setInputSource(fs[perm])
    g.align(guess)
    Tp = g.getFinalTransformation()
    assert np.linalg.norm(Tp[:3, 3].astype(np.float64) - Tg[:3, 3]) <= 1e-6 and synth.rotation_angle(Tp, Tg) <= 1e-6
