"""GPU: PCL_GICP_HIP (pcl::GeneralizedIterativeClosestPoint, /root/reference/src/mrg_slam/registrations.cpp:93-103; pclomp::GICP :104-114)
against the restated algorithm (oracle/pcl_gicp.cpp): covariances, the cost the inner BFGS minimises, whole alignments."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


def _pair(n=5000, seed=5, nsrc=4200):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    tgt = small_cloud(n, seed)
    rel = synth.make_pose([0.22, -0.13, 0.04], synth.rot_xyz(0.015, -0.01, 0.035))
    src = orc.transform_points(np.linalg.inv(rel), tgt[:nsrc])
    return tgt, src, rel


def test_covariances_and_cost_match_oracle():
    """PCL's covariance formula (raw float moments, singular values (1, 1, 1e-3)) to 1e-12; f and its gradient over the correspondences of
    the search loop to f64 summation order."""
    from mrg_slam_amd import PclGicpHip
    from oracle import oracle as orc

    tgt, src, rel = _pair()
    g, o = PclGicpHip(), orc.PclGicp(num_threads=4)
    for r in (g, o):
        r.setInputTarget(tgt)
        r.setInputSource(src)
    for which in ("source", "target"):  # (the 3 x 3 eigen solver's trigonometry differs in the last bit between the device's and the host's math library)
        np.testing.assert_allclose(g.covariances(which), o.covariances(which), rtol=0, atol=1e-12)
    for T, x in ((np.eye(4), np.zeros(6)), (rel, np.array([0.05, -0.02, 0.01, 0.004, -0.003, 0.01])), (np.eye(4), np.array([0.2, -0.1, 0.05, 0.01, -0.01, 0.03]))):
        fg, gg, ng = g.evaluate(T, x)
        fo, go, no = o.evaluate(T, x)
        assert ng == no and ng > 1000
        assert fg == pytest.approx(fo, rel=1e-12)
        np.testing.assert_allclose(gg, go, rtol=0, atol=1e-11 * max(1.0, np.abs(go).max()))


@pytest.mark.parametrize("omp", [False, True])
@pytest.mark.parametrize("eps", [0.01, 1e-4])
@pytest.mark.parametrize("guess_seed", [None, 7])
def test_pcl_gicp_align_matches_oracle(omp, eps, guess_seed):
    from mrg_slam_amd import PclGicpHip, select_registration_method, synth
    from oracle import oracle as orc

    tgt, src, rel = _pair()
    guess = np.eye(4) if guess_seed is None else synth.perturb_pose(rel, np.random.default_rng(guess_seed))
    g = select_registration_method({"registration_method": "GICP_OMP" if omp else "GICP", "reg_transformation_epsilon": eps, "reg_use_reciprocal_correspondences": True})
    assert type(g) is PclGicpHip
    o = orc.PclGicp(transformation_epsilon=eps, omp=omp, num_threads=4, sum_threads=8 if omp else 1)  # GICP_OMP: the sums of 8 OpenMP threads (the product's default)
    for r in (g, o):
        r.setInputTarget(tgt)
        r.setInputSource(src)
    aligned = g.align(guess, want_aligned=True)
    o.align(guess)
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    assert g.hasConverged() == o.hasConverged() and g.getFinalNumIteration() == o.getFinalNumIteration()
    np.testing.assert_array_equal(Tg, To)  # both formulations add their cost terms in the reference's order: the same BFGS trajectory, bit for bit
    assert np.linalg.norm(Tg[:3, 3].astype(np.float64) - rel[:3, 3]) < 5e-3  # and it is the motion
    np.testing.assert_array_equal(aligned, orc.transform_points(Tg, src))
    assert g.getFitnessScore() == pytest.approx(o.getFitnessScore(), rel=1e-3, abs=1e-9)


@pytest.mark.parametrize("threads", [2, 5, 8, 16])
def test_pclomp_sums_follow_the_stated_thread_count(threads):
    """pclomp::GICP adds per-thread partial sums over static chunks of the correspondence list (libgomp: the first m mod T threads take one more) and
    the partials in thread order: PCL_GICP_OMP_HIP with num_threads = T equals the oracle's T-thread accumulation bit for bit — cost, gradient, and
    whole alignments; another T gives other last bits"""
    from mrg_slam_amd import PclGicpHip, synth
    from oracle import oracle as orc

    tgt, src, rel = _pair(6000, 21, 5003)  # (a correspondence count that no thread count divides)
    g, o = PclGicpHip(omp=True, num_threads=threads, transformation_epsilon=1e-4), orc.PclGicp(omp=True, num_threads=4, sum_threads=threads, transformation_epsilon=1e-4)
    other = orc.PclGicp(omp=True, num_threads=4, sum_threads=1, transformation_epsilon=1e-4)
    for r in (g, o, other):
        r.setInputTarget(tgt)
        r.setInputSource(src)
    differs = False
    for T, x in ((np.eye(4), np.zeros(6)), (rel, np.array([0.05, -0.02, 0.01, 0.004, -0.003, 0.01])), (np.eye(4), np.array([0.2, -0.1, 0.05, 0.01, -0.01, 0.03]))):
        fg, gg, ng = g.evaluate(T, x)
        fo, go, no = o.evaluate(T, x)
        f1, g1, _ = other.evaluate(T, x)
        # the same additions in the same order (a term itself differs in its last bit now and then — the Mahalanobis matrices come from the device's
        # and the host's own 3 x 3 inverses —: one ulp of a sum in a few hundred evaluations, so "equal" here is 1e-14 relative, not bitwise)
        assert ng == no and fg == pytest.approx(fo, rel=1e-14)
        np.testing.assert_allclose(gg, go, rtol=0, atol=1e-14 * np.abs(go).max())
        differs = differs or fg != f1 or (gg != g1).any()
    assert differs  # (the serial chain rounds differently somewhere in three evaluations of 5000 terms)
    for guess in (np.eye(4), synth.perturb_pose(rel, np.random.default_rng(3))):
        g.align(guess)
        o.align(guess)
        np.testing.assert_array_equal(g.getFinalTransformation(), o.getFinalTransformation())
        assert (g.hasConverged(), g.getFinalNumIteration()) == (o.hasConverged(), o.getFinalNumIteration())
    # fewer correspondences than threads: empty chunks
    tiny_t, tiny_s = tgt[:40], src[:9]
    g2, o2 = PclGicpHip(omp=True, num_threads=threads), orc.PclGicp(omp=True, sum_threads=threads)
    for r in (g2, o2):
        r.setInputTarget(tiny_t)
        r.setInputSource(tiny_s)
    fg, gg, ng = g2.evaluate(np.eye(4), np.zeros(6))
    fo, go, no = o2.evaluate(np.eye(4), np.zeros(6))
    assert ng == no and fg == pytest.approx(fo, rel=1e-14) and np.allclose(gg, go, rtol=0, atol=1e-14 * max(1e-300, np.abs(go).max()))


def test_pcl_gicp_limits_and_degenerate_inputs():
    from mrg_slam_amd import BatchMatcher, PclGicpHip
    from mrg_slam_amd._lib import PCL_GICP_HIP, MrgfeError
    from mrg_slam_amd.registration import default_params
    from oracle import oracle as orc

    tgt, src, rel = _pair(3000, 9, 2500)
    one, oo = PclGicpHip(transformation_epsilon=1e-12, maximum_iterations=1), orc.PclGicp(transformation_epsilon=1e-12, maximum_iterations=1, num_threads=4)
    few, of = PclGicpHip(max_optimizer_iterations=2), orc.PclGicp(max_optimizer_iterations=2, num_threads=4)
    for g, o in ((one, oo), (few, of)):
        for r in (g, o):
            r.setInputTarget(tgt)
            r.setInputSource(src)
            r.align(np.eye(4))
        assert g.hasConverged() == o.hasConverged() and g.getFinalNumIteration() == o.getFinalNumIteration()
        np.testing.assert_allclose(g.getFinalTransformation(), o.getFinalTransformation(), atol=1e-5)
    assert one.hasConverged() and one.getFinalNumIteration() == 1  # the iteration limit counts as converged
    far, ofar = PclGicpHip(max_correspondence_distance=0.5), orc.PclGicp(max_correspondence_distance=0.5, num_threads=2)
    for r in (far, ofar):
        r.setInputTarget(tgt)
        r.setInputSource(src + np.float32([100, 0, 0, 0]))  # fewer than four correspondences: NotEnoughPointsException ends the loop unconverged
        r.align(np.eye(4))
    assert not far.hasConverged() and not ofar.hasConverged()
    np.testing.assert_array_equal(far.getFinalTransformation(), np.eye(4, dtype=np.float32))
    with pytest.raises(MrgfeError):
        BatchMatcher(default_params(PCL_GICP_HIP))


def test_pclomp_with_one_thread_is_the_serial_chain_and_more_than_sixteen_is_refused():
    """`reg_num_threads: 1` with GICP_OMP used to fall through to the block tree, which matches no reference.  One OpenMP thread adds ALL terms into its
    one partial in correspondence order (then 0 + p_0 = p_0): the serial chain.  So PCL_GICP_OMP_HIP with num_threads = 1 must equal the oracle's one-thread
    accumulation; thread counts beyond the sixteen chains the kernel has are an error, not a silent clamp."""
    from mrg_slam_amd import MrgfeError, PclGicpHip, synth
    from oracle import oracle as orc

    tgt, src, rel = _pair(6000, 22, 5003)
    g, o = PclGicpHip(omp=True, num_threads=1, transformation_epsilon=1e-4), orc.PclGicp(omp=True, num_threads=4, sum_threads=1, transformation_epsilon=1e-4)
    eight = orc.PclGicp(omp=True, num_threads=4, sum_threads=8, transformation_epsilon=1e-4)
    for r in (g, o, eight):
        r.setInputTarget(tgt)
        r.setInputSource(src)
    differs = False
    for T, x in ((np.eye(4), np.zeros(6)), (rel, np.array([0.05, -0.02, 0.01, 0.004, -0.003, 0.01])), (np.eye(4), np.array([0.2, -0.1, 0.05, 0.01, -0.01, 0.03]))):
        fg, gg, ng = g.evaluate(T, x)
        fo, go, no = o.evaluate(T, x)
        f8, g8, _ = eight.evaluate(T, x)
        assert ng == no and fg == pytest.approx(fo, rel=1e-14)
        np.testing.assert_allclose(gg, go, rtol=0, atol=1e-14 * np.abs(go).max())
        differs = differs or fg != f8 or (gg != g8).any()
    assert differs  # (eight partials round differently somewhere)
    for guess in (np.eye(4), synth.perturb_pose(rel, np.random.default_rng(4))):
        g.align(guess)
        o.align(guess)
        np.testing.assert_array_equal(g.getFinalTransformation(), o.getFinalTransformation())
        assert (g.hasConverged(), g.getFinalNumIteration()) == (o.hasConverged(), o.getFinalNumIteration())
    with pytest.raises(MrgfeError, match="1..16 threads"):
        PclGicpHip(omp=True, num_threads=17)
