"""CPU: the oracle's restatement of pcl::GeneralizedIterativeClosestPoint / pclomp::GICP (oracle/pcl_gicp.cpp, oracle/bfgs.h;
/root/reference/src/mrg_slam/registrations.cpp:93-114) — what can be pinned without the upstream libraries: the covariance formula against a
numpy restatement, the gradient against finite differences, recovery of a known motion, the termination rules."""
import numpy as np
import pytest

from conftest import small_cloud
from mrg_slam_amd import synth
from oracle import oracle as orc


def _pair(n=4000, seed=5, nsrc=3300):
    tgt = small_cloud(n, seed)
    rel = synth.make_pose([0.22, -0.13, 0.04], synth.rot_xyz(0.015, -0.01, 0.035))
    src = orc.transform_points(np.linalg.inv(rel), tgt[:nsrc])
    return tgt, src, rel


def test_covariances_are_pcls_formula():
    """computeCovariances: raw second moments of the float coordinates (float products) summed in double, / k, minus mean mean^T; singular values
    replaced by (1, 1, 1e-3) along the eigenvectors."""
    tgt = small_cloud(600, 3)
    g = orc.PclGicp(correspondence_randomness=12)
    g.setInputTarget(tgt)
    g.setInputSource(tgt[:50])
    C = g.covariances("target")
    idx, _ = orc.knn(tgt, tgt, 12)
    for i in (0, 17, 311, 599):
        nb = tgt[idx[i], :3]
        mean = nb.astype(np.float64).sum(0) / 12
        raw = np.zeros((3, 3))
        for r in range(3):
            for c in range(r + 1):
                raw[r, c] = raw[c, r] = (nb[:, r] * nb[:, c]).astype(np.float64).sum() / 12 - mean[r] * mean[c]  # float products, like PCL
        w, E = np.linalg.eigh(raw)
        order = np.argsort(-np.abs(w))
        exp = sum((1.0 if k < 2 else 1e-3) * np.outer(E[:, order[k]], E[:, order[k]]) for k in range(3))
        np.testing.assert_allclose(C[i], exp, atol=1e-9)
        assert np.linalg.eigvalsh(C[i]) == pytest.approx([1e-3, 1.0, 1.0], abs=1e-9)


def test_gradient_matches_finite_differences_and_motion_is_recovered():
    tgt, src, rel = _pair()
    g = orc.PclGicp(transformation_epsilon=1e-5, num_threads=4)
    g.setInputTarget(tgt)
    g.setInputSource(src)
    x0 = np.array([0.1, -0.05, 0.02, 0.01, -0.02, 0.03])
    f0, grad, n = g.evaluate(np.eye(4), x0)
    assert n > 0.9 * len(src) and f0 > 0
    for k in range(6):
        h = 2e-4
        xp, xm = x0.copy(), x0.copy()
        xp[k] += h
        xm[k] -= h
        num = (g.evaluate(np.eye(4), xp)[0] - g.evaluate(np.eye(4), xm)[0]) / (2 * h)
        assert grad[k] == pytest.approx(num, rel=5e-3, abs=5e-3 * np.abs(grad).max())  # the cost is evaluated through float transforms
    for omp in (False, True):
        o = orc.PclGicp(transformation_epsilon=1e-5, omp=omp, num_threads=4)
        o.setInputTarget(tgt)
        o.setInputSource(src)
        o.align(np.eye(4))
        T = o.getFinalTransformation().astype(np.float64)
        assert o.hasConverged() and np.linalg.norm(T[:3, 3] - rel[:3, 3]) < 2e-3 and synth.rotation_angle(T, rel) < 1e-3
        assert o.inner_steps >= o.getFinalNumIteration() and o.evals > o.inner_steps
        assert o.getFitnessScore() < 1e-5


def test_gpu_order_mode_adds_the_same_terms():
    """PclGicp(gpu_order=True) — the diagnostic mode the GPU soak replays against — changes the order of the f64 additions of the cost sums and
    nothing else: f and the gradient agree with the reference-order evaluation to rounding, registrations to the bar."""
    tgt, src, rel = _pair()
    a, b = orc.PclGicp(transformation_epsilon=1e-3, num_threads=1), orc.PclGicp(transformation_epsilon=1e-3, num_threads=1, gpu_order=True)
    for g in (a, b):
        g.setInputTarget(tgt)
        g.setInputSource(src)
    x0 = np.array([0.05, -0.02, 0.01, 0.004, -0.01, 0.02])
    fa, ga, na = a.evaluate(np.eye(4), x0)
    fb, gb, nb = b.evaluate(np.eye(4), x0)
    assert na == nb and fa == pytest.approx(fb, rel=1e-13)
    np.testing.assert_allclose(ga, gb, rtol=0, atol=1e-12 * np.abs(ga).max())
    for g in (a, b):
        g.align(np.eye(4))
    Ta, Tb = a.getFinalTransformation().astype(np.float64), b.getFinalTransformation().astype(np.float64)
    assert np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]) < 5e-3 and synth.rotation_angle(Ta, Tb) < 5e-3


def test_termination_rules():
    tgt, src, rel = _pair(2500, 9, 2000)
    one = orc.PclGicp(transformation_epsilon=1e-12, maximum_iterations=1)  # nr_iterations_ >= max_iterations_ counts as converged
    one.setInputTarget(tgt)
    one.setInputSource(src)
    one.align(np.eye(4))
    assert one.hasConverged() and one.getFinalNumIteration() == 1
    far = orc.PclGicp(max_correspondence_distance=0.5)  # fewer than four correspondences: NotEnoughPointsException ends the loop, converged_ stays false
    far.setInputTarget(tgt)
    far.setInputSource(src + np.float32([100, 0, 0, 0]))
    far.align(np.eye(4))
    assert not far.hasConverged() and far.getFinalNumIteration() == 0
    np.testing.assert_array_equal(far.getFinalTransformation(), np.eye(4, dtype=np.float32))
    loose = orc.PclGicp(transformation_epsilon=10.0, rotation_epsilon=10.0)  # delta < 1 after the first outer iteration
    loose.setInputTarget(tgt)
    loose.setInputSource(src)
    loose.align(np.eye(4))
    assert loose.hasConverged() and loose.getFinalNumIteration() == 1
    capped = orc.PclGicp(max_optimizer_iterations=1)  # one BFGS step per outer iteration: still converges, in more outer iterations
    full = orc.PclGicp()
    for r in (capped, full):
        r.setInputTarget(tgt)
        r.setInputSource(src)
        r.align(np.eye(4))
    assert capped.hasConverged() and capped.getFinalNumIteration() >= full.getFinalNumIteration()
    # the guess is applied to the source once and composed at the end: final = previous_transformation_ * guess
    g1 = orc.PclGicp(transformation_epsilon=1e-5)
    g1.setInputTarget(tgt)
    g1.setInputSource(src)
    g1.align(synth.perturb_pose(rel, np.random.default_rng(1)))
    T = g1.getFinalTransformation().astype(np.float64)
    assert np.linalg.norm(T[:3, 3] - rel[:3, 3]) < 2e-3
