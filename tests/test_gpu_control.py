"""GPU: the optimiser stepped on the device (ndt_reduce_kernel<true>, state in HBM, no host round trip) against the same state
machine stepped by the host (one synchronisation per round) and against the CPU oracle.  Same source (csrc/ndt_ctl.h) on both
sides; the device evaluates the float sine / cosine of the pose matrices in double and rounds, the host calls sinf / cosf."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture()
def control():
    from mrg_slam_amd._lib import lib

    def set_mode(m):
        lib().mrgfe_dbg_set_host_control(m)

    yield set_mode
    lib().mrgfe_dbg_set_host_control(-1)


@pytest.fixture()
def fused():
    from mrg_slam_amd._lib import lib

    before = lib().mrgfe_dbg_set_fused_launch(-1)
    yield lambda m: lib().mrgfe_dbg_set_fused_launch(m)
    lib().mrgfe_dbg_set_fused_launch(before)


def _scenes(n_pairs, seed):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_pairs):
        n = int(rng.integers(2000, 7000))
        tgt = small_cloud(n, int(rng.integers(1 << 30)), extent=(float(rng.uniform(15, 40)), float(rng.uniform(10, 30)), float(rng.uniform(2, 6))))
        rel = synth.make_pose(rng.normal(0, 0.3, 3), synth.rot_xyz(*rng.normal(0, 0.03, 3)))
        src = orc.transform_points(np.linalg.inv(rel), tgt[: int(n * rng.uniform(0.5, 1.0))])
        guess = synth.perturb_pose(rel if rng.random() < 0.7 else np.eye(4), rng)
        out.append((tgt, src, guess))
    return out


@pytest.mark.parametrize("eps,search", [(0.1, "DIRECT7"), (0.01, "DIRECT7"), (0.01, "DIRECT1"), (0.01, "KDTREE")])
def test_device_control_equals_host_control_and_oracle(control, eps, search):
    from mrg_slam_amd import BatchMatcher, synth
    from mrg_slam_amd.registration import result_matrix
    from oracle import oracle as orc

    scenes = _scenes(24, 17)
    res = {}
    for mode in (1, 0):
        control(mode)
        bm = BatchMatcher(transformation_epsilon=eps, search=search)
        for tgt, src, guess in scenes:
            t = bm.add_target(tgt)
            bm.add_pair(t, src, guess)
        res[mode] = bm.align(float("inf"))
    h, d = res[1], res[0]
    assert (h["converged"] == d["converged"]).all() and (h["iterations"] == d["iterations"]).all() and (h["evaluations"] == d["evaluations"]).all()
    for k in range(len(scenes)):
        Th, Td = result_matrix(h[k]), result_matrix(d[k])
        assert np.linalg.norm(Th[:3, 3].astype(np.float64) - Td[:3, 3]) <= 1e-5 and synth.rotation_angle(Th, Td) <= 1e-5
        np.testing.assert_allclose(h[k]["H"], d[k]["H"], rtol=1e-6, atol=1e-6 * np.abs(h[k]["H"]).max())
    assert np.mean([np.array_equal(h[k]["T"], d[k]["T"]) for k in range(len(scenes))]) >= 0.9  # and nearly always bit for bit
    for k, (tgt, src, guess) in enumerate(scenes):
        o = orc.Ndt(transformation_epsilon=eps, num_threads=8, search=search)
        o.setInputTarget(tgt)
        o.setInputSource(src)
        o.align(guess)
        if o.hasConverged() and o.getFinalNumIteration() < 60:
            Td, To = result_matrix(d[k]), o.getFinalTransformation()
            assert np.linalg.norm(Td[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4 and synth.rotation_angle(Td, To) <= 1e-4
            assert bool(d[k]["converged"]) and int(d[k]["iterations"]) == o.getFinalNumIteration()


def test_single_registration_both_ways(control, street_pair_vlp16):
    from mrg_slam_amd import NdtHip, synth

    tgt, src, rel = street_pair_vlp16
    out = {}
    for mode in (1, 0):
        control(mode)
        g = NdtHip(transformation_epsilon=0.01)
        g.setInputTarget(tgt)
        g.setInputSource(src)
        g.align(synth.warm_guess(rel, 3))
        out[mode] = (g.getFinalTransformation(), g.hasConverged(), g.getFinalNumIteration(), g.evals, g.getHessian())
    assert out[0][1:4] == out[1][1:4]
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-6
    np.testing.assert_allclose(out[0][4], out[1][4], rtol=1e-9)


def test_ragged_batch_with_dead_pairs_on_the_device(control):
    """pairs that finish at once (empty source, target without a grid) next to running ones; more pairs than one plan chunk"""
    from mrg_slam_amd import BatchMatcher

    control(0)
    scenes = _scenes(5, 3)
    bm = BatchMatcher(transformation_epsilon=0.01)
    t_empty = bm.add_target(np.zeros((0, 4), np.float32))
    ids = [bm.add_target(tgt) for tgt, _, _ in scenes]
    for rep in range(60):  # 300 pairs > 256: two chunks of the plan kernel's scan
        for k, (_, src, guess) in enumerate(scenes):
            bm.add_pair(ids[k], src, guess)
    bm.add_pair(ids[0], np.zeros((0, 4), np.float32), np.eye(4))
    bm.add_pair(t_empty, scenes[0][1], np.eye(4))
    r = bm.align()
    assert len(r) == 302
    for k in range(5, 300):
        assert np.array_equal(r[k]["T"], r[k % 5]["T"]) and r[k]["iterations"] == r[k % 5]["iterations"]
    assert not r[300]["converged"] and not r[301]["converged"]
    control(1)
    bm2 = BatchMatcher(transformation_epsilon=0.01)
    for tgt, src, guess in scenes:
        bm2.add_pair(bm2.add_target(tgt), src, guess)
    r2 = bm2.align()
    assert (r2["iterations"] == r[:5]["iterations"]).all()


def test_controller_math_device_equals_host():
    """csrc/ndt_ctl.h compiled for the device against the same source compiled for the host: pose matrices (float sine / cosine:
    sinf on the host, double sine rounded to float on the device), angle derivative tables (double sine / cosine of two C
    libraries) and the 6x6 Hestenes SVD solve."""
    import ctypes as C

    from mrg_slam_amd._lib import check, default_context, lib

    rng = np.random.default_rng(0)
    n = 20000
    cases = np.zeros((n, 48))
    cases[:, :3] = rng.normal(0, 5, (n, 3))
    cases[:, 3:6] = rng.normal(0, 0.5, (n, 3)) * (rng.random((n, 1)) < 0.8) + rng.normal(0, 1e-5, (n, 3))
    A = rng.normal(0, 1, (n, 6, 6))
    A = A @ A.transpose(0, 2, 1) * rng.uniform(1e-2, 1e4, (n, 1, 1)) + rng.normal(0, 1e-3, (n, 6, 6))  # nearly symmetric, like the float Hessians
    A[::7] *= np.array([1, 1, 1, 1e-7, 1, 1])  # ill-conditioned ones
    A[::501, :, 2] = A[::501, :, 1]  # singular ones
    cases[:, 6:42] = A.reshape(n, 36)
    cases[:, 42:] = rng.normal(0, 10, (n, 6))
    out = {}
    _dp, _fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    for dev in (0, 1, 2):
        M, tab, x = np.empty((n, 16), np.float32), np.empty((n, 69)), np.empty((n, 6))
        check(lib().mrgfe_dbg_ctl_math(default_context()._h, cases.ctypes.data_as(_dp), n, dev, M.ctypes.data_as(_fp), tab.ctypes.data_as(_dp), x.ctypes.data_as(_dp)))
        out[dev] = (M, tab, x)
    Mh, th, xh = out[0]
    Md, td, xd = out[1]
    same_M = (Mh == Md).all(axis=1).mean()
    print(f"pose matrices identical: {same_M:.5f}; tables identical: {(th == td).all(axis=1).mean():.5f}; solves identical: {(xh == xd).all(axis=1).mean():.5f}")
    assert np.abs(Mh - Md).max() <= 1.2e-7 and same_M >= 0.999  # at most an ulp of a rotation entry, and rarely
    np.testing.assert_allclose(td, th, rtol=0, atol=4e-16)
    ok = np.isfinite(xh).all(axis=1)
    assert (np.isfinite(xd).all(axis=1) == ok).all()
    scale = np.abs(xh[ok]).max(axis=1, keepdims=True) + 1e-300
    assert (np.abs(xd[ok] - xh[ok]) / scale).max() <= 1e-6  # ill-conditioned systems amplify the last-bit differences of sqrt / division
    assert ((xd[ok] == xh[ok]).all(axis=1)).mean() >= 0.5
    # the wavefront form of the solve (what the device controller runs) against the single-lane form, both on the device
    xw = out[2][2]
    assert (np.isfinite(xw).all(axis=1) == ok).all()
    assert (xw[ok] == xd[ok]).all()


@pytest.mark.parametrize("search", ["DIRECT7", "DIRECT1", "KDTREE"])
@pytest.mark.parametrize("host", [0, 1])
def test_fused_launch_equals_one_launch_per_variant(control, fused, search, host):
    """All three kernel variants of a round in ONE launch (items interleaved) against one launch per variant: an item's partial
    record does not depend on the launch it is computed in, so every result is the same bit for bit (device- and host-stepped)."""
    from mrg_slam_amd import BatchMatcher

    scenes = _scenes(40, 23)
    control(host)
    res = {}
    for f in (0, 1):
        assert fused(f) == f
        bm = BatchMatcher(transformation_epsilon=0.01, search=search)
        for tgt, src, guess in scenes:
            bm.add_pair(bm.add_target(tgt), src, guess)
        res[f] = bm.align(float("inf"))
        ms, launches, nbytes = bm.kernel_stats(-1)
        assert launches > 0 and nbytes > 0
    a, b = res[0], res[1]
    for field in ("T", "H", "fitness", "converged", "iterations", "evaluations", "trans_probability"):
        assert np.array_equal(a[field], b[field]), field
