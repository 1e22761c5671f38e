"""CPU: the product's NDT optimiser (mrg_slam_amd/csrc/ndt_ctl.h — the Newton / More-Thuente state machine that
ndt_reduce_kernel steps on the device and NdtController steps on the host) driven by hand through mrgfe_dbg_ctl_*, with the
CPU ORACLE supplying every derivative evaluation it asks for.  It must request exactly the evaluations the oracle's own
computeTransformation performs and end at the same transform: control flow, line search, 6x6 solve and pose arithmetic of the
product are checked here without a GPU (the derivative kernels are checked against the oracle in tests/test_gpu_ndt.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import small_cloud

_fp, _dp = C.POINTER(C.c_float), C.POINTER(C.c_double)


from oracle.replay import drive as _drive  # noqa: E402  (shared with the soak and bench.py's parity legs)


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("seed,eps,search,res", [(1, 0.1, "DIRECT7", 1.0), (2, 0.01, "DIRECT7", 1.0), (3, 0.001, "DIRECT7", 2.0), (4, 0.01, "DIRECT1", 1.0),
                                                 (5, 0.1, "KDTREE", 1.5), (6, 0.01, "DIRECT7", 0.5), (7, 0.1, "DIRECT7", 1.0), (8, 0.01, "DIRECT26", 1.0)])
def test_state_machine_follows_the_oracle(seed, eps, search, res, split, monkeypatch):
    """split: the batch engine's variant — first trial of a line search without its Hessian, fetched by a second pass at the same
    pose only when the trial is accepted (csrc/ndt_ctl.h); must end exactly where the single-pass flow ends"""
    from mrg_slam_amd import synth

    if split:
        monkeypatch.setenv("MRGFE_DBG_CTL_SPLIT", "1")
    else:
        monkeypatch.delenv("MRGFE_DBG_CTL_SPLIT", raising=False)
    from mrg_slam_amd._lib import NDT_HIP, SEARCH
    from mrg_slam_amd.registration import default_params
    from oracle import oracle as orc

    rng = np.random.default_rng(seed)
    tgt = small_cloud(4000, seed)
    rel = synth.make_pose(rng.normal(0, 0.3, 3), synth.rot_xyz(*rng.normal(0, 0.03, 3)))
    src = orc.transform_points(np.linalg.inv(rel), tgt[:3000])
    guess = synth.perturb_pose(rel if seed % 3 else np.eye(4), rng)
    o = orc.Ndt(resolution=res, transformation_epsilon=eps, maximum_iterations=64, num_threads=2, search=search)
    assert o.setInputTarget(tgt) == 0
    o.setInputSource(src)
    o.align(guess)
    p = default_params(NDT_HIP)
    p.resolution, p.transformation_epsilon, p.maximum_iterations, p.nn_search_method = res, eps, 64, SEARCH[search]
    d = orc.Ndt(resolution=res, transformation_epsilon=eps, maximum_iterations=64, num_threads=2, search=search)  # the evaluator
    assert d.setInputTarget(tgt) == 0
    d.setInputSource(src)
    T, conv, it, ev, modes = _drive(d, p, guess, len(src))
    To = o.getFinalTransformation()
    assert conv == o.hasConverged() and it == o.getFinalNumIteration()
    assert ev == o.evals  # the cached repeats of a clamped trial are counted like the reference's recomputations
    assert np.linalg.norm(T[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-6 and synth.rotation_angle(T, To) <= 1e-6
    assert modes[0] == 0 and set(modes) <= {0, 1, 2}
    if split:  # every full pass after the first is a Hessian fetch right behind the score+gradient pass of an accepted first trial
        assert all(modes[k - 1] == 1 for k in range(1, len(modes)) if modes[k] == 0)


def test_degenerate_starts():
    from mrg_slam_amd._lib import NDT_HIP, check, lib
    from mrg_slam_amd.registration import default_params

    p = default_params(NDT_HIP)
    g = np.eye(4, dtype=np.float32)
    h = C.c_void_p()
    check(lib().mrgfe_dbg_ctl_create(C.byref(p), g.ctypes.data_as(_fp), 0, C.byref(h)))  # empty source: nothing to evaluate
    assert lib().mrgfe_dbg_ctl_request(h, None, None, None) == 0
    lib().mrgfe_dbg_ctl_destroy(h)
    # a NaN Hessian ends the alignment unconverged (the reference: delta_p_norm != delta_p_norm -> converged_ = false)
    check(lib().mrgfe_dbg_ctl_create(C.byref(p), g.ctypes.data_as(_fp), 100, C.byref(h)))
    assert lib().mrgfe_dbg_ctl_request(h, None, None, None) == 1
    H = np.full((6, 6), np.nan)
    check(lib().mrgfe_dbg_ctl_result(h, 1.0, np.ones(6).ctypes.data_as(_dp), H.ctypes.data_as(_dp), 0.0))
    T, conv = np.empty((4, 4), dtype=np.float32), C.c_int(1)
    assert lib().mrgfe_dbg_ctl_request(h, None, None, None) == 0
    check(lib().mrgfe_dbg_ctl_final(h, T.ctypes.data_as(_fp), C.byref(conv), None, None))
    assert conv.value == 0
    # an all-zero gradient: zero step, converged (norm == 0)
    lib().mrgfe_dbg_ctl_destroy(h)
    check(lib().mrgfe_dbg_ctl_create(C.byref(p), g.ctypes.data_as(_fp), 100, C.byref(h)))
    check(lib().mrgfe_dbg_ctl_result(h, 1.0, np.zeros(6).ctypes.data_as(_dp), np.eye(6).ctypes.data_as(_dp), 0.0))
    assert lib().mrgfe_dbg_ctl_request(h, None, None, None) == 0
    check(lib().mrgfe_dbg_ctl_final(h, T.ctypes.data_as(_fp), C.byref(conv), None, None))
    assert conv.value == 1
    lib().mrgfe_dbg_ctl_destroy(h)


def test_float_sine_cosine_are_the_c_librarys():
    """The pose matrices of the reference come from Eigen::AngleAxisf, i.e. sinf / cosf of the C library, which are not correctly
    rounded: the optimiser restates glibc's algorithm (csrc/ndt_ctl.h, same source on host and device) and must return the C
    library's floats bit for bit — a strided sweep over every exponent from 2^-31 to 120, both signs, plus the special values."""
    import ctypes.util

    from mrg_slam_amd._lib import lib

    libm = C.CDLL(ctypes.util.find_library("m"))
    lo, hi = np.float32(2.0 ** -31).view(np.uint32), np.float32(120.0).view(np.uint32)
    bits = np.arange(int(lo), int(hi), 127, dtype=np.uint32)
    x = np.concatenate([bits.view(np.float32), -bits.view(np.float32), np.array([0.0, -0.0, 1e-40, 119.99999, 3.1415927, 1.5707964, 0.7853982, 0.78539824], np.float32)])
    s, c = np.empty_like(x), np.empty_like(x)
    lib().mrgfe_dbg_sincosf(x.ctypes.data_as(_fp), len(x), s.ctypes.data_as(_fp), c.ctypes.data_as(_fp))
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    # the C library one call at a time is slow from Python: check a random sample of the sweep plus the special values
    rng = np.random.default_rng(0)
    pick = np.concatenate([rng.choice(len(x) - 8, 200000, replace=False), np.arange(len(x) - 8, len(x))])
    ref_s = np.array([libm.sinf(float(v)) for v in x[pick]], dtype=np.float32)
    ref_c = np.array([libm.cosf(float(v)) for v in x[pick]], dtype=np.float32)
    assert (s[pick].view(np.uint32) == ref_s.view(np.uint32)).all()
    assert (c[pick].view(np.uint32) == ref_c.view(np.uint32)).all()
    # beyond the polynomial's range and for non-finite input the double routine takes over
    big = np.array([120.0, 1e6, np.inf, np.nan], np.float32)
    sb, cb = np.empty_like(big), np.empty_like(big)
    lib().mrgfe_dbg_sincosf(big.ctypes.data_as(_fp), 4, sb.ctypes.data_as(_fp), cb.ctypes.data_as(_fp))
    assert abs(sb[0] - np.sin(120.0)) < 1e-7 and abs(cb[1] - np.cos(1e6)) < 1e-7 and np.isnan(sb[2:]).all() and np.isnan(cb[2:]).all()


def test_reference_order_mode_reproduces_the_oracles_double_trajectory():
    """mrgfe_dbg_set_ndt_reference_order(1): the optimiser's Newton solve becomes Eigen's two-sided JacobiSVD restated operation for operation (csrc/ndt_ctl.h
    jacobi2_solve6) instead of the LU fast path / the one-sided SVD — those give the same step to ~1e-16, which a run to the iteration limit amplifies.  Fed
    with the reference-order oracle's evaluations (what the reference-order kernels deliver bit for bit, tests/test_gpu_ndt_reforder.py), the state machine
    must then reproduce the oracle's whole alignment in DOUBLE: the last pose vector, not only the float transformation."""
    from mrg_slam_amd._lib import NDT_HIP, SEARCH, lib
    from mrg_slam_amd.registration import default_params
    from oracle import oracle as orc
    from oracle.replay import soak_scene

    assert lib().mrgfe_dbg_set_ndt_reference_order(1) == 1
    try:
        rng = np.random.default_rng(20261004)
        iterations = 0
        for c in range(24):
            tgt, src, guess, eps = soak_scene(rng)
            res, search = float(rng.choice([0.5, 1.0, 1.5, 2.0])), str(rng.choice(["DIRECT7", "DIRECT1", "DIRECT26", "KDTREE"]))
            kw = dict(resolution=res, num_threads=8, transformation_epsilon=eps, maximum_iterations=64, search=search)
            o, d = orc.Ndt(**kw), orc.Ndt(**kw)
            for x in (o, d):
                x.setInputTarget(tgt)
                x.setInputSource(src)
            o.align(guess)
            prm = default_params(NDT_HIP)
            prm.transformation_epsilon, prm.maximum_iterations, prm.resolution, prm.nn_search_method = eps, 64, res, SEARCH[search]
            T, conv, it, ev, _ = _drive(d, prm, guess, len(src))
            np.testing.assert_array_equal(T, o.getFinalTransformation())
            assert (conv, it, ev) == (o.hasConverged(), o.getFinalNumIteration(), o.evals)
            np.testing.assert_array_equal(d.last_pose(), o.last_pose())
            iterations += it
        assert iterations > 150
    finally:
        assert lib().mrgfe_dbg_set_ndt_reference_order(0) == 0
