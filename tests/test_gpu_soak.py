"""GPU: randomised parity soak (round 1's tests/soak_parity.py as a test; fixed seed): 320 small scenes over all five methods,
resolutions, search methods, epsilons and guesses, GPU against the CPU oracle.

For every NDT scene the alignment is held against TWO oracle runs:
  reference order  the oracle as it restates the reference (per-point sums added in index order): the parity bar 1e-4 m / 1e-4 rad
                   applies wherever the optimisation settles;
  GPU order        the product's optimiser state machine stepped on the CPU (mrgfe_dbg_ctl_*) with the oracle evaluating every
                   request in the HIP kernels' summation ORDER (oracle/ndt.cpp, gpu_order_ppt): same per-pair float terms, same
                   order => the same doubles, so the whole trajectory must repeat bit for bit.  A scene where the GPU differs from
                   the reference-order run but equals this one differs by summation order alone (1e-16 per sum), amplified by an
                   optimisation that does not settle — not by an arithmetic difference.
"""
import ctypes as C

import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


def _scene(rng):
    from mrg_slam_amd import synth
    from oracle import oracle as orc

    n = int(rng.integers(1500, 9000))
    tgt = small_cloud(n, int(rng.integers(1 << 30)), extent=(float(rng.uniform(15, 45)), float(rng.uniform(10, 40)), float(rng.uniform(2, 6))))
    rel = synth.make_pose(rng.normal(0, 0.3, 3), synth.rot_xyz(*rng.normal(0, 0.03, 3)))
    src = orc.transform_points(np.linalg.inv(rel), tgt[: int(n * rng.uniform(0.5, 1.0))])
    src[:, :3] += rng.normal(0, 0.01, (len(src), 3)).astype(np.float32)
    guess = synth.perturb_pose(rel if rng.random() < 0.7 else np.eye(4), rng)
    return tgt, src, guess, float(rng.choice([0.1, 0.01, 0.001]))


def test_soak_all_methods():
    from mrg_slam_amd import GicpHip, IcpHip, NdtHip, SmallGicpHip, VgicpHip, synth
    from mrg_slam_amd._lib import NDT_HIP, SEARCH, lib
    from mrg_slam_amd.registration import default_params
    from oracle import oracle as orc
    from test_controller_cpu import _drive

    lib().mrgfe_dbg_set_host_control(-1)
    rng = np.random.default_rng(11)
    cases = 320
    stats = {"ndt": 0, "ndt_exact_ref": 0, "ndt_exact_gpu_order": 0, "ndt_settled": 0, "other": 0, "other_exact": 0}
    worst_settled = 0.0
    unexplained, over_bar = [], []
    for c in range(cases):
        tgt, src, guess, eps = _scene(rng)
        kind = rng.random()
        if kind < 0.75:
            res = float(rng.choice([0.5, 1.0, 1.5, 2.0]))
            search = str(rng.choice(["DIRECT7", "DIRECT1", "DIRECT26", "KDTREE"]))
            g = NdtHip(resolution=res, transformation_epsilon=eps, maximum_iterations=64, search=search)
            o = orc.Ndt(resolution=res, transformation_epsilon=eps, maximum_iterations=64, num_threads=8, search=search)
            tag = f"case {c}: NDT res={res} {search} eps={eps}"
        else:
            sub = rng.random()
            if sub < 0.2:
                g, o, tag = IcpHip(transformation_epsilon=eps * 1e-3), orc.Icp(transformation_epsilon=eps * 1e-3), f"case {c}: ICP"
            elif sub < 0.45:
                vres = float(rng.choice([0.5, 1.0, 2.0]))
                g, o, tag = VgicpHip(resolution=vres, transformation_epsilon=eps), orc.FastVgicp(resolution=vres, transformation_epsilon=eps, num_threads=1), f"case {c}: VGICP"
            elif sub < 0.72:
                g, o, tag = GicpHip(transformation_epsilon=eps), orc.FastGicp(transformation_epsilon=eps, num_threads=8), f"case {c}: GICP"
            else:
                g, o, tag = SmallGicpHip(transformation_epsilon=eps), orc.SmallGicp(transformation_epsilon=eps, num_threads=8), f"case {c}: SMALL_GICP"
        g.setInputTarget(tgt)
        o.setInputTarget(tgt)
        g.setInputSource(src)
        o.setInputSource(src)
        g.align(guess)
        o.align(guess)
        Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
        dt, dr = float(np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3])), synth.rotation_angle(Tg, To)
        exact = np.array_equal(Tg, To)
        if not isinstance(g, NdtHip):
            stats["other"] += 1
            stats["other_exact"] += exact
            assert g.hasConverged() == o.hasConverged(), tag
            assert dt <= 1e-4 and dr <= 1e-4, (tag, dt, dr)
            continue
        stats["ndt"] += 1
        stats["ndt_exact_ref"] += exact
        # the same alignment replayed on the CPU in the GPU's summation order
        p = default_params(NDT_HIP)
        p.resolution, p.transformation_epsilon, p.maximum_iterations, p.nn_search_method = res, eps, 64, SEARCH[search]
        d = orc.Ndt(resolution=res, transformation_epsilon=eps, maximum_iterations=64, num_threads=1, search=search, gpu_order_ppt=1)
        d.setInputTarget(tgt)
        d.setInputSource(src)
        Tr, conv_r, it_r, ev_r, _ = _drive(d, p, guess, len(src))
        same_as_replay = np.array_equal(Tg, Tr) and bool(g.hasConverged()) == conv_r and g.getFinalNumIteration() == it_r and g.evals == ev_r
        stats["ndt_exact_gpu_order"] += same_as_replay
        settled = o.hasConverged() and g.hasConverged() and o.getFinalNumIteration() <= 30 and g.getFinalNumIteration() <= 30
        if settled:
            stats["ndt_settled"] += 1
            worst_settled = max(worst_settled, dt, dr)
            if dt > 1e-4 or dr > 1e-4:
                over_bar.append((tag, dt, dr, g.getFinalNumIteration(), o.getFinalNumIteration(), same_as_replay))
        if not exact and not same_as_replay:
            drt = float(np.linalg.norm(Tg[:3, 3].astype(np.float64) - Tr[:3, 3]))
            unexplained.append((tag, dt, drt, g.getFinalNumIteration(), it_r))
    print(stats, "worst settled difference", worst_settled)
    for u in unexplained:
        print("not bit-identical to either oracle run:", u)
    for u in over_bar:
        print("settled but over the bar:", u)
    assert stats["other_exact"] == stats["other"], "ICP / GICP / VGICP results are bit-identical to the oracle's"
    # every GPU trajectory repeats bit for bit on the CPU in GPU order, up to the rare scene where the f64 Hessian pass sees the two
    # C libraries' exp differ in the last bit (it then still ends within 1e-6)
    assert stats["ndt_exact_gpu_order"] >= 0.97 * stats["ndt"]
    for tag, dt, drt, it_g, it_r in unexplained:
        assert drt <= 1e-6, (tag, drt)
    # reference order: parity wherever the optimisation settles; a settled scene over the bar must be order noise (equal to the replay)
    assert stats["ndt_exact_ref"] >= 0.9 * stats["ndt"]
    for tag, dt, dr, it_g, it_o, same_as_replay in over_bar:
        assert same_as_replay, (tag, dt, dr)
    assert len(over_bar) <= 3


def test_soak_round3_methods():
    """The same soak for the methods added in round 3: pcl::GICP (both stopping rules of its BFGS) and ICP with reciprocal correspondences,
    60 random scenes.  ICP: the bar of 1e-4 m / 1e-4 rad.  pcl::GICP: its BFGS line search amplifies the order of the f64 cost sums (the
    reference adds the terms one after the other, the kernels in a tree: 1e-16 relative) into millimetres on about one scene in eight — so
    every result must be bit-identical to the oracle run with its sums in the kernels' order (PclGicp(gpu_order=True): same decisions, same
    transform), most of them (>= 80 %) also to the reference-order oracle, and all within 5 mm of it with the same flags and iteration counts."""
    from mrg_slam_amd import IcpHip, PclGicpHip, synth
    from oracle import oracle as orc

    rng = np.random.default_rng(29)
    stats = {"gicp": 0, "gicp_exact_ref": 0, "gicp_exact_gpu_order": 0, "icp": 0, "icp_exact": 0}
    worst = 0.0
    for c in range(60):
        tgt, src, guess, eps = _scene(rng)
        kind = rng.random()
        replay = None
        if kind < 0.7:
            omp = kind >= 0.4
            g, o, tag = PclGicpHip(transformation_epsilon=eps, omp=omp), orc.PclGicp(transformation_epsilon=eps, omp=omp, num_threads=8), f"case {c}: PCL GICP{'_OMP' if omp else ''} eps={eps}"
            replay = orc.PclGicp(transformation_epsilon=eps, omp=omp, num_threads=1, gpu_order=True)
        else:
            g, o, tag = (IcpHip(transformation_epsilon=eps * 1e-3, use_reciprocal_correspondences=True),
                         orc.Icp(transformation_epsilon=eps * 1e-3, use_reciprocal_correspondences=True), f"case {c}: ICP reciprocal")
        for r in (g, o) + ((replay,) if replay else ()):
            r.setInputTarget(tgt)
            r.setInputSource(src)
            r.align(guess)
        Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
        dt, dr = float(np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3])), synth.rotation_angle(Tg, To)
        assert g.hasConverged() == o.hasConverged(), tag
        assert g.getFinalNumIteration() == o.getFinalNumIteration(), (tag, g.getFinalNumIteration(), o.getFinalNumIteration())
        if replay is None:
            stats["icp"] += 1
            stats["icp_exact"] += np.array_equal(Tg, To)
            assert dt <= 1e-4 and dr <= 1e-4, (tag, dt, dr)
            continue
        stats["gicp"] += 1
        stats["gicp_exact_ref"] += np.array_equal(Tg, To)
        same = np.array_equal(Tg, replay.getFinalTransformation()) and g.getFinalNumIteration() == replay.getFinalNumIteration()
        stats["gicp_exact_gpu_order"] += same
        assert same, (tag, "differs from the oracle in the kernels' summation order")
        assert dt <= 5e-3 and dr <= 5e-3, (tag, dt, dr)
        worst = max(worst, dt, dr)
    print(stats, "worst PCL GICP difference to the reference-order oracle", worst)
    assert stats["gicp_exact_ref"] >= 0.8 * stats["gicp"]
