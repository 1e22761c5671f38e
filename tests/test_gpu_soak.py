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
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_soak_all_methods():
    from oracle.replay import ndt_soak

    st = ndt_soak(320, 11)
    print({k: v for k, v in st.items() if k not in ("over_bar", "unexplained")})
    for u in st["unexplained"]:
        print("not bit-identical to either oracle run:", u)
    for u in st["over_bar"]:
        print("over the bar:", u)
    assert st["other_exact"] == st["other"] and st["other_flag_mismatch"] == 0, "ICP / GICP / VGICP results are bit-identical to the oracle's"
    # EVERY NDT scene equals the GPU-order replay: bit for bit, or — where the f64 Hessian pass met an argument on which the device's exp and
    # glibc's differ in the last bit — within 1e-6 m / rad of it after the same number of iterations.  No share, no allowance: a count that must
    # come out as the number of scenes.
    near = [u for u in st["unexplained"] if u["dt_vs_replay_m"] <= 1e-6 and u["iterations_hip"] == u["iterations_replay"]]
    assert len(near) == len(st["unexplained"]), [u for u in st["unexplained"] if u not in near]
    # (`unexplained` lists the scenes that equal NEITHER oracle run bit for bit; one that equals the reference-order run needs no explanation)
    # reference order: a scene over the bar — settled or not — must be summation-order noise, i.e. equal to the replay ...
    for u in st["over_bar"]:
        assert u["equal_to_gpu_order_replay"] or any(v["case"] == u["case"] for v in near), u
    # ... AND the counts are capped in absolute terms against the reference-order oracle, which does not go through the product's own optimiser (ADVICE r5:
    # the replay steps csrc/ndt_ctl.h too, so a defect in its LU fast path would be shared).  Measured on 7,505 scenes (profiles/r05_soak.json): 0.6 % over
    # the bar, 2 of them settled, worst settled 2.4 cm; on this draw of 240 NDT scenes: at most 5 over the bar, at most 1 settled, and at least 90 % bit-identical.
    assert st["ndt"] >= 200
    assert st["ndt_over_bar"] <= 5 and st["ndt_settled_over_bar"] <= 1, st["over_bar"]
    assert st["ndt_worst_settled"] <= 0.05, st["over_bar"]
    assert st["ndt_exact_ref"] >= 0.9 * st["ndt"]
    assert st["ndt_flag_or_iteration_mismatch"] <= st["ndt_over_bar"]


def test_soak_lu_fast_path_against_the_svd_solve():
    """the LU fast path of the Newton solve (ndt_ctl.h lu_solve6) replaced round 4's one-sided Jacobi SVD, and the GPU-order replay shares it.
    The same 120 scenes in two child processes — default, and MRGFE_NEWTON_SVD=1 (read once per process) — each held against the reference-order oracle,
    which knows neither: flags, iteration counts and the counts of scenes over the bar must agree between the two solves."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import json, sys; sys.path.insert(0, %r); import torch; from oracle.replay import ndt_soak; st = ndt_soak(120, 13); "
            "print(json.dumps({k: v for k, v in st.items() if k not in ('over_bar', 'unexplained')}))" % root)
    out = {}
    for name in ("lu", "svd"):
        env = dict(os.environ, OMP_NUM_THREADS="8")
        env.pop("MRGFE_NEWTON_SVD", None)
        if name == "svd":
            env["MRGFE_NEWTON_SVD"] = "1"
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = json.loads(r.stdout.strip().splitlines()[-1])
    lu, svd = out["lu"], out["svd"]
    print(lu, svd)
    for k in ("ndt", "ndt_flag_or_iteration_mismatch", "other_over_bar", "other_exact"):
        assert lu[k] == svd[k], k
    assert abs(lu["ndt_over_bar"] - svd["ndt_over_bar"]) <= 1 and abs(lu["ndt_exact_ref"] - svd["ndt_exact_ref"]) <= 2
    assert lu["ndt_settled_over_bar"] == svd["ndt_settled_over_bar"] == 0


def test_soak_round3_methods():
    """The same soak for the methods added in round 3: pcl::GICP, pclomp::GICP and ICP with reciprocal correspondences, 60 random scenes.
    ICP: the bar of 1e-4 m / 1e-4 rad.  Serial pcl::GICP is deterministic — its BFGS line search amplifies the ORDER of the f64 cost sums into
    millimetres on about one scene in fourteen, so PCL_GICP_HIP adds them in the reference's order (round 4) and every result must be
    bit-identical to the reference-order oracle.  pclomp::GICP adds per-thread partials over static chunks in thread order — fixed for a given thread
    count: PCL_GICP_OMP_HIP reproduces it for T = 8 (round 5) and must be bit-identical to the 8-thread oracle too: nothing leaves the bar."""
    from oracle.replay import round3_soak

    st = round3_soak(60, 29)
    print({k: v for k, v in st.items() if k != "over_bar"})
    for u in st["over_bar"]:
        print("over the bar:", u)
    assert st["icp_over_bar"] == 0 and st["icp_flag_or_iteration_mismatch"] == 0
    assert st["gicp_serial"] >= 10 and st["gicp_serial_exact_ref"] == st["gicp_serial"], "a serial pcl::GICP result differs from the reference-order oracle"
    assert st["gicp_omp"] >= 10 and st["gicp_omp_exact_ref"] == st["gicp_omp"], "a pclomp::GICP result differs from the 8-thread oracle"
    assert st["gicp_omp_over_bar"] == 0 and st["gicp_serial_over_bar"] == 0
    assert st["gicp_flag_or_iteration_mismatch"] == 0


def test_soak_pcl_ndt():
    """PCL_NDT_HIP (registration_method "NDT"): 120 random scenes against the reference-order oracle.  All pair terms are f64 on both sides; what
    differs is the association and order of the sums (and the device's exp against glibc's in the last bit).  On this draw: no scene over 1e-4 m /
    1e-4 rad, the same flags, iteration and evaluation counts.  (The 1200-scene soak of profiles/r05_soak.json has two scenes over the bar, both runs
    of tens of iterations that do not settle: DESIGN.md section 2.)"""
    from oracle.replay import pclndt_soak

    st = pclndt_soak(120, 31)
    print({k: v for k, v in st.items() if k != "over_bar_cases"})
    assert st["over_bar"] == 0, st["over_bar_cases"]
    assert st["flag_or_iteration_mismatch"] == 0 and st["evaluation_count_mismatch"] == 0
    assert st["worst"] <= 1e-6 and st["exact"] >= 0.95 * st["cases"]
    assert 10 <= st["one_iteration"] < st["cases"] and st["iterations_total"] > 4 * st["cases"]  # both regimes are in the sample


def test_pcl_gicp_tree_sums_still_equal_the_gpu_order_oracle():
    """The round-3 evaluation (block tree) stays behind mrgfe_dbg_set_pclgicp_reference_order(0): bit-identical to the oracle in the kernels' order."""
    from mrg_slam_amd import PclGicpHip
    from mrg_slam_amd._lib import lib
    from oracle import oracle as orc
    from oracle.replay import soak_scene

    rng = np.random.default_rng(5)
    try:
        assert lib().mrgfe_dbg_set_pclgicp_reference_order(0) == 0
        for _ in range(6):
            tgt, src, guess, eps = soak_scene(rng)
            g, r = PclGicpHip(transformation_epsilon=eps), orc.PclGicp(transformation_epsilon=eps, num_threads=1, gpu_order=True)
            for x in (g, r):
                x.setInputTarget(tgt)
                x.setInputSource(src)
                x.align(guess)
            np.testing.assert_array_equal(g.getFinalTransformation(), r.getFinalTransformation())
    finally:
        lib().mrgfe_dbg_set_pclgicp_reference_order(1)
