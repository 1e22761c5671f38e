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
    # every GPU trajectory repeats bit for bit on the CPU in GPU order, up to the rare scene where the f64 Hessian pass sees the two
    # C libraries' exp differ in the last bit (it then still ends within 1e-6)
    assert st["ndt_exact_gpu_order"] >= 0.97 * st["ndt"]
    for u in st["unexplained"]:
        assert u["dt_vs_replay_m"] <= 1e-6, u
    # reference order: parity wherever the optimisation settles; a settled scene over the bar must be order noise (equal to the replay).
    # The COUNT of such scenes is not an allowance of this test any more: bench.py prints it (`soak_over_bar`) with every run.
    assert st["ndt_exact_ref"] >= 0.9 * st["ndt"]
    for u in st["over_bar"]:
        if u["settled"]:
            assert u["equal_to_gpu_order_replay"], u
    assert st["ndt_settled_over_bar"] <= 3  # regression guard only (round 3 measured 0-2 per 232; the number itself is in the bench line)


def test_soak_round3_methods():
    """The same soak for the methods added in round 3: pcl::GICP (both stopping rules of its BFGS) and ICP with reciprocal correspondences,
    60 random scenes.  ICP: the bar of 1e-4 m / 1e-4 rad.  pcl::GICP: its BFGS line search amplifies the order of the f64 cost sums (the
    reference adds the terms one after the other, the kernels in a tree: 1e-16 relative) into millimetres on about one scene in eight — so
    every result must be bit-identical to the oracle run with its sums in the kernels' order (PclGicp(gpu_order=True): same decisions, same
    transform) with the same flags and iteration counts as the reference-order run; how many leave the bar against THAT run, and by how much,
    is reported by bench.py (`soak_over_bar.pcl_gicp`), not allowed for here."""
    from oracle.replay import round3_soak

    st = round3_soak(60, 29)
    print({k: v for k, v in st.items() if k != "over_bar"})
    for u in st["over_bar"]:
        print("over the bar:", u)
    assert st["icp_over_bar"] == 0 and st["icp_flag_or_iteration_mismatch"] == 0
    assert st["gicp_exact_gpu_order"] == st["gicp"], "a pcl::GICP result differs from the oracle in the kernels' summation order"
    assert st["gicp_flag_or_iteration_mismatch"] == 0
    assert st["gicp_over_bar_equal_to_gpu_order_replay"] == st["gicp_over_bar"]
    assert st["gicp_worst"] <= 5e-3 and st["gicp_exact_ref"] >= 0.8 * st["gicp"]  # regression guards only (round 3: 5 of 45 off by 0.1 - 1.7 mm; the numbers are in the bench line)
