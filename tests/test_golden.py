"""Frozen vectors (tests/golden/frontend_small.npz, made by tests/golden/make_golden.py from the CPU oracle).
CPU: the oracle still reproduces them bit for bit (float results: to 1e-12, they are thread-count independent).
GPU: the HIP path matches them through the C ABI on the GPU box, where /root/reference and its data do not exist."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_small.npz"))
PP = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "perpoint_small.npz"))


# ---- CPU: oracle vs its frozen outputs --------------------------------------------------------------------------
def test_oracle_prefilters_reproduce_golden():
    from oracle import oracle as orc

    d = orc.distance_filter(G["raw"], 0.1, 35.0)
    np.testing.assert_array_equal(d, G["distance_out"])
    v, _ = orc.voxelgrid(d, 0.1, 1)
    np.testing.assert_array_equal(v, G["voxel_out_0p1"])
    np.testing.assert_array_equal(orc.voxelgrid(d, 0.5, 2)[0], G["voxel_out_0p5_min2"])
    r, keep = orc.radius_outlier(v, 0.5, 2)
    np.testing.assert_array_equal(r, G["radius_out"])
    np.testing.assert_array_equal(keep, G["radius_keep"])
    s, keep = orc.statistical_outlier(v, 30, 1.2)
    np.testing.assert_array_equal(s, G["sor_out"])


@pytest.mark.parametrize("eps,tag", [(0.1, "eps0p1"), (0.01, "eps0p01")])
def test_oracle_ndt_reproduces_golden(eps, tag):
    from oracle import oracle as orc

    ndt = orc.Ndt(resolution=1.0, transformation_epsilon=eps, maximum_iterations=64, num_threads=4)
    assert ndt.setInputTarget(G["tgt"]) == 0
    ndt.setInputSource(G["src"])
    ndt.align(G["guess"])
    np.testing.assert_array_equal(ndt.getFinalTransformation(), G[f"ndt_{tag}_T"])
    np.testing.assert_array_equal(ndt.getHessian(), G[f"ndt_{tag}_H"])
    assert [int(ndt.hasConverged()), ndt.getFinalNumIteration(), ndt.evals] == G[f"ndt_{tag}_meta"].tolist()
    np.testing.assert_allclose([ndt.getFitnessScore(), ndt.getTransformationProbability()], G[f"ndt_{tag}_fitness"], rtol=1e-13)
    s, g, H = ndt.evaluate(G["eval_T"], G["eval_p"], 0)
    assert s == G["eval0_score"][0]
    np.testing.assert_array_equal(g, G["eval0_g"])
    np.testing.assert_array_equal(H, G["eval0_H"])


def test_oracle_gicp_reproduces_golden():
    from oracle import oracle as orc

    g = orc.FastGicp(transformation_epsilon=0.01, num_threads=2)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    g.align(G["guess"])
    np.testing.assert_allclose(g.getFinalTransformation(), G["gicp_T"], rtol=0, atol=1e-6)  # per-thread partial sums: order varies
    assert [int(g.hasConverged()), g.getFinalNumIteration()] == G["gicp_meta"].tolist()
    np.testing.assert_array_equal(g.covariances("source")[:64], G["gicp_src_cov"])


def test_oracle_small_gicp_reproduces_golden():
    from oracle import oracle as orc

    S = np.load(os.path.join(os.path.dirname(__file__), "golden", "small_gicp.npz"))
    g = orc.SmallGicp(transformation_epsilon=0.01, num_threads=1)  # one thread: sequential sums, exact reproduction
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    for tag, guess in (("warm", G["guess"]), ("identity", np.eye(4))):
        g.align(guess)
        np.testing.assert_array_equal(g.getFinalTransformation(), S[f"{tag}_T"])
        np.testing.assert_array_equal(g.getFinalHessian(), S[f"{tag}_H"])
        assert [int(g.hasConverged()), g.getFinalNumIteration()] == S[f"{tag}_meta"].tolist()
    e, H, b, n = g.linearize(np.asarray(G["guess"], dtype=np.float64))
    assert e == S["lin_err"][0] and n == S["lin_n"][0]
    np.testing.assert_array_equal(H, S["lin_H"])
    np.testing.assert_array_equal(b, S["lin_b"])
    # the solution is the one the fast_gicp formulation finds (same cost, different parametrisation of the step)
    assert np.linalg.norm(S["warm_T"][:3, 3] - G["gicp_T"][:3, 3]) < 2e-3
    assert np.linalg.norm(S["warm_T"][:3, 3] - G["rel"][:3, 3]) < 0.02


def test_oracle_vgicp_reproduces_golden():
    from oracle import oracle as orc

    V = np.load(os.path.join(os.path.dirname(__file__), "golden", "vgicp.npz"))
    g = orc.FastVgicp(resolution=1.0, transformation_epsilon=0.01, num_threads=1)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    for tag, guess in (("warm", G["guess"]), ("identity", np.eye(4))):
        g.align(guess)
        np.testing.assert_array_equal(g.getFinalTransformation(), V[f"{tag}_T"])
        np.testing.assert_array_equal(g.getFinalHessian(), V[f"{tag}_H"])
        assert [int(g.hasConverged()), g.getFinalNumIteration()] == V[f"{tag}_meta"].tolist()
    e, H, b, n = g.linearize(np.asarray(G["guess"], dtype=np.float64))
    assert e == V["lin_err"][0] and n == V["lin_n"][0] and g.numVoxels() == V["num_voxels"][0]
    np.testing.assert_array_equal(H, V["lin_H"])
    assert np.linalg.norm(V["warm_T"][:3, 3] - G["rel"][:3, 3]) < 0.05


def test_oracle_icp_reproduces_golden():
    from oracle import oracle as orc

    I = np.load(os.path.join(os.path.dirname(__file__), "golden", "icp.npz"))
    for tag, guess, eps in (("warm", G["guess"], 0.01), ("identity", np.eye(4), 1e-6)):
        g = orc.Icp(transformation_epsilon=eps)
        g.setInputTarget(G["tgt"])
        g.setInputSource(G["src"])
        g.align(guess)
        np.testing.assert_array_equal(g.getFinalTransformation(), I[f"{tag}_T"])
        assert [int(g.hasConverged()), g.getFinalNumIteration()] == I[f"{tag}_meta"].tolist()
    assert np.linalg.norm(I["identity_T"][:3, 3] - G["rel"][:3, 3]) < 0.02


def test_oracle_round3_methods_reproduce_golden():
    """pcl::GICP, pclomp::GICP and reciprocal ICP (tests/golden/round3.npz)."""
    from oracle import oracle as orc

    R = np.load(os.path.join(os.path.dirname(__file__), "golden", "round3.npz"))
    for tag, omp in (("gicp", False), ("gicp_omp", True)):
        g = orc.PclGicp(transformation_epsilon=0.01, omp=omp, num_threads=1)
        g.setInputTarget(G["tgt"])
        g.setInputSource(G["src"])
        g.align(G["guess"])
        np.testing.assert_array_equal(g.getFinalTransformation(), R[f"{tag}_T"])
        assert [int(g.hasConverged()), g.getFinalNumIteration()] == R[f"{tag}_meta"].tolist()
        np.testing.assert_array_equal(g.covariances("source")[:64], R[f"{tag}_src_cov"])
    g = orc.Icp(transformation_epsilon=0.01, use_reciprocal_correspondences=True)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    g.align(G["guess"])
    np.testing.assert_array_equal(g.getFinalTransformation(), R["icp_reciprocal_T"])
    assert [int(g.hasConverged()), g.getFinalNumIteration()] == R["icp_reciprocal_meta"].tolist()
    assert np.linalg.norm(R["gicp_T"][:3, 3] - G["rel"][:3, 3]) < 2e-3


def test_oracle_pcl_ndt_reproduces_golden():
    """pcl::NormalDistributionsTransform (tests/golden/pcl_ndt.npz)."""
    from oracle import oracle as orc

    P = np.load(os.path.join(os.path.dirname(__file__), "golden", "pcl_ndt.npz"))
    for eps, tag in ((0.1, "eps0p1"), (1e-6, "eps1em6")):
        g = orc.PclNdt(resolution=1.0, transformation_epsilon=eps, maximum_iterations=64)
        assert g.setInputTarget(G["tgt"]) == 0
        g.setInputSource(G["src"])
        g.align(G["guess"])
        np.testing.assert_array_equal(g.getFinalTransformation(), P[f"{tag}_T"])
        np.testing.assert_array_equal(g.getHessian(), P[f"{tag}_H"])
        assert [int(g.hasConverged()), g.getFinalNumIteration(), g.evals] == P[f"{tag}_meta"].tolist()
        assert [g.getFitnessScore(), g.getTransformationLikelihood()] == P[f"{tag}_fitness"].tolist()
    assert P["eps0p1_meta"][1] == 1 and P["eps1em6_meta"][1] > 3  # PCL's iteration rule: mrg_slam's epsilon stops after one Newton step
    assert np.linalg.norm(P["eps1em6_T"][:3, 3] - G["rel"][:3, 3]) < 0.02
    for mode in (0, 1, 2):
        s, gr, H = g.evaluate(G["eval_T"], G["eval_p"], mode)
        assert s == P[f"eval{mode}_score"][0]
        np.testing.assert_array_equal(gr, P[f"eval{mode}_g"])
        np.testing.assert_array_equal(H, P[f"eval{mode}_H"])


@pytest.mark.gpu
def test_hip_pcl_ndt_matches_golden():
    from mrg_slam_amd import PclNdtHip, synth

    P = np.load(os.path.join(os.path.dirname(__file__), "golden", "pcl_ndt.npz"))
    for eps, tag in ((0.1, "eps0p1"), (1e-6, "eps1em6")):
        g = PclNdtHip(resolution=1.0, transformation_epsilon=eps, maximum_iterations=64)
        assert g.setInputTarget(G["tgt"]) == 0
        g.setInputSource(G["src"])
        g.align(G["guess"])
        T = g.getFinalTransformation()
        assert np.linalg.norm(T[:3, 3].astype(np.float64) - P[f"{tag}_T"][:3, 3]) <= 1e-4 and synth.rotation_angle(T, P[f"{tag}_T"]) <= 1e-4
        assert [int(g.hasConverged()), g.getFinalNumIteration(), g.evals] == P[f"{tag}_meta"].tolist()
        np.testing.assert_allclose(g.getHessian(), P[f"{tag}_H"], rtol=0, atol=1e-8 * np.abs(P[f"{tag}_H"]).max())
        assert g.getFitnessScore() == pytest.approx(P[f"{tag}_fitness"][0], rel=1e-6)
    for mode in (0, 1, 2):
        s, gr, H = g.evaluate(G["eval_T"], G["eval_p"], mode)
        if mode != 2:
            assert s == pytest.approx(P[f"eval{mode}_score"][0], rel=1e-12)
            np.testing.assert_allclose(gr, P[f"eval{mode}_g"], rtol=0, atol=1e-10 * np.abs(P[f"eval{mode}_g"]).max())
        if mode != 1:
            np.testing.assert_allclose(H, P[f"eval{mode}_H"], rtol=0, atol=1e-10 * np.abs(P[f"eval{mode}_H"]).max())


def test_oracle_perpoint_passes_reproduce_golden():
    from oracle import oracle as orc

    clouds, poses = [PP[f"kf{k}_cloud"] for k in range(3)], [PP[f"kf{k}_pose"] for k in range(3)]
    np.testing.assert_array_equal(orc.map_cloud_generate(clouds, poses, [1, 0, 0], 0.5, 1, 10000.0, False)[0], PP["map_0p5"])
    np.testing.assert_array_equal(orc.map_cloud_generate(clouds, poses, [1, 0, 0], 0.25, 2, 8.0, True)[0], PP["map_0p25_min2_far8_skip"])
    np.testing.assert_array_equal(orc.map_cloud_generate(clouds, poses, [1, 0, 0], 0.0, 1, 6.0, False)[0], PP["map_full_far6"])
    kept, removed = orc.remove_points_near(clouds[0], PP["centres"], 1.5)
    np.testing.assert_array_equal(kept, PP["near_kept"])
    np.testing.assert_array_equal(removed, PP["near_removed"])
    np.testing.assert_array_equal(orc.deskew(clouds[1], PP["ang_v"], 0.1), PP["deskewed"])


# ---- GPU: HIP path vs the frozen outputs ------------------------------------------------------------------------
@pytest.mark.gpu
def test_hip_perpoint_passes_match_golden():
    from mrg_slam_amd import KeyFrameSnapshot, MapCloudGenerator, deskew, remove_points_near

    kfs = [KeyFrameSnapshot(PP[f"kf{k}_pose"], PP[f"kf{k}_cloud"], k == 0) for k in range(3)]
    gen = MapCloudGenerator()
    np.testing.assert_array_equal(gen.generate(kfs, 0.5, 1, 10000.0, False), PP["map_0p5"])
    np.testing.assert_array_equal(gen.generate(kfs, 0.25, 2, 8.0, True), PP["map_0p25_min2_far8_skip"])
    np.testing.assert_array_equal(gen.generate(kfs, 0.0, 1, 6.0, False), PP["map_full_far6"])
    kept, removed = remove_points_near(kfs[0].cloud, PP["centres"], 1.5)
    np.testing.assert_array_equal(kept, PP["near_kept"])
    np.testing.assert_array_equal(removed, PP["near_removed"])
    np.testing.assert_array_equal(deskew(kfs[1].cloud, PP["ang_v"], 0.1), PP["deskewed"])


@pytest.mark.gpu
def test_hip_prefilters_match_golden():
    from mrg_slam_amd import RadiusOutlierRemoval, StatisticalOutlierRemoval, VoxelGrid, calc_fitness_score, distance_filter

    d = distance_filter(G["raw"], 0.1, 35.0)
    np.testing.assert_array_equal(d, G["distance_out"])
    vg = VoxelGrid()
    vg.setLeafSize(0.1)
    vg.setInputCloud(d)
    v = vg.filter()
    np.testing.assert_array_equal(v, G["voxel_out_0p1"])
    vg.setLeafSize(0.5)
    vg.setMinimumPointsNumberPerVoxel(2)
    np.testing.assert_array_equal(vg.filter(), G["voxel_out_0p5_min2"])
    ro = RadiusOutlierRemoval()
    ro.setInputCloud(v)
    np.testing.assert_array_equal(ro.filter(), G["radius_out"])
    so = StatisticalOutlierRemoval()
    so.setInputCloud(v)
    np.testing.assert_array_equal(so.filter(), G["sor_out"])
    assert calc_fitness_score(G["tgt"], G["src"], G["rel"]) == pytest.approx(G["fitness_inf"][0], rel=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("eps,tag", [(0.1, "eps0p1"), (0.01, "eps0p01")])
def test_hip_ndt_matches_golden(eps, tag):
    from mrg_slam_amd import NdtHip, synth

    g = NdtHip(resolution=1.0, transformation_epsilon=eps, maximum_iterations=64)
    assert g.setInputTarget(G["tgt"]) == 0
    g.setInputSource(G["src"])
    g.align(G["guess"])
    T, To = g.getFinalTransformation(), G[f"ndt_{tag}_T"]
    assert np.linalg.norm(T[:3, 3].astype(np.float64) - To[:3, 3]) <= 1e-4  # north_star bar
    assert synth.rotation_angle(T, To) <= 1e-4
    assert [int(g.hasConverged()), g.getFinalNumIteration(), g.evals] == G[f"ndt_{tag}_meta"].tolist()
    np.testing.assert_allclose(g.getHessian(), G[f"ndt_{tag}_H"], rtol=0, atol=1e-4 * np.abs(G[f"ndt_{tag}_H"]).max())
    assert g.getFitnessScore() == pytest.approx(G[f"ndt_{tag}_fitness"][0], rel=1e-3)
    keys, npts, mean, icov = g.leaves()
    np.testing.assert_array_equal(keys, G["ndt_leaf_keys"])
    np.testing.assert_array_equal(npts, G["ndt_leaf_npts"])
    np.testing.assert_allclose(mean, G["ndt_leaf_mean"], rtol=0, atol=1e-12)
    for mode in (0, 1, 2):
        s, gr, H = g.evaluate(G["eval_T"], G["eval_p"], mode)
        if mode != 2:
            assert s == pytest.approx(G[f"eval{mode}_score"][0], rel=1e-9)
            np.testing.assert_allclose(gr, G[f"eval{mode}_g"], rtol=0, atol=1e-8 * np.abs(G[f"eval{mode}_g"]).max())
        if mode != 1:
            np.testing.assert_allclose(H, G[f"eval{mode}_H"], rtol=0, atol=2e-6 * np.abs(G[f"eval{mode}_H"]).max())
    idx, sqd = g.nearestKSearch1(G["src"][:200])
    np.testing.assert_array_equal(idx, G["nn_idx"])
    np.testing.assert_array_equal(sqd, G["nn_sqd"])


@pytest.mark.gpu
def test_hip_small_gicp_matches_golden():
    from mrg_slam_amd import SmallGicpHip, synth

    S = np.load(os.path.join(os.path.dirname(__file__), "golden", "small_gicp.npz"))
    g = SmallGicpHip(transformation_epsilon=0.01)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    for tag, guess in (("warm", G["guess"]), ("identity", np.eye(4))):
        g.align(guess)
        T = g.getFinalTransformation()
        assert np.linalg.norm(T[:3, 3].astype(np.float64) - S[f"{tag}_T"][:3, 3]) <= 1e-4
        assert synth.rotation_angle(T, S[f"{tag}_T"]) <= 1e-4
        assert [int(g.hasConverged()), g.getFinalNumIteration()] == S[f"{tag}_meta"].tolist()
    H, b, e, n = g.linearize(np.asarray(G["guess"], dtype=np.float64))
    assert n == S["lin_n"][0] and e == pytest.approx(S["lin_err"][0], rel=1e-12)
    np.testing.assert_allclose(H, S["lin_H"], rtol=0, atol=1e-12 * np.abs(S["lin_H"]).max())


@pytest.mark.gpu
def test_hip_icp_matches_golden():
    from mrg_slam_amd import IcpHip, synth

    I = np.load(os.path.join(os.path.dirname(__file__), "golden", "icp.npz"))
    for tag, guess, eps in (("warm", G["guess"], 0.01), ("identity", np.eye(4), 1e-6)):
        g = IcpHip(transformation_epsilon=eps)
        g.setInputTarget(G["tgt"])
        g.setInputSource(G["src"])
        g.align(guess)
        T = g.getFinalTransformation()
        assert np.linalg.norm(T[:3, 3].astype(np.float64) - I[f"{tag}_T"][:3, 3]) <= 1e-4
        assert synth.rotation_angle(T, I[f"{tag}_T"]) <= 1e-4
        assert [int(g.hasConverged()), g.getFinalNumIteration()] == I[f"{tag}_meta"].tolist()


@pytest.mark.gpu
def test_hip_round3_methods_match_golden():
    from mrg_slam_amd import IcpHip, PclGicpHip, synth

    R = np.load(os.path.join(os.path.dirname(__file__), "golden", "round3.npz"))
    for tag, omp in (("gicp", False), ("gicp_omp", True)):
        g = PclGicpHip(transformation_epsilon=0.01, omp=omp)
        g.setInputTarget(G["tgt"])
        g.setInputSource(G["src"])
        g.align(G["guess"])
        T = g.getFinalTransformation()
        assert np.linalg.norm(T[:3, 3].astype(np.float64) - R[f"{tag}_T"][:3, 3]) <= 1e-4
        assert synth.rotation_angle(T, R[f"{tag}_T"]) <= 1e-4
        assert [int(g.hasConverged()), g.getFinalNumIteration()] == R[f"{tag}_meta"].tolist()
        np.testing.assert_allclose(g.covariances("source")[:64], R[f"{tag}_src_cov"], rtol=0, atol=1e-12)
    g = IcpHip(transformation_epsilon=0.01, use_reciprocal_correspondences=True)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    g.align(G["guess"])
    T = g.getFinalTransformation()
    assert np.linalg.norm(T[:3, 3].astype(np.float64) - R["icp_reciprocal_T"][:3, 3]) <= 1e-4
    assert synth.rotation_angle(T, R["icp_reciprocal_T"]) <= 1e-4
    assert [int(g.hasConverged()), g.getFinalNumIteration()] == R["icp_reciprocal_meta"].tolist()


@pytest.mark.gpu
def test_hip_vgicp_matches_golden():
    from mrg_slam_amd import VgicpHip, synth

    V = np.load(os.path.join(os.path.dirname(__file__), "golden", "vgicp.npz"))
    g = VgicpHip(resolution=1.0, transformation_epsilon=0.01)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    for tag, guess in (("warm", G["guess"]), ("identity", np.eye(4))):
        g.align(guess)
        T = g.getFinalTransformation()
        assert np.linalg.norm(T[:3, 3].astype(np.float64) - V[f"{tag}_T"][:3, 3]) <= 1e-4
        assert synth.rotation_angle(T, V[f"{tag}_T"]) <= 1e-4
        assert [int(g.hasConverged()), g.getFinalNumIteration()] == V[f"{tag}_meta"].tolist()
    H, b, e, n = g.linearize(np.asarray(G["guess"], dtype=np.float64))
    assert n == V["lin_n"][0] and e == pytest.approx(V["lin_err"][0], rel=1e-12)
    np.testing.assert_allclose(H, V["lin_H"], rtol=0, atol=1e-12 * np.abs(V["lin_H"]).max())


@pytest.mark.gpu
def test_hip_gicp_matches_golden():
    from mrg_slam_amd import GicpHip, synth

    g = GicpHip(transformation_epsilon=0.01)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    g.align(G["guess"])
    T = g.getFinalTransformation()
    assert np.linalg.norm(T[:3, 3].astype(np.float64) - G["gicp_T"][:3, 3]) <= 1e-4
    assert synth.rotation_angle(T, G["gicp_T"]) <= 1e-4
    assert [int(g.hasConverged()), g.getFinalNumIteration()] == G["gicp_meta"].tolist()
