#!/usr/bin/env python3
"""Generates tests/golden/frontend_small.npz: small seeded inputs and the CPU oracle's outputs for them.

The reference (aserbremen/mrg_slam) ships no tests, fixtures or golden vectors for this path and its arithmetic lives in
un-vendored PCL / ndt_omp / fast_gicp, so these vectors are produced by this repo's own CPU restatement
(oracle/) - they freeze ITS numbers: the CPU suite checks that the oracle still reproduces them, the GPU suite checks
the HIP path against them on the GPU box.  Re-run only when the oracle is deliberately changed:
    python tests/golden/make_golden.py                      # all three files
    python tests/golden/make_golden.py --small-gicp-only    # tests/golden/small_gicp.npz only
    python tests/golden/make_golden.py --vgicp-only         # tests/golden/vgicp.npz only
    python tests/golden/make_golden.py --icp-only           # tests/golden/icp.npz only
    python tests/golden/make_golden.py --round3-only        # tests/golden/round3.npz only (pcl::GICP, pclomp::GICP, reciprocal ICP)
    python tests/golden/make_golden.py --pcl-ndt-only       # tests/golden/pcl_ndt.npz only (pcl::NormalDistributionsTransform, registration_method "NDT")
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import small_cloud  # noqa: E402
from mrg_slam_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    out = {}
    tgt = small_cloud(2500, 2024)
    rel = synth.make_pose([0.22, -0.08, 0.02], synth.rot_xyz(0.008, -0.006, 0.025))
    src = orc.transform_points(np.linalg.inv(rel), tgt)
    src[:, :3] += np.random.default_rng(99).normal(0, 0.01, (len(src), 3)).astype(np.float32)
    guess = synth.warm_guess(rel, 5)
    out.update(tgt=tgt, src=src, rel=rel, guess=guess)
    # prefilters
    raw = small_cloud(3000, 77, extent=(40, 30, 4))
    out["raw"] = raw
    out["distance_out"] = orc.distance_filter(raw, 0.1, 35.0)
    out["voxel_out_0p1"], _ = orc.voxelgrid(out["distance_out"], 0.1, 1)
    out["voxel_out_0p5_min2"], _ = orc.voxelgrid(out["distance_out"], 0.5, 2)
    out["radius_out"], out["radius_keep"] = orc.radius_outlier(out["voxel_out_0p1"], 0.5, 2)
    out["sor_out"], out["sor_keep"] = orc.statistical_outlier(out["voxel_out_0p1"], 30, 1.2)
    # NDT
    for eps, tag in ((0.1, "eps0p1"), (0.01, "eps0p01")):
        ndt = orc.Ndt(resolution=1.0, transformation_epsilon=eps, maximum_iterations=64, num_threads=2)
        assert ndt.setInputTarget(tgt) == 0
        ndt.setInputSource(src)
        ndt.align(guess)
        out[f"ndt_{tag}_T"] = ndt.getFinalTransformation()
        out[f"ndt_{tag}_H"] = ndt.getHessian()
        out[f"ndt_{tag}_meta"] = np.array([ndt.hasConverged(), ndt.getFinalNumIteration(), ndt.evals], dtype=np.int64)
        out[f"ndt_{tag}_fitness"] = np.array([ndt.getFitnessScore(), ndt.getTransformationProbability()])
    keys, npts, mean, cov, icov = ndt.leaves()
    out.update(ndt_leaf_keys=keys, ndt_leaf_npts=npts, ndt_leaf_mean=mean, ndt_leaf_icov=icov)
    p = np.array([0.2, -0.05, 0.01, 0.012, -0.006, 0.025])
    out["eval_p"] = p
    out["eval_T"] = orc.pose_to_matrix(p)
    for mode in (0, 1, 2):
        s, g, H = ndt.evaluate(out["eval_T"], p, mode)
        out[f"eval{mode}_score"], out[f"eval{mode}_g"], out[f"eval{mode}_H"] = np.array([s]), g, H
    # GICP
    gicp = orc.FastGicp(transformation_epsilon=0.01, num_threads=2)
    gicp.setInputTarget(tgt)
    gicp.setInputSource(src)
    gicp.align(guess)
    out["gicp_T"] = gicp.getFinalTransformation()
    out["gicp_H"] = gicp.getFinalHessian()
    out["gicp_meta"] = np.array([gicp.hasConverged(), gicp.getFinalNumIteration()], dtype=np.int64)
    out["gicp_src_cov"] = gicp.covariances("source")[:64]
    e, H, b, n = gicp.linearize(np.asarray(guess, dtype=np.float64))
    out["gicp_lin_err"], out["gicp_lin_H"], out["gicp_lin_b"], out["gicp_lin_n"] = np.array([e]), H, b, np.array([n])
    # fitness / NN
    out["fitness_inf"] = np.array([orc.calc_fitness_score(tgt, src, rel)])
    idx, sqd = orc.nn1_brute(tgt, src[:200])
    out.update(nn_idx=idx, nn_sqd=sqd)
    path = os.path.join(ROOT, "tests", "golden", "frontend_small.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")

    # ---- map cloud, other-robot removal, deskewing (tests/golden/perpoint_small.npz) -----
    pp = {}
    rng = np.random.default_rng(31)
    for k in range(3):
        c = rng.normal(0, 5, (900 + 50 * k, 4)).astype(np.float32)
        c[:, 3] = rng.uniform(0, 1, len(c)).astype(np.float32)
        pp[f"kf{k}_cloud"] = c
        pp[f"kf{k}_pose"] = synth.make_pose([2.0 * k, -1.0 * k, 0.05 * k], synth.rot_xyz(0.01 * k, -0.02 * k, 0.3 * k))
    clouds, poses = [pp[f"kf{k}_cloud"] for k in range(3)], [pp[f"kf{k}_pose"] for k in range(3)]
    pp["map_0p5"], _ = orc.map_cloud_generate(clouds, poses, [1, 0, 0], 0.5, 1, 10000.0, False)
    pp["map_0p25_min2_far8_skip"], _ = orc.map_cloud_generate(clouds, poses, [1, 0, 0], 0.25, 2, 8.0, True)
    pp["map_full_far6"], _ = orc.map_cloud_generate(clouds, poses, [1, 0, 0], 0.0, 1, 6.0, False)
    pp["centres"] = np.array([[1.0, 0.5, 0.0], [-2.0, 2.0, 0.3]], dtype=np.float32)
    pp["near_kept"], pp["near_removed"] = orc.remove_points_near(clouds[0], pp["centres"], 1.5)
    pp["ang_v"] = np.array([0.31, -0.22, 0.77], dtype=np.float32)
    pp["deskewed"] = orc.deskew(clouds[1], pp["ang_v"], 0.1)
    path = os.path.join(ROOT, "tests", "golden", "perpoint_small.npz")
    np.savez_compressed(path, **pp)
    print(path, os.path.getsize(path), "bytes")


def small_gicp():
    """tests/golden/small_gicp.npz: the restated small_gicp (oracle variant 1) on the inputs of frontend_small.npz
    (loaded from the committed file, which is left untouched)."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "frontend_small.npz"))
    out = {}
    g = orc.SmallGicp(transformation_epsilon=0.01, num_threads=1)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    for tag, guess in (("warm", G["guess"]), ("identity", np.eye(4))):
        g.align(guess)
        out[f"{tag}_T"] = g.getFinalTransformation()
        out[f"{tag}_H"] = g.getFinalHessian()
        out[f"{tag}_meta"] = np.array([g.hasConverged(), g.getFinalNumIteration()], dtype=np.int64)
    e, H, b, n = g.linearize(np.asarray(G["guess"], dtype=np.float64))
    out["lin_err"], out["lin_H"], out["lin_b"], out["lin_n"] = np.array([e]), H, b, np.array([n])
    path = os.path.join(ROOT, "tests", "golden", "small_gicp.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def vgicp():
    """tests/golden/vgicp.npz: the restated fast_gicp::FastVGICP (oracle variant 2) on the inputs of frontend_small.npz."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "frontend_small.npz"))
    out = {}
    g = orc.FastVgicp(resolution=1.0, transformation_epsilon=0.01, num_threads=1)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    for tag, guess in (("warm", G["guess"]), ("identity", np.eye(4))):
        g.align(guess)
        out[f"{tag}_T"] = g.getFinalTransformation()
        out[f"{tag}_H"] = g.getFinalHessian()
        out[f"{tag}_meta"] = np.array([g.hasConverged(), g.getFinalNumIteration()], dtype=np.int64)
    e, H, b, n = g.linearize(np.asarray(G["guess"], dtype=np.float64))
    out["lin_err"], out["lin_H"], out["lin_b"], out["lin_n"] = np.array([e]), H, b, np.array([n])
    out["num_voxels"] = np.array([g.numVoxels()])
    path = os.path.join(ROOT, "tests", "golden", "vgicp.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def icp():
    """tests/golden/icp.npz: the restated pcl::IterativeClosestPoint (oracle variant 3) on the inputs of frontend_small.npz."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "frontend_small.npz"))
    out = {}
    for tag, guess, eps in (("warm", G["guess"], 0.01), ("identity", np.eye(4), 1e-6)):
        g = orc.Icp(transformation_epsilon=eps)
        g.setInputTarget(G["tgt"])
        g.setInputSource(G["src"])
        g.align(guess)
        out[f"{tag}_T"] = g.getFinalTransformation()
        out[f"{tag}_meta"] = np.array([g.hasConverged(), g.getFinalNumIteration()], dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "icp.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def round3():
    """tests/golden/round3.npz: the restated pcl::GeneralizedIterativeClosestPoint ("GICP") and pclomp::GICP ("GICP_OMP", the older stopping rule
    of the inner BFGS), and pcl::IterativeClosestPoint with reciprocal correspondences, on the inputs of frontend_small.npz."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "frontend_small.npz"))
    out = {}
    for tag, omp in (("gicp", False), ("gicp_omp", True)):
        g = orc.PclGicp(transformation_epsilon=0.01, omp=omp, num_threads=1)
        g.setInputTarget(G["tgt"])
        g.setInputSource(G["src"])
        g.align(G["guess"])
        out[f"{tag}_T"] = g.getFinalTransformation()
        out[f"{tag}_meta"] = np.array([g.hasConverged(), g.getFinalNumIteration()], dtype=np.int64)
        out[f"{tag}_src_cov"] = g.covariances("source")[:64]
    g = orc.Icp(transformation_epsilon=0.01, use_reciprocal_correspondences=True)
    g.setInputTarget(G["tgt"])
    g.setInputSource(G["src"])
    g.align(G["guess"])
    out["icp_reciprocal_T"] = g.getFinalTransformation()
    out["icp_reciprocal_meta"] = np.array([g.hasConverged(), g.getFinalNumIteration()], dtype=np.int64)
    path = os.path.join(ROOT, "tests", "golden", "round3.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


def pcl_ndt():
    """tests/golden/pcl_ndt.npz: the restated pcl::NormalDistributionsTransform (oracle/pcl_ndt.cpp; registration_method "NDT", registrations.cpp:115-129)
    on the inputs of frontend_small.npz: mrg_slam's epsilon (one Newton iteration under PCL's rule) and a tight one, plus one evaluation of each kind."""
    G = np.load(os.path.join(ROOT, "tests", "golden", "frontend_small.npz"))
    out = {}
    for eps, tag in ((0.1, "eps0p1"), (1e-6, "eps1em6")):
        g = orc.PclNdt(resolution=1.0, transformation_epsilon=eps, maximum_iterations=64)
        assert g.setInputTarget(G["tgt"]) == 0
        g.setInputSource(G["src"])
        g.align(G["guess"])
        out[f"{tag}_T"] = g.getFinalTransformation()
        out[f"{tag}_H"] = g.getHessian()
        out[f"{tag}_meta"] = np.array([g.hasConverged(), g.getFinalNumIteration(), g.evals], dtype=np.int64)
        out[f"{tag}_fitness"] = np.array([g.getFitnessScore(), g.getTransformationLikelihood()])
    for mode in (0, 1, 2):
        s, gr, H = g.evaluate(G["eval_T"], G["eval_p"], mode)
        out[f"eval{mode}_score"], out[f"eval{mode}_g"], out[f"eval{mode}_H"] = np.array([s]), gr, H
    path = os.path.join(ROOT, "tests", "golden", "pcl_ndt.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if "--pcl-ndt-only" in sys.argv:
        pcl_ndt()
    elif "--round3-only" in sys.argv:
        round3()
    elif "--icp-only" in sys.argv:
        icp()
    elif "--small-gicp-only" in sys.argv:
        small_gicp()
    elif "--vgicp-only" in sys.argv:
        vgicp()
    else:
        main()
        small_gicp()
        vgicp()
        icp()
        round3()
        pcl_ndt()
