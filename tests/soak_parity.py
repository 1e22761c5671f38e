#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity soak (not collected by pytest): python tests/soak_parity.py [cases] [seed]
Random small scenes, resolutions, search methods, epsilons and guesses; prints the worst translation / rotation difference
of the final transforms and every case that is not bit-identical."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from conftest import small_cloud
    from mrg_slam_amd import GicpHip, IcpHip, NdtHip, SmallGicpHip, VgicpHip, synth
    from oracle import oracle as orc

    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    worst_t = worst_r = 0.0
    n_exact = n_conv_mismatch = 0
    for c in range(cases):
        n = int(rng.integers(1500, 9000))
        tgt = small_cloud(n, int(rng.integers(1 << 30)), extent=(float(rng.uniform(15, 45)), float(rng.uniform(10, 40)), float(rng.uniform(2, 6))))
        rel = synth.make_pose(rng.normal(0, 0.3, 3), synth.rot_xyz(*rng.normal(0, 0.03, 3)))
        src = orc.transform_points(np.linalg.inv(rel), tgt[: int(n * rng.uniform(0.5, 1.0))])
        src[:, :3] += rng.normal(0, 0.01, (len(src), 3)).astype(np.float32)
        guess = synth.perturb_pose(rel if rng.random() < 0.7 else np.eye(4), rng)
        eps = float(rng.choice([0.1, 0.01, 0.001]))
        if rng.random() < 0.8:
            res = float(rng.choice([0.5, 1.0, 1.5, 2.0]))
            search = str(rng.choice(["DIRECT7", "DIRECT1", "DIRECT26", "KDTREE"]))
            g = NdtHip(resolution=res, transformation_epsilon=eps, maximum_iterations=64, search=search)
            o = orc.Ndt(resolution=res, transformation_epsilon=eps, maximum_iterations=64, num_threads=8, search=search)
            tag = f"NDT res={res} {search} eps={eps}"
        elif rng.random() < 0.2:
            g = IcpHip(transformation_epsilon=eps * 1e-3)
            o = orc.Icp(transformation_epsilon=eps * 1e-3)
            tag = f"ICP eps={eps * 1e-3}"
        elif rng.random() < 0.3:
            vres = float(rng.choice([0.5, 1.0, 2.0]))
            g = VgicpHip(resolution=vres, transformation_epsilon=eps)
            o = orc.FastVgicp(resolution=vres, transformation_epsilon=eps, num_threads=1)
            tag = f"VGICP res={vres} eps={eps}"
        elif rng.random() < 0.5:
            g = GicpHip(transformation_epsilon=eps)
            o = orc.FastGicp(transformation_epsilon=eps, num_threads=8)
            tag = f"GICP eps={eps}"
        else:
            g = SmallGicpHip(transformation_epsilon=eps)
            o = orc.SmallGicp(transformation_epsilon=eps, num_threads=8)
            tag = f"SMALL_GICP eps={eps}"
        ok = g.setInputTarget(tgt)
        ook = o.setInputTarget(tgt)
        g.setInputSource(src)
        o.setInputSource(src)
        g.align(guess)
        o.align(guess)
        Tg, To = g.getFinalTransformation().astype(np.float64), o.getFinalTransformation().astype(np.float64)
        dt, dr = float(np.linalg.norm(Tg[:3, 3] - To[:3, 3])), synth.rotation_angle(Tg, To)
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
        exact = np.array_equal(g.getFinalTransformation(), o.getFinalTransformation())
        n_exact += exact
        if bool(g.hasConverged()) != bool(o.hasConverged()) or g.getFinalNumIteration() != o.getFinalNumIteration():
            n_conv_mismatch += 1
        if not exact:
            print(f"case {c}: {tag} n={n} dt={dt:.3e} dr={dr:.3e} conv {g.hasConverged()}/{o.hasConverged()} it {g.getFinalNumIteration()}/{o.getFinalNumIteration()}")
    print(f"{cases} cases: {n_exact} bit-identical, worst dt {worst_t:.3e} m, worst dr {worst_r:.3e} rad, convergence/iteration mismatches {n_conv_mismatch}")


if __name__ == "__main__":
    main()
