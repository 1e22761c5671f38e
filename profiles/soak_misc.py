#!/usr/bin/env python3
"""Randomised parity soak of the rows around the alignment against the CPU oracle:
    python3 profiles/soak_misc.py [cases=400] > gpurun_out/soak_misc.json
k-NN rows and 1-NN (bit-exact indices and distances), getFitnessScore / calc_fitness_score (relative 1e-12: fixed-order f64 sums of float distances on both
sides, in different orders), the information matrix, the map cloud, other-robot point removal and deskewing (bit-exact).  Kept as profiles/<tag>_soak_misc.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from mrg_slam_amd import InformationMatrixCalculator, KeyFrameSnapshot, MapCloudGenerator, calc_fitness_score, deskew, knn, remove_points_near, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle.replay import small_cloud  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(5105)
names = ("knn_rows", "fitness_score", "information_matrix", "map_cloud", "remove_points_near", "deskew")
tally = {k: [0, 0] for k in names}
worst_fit_rel = 0.0
bad = []
t0 = time.time()
gen = MapCloudGenerator()


def note(name, ok, what):
    tally[name][1] += 1
    if ok:
        tally[name][0] += 1
    elif len(bad) < 40:
        bad.append(f"{name}: {what}")


for t in range(cases):
    n = int(rng.integers(200, 9000))
    ext = (rng.uniform(3, 30), rng.uniform(3, 30), rng.uniform(0.5, 5))
    a = small_cloud(n, 7000 + t, extent=ext)
    rel = synth.make_pose(rng.normal(0, 0.6, 3), synth.rot_xyz(*rng.normal(0, 0.05, 3)))
    b = orc.transform_points(np.linalg.inv(rel), a[: int(rng.integers(100, n + 1))]) + np.float32(0)
    b[:, :3] += rng.normal(0, 0.01, (len(b), 3)).astype(np.float32)
    what = f"case {t}: n={n} extent={tuple(round(e, 2) for e in ext)}"
    # k-NN rows of a cloud's own points (the GICP covariances' and the statistical filter's search) and of another cloud's points
    k = int(rng.choice([1, 5, 20, 31]))
    if n > k:
        q = a if t % 2 == 0 else b[:3000]
        gi, gd = knn(a, q, k)
        oi, od = orc.knn(a, q, k)
        note("knn_rows", np.array_equal(gi, oi) and np.array_equal(gd, od), what + f" k={k}")
    # getFitnessScore / calc_fitness_score
    guess = rel @ synth.make_pose(rng.normal(0, 0.05, 3), synth.rot_xyz(*rng.normal(0, 0.01, 3)))
    max_range = float(rng.choice([np.inf, 2.0, 0.5]))
    g, o = calc_fitness_score(a, b, guess, max_range), orc.calc_fitness_score(a, b, guess, max_range)
    r = abs(g - o) / max(abs(o), 1e-300) if np.isfinite(o) and o != 0 else (0.0 if g == o or (not np.isfinite(g) and not np.isfinite(o)) else 1.0)
    worst_fit_rel = max(worst_fit_rel, r)
    note("fitness_score", r <= 1e-12, what + f" max_range={max_range} hip={g!r} oracle={o!r}")
    params = {"var_gain_a": float(rng.choice([2.0, 20.0])), "fitness_score_thresh": float(rng.choice([0.5, 1.25, 2.5])), "use_const_inf_matrix": bool(t % 9 == 8)}
    gm = InformationMatrixCalculator(params).calc_information_matrix(a, b, guess)
    om, ofit = orc.calc_information_matrix(a, b, guess, params)
    note("information_matrix", np.allclose(gm, om, rtol=1e-11, atol=0.0), what + f" {params}")
    # map cloud of a few keyframes
    K = int(rng.integers(1, 6))
    kposes = [synth.make_pose(rng.normal(0, 4.0, 3), synth.rot_z(rng.normal(0, 0.5))) for _ in range(K)]
    kclouds = [small_cloud(int(rng.integers(50, 4000)), 9000 + 10 * t + j, extent=ext) for j in range(K)]
    res, minp, far, skip = float(rng.choice([0.05, 0.1, 0.5])), int(rng.choice([1, 1, 2])), float(rng.choice([1e4, 15.0])), bool(t % 5 == 4)
    gm_ = gen.generate([KeyFrameSnapshot(kposes[j], kclouds[j], j == 0) for j in range(K)], res, minp, far, skip)
    om_, st = orc.map_cloud_generate(kclouds, kposes, [1 if j == 0 else 0 for j in range(K)], res, minp, far, skip)
    note("map_cloud", (gm_ is None and st != 0) or (gm_ is not None and st == 0 and np.array_equal(gm_, om_)), what + f" K={K} res={res} min={minp} far={far} skip={skip}")
    # other robots' points and deskewing
    ctr = rng.uniform(-ext[0], ext[0], (int(rng.integers(1, 4)), 3))
    rad = float(rng.choice([0.5, 2.0, 5.0]))
    gk, gr = remove_points_near(a, ctr, rad)
    ok_, or_ = orc.remove_points_near(a, ctr, rad)
    note("remove_points_near", np.array_equal(gk, ok_) and np.array_equal(gr, or_), what + f" radius={rad}")
    av = rng.normal(0, 0.5, 3)
    note("deskew", np.array_equal(deskew(a, av, 0.1), orc.deskew(a, av, 0.1)), what)
    if t % 100 == 99:
        print(f"[soak_misc] {t + 1} cases, {time.time() - t0:.0f} s", file=sys.stderr)
print(json.dumps({"cases": cases, "seed": 5105, "within_bar_of_run": {k: f"{v[0]}/{v[1]}" for k, v in tally.items()},
                  "bars": {"knn_rows": "bit-exact indices and squared distances", "fitness_score": "relative 1e-12", "information_matrix": "relative 1e-11", "map_cloud": "bit-exact",
                           "remove_points_near": "bit-exact", "deskew": "bit-exact"}, "worst_fitness_relative_difference": worst_fit_rel, "outside": bad, "seconds": time.time() - t0}))
