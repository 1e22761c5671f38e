#!/usr/bin/env python3
"""Randomised parity soak of the prefilter rows against the CPU oracle, bit for bit:
    python3 profiles/soak_filters.py [cases=600] > gpurun_out/soak_filters.json
Every case: a random structured cloud or a synthetic VLP-16 scan (sizes around the kernels' tile boundaries included, some with non-finite
coordinates), random reference parameters; each filter on its own and the whole chain of PrefilteringComponent::cloud_callback through
mrgfe_prefilter, device-driven (one host wait) and host-driven.  Kept as profiles/<tag>_soak_filters.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (before libmrgfe)

from mrg_slam_amd import ApproximateVoxelGrid, RadiusOutlierRemoval, StatisticalOutlierRemoval, VoxelGrid, distance_filter, prefilter, synth  # noqa: E402
from mrg_slam_amd._lib import lib  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle.replay import small_cloud  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
rng = np.random.default_rng(4104)
scene = synth.street_scene()
scans = [synth.synth_lidar(scene, synth.make_pose([3.0 * k, 0.5 * k, 0.0], synth.rot_z(0.05 * k)), "VLP16", 9000 + k) for k in range(6)]
names = ("distance_filter", "voxelgrid", "approx_voxelgrid", "radius_outlier", "statistical_outlier", "chain_device_driven", "chain_host_driven", "chain_statistical", "chain_approx")
tally = {k: [0, 0] for k in names}  # exact, run
bad = []
sizes = [1, 2, 3, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4095, 4096, 4097]
t0 = time.time()


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def note(name, ok, what):
    tally[name][1] += 1
    if ok:
        tally[name][0] += 1
    elif len(bad) < 40:
        bad.append(f"{name}: {what}")


for t in range(cases):
    if t % 4 == 3:
        cloud = scans[t % len(scans)][: int(rng.integers(2000, len(scans[t % len(scans)])))].copy()
    else:
        n = sizes[t % len(sizes)] if t % 3 == 0 else int(rng.integers(50, 25000))
        cloud = small_cloud(n, 5000 + t, extent=(rng.uniform(2, 40), rng.uniform(2, 40), rng.uniform(0.5, 6)))
    n = len(cloud)
    if t % 7 == 5 and n > 20:
        cloud[rng.choice(n, max(1, n // 40), replace=False), rng.integers(0, 3)] = rng.choice([np.nan, np.inf, -np.inf])
    near, far = float(rng.choice([0.1, 0.5, 2.0])), float(rng.choice([8.0, 35.0, 1e4]))
    leaf = float(rng.choice([0.05, 0.1, 0.25, 0.5, 1.0]))
    min_pts = int(rng.choice([1, 1, 2, 3]))
    radius, min_nb = float(rng.choice([0.3, 0.5, 1.0])), int(rng.choice([1, 2, 4]))
    mean_k, stddev = int(rng.choice([8, 20, 30])), float(rng.choice([0.8, 1.2, 2.0]))
    what = f"case {t}: n={n} near={near} far={far} leaf={leaf} min_pts={min_pts} radius={radius} min_nb={min_nb} mean_k={mean_k} stddev={stddev}"
    note("distance_filter", same(distance_filter(cloud, near, far), orc.distance_filter(cloud, near, far)), what)
    finite = cloud[np.isfinite(cloud[:, :3]).all(axis=1)]
    vg = VoxelGrid(); vg.setLeafSize(leaf); vg.setMinimumPointsNumberPerVoxel(min_pts); vg.setInputCloud(cloud)
    note("voxelgrid", same(vg.filter(), orc.voxelgrid(cloud, leaf, min_pts)[0]), what)
    av = ApproximateVoxelGrid(); av.setLeafSize(leaf, leaf, leaf); av.setInputCloud(cloud)
    note("approx_voxelgrid", same(av.filter(), orc.approx_voxelgrid(cloud, leaf)), what)
    sub = finite[:6000]  # (the oracle's outlier filters are brute force)
    if len(sub):
        ro = RadiusOutlierRemoval(); ro.setRadiusSearch(radius); ro.setMinNeighborsInRadius(min_nb); ro.setInputCloud(sub)
        note("radius_outlier", same(ro.filter(), orc.radius_outlier(sub, radius, min_nb)[0]), what)
        so = StatisticalOutlierRemoval(); so.setMeanK(mean_k); so.setStddevMulThresh(stddev); so.setInputCloud(sub)
        note("statistical_outlier", same(so.filter(), orc.statistical_outlier(sub, mean_k, stddev)[0]), what)
    p = {"distance_near_thresh": near, "distance_far_thresh": far, "downsample_resolution": leaf, "downsample_min_points_per_voxel": min_pts, "radius_radius": radius,
         "radius_min_neighbors": min_nb, "statistical_mean_k": mean_k, "statistical_stddev": stddev}
    c1 = orc.distance_filter(cloud, near, far)
    c2 = orc.voxelgrid(c1, leaf, min_pts)[0]
    exp = orc.radius_outlier(c2, radius, min_nb)[0]
    lib().mrgfe_dbg_set_prefilter_device_driven(1)
    note("chain_device_driven", same(prefilter(cloud, p), exp), what)
    lib().mrgfe_dbg_set_prefilter_device_driven(0)
    note("chain_host_driven", same(prefilter(cloud, p), exp), what)
    lib().mrgfe_dbg_set_prefilter_device_driven(1)
    if t % 3 == 1:
        note("chain_statistical", same(prefilter(cloud, dict(p, outlier_removal_method="STATISTICAL")), orc.statistical_outlier(c2, mean_k, stddev)[0] if len(c2) else c2), what)
        c2a = orc.approx_voxelgrid(c1, leaf)
        note("chain_approx", same(prefilter(cloud, dict(p, downsample_method="APPROX_VOXELGRID")), orc.radius_outlier(c2a, radius, min_nb)[0] if len(c2a) else c2a), what)
    if t % 100 == 99:
        print(f"[soak_filters] {t + 1} cases, {time.time() - t0:.0f} s", file=sys.stderr)
print(json.dumps({"cases": cases, "seed": 4104, "exact_of_run": {k: f"{v[0]}/{v[1]}" for k, v in tally.items()}, "not_exact": bad, "seconds": time.time() - t0}))
