#!/usr/bin/env python3
"""k-NN microbenchmark: one distance-filtered VLP-64 scan (~130k points), k = 20 self-queries (the GICP covariance search)."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mrg_slam_amd import Context, distance_filter, synth
from mrg_slam_amd.filters import knn
from mrg_slam_amd._lib import lib

ctx = Context(0)
scene = synth.street_scene()
poses = synth.arc_trajectory(2)
raw = synth.synth_lidar(scene, poses[0], "VLP64", synth.BASE_SEED)
scan = distance_filter(raw, 0.1, 35.0, ctx=ctx)
lib().mrgfe_dbg_set_fit_stats(int(os.environ.get("STATS", "1")))
for mode in (0,):
    for k in (20, 10, 31):
        ms = []
        for _ in range(6):
            idx, sqd = knn(scan, scan, k, ctx=ctx)
            s = ctx.knn_stats()
            ms.append(s["ms"])
        print("pair", mode, "k", k, "points", len(scan), "ms", np.round(ms, 3), "candidates/query", s["candidates"] / max(1.0, s["queries"]),
              "digest", hashlib.sha256(idx.tobytes() + sqd.tobytes()).hexdigest()[:16], flush=True)
