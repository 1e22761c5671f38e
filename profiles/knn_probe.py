#!/usr/bin/env python3
"""k-NN kernel probe (nn_knn_kernel, csrc/nn_grid.hip):  python3 profiles/knn_probe.py [lib.so]
Self k-NN of a ~130k-point VLP-64 cloud (distance filter only: the GICP covariance search of BASELINE config[2]) and of its prefiltered ~33k-point
form (StatisticalOutlierRemoval's search), k = 20 and 31: kernel ms (HIP events, mrgfe_ctx_knn_stats), candidates measured per query, an order-
independent digest of all rows, and a sample of rows against a full sort by (distance, index).  MRGFE_LIB names another build of the library (A/B)."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    os.environ["MRGFE_LIB"] = os.path.abspath(sys.argv[1])


def main():
    from mrg_slam_amd import Context, distance_filter, knn, prefilter, synth
    from mrg_slam_amd._lib import lib

    ctx = Context(0)
    scene = synth.street_scene()
    poses = synth.arc_trajectory(2)
    raw = synth.synth_lidar(scene, poses[0], "VLP64", synth.BASE_SEED)
    clouds = {"130k": distance_filter(raw, 0.1, 35.0, ctx=ctx), "prefiltered": prefilter(raw, ctx=ctx)}
    rng = np.random.default_rng(5)
    out = {"lib": os.environ.get("MRGFE_LIB", "libmrgfe.so")}
    for name, t in clouds.items():
        for k in (20, 31):
            ms = []
            for rep in range(4):
                idx, sqd = knn(t, t, k, ctx=ctx)
                ms.append(ctx.knn_stats()["ms"])
            lib().mrgfe_dbg_set_fit_stats(1)
            knn(t, t, k, ctx=ctx)
            ks = ctx.knn_stats()
            lib().mrgfe_dbg_set_fit_stats(0)
            tx, ty, tz = (t[:, a] for a in range(3))
            bad = 0
            for i in rng.choice(len(t), 200, replace=False):
                dx, dy, dz = tx - t[i, 0], ty - t[i, 1], tz - t[i, 2]
                d = (dx * dx + dy * dy) + dz * dz
                order = np.lexsort((np.arange(len(t)), d))[:k]
                bad += int(not (np.array_equal(idx[i], order.astype(np.int32)) and np.array_equal(sqd[i], d[order])))
            out[f"{name}_k{k}"] = {"points": len(t), "ms": float(np.median(ms[1:])), "candidates_per_query": ks["candidates"] / max(ks["queries"], 1.0),
                                   "rows_sha256_16": hashlib.sha256(idx.tobytes() + sqd.tobytes()).hexdigest()[:16], "sample_rows_wrong": bad}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
