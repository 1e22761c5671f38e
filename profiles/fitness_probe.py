"""Where the getFitnessScore(inf) time of config[3] goes: the batch's fitness epilogue timed for a few max_range values, and the
distribution of the nearest-neighbour distances of one candidate pair.  python3 profiles/fitness_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import bench
    from mrg_slam_amd import BatchMatcher, distance_filter

    raw, pairs = bench.make_loop_workload()
    scans = [distance_filter(s, 0.1, 35.0) for s in raw]
    dev = [torch.from_numpy(s).cuda() for s in scans]
    ids = list(range(256))
    targets = sorted({pairs[i][0] for i in ids})
    tpos = {a: k for k, a in enumerate(targets)}
    args = ([dev[a].data_ptr() for a in targets], [len(scans[a]) for a in targets], np.array([tpos[pairs[i][0]] for i in ids], dtype=np.int32),
            [dev[pairs[i][1]].data_ptr() for i in ids], [len(scans[pairs[i][1]]) for i in ids], np.stack([pairs[i][2] for i in ids]))
    bm = BatchMatcher(transformation_epsilon=0.1, maximum_iterations=64)
    for mr in (-1.0, 0.05, 0.25, 1.0, 4.0, 25.0, float("inf")):
        ts = []
        for _ in range(4):
            bm.clear()
            bm.add_device(*args)
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = bm.align(mr)
            ts.append(1e3 * (time.perf_counter() - t))
        print(f"max_range {mr:6}: {min(ts):7.2f} ms  mean fitness {np.mean(r['fitness'][r['fitness'] < 1e30]) if mr >= 0 else 0:.4f}")
    # one pair: distances
    from mrg_slam_amd import NdtHip
    from mrg_slam_amd.registration import result_matrix
    i = 5
    a, b, guess, _ = pairs[i]
    T = result_matrix(r[i])
    src = scans[b][:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    from scipy.spatial import cKDTree
    d, _ = cKDTree(scans[a][:, :3].astype(np.float64)).query(src)
    qs = [0.1, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99, 1.0]
    print("pair", i, "n", len(d), "NN distance quantiles", {q: round(float(np.quantile(d, q)), 3) for q in qs})
    for thr in (0.125, 0.25, 0.375, 0.5, 1.0, 2.0, 4.0, 8.0):
        print(f"  > {thr} m: {float((d > thr).mean()):.3f}")


main()

