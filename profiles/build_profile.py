#!/usr/bin/env python3
"""The NDT target build alone (setInputTarget of a batch: bounding box, voxel keys, radix sort, run heads, per-voxel sums, leaves):
    rocprofv3 --kernel-trace --stats -- python3 profiles/build_profile.py [targets=256] [steps=10]
`targets` config[1]-shaped clouds (VLP-64 street scans after the 0.1 - 35 m distance filter, ~129k points, nine distinct ones cycled) resident in HBM,
mrgfe_batch_build_targets with a stream synchronisation either side; last line: JSON with the median ms per build and the §8(d) byte model."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from mrg_slam_amd import BatchMatcher, Context, distance_filter, synth

    n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    ctx = Context(0)
    scene = synth.street_scene()
    n_scans = 9
    poses = synth.arc_trajectory(n_scans)
    scans = [distance_filter(synth.synth_lidar(scene, poses[k], "VLP64", synth.BASE_SEED + k), 0.1, 35.0, ctx=ctx) for k in range(n_scans)]
    dev = [torch.from_numpy(s).to("cuda:0") for s in scans]
    bm = BatchMatcher(transformation_epsilon=0.1, maximum_iterations=64, ctx=ctx)
    idx = [b % n_scans for b in range(n_t)]
    args = ([dev[a].data_ptr() for a in idx], [len(scans[a]) for a in idx], np.zeros(0, dtype=np.int32), [], [], np.zeros((0, 4, 4)))
    t = []
    for rep in range(steps + 2):
        bm.clear()
        bm.add_device(*args)
        ctx.synchronize()
        t0 = time.perf_counter()
        bm.build_targets()
        ctx.synchronize()
        if rep >= 2:
            t.append(1e3 * (time.perf_counter() - t0))
    pts = float(sum(len(scans[a]) for a in idx))
    alg = pts * 32.0  # N*16 (keys) + N*16 (leaf sums) + V*64 (leaves: a few per cent, left out here)
    ms = float(np.median(t))
    print(json.dumps({"targets": n_t, "points": pts, "steps": steps, "ms_per_build": ms, "min_ms": float(min(t)), "alg_bytes": alg, "GBps": alg / 1e9 / (ms / 1e3),
                      "frac_of_8TBps": alg / 1e9 / (ms / 1e3) / 8000.0}))


main()
