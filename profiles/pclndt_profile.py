#!/usr/bin/env python3
"""Kernel-trace / counter workload for PCL_NDT_HIP (registration_method "NDT" = pcl::NormalDistributionsTransform, the f64 formulation):
    rocprofv3 --kernel-trace --stats -- python3 profiles/pclndt_profile.py [pairs=64] [eps=1e-5] [steps=3]
`pairs` config[1]-shaped pairs (VLP-64 street scans, ~130k points, consecutive poses, warm guesses) through a PCL_NDT_HIP batch, clouds resident in HBM;
the last line is JSON: ms per step, evaluations, the library's HIP-event time and algorithmic bytes of ndt_derivatives_f64_all_kernel."""
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
    from mrg_slam_amd._lib import PCL_NDT_HIP
    from mrg_slam_amd.registration import default_params

    n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = Context(0)
    scene = synth.street_scene()
    n_scans = 9
    poses = synth.arc_trajectory(n_scans)
    raw = [synth.synth_lidar(scene, poses[k], "VLP64", synth.BASE_SEED + k) for k in range(n_scans)]
    scans = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw]
    dev = [torch.from_numpy(s).to("cuda:0") for s in scans]
    prm = default_params(PCL_NDT_HIP)
    prm.transformation_epsilon, prm.maximum_iterations, prm.resolution = eps, 64, 1.0
    bm = BatchMatcher(prm, ctx)
    idx = [(b % (n_scans - 1), b % (n_scans - 1) + 1) for b in range(n_pairs)]
    guesses = np.stack([synth.warm_guess(synth.rel_pose(poses[a], poses[c]), b) for b, (a, c) in enumerate(idx)])
    args = ([dev[a].data_ptr() for a, _ in idx], [len(scans[a]) for a, _ in idx], np.arange(n_pairs, dtype=np.int32), [dev[c].data_ptr() for _, c in idx],
            [len(scans[c]) for _, c in idx], guesses)
    t, k, pts, nb = [], np.zeros(3), 0.0, 0.0
    res = None
    for rep in range(steps + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        bm.clear()
        bm.add_device(*args)
        res = bm.align()
        ctx.synchronize()
        if rep:
            t.append(1e3 * (time.perf_counter() - t0))
            k += np.array(bm.kernel_stats())
            a, b = bm.pair_counts()
            pts, nb = pts + a, nb + b
    print(json.dumps({"pairs": n_pairs, "eps": eps, "steps": steps, "ms_per_step": float(np.median(t)), "iterations_per_alignment": float(res["iterations"].mean()),
                      "evaluations_per_alignment": float(res["evaluations"].mean()), "kernel": "ndt_derivatives_f64_all_kernel", "hip_event_ms": float(k[0]), "launches": int(k[1]),
                      "avg_launch_ms": float(k[0] / k[1]) if k[1] else None, "alg_bytes": float(k[2]), "alg_bytes_per_launch": float(k[2] / k[1]) if k[1] else None,
                      "achieved_GBps": float(k[2] / 1e9 / (k[0] / 1e3)) if k[0] else None, "frac": float(k[2] / 1e9 / (k[0] / 1e3) / 8000.0) if k[0] else None,
                      "mean_neighbours_per_point": nb / pts if pts else None, "byte_model": "N * (16 + 27*8) + neighbours * 112 per evaluation"}))


if __name__ == "__main__":
    main()
