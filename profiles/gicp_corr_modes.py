#!/usr/bin/env python3
"""GICP frame time against the displacement between frame and keyframe, by correspondence-search mode (MRGFE_GICP_CORR_PASSES 0: one lane group per
query, 2: the passes of getFitnessScore carrying the index):  python3 profiles/gicp_corr_modes.py
BASELINE config[2] shape: ~130k-point frames (distance filter only), resident, setInputSourceDevice + align against a ~130k-point keyframe 1 .. 6 m away."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from mrg_slam_amd import Context, SmallGicpHip, distance_filter, prefilter, synth

    ctx = Context(0)
    scene = synth.street_scene()
    K = 7
    poses = synth.arc_trajectory(K)
    raw = [synth.synth_lidar(scene, poses[k], "VLP64", synth.BASE_SEED + k) for k in range(K)]
    pre = len(sys.argv) > 1 and sys.argv[1] == "prefiltered"  # the ~33k-point clouds of the odometry / loop-closure path instead of the ~130k-point ones
    scans = [prefilter(s, ctx=ctx) if pre else distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw]
    dev = [torch.from_numpy(s).to("cuda:0") for s in scans]
    odo = SmallGicpHip(transformation_epsilon=0.1, ctx=ctx)
    odo.setInputTarget(scans[0])
    out = {"mode": os.environ.get("MRGFE_GICP_CORR_PASSES", "default"), "points": len(scans[1])}
    far = len(sys.argv) > 2 and sys.argv[2] == "far"  # loop-closure-sized guesses (0.5 m / 2 deg): several outer iterations
    for k in range(1, K):
        guess = synth.warm_guess(np.linalg.inv(poses[0]) @ poses[k], k)
        if far:
            guess = synth.perturb_pose(np.linalg.inv(poses[0]) @ poses[k], np.random.default_rng(4242 + k), sigma_t=(0.5, 0.5, 0.1), sigma_r_deg=(0.5, 0.5, 2.0))
        tf, fin = [], None
        for rep in range(6):
            ctx.synchronize()
            t1 = time.perf_counter()
            odo.setInputSourceDevice(dev[k].data_ptr(), len(scans[k]))
            odo.align(guess)
            tf.append(time.perf_counter() - t1)
            fin = odo.getFinalTransformation()
        out[f"{k} m"] = {"frame_ms": round(1e3 * float(np.median(tf[2:])), 4), "iterations": int(odo.getFinalNumIteration()),
                         "T_sha": __import__("hashlib").sha256(np.ascontiguousarray(fin).tobytes()).hexdigest()[:8]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
