#!/usr/bin/env python3
"""Kernel-trace workloads for the GICP path (BASELINE config[2] shapes):  rocprofv3 --kernel-trace --stats -- python3 profiles/gicp_profile.py batch|frame
batch: 32 candidate clouds of ~130k points against one keyframe, SMALL_GICP_HIP, covariances recomputed every call (3 calls)
frame: raw VLP-64 scan -> mrgfe_prefilter_device -> setInputSourceDevice -> align against a keyframe, SMALL_GICP_HIP (8 frames)
frame130: distance-filtered ~130k-point frames, resident, setInputSourceDevice + align against a ~130k-point keyframe (config[2] shape)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    from mrg_slam_amd import BatchMatcher, Context, SmallGicpHip, distance_filter, prefilter, prefilter_to_device, synth
    from mrg_slam_amd._lib import SMALL_GICP_HIP
    from mrg_slam_amd.registration import default_params

    which = sys.argv[1] if len(sys.argv) > 1 else "batch"
    ctx = Context(0)
    scene = synth.street_scene()
    poses = synth.arc_trajectory(5)
    raw = [synth.synth_lidar(scene, poses[k], "VLP64", synth.BASE_SEED + k) for k in range(5)]
    if which == "batch":
        scans = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw]
        gp = default_params(SMALL_GICP_HIP)
        gp.transformation_epsilon = 0.1
        gb = BatchMatcher(gp, ctx)
        gt = gb.add_target(scans[0])
        for b in range(32):
            gb.add_pair(gt, scans[1 + b % 4], synth.warm_guess(np.linalg.inv(poses[0]) @ poses[1 + b % 4], b))
        gb.align(-1.0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r = gb.align(-1.0)
        ctx.synchronize()
        print("batch ms", 1e3 * (time.perf_counter() - t0) / 3, "converged", int(r["converged"].sum()), "iterations", r["iterations"].mean())
    elif which == "frame130":  # BASELINE config[2] shape: ~130k-point frames (distance filter only) against a ~130k-point keyframe, clouds resident
        scans = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw]
        dev = [torch.from_numpy(s).to("cuda:0") for s in scans]
        odo = SmallGicpHip(transformation_epsilon=0.1, ctx=ctx)
        odo.setInputTarget(scans[0])
        tf = []
        for k in (1, 2, 3, 4, 1, 2, 3, 4, 1, 2):
            ctx.synchronize()
            t1 = time.perf_counter()
            odo.setInputSourceDevice(dev[k].data_ptr(), len(scans[k]))
            t2 = time.perf_counter()
            odo.align(synth.warm_guess(np.linalg.inv(poses[0]) @ poses[k], k))
            tf.append((time.perf_counter() - t1, t2 - t1))
        print("frame130 ms", 1e3 * float(np.median([a for a, _ in tf[2:]])), "of which setInputSource", 1e3 * float(np.median([b for _, b in tf[2:]])), "points", len(scans[1]), "iterations", odo.getFinalNumIteration())
    else:
        kf = prefilter(raw[0], ctx=ctx)
        dbuf = torch.empty((len(raw[1]) + 1000, 4), dtype=torch.float32, device="cuda:0")
        odo = SmallGicpHip(transformation_epsilon=0.1, ctx=ctx)
        odo.setInputTarget(kf)
        tf = []
        for k in (1, 2, 3, 4, 1, 2, 3, 4):
            ctx.synchronize()
            t1 = time.perf_counter()
            m = prefilter_to_device(raw[k], dbuf.data_ptr(), len(raw[k]), ctx=ctx)
            t2 = time.perf_counter()
            odo.setInputSourceDevice(dbuf.data_ptr(), m)
            odo.align(synth.warm_guess(np.linalg.inv(poses[0]) @ poses[k], k))
            tf.append((time.perf_counter() - t1, t2 - t1))
        print("frame ms", 1e3 * float(np.median([a for a, _ in tf[2:]])), "of which prefilter", 1e3 * float(np.median([b for _, b in tf[2:]])), "points", m, "iterations", odo.getFinalNumIteration())


if __name__ == "__main__":
    main()
