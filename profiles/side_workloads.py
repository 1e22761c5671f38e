#!/usr/bin/env python3
"""Small fixed workload for kernel traces of the non-headline paths: rocprofv3 --kernel-trace --stats -- python3 profiles/side_workloads.py
GICP_HIP (3 full set-target/set-source/align cycles on one prefiltered VLP-64 pair), the prefilter chain (3 raw scans)
and calc_fitness_score (3 calls)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from mrg_slam_amd import Context, GicpHip, calc_fitness_score, prefilter, synth

    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    ctx = Context(0)
    scene = synth.street_scene()
    poses = synth.arc_trajectory(3)
    raw = [synth.synth_lidar(scene, poses[k], "VLP64", synth.BASE_SEED + k) for k in range(3)]
    rel = np.linalg.inv(poses[0]) @ poses[1]
    ft, fs = prefilter(raw[0], ctx=ctx), prefilter(raw[1], ctx=ctx)
    if which in ("all", "prefilter"):
        for k in range(3):
            prefilter(raw[k], ctx=ctx)
    if which in ("all", "gicp"):
        g = GicpHip(transformation_epsilon=0.01, ctx=ctx)
        for k in range(3):
            g.setInputTarget(ft)
            g.setInputSource(fs)
            g.align(rel @ synth.make_pose([0.3, -0.2, 0.05], synth.rot_z(0.03)))
        print("gicp iterations", g.getFinalNumIteration())
    if which == "gicp_full":  # ~130k-point clouds (distance filter only): BASELINE config[2] shape
        from mrg_slam_amd import distance_filter

        a, b = distance_filter(raw[0], 0.1, 35.0, ctx=ctx), distance_filter(raw[1], 0.1, 35.0, ctx=ctx)
        g = GicpHip(transformation_epsilon=0.1, ctx=ctx)
        for k in range(3):
            g.setInputTarget(a)
            g.setInputSource(b)
            g.align(rel @ synth.make_pose([0.3, -0.2, 0.05], synth.rot_z(0.03)))
        print("gicp iterations", g.getFinalNumIteration())
    if which == "odo_ndt":  # odometry frames: raw scan -> prefilter (result left in HBM) -> scan-to-keyframe NDT_HIP align, six frames
        import torch

        from mrg_slam_amd import NdtHip, prefilter_to_device

        o = NdtHip(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, ctx=ctx)
        o.setInputTarget(ft)
        buf = torch.empty((len(raw[1]) + 16, 4), dtype=torch.float32, device="cuda:0")
        for k in range(6):
            m = prefilter_to_device(raw[1 + k % 2], buf.data_ptr(), buf.shape[0], ctx=ctx)
            o.setInputSourceDevice(buf.data_ptr(), m)
            o.align(synth.warm_guess(np.linalg.inv(poses[0]) @ poses[1 + k % 2], k))
        print("iterations", o.getFinalNumIteration())
    if which == "mapcloud":  # map cloud of 200 prefiltered keyframes out of the HBM map store (map_cloud_generator.cpp:14-86), three times
        from mrg_slam_amd import MapCloudStore

        kf = [prefilter(raw[k % 3], ctx=ctx) for k in range(3)]
        K = 200
        kposes = [synth.make_pose([1.0 * k, 0.3 * k, 0.0], synth.rot_z(0.01 * k)) for k in range(K)]
        ms = MapCloudStore(ctx)
        for k in range(K):
            ms.add(k + 1, kf[k % 3])
        for _ in range(3):
            out = ms.generate(list(range(1, K + 1)), kposes, None, 0.1)
        print("map points", len(out))
    if which == "lc":  # loop-closure batch: one 130k-point keyframe against 64 candidates, getFitnessScore(inf) of each
        from mrg_slam_amd import BatchMatcher, distance_filter

        more = synth.arc_trajectory(5)
        sc = [distance_filter(synth.synth_lidar(scene, more[k], "VLP64", synth.BASE_SEED + k), 0.1, 35.0, ctx=ctx) for k in range(5)]
        bm = BatchMatcher(ctx=ctx, transformation_epsilon=0.1)
        t = bm.add_target(sc[0])
        for b in range(64):
            bm.add_pair(t, sc[1 + b % 4], synth.warm_guess(np.linalg.inv(more[0]) @ more[1 + b % 4], b))
        for k in range(2):
            res = bm.align(float("inf"))
        print("fitness", res["fitness"][:4])
    if which == "gicp_lc":  # loop-detection calls with GICP_HIP: one 130k-point keyframe, 32 candidates named by keyframe id
        from mrg_slam_amd import BatchMatcher, distance_filter
        from mrg_slam_amd._lib import GICP_HIP
        from mrg_slam_amd.registration import default_params

        more = synth.arc_trajectory(5)
        sc = [distance_filter(synth.synth_lidar(scene, more[k], "VLP64", synth.BASE_SEED + k), 0.1, 35.0, ctx=ctx) for k in range(5)]
        prm = default_params(GICP_HIP)
        prm.transformation_epsilon = 0.1
        bm = BatchMatcher(prm, ctx)
        for call in range(3):  # the first call fills the keyframe store, the others find the candidates (and covariances) resident
            bm.clear()
            t = bm.add_target(sc[0])
            for b in range(32):
                bm.add_pair(t, sc[1 + b % 4] if bm.has_cloud(100 + b) is None else None, synth.warm_guess(np.linalg.inv(more[0]) @ more[1 + b % 4], b), key=100 + b)
            res = bm.align(-1.0)
        print("converged", int(res["converged"].sum()), "store MB", bm.store_bytes() >> 20)
    if which in ("all", "fitness"):
        for k in range(3):
            print("fitness", calc_fitness_score(ft, fs, rel, 2.0, ctx=ctx))
    ctx.synchronize()


if __name__ == "__main__":
    main()
