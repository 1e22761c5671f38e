#!/usr/bin/env python3
"""Side measurements quoted in DESIGN.md §6 (not the headline metric): run on the GPU box, prints one JSON object.
Lives under tests/ because it times the CPU oracle beside the GPU path (only tests/, smoke() and bench.py may use oracle/).

* host-pointer (PCIe-inclusive) batched NDT throughput: clouds handed over as host buffers every step
* single-pair NDT latency through one pcl::Registration-style handle
* GICP_HIP vs the CPU oracle (restated fast_gicp) on one VLP-64 pair
* prefilter chain (distance + 0.1 m voxel + radius outlier) per raw VLP-64 scan, GPU vs CPU oracle
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from mrg_slam_amd import BatchMatcher, Context, GicpHip, NdtHip, distance_filter, prefilter, synth
    from mrg_slam_amd._lib import NDT_HIP, SEARCH
    from mrg_slam_amd.registration import default_params
    from oracle import oracle as orc

    ctx = Context(0)
    scene = synth.street_scene()
    poses = synth.arc_trajectory(5)
    raw = [synth.synth_lidar(scene, poses[k], "VLP64", synth.BASE_SEED + k) for k in range(5)]
    scans = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw]
    rels = [np.linalg.inv(poses[k]) @ poses[k + 1] for k in range(4)]
    out = {"points_per_scan": float(np.mean([len(s) for s in scans]))}

    # ---- host-pointer batched NDT -------------------------------------------------------------------------------
    prm = default_params(NDT_HIP)
    prm.transformation_epsilon, prm.maximum_iterations, prm.resolution, prm.nn_search_method = 0.1, 64, 1.0, SEARCH["DIRECT7"]
    bm = BatchMatcher(prm, ctx)
    B = 64
    pairs = [(b % 4, b % 4 + 1, synth.warm_guess(rels[b % 4], b)) for b in range(B)]

    def step_host():
        bm.clear()
        for ti, si, g in pairs:
            t = bm.add_target(scans[ti])
            bm.add_pair(t, scans[si], g)
        return bm.align()

    step_host()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step_host()
    ctx.synchronize()
    out["ndt_host_pointer_alignments_per_s"] = 3 * B / (time.perf_counter() - t0)
    out["ndt_host_pointer_note"] = f"{B} pairs per step, both clouds of every pair copied host->device (pinned staging) inside the timed region"

    # ---- loop-closure style batch: one new keyframe (target) against B candidate clouds, getFitnessScore(inf) per pair --------
    bm.clear()
    t = bm.add_target(scans[0])
    for b in range(B):
        bm.add_pair(t, scans[1 + b % 4], synth.warm_guess(np.linalg.inv(poses[0]) @ poses[1 + b % 4], b))
    lc = {}
    for name, rng in (("align_only_ms", -1.0), ("align_plus_fitness_inf_ms", float("inf")), ("align_plus_fitness_2m2_ms", 2.0)):
        bm.align(rng)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            res = bm.align(rng)
        ctx.synchronize()
        lc[name] = 1e3 * (time.perf_counter() - t0) / 3
    lc["pairs"] = B
    lc["fitness_sample"] = [float(res["fitness"][0]), float(res["fitness"][1])]
    out["loop_closure_batch"] = lc
    if "--lc-only" in sys.argv:
        print(json.dumps(out))
        return

    # ---- GICP loop-closure batch: 32 candidate clouds against one 130k-point keyframe -------------------------------
    def gicp_batch(method):
        gp = default_params(method)
        gp.transformation_epsilon = 0.1
        gb = BatchMatcher(gp, ctx)
        gt = gb.add_target(scans[0])
        for b in range(32):
            gb.add_pair(gt, scans[1 + b % 4], synth.warm_guess(np.linalg.inv(poses[0]) @ poses[1 + b % 4], b))
        gb.align(-1.0)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            r = gb.align(-1.0)
        ctx.synchronize()
        return 1e3 * (time.perf_counter() - t0) / 2, int(np.sum(r["converged"]))

    from mrg_slam_amd._lib import GICP_HIP, SMALL_GICP_HIP, VGICP_HIP
    ms_f, conv_f = gicp_batch(GICP_HIP)
    ms_s, conv_s = gicp_batch(SMALL_GICP_HIP)
    out["gicp_batch_32x130k"] = {"gicp_hip_ms": ms_f, "small_gicp_hip_ms": ms_s, "converged": [conv_f, conv_s]}

    # ---- one loop-detection call end to end, host clouds in: new keyframe + 32 recurring candidates -------------------
    # (clear, add_target, add_pair x 32, align): candidates handed over as host clouds every call vs named by keyframe id
    def lc_call(method, keyed, calls=3):
        lp = default_params(method)
        lp.transformation_epsilon, lp.maximum_iterations, lp.resolution, lp.nn_search_method = 0.1, 64, 1.0, SEARCH["DIRECT7"]
        lb = BatchMatcher(lp, ctx)
        times = []
        for c in range(calls + 1):
            ctx.synchronize()
            t0 = time.perf_counter()
            lb.clear()
            lt = lb.add_target(scans[0])
            for b in range(32):
                k = 1 + b % 4
                guess = synth.warm_guess(np.linalg.inv(poses[0]) @ poses[k], b)
                if keyed:
                    lb.add_pair(lt, scans[k] if lb.has_cloud(100 + b) is None else None, guess, key=100 + b)
                else:
                    lb.add_pair(lt, scans[k], guess)
            lb.align(-1.0)
            ctx.synchronize()
            times.append(1e3 * (time.perf_counter() - t0))
        return float(np.median(times[1:]))

    out["loop_detection_call_32_candidates_ms"] = {
        "ndt_host_clouds": lc_call(NDT_HIP, False), "ndt_keyframe_store": lc_call(NDT_HIP, True),
        "gicp_host_clouds": lc_call(GICP_HIP, False), "gicp_keyframe_store": lc_call(GICP_HIP, True),
        "small_gicp_host_clouds": lc_call(SMALL_GICP_HIP, False), "small_gicp_keyframe_store": lc_call(SMALL_GICP_HIP, True)}
    if "--gicp-batch-only" in sys.argv:
        print(json.dumps(out))
        return

    # ---- single pair latency ----------------------------------------------------------------------------------------
    reg = NdtHip(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, ctx=ctx)
    lat = []
    for _ in range(8):
        ctx.synchronize()
        t1 = time.perf_counter()
        reg.setInputTarget(scans[0])
        reg.setInputSource(scans[1])
        reg.align(pairs[0][2])
        lat.append(time.perf_counter() - t1)
    out["ndt_single_pair_latency_ms_host_pointers"] = 1e3 * float(np.median(lat[2:]))
    out["ndt_single_pair_evaluations"] = reg.evals

    # ---- GICP ---------------------------------------------------------------------------------------------------------
    ft, fs = prefilter(raw[0], ctx=ctx), prefilter(raw[1], ctx=ctx)
    g = GicpHip(transformation_epsilon=0.1, ctx=ctx)
    guess = synth.warm_guess(rels[0], 0)
    tg = []
    for _ in range(4):
        ctx.synchronize()
        t1 = time.perf_counter()
        g.setInputTarget(ft)
        g.setInputSource(fs)
        g.align(guess)
        tg.append(time.perf_counter() - t1)
    cores = min(32, os.cpu_count() or 1)
    o = orc.FastGicp(transformation_epsilon=0.1, num_threads=cores)
    t1 = time.perf_counter()
    o.setInputTarget(ft)
    o.setInputSource(fs)
    o.align(guess)
    tcpu = time.perf_counter() - t1
    Tg, To = g.getFinalTransformation(), o.getFinalTransformation()
    out["gicp"] = {"points": [len(ft), len(fs)], "gpu_ms": 1e3 * float(np.median(tg[1:])), "cpu_oracle_ms": 1e3 * tcpu, "cpu_threads": cores,
                   "dt_m": float(np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3])), "dr_rad": synth.rotation_angle(Tg, To),
                   "iterations": [g.getFinalNumIteration(), o.getFinalNumIteration()]}

    # ---- GICP on the full-size (distance-filtered, ~130k point) scans: BASELINE config[2] shape -------------------------
    g2 = GicpHip(transformation_epsilon=0.1, ctx=ctx)
    tg = []
    for _ in range(3):
        ctx.synchronize()
        t1 = time.perf_counter()
        g2.setInputTarget(scans[0])
        g2.setInputSource(scans[1])
        g2.align(guess)
        tg.append(time.perf_counter() - t1)
    o2 = orc.FastGicp(transformation_epsilon=0.1, num_threads=cores)
    t1 = time.perf_counter()
    o2.setInputTarget(scans[0])
    o2.setInputSource(scans[1])
    o2.align(guess)
    tcpu = time.perf_counter() - t1
    Tg, To = g2.getFinalTransformation(), o2.getFinalTransformation()
    out["gicp_full_size"] = {"points": [len(scans[0]), len(scans[1])], "gpu_ms": 1e3 * float(np.median(tg[1:])), "cpu_oracle_ms": 1e3 * tcpu, "cpu_threads": cores,
                             "dt_m": float(np.linalg.norm(Tg[:3, 3].astype(np.float64) - To[:3, 3])), "dr_rad": synth.rotation_angle(Tg, To),
                             "iterations": [g2.getFinalNumIteration(), o2.getFinalNumIteration()]}

    # ---- one odometry frame: raw scan in host memory -> prefilter -> scan-to-keyframe NDT against a resident keyframe ----
    import torch

    from mrg_slam_amd import prefilter_to_device

    kf = prefilter(raw[0], ctx=ctx)
    odo = NdtHip(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, ctx=ctx)
    odo.setInputTarget(kf)
    dbuf = torch.empty((len(raw[1]), 4), dtype=torch.float32, device="cuda:0")
    prev = np.eye(4)
    tf = []
    for k in (1, 2, 3, 4, 1, 2, 3, 4):
        ctx.synchronize()
        t1 = time.perf_counter()
        m = prefilter_to_device(raw[k], dbuf.data_ptr(), len(raw[k]), ctx=ctx)
        odo.setInputSourceDevice(dbuf.data_ptr(), m)
        odo.align(synth.warm_guess(np.linalg.inv(poses[0]) @ poses[k], k))
        tf.append(time.perf_counter() - t1)
    out["odometry_frame_ms"] = {"raw_points": len(raw[1]), "filtered_points": int(m), "prefilter_plus_scan_to_keyframe_ndt_ms": 1e3 * float(np.median(tf[2:]))}
    from mrg_slam_amd import SmallGicpHip

    odo_g = SmallGicpHip(transformation_epsilon=0.1, ctx=ctx)  # the YAML default registration (config/mrg_slam.yaml:100)
    odo_g.setInputTarget(kf)
    tf = []
    for k in (1, 2, 3, 4, 1, 2, 3, 4):
        ctx.synchronize()
        t1 = time.perf_counter()
        m = prefilter_to_device(raw[k], dbuf.data_ptr(), len(raw[k]), ctx=ctx)
        odo_g.setInputSourceDevice(dbuf.data_ptr(), m)
        odo_g.align(synth.warm_guess(np.linalg.inv(poses[0]) @ poses[k], k))
        tf.append(time.perf_counter() - t1)
    out["odometry_frame_ms"]["prefilter_plus_scan_to_keyframe_small_gicp_ms"] = 1e3 * float(np.median(tf[2:]))
    from mrg_slam_amd import VgicpHip

    odo_v = VgicpHip(resolution=1.0, transformation_epsilon=0.1, ctx=ctx)  # FAST_VGICP / the reference's FAST_VGICP_CUDA slot
    odo_v.setInputTarget(kf)
    tf = []
    for k in (1, 2, 3, 4, 1, 2, 3, 4):
        ctx.synchronize()
        t1 = time.perf_counter()
        m = prefilter_to_device(raw[k], dbuf.data_ptr(), len(raw[k]), ctx=ctx)
        odo_v.setInputSourceDevice(dbuf.data_ptr(), m)
        odo_v.align(synth.warm_guess(np.linalg.inv(poses[0]) @ poses[k], k))
        tf.append(time.perf_counter() - t1)
    out["odometry_frame_ms"]["prefilter_plus_scan_to_keyframe_vgicp_ms"] = 1e3 * float(np.median(tf[2:]))
    ms_v, conv_v = gicp_batch(VGICP_HIP)
    out["gicp_batch_32x130k"]["vgicp_hip_ms"] = ms_v
    out["gicp_batch_32x130k"]["converged"].append(conv_v)

    # ---- BASELINE config[4] shape on one GPU: two robots' odometry streams (threads / contexts A, B) while a loop-closure
    #      batch stream (context C: 64 NDT candidates per call, keyframe store) keeps the GPU busy ---------------------------
    import threading

    def odo_stream(cx, frames, lat):
        o = NdtHip(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, ctx=cx)
        o.setInputTarget(kf)
        buf = torch.empty((len(raw[1]), 4), dtype=torch.float32, device="cuda:0")
        for f in range(frames):
            k = 1 + f % 4
            t1 = time.perf_counter()
            mm = prefilter_to_device(raw[k], buf.data_ptr(), len(raw[k]), ctx=cx)
            o.setInputSourceDevice(buf.data_ptr(), mm)
            o.align(synth.warm_guess(np.linalg.inv(poses[0]) @ poses[k], k))
            lat.append(time.perf_counter() - t1)

    def lc_stream(cx, stop, count):
        lb = BatchMatcher(prm, cx)
        while not stop.is_set():
            lb.clear()
            lt = lb.add_target(scans[0])
            for b in range(64):
                k = 1 + b % 4
                lb.add_pair(lt, scans[k] if lb.has_cloud(500 + b) is None else None, synth.warm_guess(np.linalg.inv(poses[0]) @ poses[k], b), key=500 + b)
            lb.align(-1.0)
            count.append(1)

    def concurrent(high_priority, reserve_cus=0):
        cA, cB, cC = Context(0, high_priority), Context(0, high_priority), Context(0, reserve_cus=reserve_cus)
        alone = []
        odo_stream(cA, 24, alone)
        latA, latB, calls, stop = [], [], [], threading.Event()
        tl = threading.Thread(target=lc_stream, args=(cC, stop, calls))
        tl.start()
        time.sleep(0.05)
        t_begin = time.perf_counter()
        ta, tb = threading.Thread(target=odo_stream, args=(cA, 240, latA)), threading.Thread(target=odo_stream, args=(cB, 240, latB))
        ta.start(); tb.start(); ta.join(); tb.join()
        span = time.perf_counter() - t_begin
        stop.set()
        tl.join()
        both = np.array(latA[4:] + latB[4:])
        return {"odometry_frame_alone_ms": 1e3 * float(np.median(alone[4:])), "odometry_frame_median_ms": 1e3 * float(np.median(both)),
                "odometry_frame_p95_ms": 1e3 * float(np.percentile(both, 95)), "odometry_frames_per_s_both_robots": 480 / span,
                "loop_closure_calls_per_s_64_candidates": len(calls) / span}

    out["two_odometry_streams_plus_loop_closure_stream"] = concurrent(False)
    out["two_high_priority_odometry_streams_plus_loop_closure_stream"] = concurrent(True)
    # the loop-closure context confined to 224 / 192 of the 256 compute units (mrgfe_ctx_create_reserving): the odometry launches find the rest free
    out["two_high_priority_odometry_streams_plus_loop_closure_stream_reserving_32_cus"] = concurrent(True, 32)
    out["two_high_priority_odometry_streams_plus_loop_closure_stream_reserving_64_cus"] = concurrent(True, 64)
    out["two_odometry_streams_plus_loop_closure_stream_reserving_32_cus"] = concurrent(False, 32)

    # ---- map cloud of 200 prefiltered keyframes (6.5 M points): host clouds every call vs the HBM map store -------------
    from mrg_slam_amd import KeyFrameSnapshot, MapCloudGenerator, MapCloudStore

    kf_clouds = [prefilter(raw[k % 5], ctx=ctx) for k in range(5)]
    K = 200
    kposes = [synth.make_pose([1.0 * k, 0.3 * k, 0.0], synth.rot_z(0.01 * k)) for k in range(K)]
    gen, mstore = MapCloudGenerator(ctx), MapCloudStore(ctx)
    for k in range(K):
        mstore.add(k + 1, kf_clouds[k % 5])
    snaps = [KeyFrameSnapshot(kposes[k], kf_clouds[k % 5], k == 0) for k in range(K)]
    tm = {}
    for name, fn in (("host_clouds_ms", lambda: gen.generate(snaps, 0.1)), ("map_store_ms", lambda: mstore.generate(list(range(1, K + 1)), kposes, None, 0.1)),
                     ("map_store_view_ms", lambda: mstore.generate(list(range(1, K + 1)), kposes, None, 0.1, copy=False))):
        fn()
        ts = []
        for _ in range(3):
            ctx.synchronize()
            t1 = time.perf_counter()
            res_map = fn()
            ts.append(time.perf_counter() - t1)
        tm[name] = 1e3 * float(np.median(ts))
    tm.update(keyframes=K, points_in=int(sum(len(kf_clouds[k % 5]) for k in range(K))), points_out=int(len(res_map)))
    out["map_cloud_200_keyframes"] = tm

    # ---- prefilter chain --------------------------------------------------------------------------------------------
    tp = []
    for _ in range(5):
        ctx.synchronize()
        t1 = time.perf_counter()
        f = prefilter(raw[2], ctx=ctx)
        tp.append(time.perf_counter() - t1)
    t1 = time.perf_counter()
    e = orc.distance_filter(raw[2], 0.1, 35.0)
    e, _ = orc.voxelgrid(e, 0.1, 1)
    e, _ = orc.radius_outlier(e, 0.5, 2)
    tcpu = time.perf_counter() - t1
    out["prefilter"] = {"raw_points": len(raw[2]), "out_points": len(f), "gpu_ms_host_pointers": 1e3 * float(np.median(tp[1:])), "cpu_oracle_ms_1_thread": 1e3 * tcpu,
                        "exact": bool(f.shape == e.shape and (f == e).all())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
