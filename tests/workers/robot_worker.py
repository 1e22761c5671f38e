"""One robot of BASELINE config[4] as its own process (kitti_multirobot_processor.py:164-172 starts one SLAM instance per robot):
an odometry stream over VLP-64 scans — raw scan in host memory -> mrgfe_prefilter_device -> setInputSource / align against the keyframe,
keyframe switches as in ScanMatchingOdometryComponent::matching (scan_matching_odometry_component.cpp:195-350) — and then an inter-robot
loop-closure batch: the robot's last keyframe against 64 candidates taken from the OTHER robot's scans (loop_detector.cpp:104,126-145,
getFitnessScore(inf)).  Everything is repeated on the CPU oracle, sequentially; the process prints one JSON line with the differences.

    python tests/workers/robot_worker.py <robot 0|1> <frames> <out.json>
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    robot, frames, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch  # before libmrgfe: one HIP runtime in the process (device buffers of the filtered scans)

    from mrg_slam_amd import BatchMatcher, Context, NdtHip, prefilter_to_device, synth
    from mrg_slam_amd.odometry import ScanMatchingOdometry
    from mrg_slam_amd.registration import result_matrix
    from oracle import oracle as orc

    scene = synth.street_scene()
    # robot 0 drives the street forward from x = 0, robot 1 comes the other way from 1 m/scan x frames ahead of it (they meet in the middle)
    def pose_of(r, k):
        x = 1.0 * k if r == 0 else 1.0 * (frames - 1 - k) + 6.0
        yaw = np.deg2rad(1.5 * k) * (1 if r == 0 else -1) + (0.0 if r == 0 else np.pi)
        return synth.make_pose([x, 0.3 * (1 - 2 * r), 0.0], synth.rot_xyz(0.0, 0.0, yaw))

    raw = {r: [synth.synth_lidar(scene, pose_of(r, k), "VLP64", seed=20251003 + 1000 * r + k) for k in range(frames)] for r in (0, 1)}
    poses = {r: [pose_of(r, k) for k in range(frames)] for r in (0, 1)}
    ctx = Context(0)
    t_start = time.time()

    # ---- odometry stream on the GPU: device-resident filtered scans
    reg = NdtHip(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, ctx=ctx)
    def to_dev(cloud):  # the odometry object keeps the keyframe's buffer alive, the frame's lives until the next frame
        buf = torch.empty((len(cloud), 4), dtype=torch.float32, device="cuda:0")
        n = prefilter_to_device(cloud, buf.data_ptr(), len(cloud), ctx=ctx)
        return buf, n

    odo = ScanMatchingOdometry(reg, set_target=lambda c: reg.setInputTargetDevice(c[0].data_ptr(), c[1]), set_source=lambda c: reg.setInputSourceDevice(c[0].data_ptr(), c[1]),
                               downsample=to_dev)
    # the initial guess of a frame is prev_trans * msf_delta (:265-266); msf_delta = the robot's own odometry between the two frames
    # (enable_robot_odometry_init_guess, :226-262), here the true motion perturbed like the bench's warm guesses — an identity guess leaves
    # NDT with mrg_slam's parameters (steps clamped to 0.1 m, "converged" at the first shorter one) at the keyframe in a street canyon
    deltas = [np.eye(4)] + [synth.warm_guess(np.linalg.inv(poses[robot][k - 1]) @ poses[robot][k], 100 * robot + k) for k in range(1, frames)]
    gpu_odom, gpu_iters = [], []
    for k in range(frames):
        gpu_odom.append(odo.matching(0.1 * k, raw[robot][k], deltas[k]))
        gpu_iters.append(reg.getFinalNumIteration() if k else 0)
    t_odo = time.time() - t_start

    # ---- the same stream on the CPU oracle (host clouds through the oracle's prefilter chain)
    def cpu_prefilter(c):
        c = orc.distance_filter(c, 0.1, 35.0)
        c, _ = orc.voxelgrid(c, 0.1, 1)
        c, _ = orc.radius_outlier(c, 0.5, 2)
        return c

    filtered = {r: [cpu_prefilter(c) for c in raw[r]] for r in (0, 1)}
    oreg = orc.Ndt(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, num_threads=8)
    oodo = ScanMatchingOdometry(oreg)
    cpu_odom, cpu_iters = [], []
    for k in range(frames):
        cpu_odom.append(oodo.matching(0.1 * k, filtered[robot][k], deltas[k]))
        cpu_iters.append(oreg.getFinalNumIteration() if k else 0)
    odo_dt = max(float(np.linalg.norm(a[:3, 3].astype(np.float64) - b[:3, 3])) for a, b in zip(gpu_odom, cpu_odom))
    odo_dr = max(synth.rotation_angle(a.astype(np.float64), b.astype(np.float64)) for a, b in zip(gpu_odom, cpu_odom))

    # ---- inter-robot batch: this robot's last keyframe against 64 candidates from the other robot's scans
    other = 1 - robot
    kf_index = frames - 1  # the robot's newest scan plays the new keyframe (loop_detector.cpp:104)
    tgt = filtered[robot][kf_index]
    rng = np.random.default_rng(4242 + robot)
    cand = []
    for c in range(64):
        k = c % frames
        rel = np.linalg.inv(poses[robot][kf_index]) @ poses[other][k]
        guess = rel @ synth.make_pose(rng.normal(0, 0.25, 3) * [1, 1, 0.2], synth.rot_xyz(*np.deg2rad(rng.normal(0, 1.0, 3))))
        cand.append((k, guess))
    bm = BatchMatcher(transformation_epsilon=0.1, maximum_iterations=64, ctx=ctx)
    t_b = time.time()
    t = bm.add_target(tgt)
    for k, g in cand:
        bm.add_pair(t, filtered[other][k], g, key=1000 * other + k + 1)
    res = bm.align(float("inf"))
    t_batch = time.time() - t_b
    from mrg_slam_amd import loop_closure as lc

    gbest, gscore = lc.select_best(res)
    ob = orc.Ndt(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, num_threads=8)
    ob.setInputTarget(tgt)
    best, best_score, b_dt, b_dr, b_fit, b_mis = None, np.finfo(np.float64).max, 0.0, 0.0, 0.0, 0
    for i, (k, g) in enumerate(cand):
        ob.setInputSource(filtered[other][k])
        ob.align(g)
        To = ob.getFinalTransformation()
        score = ob.getFitnessScore(float("inf"))
        Tg = result_matrix(res[i]).astype(np.float64)
        settled = ob.hasConverged() and ob.getFinalNumIteration() <= 30
        if settled:
            b_dt = max(b_dt, float(np.linalg.norm(Tg[:3, 3] - To[:3, 3])))
            b_dr = max(b_dr, synth.rotation_angle(Tg, To.astype(np.float64)))
            b_fit = max(b_fit, abs(float(res[i]["fitness"]) - score) / max(score, 1e-12))
            b_mis += int(bool(res[i]["converged"]) != ob.hasConverged() or int(res[i]["iterations"]) != ob.getFinalNumIteration())
        if not ob.hasConverged() or score > best_score:
            continue
        best_score, best = score, i
    out = {"robot": robot, "frames": frames, "points_per_filtered_scan": float(np.mean([len(c) for c in filtered[robot]])), "keyframes_gpu": odo.keyframes, "keyframes_cpu": oodo.keyframes,
           "odometry_max_dt_m": odo_dt, "odometry_max_dr_rad": odo_dr, "odometry_same_iterations": gpu_iters == cpu_iters, "odometry_s": t_odo,
           "batch_candidates": len(cand), "batch_max_dt_m": b_dt, "batch_max_dr_rad": b_dr, "batch_max_rel_fitness_diff": b_fit, "batch_mismatches": b_mis,
           "batch_best_gpu": gbest, "batch_best_cpu": best, "batch_s": t_batch, "final_odom_error_vs_truth_m":
           float(np.linalg.norm((np.linalg.inv(poses[robot][0]) @ poses[robot][frames - 1])[:3, 3] - gpu_odom[-1][:3, 3].astype(np.float64)))}
    with open(out_path, "w") as f:
        json.dump(out, f)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
