"""A scripted loop-closure session for the LoopDetector tests: one robot drives 1.25 laps of the 40 m ring road (its keyframes' graph
estimates drift off the truth), a second robot contributes a few keyframes near the start; the keyframes arrive one by one as "new
keyframes" the way LoopDetector::detect receives them (/root/reference/src/mrg_slam/loop_detector.cpp:14-38)."""
import numpy as np


def make_ring_session(n_keyframes=64, model="VLP16", laps=1.25, radius=40.0, prefilter=None, seed=11):
    from mrg_slam_amd import synth
    from mrg_slam_amd.loop_detector import Edge, KeyFrame

    scene = synth.loop_scene(radius=radius)
    rng = np.random.default_rng(seed)
    n_a = n_keyframes - 6
    kfs = []
    # robot a: around the ring
    truth = []
    for k in range(n_a):
        a = 2 * np.pi * laps * k / n_a
        truth.append(synth.make_pose([radius * np.cos(a), radius * np.sin(a), 0.0], synth.rot_z(a + np.pi / 2)))
    # robot b: six keyframes driving the first stretch the other way
    truth_b = []
    for k in range(6):
        a = 2 * np.pi * (0.10 - 0.02 * k)
        truth_b.append(synth.make_pose([(radius + 1.0) * np.cos(a), (radius + 1.0) * np.sin(a), 0.0], synth.rot_z(a - np.pi / 2)))
    scans = synth.synth_lidar_many(scene, truth + truth_b, model, [synth.BASE_SEED + 9000 + k for k in range(n_keyframes)], cache_tag=f"ringsession_{model}_{n_keyframes}")
    if prefilter is not None:
        scans = [prefilter(s) for s in scans]
    for r, (poses, name) in enumerate(((truth, "robot_a"), (truth_b, "robot_b"))):
        accum, drift = 0.0, np.eye(4)
        first = len(kfs)
        for k, T in enumerate(poses):
            if k:
                accum += float(np.linalg.norm(T[:3, 3] - poses[k - 1][:3, 3]))
                # odometry drift: a small random walk on the graph estimate
                drift = drift @ synth.make_pose(rng.normal(0, 0.02, 3) * [1, 1, 0.2], synth.rot_xyz(*np.deg2rad(rng.normal(0, 0.08, 3))))
            kf = KeyFrame(id=len(kfs) + 1, cloud=scans[len(kfs)], estimate=T @ drift, accum_distance=accum, slam_uuid=name, first_keyframe=(k == 0))
            if k:
                prev = kfs[-1]
                rel = np.linalg.inv(kf.estimate) @ prev.estimate  # this keyframe -> the previous one
                kf.prev_edge = Edge(kf, prev, rel)
                prev.next_edge = Edge(kf, prev, rel)  # "the one after it -> this keyframe": next_edge->from_keyframe is the later one
                kf.connected.add(prev.id)
                prev.connected.add(kf.id)
            kfs.append(kf)
        assert len(kfs) - first == len(poses)
    # arrival order: robot a's keyframes, with robot b's slipped in after the first lap has started
    order = list(range(n_a))
    for j in range(6):
        order.insert(20 + 2 * j, n_a + j)
    return kfs, order


def run_session(detector, kfs, order, group=1, batched=False):
    """Feeds the keyframes `group` at a time (the reference's detect() receives every keyframe added since the last optimisation, loop_detector.cpp:18-21);
    every detected loop becomes a graph edge (as mrg_slam_component does), so later calls see it.  batched: detect_batched instead of detect."""
    known, loops = [], []
    for g0 in range(0, len(order), group):
        new = [kfs[i] for i in order[g0:g0 + group]]
        found = detector.detect_batched(known, new) if batched else detector.detect(known, new)
        for lp in found:
            lp.key1.connected.add(lp.key2.id)
            lp.key2.connected.add(lp.key1.id)
        loops += found
        known += new
    return loops
