#!/usr/bin/env python3
"""BASELINE config[2] at its stated size (scan-to-keyframe GICP on ~130k-point VLP-64 scans) against the oracle over MANY frames — bench.py's line
holds 12 per method:   python3 profiles/config2_parity.py [frames=48] > gpurun_out/config2_parity.json
Per method (SMALL_GICP_HIP = the YAML default, GICP_HIP = fast_gicp, VGICP_HIP = fast_vgicp): a keyframe every eight scans, every frame aligned to its keyframe
from a warm guess and from a loop-closure-sized one (0.5 m / 2 deg); counts of bit-identical transformations, frames within the 1e-4 m / 1e-4 rad bar, equal
iteration counts and convergence flags."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from mrg_slam_amd import Context, GicpHip, SmallGicpHip, VgicpHip, distance_filter, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 48
ctx = Context(0)
scene, poses, raw = bench.make_workload(256, 256, 0, "distance")
host = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw[: n_frames + 8]]
dev = [torch.from_numpy(s).to("cuda:0") for s in host]
threads = min(32, os.cpu_count() or 8)
out = {"workload": f"{n_frames} frames of ~{int(np.mean([len(h) for h in host]))} points, keyframe every 8 scans, warm and loop-closure-sized guesses, max_correspondence_distance 2.0, k = 20, eps 0.1",
       "oracle_threads": threads, "methods": {}}


def rot_angle(Ra, Rb):
    return 0.0 if np.array_equal(Ra, Rb) else float(synth.rotation_angle(Ra, Rb))


for name, cls, ocls, kw in (("SMALL_GICP_HIP", SmallGicpHip, orc.SmallGicp, {}), ("GICP_HIP", GicpHip, orc.FastGicp, {}), ("VGICP_HIP", VgicpHip, orc.FastVgicp, {"resolution": 1.0})):
    t0 = time.time()
    reg = cls(transformation_epsilon=0.1, ctx=ctx, **kw)
    o = ocls(transformation_epsilon=0.1, num_threads=threads, **kw)
    tally = {"frames": 0, "bit_identical": 0, "within_bar": 0, "same_iterations_and_convergence": 0, "max_dt_m": 0.0, "max_dr_rad": 0.0, "outer_iterations": []}
    kf = -1
    for f in range(1, n_frames + 1):
        k = (f - 1) // 8 * 8
        if k != kf:
            kf = k
            reg.setInputTargetDevice(dev[k].data_ptr(), len(host[k]))
            o.setInputTarget(host[k])
        rel = synth.rel_pose(poses[k], poses[f])
        guesses = [synth.warm_guess(rel, 6000 + f)]
        if f % 2 == 0:
            guesses.append(synth.perturb_pose(rel, np.random.default_rng(4300 + f), sigma_t=(0.5, 0.5, 0.1), sigma_r_deg=(0.5, 0.5, 2.0)))
        reg.setInputSourceDevice(dev[f].data_ptr(), len(host[f]))
        o.setInputSource(host[f])
        for g in guesses:
            reg.align(g)
            o.align(g)
            Th, To = reg.getFinalTransformation(), o.getFinalTransformation()
            dt, dr = float(np.linalg.norm(Th[:3, 3] - To[:3, 3])), rot_angle(Th[:3, :3], To[:3, :3])
            tally["frames"] += 1
            tally["bit_identical"] += int(np.array_equal(Th, To))
            tally["within_bar"] += int(dt <= 1e-4 and dr <= 1e-4)
            tally["same_iterations_and_convergence"] += int(reg.getFinalNumIteration() == o.getFinalNumIteration() and bool(reg.hasConverged()) == bool(o.hasConverged()))
            tally["max_dt_m"], tally["max_dr_rad"] = max(tally["max_dt_m"], dt), max(tally["max_dr_rad"], dr)
            tally["outer_iterations"].append(int(reg.getFinalNumIteration()))
    its = tally.pop("outer_iterations")
    tally["outer_iterations_mean_max"] = [float(np.mean(its)), int(np.max(its))]
    tally["seconds"] = time.time() - t0
    out["methods"][name] = tally
    print(f"[config2_parity] {name}: {tally}", file=sys.stderr)
print(json.dumps(out))
