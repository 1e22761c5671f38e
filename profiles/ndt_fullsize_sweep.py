#!/usr/bin/env python3
"""NDT at the stated size (129k-point VLP-64 pairs) over the parameters bench.py's workload fixes — every neighbourhood of `reg_nn_search_method`
(DIRECT1 / DIRECT7 / DIRECT26 / KDTREE, registrations.cpp:136-146), resolutions 0.5 / 1 / 2 m, eps 0.1 / 0.01, warm and identity guesses — single registrations
against the reference-order oracle:   python3 profiles/ndt_fullsize_sweep.py [pairs=12] > gpurun_out/ndt_fullsize_sweep.json"""
import itertools
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from mrg_slam_amd import Context, NdtHip, PclNdtHip, distance_filter, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = Context(0)
scene, poses, raw = bench.make_workload(256, 256, 0, "distance")
step = 256 // n_pairs
idx = [k * step for k in range(n_pairs)]
host = {k: distance_filter(raw[k], 0.1, 35.0, ctx=ctx) for k in set(idx) | {k + 1 for k in idx}}
dev = {k: torch.from_numpy(v).to("cuda:0") for k, v in host.items()}
cases = []
for (search, res, eps) in itertools.product(("DIRECT1", "DIRECT7", "DIRECT26", "KDTREE"), (0.5, 1.0, 2.0), (0.1, 0.01)):
    for j, k in enumerate(idx):
        rel = synth.rel_pose(poses[k], poses[k + 1])
        cases.append((search, res, eps, k, np.eye(4) if j % 4 == 3 else synth.warm_guess(rel, 8000 + k)))
# round 5: registration_method "NDT" = pcl::NormalDistributionsTransform (PCL_NDT_HIP): its one neighbourhood (radius search over the voxel centroids),
# the same resolutions, eps 0.1 (mrg_slam's value: ONE Newton iteration by PCL's rule) / 0.001 / 1e-5 (runs to the zero-step or iteration limit)
for (res, eps) in itertools.product((0.5, 1.0, 2.0), (0.1, 1e-3, 1e-5)):
    for j, k in enumerate(idx[:: max(1, len(idx) // 16)]):
        rel = synth.rel_pose(poses[k], poses[k + 1])
        cases.append(("PCL_NDT", res, eps, k, np.eye(4) if j % 4 == 3 else synth.warm_guess(rel, 8000 + k)))
t0 = time.time()
hip = []
regs = {}
for (search, res, eps, k, guess) in cases:
    key = (search, res, eps)
    if key not in regs:
        regs[key] = (PclNdtHip(resolution=res, transformation_epsilon=eps, maximum_iterations=64, ctx=ctx) if search == "PCL_NDT" else
                     NdtHip(resolution=res, transformation_epsilon=eps, maximum_iterations=64, search=search, ctx=ctx))
    r = regs[key]
    r.setInputTargetDevice(dev[k].data_ptr(), len(host[k]))
    r.setInputSourceDevice(dev[k + 1].data_ptr(), len(host[k + 1]))
    r.align(guess)
    hip.append((r.getFinalTransformation().copy(), bool(r.hasConverged()), int(r.getFinalNumIteration())))
t_hip = time.time() - t0


def oracle_case(c):
    search, res, eps, k, guess = c
    o = (orc.PclNdt(resolution=res, transformation_epsilon=eps, maximum_iterations=64, num_threads=2) if search == "PCL_NDT" else
         orc.Ndt(resolution=res, transformation_epsilon=eps, maximum_iterations=64, num_threads=2, search=search))
    o.setInputTarget(host[k])
    o.setInputSource(host[k + 1])
    o.align(guess)
    return o.getFinalTransformation(), bool(o.hasConverged()), int(o.getFinalNumIteration())


with ThreadPoolExecutor(max(1, min(16, (os.cpu_count() or 2) // 2))) as ex:
    ora = list(ex.map(oracle_case, cases))
out = {"workload": f"{n_pairs} pairs of ~129k points x 4 neighbourhoods x 3 resolutions x 2 eps, plus up to 16 pairs x 3 resolutions x 3 eps through PCL_NDT_HIP = {len(cases)} single registrations, every fourth from the identity", "by_search": {},
       "hip_seconds": t_hip, "seconds": None}
over = []
for (c, h, o) in zip(cases, hip, ora):
    t = out["by_search"].setdefault(c[0], {"alignments": 0, "bit_identical": 0, "within_bar": 0, "same_iterations_and_convergence": 0, "at_the_iteration_limit": 0, "max_dt_m": 0.0, "max_dr_rad": 0.0})
    same = np.array_equal(h[0], o[0])
    dt = float(np.linalg.norm(h[0][:3, 3] - o[0][:3, 3]))
    dr = 0.0 if same else float(synth.rotation_angle(h[0][:3, :3], o[0][:3, :3]))
    t["alignments"] += 1
    t["bit_identical"] += int(same)
    ok = dt <= 1e-4 and dr <= 1e-4
    t["within_bar"] += int(ok)
    t["same_iterations_and_convergence"] += int(h[1] == o[1] and h[2] == o[2])
    t["at_the_iteration_limit"] += int(h[2] >= 64 or o[2] >= 64)
    t["max_dt_m"], t["max_dr_rad"] = max(t["max_dt_m"], dt), max(t["max_dr_rad"], dr)
    if not ok:
        over.append({"search": c[0], "resolution": c[1], "eps": c[2], "pair": int(c[3]), "dt_m": dt, "dr_rad": dr, "iterations_hip": h[2], "iterations_oracle": o[2]})
out["over_bar"] = over
out["seconds"] = time.time() - t0
print(json.dumps(out))
