#!/usr/bin/env python3
"""BASELINE config[1] shape (256 distinct VLP-64 scan pairs, ~129k points, NDT_OMP DIRECT7 res 1.0 eps 0.1) against the oracle for the workloads of OTHER ranks —
every rank of `bench.py --gpus N` drives its own street (scene seed 1234 + rank), and bench.py's line checks rank 0's 256 pairs only:
    python3 profiles/config1_parity_ranks.py [ranks=1,2,3] > gpurun_out/config1_parity_ranks.json
Per rank: pairs within the 1e-4 m / 1e-4 rad bar, bit-identical transformations, equal iteration counts and convergence flags; every eighth pair starts from the
identity (cold) instead of the perturbed true motion."""
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
from mrg_slam_amd import BatchMatcher, Context, distance_filter, synth  # noqa: E402
from mrg_slam_amd._lib import NDT_HIP, SEARCH  # noqa: E402
from mrg_slam_amd.registration import default_params, result_matrix  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ranks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]
B = 256
ctx = Context(0)
prm = default_params(NDT_HIP)
prm.transformation_epsilon = 0.1
prm.maximum_iterations = 64
prm.resolution = 1.0
prm.nn_search_method = SEARCH["DIRECT7"]
out = {"workload": "256 distinct pairs per rank, scan k -> scan k + 1 of that rank's street, warm guesses (seed 1000 rank + b), every eighth pair from the identity", "ranks": {}}
tot = {"pairs": 0, "bit_identical": 0, "within_bar": 0, "same_iterations_and_convergence": 0, "max_dt_m": 0.0, "max_dr_rad": 0.0}
for rank in ranks:
    t0 = time.time()
    scene, poses, raw = bench.make_workload(B, B, rank, "distance")
    host = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw]
    dev = [torch.from_numpy(s).to("cuda:0") for s in host]
    rels = [synth.rel_pose(poses[k], poses[k + 1]) for k in range(B)]
    guesses = np.stack([np.eye(4) if b % 8 == 7 else synth.warm_guess(rels[b], 1000 * rank + b) for b in range(B)])
    bm = BatchMatcher(prm, ctx)
    bm.add_device([dev[k].data_ptr() for k in range(B)], [len(host[k]) for k in range(B)], np.arange(B, dtype=np.int32), [dev[k + 1].data_ptr() for k in range(B)],
                  [len(host[k + 1]) for k in range(B)], guesses)
    res = bm.align()

    def oracle_pair(b):
        o = orc.Ndt(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, num_threads=2)
        o.setInputTarget(host[b])
        o.setInputSource(host[b + 1])
        o.align(guesses[b])
        return o.getFinalTransformation(), bool(o.hasConverged()), int(o.getFinalNumIteration())

    with ThreadPoolExecutor(max(1, min(16, (os.cpu_count() or 2) // 2))) as ex:
        ora = list(ex.map(oracle_pair, range(B)))
    t = {"pairs": B, "bit_identical": 0, "within_bar": 0, "same_iterations_and_convergence": 0, "max_dt_m": 0.0, "max_dr_rad": 0.0, "cold_pairs": B // 8}
    for b in range(B):
        Th, (To, oc, oi) = result_matrix(res[b]), ora[b]
        same = np.array_equal(Th, To)
        dt = float(np.linalg.norm(Th[:3, 3] - To[:3, 3]))
        dr = 0.0 if same else float(synth.rotation_angle(Th[:3, :3], To[:3, :3]))
        t["bit_identical"] += int(same)
        t["within_bar"] += int(dt <= 1e-4 and dr <= 1e-4)
        t["same_iterations_and_convergence"] += int(int(res[b]["iterations"]) == oi and bool(res[b]["converged"]) == oc)
        t["max_dt_m"], t["max_dr_rad"] = max(t["max_dt_m"], dt), max(t["max_dr_rad"], dr)
    t["mean_iterations"] = float(np.mean(res["iterations"]))
    t["seconds"] = time.time() - t0
    out["ranks"][str(rank)] = t
    for k in ("pairs", "bit_identical", "within_bar", "same_iterations_and_convergence"):
        tot[k] += t[k]
    tot["max_dt_m"], tot["max_dr_rad"] = max(tot["max_dt_m"], t["max_dt_m"]), max(tot["max_dr_rad"], t["max_dr_rad"])
    print(f"[config1_parity_ranks] rank {rank}: {t}", file=sys.stderr)
out["total"] = tot
print(json.dumps(out))
