"""Diagnostic: derivative evaluations per alignment and kernel variant on the bench workload (python profiles/mode_counts.py)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mrg_slam_amd import BatchMatcher, Context, distance_filter, synth
from mrg_slam_amd._lib import NDT_HIP, SEARCH
from mrg_slam_amd.registration import default_params
scene, poses, raw = bench.make_workload(256, 256, 0, "distance")
ctx = Context(0)
scans = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw]
dev = [torch.from_numpy(s).cuda() for s in scans]
rels = [np.linalg.inv(poses[k]) @ poses[k + 1] for k in range(256)]
prm = default_params(NDT_HIP); prm.transformation_epsilon = 0.1; prm.maximum_iterations = 64
bm = BatchMatcher(prm, ctx)
guesses = np.stack([np.eye(4) if b % 4 == 3 else synth.warm_guess(rels[b], b) for b in range(256)])
bm.add_device([d.data_ptr() for d in dev[:256]], [len(s) for s in scans[:256]], np.arange(256, dtype=np.int32), [d.data_ptr() for d in dev[1:257]], [len(s) for s in scans[1:257]], guesses)
res = bm.align()
n = np.mean([len(s) for s in scans[1:257]])
for m in range(3):
    p, nb = bm.pair_counts(m)
    print("mode", m, "evaluations per alignment", p / n / 256)
print("iterations", res["iterations"].mean(), "evaluations", res["evaluations"].mean(), "rounds", bm.rounds())
