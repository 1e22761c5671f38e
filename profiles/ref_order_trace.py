"""Three single 130k-point registrations in reference order (for a kernel trace: rocprofv3 --kernel-trace --stats -- python3 profiles/ref_order_trace.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from mrg_slam_amd import Context, NdtHip, distance_filter, synth
from mrg_slam_amd._lib import lib
scene, poses, raw = bench.make_workload(256, 256, 0, "distance")
ctx = Context(0)
scans = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw[:2]]
dev = [torch.from_numpy(s).cuda() for s in scans]
lib().mrgfe_dbg_set_ndt_reference_order(1)
reg = NdtHip(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, ctx=ctx)
for rep in range(3):
    t0 = time.perf_counter()
    reg.setInputTargetDevice(dev[0].data_ptr(), len(scans[0])); reg.setInputSourceDevice(dev[1].data_ptr(), len(scans[1])); reg.align(synth.warm_guess(synth.rel_pose(poses[0], poses[1]), 0))
    print("ms", 1e3 * (time.perf_counter() - t0), "evals", reg.evals, file=sys.stderr)
