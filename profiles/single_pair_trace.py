"""Diagnostic: per-round timing of one 130k-point NDT registration (MRGFE_TRACE=1 python profiles/single_pair_trace.py)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from mrg_slam_amd import Context, NdtHip, distance_filter, synth  # noqa: E402

ctx = Context(0)
scene = synth.street_scene()
tgt, src, rel = synth.scan_pair(0, "VLP64", scene)
ft, fs = distance_filter(tgt, ctx=ctx), distance_filter(src, ctx=ctx)
dt, ds = torch.from_numpy(ft).cuda(), torch.from_numpy(fs).cuda()
reg = NdtHip(transformation_epsilon=0.1, ctx=ctx)
guess = synth.warm_guess(rel, 0)
for it in range(8):
    ctx.synchronize()
    t0 = time.perf_counter()
    reg.setInputTargetDevice(dt.data_ptr(), len(ft))
    t1 = time.perf_counter()
    reg.setInputSourceDevice(ds.data_ptr(), len(fs))
    reg.align(guess)
    t2 = time.perf_counter()
    print(f"setInputTarget {1e6 * (t1 - t0):.0f} us, align {1e6 * (t2 - t1):.0f} us, evaluations {reg.evals}, iterations {reg.getFinalNumIteration()}", file=sys.stderr)
