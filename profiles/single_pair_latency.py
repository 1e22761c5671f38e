"""NDT single registration latency, host and device pointers (setInputTarget + setInputSource + align of one 130k-point pair): python3 profiles/single_pair_latency.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from mrg_slam_amd import Context, NdtHip, distance_filter, synth  # noqa: E402

ctx = Context(0)
scene = synth.street_scene()
tgt, src, rel = synth.scan_pair(0, "VLP64", scene)
ft, fs = distance_filter(tgt, ctx=ctx), distance_filter(src, ctx=ctx)
dt, ds = torch.from_numpy(ft).cuda(), torch.from_numpy(fs).cuda()
reg = NdtHip(transformation_epsilon=0.1, ctx=ctx)
guess = synth.warm_guess(rel, 0)
res = {}
for mode in ("host", "device"):
    lat, lt = [], []
    for it in range(40):
        ctx.synchronize()
        t0 = time.perf_counter()
        if mode == "host":
            reg.setInputTarget(ft)
            t1 = time.perf_counter()
            reg.setInputSource(fs)
        else:
            reg.setInputTargetDevice(dt.data_ptr(), len(ft))
            t1 = time.perf_counter()
            reg.setInputSourceDevice(ds.data_ptr(), len(fs))
        reg.align(guess)
        lat.append(time.perf_counter() - t0)
        lt.append(t1 - t0)
    res[mode] = (1e3 * float(np.median(lat[5:])), 1e3 * float(np.median(lt[5:])))
print("single pair ms (total, of which setInputTarget call): host pointers %.3f / %.3f, device pointers %.3f / %.3f, evaluations %d" % (*res["host"], *res["device"], reg.evals))
