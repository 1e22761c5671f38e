import sys, time, numpy as np
sys.path.insert(0, ".")
import torch
from mrg_slam_amd import Context, prefilter_to_device, synth
from mrg_slam_amd._lib import lib
ctx = Context(0)
sc = synth.street_scene()
raw = synth.synth_lidar(sc, np.eye(4), "VLP64", synth.BASE_SEED)
buf = torch.empty((len(raw) + 16, 4), dtype=torch.float32, device="cuda:0")
for mode in (1, 0, 1, 0):
    lib().mrgfe_dbg_set_prefilter_device_driven(mode)
    ts = []
    for i in range(40):
        ctx.synchronize()
        t0 = time.perf_counter()
        m = prefilter_to_device(raw, buf.data_ptr(), buf.shape[0], ctx=ctx)
        ts.append(1e3 * (time.perf_counter() - t0))
    print("mode", mode, "median ms", round(float(np.median(ts[5:])), 4), "min", round(min(ts[5:]), 4), "points", m)
