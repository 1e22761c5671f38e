"""What the reference's summation order costs (MRGFE_NDT_REFERENCE_ORDER): config[1] pairs, a single registration and batches of 16 / 64 / 256, default order against reference order."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from mrg_slam_amd import BatchMatcher, Context, NdtHip, distance_filter, synth
from mrg_slam_amd._lib import NDT_HIP, SEARCH, lib
from mrg_slam_amd.registration import default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene, poses, raw = bench.make_workload(256, 256, 0, "distance")
ctx = Context(0)
scans = [distance_filter(s, 0.1, 35.0, ctx=ctx) for s in raw[: n + 1]]
dev = [torch.from_numpy(s).cuda() for s in scans]
prm = default_params(NDT_HIP); prm.transformation_epsilon, prm.maximum_iterations, prm.resolution, prm.nn_search_method = 0.1, 64, 1.0, SEARCH["DIRECT7"]
guesses = [np.eye(4) if k % 4 == 3 else synth.warm_guess(synth.rel_pose(poses[k], poses[k + 1]), k) for k in range(n)]
out = {}
for mode in (0, 1):
    lib().mrgfe_dbg_set_ndt_reference_order(mode)
    reg = NdtHip(resolution=1.0, transformation_epsilon=0.1, maximum_iterations=64, ctx=ctx)
    lat = []
    for rep in range(6):
        ctx.synchronize(); t0 = time.perf_counter()
        reg.setInputTargetDevice(dev[0].data_ptr(), len(scans[0])); reg.setInputSourceDevice(dev[1].data_ptr(), len(scans[1])); reg.align(guesses[0])
        lat.append(1e3 * (time.perf_counter() - t0))
    rec = {"single_pair_ms": float(np.median(lat[1:])), "evaluations": reg.evals}
    for b in (16, 64, n):
        if b > n: continue
        bm = BatchMatcher(prm, ctx)
        args = ([dev[k].data_ptr() for k in range(b)], [len(scans[k]) for k in range(b)], np.arange(b, dtype=np.int32), [dev[k + 1].data_ptr() for k in range(b)], [len(scans[k + 1]) for k in range(b)], np.stack(guesses[:b]))
        ts = []
        for rep in range(3):
            bm.clear(); bm.add_device(*args); ctx.synchronize(); t0 = time.perf_counter(); r = bm.align(); ts.append(1e3 * (time.perf_counter() - t0))
        rec[f"batch_{b}_ms"] = float(np.median(ts[1:])); rec[f"batch_{b}_T_sha"] = bench.sha16([r["T"]])
    out["reference_order" if mode else "default_order"] = rec
lib().mrgfe_dbg_set_ndt_reference_order(0)
print(json.dumps(out))
