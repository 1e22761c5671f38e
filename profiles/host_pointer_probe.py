"""The PCIe-inclusive rate of the headline workload (both clouds of every pair handed over as page-locked HOST pointers inside the step), one batch at a time
and two in flight:  python profiles/host_pointer_probe.py [pairs]   [pageable]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from mrg_slam_amd import BatchMatcher, Context, distance_filter, synth
from mrg_slam_amd._lib import NDT_HIP, SEARCH, lib
from mrg_slam_amd.registration import default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pageable = len(sys.argv) > 2 and sys.argv[2] == "pageable"  # the clouds stay ordinary host memory: the staging ring
scene, poses, raw = bench.make_workload(256, 256, 0, "distance")
ctx = Context(0)
scans = [np.ascontiguousarray(distance_filter(s, 0.1, 35.0, ctx=ctx)) for s in raw[: n + 1]]
prm = default_params(NDT_HIP); prm.transformation_epsilon, prm.maximum_iterations, prm.resolution, prm.nn_search_method = 0.1, 64, 1.0, SEARCH["DIRECT7"]
guesses = [np.eye(4) if k % 4 == 3 else synth.warm_guess(synth.rel_pose(poses[k], poses[k + 1]), k) for k in range(n)]
bms = [BatchMatcher(prm, ctx), BatchMatcher(prm, Context(0))]
nbytes = sum(scans[k].nbytes + scans[k + 1].nbytes for k in range(n))
if not pageable:
    for b in bms: b._ctx.set_zero_copy_uploads(True)
    for sc in scans: assert lib().mrgfe_pin_host_buffer(ctx._h, sc.ctypes.data_as(C.c_void_p), sc.nbytes) == 0
def fill(b):
    b.clear()
    for k in range(n): b.add_pair(b.add_target(scans[k]), scans[k + 1], guesses[k])
out = {"pairs": n, "MB_per_step": nbytes / 1e6, "host_memory": "pageable" if pageable else "page-locked", "env": {k: v for k, v in os.environ.items() if k.startswith("MRGFE_")}}
fill(bms[0]); ref = bms[0].align(); ctx.synchronize()
t0 = time.perf_counter(); fill(bms[0]); t_add = time.perf_counter() - t0; ctx.synchronize(); t_up = time.perf_counter() - t0; bms[0].align()
out["add_calls_ms"], out["uploads_landed_ms"], out["upload_GBps"] = 1e3 * t_add, 1e3 * t_up, nbytes / 1e9 / t_up
t0 = time.perf_counter()
for _ in range(4): fill(bms[0]); r = bms[0].align()
ctx.synchronize(); t = (time.perf_counter() - t0) / 4
out["one_at_a_time"] = {"ms_per_step": 1e3 * t, "alignments_per_s": n / t, "GBps": nbytes / 1e9 / t, "same": bool(np.array_equal(r["T"], ref["T"]))}
def submit(b): fill(b); b.align_async()
submit(bms[0]); bms[0].wait(); ctx.synchronize()
t0 = time.perf_counter(); live = [False, False]; steps = 8
for it in range(steps):
    k = it % 2
    if live[k]: r = bms[k].wait()
    submit(bms[k]); live[k] = True
for k in ((steps) % 2, (steps + 1) % 2):
    if live[k]: r = bms[k].wait()
t = (time.perf_counter() - t0) / steps
out["two_in_flight"] = {"ms_per_step": 1e3 * t, "alignments_per_s": n / t, "GBps": nbytes / 1e9 / t, "same": bool(np.array_equal(r["T"], ref["T"]))}
if not pageable:
    for b in bms: b._ctx.set_zero_copy_uploads(False)
    for sc in scans: lib().mrgfe_unpin_host_buffer(ctx._h, sc.ctypes.data_as(C.c_void_p))
print(json.dumps(out))
