"""Side measurement: the 128 pairs of a bench step split over N contexts (= N HIP streams) driven by N host
threads, so one half-batch's straggler rounds overlap the other's full launches.  python profiles/two_streams.py [N]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mrg_slam_amd import BatchMatcher, Context, distance_filter, synth
from mrg_slam_amd._lib import NDT_HIP, SEARCH
from mrg_slam_amd.registration import default_params
import gc
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 128
scene = synth.street_scene()
poses = synth.arc_trajectory(9)
ctx0 = Context(0)
scans = [distance_filter(synth.synth_lidar(scene, poses[k], "VLP64", synth.BASE_SEED + k), 0.1, 35.0, ctx=ctx0) for k in range(9)]
dev = [torch.from_numpy(s).cuda(0) for s in scans]
rels = [np.linalg.inv(poses[k]) @ poses[k + 1] for k in range(8)]
pairs = [(b % 8, b % 8 + 1, synth.warm_guess(rels[b % 8], b)) for b in range(B)]
prm = default_params(NDT_HIP)
prm.transformation_epsilon, prm.maximum_iterations, prm.resolution, prm.nn_search_method = 0.1, 64, 1.0, SEARCH["DIRECT7"]
ctxs = [Context(0) for _ in range(NS)]
bms = [BatchMatcher(prm, c) for c in ctxs]
def work(i, out):
    bm = bms[i]
    bm.clear()
    for (ti, si, g) in pairs[i::NS]:
        t = bm.add_target_device(dev[ti].data_ptr(), len(scans[ti]))
        bm.add_pair_device(t, dev[si].data_ptr(), len(scans[si]), g)
    out[i] = bm.align()
def step():
    out = [None] * NS
    th = [threading.Thread(target=work, args=(i, out)) for i in range(NS)]
    for t in th: t.start()
    for t in th: t.join()
    return out
for _ in range(3): step()
gc.collect(); gc.freeze()
t0 = time.perf_counter()
for _ in range(8): step()
dt = (time.perf_counter() - t0) / 8
print(f"streams={NS}: {1e3*dt:.2f} ms per 128 pairs, {B/dt:.0f} alignments/s")
