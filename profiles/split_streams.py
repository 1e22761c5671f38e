"""Experiment: the 256 pairs of a bench step as S independent batches on S contexts (streams), aligned by S host threads at once.
What it measures: how much of a lock-step round's serial tail (controller kernel, plan kernel, queue gaps — the GPU is idle behind one
wavefront per pair) another batch's derivative kernels can fill.   python3 profiles/split_streams.py [S ...]"""
import os, sys, threading, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

B = int(os.environ.get("SPLIT_BATCH", "256"))
scene, poses, raw = bench.make_workload(B, B, 0, "distance")
import torch  # noqa: E402
from mrg_slam_amd import BatchMatcher, Context, distance_filter, synth  # noqa: E402
from mrg_slam_amd._lib import NDT_HIP, SEARCH  # noqa: E402
from mrg_slam_amd.registration import default_params  # noqa: E402

ctx0 = Context(0)
host = [distance_filter(s, 0.1, 35.0, ctx=ctx0) for s in raw]
dev = [torch.from_numpy(s).to("cuda:0") for s in host]
rels = [np.linalg.inv(poses[k]) @ poses[k + 1] for k in range(B)]
guess = [np.eye(4) if b % 4 == 3 else synth.warm_guess(rels[b], b) for b in range(B)]
prm = default_params(NDT_HIP)
prm.transformation_epsilon = 0.1
prm.maximum_iterations = 64
prm.resolution = 1.0
prm.nn_search_method = SEARCH["DIRECT7"]

for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    ctxs = [ctx0] + [Context(0) for _ in range(S - 1)]
    bms = [BatchMatcher(prm, c) for c in ctxs]
    parts = [list(range(s, B, S)) for s in range(S)]  # interleaved: warm and cold guesses in every part
    args = [([dev[k].data_ptr() for k in p], [len(host[k]) for k in p], np.arange(len(p), dtype=np.int32), [dev[k + 1].data_ptr() for k in p], [len(host[k + 1]) for k in p],
             np.stack([guess[k] for k in p])) for p in parts]

    def one(s):
        bms[s].clear()
        bms[s].add_device(*args[s])
        return bms[s].align()

    def step():
        if S == 1:
            return [one(0)]
        out = [None] * S
        th = [threading.Thread(target=lambda s=s: out.__setitem__(s, one(s))) for s in range(S)]
        for t in th: t.start()
        for t in th: t.join()
        return out

    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 8
    for _ in range(n): res = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"{S} batch(es) of {B // S} pairs side by side: {ms:.3f} ms per {B} pairs = {B / ms * 1e3:.0f} alignments/s, converged {sum(int(r['converged'].sum()) for r in res)}", flush=True)
