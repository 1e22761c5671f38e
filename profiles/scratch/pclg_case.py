import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_soak import _scene
from mrg_slam_amd import PclGicpHip, IcpHip, synth
from oracle import oracle as orc
rng = np.random.default_rng(29)
for c in range(60):
    tgt, src, guess, eps = _scene(rng)
    kind = rng.random()
    if kind >= 0.7:
        continue
    omp = kind >= 0.4
    res = {}
    for name, mk in (("gpu", lambda: PclGicpHip(transformation_epsilon=eps, omp=omp)), ("o1", lambda: orc.PclGicp(transformation_epsilon=eps, omp=omp, num_threads=1)),
                     ("o8", lambda: orc.PclGicp(transformation_epsilon=eps, omp=omp, num_threads=8)), ("og", lambda: orc.PclGicp(transformation_epsilon=eps, omp=omp, num_threads=1, gpu_order=True))):
        r = mk(); r.setInputTarget(tgt); r.setInputSource(src); r.align(guess)
        res[name] = (r.getFinalTransformation().astype(np.float64), r.getFinalNumIteration(), bool(r.hasConverged()))
    d = lambda a, b: float(np.linalg.norm(res[a][0][:3, 3] - res[b][0][:3, 3]))
    print(c, "omp" if omp else "pcl", eps, "iters", [res[k][1] for k in res], "gpu-o8 %.2e gpu-o1 %.2e o1-o8 %.2e gpu-og %.2e" % (d("gpu", "o8"), d("gpu", "o1"), d("o1", "o8"), d("gpu", "og")), "exact" if np.array_equal(res["gpu"][0], res["og"][0]) else "DIFF", flush=True)
