import time, numpy as np, sys
sys.path.insert(0, ".")
import torch
from mrg_slam_amd import PclGicpHip, distance_filter, prefilter, synth
from mrg_slam_amd._lib import lib
sc = synth.street_scene()
t, s, rel = synth.scan_pair(0, "VLP64", sc)
for name, f in (("130k", lambda c: distance_filter(c, 0.1, 35.0)), ("33k", prefilter)):
    a, b = f(t), f(s)
    for mode in (1, 0):
        lib().mrgfe_dbg_set_pclgicp_reference_order(mode)
        g = PclGicpHip(transformation_epsilon=0.01)
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); g.setInputTarget(a); g.setInputSource(b); g.align(synth.warm_guess(rel, 0)); ts.append(1e3 * (time.perf_counter() - t0))
        print(name, len(a), "reference order" if mode else "tree", "ms", round(min(ts), 2), "iterations", g.getFinalNumIteration(), "evaluations", g.evals)
