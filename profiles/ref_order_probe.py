import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
from conftest import small_cloud
from mrg_slam_amd import NdtHip, synth
from mrg_slam_amd._lib import lib
from oracle import oracle as orc
lib().mrgfe_dbg_set_ndt_reference_order(1)
for seed in (3, 4):
    tgt = small_cloud(5000, seed)
    rel = synth.make_pose([0.25, -0.1, 0.03], synth.rot_xyz(0.01, -0.008, 0.03))
    src = orc.transform_points(np.linalg.inv(rel), tgt)
    for search in ("DIRECT7", "DIRECT1", "DIRECT26", "KDTREE"):
        g = NdtHip(search=search); o = orc.Ndt(search=search, num_threads=4)
        g.setInputTarget(tgt); o.setInputTarget(tgt); g.setInputSource(src); o.setInputSource(src)
        for p in (np.array([0.2, -0.05, 0.01, 0.012, -0.006, 0.025]), np.zeros(6)):
            T = orc.pose_to_matrix(p)
            for mode in (0, 1, 2):
                gs, gg, gH = g.evaluate(T, p, mode); os_, og, oH = o.evaluate(T, p, mode)
                print(seed, search, "p0" if p.any() else "pz", "mode", mode, "score", gs == os_ if mode != 2 else "-", "grad bad", int((gg != og).sum()) if mode != 2 else "-",
                      "hess bad", int((gH != oH).sum()) if mode != 1 else "-", "maxrel", float(np.max(np.abs(gH - oH) / (np.abs(oH) + 1e-300))) if mode != 1 else "-")
