import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
from conftest import small_cloud
from mrg_slam_amd import NdtHip, synth
from mrg_slam_amd._lib import lib
from oracle import oracle as orc
lib().mrgfe_dbg_set_ndt_reference_order(1)
tgt = small_cloud(5000, 3)
rel = synth.make_pose([0.25, -0.1, 0.03], synth.rot_xyz(0.01, -0.008, 0.03))
src = orc.transform_points(np.linalg.inv(rel), tgt)
src[:, :3] += np.random.default_rng(4).normal(0, 0.01, (len(src), 3)).astype(np.float32)
for res in (1.0, 2.0):
    g = NdtHip(search="DIRECT7", resolution=res); o = orc.Ndt(search="DIRECT7", num_threads=4, resolution=res)
    g.setInputTarget(tgt); o.setInputTarget(tgt); g.setInputSource(src); o.setInputSource(src)
    for p in (np.array([0.2, -0.05, 0.01, 0.012, -0.006, 0.025]), np.zeros(6), np.array([-0.4, 0.3, 0.05, -0.02, 0.03, -0.1])):
        T = orc.pose_to_matrix(p)
        for mode in (0, 1, 2):
            gs, gg, gH = g.evaluate(T, p, mode); os_, og, oH = o.evaluate(T, p, mode)
            if np.isnan(gH).any(): print("NaN entries", np.argwhere(np.isnan(gH)).tolist())
            print(res, "mode", mode, "score", gs, os_, "grad nan", int(np.isnan(gg).sum()), "H nan", int(np.isnan(gH).sum()), "H bad", int((gH != oH).sum()), "grad bad", int((gg != og).sum()))
