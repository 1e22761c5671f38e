"""Diagnostic: the product's NDT optimiser stepped by hand (mrgfe_dbg_ctl_*); every request is evaluated by the GPU in reference order AND by the
reference-order oracle and the two are compared bit for bit; the oracle's answer is fed back."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ctypes as C
import numpy as np
import torch
from conftest import small_cloud
from mrg_slam_amd import NdtHip, synth
from mrg_slam_amd._lib import lib, check, NDT_HIP
from mrg_slam_amd.registration import default_params
from oracle import oracle as orc
_fp, _dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
lib().mrgfe_dbg_set_ndt_reference_order(1)
tgt = small_cloud(6000, 5)
rel = synth.make_pose([0.25, -0.1, 0.03], synth.rot_xyz(0.01, -0.008, 0.03))
src = orc.transform_points(np.linalg.inv(rel), tgt)
src[:, :3] += np.random.default_rng(6).normal(0, 0.01, (len(src), 3)).astype(np.float32)
eps = 0.1
g = NdtHip(transformation_epsilon=eps, maximum_iterations=64); o = orc.Ndt(num_threads=4, transformation_epsilon=eps, maximum_iterations=64)
g.setInputTarget(tgt); o.setInputTarget(tgt); g.setInputSource(src); o.setInputSource(src)
prm = default_params(NDT_HIP); prm.transformation_epsilon = eps; prm.maximum_iterations = 64
h = C.c_void_p()
guess = np.ascontiguousarray(np.eye(4, dtype=np.float32))
check(lib().mrgfe_dbg_ctl_create(C.byref(prm), guess.ctypes.data_as(_fp), len(src), C.byref(h)))
mode, Tc, p = C.c_int(0), np.empty((4, 4), dtype=np.float32), np.empty(6)
k = 0
while lib().mrgfe_dbg_ctl_request(h, C.byref(mode), Tc.ctypes.data_as(_fp), p.ctypes.data_as(_dp)):
    T = Tc.T.copy()
    gs, gg, gH = g.evaluate(T, p, mode.value)
    os_, og, oH = o.evaluate(T, p, mode.value)
    print("request", k, "mode", mode.value, "p", p, "score same", gs == os_, "grad bad", int((gg != og).sum()), "H bad", int((gH != oH).sum()))
    check(lib().mrgfe_dbg_ctl_result(h, os_, np.ascontiguousarray(og).ctypes.data_as(_dp), np.ascontiguousarray(oH).ctypes.data_as(_dp), 0.0))
    k += 1
g.align(np.eye(4)); o.align(np.eye(4))
print("align: T same", np.array_equal(g.getFinalTransformation(), o.getFinalTransformation()), "H bad", int((g.getHessian() != o.getHessian()).sum()), "vs last oracle evaluation", int((o.getHessian() != oH).sum()), int((g.getHessian() != gH).sum()))
