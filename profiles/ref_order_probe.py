"""Diagnostic: the product's NDT optimiser stepped by hand (mrgfe_dbg_ctl_*) on saved scenes (tests/golden/reforder_case_*.npz); every request is evaluated by the GPU in
reference order AND by the reference-order oracle, compared bit for bit; the oracle's answer is fed back."""
import sys, os, glob
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import ctypes as C
import numpy as np
import torch
from mrg_slam_amd import NdtHip
from mrg_slam_amd._lib import lib, check, NDT_HIP, SEARCH
from mrg_slam_amd.registration import default_params
from oracle import oracle as orc
_fp, _dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
lib().mrgfe_dbg_set_ndt_reference_order(1)
for f in sorted(glob.glob("tests/golden/reforder_case_*.npz")):
    z = np.load(f)
    tgt, src, guess, eps, res, search = z["tgt"], z["src"], z["guess"], float(z["eps"]), float(z["res"]), str(z["search"])
    g = NdtHip(resolution=res, transformation_epsilon=eps, maximum_iterations=64, search=search); o = orc.Ndt(resolution=res, num_threads=8, transformation_epsilon=eps, maximum_iterations=64, search=search)
    g.setInputTarget(tgt); o.setInputTarget(tgt); g.setInputSource(src); o.setInputSource(src)
    prm = default_params(NDT_HIP); prm.transformation_epsilon = eps; prm.maximum_iterations = 64; prm.resolution = res; prm.nn_search_method = SEARCH[search]
    h = C.c_void_p()
    gc = np.ascontiguousarray(np.asarray(guess, dtype=np.float32).T)
    check(lib().mrgfe_dbg_ctl_create(C.byref(prm), gc.ctypes.data_as(_fp), len(src), C.byref(h)))
    mode, Tc, p = C.c_int(0), np.empty((4, 4), dtype=np.float32), np.empty(6)
    k = 0
    while lib().mrgfe_dbg_ctl_request(h, C.byref(mode), Tc.ctypes.data_as(_fp), p.ctypes.data_as(_dp)):
        T = Tc.T.copy()
        gs, gg, gH = g.evaluate(T, p, mode.value)
        os_, og, oH = o.evaluate(T, p, mode.value)
        bad = (mode.value != 2 and (gs != os_ or (gg != og).any())) or (mode.value != 1 and (gH != oH).any())
        if bad:
            print(f, "request", k, "mode", mode.value, "score", gs == os_, gs - os_, "grad bad", int((gg != og).sum()), "H bad", int((gH != oH).sum()), "maxrel H", float(np.max(np.abs(gH - oH) / (np.abs(oH) + 1e-300))))
        check(lib().mrgfe_dbg_ctl_result(h, os_, np.ascontiguousarray(og).ctypes.data_as(_dp), np.ascontiguousarray(oH).ctypes.data_as(_dp), 0.0))
        k += 1
    g.align(guess); o.align(guess)
    print(f, "requests", k, "align: T same", np.array_equal(g.getFinalTransformation(), o.getFinalTransformation()), g.getFinalNumIteration(), o.getFinalNumIteration())
