#!/usr/bin/env python3
"""The randomised parity soak at a size the GPU suite and bench.py do not run every time:
    python3 profiles/soak.py [ndt_cases=2000] [round3_cases=600] [pcl_ndt_cases=500] [reference_order_cases=ndt_cases] > gpurun_out/soak.json
Counts per method of scenes over the 1e-4 m / 1e-4 rad bar against the reference-order oracle, of bit-identical results, and of over-the-bar scenes
that equal the GPU-order replay (oracle/replay.py).  The summary is kept as profiles/<tag>_soak.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (before libmrgfe)

from oracle.replay import ndt_reference_order_soak, ndt_soak, pclndt_soak, round3_soak  # noqa: E402

a_n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
b_n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
c_n = int(sys.argv[3]) if len(sys.argv) > 3 else 500
t0 = time.time()
a = ndt_soak(a_n, 77)
t1 = time.time()
print(f"[soak] {a_n} scenes in {t1 - t0:.0f} s", file=sys.stderr)
b = round3_soak(b_n, 78)
print(f"[soak] {b_n} scenes in {time.time() - t1:.0f} s", file=sys.stderr)
t2 = time.time()
c = pclndt_soak(c_n, 79)
print(f"[soak] {c_n} scenes in {time.time() - t2:.0f} s", file=sys.stderr)
# NDT_HIP with its sums (and Newton solve) in the reference's order (mrgfe_dbg_set_ndt_reference_order): the same kind of scenes, every one must be bit-identical
from mrg_slam_amd._lib import lib  # noqa: E402

d_n = int(sys.argv[4]) if len(sys.argv) > 4 else a_n
t3 = time.time()
lib().mrgfe_dbg_set_ndt_reference_order(1)
d = ndt_reference_order_soak(d_n, 80)
lib().mrgfe_dbg_set_ndt_reference_order(0)
d["seconds"] = time.time() - t3
print(f"[soak] {d_n} reference-order scenes in {time.time() - t3:.0f} s", file=sys.stderr)
print(json.dumps({"all_methods": a, "pcl_gicp_and_reciprocal_icp": b, "pcl_ndt": c, "ndt_reference_order": d, "seconds": time.time() - t0}))
