"""Calls every entry point of include/mrgfe.h with NULL handles / NULL pointers / zero sizes (argument types from the binding table): each must
come back — with an error code where it has one — instead of crashing.  Prints the name before every call so that a crash names its function.
Runs without a GPU (nothing gets as far as HIP); tests/test_hardening_cpu.py starts it as a child process."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MRGFE_NO_TORCH", "1")
from mrg_slam_amd import _lib  # noqa: E402

L = _lib.lib()
skip = {"mrgfe_last_error", "mrgfe_version"}
bad = []
for name, (res, args) in sorted({**_lib.SIGNATURES, **_lib.DEBUG_SIGNATURES}.items()):
    if name in skip:
        continue
    print("calling", name, flush=True)
    vals = []
    for a in args:
        if a in (C.c_int, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_size_t, C.c_long, C.c_uint8):
            vals.append(a(0))
        elif a in (C.c_double, C.c_float):
            vals.append(a(0.0))
        else:
            vals.append(None)  # every pointer NULL
    r = getattr(L, name)(*vals)
    if res is C.c_int and name.startswith(("mrgfe_reg_", "mrgfe_batch_", "mrgfe_node_", "mrgfe_map_store_", "mrgfe_ctx_")) and not name.endswith(("_has_converged", "_iterations", "_evaluations", "_num_pairs", "_num_members", "_has_cloud", "_has", "_rounds", "_last_gather", "_select_best")):
        if r >= 0:
            bad.append((name, r))
print("done", flush=True)
if bad:
    print("accepted NULL handles:", bad)
    sys.exit(3)
