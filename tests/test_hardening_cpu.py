"""CPU hardening (the GPU pool offers no sanitizers): the product's host-only arithmetic and the CPU oracle under AddressSanitizer +
UndefinedBehaviorSanitizer, and every C-ABI entry point with NULL handles / pointers and zero sizes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")


def _build_and_run(tmp_path, name, args, timeout=600):
    exe = str(tmp_path / name)
    b = subprocess.run(["g++"] + SAN + args + ["-o", exe], capture_output=True, text=True, timeout=timeout)
    assert b.returncode == 0, b.stderr[-4000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=timeout, env=ENV)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-6000:]
    return r.stdout


def test_optimiser_and_bfgs_are_clean_under_asan_ubsan(tmp_path):
    """csrc/ndt_ctl.h (the NDT state machine: both formulations, three epsilons, NaN / zero-gradient / rank-deficient evaluations, LU against SVD,
    the float sine / cosine) and csrc/bfgs.h, compiled for the host with g++ and both sanitizers"""
    out = _build_and_run(tmp_path, "host_san", ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "mrg_slam_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                                                os.path.join(ROOT, "tests", "sanitize", "host_paths_san.cpp")])
    assert "0 failed checks" in out


def test_oracle_is_clean_under_asan_ubsan(tmp_path):
    """every restated algorithm of oracle/ once on a small cloud (plus empty inputs), sources linked straight into the sanitized program"""
    src = [os.path.join(ROOT, "oracle", f) for f in ("ndt.cpp", "pcl_ndt.cpp", "filters.cpp", "mapcloud.cpp", "gicp.cpp", "pcl_gicp.cpp")]
    out = _build_and_run(tmp_path, "oracle_san", ["-fopenmp", "-mfma", "-ffp-contract=off", "-I" + os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "sanitize", "oracle_san.cpp")] + src)
    assert "done" in out


def test_every_entry_point_survives_null_arguments():
    """NULL handles, NULL pointers, zero sizes into every function include/mrgfe.h declares: an error code (or a harmless default), never a crash"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "workers", "null_args_worker.py")], capture_output=True, text=True, timeout=300)
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("calling")]
    assert r.returncode == 0, (last[-1] if last else None, r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.strip().endswith("done") and len(last) >= 120
