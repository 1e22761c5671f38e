"""CPU: glibc's double exp() restated in mrg_slam_amd/csrc/glibc_exp.h — what the f64 passes of the NDT kernels call, because the reference computes its
per-pair weights with the host's libm and exp is not correctly rounded (the ROCm device library's result differs from glibc's in the last bit on about one
argument in ten).  Two independent pins:
  * the 128-entry table is regenerated from first principles (2^(i/128) to 90 digits; head = nearest double, tail = 2^(i/128) / head - 1 rounded) and must equal
    the one in the header;
  * the host build of the header (mrgfe_dbg_exp, on_device = 0) equals THIS host's C library (math.exp = libm's exp) on a million arguments over every branch:
    |x| < 2^-54, the main path, 512 <= |x| < 1024 with its subnormal fix-up, overflow / underflow, infinities and NaN."""
import ctypes as C
import math
import os
import re
import struct
from decimal import Decimal, getcontext

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_table_regenerated_from_first_principles():
    getcontext().prec = 90
    ln2 = Decimal(2).ln()
    want = []
    for i in range(128):
        v = (ln2 * Decimal(i) / Decimal(128)).exp()
        head = float(v)
        tail = float(v / Decimal(head) - 1)
        want += [struct.unpack("<Q", struct.pack("<d", tail))[0], struct.unpack("<Q", struct.pack("<d", head))[0] - ((i << 52) // 128)]
    src = open(os.path.join(ROOT, "mrg_slam_amd", "csrc", "glibc_exp.h")).read()
    body = src[src.index("kGlibcExpTab[256] = {"):]
    got = [int(h, 16) for h in re.findall(r"0x([0-9a-f]{16})ULL", body[: body.index("};")])]
    assert len(got) == 256 and got == want


def _args(n, seed):
    rng = np.random.default_rng(seed)
    u = rng.random(n)
    x = np.concatenate([-u[: n // 8] * 1e-3, -760 * u[n // 8: n // 4], 720 * u[n // 4: 3 * n // 8], -60 * u[3 * n // 8: n // 2] ** 2, -u[n // 2: 5 * n // 8] * 1e-17,
                        rng.integers(0, 1 << 63, n // 8).astype(np.uint64).view(np.float64) * rng.choice([-1.0, 1.0], n // 8), -30 * u[3 * n // 4: 7 * n // 8], -2 * u[7 * n // 8:]])
    return np.ascontiguousarray(np.concatenate([x, [0.0, -0.0, np.inf, -np.inf, np.nan, 709.782712893384, 709.7827128933841, -745.1332191019411, -745.1332191019412, -708.3964185322641,
                                                    512.0, -512.0, 1024.0, -1024.0, 2.0 ** -54, -(2.0 ** -54), 1e-320, -1e-320]]))


def test_host_build_equals_this_hosts_libm():
    from mrg_slam_amd import _lib

    x = _args(1_000_000, 7)
    out = np.empty_like(x)
    dp = C.POINTER(C.c_double)
    assert _lib.lib().mrgfe_dbg_exp(None, x.ctypes.data_as(dp), len(x), 0, out.ctypes.data_as(dp)) == 0

    def libm(v):
        try:
            return math.exp(v)
        except OverflowError:
            return math.inf

    want = np.array([libm(float(v)) for v in x])
    same = (out == want) | (np.isnan(out) & np.isnan(want))
    assert same.all(), [(float(a).hex(), float(b).hex(), float(c).hex()) for a, b, c in zip(x[~same][:5], out[~same][:5], want[~same][:5])]
