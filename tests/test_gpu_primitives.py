"""GPU: the batched device primitives under every grid structure (cellsort.hip), checked bit-exactly against numpy."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_u32p = C.POINTER(C.c_uint32)
_fp = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def ctx():
    from mrg_slam_amd import default_context

    return default_context()


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 2047, 2048, 2049, 4096, 10000, 131072, 250001])
@pytest.mark.parametrize("bits", [1, 7, 8, 9, 17, 24, 31])
def test_radix_sort_is_a_stable_sort(ctx, n, bits):
    from mrg_slam_amd._lib import check, lib

    rng = np.random.default_rng(n * 37 + bits)
    hi = min(1 << bits, 1 << 31)
    keys = rng.integers(0, hi, n, dtype=np.uint32)
    if n > 10:
        keys[: n // 3] = keys[0]  # long runs of equal keys stress the stability logic
    vals = np.arange(n, dtype=np.uint32)
    ok, ov = np.empty_like(keys), np.empty_like(vals)
    check(lib().mrgfe_dbg_sort_pairs(ctx._h, keys.ctypes.data_as(_u32p), vals.ctypes.data_as(_u32p), n, bits, ok.ctypes.data_as(_u32p), ov.ctypes.data_as(_u32p)))
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(ok, keys[order])
    np.testing.assert_array_equal(ov, order.astype(np.uint32))


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 2047, 2048, 2049, 100000, 600001])
def test_exclusive_scan(ctx, n):
    from mrg_slam_amd._lib import check, lib

    rng = np.random.default_rng(n + 1)
    a = rng.integers(0, 5, n, dtype=np.uint32)
    out = np.empty_like(a)
    tot = C.c_uint32(123)
    check(lib().mrgfe_dbg_exclusive_scan(ctx._h, a.ctypes.data_as(_u32p), n, out.ctypes.data_as(_u32p), C.byref(tot)))
    exp = np.concatenate([[0], np.cumsum(a, dtype=np.uint64)[:-1]]).astype(np.uint32) if n else a
    np.testing.assert_array_equal(out, exp)
    assert tot.value == int(a.sum())


@pytest.mark.parametrize("n", [1, 100, 2048, 5000, 131072])
def test_bounding_box_skips_non_finite_points(ctx, n):
    from mrg_slam_amd._lib import check, lib

    rng = np.random.default_rng(n)
    c = rng.normal(0, 30, (n, 4)).astype(np.float32)
    if n >= 100:
        c[3, 0] = np.nan
        c[17, 1] = np.inf
        c[55, 2] = -np.inf
    mn, mx = np.empty(3, np.float32), np.empty(3, np.float32)
    cnt = C.c_uint32(0)
    check(lib().mrgfe_dbg_minmax(ctx._h, c.ctypes.data_as(_fp), n, mn.ctypes.data_as(_fp), mx.ctypes.data_as(_fp), C.byref(cnt)))
    fin = np.isfinite(c[:, :3]).all(1)
    assert cnt.value == fin.sum()
    np.testing.assert_array_equal(mn, c[fin, :3].min(0))
    np.testing.assert_array_equal(mx, c[fin, :3].max(0))


@pytest.mark.parametrize("n_vals", [44, 37, 1])
def test_folded_wave_reduction_equals_the_plain_one(n_vals):
    """wave_sum_fold (the derivative kernels' epilogue: N values per lane folded in six shuffle steps, N / 2 values moved per step)
    adds every value in wave_sum's tree — (lane i) + (lane i + 32), + 16, ... — so its totals are the same doubles as N separate
    wave_sum calls, whatever the magnitudes; and both equal that tree evaluated in numpy."""
    from mrg_slam_amd import Context
    from mrg_slam_amd._lib import check, lib

    ctx = Context(0)
    rng = np.random.default_rng(5)
    cases = 64
    x = rng.normal(0, 1, (cases, 64, n_vals)) * 10.0 ** rng.integers(-12, 12, (cases, 64, n_vals))
    x[0] = 0.0
    x[1, ::2] *= -1.0
    x = np.ascontiguousarray(x)
    fold, plain = np.empty((cases, n_vals)), np.empty((cases, n_vals))
    dp = C.POINTER(C.c_double)
    check(lib().mrgfe_dbg_wave_sums(ctx._h, n_vals, x.ctypes.data_as(dp), cases, fold.ctypes.data_as(dp), plain.ctypes.data_as(dp)))
    assert np.array_equal(fold, plain)
    v = x.copy()
    off = 32
    while off:
        v[:, :off] = v[:, :off] + v[:, off:2 * off]
        off //= 2
    assert np.array_equal(plain, v[:, 0])
