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


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 6, 17, 64, 200, 256])
def test_radix_sort_with_m_distinct_digits_per_wavefront(ctx, m):
    """The kernels match the lanes of a wavefront that hold the same digit by peeling the first few distinct digits and fall back to the bit-by-bit
    match for what is left (dev_utils.h wave_match_digit8, wave_hist_add): m distinct digits interleaved lane by lane — below, at and above the peel
    depth — in BOTH passes of a 16-bit sort, plus runs of 70 equal keys that cross wavefront and round boundaries."""
    from mrg_slam_amd._lib import check, lib

    n = 50001
    i = np.arange(n, dtype=np.uint64)
    lo = (i * 3) % m
    hi = ((i // 7) * 5) % m
    keys = (lo | (hi << 8)).astype(np.uint32)
    keys[20000:20070] = keys[20000]
    keys[30000:33000] = np.repeat(keys[30000:30000 + 43], 70)[:3000]
    vals = np.arange(n, dtype=np.uint32)
    ok, ov = np.empty_like(keys), np.empty_like(vals)
    check(lib().mrgfe_dbg_sort_pairs(ctx._h, keys.ctypes.data_as(_u32p), vals.ctypes.data_as(_u32p), n, 16, ok.ctypes.data_as(_u32p), ov.ctypes.data_as(_u32p)))
    order = np.argsort(keys, kind="stable")
    np.testing.assert_array_equal(ok, keys[order])
    np.testing.assert_array_equal(ov, order.astype(np.uint32))


@pytest.mark.parametrize("n,bits", [(1048577, 20), (1150000, 9), (2500003, 31), (6600000, 26)])
def test_radix_sort_of_thousands_of_tiles(ctx, n, bits):
    """More than 512 tiles in one problem (the map cloud's 6.5 M points): the digit counts are scanned chunk by chunk (rs_chunk_sums_kernel,
    rs_scan_chunks_kernel) instead of by one workgroup per problem; sizes just past the switch, with ragged last chunks and a ragged last tile."""
    from mrg_slam_amd._lib import check, lib

    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, min(1 << bits, 1 << 31), n, dtype=np.uint32)
    keys[n // 2: n // 2 + n // 5] = keys[7]
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


def _brute_knn(cloud, q, k):
    """(index, squared distance) of the k nearest points, ascending by (distance, index), float32 arithmetic in the kernel's order"""
    c, qq = cloud[:, :3].astype(np.float32), q[:, :3].astype(np.float32)
    idx = np.full((len(qq), k), -1, dtype=np.int32)
    sqd = np.full((len(qq), k), -1.0, dtype=np.float32)
    fin = np.isfinite(c).all(axis=1)
    ids = np.nonzero(fin)[0]
    for i, p in enumerate(qq):
        if not np.isfinite(p).all() or len(ids) == 0:
            continue
        d = c[ids] - p
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        o = np.lexsort((ids, d2))[:k]
        idx[i, : len(o)] = ids[o]
        sqd[i, : len(o)] = d2[o]
    return idx, sqd


@pytest.mark.parametrize("k", [1, 5, 20])
def test_grid_set_members_answer_like_single_grids(ctx, k):
    """NnGridSet (all members per launch, csrc/nn_grid.hip) against the single-cloud build and against brute force: clouds of very different
    size, extent and density in one set, an empty one, one with non-finite points; the rebuilt set (cell edges reused) answers the same."""
    from mrg_slam_amd import synth
    from mrg_slam_amd.filters import grid_set_query, knn

    rng = np.random.default_rng(100 + k)
    scene = synth.street_scene()
    poses = synth.arc_trajectory(3)
    scan = synth.synth_lidar(scene, poses[0], "VLP16", synth.BASE_SEED)
    scan2 = synth.synth_lidar(scene, poses[2], "VLP16", synth.BASE_SEED + 2)[::3]
    blob = np.zeros((3000, 4), dtype=np.float32)
    blob[:, :3] = rng.normal(0, 0.05, (3000, 3))  # everything inside one coarse cell: the adaptive edge halves four times
    few = np.zeros((7, 4), dtype=np.float32)
    few[:, :3] = rng.uniform(-30, 30, (7, 3))
    holes = scan2.copy()
    holes[::17, 1] = np.nan
    holes[5::29, 0] = np.inf
    line = np.zeros((500, 4), dtype=np.float32)
    line[:, 0] = np.linspace(-100, 100, 500)  # a degenerate extent
    clouds = [scan, blob, np.zeros((0, 4), dtype=np.float32), few, holes, line, scan2]
    q = np.concatenate([scan[::40], blob[::300], few, (rng.uniform(-60, 60, (200, 4))).astype(np.float32)])
    q[3, 2] = np.nan
    for rounds in (1, 3):
        idx, sqd = grid_set_query(clouds, q, k, rounds=rounds, ctx=ctx)
        for m, c in enumerate(clouds):
            bi, bd = _brute_knn(c, q, k)
            if k == 1:
                bi[~np.isfinite(q[:, :3]).all(axis=1)] = -1
            np.testing.assert_array_equal(idx[m], bi, err_msg=f"member {m} rounds {rounds}")
            ok = bi >= 0
            np.testing.assert_array_equal(sqd[m][ok], bd[ok], err_msg=f"member {m} rounds {rounds}")
            if k > 1 and len(c):
                si, sd = knn(c, q, k, ctx=ctx)
                np.testing.assert_array_equal(idx[m], si)
                np.testing.assert_array_equal(sqd[m], sd)


@pytest.mark.parametrize("k", [1, 4])
def test_grid_set_degenerate_sets(ctx, k):
    """A set of one cloud, a set of empty clouds only, a set whose clouds hold nothing but non-finite points, one point per cloud."""
    from mrg_slam_amd.filters import grid_set_query

    rng = np.random.default_rng(7)
    q = rng.uniform(-5, 5, (50, 4)).astype(np.float32)
    one = np.zeros((300, 4), dtype=np.float32)
    one[:, :3] = rng.uniform(-4, 4, (300, 3))
    nan_only = np.full((5, 4), np.nan, dtype=np.float32)
    single = np.zeros((1, 4), dtype=np.float32)
    single[0, :3] = [1.0, 2.0, 3.0]
    for clouds in ([one], [np.zeros((0, 4), dtype=np.float32)] * 3, [nan_only, nan_only], [single, single + 1.0], [one, nan_only, single]):
        idx, sqd = grid_set_query(clouds, q, k, rounds=2, ctx=ctx)
        for m, c in enumerate(clouds):
            bi, bd = _brute_knn(c, q, k)
            np.testing.assert_array_equal(idx[m], bi)
            ok = bi >= 0
            np.testing.assert_array_equal(sqd[m][ok], bd[ok])



def test_glibc_exp_on_the_device_equals_the_host_build():
    """csrc/glibc_exp.h compiled for gfx950 against its host build (which tests/test_glibc_exp.py holds against the C library): bit for bit on a million
    arguments over every branch — the f64 passes of the NDT kernels then weigh every (point, voxel) pair with the double the reference's host computes."""
    import ctypes as C

    from mrg_slam_amd import Context
    from mrg_slam_amd._lib import lib
    from test_glibc_exp import _args

    ctx = Context()
    x = _args(1_000_000, 11)
    dev, host = np.empty_like(x), np.empty_like(x)
    dp = C.POINTER(C.c_double)
    assert lib().mrgfe_dbg_exp(ctx._h, x.ctypes.data_as(dp), len(x), 1, dev.ctypes.data_as(dp)) == 0
    assert lib().mrgfe_dbg_exp(None, x.ctypes.data_as(dp), len(x), 0, host.ctypes.data_as(dp)) == 0
    same = (dev == host) | (np.isnan(dev) & np.isnan(host))
    assert same.all(), [(float(a).hex(), float(b).hex(), float(c).hex()) for a, b, c in zip(x[~same][:5], dev[~same][:5], host[~same][:5])]
