"""CPU: pin the prefilter / nearest-neighbour oracle against brute-force numpy restatements."""
import numpy as np
import pytest

from conftest import small_cloud
from oracle import oracle as orc


def test_distance_filter_matches_numpy():
    c = small_cloud(5000, 3, extent=(45, 45, 3))
    c[:5, :3] = 0.01  # too near
    out = orc.distance_filter(c, 0.1, 35.0)
    d = np.sqrt((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2]).astype(np.float64)
    np.testing.assert_array_equal(out, c[(d > 0.1) & (d < 35.0)])
    assert len(orc.distance_filter(np.zeros((0, 4), np.float32))) == 0


@pytest.mark.parametrize("leaf,min_pts", [(0.1, 1), (0.5, 1), (0.5, 3), (2.0, 1)])
def test_voxelgrid_matches_dict_bruteforce(leaf, min_pts):
    c = small_cloud(4000, 5)
    out, status = orc.voxelgrid(c, leaf, min_pts, orc.ORDER_STABLE)
    assert status == 0
    inv = np.float32(1.0) / np.float32(leaf)
    ijk = np.floor(c[:, :3] * inv).astype(np.int64)
    mn, mx = ijk.min(0), ijk.max(0)
    div = mx - mn + 1
    lin = (ijk - mn) @ np.array([1, div[0], div[0] * div[1]])
    cells = {}
    for i, k in enumerate(lin):
        cells.setdefault(int(k), []).append(i)
    exp = []
    for k in sorted(cells):
        idx = cells[k]
        if len(idx) < min_pts:
            continue
        acc = np.zeros(4, dtype=np.float32)
        for i in idx:
            acc = acc + c[i]  # float32 running sum in point order
        exp.append(acc / np.float32(len(idx)))
    exp = np.asarray(exp, dtype=np.float32).reshape(-1, 4)
    np.testing.assert_array_equal(out, exp)
    # std::sort order (PCL's) may permute the additions inside a voxel: same voxels, centroids within float rounding
    out2, _ = orc.voxelgrid(c, leaf, min_pts, orc.ORDER_STD_SORT)
    assert out2.shape == out.shape
    np.testing.assert_allclose(out2, out, rtol=0, atol=2e-5)


def test_voxelgrid_overflow_returns_input():
    c = small_cloud(100)
    c[0, :3] = [3e5, -3e5, 3e5]
    out, status = orc.voxelgrid(c, 0.1, 1)
    assert status == 1
    np.testing.assert_array_equal(out, c)
    out, status = orc.voxelgrid(np.zeros((0, 4), np.float32), 0.1, 1)
    assert len(out) == 0


def _sqd(a, b):
    d = a[:, None, :3] - b[None, :, :3]
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]  # float32, FLANN L2_Simple order


def test_radius_outlier_matches_bruteforce():
    c = small_cloud(1500, 7)
    c[:20, :3] += 100.0  # isolated points
    out, keep = orc.radius_outlier(c, 0.5, 2)
    cnt = (_sqd(c, c).astype(np.float64) <= 0.25).sum(1)  # includes the point itself
    np.testing.assert_array_equal(keep, cnt >= 3)
    np.testing.assert_array_equal(out, c[keep])
    # lattice with spacing exactly r: neighbours at sqdist == r^2 count (inclusive compare)
    g = np.zeros((27, 4), np.float32)
    g[:, :3] = np.stack(np.meshgrid(*[np.arange(3) * 0.5] * 3, indexing="ij"), -1).reshape(-1, 3)
    _, keep = orc.radius_outlier(g, 0.5, 3)
    assert keep.all()  # corner points have exactly 3 neighbours at distance r, plus themselves


def test_knn_matches_bruteforce():
    t, q = small_cloud(2500, 11), small_cloud(300, 12, extent=(30, 20, 5))
    idx, sqd = orc.knn(t, q, 5)
    D = _sqd(q, t)
    order = np.lexsort((np.broadcast_to(np.arange(D.shape[1]), D.shape), D), axis=1)[:, :5]
    np.testing.assert_array_equal(idx, order)
    np.testing.assert_array_equal(sqd, np.take_along_axis(D, order, 1))
    bi, bd = orc.nn1_brute(t, q)
    np.testing.assert_array_equal(bi, idx[:, 0])
    np.testing.assert_array_equal(bd, sqd[:, 0])


def test_statistical_outlier_matches_numpy():
    c = small_cloud(1200, 13)
    c[:10, :3] += 60.0
    k = 10
    out, keep = orc.statistical_outlier(c, k, 1.2)
    D = np.sort(_sqd(c, c), axis=1)[:, 1 : k + 1]
    dist = (np.sqrt(D).astype(np.float64).sum(1) / k).astype(np.float32)
    s, ss = dist.astype(np.float64).sum(), (dist * dist).astype(np.float64).sum()
    n = len(c)
    thr = s / n + 1.2 * np.sqrt((ss - s * s / n) / (n - 1))
    np.testing.assert_array_equal(keep, ~(dist > thr))
    assert not keep[:10].any() and keep.sum() > 0.9 * n
    np.testing.assert_array_equal(out, c[keep])


def test_fitness_score_semantics():
    t = small_cloud(2000, 21)
    rel = np.eye(4)
    rel[:3, 3] = [0.05, 0.0, 0.0]
    s = orc.calc_fitness_score(t, t, rel)
    _, d = orc.nn1_brute(t, orc.transform_points(rel, t))
    assert s == pytest.approx(d.astype(np.float64).mean(), rel=1e-12)
    # squared distance is compared with the UN-squared max_range (reference quirk)
    s2 = orc.calc_fitness_score(t, t, rel, max_range=0.001)
    sel = d.astype(np.float64) <= 0.001
    assert s2 == pytest.approx(d[sel].astype(np.float64).mean(), rel=1e-12)
    assert orc.calc_fitness_score(t, t, np.diag([1, 1, 1, 1.0]) + np.eye(4, k=3) * 500, max_range=1.0) == np.finfo(np.float64).max


def test_information_matrix_weights_against_numpy():
    """InformationMatrixCalculator::weight / calc_information_matrix (information_matrix_calculator.cpp:14-44,83-88): the oracle's
    restatement and the product's host arithmetic (mrgfe_inf_weight / mrgfe_inf_matrix_from_fitness, no GPU involved) against the
    formula written out in numpy."""
    import ctypes as C

    from mrg_slam_amd import _lib
    from mrg_slam_amd._lib import lib

    rng = np.random.default_rng(0)
    p = _lib.InfParams()
    lib().mrgfe_inf_default_params(C.byref(p))
    assert (p.use_const_inf_matrix, p.const_stddev_x, p.const_stddev_q, p.var_gain_a, p.min_stddev_x, p.max_stddev_x, p.min_stddev_q, p.max_stddev_q,
            p.fitness_score_thresh) == (0, 0.5, 0.1, 2.0, 0.1, 0.75, 0.05, 0.2, 1.25)
    for fit in [0.0, 1e-3, 0.2, 1.25, 3.0, np.finfo(np.float64).max] + list(rng.uniform(0, 2, 20)):
        w = {}
        for tag, (lo, hi) in {"x": (0.1, 0.75), "q": (0.05, 0.2)}.items():
            y = (1.0 - np.exp(-2.0 * fit)) / (1.0 - np.exp(-2.0 * 1.25))
            w[tag] = lo ** 2 + (hi ** 2 - lo ** 2) * y
            wo = orc.inf_weight(2.0, 1.25, lo ** 2, hi ** 2, fit)
            assert wo == pytest.approx(w[tag], rel=1e-14)  # numpy's exp and the C library's differ in the last bit
            assert lib().mrgfe_inf_weight(2.0, 1.25, lo ** 2, hi ** 2, fit) == wo  # oracle and product: the same C library
            w[tag] = wo
        exp = np.diag([1 / w["x"]] * 3 + [1 / w["q"]] * 3)
        inf = np.empty((6, 6))
        assert lib().mrgfe_inf_matrix_from_fitness(C.byref(p), fit, inf.ctypes.data_as(C.POINTER(C.c_double))) == 0
        np.testing.assert_array_equal(inf, exp)
        v = np.array([0, 0.5, 0.1, 2.0, 0.1, 0.75, 0.05, 0.2, 1.25])
        o_inf = np.empty((6, 6))
        orc.lib().orc_inf_matrix(v.ctypes.data_as(C.POINTER(C.c_double)), float(fit), o_inf.ctypes.data_as(C.POINTER(C.c_double)))
        np.testing.assert_array_equal(o_inf, inf)
    # constant matrix: the reference divides by the standard deviation itself (:21-22)
    p.use_const_inf_matrix = 1
    inf = np.empty((6, 6))
    lib().mrgfe_inf_matrix_from_fitness(C.byref(p), 123.0, inf.ctypes.data_as(C.POINTER(C.c_double)))
    np.testing.assert_array_equal(inf, np.diag([2.0] * 3 + [10.0] * 3))
    o_inf, o_fit = orc.calc_information_matrix(np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32), np.eye(4), {"use_const_inf_matrix": True})
    np.testing.assert_array_equal(o_inf, inf)


def _approx_voxelgrid_py(c, leaf):
    """pcl::ApproximateVoxelGrid as a plain Python loop over a dict-backed 512-entry history (independent restatement of SURVEY.md A.1)."""
    inv = np.float32(1.0) / np.float32(leaf)
    hist, out = {}, []
    for p in c:
        ijk = tuple(int(np.floor(np.float32(p[k]) * inv)) for k in range(3))
        h = (ijk[0] * 7171 + ijk[1] * 3079 + ijk[2] * 4231) & 511
        e = hist.get(h)
        if e is not None and e[0] != ijk:
            out.append(e[1] / np.float32(e[2]))
            e = None
        if e is None:
            e = [ijk, np.zeros(4, np.float32), 0]
        e[1] = e[1] + p
        e[2] += 1
        hist[h] = e
    for h in sorted(hist):
        out.append(hist[h][1] / np.float32(hist[h][2]))
    return np.array(out, dtype=np.float32).reshape(-1, 4)


@pytest.mark.parametrize("leaf", [0.05, 0.3, 2.0, 50.0])
def test_approx_voxelgrid_oracle_matches_the_plain_loop(leaf):
    from oracle import oracle as orc

    c = small_cloud(3000, 21, extent=(25.0, 15.0, 4.0))
    np.testing.assert_array_equal(orc.approx_voxelgrid(c, leaf), _approx_voxelgrid_py(c, leaf))
    # order dependent: another arrival order gives another cloud (as many or more points: a cell can be flushed more than once)
    shuffled = c[np.random.default_rng(3).permutation(len(c))]
    a, b = orc.approx_voxelgrid(c, leaf), orc.approx_voxelgrid(shuffled, leaf)
    np.testing.assert_array_equal(b, _approx_voxelgrid_py(shuffled, leaf))
    assert leaf >= 50.0 or not np.array_equal(a, b)
    assert len(orc.approx_voxelgrid(np.zeros((0, 4), np.float32), leaf)) == 0
    one = orc.approx_voxelgrid(c[:1], leaf)
    np.testing.assert_array_equal(one, c[:1])
