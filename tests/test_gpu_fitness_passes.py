"""GPU: getFitnessScore's passes (block -> seed + sweep -> pyramid walk of the unseeded rest) give the exact nearest distances: equal
to round 2's pyramid walk for every queued query, to the brute-force oracle, and independent of where a query is settled.
Reference: pcl::Registration::getFitnessScore as called at src/mrg_slam/loop_detector.cpp:137 (max_range = inf there)."""
import numpy as np
import pytest

from conftest import small_cloud

pytestmark = pytest.mark.gpu


def _set_sweep(mode):
    from mrg_slam_amd._lib import lib

    return lib().mrgfe_dbg_set_fit_sweep(mode)


@pytest.fixture()
def sweep_mode():
    before = _set_sweep(-1)
    yield
    _set_sweep(before)


def _clouds(seed):
    """A target with dense and sparse regions and sources whose points lie at every distance from it: on it, centimetres off,
    decimetres off (the ring-spacing case of a loop-closure candidate), metres off and far outside its bounding box."""
    rng = np.random.default_rng(seed)
    tgt = small_cloud(60000, seed, extent=(35.0, 20.0, 4.0))
    srcs = []
    base = small_cloud(30000, seed + 1, extent=(35.0, 20.0, 4.0))
    srcs.append(base)
    off = base.copy()
    off[:, 2] += rng.uniform(0.1, 0.9, len(off)).astype(np.float32)  # decimetres above every surface
    srcs.append(off)
    far = base[:8000].copy()
    far[:, :3] += rng.normal(0, 3.0, (len(far), 3)).astype(np.float32)  # metres off, some outside the box
    far[:200, :3] += 200.0
    srcs.append(far)
    srcs.append(base[:1])  # a single query
    return tgt, srcs


@pytest.mark.parametrize("max_range", [float("inf"), 4.0, 0.04])
def test_sweep_equals_pyramid_walk_and_oracle(sweep_mode, max_range):
    from mrg_slam_amd import default_context
    from mrg_slam_amd.filters import calc_fitness_score
    from oracle import oracle as orc

    tgt, srcs = _clouds(5)
    ctx_stats = []
    got = {}
    for mode in (1, 0):
        assert _set_sweep(mode) == mode
        vals = []
        for s in srcs:
            vals.append(calc_fitness_score(tgt, s, np.eye(4), max_range))
        got[mode] = vals
        ctx_stats.append(default_context().fitness_stats())
    assert got[1] == got[0], (got, ctx_stats)  # bit for bit: both are the exact nearest distances summed in the same order
    for s, v in zip(srcs, got[1]):
        exp = orc.calc_fitness_score(tgt, s, np.eye(4), max_range)
        assert v == pytest.approx(exp, rel=1e-12), (v, exp)
    assert ctx_stats[0]["queued"] >= ctx_stats[0]["queued_far"]


def test_batched_jobs_and_sparse_targets(sweep_mode):
    """Many jobs in one launch (ragged sizes), a target so sparse that most seeds come from the super-brick and block levels (some
    queries find none and take the pyramid walk), and a tiny target: every score equals the other mode's and the oracle's."""
    from mrg_slam_amd import BatchMatcher
    from oracle import oracle as orc

    rng = np.random.default_rng(11)
    sparse = np.zeros((400, 4), np.float32)
    sparse[:, :3] = rng.uniform(-40, 40, (400, 3))
    tiny = np.zeros((3, 4), np.float32)
    tiny[:, :3] = [[0, 0, 0], [1, 0, 0], [0, 30, 0]]
    dense = small_cloud(50000, 21, extent=(30.0, 30.0, 3.0))
    targets = [sparse, tiny, dense]
    srcs = [small_cloud(3000 + 777 * k, 40 + k, extent=(30.0, 30.0, 3.0)) for k in range(7)]
    res = {}
    for mode in (1, 0):
        assert _set_sweep(mode) == mode
        bm = BatchMatcher(transformation_epsilon=0.1, maximum_iterations=1)
        tids = [bm.add_target(t) for t in targets]
        for k, s in enumerate(srcs):
            bm.add_pair(tids[k % 3], s, np.eye(4))
        res[mode] = bm.align(float("inf"))
    np.testing.assert_array_equal(res[1]["fitness"], res[0]["fitness"])
    np.testing.assert_array_equal(res[1]["T"], res[0]["T"])
    for k, s in enumerate(srcs):
        T = np.asarray(res[1][k]["T"], np.float32).reshape(4, 4).T.astype(np.float64)
        exp = orc.calc_fitness_score(targets[k % 3], s, T, float("inf"))
        assert res[1][k]["fitness"] == pytest.approx(exp, rel=1e-9)
